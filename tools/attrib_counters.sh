#!/bin/bash
# Binding-resource attribution of one train step's entry points from SQ counters (verdict r04 item 1):
#   bash tools/attrib_counters.sh <tag> [pmc_step.py arguments, default: configs[4] in bf16 mode]
# Five separate rocprofv3 --pmc passes of tools/pmc_step.py (counters only, with --kernel-trace; 8 SQ slots per pass):
#   A  issue / wait split of wave time + matrix pipe busy      B  LDS array + instruction mix      C  memory-side
#   D  FETCH_SIZE      E  WRITE_SIZE   (PASSES="D E" runs a subset)
# tools/attrib_table.py turns gpurun_out/attrib_<tag>_{A,B,C} into the factored table committed under profiles/.
set -o pipefail
TAG=${1:-r05}; shift
ARGS=${@:---labels 1 --length 5000 --dtype bf16}
R=$PWD; O=$R/gpurun_out; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/attrib_${TAG}_counters_available.txt" 2>&1 || true
have() { grep -qw "$1" "$O/attrib_${TAG}_counters_available.txt"; }
pick() { local out=""; for c in "$@"; do have "$c" && out="$out $c"; done; echo $out; }
A=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE)
B=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE)
C=$(pick SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE)
D="FETCH_SIZE"
E="WRITE_SIZE"
S="$R/tools/pmc_step.py"
for P in ${PASSES:-A B C D E}; do
  eval "CN=\$$P"
  echo "pass $P: $CN"
  rm -rf "$O/attrib_${TAG}_$P"
  rocprofv3 --pmc $CN --kernel-trace --output-format csv -d "$O/attrib_${TAG}_$P" -- python3 $S $ARGS --log "$O/attrib_${TAG}_$P.order.json" > "$O/attrib_${TAG}_$P.out" 2> "$O/attrib_${TAG}_$P.err" || { tail -5 "$O/attrib_${TAG}_$P.err"; exit 1; }
done
echo "attrib passes done"
