set -o pipefail
R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/layer_bench.py --dtype bf16 --length 5000 --reps 3 --blocks 3"
rm -rf $O/pb1 $O/pb2 $O/pb3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pb1 -- $CMD > /dev/null 2> $O/pb1.err || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pb2 -- $CMD > /dev/null 2> $O/pb2.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pb3 -- $CMD > /dev/null 2> $O/pb3.err || exit 1
for d in pb1 pb2 pb3; do python3 $R/tools/pmc_summary.py $O/$d conv1d; done
