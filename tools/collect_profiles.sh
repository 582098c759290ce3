#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r05 [quick]
# Writes under gpurun_out/: the bench line, a rocprofv3 --kernel-trace --stats run of bench.py, and three
# separate --pmc passes of tools/pmc_step.py (FETCH_SIZE / WRITE_SIZE / MFMA-busy; counters are never combined
# with other trace domains).  tools/make_profile_artifacts.py turns them into the files committed under profiles/.
set -o pipefail
TAG=${1:-r05}
R=$PWD
O=$R/gpurun_out
mkdir -p "$O"
rm -rf "$O/prof_$TAG" "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG" "$O/pmc_sq_$TAG"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-also --priming-seconds 0 --detail $O/prof_${TAG}_detail.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -- python3 $B --steps 20 --warmup 5 --priming 0 > "$O/prof_$TAG.json" 2> "$O/prof_$TAG.err" || exit 1
rm -rf "$O/prof_${TAG}_c5bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${TAG}_c5bf16" -- python3 $B --dtype bf16 --length 5000 --labels 1 --steps 20 --warmup 5 --priming 0 > "$O/prof_${TAG}_c5bf16.json" 2> "$O/prof_${TAG}_c5bf16.err" || exit 1
echo "kernel traces done"
S="$R/tools/pmc_step.py"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$TAG" -- python3 $S --log "$O/pmc_fetch_$TAG.order.json" > /dev/null 2> "$O/pmc_fetch_$TAG.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$TAG" -- python3 $S --log "$O/pmc_write_$TAG.order.json" > /dev/null 2> "$O/pmc_write_$TAG.err" || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/pmc_sq_$TAG" -- python3 $S --log "$O/pmc_sq_$TAG.order.json" > /dev/null 2> "$O/pmc_sq_$TAG.err" || exit 1
# the same three passes for the other legs of the bench line: ECGMultimodal, and BASELINE configs[4] (12x5000) in fp32 and bf16 mode
for V in "mm --model multimodal" "c5f32 --labels 1 --length 5000" "c5bf16 --labels 1 --length 5000 --dtype bf16"; do
set -- $V; T=$1; shift
rm -rf "$O/pmc_fetch_${TAG}_$T" "$O/pmc_write_${TAG}_$T" "$O/pmc_sq_${TAG}_$T"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_${TAG}_$T" -- python3 $S "$@" --log "$O/pmc_fetch_${TAG}_$T.order.json" > /dev/null 2> "$O/pmc_fetch_${TAG}_$T.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_${TAG}_$T" -- python3 $S "$@" --log "$O/pmc_write_${TAG}_$T.order.json" > /dev/null 2> "$O/pmc_write_${TAG}_$T.err" || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/pmc_sq_${TAG}_$T" -- python3 $S "$@" --log "$O/pmc_sq_${TAG}_$T.order.json" > /dev/null 2> "$O/pmc_sq_${TAG}_$T.err" || exit 1
done
echo "pmc done"
# The bench lines come LAST, with the counter traffic of THIS collection in place (profiles/pmc_traffic.json is rebuilt here,
# on the box, from the passes above: a bench line that says "stale" was run against someone else's counters)
cd "$R" && python3 tools/make_profile_artifacts.py "$TAG" --traffic-only > "$O/${TAG}_traffic_only.log" 2>&1 || { tail -5 "$O/${TAG}_traffic_only.log"; exit 1; }
python bench.py --steps 100 --warmup 20 --detail "$O/${TAG}_bench_detail.json" > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err" || exit 1
echo "bench done"
if [ "$2" != "quick" ]; then
python bench.py --steps 20 --warmup 5 --detail "$O/${TAG}_bench_driver_form_detail.json" > "$O/${TAG}_bench_driver_form.json" 2>/dev/null || exit 1
python bench.py --steps 50 --warmup 10 --graph --no-cpu-baseline --no-also --detail "$O/${TAG}_bench_graph_replay_detail.json" > "$O/${TAG}_bench_graph_replay.json" 2>/dev/null || exit 1
python bench.py --steps 100 --warmup 20 --dtype bf16 --graph --no-cpu-baseline --no-also --detail "$O/${TAG}_bench_bf16_graph_replay_detail.json" > "$O/${TAG}_bench_bf16_graph_replay.json" 2>/dev/null || exit 1
python bench.py --workload input > "$O/${TAG}_input_pipeline.json" 2>/dev/null || exit 1
python tools/bench_eval.py > "$O/${TAG}_inference.json" 2>/dev/null || exit 1
echo "variants done"
fi
