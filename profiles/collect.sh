#!/bin/bash
# Collects the per-kernel evidence bench.py's roofline leg refers to.  Run on the GPU box from the repo
# root:   bash profiles/collect.sh <tag> [workload]
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline`
#   2. two separate PMC passes (FETCH_SIZE, then WRITE_SIZE) of `bench.py --steps 2 --warmup 0`
# Raw outputs go to gpurun_out/prof_<tag>/ (scratch); the summaries are written to gpurun_out/ as
# <tag>_kernel_stats_<workload>.csv and <tag>_pmc_<workload>.json — copy them into profiles/.
set -e -o pipefail
TAG=${1:-rXX}
WL=${2:-random64}   # a workload of bench.py, or "default": the driver's command line (random64 leg + text64 leg in one process)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_$WL
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
WLARG="--workload $WL"
if [ "$WL" = "default" ]; then WLARG=""; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline $WLARG > "$OUT/stats.log" 2>&1
echo "[collect] stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline $WLARG > "$OUT/fetch.log" 2>&1
echo "[collect] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline $WLARG > "$OUT/write.log" 2>&1
echo "[collect] WRITE_SIZE pass done"
cd "$ROOT"
python3 profiles/summarize.py "$OUT" "$TAG" "$WL"
