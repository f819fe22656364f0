set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-r04_h}
python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "[all] bench rc=$?"
for wl in random64 text64; do bash profiles/collect.sh $TAG $wl > gpurun_out/${TAG}_collect_$wl.log 2>&1; echo "[all] collect $wl rc=$?"; done
for wl in batch1m lowent256 lowent64 zlibtext64; do
  OUT=gpurun_out/prof_${TAG}_$wl; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --workload $wl > $GRAFT_REPO_ROOT/$OUT/fetch.log 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --workload $wl > $GRAFT_REPO_ROOT/$OUT/write.log 2>&1 )
  python3 profiles/summarize.py $OUT $TAG $wl; echo "[all] pmc $wl done"
done
for wl in random64 text64; do bash profiles/collect_sq.sh $TAG $wl > gpurun_out/${TAG}_sq_$wl.log 2>&1; echo "[all] sq $wl rc=$?"; done
ZES_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_two_ranks_gloo_rehearsal.json 2>/dev/null; echo "[all] two ranks rc=$?"
