#!/bin/bash
# Per-wave issue statistics of every kernel of one workload (one PMC pass):
#   bash profiles/collect_sq.sh <tag> [workload]   -> gpurun_out/<tag>_sq_<workload>.json (copy it into profiles/)
set -e -o pipefail
TAG=${1:-rXX}
WL=${2:-random64}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG/sq_$WL
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT" -o sq -- python3 "$ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --workload "$WL" > "$OUT/run.log" 2>&1
cd "$ROOT"
python3 - "$OUT" "$TAG" "$WL" <<'PY'
import csv, glob, json, os, sys
out, tag, wl = sys.argv[1:4]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"_note": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY on `bench.py --steps 2 --warmup 0 --workload %s`; "
                "fractions of a wave's cycles: wait_any = parked on s_waitcnt/barrier, active_inst = issuing, wait_inst = issue stall" % wl}
for k, v in acc.items():
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0 or not k.startswith("k_"):
        continue
    res[k] = {"wave_cycles": wc, "wait_any": round(v.get("SQ_WAIT_ANY", 0) / wc, 3), "active_inst": round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
              "wait_inst": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(out)), "%s_sq_%s.json" % (tag, wl)), "w"), indent=1)
print("[collect_sq] %d kernels" % (len(res) - 1))
PY
