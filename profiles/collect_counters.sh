#!/bin/bash
# Instruction-mix and LDS counters of the deflate kernels on one 64 MiB buffer (several PMC passes of
# tools/gpu_deflate_kernels.py):   bash profiles/collect_counters.sh <tag> [kind]   -> gpurun_out/<tag>_counters_<kind>.json
set -e -o pipefail
TAG=${1:-rXX}
KIND=${2:-itext}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_counters_${TOOL:+inf_}$KIND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o c -- python3 "$ROOT/tools/${TOOL:-gpu_deflate_kernels.py}" $KIND > "$OUT/p$i.log" 2>&1 || echo "[collect_counters] pass $i failed"
  echo "[collect_counters] pass $i done"
done
cd "$ROOT"
python3 - "$OUT" "$TAG" "$KIND" <<'PY'
import csv, glob, json, os, sys
out, tag, kind = sys.argv[1:4]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        d = acc.setdefault(k, {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_n_" + r["Counter_Name"]] = d.get("_n_" + r["Counter_Name"], 0) + 1
res = {"_note": "rocprofv3 --pmc passes of tools/gpu_deflate_kernels.py %s (4 deflate calls of 64 MiB each): sums over all launches of a kernel; _n_* = dispatches summed" % kind}
for k, v in acc.items():
    res[k] = {c: x for c, x in v.items()}
json.dump(res, open(os.path.join(os.path.dirname(out), "%s_counters_%s%s.json" % (tag, "inf_" if os.environ.get("TOOL") else "", kind)), "w"), indent=1)
print("[collect_counters] %d kernels" % (len(res) - 1))
PY
