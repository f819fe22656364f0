#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh into the two summaries kept under profiles/:
<tag>_kernel_stats_<workload>.csv (name, calls, total/avg/min/max ns, %) and <tag>_pmc_<workload>.json
(per kernel: launches, FETCH_SIZE and WRITE_SIZE sums in KiB as reported, and HBM bytes per launch with
the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE x 2)."""
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.split("(")[0].strip()
    return name.split(" ")[-1].split("::")[-1]


def main():
    out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    dest = os.path.join(os.path.dirname(out.rstrip("/")), "")
    stats = find(os.path.join(out, "stats"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(dest, "%s_kernel_stats_%s.csv" % (tag, wl)), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        print("[summarize] kernel stats: %d kernels" % len(rows))
    pmc = {}
    for which in ("fetch", "write"):
        path = find(os.path.join(out, which), "*counter_collection.csv")
        if not path:
            continue
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            e = pmc.setdefault(k, {"launches": {}, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
            e[r["Counter_Name"]] += float(r["Counter_Value"])
            e["launches"][which] = e["launches"].get(which, 0) + 1
    res = {"_note": "KiB as reported by rocprofv3 --pmc (separate passes); hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches"}
    for k, e in pmc.items():
        nf, nw = e["launches"].get("fetch", 0), e["launches"].get("write", 0)
        per = 0.0
        if nf:
            per += 2.0 * e["FETCH_SIZE"] * 1024.0 / nf
        if nw:
            per += e["WRITE_SIZE"] * 1024.0 / nw
        res[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "FETCH_SIZE_KiB_sum": e["FETCH_SIZE"],
                  "WRITE_SIZE_KiB_sum": e["WRITE_SIZE"], "hbm_bytes_per_launch": int(per)}
    json.dump(res, open(os.path.join(dest, "%s_pmc_%s.json" % (tag, wl)), "w"), indent=1)
    print("[summarize] pmc: %d kernels" % len(pmc))


if __name__ == "__main__":
    main()
