#!/usr/bin/env python3
"""The measured table of DESIGN.md §6 from one bench.py line:  python3 profiles/make_table.py <bench json> [--write]
(--write: replaces the text between the MEASURED markers of DESIGN.md)"""
import json
import os
import re
import sys

here = os.path.dirname(os.path.abspath(__file__))
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rows = []
def pct(r):
    return "%.2f %%" % (100 * r["frac"]) if r else "—"
def row(name, leg, deflate, inflate, rt, c_over_n):
    rd, ri = leg.get("roofline_deflate"), leg.get("roofline")
    if name == "random64":
        rd = ri = leg.get("roofline")
    dom = " / ".join("`%s` %.2f ms" % (r["kernel"], r["avg_launch_ms"]) for r in (rd, ri) if r and (name != "random64" or r is ri))
    fr = " / ".join(pct(r) for r in (rd, ri) if r and (name != "random64" or r is ri))
    rows.append("| %s | %s | %s | %s | %s | %s | %s |" % (name, c_over_n, deflate, inflate, rt, dom, fr))
n = d["config"]["bytes_per_buffer"] * d["config"]["buffers_per_step"] // d["n_gpus"]
row("random64 (`configs[1]`, `value`)", d, "%.1f" % d["deflate_gibs_per_gpu"], "%.1f" % d["inflate_gibs_per_gpu"],
    "**%.1f** (%.1f – %.1f)" % (d["value"], d["value_spread"]["min"], d["value_spread"]["max"]), "%.3f" % (d["config"]["compressed_bytes"] / n))
sizes = {"text64": 64 << 20, "batch1m": 128 << 20, "lowent256": 256 << 20, "zlibtext64": 64 << 20}
names = {"text64": "text64 (`configs[2]`)", "batch1m": "batch1m (`configs[3]` share)", "lowent256": "lowent256 (`configs[4]`)", "zlibtext64": "zlibtext64 (§8f.1, T2)"}
for k in ("text64", "batch1m", "lowent256", "zlibtext64"):
    l = d.get(k)
    if not l:
        continue
    row(names[k], l, "%.1f" % l["deflate_gibs_per_gpu"] if "deflate_gibs_per_gpu" in l else "—", "%.1f" % l["inflate_gibs_per_gpu"],
        "%.1f" % l["round_trip_gibs_per_gpu"] if "round_trip_gibs_per_gpu" in l else "—", "%.3f" % (l["compressed_bytes"] / sizes[k]))
out = ["| workload | c / n | deflate GiB/s | inflate GiB/s | round trip | dominant kernel (deflate / inflate) | (n+c)/t vs 8 TB/s |", "|---|---|---|---|---|---|---|"] + rows
h = d.get("host_api")
if h and "rows" in h:
    out += ["", "Host-pointer API, 64 MiB calls, PCIe included (never `value`); link measured on the same box: %.0f GB/s up, %.0f down:" % (h["link"]["h2d_GBs"], h["link"]["d2h_GBs"]), "",
            "| | deflate GiB/s | inflate (`zes_inflate_alloc`, early estimate / exact size) GiB/s |", "|---|---|---|"]
    for k, v in h["rows"].items():
        out.append("| C-ABI, %s | %.1f | %.1f / %.1f |" % (k.replace("_", ", "), v["deflate_gibs"], v["inflate_gibs"], v.get("inflate_exact_alloc_gibs", 0)))
    nd = h.get("node") or {}
    for k, v in (nd.get("rows") or {}).items():
        out.append("| Node façade, %s | %.1f | %.1f |" % (k.replace("_", ", "), v["deflate_gibs"], v["inflate_gibs"]))
cb = d.get("cpu_baseline")
if cb:
    tb = (d.get("text64") or {}).get("cpu_baseline") or {}
    out += ["", "CPU baseline on the same box (the oracle = a port of the reference's algorithm, 1 thread of the host's cores): random64 %.3f GiB/s deflate / %.3f inflate; text64 %s / %s.  The reference itself under Node 12 in the build container: 1.23 MiB/s deflate, 34 MiB/s inflate (BASELINE.md)." % (
        cb.get("deflate_gibs", 0), cb.get("inflate_gibs", 0), tb.get("deflate_gibs", "—"), tb.get("inflate_gibs", "—"))]
text = "\n".join(out)
print(text)
if "--write" in sys.argv:
    p = os.path.join(os.path.dirname(here), "DESIGN.md")
    s = open(p).read()
    if "MEASURED_TABLE" in s:
        s = s.replace("MEASURED_TABLE", "<!-- MEASURED:BEGIN (profiles/make_table.py) -->\n" + text + "\n<!-- MEASURED:END -->")
    else:
        s = re.sub(r"<!-- MEASURED:BEGIN.*?<!-- MEASURED:END -->", lambda m: "<!-- MEASURED:BEGIN (profiles/make_table.py) -->\n" + text + "\n<!-- MEASURED:END -->", s, flags=re.S)
    open(p, "w").write(s)
