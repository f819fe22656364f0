#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py's roofline.traffic reads) from the per-workload PMC summaries of one round:
   python3 profiles/make_traffic.py r02_a random64 text64 ..."""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
tag, wls = sys.argv[1], sys.argv[2:]
out = {"_note": "HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of `bench.py --steps 2 "
                "--workload W`, profiles/collect.sh): (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / launches; FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 reports half of a wide streaming read; dword-wide and byte loads are uncalibrated, so the read "
                "side is an upper estimate). Source: profiles/%s_pmc_{%s}.json" % (tag, ",".join(wls))}
for wl in wls:
    d = json.load(open(os.path.join(here, "%s_pmc_%s.json" % (tag, wl))))
    out[wl] = {k: v["hbm_bytes_per_launch"] for k, v in d.items() if isinstance(v, dict) and k.startswith("k_")}
json.dump(out, open(os.path.join(here, "traffic.json"), "w"), indent=1)
print("traffic.json:", ", ".join("%s (%d kernels)" % (w, len(out[w])) for w in wls))
