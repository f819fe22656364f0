#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: every launch with its duration and the idle gap
in front of it (host work, copies and synchronisations between launches show up as gaps).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload random64
    python3 tools/trace_gaps.py gpurun_out/tr/*/tr_kernel_trace.csv [first_kernel_of_a_step]
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "k_make_blks"
starts = [i for i, r in enumerate(rows) if r[2] == first]
if len(starts) < 2:
    sys.exit("fewer than two steps in the trace")
a, b = starts[-2], starts[-1]  # the last complete step
step = rows[a:b]
busy = sum(e - s for s, e, _ in step)
span = rows[b][0] - step[0][0]
prev = step[0][0]
for s, e, n in step:
    print(f"{(s - step[0][0]) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  run {(e - s) / 1e3:7.1f}  {n}")
    prev = e
print(f"step {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us ({100 * (span - busy) / span:.1f} %)")
