#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel_trace.csv, for the decode step: groups gaps by the
kernel that follows.  Usage: gap_stats.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
gaps = collections.defaultdict(list)
busy = idle = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 20000:            # host-side pauses (syncs, python) are not the chain's cost
        continue
    gaps[n1].append(g)
    idle += max(g, 0)
    busy += e1 - s1
print(f"kernels {len(rows)}  busy {busy/1e6:.1f} ms  idle between kernels (gaps < 20 us) {idle/1e6:.1f} ms")
for n, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort()
    print(f"{n:42s} n={len(v):6d} median gap {v[len(v)//2]/1e3:6.2f} us  mean {sum(v)/len(v)/1e3:6.2f} us  total {sum(v)/1e6:7.1f} ms")
