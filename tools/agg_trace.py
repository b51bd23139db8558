#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid): calls, avg/median/min us, total ms."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:48], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e6:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{k[0]:50s} grid=({k[1]},{k[2]},{k[3]}) wg={k[4]:>4s} n={len(v):6d} avg={sum(v)/len(v)/1e3:9.1f}us med={sorted(v)[len(v)//2]/1e3:8.1f}us min={min(v)/1e3:8.1f}us total={sum(v)/1e6:8.1f}ms {100*sum(v)/tot:5.1f}%")
