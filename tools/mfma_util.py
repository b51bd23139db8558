#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 counter-collection CSV taken with
`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`: busy cycles summed over the 1,024 SIMDs against elapsed cycles
(GRBM_GUI_ACTIVE is summed over the 8 XCDs): util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024); effective clock = GUI_ACTIVE / 8 /
kernel time when a kernel trace CSV is given too.  Usage: mfma_util.py <counter_collection.csv> [kernel_trace.csv] [out.json]"""
import collections
import csv
import json
import sys

busy, gui = collections.defaultdict(list), collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"][:50], int(r["Grid_Size"]))
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[key].append(float(r["Counter_Value"]))
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui[key].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
if len(sys.argv) > 2 and sys.argv[2].endswith(".csv"):
    for r in csv.DictReader(open(sys.argv[2])):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        dur[(r["Kernel_Name"][:50], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for key in busy:
    if key not in gui or not sum(gui[key]):
        continue
    b, g = sum(busy[key]) / len(busy[key]), sum(gui[key]) / len(gui[key])
    row = {"kernel": key[0], "grid_threads": key[1], "dispatches": len(busy[key]), "mfma_busy_cycles": round(b),
           "elapsed_cycles": round(g / 8), "mfma_util": round(b / (g / 8 * 1024), 4)}
    if dur.get(key):
        ns = sum(dur[key]) / len(dur[key])
        row["avg_us"] = round(ns / 1e3, 1)
        row["effective_clock_GHz"] = round(g / 8 / ns, 3)
    rows.append(row)
rows.sort(key=lambda r: -r["mfma_busy_cycles"] * r["dispatches"])
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 tools/pmc_probe.py encoder,prefill "
                  "(tools/gpu_round.sh <tag> mfma): one mel + encoder pass and one prefill of 32 x 30 s clips at the 1.7B shapes, in situ",
       "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "kernels": rows[:20]}
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
for r in rows[:12]:
    print(r)
