#!/usr/bin/env python3
"""The decode step's weight-streaming GEMM launches IN SITU: average duration of every skinny4 / skinny3 launch inside the
captured batch-32 step (attention, residual+norm and sampling launches between them), and the step's own period, from a
rocprofv3 --kernel-trace of `tools/pmc_probe.py graph4`.  Stamped with the hash of the library that ran (pmc_probe_meta.json):
bench.py quotes it as roofline.in_situ only for the same build.
Usage: in_situ.py <kernel_trace.csv> <pmc_probe_meta.json> <out.json>"""
import csv
import json
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
meta = json.load(open(sys.argv[2]))
gemm = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if any(k in r["Kernel_Name"] for k in ("skinny4_kernel", "skinny3_kernel", "head_stream_kernel"))]
starts = [int(r["Start_Timestamp"]) for r in rows if "embed_norm_x16" in r["Kernel_Name"]]
periods = sorted(b - a for a, b in zip(starts, starts[1:]))
# consecutive steps of one graph replay sit back to back: the lower half of the periods excludes replay boundaries and the eager step
per = periods[: max(1, len(periods) // 2)]
by_kernel = {}
for r in rows:
    if "skinny" in r["Kernel_Name"] or "head_stream" in r["Kernel_Name"] or "attn_decode" in r["Kernel_Name"] or "resnorm" in r["Kernel_Name"] or "sample_" in r["Kernel_Name"] or "embed_norm" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"])
        by_kernel.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"lib_sha16": meta.get("lib_sha16"), "batch": meta.get("batch"), "launches": len(gemm), "avg_launch_us": sum(gemm) / len(gemm) / 1e3,
       "steps_seen": len(starts), "whole_step_us": sum(per) / len(per) / 1e3,
       "kernels": [{"kernel": k[0], "grid": [int(k[1]), int(k[2])], "n": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2)}
                   for k, v in sorted(by_kernel.items(), key=lambda kv: -sum(kv[1]))],
       "command": "rocprofv3 --kernel-trace --output-format csv -- python3 tools/pmc_probe.py graph4 (tools/gpu_round.sh <tag> trace_graph)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
