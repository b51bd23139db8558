#!/usr/bin/env python3
"""Merge two rocprofv3 counter-collection CSVs (one pass --pmc FETCH_SIZE, one pass --pmc WRITE_SIZE) into a per-kernel
HBM traffic summary.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request of a
wide coalesced stream -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is.
Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [pmc_probe_meta.json]
The meta file (written by tools/pmc_probe.py on the GPU box) stamps the summary with the sha-256 of the library that ran;
bench.py reports `roofline.traffic` only from a summary whose stamp equals the library it is running."""
import collections
import csv
import json
import sys


def load(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            key = (r["Kernel_Name"][:60], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            agg[key].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
    f = fetch[key]
    w = write.get(key, [0.0])
    rows.append({"kernel": key[0], "grid_threads": key[1], "wg": key[2], "dispatches": len(f),
                 "FETCH_SIZE_KB_avg": round(sum(f) / len(f), 1), "hbm_read_MB_corrected_x2": round(2 * sum(f) / len(f) * 1024 / 1e6, 2),
                 "WRITE_SIZE_KB_avg": round(sum(w) / len(w), 1), "hbm_write_MB": round(sum(w) / len(w) * 1024 / 1e6, 2)})
sk = [r for r in rows if any(k in r["kernel"] for k in ("skinny4_kernel", "skinny3_kernel", "head_stream_kernel"))]
n = sum(r["dispatches"] for r in sk)
meta = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/pmc_probe.py " + ",".join(meta.get("ran", ["?"])) +
                  "  (second pass with --pmc WRITE_SIZE)",
       "lib_sha16": meta.get("lib_sha16"), "probe": meta,
       "correction": "gfx950: read bytes = 2 * FETCH_SIZE * 1024; write bytes = WRITE_SIZE * 1024",
       "decode_gemm_hbm_bytes_per_launch": int(sum((r["hbm_read_MB_corrected_x2"] + r["hbm_write_MB"]) * 1e6 * r["dispatches"] for r in sk) / max(n, 1)),
       "kernels": rows[:40]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
for r in rows[:14]:
    print(r)
