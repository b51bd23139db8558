#!/usr/bin/env python3
"""Kernel sequence of one bench step (between two mel_logpower launches of the 32-clip batch) from a rocprofv3
kernel_trace.csv, with run-length collapsing; the decode graph replays are summarised.  Usage: step_sequence.py <csv>"""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:46], r["Grid_Size_X"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
mel = [i for i, r in enumerate(rows) if r[2].startswith("mel_logpower") and int(r[3]) > 40000]
a, b = mel[-2], mel[-1]
seq = rows[a:b]
print(f"step: {len(seq)} kernels, {(seq[-1][1]-seq[0][0])/1e6:.2f} ms")
out = []
i = 0
while i < len(seq):
    if seq[i][2].startswith("_Z21embed_norm_x16"):
        j = i
        while j < len(seq) and (seq[j][2].startswith(("_Z21embed_norm", "void skinny3", "void skinny4", "_Z24attn_decode", "_Z23slab_resnorm", "sample_"))):
            j += 1
        out.append(f"  [decode graph replays: {j-i} kernels, {(seq[j-1][1]-seq[i][0])/1e6:.2f} ms]")
        i = j
        continue
    n, g, s0 = seq[i][2], seq[i][3], seq[i][0]
    k = i
    while k + 1 < len(seq) and seq[k + 1][2] == n and seq[k + 1][3] == g:
        k += 1
    dur = sum(r[1] - r[0] for r in seq[i:k + 1]) / 1e3
    gap = (seq[i][0] - seq[i - 1][1]) / 1e3 if i else 0
    out.append(f"  +{gap:6.1f}us gap | {k-i+1:3d} x {n:46s} grid {g:>9s} {dur:9.1f} us")
    i = k + 1
print("\n".join(out))
