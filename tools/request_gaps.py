#!/usr/bin/env python3
"""Where one timed request of bench.py spends wall time that is NOT kernel time: from a rocprofv3 kernel_trace.csv, the last
`mel_logpower_kernel` marks the start of the last request; prints, for that request, the wall span, the summed kernel time and every
idle gap above 30 us with the kernels on either side (host work between stages, synchronisations, graph replay boundaries).
Usage: request_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(sys.argv[1]))))
mels = [i for i, r in enumerate(rows) if "mel_logpower" in r[2]]
# the bench's timed steps are full batches: take the last mel launch that is followed by a sampler launch
i0 = mels[-1]
for cand in reversed(mels):
    nxt = [m for m in mels if m > cand]
    end = nxt[0] if nxt else len(rows)
    if sum("sample_" in r[2] for r in rows[cand:end]) >= 100:
        i0, i1 = cand, end
        break
req = rows[i0:i1]
span = req[-1][1] - req[0][0]
busy = sum(e - s for s, e, _ in req)
gaps = [(b[0] - a[1], a[2], b[2], (a[1] - req[0][0]) / 1e6) for a, b in zip(req, req[1:]) if b[0] - a[1] > 30000]
print(f"request: {len(req)} kernels, wall span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle {100 * (span - busy) / span:.1f} %")
print(f"gaps > 30 us: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.2f} ms; all gaps {(span - busy) / 1e6:.2f} ms")
for g, a, b, at in sorted(gaps, reverse=True)[:25]:
    print(f"  {g / 1e3:8.1f} us at +{at:7.2f} ms  after {a:44s} before {b}")
