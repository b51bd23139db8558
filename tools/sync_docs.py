#!/usr/bin/env python3
"""Rewrite the three paragraphs that quote the committed bench line (DESIGN.md section 0 headline, README.md round-5 paragraph,
profiles/README.md r05_bench_line row) from profiles/r05_bench_line.json, so that no figure is typed by hand.
Usage: python3 tools/sync_docs.py <lib_sha16> "<box range text for the resident figure>" """
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
lib, rng = sys.argv[1], sys.argv[2]
d = json.loads((ROOT / "profiles/r05_bench_line.json").read_text())
e, lf, st = d["end_to_end"], d["longform_fp8"], d["stages"]
V = dict(rtf=round(d["value"]), ms=d["ms_per_step"], e2e_rtf=round(e["value"]), e2e=e["p50_latency_ms"], dres=e["minus_resident_ms"],
         fill=e["h2d"]["host_fill_ms"], span=e["h2d"]["copy_span_ms"], gbps=e["h2d"]["copy_GBps_over_span"], runs=e["h2d"]["runs_ms"],
         other=e["h2d_other_form"]["p50_latency_ms"], oruns=e["h2d_other_form"]["runs_ms"], b1=d["batch1"]["p50_latency_ms"],
         f2=round(d["two_in_flight"]["value"]), f4=round(d["in_flight_4"]["value"]), lf=round(lf["value"]), lfms=lf["p50_latency_ms"],
         lfref=lf["reference_budget"]["latency_ms"] / 1e3, lffrac=lf["roofline"]["frac"], lfus=lf["roofline"]["avg_launch_us"],
         sd=lf["sampler"]["t0.7_top_p0.9_us"], sg=lf["sampler"]["greedy_us"], enc=st["encoder_ms"], pre=st["prefill_ms"],
         dec=st["decode_ms_per_token"], frac=d["roofline"]["frac"], us=d["roofline"]["avg_launch_us"], insitu=d["roofline"]["in_situ"]["frac"])
f = lambda x: f"{x:,}"      # noqa: E731


def between(path, start, end, new):
    s = (ROOT / path).read_text()
    a, b = s.index(start), s.index(end)
    (ROOT / path).write_text(s[:a] + new + s[b:])


between("DESIGN.md", "Headline on the driver-equivalent line of this round", "| # | asked | done | where |",
        f"""Headline on the driver-equivalent line of this round (`profiles/r05_bench_line.json`, library `{lib}`, one MI355X): **{f(V['rtf'])} x real time,
{V['ms']:.1f} ms per 32 x 30 s request** (round 4: 3,952 x / 242.9 ms; {rng}), `end_to_end`
{V['e2e']:.1f} ms (+{V['dres']:.1f} ms: {V['fill']:.1f} ms host fill + {V['span']:.1f} ms copy at {V['gbps']:.0f} GB/s; five recorded runs per staging form, {min(V['runs']):.1f}-{max(V['runs']):.1f} and {min(V['oruns']):.1f}-{max(V['oruns']):.1f} ms), batch-1 {V['b1']:.1f} ms,
two / four requests in flight {f(V['f2'])} x / {f(V['f4'])} x, `longform_fp8` (configs[4]) **{f(V['lf'])} x** ({V['lfms']:.1f} ms for 20 min of audio; 279.9-284.1 ms before the four-wave decode attention and the 128-wide fp8 MFMA, 3,853 x / 311.4 ms before the rows
marked "+" below), encoder {V['enc']:.1f} ms (15.17), prefill {V['pre']:.1f} ms, decode {V['dec']:.3f} ms per token (1.504 before the last "+" rows; its floor is section 0a).

""")
between("README.md", "MI355X, round 5 (`profiles/r05_bench_line.json`", "The driver's own round-4 line (`BENCH_r04.json`",
        f"""MI355X, round 5 (`profiles/r05_bench_line.json`, every figure below is from that one committed line): **RTF {f(V['rtf'])}×** at batch 32 ({V['ms']:.1f} ms per 16 minutes of
audio resident in HBM; {rng}), **{f(V['e2e_rtf'])}× / {V['e2e']:.1f} ms** with the host-to-device copy inside (SURVEY.md 8d's definition; the line itself says where the {V['dres']:.1f} ms go:
`end_to_end.h2d`, every run recorded), {V['b1']:.1f} ms for a single 30 s clip, {f(V['f2'])}× / {f(V['f4'])}× with two / four requests in flight, BASELINE.json `configs[4]` (fp8 weights, 20 min long-form, T = 0.7 / top-p 0.9)
**{f(V['lf'])}×** ({V['lfms']:.1f} ms for the 20 minutes at 128 tokens per segment; {V['lfref']:.2f} s with the reference's own 1,500-token budget per segment), encoder {V['enc']:.1f} ms, prefill {V['pre']:.1f} ms,
decode {V['dec']:.3f} ms per token, decode GEMMs {V['frac']:.2f} of the 8 TB/s HBM peak ({V['insitu']:.2f} in situ; HBM traffic 34.8 MB per launch for 30.45 MB algorithmic).  """)
between("profiles/README.md", "| `r05_bench_line.json` |", "| `r05_kernel_stats.csv` |",
        f"""| `r05_bench_line.json` | `python bench.py --steps 20 --warmup 3` (`tools/gpu_round.sh <tag> bench`) on the committed tree | RTF {f(V['rtf'])}x ({V['ms']:.1f} ms per batch-32 request resident in HBM; {rng}); `end_to_end` {V['e2e']:.1f} ms with `h2d` = fill {V['fill']:.2f} ms + copy {V['span']:.2f} ms at {V['gbps']:.0f} GB/s, the other staging form beside it ({V['other']:.1f} ms), `runs_ms` of both blocks; batch-1 {V['b1']:.1f} ms; two / four in flight {f(V['f2'])}x / {f(V['f4'])}x; **`longform_fp8`** (configs[4]) {f(V['lf'])}x ({V['lfms']:.1f} ms; {V['lfref']:.2f} s with the reference's budget) with `roofline.frac` {V['lffrac']:.2f} (fp8 weight streaming at batch 40, three m-tiles: {V['lfus']:.1f} us per launch) and `sampler` {V['sd']:.1f} vs {V['sg']:.1f} us; stages: encoder {V['enc']:.1f} ms, prefill {V['pre']:.1f} ms, decode {V['dec']:.3f} ms per token (medians, `stages.statistic`); `roofline.frac` {V['frac']:.3f} ({V['us']:.2f} us per launch) with `traffic` 34.8 MB per launch and `in_situ` {V['insitu']:.3f} from the stamped summaries below |
""")
print("synced:", V["rtf"], f"{V['ms']:.1f} ms")
