#!/usr/bin/env python3
"""Time q3asr_conv3x3s2_bf16 on the encoder stem shapes of the headline workload (960 chunks of 100 frames)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from qwen3_asr_mlx_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
for (imgs, H, W, C, name) in [(960, 64, 50, 480, "conv2"), (960, 32, 25, 480, "conv3")]:
    x = torch.randn(imgs, H, W, C, device=dev).to(torch.bfloat16)
    w = ops.conv_weight_chunked((torch.randn(C, 3, 3, C, device=dev) * 0.02).to(torch.bfloat16))
    b = torch.zeros(C, device=dev)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((imgs, Ho, Wo, C), dtype=torch.bfloat16, device=dev)
    ts = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.conv3x3s2(x, imgs, H, W, C, w, C, b, 1, out)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / 3)
    ts.sort()
    fl = 2.0 * imgs * Ho * Wo * C * 9 * C
    print(f"{name}: median {ts[len(ts)//2]*1e3:9.1f} us  {fl/ts[len(ts)//2]/1e9:7.1f} TFLOP/s")

# conv1 (direct K = 9 conv + GELU, 960 chunks of 100 frames): VALU-bound on the GELU
mel = torch.randn(32, 128, 3000, device=dev)
cb = torch.arange(960, dtype=torch.int32, device=dev) // 30
cf = (torch.arange(960, dtype=torch.int32, device=dev) % 30) * 100
w9 = torch.randn(480, 9, device=dev) * 0.3
b1 = torch.zeros(480, device=dev)
out1 = torch.empty((960, 50, 64, 480), dtype=torch.bfloat16, device=dev)
ts = []
for r in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.call("q3asr_conv1_gelu", ops.ptr(mel), 3000, ops.ptr(cb), ops.ptr(cf), 960, ops.ptr(w9), ops.ptr(b1), ops.ptr(out1), 480, ops.sp())
    e1.record()
    torch.cuda.synchronize()
    if r:
        ts.append(e0.elapsed_time(e1) / 3)
ts.sort()
print(f"conv1: median {ts[len(ts)//2]*1e3:9.1f} us  ({out1.numel() * 2 / ts[len(ts)//2] / 1e6:.0f} GB/s written)")
