#!/usr/bin/env python3
"""Source-level view of the same hazard (tools/pk_fma_hazard.py): copies of conv1_gelu_kernel with one part replaced each, beside an
MFMA-only kernel on another stream.  The copies whose 3x3 accumulation the compiler turns into v_pk_fma_f32 (op_sel) are
wrong in even channels of lanes 48-63; with the accumulation as scalar v_fma_f32 the output is exact."""
import ctypes as C
import sys
import threading
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
dev = torch.device("cuda:0")
torch.manual_seed(0)
SO = ROOT / "tools" / "micro" / "libaggr.so"
if not SO.exists():
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", str(ROOT / "include"),
                    "-o", str(SO), str(ROOT / "tools" / "micro" / "aggr.hip")], check=True)
lib = C.CDLL(str(SO))
lib.aggr_launch.argtypes = [C.c_int, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p]
lib.conv1_variant_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
C1, Ct, F = 480, 240, 3000
mel = torch.randn(8, 128, F, device=dev)
chunk_b = (torch.arange(Ct, device=dev, dtype=torch.int32) // 30).contiguous()
chunk_f0 = ((torch.arange(Ct, device=dev, dtype=torch.int32) % 30) * 100).contiguous()
w9 = (torch.randn(C1, 9, device=dev) * 0.25).contiguous()
bias = (torch.randn(C1, device=dev) * 0.02).contiguous()
out = torch.empty((Ct, 50, 64, C1), dtype=torch.bfloat16, device=dev)
sink = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
s_main, s_load = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
p_main, p_load = C.c_void_p(s_main.cuda_stream), C.c_void_p(s_load.cuda_stream)
torch.cuda.synchronize()
NAMES = ["as shipped (erfc-form GELU)", "no GELU", "GELU through erff", "no GELU, v_fma_f32 by inline asm"]
stop = threading.Event()


def aggressor():
    while not stop.is_set():
        for _ in range(20):
            assert lib.aggr_launch(1, C.c_void_p(sink.data_ptr()), sink.numel(), 1024, 20000, p_load) == 0
        s_load.synchronize()


def victim(var):
    with torch.cuda.stream(s_main):
        out.zero_()
        assert lib.conv1_variant_launch(var, C.c_void_p(mel.data_ptr()), F, C.c_void_p(chunk_b.data_ptr()), C.c_void_p(chunk_f0.data_ptr()), Ct,
                                        C.c_void_p(w9.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), C1, p_main) == 0
        r = out.clone()
    s_main.synchronize()
    return r


refs = [victim(v) for v in range(4)]
print("quiet repeat identical:", all(torch.equal(victim(v).view(torch.int16), refs[v].view(torch.int16)) for v in range(4)), flush=True)
th = threading.Thread(target=aggressor)
th.start()
for v, name in enumerate(NAMES):
    bad_runs, bad_el = 0, 0
    par = torch.zeros(8, dtype=torch.long, device=dev)
    for r in range(12):
        g = victim(v)
        nz = (g.view(torch.int16) != refs[v].view(torch.int16)).flatten().nonzero().flatten()
        bad_runs += nz.numel() > 0
        bad_el += nz.numel()
        par += torch.bincount((nz % 480) % 8, minlength=8)
    print(f"victim conv1 {name:34s}: runs wrong {bad_runs}/12, wrong elements {bad_el}, by channel-in-octet {par.tolist()}", flush=True)
stop.set()
th.join()
