#!/usr/bin/env python3
"""Reproducer of the hazard in DESIGN.md section 4: kernels that each loop over ONE VALU instruction kind (tools/micro/aggr.hip,
inline asm, results a function of the thread index only) run on one stream while a kernel that only issues MFMAs runs on
another.  On the MI355X boxes of this pool `v_pk_fma_f32 ... op_sel:[0,1,0]` returns wrong values in lanes 48-63 in every
run; every other kind, and every kind beside a VALU-only kernel, is exact.  (profiles/r01_pk_fma_hazard.txt)"""
import ctypes as C
import sys
import threading
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
SO = ROOT / "tools" / "micro" / "libaggr.so"
if not SO.exists():
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", str(ROOT / "include"),
                    "-o", str(SO), str(ROOT / "tools" / "micro" / "aggr.hip")], check=True)
lib = C.CDLL(str(SO))
lib.aggr_launch.argtypes = [C.c_int, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p]
lib.victim_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
GRID, ITERS = 4096, 2000
out = torch.zeros(GRID * 256, device=dev)
sink = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
s_main, s_load = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
p_main, p_load = C.c_void_p(s_main.cuda_stream), C.c_void_p(s_load.cuda_stream)
torch.cuda.synchronize()
NAMES = ["v_pk_fma_f32", "v_pk_mul_f32 + v_pk_add_f32", "v_fma_f32", "v_fmaak_f32", "v_exp_f32", "v_rcp_f32", "v_mul_f32 + v_add_f32", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel:[0,1,0]",
         "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32", "v_pk_mul_f32 + v_pk_add_f32 op_sel:[0,1]",
         "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]"]
AGGR = {"mfma": (1, 20000), "valu": (2, 50000)}
stop = threading.Event()


def aggressor(kind):
    k, it = AGGR[kind]
    while not stop.is_set():
        for _ in range(20):
            assert lib.aggr_launch(k, C.c_void_p(sink.data_ptr()), sink.numel(), 1024, it, p_load) == 0
        s_load.synchronize()


def victim(kind):
    with torch.cuda.stream(s_main):
        out.zero_()
        assert lib.victim_launch(kind, C.c_void_p(out.data_ptr()), GRID, ITERS, p_main) == 0
        r = out.clone()
    s_main.synchronize()
    return r


refs = [victim(k) for k in range(len(NAMES))]
print("quiet repeat identical:", all(torch.equal(victim(k).view(torch.int32), refs[k].view(torch.int32)) for k in range(len(NAMES))), flush=True)
for ag in ("mfma",):
    stop.clear()
    th = threading.Thread(target=aggressor, args=(ag,))
    th.start()
    for k, name in enumerate(NAMES):
        bad_runs, bad_el = 0, 0
        lanes = torch.zeros(64, dtype=torch.long, device=dev)
        for r in range(20):
            g = victim(k)
            nz = (g.view(torch.int32) != refs[k].view(torch.int32)).nonzero().flatten()
            bad_runs += nz.numel() > 0
            bad_el += nz.numel()
            lanes += torch.bincount(nz % 64, minlength=64)
        rows = lanes.view(4, 16).sum(1).tolist()
        print(f"aggressor {ag:5s} victim {name:42s}: runs wrong {bad_runs}/20, wrong threads {bad_el}, by 16-lane row {rows}", flush=True)
    stop.set()
    th.join()

# ---- the same two instruction streams inside ONE kernel (waves 0-1 packed FMA with op_sel, waves 2-3 MFMA or idle)
lib.same_kernel_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
fsink = torch.zeros(16, device=dev)


def mix(mfma):
    with torch.cuda.stream(s_main):
        out.zero_()
        assert lib.same_kernel_launch(C.c_void_p(out.data_ptr()), C.c_void_p(fsink.data_ptr()), GRID, ITERS, mfma, p_main) == 0
        r = out.clone()
    s_main.synchronize()
    return r


quiet = mix(0)
bad_runs, lanes = 0, torch.zeros(64, dtype=torch.long, device=dev)
for r in range(20):
    nz = (mix(1).view(torch.int32) != quiet.view(torch.int32)).nonzero().flatten()
    bad_runs += nz.numel() > 0
    lanes += torch.bincount(nz % 64, minlength=64)
print(f"one kernel, waves 0-1 v_pk_fma_f32 op_sel:[0,1,0] beside its own waves 2-3 issuing MFMAs: runs wrong {bad_runs}/20, "
      f"by 16-lane row {lanes.view(4, 16).sum(1).tolist()}", flush=True)
