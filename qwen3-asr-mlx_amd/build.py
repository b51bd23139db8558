"""Build libq3asr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libq3asr_hip.so"
SOURCES = ["core.hip", "mel.hip", "gemm.hip", "norm.hip", "elementwise.hip", "attn.hip", "decode.hip", "sampling.hip", "parity.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", str(HERE.parent / "include")]
if os.environ.get("Q3ASR_PACKED_F32", "0") != "1":
    # no v_pk_*_f32: see DESIGN.md section 4 (wrong results beside a foreign MFMA kernel).  The host pass prints "not a
    # recognized feature" for it and ignores it.
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


if os.environ.get("Q3ASR_KERNARG_PRELOAD", "1") != "0":
    # gfx950 can hand the first 16 dwords of a kernel's arguments to the wave in SGPRs at dispatch instead of a scalar load from a
    # cold cache at the top of every kernel: the decode step is 199 dependent launches of 4-14 us, and each one's first loads
    # sit behind that load (tools/micro/adf_time.hip: qkv projection + decode attention 13.69 -> 13.32 us per layer at batch 1,
    # 19.81 -> 19.03 at batch 32).  Code generation only: same results.  Q3ASR_KERNARG_PRELOAD=0 builds without it (A/B).
    FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _stale(obj: Path, deps) -> bool:
    return (not obj.exists()) or any(d.stat().st_mtime > obj.stat().st_mtime for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)
    common = [CSRC / "common.h", HERE.parent / "include" / "q3asr.h", Path(__file__)]
    jobs = []
    for s in SOURCES:
        src, obj = CSRC / s, objdir / (s + ".o")
        if force or _stale(obj, [src] + common):
            jobs.append([hipcc, *FLAGS, "-c", str(src), "-o", str(obj)])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [str(objdir / (s + ".o")) for s in SOURCES]
    if force or jobs or not LIB.exists():
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
