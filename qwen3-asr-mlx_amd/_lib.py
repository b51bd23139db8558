"""ctypes binding of libq3asr_hip.so (include/q3asr.h).  No CPU fallback: if the library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libq3asr_hip.so"
_lib = None
_lock = threading.Lock()

_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGS = {
    "q3asr_version": [],
    "q3asr_arch": [],
    "q3asr_last_error": [],
    "q3asr_log_mel": [_vp, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "q3asr_log_mel_generic": [_vp, _i64, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "q3asr_conv1_gelu": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "q3asr_conv1_gelu_f32": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "q3asr_split3_f32": [_vp, _i64, _vp, _i64, _i, _vp],
    "q3asr_layernorm_f32_split3": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "q3asr_attn_fwd_f32": [_vp, _i64, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i64, _vp],
    "q3asr_conv3x3s2_bf16": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _vp, _vp],
    "q3asr_gemm_bf16": [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _i, _i64, _vp, _vp],
    "q3asr_gemm_fp8": [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _vp, _i, _i64, _vp, _i64, _vp],
    "q3asr_rmsnorm_bf16_fp8": [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _f, _vp],
    "q3asr_quantize_rows_fp8": [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp],
    "q3asr_layernorm_f32_bf16": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "q3asr_rmsnorm_bf16": [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i, _f, _vp],
    "q3asr_attn_fwd": [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "q3asr_attn_set_stamp": [_vp, _i],
    "q3asr_embed_merge": [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp],
    "q3asr_qk_norm_rope_kv": [_vp, _i, _i64, _i, _i64, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "q3asr_skinny3_gemm": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp],
    "q3asr_skinny4_gemm": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp],
    "q3asr_skinny4_gemm_norm": [_vp, _i, _i64, _vp, _vp, _vp, _f, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp],
    "q3asr_skinny3_gemm_norm": [_vp, _i, _i64, _vp, _vp, _vp, _f, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp],
    "q3asr_slab_resnorm_x16": [_vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "q3asr_embed_norm_x16": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp],
    "q3asr_attn_decode_fused_workspace": [_i, _i, _i],
    "q3asr_attn_decode_set_chunks_per_wg": [_i],
    "q3asr_attn_decode_fused": [_vp, _i64, _i, _i64, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "q3asr_sample_step": [_vp, _i64, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _i, _u64, _i, _i, _i, _vp],
    "q3asr_graph_begin": [_vp],
    "q3asr_graph_end": [_vp, C.POINTER(_vp)],
    "q3asr_graph_launch": [_vp, _vp],
    "q3asr_graph_destroy": [_vp],
    "q3asr_gemm_force_kernel": [_i],
    "q3asr_gemm_set_interior": [_i],
    "q3asr_set_option": [_i, _i],
    "q3asr_gemm_set_stamp": [_vp],
    "q3asr_gemm_set_stamp_wg": [_i],
    "q3asr_event_create": [C.POINTER(_vp)],
    "q3asr_event_record": [_vp, _vp],
    "q3asr_event_elapsed_ms": [_vp, _vp, C.POINTER(_f)],
    "q3asr_event_destroy": [_vp],
}
_RESTYPES = {"q3asr_arch": C.c_char_p, "q3asr_last_error": C.c_char_p, "q3asr_attn_decode_fused_workspace": C.c_int64}
EXPORTS = tuple(_SIGS)


class Q3AsrError(RuntimeError):
    pass


def load():
    """Return the loaded library (loading it on first use).  Raises if it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not LIB_PATH.exists():
                    raise Q3AsrError(
                        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
                # PyTorch-ROCm ships its own libamdhip64; the library must bind to THAT runtime (the one whose device pointers
                # it is handed), so torch is imported first: loaded the other way round, the two runtimes do not see each
                # other's device ("no ROCm-capable device is detected" at the first launch).
                import torch  # noqa: F401
                lib = C.CDLL(str(LIB_PATH))
                for name, args in _SIGS.items():
                    fn = getattr(lib, name)
                    fn.argtypes = args
                    fn.restype = _RESTYPES.get(name, C.c_int)
                _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().q3asr_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise Q3AsrError(f"{what}: {msg} (rc={rc})")


PROFILE = None          # bench.py: {"names": set, "stream": c_void_p, "events": [(name, start, stop)]}


def call(name: str, *args) -> None:
    lib = load()
    prof = PROFILE
    if prof is not None and name in prof["names"]:
        e0, e1 = C.c_void_p(), C.c_void_p()
        lib.q3asr_event_create(C.byref(e0))
        lib.q3asr_event_create(C.byref(e1))
        lib.q3asr_event_record(e0, prof["stream"])
        check(getattr(lib, name)(*args), name)
        lib.q3asr_event_record(e1, prof["stream"])
        prof["events"].append((name, e0, e1, args))
        return
    check(getattr(lib, name)(*args), name)


def drain_profile(prof) -> list:
    """[(name, ms, args)] for the recorded calls; destroys the events."""
    lib = load()
    out = []
    for name, e0, e1, args in prof["events"]:
        ms = C.c_float()
        check(lib.q3asr_event_elapsed_ms(e0, e1, C.byref(ms)), "event_elapsed")
        out.append((name, float(ms.value), args))
        lib.q3asr_event_destroy(e0)
        lib.q3asr_event_destroy(e1)
    prof["events"].clear()
    return out


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
