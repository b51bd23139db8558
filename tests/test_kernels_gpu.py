"""Per-kernel parity on the GPU: every C-ABI entry point against the oracle / a float32 torch
restatement of the same op, on seeded inputs, including the ragged / edge shapes."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from signals import signal
from oracle.decoder import swiglu as swiglu_ref      # the reference's three rounding points (decoder.py:96-97)

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def bfr(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rnd(shape, dev, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


# ----------------------------------------------------------------------------- mel
def test_mel_matches_reference_golden_and_oracle(dev):
    import oracle
    import qwen3_asr_mlx_amd as q
    full = np.load(GOLD / "mel_full.npz")
    for key in full.files:
        if "_n" in key:
            x = signal("noise", int(key.split("_n")[1]))
        else:
            kind, s = key.split("_")
            x = signal(kind, int(float(s[:-1]) * 16000))
        got = q.log_mel_spectrogram(x).cpu().numpy()
        assert got.shape == full[key].shape
        # tolerance: 2e-6 abs on the normalised log-mel (fp64-accumulated DFT vs pocketfft float32)
        np.testing.assert_allclose(got, full[key], atol=2e-6, rtol=0, err_msg=key)
    dg = json.loads((GOLD / "mel_digest.json").read_text())
    for kind in ("tone", "noise", "speechlike", "silence"):
        x = signal(kind, 480000)
        got = q.log_mel_spectrogram(x).cpu().numpy()
        assert list(got.shape) == dg[f"{kind}_30s"]["shape"]
        ref = np.array(dg[f"{kind}_30s"]["every97"], dtype=np.float32)
        np.testing.assert_allclose(got.reshape(-1)[::97][:4096], ref, atol=2e-6, rtol=0)
        np.testing.assert_allclose(got, oracle.log_mel_spectrogram(x), atol=2e-6, rtol=0)


def test_mel_non_default_parameters_match_reference_golden(dev):
    """log_mel_spectrogram(audio, n_fft, hop_length, n_mels, sample_rate, f_min, f_max) for triples other than the encoder's
    (q3asr_log_mel_generic) against fixtures from the live reference; same tolerance as the default path."""
    import qwen3_asr_mlx_amd as q
    from test_oracle_cpu import mel_param_cases
    from qwen3_asr_mlx_amd import audio
    for name, x, params, want in mel_param_cases():
        got = q.log_mel_spectrogram(x, **params).cpu().numpy()
        assert got.shape == want.shape, name
        np.testing.assert_allclose(got, want, atol=2e-6, rtol=0, err_msg=name)
    # a ragged batch through the generic kernel: frames beyond an utterance's count are zero, rows are batch-invariant
    fe = audio.get_frontend(dev, n_fft=512, hop_length=128, n_mels=80)
    xs = [signal("noise", 16000), signal("tone", 3000), signal("speechlike", 7777)]
    a, ld, lens = audio.pack_audio(xs, dev)
    mel, nfr = fe(a, ld, lens)
    assert nfr == [len(x) // 128 for x in xs] and mel.shape == (3, 80, 125)
    for i, x in enumerate(xs):
        assert torch.equal(mel[i, :, : nfr[i]], q.log_mel_spectrogram(x, n_fft=512, hop_length=128, n_mels=80))
        if nfr[i] < mel.shape[2]:
            assert float(mel[i, :, nfr[i]:].abs().max()) == 0.0


def test_mel_batch_is_ragged_safe_and_batch_invariant(dev):
    import oracle
    from qwen3_asr_mlx_amd import audio
    xs = [signal("noise", 80000), signal("tone", 16000), signal("noise", 1234), signal("silence", 8000),
          signal("speechlike", 48077)]
    fe = audio.get_frontend(dev)
    a, ld, lens = audio.pack_audio(xs, dev)
    mel, nfr = fe(a, ld, lens)
    assert nfr == [len(x) // 160 for x in xs]
    for i, x in enumerate(xs):
        got = mel[i, :, : nfr[i]].cpu().numpy()
        np.testing.assert_allclose(got, oracle.log_mel_spectrogram(x), atol=2e-6, rtol=0)
        assert float(mel[i, :, nfr[i]:].abs().max().item() if nfr[i] < mel.shape[2] else 0.0) == 0.0
        single = audio.log_mel_spectrogram(x)
        assert torch.equal(single, mel[i, :, : nfr[i]])          # bitwise: independent of batch mates
    assert np.allclose(mel[3, :, : nfr[3]].cpu().numpy(), -1.5, atol=1e-6)   # reference test_audio.py:80-89


def test_mel_rejects_too_short(dev):
    import qwen3_asr_mlx_amd as q
    with pytest.raises(ValueError):
        q.log_mel_spectrogram(np.zeros(100, dtype=np.float32))


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("M,D", [(1, 256), (7, 1024), (390, 1024), (33, 2048)])
def test_layernorm(dev, M, D):
    from qwen3_asr_mlx_amd import ops
    x = rnd((M, D), dev, 1, 2.0, torch.float32) + 0.5
    w = rnd((D,), dev, 2, 1.0, torch.float32)
    b = rnd((D,), dev, 3, 1.0, torch.float32)
    y = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    ops.layernorm(x, w, b, y, M, D, 1e-5)
    ref = F.layer_norm(x.cpu(), (D,), w.cpu(), b.cpu(), 1e-5)
    # bf16 output: half an ulp of the bf16 result plus float32 reduction-order noise
    torch.testing.assert_close(y.float().cpu(), bfr(ref), atol=2e-2, rtol=8e-3)
    assert (y.float().cpu() - ref).abs().max() < 4e-2


@pytest.mark.parametrize("M,D", [(1, 512), (5, 2048), (408, 2048), (9, 4096)])
def test_rmsnorm_and_row_gather(dev, M, D):
    import oracle.decoder as od
    from qwen3_asr_mlx_amd import ops
    x = rnd((M, D), dev, 4, 1.5)
    w = rnd((D,), dev, 5, 1.0)
    y = torch.empty_like(x)
    ops.rmsnorm(x, w, y, M, D, 1e-6)
    ref = od.rms_norm(x.float().cpu(), w.float().cpu(), 1e-6)
    torch.testing.assert_close(y.float().cpu(), ref, atol=0, rtol=8e-3)   # <= 1 bf16 ulp
    idx = torch.tensor([M - 1, 0], dtype=torch.int32, device=dev)
    y2 = torch.empty((2, D), dtype=torch.bfloat16, device=dev)
    ops.rmsnorm(x, w, y2, 2, D, 1e-6, row_idx=idx)
    assert torch.equal(y2[0], y[M - 1]) and torch.equal(y2[1], y[0])


# ----------------------------------------------------------------------------- GEMM
def _gemm_ref(A, W, bias=None, act=0, res=None, res_kind=0, res_mod=0):
    c = A.float().cpu() @ W.float().cpu().T
    if bias is not None:
        c = c + bias.cpu()
    if act == 1:
        c = F.gelu(c)
    if res_kind == 1:
        r = res.cpu()
        c = c + (r[torch.arange(c.shape[0]) % res_mod] if res_mod else r)
    elif res_kind == 2:
        c = bfr(c) + res.float().cpu()
    return c


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (130, 256, 192), (257, 480, 4320), (1000, 384, 1024), (64, 1024, 7680)])
def test_gemm_plain_and_edges(dev, M, N, K):
    from qwen3_asr_mlx_amd import ops
    A = rnd((M, K), dev, 10, 1.0)
    W = rnd((N, K), dev, 11, 1.0 / math.sqrt(K))
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm(A, W, M, N, K, out, out_fp32=True)
    ref = _gemm_ref(A, W)
    torch.testing.assert_close(out.cpu(), ref, atol=2e-3, rtol=1e-3)      # float32 accumulation order only
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, M, N, K, outb)
    torch.testing.assert_close(outb.float().cpu(), bfr(ref), atol=2e-2, rtol=8e-3)


def test_gemm_asymmetric_identity(dev):
    """A = I against an asymmetric W catches a transposed C write or operand swap exactly."""
    from qwen3_asr_mlx_amd import ops
    K = 256
    A = torch.eye(K, dtype=torch.bfloat16, device=dev)
    W = (torch.arange(384 * K, dtype=torch.float32).reshape(384, K) % 251 - 125).to(torch.bfloat16).to(dev)
    out = torch.empty((K, 384), dtype=torch.float32, device=dev)
    ops.gemm(A, W, K, 384, K, out, out_fp32=True)
    assert torch.equal(out.cpu(), W.float().cpu().T)


def test_gemm_epilogues(dev):
    from qwen3_asr_mlx_amd import ops
    M, N, K = 300, 512, 256
    A = rnd((M, K), dev, 20)
    W = rnd((N, K), dev, 21, 1.0 / math.sqrt(K))
    bias = rnd((N,), dev, 22, 0.5, torch.float32)
    # bias + GELU -> bf16
    o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(A, W, M, N, K, o, bias=bias, act=1)
    torch.testing.assert_close(o.float().cpu(), bfr(_gemm_ref(A, W, bias, 1)), atol=2e-2, rtol=8e-3)
    # bias + fp32 residual, in place
    h = rnd((M, N), dev, 23, 1.0, torch.float32)
    h0 = h.clone()
    ops.gemm(A, W, M, N, K, h, bias=bias, res=h, res_kind=1, ldres=N, out_fp32=True)
    torch.testing.assert_close(h.cpu(), _gemm_ref(A, W, bias, 0, h0, 1), atol=2e-3, rtol=1e-3)
    # bf16 residual with MLX rounding
    hb = rnd((M, N), dev, 24)
    hb0 = hb.clone()
    ops.gemm(A, W, M, N, K, hb, res=hb, res_kind=2, ldres=N)
    torch.testing.assert_close(hb.float().cpu(), bfr(_gemm_ref(A, W, None, 0, hb0, 2)), atol=2e-2, rtol=8e-3)
    # residual indexed modulo 13 + row map with dropped rows (conv_out + PE + pad strip)
    pe = rnd((13, N), dev, 25, 1.0, torch.float32)
    rm = torch.tensor([(i * 7) % M if i % 5 else -1 for i in range(M)], dtype=torch.int32)
    keep = rm >= 0
    rm[keep] = torch.arange(int(keep.sum()), dtype=torch.int32)
    out = torch.zeros((int(keep.sum()), N), dtype=torch.float32, device=dev)
    ops.gemm(A, W, M, N, K, out, res=pe, res_kind=1, ldres=N, res_mod=13, out_fp32=True, row_map=rm.to(dev))
    torch.testing.assert_close(out.cpu(), _gemm_ref(A, W, None, 0, pe, 1, 13)[keep], atol=2e-3, rtol=1e-3)


def test_gemm_swiglu(dev):
    from qwen3_asr_mlx_amd import ops
    M, I, K = 200, 256, 512
    A = rnd((M, K), dev, 30)
    gate = rnd((I, K), dev, 31, 2.0 / math.sqrt(K))
    up = rnd((I, K), dev, 32, 2.0 / math.sqrt(K))
    gu = torch.stack([gate.reshape(I // 16, 16, K), up.reshape(I // 16, 16, K)], 1).reshape(2 * I, K).contiguous()
    o = torch.empty((M, I), dtype=torch.bfloat16, device=dev)
    ops.gemm(A, gu, M, 2 * I, K, o, act=2)
    g, u = bfr(A.float().cpu() @ gate.float().cpu().T), bfr(A.float().cpu() @ up.float().cpu().T)
    ref = swiglu_ref(g, u)
    torch.testing.assert_close(o.float().cpu(), ref, atol=2e-2, rtol=1.6e-2)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 512, 384), (1000, 768, 1024), (2050, 1280, 2048), (515, 256, 6144),
                                   (777, 992, 512),               # ragged N as well (last tile column 224 wide)
                                   (4396, 4096, 1024),            # 288 tiles: the 8-phase kernel walks two tiles per workgroup
                                   (12480, 1024, 1024),           # the encoder's out projection: 196 tiles of 256 rows, 224 of 224
                                   (70, 128, 64), (130, 192, 192)])   # fewer K tiles than the 64^2 kernel keeps in flight (8-phase n/a: 128^2 twice)
def test_gemm_kernels_agree_bitwise(dev, M, N, K):
    """The 64x64, the 128x128 and the 8-phase LDS-DMA kernel (256x256, 224x256 and 208x256 = 112 + 96 row tiles) accumulate every output in the same order
    (k ascending, one MFMA chain): bit-identical results, plain, fp32-out and through the epilogues (bias+GELU, bf16 residual, SwiGLU),
    including ragged M edges -- which is what makes a clip's result independent of the batch it is decoded in."""
    from qwen3_asr_mlx_amd import _lib, ops
    A = rnd((M, K), dev, 40, 1.0)
    W = rnd((N, K), dev, 41, 1.0 / math.sqrt(K))
    bias = rnd((N,), dev, 42, 0.5, torch.float32)
    hb = rnd((M, N), dev, 43)
    h0 = rnd((M, N), dev, 44, 1.0, torch.float32)
    pe = rnd((13, N), dev, 45, 1.0, torch.float32)
    rm = torch.tensor([i if i % 5 else -1 for i in range(M)], dtype=torch.int32)
    keep = rm >= 0
    n_keep = int(keep.sum())
    rm[keep] = torch.arange(n_keep, dtype=torch.int32)
    rm = rm.to(dev)
    outs = {}
    try:
        for mode in (128, 8, 64, 224, 208):
            _lib.call("q3asr_gemm_force_kernel", mode)
            o32 = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
            ops.gemm(A, W, M, N, K, o32, out_fp32=True)
            og = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, M, N, K, og, bias=bias, act=1)
            orr = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, M, N, K, orr, res=hb, res_kind=2, ldres=N)
            osw = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, M, N, K, osw, act=2)
            h32 = h0.clone()                                                  # bias + fp32 residual in place (encoder out / fc2)
            ops.gemm(A, W, M, N, K, h32, bias=bias, res=h32, res_kind=1, ldres=N, out_fp32=True)
            opk = torch.zeros((n_keep, N), dtype=torch.float32, device=dev)   # modulo-row residual + row map (conv_out + PE + pad strip)
            ops.gemm(A, W, M, N, K, opk, res=pe, res_kind=1, ldres=N, res_mod=13, out_fp32=True, row_map=rm)
            outs[mode] = (o32, og, orr, osw, h32, opk)
    finally:
        _lib.call("q3asr_gemm_force_kernel", 0)
    torch.testing.assert_close(outs[8][0].cpu(), _gemm_ref(A, W), atol=2e-3, rtol=1e-3)
    for mode in (8, 64, 224, 208):
        for a, b in zip(outs[128], outs[mode]):
            assert torch.equal(a, b), mode


def test_gemm_8phase_direct_epilogue_fallback(dev):
    """N % 8 == 4: the whole-row (LDS-transposed) epilogue's precondition fails and the 8-phase kernel stores straight from
    the accumulator layout; results still equal the 128x128 kernel bit for bit."""
    from qwen3_asr_mlx_amd import _lib, ops
    M, N, K = 530, 996, 512
    A = rnd((M, K), dev, 46, 1.0)
    W = rnd((N, K), dev, 47, 1.0 / math.sqrt(K))
    bias = rnd((N,), dev, 48, 0.5, torch.float32)
    h0 = rnd((M, N), dev, 49, 1.0, torch.float32)
    outs = {}
    try:
        for mode in (128, 8):
            _lib.call("q3asr_gemm_force_kernel", mode)
            og = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            ops.gemm(A, W, M, N, K, og, bias=bias, act=1)
            h32 = h0.clone()
            ops.gemm(A, W, M, N, K, h32, bias=bias, res=h32, res_kind=1, ldres=N, out_fp32=True)
            outs[mode] = (og, h32)
    finally:
        _lib.call("q3asr_gemm_force_kernel", 0)
    torch.testing.assert_close(outs[8][1].cpu(), _gemm_ref(A, W, bias, 0, h0, 1), atol=2e-3, rtol=1e-3)
    for a, b in zip(outs[128], outs[8]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("mode", [8, 224, 208])
def test_gemm_8phase_is_stable_under_load(dev, mode):
    """Race screen for the 8-phase pipeline (LDS-DMA data is ordered only by counted waits + barriers): 30 back-to-back
    launches of a chip-filling shape, each compared bit for bit with the 128x128 kernel's result; both tile heights."""
    from qwen3_asr_mlx_amd import _lib, ops
    M, N, K = 4096, 4096 + 256, 2048 + 128
    A = rnd((M, K), dev, 50, 1.0)
    W = rnd((N, K), dev, 51, 1.0 / math.sqrt(K))
    try:
        _lib.call("q3asr_gemm_force_kernel", 128)
        ref = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ops.gemm(A, W, M, N, K, ref)
        _lib.call("q3asr_gemm_force_kernel", mode)
        outs = [torch.empty((M, N), dtype=torch.bfloat16, device=dev) for _ in range(30)]
        for o in outs:
            ops.gemm(A, W, M, N, K, o)
    finally:
        _lib.call("q3asr_gemm_force_kernel", 0)
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
    assert not bad, bad


@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (1000, 768, 1024), (4396, 2048, 2048), (130, 1024, 6144)])
def test_gemm_fp8_on_the_fp8_mfma(dev, M, N, K):
    """q3asr_gemm_fp8: per-token quantised activations x per-row quantised weights on v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales) against
    the same product in float64 of the dequantised operands (exact inputs: only the fp32 accumulation order differs), plain,
    with the bf16 residual and through SwiGLU; the quantisers against torch's e4m3 conversion."""
    from qwen3_asr_mlx_amd import ops
    x = rnd((M, K), dev, 70, 1.0)
    x[3] *= 37.0                                                       # rows with very different scales
    x[5] = 0                                                           # an all-zero row: scale 1
    W = rnd((N, K), dev, 71, 1.0 / math.sqrt(K))
    q8, sw, w_eff = ops.quantize_fp8_rows(W)
    a8 = torch.empty((M, K), dtype=torch.uint8, device=dev)
    sa = torch.empty((M,), dtype=torch.float32, device=dev)
    ops.quantize_rows_fp8(x, a8, sa, M, K)
    amax = x.float().abs().amax(1)
    want_s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    torch.testing.assert_close(sa, want_s, rtol=1e-6, atol=0)
    want_q = (x.float() * (1.0 / sa)[:, None]).to(torch.float8_e4m3fn)
    assert torch.equal(a8.view(torch.float8_e4m3fn).float(), want_q.float())
    a_deq = a8.view(torch.float8_e4m3fn).double() * sa.double()[:, None]
    ref = (a_deq @ w_eff.double().T).cpu()                             # w_eff = scale * fp8 exactly
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8(a8, sa, q8, sw, M, N, K, out)
    torch.testing.assert_close(out.float().cpu(), bfr(ref.float()), atol=2e-2, rtol=8e-3)
    hb = rnd((M, N), dev, 72)
    out2 = torch.empty_like(out)
    ops.gemm_fp8(a8, sa, q8, sw, M, N, K, out2, res=hb)
    torch.testing.assert_close(out2.float().cpu(), bfr(bfr(ref.float()) + hb.float().cpu()), atol=3e-2, rtol=8e-3)
    # SwiGLU over 16-row interleaved (gate, up) weights, as q3asr_gemm_bf16 act=2
    I = N // 2
    osw = torch.empty((M, I), dtype=torch.bfloat16, device=dev)
    ops.gemm_fp8(a8, sa, q8, sw, M, N, K, osw, act=2)
    r3 = bfr(ref.float()).reshape(M, I // 16, 2, 16)
    g, u = r3[:, :, 0].reshape(M, I), r3[:, :, 1].reshape(M, I)
    # the reference's three bf16 roundings (swiglu_ref) pass a one-ulp difference of gate / up -- fp32 MFMA accumulation against this
    # test's float64 product -- on as up to ~5 ulps of the output: 4 % relative
    torch.testing.assert_close(osw.float().cpu(), swiglu_ref(g, u), atol=4e-2, rtol=4e-2)
    # both tile heights of the 8-phase kernel give the same bits (one k-ascending chain per output)
    from qwen3_asr_mlx_amd import _lib
    by_tile = {}
    try:
        for mode in (8, 224, 208):
            _lib.call("q3asr_gemm_force_kernel", mode)
            o_a, o_b, o_c = torch.empty_like(out), torch.empty_like(out), torch.empty_like(osw)
            ops.gemm_fp8(a8, sa, q8, sw, M, N, K, o_a)
            ops.gemm_fp8(a8, sa, q8, sw, M, N, K, o_b, res=hb)
            ops.gemm_fp8(a8, sa, q8, sw, M, N, K, o_c, act=2)
            by_tile[mode] = (o_a, o_b, o_c)
    finally:
        _lib.call("q3asr_gemm_force_kernel", 0)
    for x8, x224, x208, auto in zip(by_tile[8], by_tile[224], by_tile[208], (out, out2, osw)):
        assert torch.equal(x8, x224) and torch.equal(x8, x208) and torch.equal(x8, auto)
    # RMSNorm fused quantiser == rmsnorm_bf16 followed by the plain quantiser
    if K in (1024, 2048):
        wn = rnd((K,), dev, 73) + 1
        y = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
        ops.rmsnorm(x, wn, y, M, K, 1e-6)
        b8, sb = torch.empty_like(a8), torch.empty_like(sa)
        ops.quantize_rows_fp8(y, b8, sb, M, K)
        c8, sc = torch.empty_like(a8), torch.empty_like(sa)
        ops.rmsnorm_fp8(x, wn, c8, sc, M, K, 1e-6)
        assert torch.equal(b8, c8) and torch.equal(sb, sc)


def test_gemm_rejects_bad_shapes(dev):
    from qwen3_asr_mlx_amd import ops
    A = rnd((4, 60), dev, 1)
    W = rnd((8, 60), dev, 2)
    with pytest.raises(ValueError):
        ops.gemm(A, W, 4, 8, 60, torch.empty((4, 8), dtype=torch.bfloat16, device=dev))   # K % 8 != 0


# ----------------------------------------------------------------------------- conv stem
def test_conv1_gelu(dev):
    from qwen3_asr_mlx_amd import ops
    C1, B, Fmax = 32, 2, 230
    mel = rnd((B, 128, Fmax), dev, 40, 1.0, torch.float32)
    w = rnd((C1, 3, 3, 1), dev, 41, 0.3)                       # MLX (O, kH=freq, kW=time, I)
    b = rnd((C1,), dev, 42, 0.2, torch.float32)
    chunks = [(0, 0), (0, 100), (0, 200), (1, 0), (1, 100)]    # (utterance, first frame); 200.. is ragged (30 frames)
    cb = torch.tensor([c[0] for c in chunks], dtype=torch.int32, device=dev)
    cf = torch.tensor([c[1] for c in chunks], dtype=torch.int32, device=dev)
    out = torch.empty((len(chunks), 50, 64, C1), dtype=torch.bfloat16, device=dev)
    w9 = w.float()[..., 0].permute(0, 2, 1).contiguous()
    ops.call("q3asr_conv1_gelu", ops.ptr(mel), Fmax, ops.ptr(cb), ops.ptr(cf), len(chunks), ops.ptr(w9), ops.ptr(b),
             ops.ptr(out), C1, ops.sp())
    melc = mel.cpu()
    for i, (ub, f0) in enumerate(chunks):
        x = torch.zeros(128, 100)
        n = max(0, min(100, Fmax - f0))
        x[:, :n] = melc[ub, :, f0: f0 + n]
        ref = F.gelu(F.conv2d(x[None, None], w.float().cpu().permute(0, 3, 1, 2), b.cpu(), stride=2, padding=1))[0]
        got = out[i].float().cpu().permute(2, 1, 0)            # [t][f][c] -> [c][f][t]
        torch.testing.assert_close(got, bfr(ref), atol=1e-2, rtol=8e-3)


@pytest.mark.parametrize("imgs,H,W_,C", [(3, 50, 64, 32), (2, 25, 32, 32), (5, 13, 16, 480)])
def test_conv3x3s2_implicit_gemm(dev, imgs, H, W_, C):
    from qwen3_asr_mlx_amd import ops
    x = rnd((imgs, H, W_, C), dev, 50)
    w = rnd((C, 3, 3, C), dev, 51, 1.0 / math.sqrt(9 * C))
    b = rnd((C,), dev, 52, 0.3, torch.float32)
    Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    out = torch.empty((imgs, Ho, Wo, C), dtype=torch.bfloat16, device=dev)
    ops.conv3x3s2(x, imgs, H, W_, C, ops.conv_weight_chunked(w), C, b, 1, out)
    ref = F.gelu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), b.cpu(), stride=2, padding=1))
    torch.testing.assert_close(out.float().cpu(), bfr(ref.permute(0, 2, 3, 1)), atol=2e-2, rtol=8e-3)


@pytest.mark.parametrize("imgs,H,W_,C,Co", [(40, 25, 32, 480, 480), (9, 64, 50, 96, 256), (33, 13, 16, 64, 512)])
def test_conv3x3s2_kernels_agree_bitwise(dev, imgs, H, W_, C, Co):
    """The implicit-GEMM conv through the 128x128 and the 8-phase kernels (im2col addresses generated in the DMA's
    source address, padding taps and the K tail read from a zero line): identical bits; checked against torch too."""
    from qwen3_asr_mlx_amd import _lib, ops
    x = rnd((imgs, H, W_, C), dev, 53)
    w = rnd((Co, 3, 3, C), dev, 54, 1.0 / math.sqrt(9 * C))
    b = rnd((Co,), dev, 55, 0.3, torch.float32)
    Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    outs = {}
    try:
        for mode in (128, 8):
            _lib.call("q3asr_gemm_force_kernel", mode)
            o = torch.empty((imgs, Ho, Wo, Co), dtype=torch.bfloat16, device=dev)
            ops.conv3x3s2(x, imgs, H, W_, C, ops.conv_weight_chunked(w), Co, b, 1, o)
            outs[mode] = o
    finally:
        _lib.call("q3asr_gemm_force_kernel", 0)
    ref = F.gelu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), b.cpu(), stride=2, padding=1))
    torch.testing.assert_close(outs[8].float().cpu(), bfr(ref.permute(0, 2, 3, 1)), atol=2e-2, rtol=8e-3)
    assert torch.equal(outs[8], outs[128])


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale, causal):
    """q (Tq,D) k,v (Tk,D) float32; softmax fp32, P rounded to bf16 for PV, sum unrounded."""
    s = (q @ k.T) * scale
    if causal:
        s = s.masked_fill(torch.arange(k.shape[0])[None] > torch.arange(q.shape[0])[:, None], float("-inf"))
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    return (bfr(p) @ v) / p.sum(-1, keepdim=True)


def test_attn_fwd_encoder_windows(dev):
    from qwen3_asr_mlx_amd import ops
    nh, hd = 4, 64
    d = nh * hd
    lens = [104, 104, 27, 1, 65, 104]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).tolist()
    N = sum(lens)
    qkv = rnd((N, 3 * d), dev, 60, 1.0)
    out = torch.zeros((N, d), dtype=torch.bfloat16, device=dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    kv_off = torch.tensor([s * 3 * d for s in starts], dtype=torch.int64, device=dev)
    ops.attn_fwd(ops.ptr(qkv), 3 * d, ops.off_ptr(qkv, d), ops.off_ptr(qkv, 2 * d), kv_off, hd, 3 * d, out, d,
                 i32(starts), i32(lens), len(lens), max(lens), nh, nh, hd, False, hd ** -0.5)
    c = qkv.float().cpu()
    for s, n in zip(starts, lens):
        for h in range(nh):
            q_, k_, v_ = (c[s: s + n, o + h * hd: o + (h + 1) * hd] for o in (0, d, 2 * d))
            ref = _attn_ref(q_, k_, v_, hd ** -0.5, False)
            torch.testing.assert_close(out[s: s + n, h * hd: (h + 1) * hd].float().cpu(), bfr(ref), atol=2e-2, rtol=1.6e-2)


@pytest.mark.parametrize("lens", [[1], [64], [65, 130, 7], [408, 83]])
def test_attn_fwd_decoder_causal_gqa(dev, lens):
    from qwen3_asr_mlx_amd import ops
    nh, nkv, hd, Lmax = 4, 2, 128, 448
    B = len(lens)
    M = sum(lens)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).tolist()
    q = rnd((M, nh * hd), dev, 70)
    kc = rnd((B, nkv, Lmax, hd), dev, 71)
    vc = rnd((B, nkv, Lmax, hd), dev, 72)
    out = torch.zeros((M, nh * hd), dtype=torch.bfloat16, device=dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    kv_off = torch.tensor([b * nkv * Lmax * hd for b in range(B)], dtype=torch.int64, device=dev)
    ops.attn_fwd(ops.ptr(q), nh * hd, ops.ptr(kc), ops.ptr(vc), kv_off, Lmax * hd, hd, out, nh * hd, i32(starts), i32(lens),
                 B, max(lens), nh, nkv, hd, True, hd ** -0.5)
    qc, kcc, vcc = q.float().cpu(), kc.float().cpu(), vc.float().cpu()
    for b, (s, n) in enumerate(zip(starts, lens)):
        for h in range(nh):
            ref = _attn_ref(qc[s: s + n, h * hd: (h + 1) * hd], kcc[b, h // 2, :n], vcc[b, h // 2, :n], hd ** -0.5, True)
            torch.testing.assert_close(out[s: s + n, h * hd: (h + 1) * hd].float().cpu(), bfr(ref), atol=2e-2, rtol=1.6e-2)


def test_attn_online_softmax_rescale_branch(dev):
    """Force the running max to jump at a later key tile (a spike in tile 3) -- the rescale path."""
    from qwen3_asr_mlx_amd import ops
    nh, nkv, hd, T = 2, 1, 128, 256
    q = rnd((T, nh * hd), dev, 73, 0.3)
    kc = rnd((1, nkv, T, hd), dev, 74, 0.3)
    vc = rnd((1, nkv, T, hd), dev, 75)
    kc[0, 0, 200] = q[255, :hd] * 6.0                           # key 200 matches the last query strongly
    out = torch.zeros((T, nh * hd), dtype=torch.bfloat16, device=dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    ops.attn_fwd(ops.ptr(q), nh * hd, ops.ptr(kc), ops.ptr(vc), torch.zeros(1, dtype=torch.int64, device=dev), T * hd, hd,
                 out, nh * hd, i32([0]), i32([T]), 1, T, nh, nkv, hd, True, hd ** -0.5)
    for h in range(nh):
        ref = _attn_ref(q.float().cpu()[:, h * hd: (h + 1) * hd], kc.float().cpu()[0, 0], vc.float().cpu()[0, 0], hd ** -0.5, True)
        torch.testing.assert_close(out[:, h * hd: (h + 1) * hd].float().cpu(), bfr(ref), atol=2e-2, rtol=1.6e-2)


# ----------------------------------------------------------------------------- rope / kv / embed
@pytest.mark.parametrize("nh,nkv", [(4, 2), (16, 8), (10, 5)])
def test_qk_norm_rope_kv_bf16_and_slab(dev, nh, nkv):
    """bf16 rows go through the token-per-workgroup kernel (16-byte accesses, shared cos/sin), fp32 slabs through the
    wave-per-head kernel: both against the oracle, and bit-identical to each other."""
    import oracle.decoder as od
    from qwen3_asr_mlx_amd import ops
    hd, Lmax, B = 128, 64, 3
    rows_b = [0, 0, 0, 1, 2, 2]
    rows_p = [0, 1, 2, 40, 7, 63]
    M = len(rows_b)
    qkvd = (nh + 2 * nkv) * hd
    qkv = rnd((M, qkvd), dev, 90, 2.0)
    qn, kn = rnd((hd,), dev, 91, 1.0) + 1, rnd((hd,), dev, 92, 1.0) + 1
    theta = 1e6
    inv = (theta ** (-torch.arange(64, dtype=torch.float32) / 64)).to(dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)

    rb_t, rp_t = i32(rows_b), i32(rows_p)                       # keep alive: the launch only sees raw pointers

    def run(src, is_slab, n_slabs, stride):
        qo = torch.zeros((M, nh * hd), dtype=torch.bfloat16, device=dev)
        kc = torch.zeros((B, nkv, Lmax, hd), dtype=torch.bfloat16, device=dev)
        vc = torch.zeros_like(kc)
        ops.call("q3asr_qk_norm_rope_kv", ops.ptr(src), is_slab, qkvd, n_slabs, stride, ops.ptr(rb_t), ops.ptr(rp_t),
                 ops.ptr(qn), ops.ptr(kn), ops.ptr(inv), 1e-6, ops.ptr(qo), ops.ptr(kc), ops.ptr(vc), M, nh, nkv, hd, Lmax, ops.sp())
        return qo, kc, vc

    qo, kc, vc = run(qkv, 0, 0, 0)
    c = qkv.float().cpu()
    for m in range(M):
        pos = torch.tensor([rows_p[m]])
        qr = od.rope(od.rms_norm(c[m, : nh * hd].reshape(nh, 1, hd), qn.float().cpu(), 1e-6), pos, theta)
        kr = od.rope(od.rms_norm(c[m, nh * hd: (nh + nkv) * hd].reshape(nkv, 1, hd), kn.float().cpu(), 1e-6), pos, theta)
        # <= 1 bf16 ulp (device sincos vs torch) on O(1) values
        torch.testing.assert_close(qo[m].float().cpu().reshape(nh, 1, hd), qr, atol=4e-2, rtol=1.6e-2)
        torch.testing.assert_close(kc[rows_b[m], :, rows_p[m]].float().cpu()[:, None], kr, atol=4e-2, rtol=1.6e-2)
        assert torch.equal(vc[rows_b[m], :, rows_p[m]].cpu().reshape(-1), qkv[m, (nh + nkv) * hd:].cpu())
    # norm-preservation (reference test_decoder.py:107-126): rotation keeps the per-head norm
    n_in = od.rms_norm(c[:, : nh * hd].reshape(M, nh, hd), qn.float().cpu(), 1e-6).norm(dim=-1)
    torch.testing.assert_close(qo.float().cpu().reshape(M, nh, hd).norm(dim=-1), n_in, atol=0, rtol=2e-2)
    # slab input: three fp32 partials whose sum rounds to the same bf16 values
    parts = torch.stack([c * 0.5, c * 0.25, c * 0.25]).to(dev)
    qo2, kc2, vc2 = run(parts, 1, 3, M * qkvd)
    assert torch.equal(qo2, qo) and torch.equal(kc2, kc) and torch.equal(vc2, vc)


def test_embed_merge(dev):
    from qwen3_asr_mlx_amd import ops
    V, H = 300, 512
    E = rnd((V, H), dev, 100)
    enc = rnd((5, H), dev, 101, 1.0, torch.float32)
    ids = torch.tensor([3, 7, 299, 299, 0, 299, 299, 299, 8], dtype=torch.int32, device=dev)
    src = torch.tensor([-1, -1, 0, 1, -1, 2, 3, 4, -1], dtype=torch.int32, device=dev)
    out = torch.empty((9, H), dtype=torch.bfloat16, device=dev)
    ops.call("q3asr_embed_merge", ops.ptr(ids), ops.ptr(src), ops.ptr(E), V, ops.ptr(enc), 5, ops.ptr(out), 9, H, ops.sp())
    for m in range(9):
        want = enc[src[m]].to(torch.bfloat16) if src[m] >= 0 else E[ids[m]]
        assert torch.equal(out[m], want)
    # indices the host cannot validate (device-resident): ids outside [0, V) and encoder rows >= n_enc give zero rows,
    # never an out-of-bounds read (0x7fffffff is the greedy sampler's "no maximum" sentinel)
    bad_ids = torch.tensor([3, V, -1, 0x7fffffff, 299, 299], dtype=torch.int32, device=dev)
    bad_src = torch.tensor([-1, -1, -1, -1, 5, 4], dtype=torch.int32, device=dev)
    out2 = torch.full((6, H), 7.0, dtype=torch.bfloat16, device=dev)
    ops.call("q3asr_embed_merge", ops.ptr(bad_ids), ops.ptr(bad_src), ops.ptr(E), V, ops.ptr(enc), 5, ops.ptr(out2), 6, H, ops.sp())
    assert torch.equal(out2[0], E[3]) and torch.equal(out2[5], enc[4].to(torch.bfloat16))
    assert not out2[1:5].any()


# ----------------------------------------------------------------------------- decode step in streaming layouts (W16 / X16)
def test_w16_x16_layout_roundtrip(dev):
    from qwen3_asr_mlx_amd import ops
    X = rnd((21, 256), dev, 129)
    assert torch.equal(ops.from_x16(ops.to_x16(X), 21), X)
    W = torch.arange(32 * 64, dtype=torch.float32).reshape(32, 64).to(torch.bfloat16).to(dev)
    w16 = ops.to_w16(W)
    assert w16.shape == (2, 2, 4, 16, 8)
    rt, ks, g, li, j = 1, 1, 2, 5, 3
    assert w16[rt, ks, g, li, j] == W[16 * rt + li, 32 * ks + 8 * g + j]


@pytest.mark.parametrize("M,N,K,S,waves", [(1, 1024, 512, 4, 4), (7, 4096, 2048, 4, 4), (32, 2048, 6144, 8, 4),
                                           (32, 2048, 2048, 8, 4), (33, 512, 1024, 1, 2), (64, 2048, 2048, 2, 4),
                                           (16, 48, 256, 1, 4), (20, 2048, 6144, 3, 1)])
def test_skinny3_slab_and_rowmajor(dev, M, N, K, S, waves):
    from qwen3_asr_mlx_amd import ops
    X = rnd((M, K), dev, 130)
    W = rnd((N, K), dev, 131, 1.0 / math.sqrt(K))
    x16, w16 = ops.to_x16(X), ops.to_w16(W)
    ref = X.float().cpu() @ W.float().cpu().T
    for rt in (1, 2, 4):
        slab = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny3(x16, M, w16, N, K, 0, S, waves, slab=slab, rt=rt)
        torch.testing.assert_close(slab.sum(0).cpu(), ref, atol=2e-3, rtol=1e-3)
        if S == 1:
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            ops.skinny3(x16, M, w16, N, K, 1, 1, waves, out=out, rt=rt)
            torch.testing.assert_close(out.float().cpu(), bfr(ref), atol=2e-2, rtol=8e-3)
        if (K // S) % (32 * waves) == 0:                      # waves of a workgroup split K, LDS reduce
            slab = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
            ops.skinny3(x16, M, w16, N, K, 0, S, waves, slab=slab, rt=rt, ksplit=True)
            torch.testing.assert_close(slab.sum(0).cpu(), ref, atol=2e-3, rtol=1e-3)


def test_skinny3_identity_layout(dev):
    from qwen3_asr_mlx_amd import ops
    K = 1024
    X = torch.eye(K, dtype=torch.bfloat16, device=dev)[300:348].contiguous()
    W = (torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 241 - 120).to(torch.bfloat16).to(dev)
    out = torch.empty((48, 192), dtype=torch.bfloat16, device=dev)
    ops.skinny3(ops.to_x16(X), 48, ops.to_w16(W), 192, K, 1, out=out)
    assert torch.equal(out.float().cpu(), W.float().cpu()[:, 300:348].T)


@pytest.mark.parametrize("M,I,K,waves", [(32, 6144, 2048, 2), (5, 1024, 512, 4), (64, 256, 256, 2)])
def test_skinny3_swiglu_to_x16(dev, M, I, K, waves):
    from qwen3_asr_mlx_amd import ops
    A = rnd((M, K), dev, 132)
    gate = rnd((I, K), dev, 133, 2.0 / math.sqrt(K))
    up = rnd((I, K), dev, 134, 2.0 / math.sqrt(K))
    gu = torch.stack([gate.reshape(I // 16, 16, K), up.reshape(I // 16, 16, K)], 1).reshape(2 * I, K).contiguous()
    MT = ops.mt_of(M)
    g, u = bfr(A.float().cpu() @ gate.float().cpu().T), bfr(A.float().cpu() @ up.float().cpu().T)
    for rt, ksplit in ((2, False), (4, False), (2, True), (4, True)):
        o16 = torch.zeros((I // 32, MT, 64, 8), dtype=torch.bfloat16, device=dev)
        ops.skinny3(ops.to_x16(A), M, ops.to_w16(gu), 2 * I, K, 2, 1, waves, out=o16, rt=rt, ksplit=ksplit)
        got = ops.from_x16(o16, M).float().cpu()
        torch.testing.assert_close(got, swiglu_ref(g, u), atol=2e-2, rtol=1.6e-2)


@pytest.mark.parametrize("M,N,K,S_in,S,rt,ksplit,epi,fp8", [(1, 4096, 2048, 8, 8, 1, False, 0, False), (2, 1024, 512, 3, 2, 2, False, 0, False),
                                                              (1, 12288, 2048, 8, 1, 2, True, 2, False), (2, 512, 1024, 6, 1, 4, True, 2, False),
                                                              (1, 4096, 2048, 6, 4, 1, False, 0, True), (1, 2048, 1024, 1, 1, 2, True, 2, True)])
def test_skinny3_fused_norm_equals_separate_kernels(dev, M, N, K, S_in, S, rt, ksplit, epi, fp8):
    """q3asr_skinny3_gemm_norm == q3asr_slab_resnorm_x16 followed by q3asr_skinny3_gemm, bit for bit: the residual
    stream it writes, and the GEMM output (slabs or SwiGLU activations)."""
    from qwen3_asr_mlx_amd import ops
    in_slab = rnd((S_in, M, K), dev, 170, 0.5, torch.float32)
    h = rnd((M, K), dev, 171)
    w = rnd((K,), dev, 172) + 1
    W = rnd((N, K), dev, 173, 1.5 / math.sqrt(K))
    if fp8:
        q8, scale, _ = ops.quantize_fp8_rows(W)
        wt = ops.to_w16f8(q8)
    else:
        wt, scale = ops.to_w16(W), None
    # separate kernels
    h1 = torch.empty_like(h)
    x16 = torch.zeros((K // 32, 1, 64, 8), dtype=torch.bfloat16, device=dev)
    ops.call("q3asr_slab_resnorm_x16", ops.ptr(in_slab), S_in, M * K, ops.ptr(h), ops.ptr(h1), ops.ptr(w), ops.ptr(x16), M, K, 1e-6, ops.sp())
    # fused
    h2 = torch.zeros_like(h)
    if epi == 0:
        ref = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny3(x16, M, wt, N, K, 0, S, 4, slab=ref, rt=rt, ksplit=ksplit, scale=scale)
        got = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny3_norm(in_slab, S_in, M * K, h, h2, w, 1e-6, M, wt, N, K, 0, S, slab=got, rt=rt, ksplit=ksplit, scale=scale)
    else:
        ref = torch.zeros((N // 2 // 32, 1, 64, 8), dtype=torch.bfloat16, device=dev)
        ops.skinny3(x16, M, wt, N, K, 2, 1, 4, out=ref, rt=rt, ksplit=ksplit, scale=scale)
        got = torch.zeros_like(ref)
        ops.skinny3_norm(in_slab, S_in, M * K, h, h2, w, 1e-6, M, wt, N, K, 2, 1, out=got, rt=rt, ksplit=ksplit, scale=scale)
        ref, got = ops.from_x16(ref, M), ops.from_x16(got, M)
    assert torch.equal(h2, h1)
    assert torch.equal(got, ref)
    with pytest.raises(ValueError):                            # h_in == h_out is refused: neighbours still read h_in
        ops.skinny3_norm(in_slab, S_in, M * K, h, h, w, 1e-6, M, wt, N, K, epi, S if epi == 0 else 1,
                         slab=torch.empty((S, M, N), dtype=torch.float32, device=dev) if epi == 0 else None,
                         out=None if epi == 0 else torch.zeros((N // 2 // 32, 1, 64, 8), dtype=torch.bfloat16, device=dev), rt=rt, ksplit=ksplit, scale=scale)


@pytest.mark.parametrize("M,N,K,S_in,geo,epi,fp8", [(1, 4096, 2048, 4, (2, 2, 8), 0, False), (2, 1024, 512, 3, (2, 2, 8), 0, False),
                                                    (1, 12288, 2048, 4, (4, 4, 16), 2, False), (2, 2048, 512, 6, (2, 2, 8), 2, False),
                                                    (1, 4096, 2048, 6, (2, 4, 8), 0, True), (2, 12288, 2048, 1, (4, 4, 16), 2, True),
                                                    (1, 2048, 4096, 2, (2, 4, 8), 0, False)])
def test_skinny4_fused_norm_equals_separate_kernels(dev, M, N, K, S_in, geo, epi, fp8):
    """q3asr_skinny4_gemm_norm == q3asr_slab_resnorm_x16 followed by q3asr_skinny4_gemm, bit for bit (residual stream and
    GEMM output), for 512- and 1024-thread workgroups."""
    from qwen3_asr_mlx_amd import ops
    tpw, nks, waves = geo
    S = K // (32 * nks * waves)
    in_slab = rnd((S_in, M, K), dev, 174, 0.5, torch.float32)
    h = rnd((M, K), dev, 175)
    w = rnd((K,), dev, 176) + 1
    W = rnd((N, K), dev, 177, 1.5 / math.sqrt(K))
    if fp8:
        q8, scale, _ = ops.quantize_fp8_rows(W)
        wt = ops.to_w16f8(q8)
    else:
        wt, scale = ops.to_w16(W), None
    h1 = torch.empty_like(h)
    x16 = torch.zeros((K // 32, 1, 64, 8), dtype=torch.bfloat16, device=dev)
    ops.call("q3asr_slab_resnorm_x16", ops.ptr(in_slab), S_in, M * K, ops.ptr(h), ops.ptr(h1), ops.ptr(w), ops.ptr(x16), M, K, 1e-6, ops.sp())
    h2 = torch.zeros_like(h)
    if epi == 0:
        ref = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny4(x16, M, wt, N, K, 0, tpw, nks, waves, slab=ref, scale=scale)
        got = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny4_norm(in_slab, S_in, M * K, h, h2, w, 1e-6, M, wt, N, K, 0, tpw, nks, waves, slab=got, scale=scale)
    else:
        ref = torch.zeros((N // 2 // 32, 1, 64, 8), dtype=torch.bfloat16, device=dev)
        ops.skinny4(x16, M, wt, N, K, 2, tpw, nks, waves, out=ref, scale=scale)
        got = torch.zeros_like(ref)
        ops.skinny4_norm(in_slab, S_in, M * K, h, h2, w, 1e-6, M, wt, N, K, 2, tpw, nks, waves, out=got, scale=scale)
        ref, got = ops.from_x16(ref, M), ops.from_x16(got, M)
    assert torch.equal(h2, h1)
    assert torch.equal(got, ref)
    with pytest.raises(ValueError):                            # h_in == h_out is refused: neighbours still read h_in
        ops.skinny4_norm(in_slab, S_in, M * K, h, h, w, 1e-6, M, wt, N, K, epi, tpw, nks, waves,
                         slab=torch.empty((S, M, N), dtype=torch.float32, device=dev) if epi == 0 else None,
                         out=None if epi == 0 else torch.zeros((N // 2 // 32, 1, 64, 8), dtype=torch.bfloat16, device=dev), scale=scale)


@pytest.mark.parametrize("M,N,S", [(5, 2048, 3), (32, 2048, 8), (33, 512, 1), (2, 4096, 2)])
def test_slab_resnorm_and_embed_norm_x16(dev, M, N, S):
    import oracle.decoder as od
    from qwen3_asr_mlx_amd import ops
    slab = rnd((S, M, N), dev, 120, 0.5, torch.float32)
    h = rnd((M, N), dev, 121)
    w = rnd((N,), dev, 122) + 1
    MT = ops.mt_of(M)
    ho = torch.empty_like(h)
    x16 = torch.zeros((N // 32, MT, 64, 8), dtype=torch.bfloat16, device=dev)
    ops.call("q3asr_slab_resnorm_x16", ops.ptr(slab), S, M * N, ops.ptr(h), ops.ptr(ho), ops.ptr(w), ops.ptr(x16), M, N, 1e-6, ops.sp())
    tot = slab[0].clone()
    for s in range(1, S):
        tot += slab[s]
    href = bfr(bfr(tot.cpu()) + h.float().cpu())
    torch.testing.assert_close(ho.float().cpu(), href, atol=0, rtol=8e-3)        # summation order of S slabs: <= 1 ulp
    xr = od.rms_norm(ho.float().cpu(), w.float().cpu(), 1e-6)
    torch.testing.assert_close(ops.from_x16(x16, M).float().cpu(), xr, atol=0, rtol=8e-3)
    E = rnd((50, N), dev, 124)
    tok = torch.randint(0, 50, (M,), dtype=torch.int32).to(dev)
    ops.call("q3asr_embed_norm_x16", ops.ptr(tok), ops.ptr(E), 50, ops.ptr(ho), ops.ptr(w), ops.ptr(x16), M, N, 1e-6, ops.sp())
    assert torch.equal(ho, E[tok.long()])
    torch.testing.assert_close(ops.from_x16(x16, M).float().cpu(), od.rms_norm(E[tok.long()].float().cpu(), w.float().cpu(), 1e-6),
                               atol=0, rtol=8e-3)
    # an id outside [0, V) (e.g. the sampler's 0x7fffffff sentinel) gathers a zero row instead of faulting
    bad = tok.clone()
    bad[0] = 0x7fffffff
    if M > 1:
        bad[M - 1] = -5
    ops.call("q3asr_embed_norm_x16", ops.ptr(bad), ops.ptr(E), 50, ops.ptr(ho), ops.ptr(w), ops.ptr(x16), M, N, 1e-6, ops.sp())
    assert not ho[0].any() and not ops.from_x16(x16, M)[0].any() and (M == 1 or not ho[M - 1].any())
    if M > 2:
        assert torch.equal(ho[1:M - 1], E[tok[1:M - 1].long()])


@pytest.mark.parametrize("B,poss,S,Lmax", [(1, [0], 1, 576), (3, [5, 127, 200], 4, 576), (2, [408, 535], 2, 576),
                                              (5, [255, 1023, 1024, 2047, 2900], 3, 3072), (40, None, 2, 1920)])
def test_attn_decode_fused_equals_separate_kernels(dev, B, poss, S, Lmax):
    """Fused prologue (slab input, X16 output, 1024-token chunks merged by the last workgroup to arrive) ==
    q3asr_qk_norm_rope_kv (the prefill kernel, slab form) followed by a float32 softmax attention over the cache in torch;
    the workspace is left ready for the next launch, and the result does not depend on how many chunks a workgroup sweeps."""
    from qwen3_asr_mlx_amd import _lib, ops
    nh, nkv, hd = 4, 2, 128
    qkvd = (nh + 2 * nkv) * hd
    if poss is None:
        poss = [(37 * j * j + 11) % Lmax for j in range(B)]
    parts = rnd((S, B, qkvd), dev, 140, 1.0, torch.float32)
    qn, kn = rnd((hd,), dev, 141) + 1, rnd((hd,), dev, 142) + 1
    inv = (1e6 ** (-torch.arange(64, dtype=torch.float32) / 64)).to(dev)
    kc0, vc0 = rnd((B, nkv, Lmax, hd), dev, 143), rnd((B, nkv, Lmax, hd), dev, 144)
    pos = torch.tensor(poss, dtype=torch.int32, device=dev)
    rb = torch.arange(B, dtype=torch.int32, device=dev)
    kc1, vc1 = kc0.clone(), vc0.clone()
    q = torch.zeros((B, nh * hd), dtype=torch.bfloat16, device=dev)
    ops.call("q3asr_qk_norm_rope_kv", ops.ptr(parts), 1, qkvd, S, B * qkvd, ops.ptr(rb), ops.ptr(pos), ops.ptr(qn), ops.ptr(kn),
             ops.ptr(inv), 1e-6, ops.ptr(q), ops.ptr(kc1), ops.ptr(vc1), B, nh, nkv, hd, Lmax, ops.sp())
    ref = torch.zeros((B, nh * hd), dtype=torch.float32)
    for b in range(B):
        L = poss[b] + 1
        for h in range(nh):
            ref[b, h * hd: (h + 1) * hd] = _attn_ref(q.float().cpu()[b: b + 1, h * hd: (h + 1) * hd], kc1.float().cpu()[b, h // 2, :L],
                                                      vc1.float().cpu()[b, h // 2, :L], hd ** -0.5, False)[0]
    ref = bfr(ref).to(dev)
    MT = ops.mt_of(B)
    nbytes = _lib.load().q3asr_attn_decode_fused_workspace(B, nkv, Lmax)
    assert nbytes == 4 * (B * nkv * ((Lmax + 1023) // 1024) * 260 + B * nkv)   # ADF_CHUNK = 1024
    wsp = torch.zeros((nbytes // 4,), dtype=torch.float32, device=dev)
    outs = []
    for rep in range(3):                                       # same workspace, back to back
        kc2, vc2 = kc0.clone(), vc0.clone()
        o16 = torch.zeros((nh * hd // 32, MT, 64, 8), dtype=torch.bfloat16, device=dev)
        ops.call("q3asr_attn_decode_fused", ops.ptr(parts), qkvd, S, B * qkvd, ops.ptr(qn), ops.ptr(kn), ops.ptr(inv), 1e-6,
                 ops.ptr(kc2), ops.ptr(vc2), ops.ptr(pos), ops.ptr(o16), B, nh, nkv, hd, Lmax, hd ** -0.5, ops.ptr(wsp), ops.sp())
        assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
        outs.append(ops.from_x16(o16, B))
        assert int(wsp.view(torch.int32)[-B * nkv:].abs().sum()) == 0            # arrival counters back at zero
    # same maths: bf16 probabilities in 32-key MFMA tiles, per-wave and per-chunk partials merged in a fixed order
    torch.testing.assert_close(outs[0].float(), ref.float(), atol=2e-2, rtol=1.6e-2)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])      # arrival order does not matter
    # any number of chunks per workgroup gives the same bits (per-chunk partials merged in chunk order)
    try:
        for n_chunks in (1, 2):
            _lib.call("q3asr_attn_decode_set_chunks_per_wg", n_chunks)
            kc3, vc3 = kc0.clone(), vc0.clone()
            o16 = torch.zeros((nh * hd // 32, MT, 64, 8), dtype=torch.bfloat16, device=dev)
            ops.call("q3asr_attn_decode_fused", ops.ptr(parts), qkvd, S, B * qkvd, ops.ptr(qn), ops.ptr(kn), ops.ptr(inv), 1e-6,
                     ops.ptr(kc3), ops.ptr(vc3), ops.ptr(pos), ops.ptr(o16), B, nh, nkv, hd, Lmax, hd ** -0.5, ops.ptr(wsp), ops.sp())
            assert torch.equal(kc3, kc1) and torch.equal(vc3, vc1) and torch.equal(ops.from_x16(o16, B), outs[0])
    finally:
        _lib.call("q3asr_attn_decode_set_chunks_per_wg", 0)


def test_attn_decode_fused_is_independent_of_batch_and_capacity(dev):
    """A sequence's attention output is bit-identical whatever batch it sits in and whatever the cache capacity
    (chunks are absolute 1024-token ranges merged in chunk order)."""
    from qwen3_asr_mlx_amd import _lib, ops
    nh, nkv, hd, S = 4, 2, 128, 2
    qkvd = (nh + 2 * nkv) * hd
    qn, kn = rnd((hd,), dev, 141) + 1, rnd((hd,), dev, 142) + 1
    inv = (1e6 ** (-torch.arange(64, dtype=torch.float32) / 64)).to(dev)

    def run(parts, kc, vc, poss, Lmax):
        B = len(poss)
        pos = torch.tensor(poss, dtype=torch.int32, device=dev)
        wsp = torch.zeros((_lib.load().q3asr_attn_decode_fused_workspace(B, nkv, Lmax) // 4,), dtype=torch.float32, device=dev)
        o16 = torch.zeros((nh * hd // 32, ops.mt_of(B), 64, 8), dtype=torch.bfloat16, device=dev)
        ops.call("q3asr_attn_decode_fused", ops.ptr(parts), qkvd, S, B * qkvd, ops.ptr(qn), ops.ptr(kn), ops.ptr(inv), 1e-6,
                 ops.ptr(kc), ops.ptr(vc), ops.ptr(pos), ops.ptr(o16), B, nh, nkv, hd, Lmax, hd ** -0.5, ops.ptr(wsp), ops.sp())
        return ops.from_x16(o16, B)

    B, Lmax = 20, 2560
    poss = [700, 3, 1024, 2400, 1023] * 4
    parts = rnd((S, B, qkvd), dev, 160, 1.0, torch.float32)
    kc, vc = rnd((B, nkv, Lmax, hd), dev, 161), rnd((B, nkv, Lmax, hd), dev, 162)
    full = run(parts, kc.clone(), vc.clone(), poss, Lmax)
    for j in (0, 3, 4):
        big = torch.zeros((1, nkv, 4096, hd), dtype=torch.bfloat16, device=dev)
        bigv = torch.zeros_like(big)
        big[:, :, :Lmax], bigv[:, :, :Lmax] = kc[j:j + 1], vc[j:j + 1]
        one = run(parts[:, j:j + 1].contiguous(), big, bigv, [poss[j]], 4096)
        assert torch.equal(one[0], full[j])


@pytest.mark.parametrize("B,nkv,Lmax,chunks", [(40, 8, 640, 0), (64, 8, 576, 0), (64, 2, 3072, 1), (33, 4, 2560, 1)])
def test_attn_decode_fused_four_wave_workgroups_give_the_same_bits(dev, B, nkv, Lmax, chunks):
    """More (sequence, kv head, chunk group) workgroups than the chip has CUs (33-64 rows at 8 kv heads): the launch uses four-wave
    workgroups, two per CU, each wave computing the partials of two of the eight logical waves (Q3ASR_OPT_ATTN_HALF_WG).  Outputs, appended
    rows and the workspace state are bit-identical to the eight-wave form -- one chunk per sequence and several (cross-workgroup merge)."""
    from qwen3_asr_mlx_amd import _lib, ops
    lib = _lib.load()
    nh, hd, S = 2 * nkv, 128, 2
    qkvd = (nh + 2 * nkv) * hd
    poss = [(p * 197 + 11) % Lmax for p in range(B)]
    poss[0], poss[1], poss[2] = Lmax - 1, 0, min(Lmax - 1, 470)
    parts = rnd((S, B, qkvd), dev, 190, 1.0, torch.float32)
    qn, kn = rnd((hd,), dev, 191) + 1, rnd((hd,), dev, 192) + 1
    inv = (1e6 ** (-torch.arange(64, dtype=torch.float32) / 64)).to(dev)
    kc0, vc0 = rnd((B, nkv, Lmax, hd), dev, 193), rnd((B, nkv, Lmax, hd), dev, 194)
    pos = torch.tensor(poss, dtype=torch.int32, device=dev)
    res = {}
    try:
        if chunks:
            _lib.call("q3asr_attn_decode_set_chunks_per_wg", chunks)
        for half in (1, 0):
            lib.q3asr_set_option(5, half)
            kc, vc = kc0.clone(), vc0.clone()
            wsp = torch.zeros((lib.q3asr_attn_decode_fused_workspace(B, nkv, Lmax) // 4,), dtype=torch.float32, device=dev)
            outs = []
            for rep in range(2):                                  # the same workspace twice: the arrival counters are left at zero
                o16 = torch.zeros((nh * hd // 32, ops.mt_of(B), 64, 8), dtype=torch.bfloat16, device=dev)
                ops.call("q3asr_attn_decode_fused", ops.ptr(parts), qkvd, S, B * qkvd, ops.ptr(qn), ops.ptr(kn), ops.ptr(inv), 1e-6,
                         ops.ptr(kc), ops.ptr(vc), ops.ptr(pos), ops.ptr(o16), B, nh, nkv, hd, Lmax, hd ** -0.5, ops.ptr(wsp), ops.sp())
                outs.append(ops.from_x16(o16, B))
            assert torch.equal(outs[0], outs[1])
            res[half] = (outs[0], kc, vc)
    finally:
        lib.q3asr_set_option(5, 1)
        _lib.call("q3asr_attn_decode_set_chunks_per_wg", 0)
    assert nkv * B * (1 if not chunks else (Lmax + 1023) // 1024) > 256        # the case really takes the four-wave form
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    assert bool(torch.isfinite(res[1][0].float()).all())


@pytest.mark.parametrize("Lmax", [64, 576, 2560])
def test_attn_decode_fused_never_uses_cache_rows_at_or_beyond_the_position(dev, Lmax):
    """Rows of the cache at or beyond a sequence's position hold stale or never-written bits (a previous request's rows, a
    failed row's NaNs).  The kernel must not USE them, whatever it fetches: with NaN, +-Inf and huge finite values in every
    such row (keys and values), the outputs and the appended cache row are bit-identical to a run on a cache whose tail is
    zero -- position 0 (where the kernel's row clamp lands on a row that was never written: this test found that case unguarded),
    positions inside the first tile, at tile and chunk boundaries, and at the last row of the cache.  (Round 5 tried
    requesting the first key tile before the position is known -- rows clamped to the capacity instead of to the last written
    row; this test is what that form had to pass, and did.  It gained nothing: at batch 1 a workgroup's 43 loads per wave are
    issue-bound on its CU's address path, not waiting for the position -- profiles/r05_decode_attention_stamps.txt.)"""
    from qwen3_asr_mlx_amd import _lib, ops
    nh, nkv, hd, S = 4, 2, 128, 2
    qkvd = (nh + 2 * nkv) * hd
    poss = [p for p in (0, 1, 5, 31, 32, 33, 63, 200, 255, 256, 257, 500, 1023, 1024, 1025, 2047, 2559) if p < Lmax] + [Lmax - 1]
    B = len(poss)
    parts = rnd((S, B, qkvd), dev, 170, 1.0, torch.float32)
    qn, kn = rnd((hd,), dev, 171) + 1, rnd((hd,), dev, 172) + 1
    inv = (1e6 ** (-torch.arange(64, dtype=torch.float32) / 64)).to(dev)
    kc0, vc0 = rnd((B, nkv, Lmax, hd), dev, 173), rnd((B, nkv, Lmax, hd), dev, 174)
    pos = torch.tensor(poss, dtype=torch.int32, device=dev)
    poison = torch.tensor([float("nan"), float("inf"), float("-inf"), 3.0e38, -3.0e38], dtype=torch.bfloat16, device=dev)

    def run(fill):
        kc, vc = kc0.clone(), vc0.clone()
        for b, p_ in enumerate(poss):
            n = Lmax - p_
            if fill == "zero":
                kc[b, :, p_:], vc[b, :, p_:] = 0, 0
            else:
                pat = poison[(torch.arange(n * hd, device=dev) + b) % 5].reshape(n, hd)
                kc[b, :, p_:], vc[b, :, p_:] = pat, pat
        wsp = torch.zeros((_lib.load().q3asr_attn_decode_fused_workspace(B, nkv, Lmax) // 4,), dtype=torch.float32, device=dev)
        o16 = torch.zeros((nh * hd // 32, ops.mt_of(B), 64, 8), dtype=torch.bfloat16, device=dev)
        ops.call("q3asr_attn_decode_fused", ops.ptr(parts), qkvd, S, B * qkvd, ops.ptr(qn), ops.ptr(kn), ops.ptr(inv), 1e-6,
                 ops.ptr(kc), ops.ptr(vc), ops.ptr(pos), ops.ptr(o16), B, nh, nkv, hd, Lmax, hd ** -0.5, ops.ptr(wsp), ops.sp())
        rows = torch.stack([torch.stack([kc[b, :, p_], vc[b, :, p_]]) for b, p_ in enumerate(poss)])      # the appended rows
        return ops.from_x16(o16, B), rows

    clean, rows_clean = run("zero")
    dirty, rows_dirty = run("poison")
    assert bool(torch.isfinite(clean.float()).all())
    assert torch.equal(clean, dirty) and torch.equal(rows_clean, rows_dirty)


# ----------------------------------------------------------------------------- fp8 weights (BASELINE.json configs[4])
def test_fp8_row_quantisation_is_exact_in_bf16(dev):
    from qwen3_asr_mlx_amd import ops
    W = rnd((64, 256), dev, 150, 0.05)
    q8, scale, w_eff = ops.quantize_fp8_rows(W)
    assert q8.dtype == torch.uint8 and scale.shape == (64,)
    assert torch.equal(torch.exp2(torch.round(torch.log2(scale))), scale)                  # powers of two
    back = q8.view(torch.float8_e4m3fn).float() * scale[:, None]
    assert torch.equal(back, w_eff.float())                                                 # scale * fp8 is exact in bf16
    assert (w_eff.float() - W.float()).abs().max() <= 0.07 * W.float().abs().max()          # e4m3: 3 mantissa bits
    t = ops.to_w16f8(q8)
    rt, kq, g, li, half, j = 2, 3, 1, 7, 1, 5
    assert t[rt, kq, g, li, half, j] == q8[16 * rt + li, 64 * kq + 32 * half + 8 * g + j]


@pytest.mark.parametrize("M,N,K,S,waves,rt,ksplit,epi", [(32, 4096, 2048, 8, 2, 1, False, 0), (7, 2048, 6144, 6, 4, 2, True, 0),
                                                         (32, 12288, 2048, 1, 4, 2, True, 2), (33, 1024, 512, 1, 2, 2, False, 1),
                                                         (1, 2048, 2048, 4, 4, 1, False, 0)])
def test_skinny3_fp8_weights(dev, M, N, K, S, waves, rt, ksplit, epi):
    from qwen3_asr_mlx_amd import ops
    X = rnd((M, K), dev, 151)
    W = rnd((N, K), dev, 152, 1.5 / math.sqrt(K))
    q8, scale, w_eff = ops.quantize_fp8_rows(W)
    x16, w8 = ops.to_x16(X), ops.to_w16f8(q8)
    ref = X.float().cpu() @ w_eff.float().cpu().T
    if epi == 0:
        slab = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny3(x16, M, w8, N, K, 0, S, waves, slab=slab, rt=rt, ksplit=ksplit, scale=scale)
        torch.testing.assert_close(slab.sum(0).cpu(), ref, atol=2e-3, rtol=1e-3)
        # identical effective weights: the bf16 path on W_eff gives the same numbers up to summation order
        slab2 = torch.empty_like(slab)
        ops.skinny3(x16, M, ops.to_w16(w_eff), N, K, 0, S, waves, slab=slab2, rt=rt, ksplit=ksplit)
        torch.testing.assert_close(slab.sum(0), slab2.sum(0), atol=1e-4, rtol=1e-4)
    elif epi == 1:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ops.skinny3(x16, M, w8, N, K, 1, 1, waves, out=out, rt=rt, scale=scale)
        torch.testing.assert_close(out.float().cpu(), bfr(ref), atol=2e-2, rtol=8e-3)
    else:
        I = N // 2
        o16 = torch.zeros((I // 32, ops.mt_of(M), 64, 8), dtype=torch.bfloat16, device=dev)
        ops.skinny3(x16, M, w8, N, K, 2, 1, waves, out=o16, rt=rt, ksplit=ksplit, scale=scale)
        r = ref.reshape(M, I // 16, 2, 16)
        g, u = bfr(r[:, :, 0].reshape(M, I)), bfr(r[:, :, 1].reshape(M, I))
        torch.testing.assert_close(ops.from_x16(o16, M).float().cpu(), swiglu_ref(g, u), atol=2e-2, rtol=1.6e-2)


SK4_CASES = [  # N, K, epi, (tiles per workgroup, k-steps per wave, waves)
    (4096, 2048, 0, (2, 2, 8)), (2048, 2048, 0, (2, 4, 8)), (2048, 6144, 0, (2, 3, 16)), (2048, 6144, 0, (2, 6, 8)),
    (1024, 512, 0, (2, 2, 8)), (1024, 512, 1, (2, 2, 8)), (9504, 2048, 1, (2, 8, 8)), (9472, 2048, 1, (4, 8, 8)),
    (2048, 512, 2, (2, 2, 8)), (12288, 2048, 2, (4, 4, 16)), (12288, 2048, 2, (2, 4, 16)),
]


def _sk4_run(ops, dev, x16, M, wt, scale, N, K, epi, geo):
    tpw, nks, waves = geo
    S = K // (32 * nks * waves)
    if epi == 0:
        slab = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.skinny4(x16, M, wt, N, K, 0, tpw, nks, waves, slab=slab, scale=scale)
        return slab
    if epi == 1:
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.skinny4(x16, M, wt, N, K, 1, tpw, nks, waves, out=out, scale=scale)
        return out
    o16 = torch.zeros((N // 64, ops.mt_of(M), 64, 8), dtype=torch.bfloat16, device=dev)
    ops.skinny4(x16, M, wt, N, K, 2, tpw, nks, waves, out=o16, scale=scale)
    return o16


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("N,K,epi,geo", SK4_CASES)
def test_skinny4_matches_reference_and_is_batch_invariant(dev, N, K, epi, geo, fp8):
    """q3asr_skinny4_gemm against an fp32 product for every instantiated geometry and epilogue, bf16 and fp8 weights, at
    M = 1 .. 64 (all four m-tile counts -- 1, 2, 3 for 33-48 rows, 4 -- ragged last tile); and the rows of the largest launch
    equal, bit for bit, the same rows computed alone or in any smaller launch (the k partition depends on the geometry only)."""
    from qwen3_asr_mlx_amd import ops
    tpw, nks, waves = geo
    if fp8 and nks % 2:
        pytest.skip("fp8 fragments come in pairs of k steps")
    X = rnd((64, K), dev, 160)
    W = rnd((N, K), dev, 161, (2.0 if epi == 2 else 1.0) / math.sqrt(K))
    if fp8:
        q8, scale, w_eff = ops.quantize_fp8_rows(W)
        wt = ops.to_w16f8(q8)
    else:
        scale, w_eff, wt = None, W, ops.to_w16(W)
    ref_all = X.float().cpu() @ w_eff.float().cpu().T
    results = {}
    for M in (64, 48, 40, 33, 32, 20, 16, 7, 1):
        if M > 32 and waves * tpw * ops.mt_of(M) > 160:
            continue                                      # 3 / 4 m-tiles of partial sums do not fit LDS: the decoder halves the tiles
        got = _sk4_run(ops, dev, ops.to_x16(X[:M].contiguous()), M, wt, scale, N, K, epi, geo)
        ref = ref_all[:M]
        if epi == 0:
            torch.testing.assert_close(got.sum(0).cpu(), ref, atol=2e-3, rtol=1e-3)
            results[M] = got
        elif epi == 1:
            torch.testing.assert_close(got.float().cpu(), bfr(ref), atol=2e-2, rtol=8e-3)
            results[M] = got
        else:
            I = N // 2
            r = ref.reshape(M, I // 16, 2, 16)
            g, u = bfr(r[:, :, 0].reshape(M, I)), bfr(r[:, :, 1].reshape(M, I))
            act = ops.from_x16(got, M)
            torch.testing.assert_close(act.float().cpu(), swiglu_ref(g, u), atol=2e-2, rtol=1.6e-2)
            results[M] = act
    Ms = sorted(results)
    big = results[Ms[-1]]
    for M in Ms[:-1]:
        if epi == 0:
            assert torch.equal(results[M], big[:, :M]), f"rows differ between M={M} and M={Ms[-1]}"
        else:
            assert torch.equal(results[M], big[:M]), f"rows differ between M={M} and M={Ms[-1]}"


@pytest.mark.parametrize("N,M", [(151936, 32), (151936, 17), (9504, 32), (4128, 20), (224, 32), (32, 17)])
def test_tied_head_launch_forms_give_identical_logits(dev, N, M):
    """The tied head at 17-32 rows (bf16, K = 2048, geometry 2/8/8) runs as one workgroup per CU walking its tile groups
    (head_stream_kernel); q3asr_set_option switches to one workgroup per group and to grid-order tile groups.  All forms:
    the same logits bit for bit, for group counts that are / are not multiples of 8 and fewer groups than XCDs."""
    from qwen3_asr_mlx_amd import _lib, ops
    K = 2048
    X, W = rnd((M, K), dev, 172), rnd((N, K), dev, 173, 1.0 / math.sqrt(K))
    x16, wt = ops.to_x16(X), ops.to_w16(W)
    lib = _lib.load()
    outs = {}
    try:
        for stream, xmap in ((1, 1), (0, 1), (0, 0)):
            lib.q3asr_set_option(2, stream)
            lib.q3asr_set_option(1, xmap)
            out = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev)      # a guard row behind the last one
            ops.skinny4(x16, M, wt, N, K, 1, 2, 8, 8, out=out)
            outs[(stream, xmap)] = out
    finally:
        lib.q3asr_set_option(2, 1)
        lib.q3asr_set_option(1, 1)
    ref = bfr(X.float().cpu() @ W.float().cpu().T)
    torch.testing.assert_close(outs[(1, 1)][:M].float().cpu(), ref, atol=2e-2, rtol=8e-3)
    assert torch.equal(outs[(1, 1)], outs[(0, 1)]) and torch.equal(outs[(1, 1)], outs[(0, 0)])
    assert bool((outs[(1, 1)][M] == 7.0).all())


def test_skinny4_tile_count_does_not_change_bits(dev):
    """(4, 4, 16) and (2, 4, 16) -- what the decoder switches between at 32 / 33 rows -- give identical SwiGLU outputs."""
    from qwen3_asr_mlx_amd import ops
    N, K, M = 12288, 2048, 32
    X = rnd((M, K), dev, 162)
    W = rnd((N, K), dev, 163, 2.0 / math.sqrt(K))
    a = _sk4_run(ops, dev, ops.to_x16(X), M, ops.to_w16(W), None, N, K, 2, (4, 4, 16))
    b = _sk4_run(ops, dev, ops.to_x16(X), M, ops.to_w16(W), None, N, K, 2, (2, 4, 16))
    assert torch.equal(a, b)


def test_skinny4_rejects_uninstantiated_geometry(dev):
    from qwen3_asr_mlx_amd import _lib, ops
    X = rnd((4, 512), dev, 164)
    W = rnd((64, 512), dev, 165)
    slab = torch.empty((1, 4, 64), dtype=torch.float32, device=dev)
    with pytest.raises((_lib.Q3AsrError, ValueError), match="not instantiated"):
        ops.skinny4(ops.to_x16(X), 4, ops.to_w16(W), 64, 512, 0, 1, 2, 8, slab=slab)
    with pytest.raises((_lib.Q3AsrError, ValueError), match="multiple"):
        ops.skinny4(ops.to_x16(X), 4, ops.to_w16(W), 64, 512, 0, 2, 3, 16, slab=slab)


@pytest.mark.parametrize("M,S", [(32, 2), (32, 4), (32, 8), (7, 3), (64, 8), (1, 4)])
def test_slab_resnorm_slab_count_specialisations_sum_in_slab_order(dev, M, S):
    """Hidden size 2048 with 2 / 4 / 8 slabs runs a fully unrolled specialisation (all loads first); like the generic loop
    (S = 3 here) it must add the slabs in index order: the residual stream equals the sequential fp32 sum exactly, and the
    specialised and generic forms agree bit for bit on the same values (a zero slab appended to S = 3 gives S = 4)."""
    from qwen3_asr_mlx_amd import ops
    N = 2048
    slab = rnd((S, M, N), dev, 220 + S, 0.5, torch.float32)
    h = rnd((M, N), dev, 231)
    w = rnd((N,), dev, 232) + 1
    MT = ops.mt_of(M)

    def run(sl, n):
        ho = torch.empty_like(h)
        x16 = torch.zeros((N // 32, MT, 64, 8), dtype=torch.bfloat16, device=dev)
        ops.call("q3asr_slab_resnorm_x16", ops.ptr(sl), n, M * N, ops.ptr(h), ops.ptr(ho), ops.ptr(w), ops.ptr(x16), M, N, 1e-6, ops.sp())
        return ho, x16

    ho, x16 = run(slab, S)
    tot = slab[0].clone()
    for s_ in range(1, S):
        tot = tot + slab[s_]
    href = (tot.to(torch.bfloat16).float() + h.float()).to(torch.bfloat16)
    assert torch.equal(ho, href)
    if S == 3:
        padded = torch.cat([slab, torch.zeros_like(slab[:1])])               # + 0.0 changes nothing (no -0.0 sums here)
        ho4, x4 = run(padded, 4)
        assert torch.equal(ho4, ho) and torch.equal(x4, x16)
