"""Token selection on the device: the reference's sampling known-answers (tests/test_generate.py
:132-194) restated against q3asr_sample_step, plus tie-break, penalty rounding, top-k/top-p support
sets and a distribution check against the oracle's filtered softmax."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_greedy_returns_argmax(dev):
    from qwen3_asr_mlx_amd import sample
    assert sample(torch.tensor([[0.1, 0.5, 0.2, 0.9, 0.3]]), temperature=0.0) == 3


def test_greedy_lowest_index_on_ties_and_big_vocab(dev):
    from qwen3_asr_mlx_amd import sample
    g = torch.Generator().manual_seed(0)
    x = torch.randn(151936, generator=g).to(torch.bfloat16).float()
    x[[777, 150000, 90001]] = 7.0                     # exact three-way tie at the top
    assert sample(x, temperature=0.0) == 777
    x[5] = 7.0
    assert sample(x, temperature=0.0) == 5
    assert sample(x, temperature=0.0) == int(torch.argmax(x))


def test_nan_logits_never_yield_an_out_of_range_token(dev):
    """A row without a maximum (all NaN) must not hand the argmax sentinel to the next step's embedding gather."""
    from qwen3_asr_mlx_amd import sample
    x = torch.full((1, 4096), float("nan"))
    assert 0 <= sample(x, temperature=0.0) < 4096
    for kw in (dict(temperature=0.7, top_p=0.9), dict(temperature=1.0, top_k=5), dict(temperature=1.0)):
        assert 0 <= sample(x, seed=3, **kw) < 4096
    x = torch.randn(1, 4096)
    x[0, 100:200] = float("nan")                       # NaNs beside finite logits: still a valid index
    assert 0 <= sample(x, temperature=0.0) < 4096


def test_nan_logits_raise_from_the_generation_loop(dev):
    """The device-side error flag: a row whose logits have no maximum records token 0, stops with finished = Q3ASR_ROW_FAILED
    (include/q3asr.h), and the host raises when it reads the flag -- a NaN row never decodes on as token 0."""
    import qwen3_asr_mlx_amd as q
    from qwen3_asr_mlx_amd import synth
    from qwen3_asr_mlx_amd._lib import Q3AsrError
    from signals import signal
    cfg = synth.small_config()
    state = synth.random_state(cfg, seed=4, bias_std=0.02, norm_jitter=0.1)
    state["model.norm.weight"] = torch.full_like(state["model.norm.weight"], float("nan"))      # every logit NaN
    model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
    clips = [signal("speechlike", 16000, seed=5), signal("speechlike", 24000, seed=6)]
    with pytest.raises(Q3AsrError, match="non-finite logits"):
        model.generate_tokens(clips, [8, 8], "English")
    with pytest.raises(Q3AsrError, match="non-finite logits"):
        model.transcribe(clips[0], max_tokens=8)
    model.use_graph = False
    with pytest.raises(Q3AsrError, match="non-finite logits"):
        model.generate_tokens(clips, [8, 8], "English")


def test_temperature_sampling_is_stochastic(dev):
    from qwen3_asr_mlx_amd import sample
    logits = torch.ones(1, 512)
    assert len({sample(logits, temperature=2.0) for _ in range(20)}) > 1


def test_top_k_1_always_argmax(dev):
    from qwen3_asr_mlx_amd import sample
    logits = torch.tensor([[0.1, 0.5, 0.2, 0.9, 0.3]])
    assert all(sample(logits, temperature=1.0, top_k=1) == 3 for _ in range(10))


def test_top_k_keeps_exactly_k_candidates(dev):
    from qwen3_asr_mlx_amd import sample
    logits = torch.tensor([0.1, 0.9, 0.5, 0.3, 0.7])
    seen = {sample(logits, temperature=1.0, top_k=2, seed=s) for s in range(200)}
    assert seen == {1, 4}


def test_top_k_keeps_ties_at_threshold(dev):
    from qwen3_asr_mlx_amd import sample
    logits = torch.tensor([0.5, 2.0, 1.0, 1.0, 1.0, -3.0])       # k=2 -> threshold 1.0, ties kept (generate.py:141)
    seen = {sample(logits, temperature=1.0, top_k=2, seed=s) for s in range(400)}
    assert seen == {1, 2, 3, 4}


def test_top_p_restricts_to_nucleus(dev):
    from qwen3_asr_mlx_amd import sample
    logits = torch.tensor([100.0, 0.0, 0.0, 0.0])
    assert {sample(logits, temperature=1.0, top_p=0.99, seed=s) for s in range(50)} == {0}


def test_top_p_support_matches_oracle(dev):
    import oracle
    from qwen3_asr_mlx_amd import sample
    logits = torch.tensor([2.0, 1.5, 1.5, 1.0, 0.5, 0.0, -1.0, -2.0]).to(torch.bfloat16).float()
    for top_p in (0.3, 0.5, 0.7, 0.9):
        keep = torch.isfinite(oracle.top_p_filter(logits, top_p)).nonzero().flatten().tolist()
        seen = {sample(logits, temperature=1.0, top_p=top_p, seed=s) for s in range(600)}
        assert seen == set(keep), (top_p, seen, keep)


def test_repetition_penalty_semantics(dev):
    """reference test_generate.py:176-194: listed ids only, l>0 -> l/p, l<0 -> l*p; p=1 identity;
    p is rounded to bf16 (1.2 -> 1.203125) like the weakly-typed MLX scalar."""
    from qwen3_asr_mlx_amd import sample
    # token 0 leads unpenalised; with penalty 2.0 on [0] it drops to 1.5 < 2.0
    logits = torch.tensor([3.0, 2.0, 1.0, -1.0])
    assert sample(logits, repetition_penalty=1.0, recent_tokens=[0]) == 0
    assert sample(logits, repetition_penalty=2.0, recent_tokens=[0]) == 1
    assert sample(logits, repetition_penalty=2.0, recent_tokens=[0, 0, 0]) == 1          # unique ids: applied once
    assert sample(logits, repetition_penalty=2.0, recent_tokens=None) == 0
    # negative logits are multiplied: -1 * 2 = -2 < -1.5
    neg = torch.tensor([-1.0, -1.5, -4.0, -8.0])
    assert sample(neg, repetition_penalty=2.0, recent_tokens=[0]) == 1
    # bf16 rounding of the scalar: 1.2 -> 1.203125; 2.40625/1.203125 = 2.0 exactly
    tie = torch.tensor([2.40625, 2.0, 0.0, 0.0])
    assert sample(tie, repetition_penalty=1.2, recent_tokens=[0]) == 0                    # 2.0 vs 2.0 -> lowest index
    tie2 = torch.tensor([2.0, 2.40625, 0.0, 0.0])
    assert sample(tie2, repetition_penalty=1.2, recent_tokens=[1]) == 0
    # out-of-range ids are ignored
    assert sample(logits, repetition_penalty=2.0, recent_tokens=[99999, -5]) == 0


def test_sampling_distribution_matches_filtered_softmax(dev):
    import oracle
    from qwen3_asr_mlx_amd import sample
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(64, generator=g) * 2).to(torch.bfloat16).float()
    T, k, p = 0.7, 20, 0.9
    x = (logits / torch.tensor(T).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    probs = torch.softmax(oracle.top_p_filter(oracle.top_k_filter(x, k), p), dim=-1).numpy()
    n = 4000
    counts = np.bincount([sample(logits, temperature=T, top_k=k, top_p=p, seed=s) for s in range(n)], minlength=64)
    assert set(np.nonzero(counts)[0]) <= set(np.nonzero(probs)[0])
    # total-variation distance of an n-sample empirical law from its source is ~ sqrt(K/n)
    assert 0.5 * np.abs(counts / n - probs).sum() < 0.06


def test_bookkeeping_eos_budget_and_finished(dev):
    from qwen3_asr_mlx_amd import ops
    from qwen3_asr_mlx_amd.tokenizer import IM_END_TOKEN_ID, ENDOFTEXT_TOKEN_ID
    V, B = 1024, 3
    logits = torch.zeros((B, V), dtype=torch.bfloat16, device=dev)
    logits[0, 5] = 3.0
    logits[1, 7] = 3.0
    logits[2, 9] = 3.0
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    out = torch.full((B, 4), -1, dtype=torch.int32, device=dev)
    n_gen, fin, cur, pos, mx = i32([0, 0, 0]), i32([0, 0, 1]), i32([0, 0, 42]), i32([10, 20, 30]), i32([4, 1, 4])

    def step(eos0=5, eos1=ENDOFTEXT_TOKEN_ID):
        ops.call("q3asr_sample_step", ops.ptr(logits), V, B, V, ops.ptr(out), 4, ops.ptr(n_gen), ops.ptr(fin), ops.ptr(cur),
                 ops.ptr(pos), ops.ptr(mx), 0.0, 1.0, 0, 1.0, 100, 0, eos0, eos1, 0, ops.sp())
    step()
    assert out[0, 0].item() == 5 and fin[0].item() == 1 and pos[0].item() == 10      # EOS recorded, then stop
    assert out[1, 0].item() == 7 and fin[1].item() == 1                               # budget of 1 reached
    assert out[2, 0].item() == -1 and cur[2].item() == 42 and n_gen[2].item() == 0    # already finished: untouched
    step()
    assert n_gen.tolist() == [1, 1, 0] and out[0, 1].item() == -1


# ---------------------------------------------------------------------------------------------------
# full-vocabulary rows through the C-ABI, many independent draws per launch (row b draws with Philox
# counter (n_gen, b), so B copies of one row are B independent draws)
def _draw_many(row_bf16, n_draws, temperature, top_p=1.0, top_k=0, B=64, row_base=0):
    from qwen3_asr_mlx_amd import ops
    dev = row_bf16.device
    V = row_bf16.numel()
    logits = row_bf16.reshape(1, V).repeat(B, 1).contiguous()
    out = []
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
    for it in range((n_draws + B - 1) // B):
        out_tokens = torch.zeros(B, 1, dtype=torch.int32, device=dev)
        n_gen, fin, cur, pos, mx = i32([0] * B), i32([0] * B), i32([0] * B), i32([0] * B), i32([1] * B)
        ops.call("q3asr_sample_step", ops.ptr(logits), V, B, V, ops.ptr(out_tokens), 1, ops.ptr(n_gen), ops.ptr(fin),
                 ops.ptr(cur), ops.ptr(pos), ops.ptr(mx), float(temperature), float(top_p), int(top_k), 1.0, 1,
                 1000 + it, -1, -1, row_base, ops.sp())
        out += cur.tolist()
    return np.array(out[:n_draws])


def test_draw_is_keyed_by_the_global_row_index(dev):
    """row_base: rows 16..31 of a 32-row launch draw exactly what a 16-row launch with row_base = 16 draws (a request decoded
    as two half-batches samples what the whole batch would)."""
    torch.manual_seed(3)
    row = (torch.randn(4096) * 2).to(torch.bfloat16).to(dev)
    whole = _draw_many(row, 32, 0.9, top_p=0.95, B=32)
    half = _draw_many(row, 16, 0.9, top_p=0.95, B=16, row_base=16)
    assert np.array_equal(whole[16:], half) and len(set(whole.tolist())) > 4
    assert not np.array_equal(whole[:16], half)


def _oracle_probs(row_bf16, temperature, top_p, top_k):
    import oracle
    x = (row_bf16.float().cpu() / torch.tensor(temperature).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    if top_k > 0:
        x = oracle.top_k_filter(x, top_k)
    if top_p < 1.0:
        x = oracle.top_p_filter(x, top_p)
    return torch.softmax(x.double(), dim=-1).numpy()


@pytest.mark.parametrize("temperature,top_p,top_k", [(0.7, 0.9, 0), (1.0, 1.0, 50), (0.7, 0.8, 40), (1.3, 0.5, 0),
                                                     (1.0, 1.0, 0), (0.5, 0.95, 2000)])
def test_full_vocab_draws_follow_the_filtered_softmax(dev, temperature, top_p, top_k):
    """V = 151936 (Qwen3-ASR's vocabulary), gaussian logits (heavy bf16 ties): every draw lies in the oracle's kept
    set, and the draws' mean log-probability matches the kept law's negative entropy."""
    V = 151936
    g = torch.Generator().manual_seed(11)
    row = (torch.randn(V, generator=g) * 2.5).to(torch.bfloat16).to(dev)
    probs = _oracle_probs(row, temperature, top_p, top_k)
    n = 4096
    draws = _draw_many(row, n, temperature, top_p, top_k)
    kept = probs > 0
    # The nucleus boundary may sit inside a tie; how many of the tied logits survive is decided by (target - above) / p_tie,
    # which the oracle evaluates with an fp32 softmax that sums to 1 + 1e-5 and the kernel with exact-key masses: they may
    # differ by one tied token of probability ~1e-5.  Anything else outside the kept set is an error.
    x = (row.float().cpu() / torch.tensor(temperature).to(torch.bfloat16).float()).to(torch.bfloat16).float().numpy()
    outside = draws[~kept[draws]]
    assert len(outside) <= 2 and (x[outside] == x[kept].min()).all(), ("draw outside the kept set", x[outside], x[kept].min())
    draws = draws[kept[draws]]
    logp = np.log(probs[kept])
    mean, var = (probs[kept] * logp).sum(), (probs[kept] * logp ** 2).sum() - (probs[kept] * logp).sum() ** 2
    got = np.log(probs[draws]).mean()
    assert abs(got - mean) < 5 * np.sqrt(var / len(draws)) + 1e-3, (got, mean, np.sqrt(var / n))
    if top_k and top_k <= 50:                                            # small kept set: all of it is seen, TV distance small
        assert set(np.unique(draws)) == set(np.nonzero(kept)[0])
        counts = np.bincount(draws, minlength=V)
        assert 0.5 * np.abs(counts / n - probs).sum() < 0.08


def test_full_vocab_nucleus_cuts_inside_a_tie_in_index_order(dev):
    """10 logits at 5.0, 1000 tied at 4.0 scattered over the vocabulary, the rest far below: the nucleus boundary falls
    inside the tie and keeps its first tokens in index order (stable sort + exclusive cumsum, generate.py:152-163)."""
    import oracle
    V = 151936
    g = torch.Generator().manual_seed(5)
    perm = torch.randperm(V, generator=g)
    row = torch.full((V,), -20.0)
    row[perm[:10]] = 5.0
    row[perm[10:1010]] = 4.0
    row = row.to(torch.bfloat16).to(dev)
    for top_p in (0.3, 0.6):
        keep = torch.isfinite(oracle.top_p_filter(row.float().cpu(), top_p)).nonzero().flatten().numpy()
        assert 10 < len(keep) < 1010
        draws = _draw_many(row, 16384, 1.0, top_p=top_p)
        assert set(np.unique(draws)) == set(keep), (top_p, len(set(draws)), len(keep))


def test_full_vocab_top_k_keeps_ties_at_threshold(dev):
    V = 151936
    row = torch.full((V,), -3.0)
    row[[7, 70000, 151935]] = 2.0
    row[[100, 101, 90000, 151000]] = 1.0                                 # k = 5 lands inside this tie: all four stay
    row = row.to(torch.bfloat16).to(dev)
    draws = _draw_many(row, 2048, 1.0, top_k=5)
    assert set(np.unique(draws)) == {7, 70000, 151935, 100, 101, 90000, 151000}


def test_sampling_is_reproducible_for_a_seed(dev):
    V = 151936
    g = torch.Generator().manual_seed(2)
    row = (torch.randn(V, generator=g) * 3).to(torch.bfloat16).to(dev)
    a = _draw_many(row, 256, 0.8, top_p=0.9, top_k=100)
    b = _draw_many(row, 256, 0.8, top_p=0.9, top_k=100)
    assert (a == b).all() and len(set(a.tolist())) > 1


def _step_with_penalty(dev, logits, temperature, top_p, top_k, seed, history):
    """One q3asr_sample_step over a batch whose rows already generated `history` (the penalty window)."""
    from qwen3_asr_mlx_amd import ops
    B, V = logits.shape
    n = len(history[0])
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
    out_tokens = i32([list(h) + [0] for h in history])
    n_gen, fin, cur, pos, mx = i32([n] * B), i32([0] * B), i32([0] * B), i32([0] * B), i32([n + 1] * B)
    ops.call("q3asr_sample_step", ops.ptr(logits), V, B, V, ops.ptr(out_tokens), n + 1, ops.ptr(n_gen), ops.ptr(fin),
             ops.ptr(cur), ops.ptr(pos), ops.ptr(mx), float(temperature), float(top_p), int(top_k), 1.2, 100, seed, -1, -1, 0, ops.sp())
    return cur.tolist()


@pytest.mark.parametrize("V", [151936, 4096, 1056, 155648, 155680, 200000])
def test_resident_and_streaming_sampler_forms_pick_the_same_tokens(dev, V):
    """Rows of <= 155,648 logits are held in registers / LDS across the sampler's passes, longer rows are re-read by every
    pass; q3asr_set_option(Q3ASR_OPT_SAMPLER_RESIDENT, 0) forces the streaming form.  Same tokens from both, greedy and
    drawn, with a repetition penalty in force, for vocabularies that end inside a pass, at the capacity, and beyond it
    (where both settings stream)."""
    from qwen3_asr_mlx_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(V)
    B = 24
    logits = (torch.randn(B, V, generator=g) * 2.5).to(torch.bfloat16).to(dev)
    history = [[int(t) for t in torch.randint(0, V, (20,), generator=g)] for _ in range(B)]
    for b in range(B):                                   # the penalty must matter: the row's best logits are in the window
        history[b][:3] = [int(i) for i in torch.topk(logits[b].float(), 3).indices]
    got = {}
    try:
        for res in (1, 0):
            lib.q3asr_set_option(4, res)
            got[res] = [_step_with_penalty(dev, logits, *cfg, history) for cfg in
                        ((0.0, 1.0, 0, 1), (0.7, 0.9, 0, 2), (1.0, 1.0, 50, 3), (0.8, 0.95, 200, 4), (1.3, 1.0, 0, 5))]
    finally:
        lib.q3asr_set_option(4, 1)
    assert got[1] == got[0]
    greedy = got[1][0]
    for b in range(B):                                   # and the greedy pick is the penalised argmax (oracle arithmetic)
        import oracle
        ref = oracle.apply_repetition_penalty(logits[b].float().cpu(), history[b], 1.2)
        assert greedy[b] == int(torch.argmax(ref))
    assert len(set(got[1][1])) > 1
