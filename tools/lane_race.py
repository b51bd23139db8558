#!/usr/bin/env python3
"""Model-level check behind DESIGN.md section 4: where do two lanes in flight diverge from one?  Same batches run alone and then concurrently on two lanes; compares encoder
features, first tokens (prefill) and the first differing decode step.  Usage: lane_race.py [n_batches] [B] [small|full]"""
import sys
import threading
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import qwen3_asr_mlx_amd as q  # noqa: E402
from qwen3_asr_mlx_amd import synth  # noqa: E402
from qwen3_asr_mlx_amd.audio import get_frontend, pack_audio  # noqa: E402
from qwen3_asr_mlx_amd.config import ModelConfig  # noqa: E402
from qwen3_asr_mlx_amd.decoder import set_lane  # noqa: E402
from qwen3_asr_mlx_amd.generate import generate_batch  # noqa: E402
from qwen3_asr_mlx_amd.tokenizer import build_prompt  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
full = (sys.argv[3] if len(sys.argv) > 3 else "full") == "full"
dev = torch.device("cuda:0")
cfg = ModelConfig() if full else synth.small_config()
state = synth.random_state(cfg, seed=0, device=dev)
model = q.Qwen3ASR.from_state_dict(cfg, state, synth.SyntheticTokenizer(), dev)
del state
rng = np.random.default_rng(7)
durs = np.sort(rng.uniform(5.0, 30.0, NB * B))[::-1]
xs = [(0.1 * np.random.default_rng(1234 + i).standard_normal(int(d * 16000))).astype(np.float32) for i, d in enumerate(durs)]
budgets = [max(1, int(round(d * 128 / 30))) for d in durs]
lang_ids = model._tokenizer.encode(" English")


def run(bi):
    sel = range(bi * B, (bi + 1) * B)
    audio, lens_dev, lens = pack_audio([xs[i] for i in sel], dev)
    mel, n_frames = get_frontend(dev)(audio, lens_dev, lens)
    feats, counts = model._encoder.forward_batch(mel, n_frames)
    prompts = [build_prompt(n, lang_ids) for n in counts]
    toks = generate_batch(model._decoder, feats, counts, prompts, [budgets[i] for i in sel], repetition_penalty=1.2)
    return mel.clone(), feats.clone(), toks


ref = [run(bi) for bi in range(NB)]
again = [run(bi) for bi in range(NB)]
print("one lane, repeated: identical", all(torch.equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(ref, again)), flush=True)
for rep in range(3):
    got, errs = [None] * NB, []
    nxt, lock = iter(range(NB)), threading.Lock()

    def worker(lane):
        try:
            set_lane(lane)
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                while True:
                    with lock:
                        bi = next(nxt, None)
                    if bi is None:
                        break
                    got[bi] = run(bi)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(lane,)) for lane in (1, 2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    for bi in range(NB):
        r, g = ref[bi], got[bi]
        mel_eq, feats_eq = torch.equal(r[0], g[0]), torch.equal(r[1], g[1])
        rows = (r[1] != g[1]).any(dim=1).sum().item() if not feats_eq else 0
        bad = [(i, next((s for s, (x, y) in enumerate(zip(a, b)) if x != y), -1), len(a)) for i, (a, b) in enumerate(zip(r[2], g[2])) if a != b]
        print(f"rep {rep} batch {bi}: mel equal {mel_eq}, features equal {feats_eq} ({rows} rows differ), "
              f"sequences differing {len(bad)} (row, first differing step, length): {bad[:10]}", flush=True)
