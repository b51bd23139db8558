"""High-level API (mirror of reference model.py): Qwen3ASR.from_pretrained / transcribe /
warm_up / close, plus transcribe_batch (new: the reference is batch-1 and single-GPU)."""
from __future__ import annotations

import gc
import threading
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from .audio import get_frontend, load_audio, pack_audio
from .config import ModelConfig
from .decoder import TextDecoder, load_decoder_weights
from .encoder import AudioEncoder, load_encoder_weights
from .generate import generate_batch, _next_seed
from .tokenizer import EOS_TOKEN_IDS, Tokenizer

# ISO 639-1 -> language name used in the prompt (data table of reference model.py:28-96)
LANGUAGE_MAP: dict[str, str] = dict(
    af="Afrikaans", ar="Arabic", az="Azerbaijani", be="Belarusian", bg="Bulgarian", bn="Bengali", bs="Bosnian",
    ca="Catalan", cs="Czech", cy="Welsh", da="Danish", de="German", el="Greek", en="English", es="Spanish",
    et="Estonian", fa="Persian", fi="Finnish", fr="French", gl="Galician", gu="Gujarati", he="Hebrew", hi="Hindi",
    hr="Croatian", hu="Hungarian", hy="Armenian", id="Indonesian", it="Italian", ja="Japanese", ka="Georgian",
    kk="Kazakh", kn="Kannada", ko="Korean", lt="Lithuanian", lv="Latvian", mk="Macedonian", ml="Malayalam",
    mn="Mongolian", mr="Marathi", ms="Malay", my="Burmese", ne="Nepali", nl="Dutch", no="Norwegian", pa="Punjabi",
    pl="Polish", pt="Portuguese", ro="Romanian", ru="Russian", si="Sinhala", sk="Slovak", sl="Slovenian",
    sq="Albanian", sr="Serbian", sv="Swedish", sw="Swahili", ta="Tamil", te="Telugu", th="Thai", tl="Filipino",
    tr="Turkish", uk="Ukrainian", ur="Urdu", uz="Uzbek", vi="Vietnamese", zh="Chinese",
)
LANGUAGE_MAP["is"] = "Icelandic"     # `is` is a keyword, cannot be a dict() kwarg


@dataclass
class TranscriptionResult:
    text: str
    language: str
    duration: float


def _find_split_points(samples: np.ndarray, chunk_samples: int, search_samples: int, frame_samples: int = 480) -> list[int]:
    """Low-energy split points for long audio (reference model.py:454-513): RMS of 480-sample
    frames (tail dropped); for every boundary k*chunk_samples the first minimum within
    +-search_samples, snapped to the frame start."""
    total = len(samples)
    n_frames = total // frame_samples
    if n_frames == 0:
        return []
    frames = np.asarray(samples[: n_frames * frame_samples], dtype=np.float32).reshape(n_frames, frame_samples)
    radius = search_samples // frame_samples
    boundaries = range(chunk_samples, total, chunk_samples)
    # The reference takes the RMS of every frame and then looks at 2 * radius + 1 of them per boundary.  A frame's energy is a
    # row-wise reduction (independent of the other rows), so the search windows alone give the same float32 values: a third of
    # the samples at 30 s chunks (12 -> 4 ms of host time per 20 min of audio).  Overlapping windows (chunks shorter than the
    # search span) fall back to one pass over everything.
    windowed = len(boundaries) * (2 * radius + 1) < n_frames
    energy = None if windowed else np.sqrt(np.mean(frames ** 2, axis=1)).astype(np.float32)
    points: list[int] = []
    for boundary in boundaries:
        centre = boundary // frame_samples
        lo, hi = max(0, centre - radius), min(n_frames - 1, centre + radius)
        if lo >= hi:
            points.append(boundary)
        else:
            e = np.sqrt(np.mean(frames[lo: hi + 1] ** 2, axis=1)) if windowed else energy[lo: hi + 1]
            points.append((int(np.argmin(e)) + lo) * frame_samples)
    return points


def plan_jobs(samples: Sequence[np.ndarray], chunk_duration: float, max_tokens: Optional[int]):
    """Decode jobs of a request: one per utterance, or one per low-energy-split segment for audio longer than
    `chunk_duration` (reference model.py:313,382-447; a segment's budget is max(256, 50 tokens per second)).
    Returns (jobs, seg_of): jobs = [(utterance index, samples, max_tokens)], seg_of[utterance] = its job indices
    (empty utterances have none)."""
    jobs: list = []
    seg_of: dict[int, list[int]] = {}
    for i, s in enumerate(samples):
        if len(s) == 0:
            continue
        dur = len(s) / 16_000
        if dur > chunk_duration:
            prev = 0
            for sp in _find_split_points(s, int(chunk_duration * 16_000), int(5.0 * 16_000)) + [len(s)]:
                seg, prev = s[prev:sp], sp
                if len(seg):
                    seg_of.setdefault(i, []).append(len(jobs))
                    jobs.append((i, seg, max(256, int(len(seg) / 16_000 * 50))))
        else:
            seg_of[i] = [len(jobs)]
            jobs.append((i, s, max_tokens if max_tokens is not None else max(256, int(dur * 50))))
    return jobs, seg_of


class Qwen3ASR:
    """Qwen3-ASR speech-to-text on one MI355X (reference model.py:121-447)."""

    MAX_BATCH = 64

    def __init__(self, config: ModelConfig, encoder: AudioEncoder, decoder: TextDecoder, tokenizer: Tokenizer) -> None:
        self._config = config
        self._encoder = encoder
        self._decoder = decoder
        self._tokenizer = tokenizer
        self._lock = threading.Lock()
        self.use_graph = True
        self.lanes = 2              # batches in flight when a request needs more than MAX_BATCH utterances/segments (1: sequential)
        self.continuous = True      # such requests refill decode rows as sequences end (generate_continuous) instead of
                                    # decoding fixed batches to their longest member
        self.min_admit = None       # free rows to collect before a refill prefill (None: rows / 8)
        self.continuous_lanes = 1   # lanes in that mode (measured, tools/bench_continuous.py: one lane of 64 refilled rows beats two
                                    # when token counts cannot be predicted from clip length; two win by 6 % when they can)

    @classmethod
    def from_pretrained(cls, model_id_or_path: str | Path, **kwargs) -> "Qwen3ASR":
        """Local model directory (config.json, model.safetensors, tokenizer files) or a Hub repo id
        (downloaded with snapshot_download(**kwargs); needs network).  Engine keywords, consumed here (everything else goes to
        snapshot_download as in the reference): ``device``; ``weight_dtype`` "bf16" | "fp8" | "fp8w" (BASELINE.json configs[4]);
        ``decode_gemm`` "sk4" | "sk3" (decode-step GEMM kernel, A/B); ``use_graph`` (hipGraph decode, default True);
        ``encoder_precision`` "bf16" | "fp32" (the reference-semantics encoder mode)."""
        device = kwargs.pop("device", None)
        weight_dtype = kwargs.pop("weight_dtype", "bf16")          # "fp8": BASELINE.json configs[4]
        knobs = {k: kwargs.pop(k) for k in ("decode_gemm", "use_graph", "encoder_precision") if k in kwargs}
        path = Path(model_id_or_path)
        if not path.is_dir():
            from huggingface_hub import snapshot_download
            path = Path(snapshot_download(repo_id=str(model_id_or_path), **kwargs))
        config = ModelConfig.from_pretrained(path)
        encoder = AudioEncoder(config.audio_encoder)
        load_encoder_weights(encoder, path, device)
        decoder = TextDecoder(config.text_decoder)
        load_decoder_weights(decoder, path, device, weight_dtype)
        return cls(config, encoder, decoder, Tokenizer(path))._configure(**knobs)

    @classmethod
    def from_state_dict(cls, config: ModelConfig, state: dict, tokenizer, device=None, weight_dtype: str = "bf16", **knobs) -> "Qwen3ASR":
        """Build from in-memory checkpoint tensors (keys as in model.safetensors); `knobs` as in from_pretrained."""
        import torch
        dev = device or f"cuda:{torch.cuda.current_device()}"
        encoder = AudioEncoder(config.audio_encoder)
        encoder.load_state({k[len("audio_tower."):]: v for k, v in state.items() if k.startswith("audio_tower.")}, dev)
        decoder = TextDecoder(config.text_decoder)
        decoder.load_state({k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}, dev, weight_dtype)
        return cls(config, encoder, decoder, tokenizer)._configure(**knobs)

    def _configure(self, decode_gemm: Optional[str] = None, use_graph: Optional[bool] = None,
                   encoder_precision: Optional[str] = None) -> "Qwen3ASR":
        """The engine's explicit switches (no environment variables are read anywhere in the package)."""
        if decode_gemm is not None:
            if decode_gemm not in ("sk4", "sk3"):
                raise ValueError(f"decode_gemm must be 'sk4' or 'sk3', got {decode_gemm!r}")
            self._decoder.decode_gemm = decode_gemm
        if use_graph is not None:
            self.use_graph = bool(use_graph)
        if encoder_precision is not None:
            if encoder_precision not in ("bf16", "fp32"):
                raise ValueError(f"encoder_precision must be 'bf16' or 'fp32', got {encoder_precision!r}")
            self._encoder.precision = encoder_precision
        return self

    # ------------------------------------------------------------------ public API
    def transcribe(self, audio, language: Optional[str] = None, temperature: float = 0.0, top_p: float = 1.0,
                   top_k: int = 0, repetition_penalty: float = 1.2, max_tokens: Optional[int] = None,
                   repetition_context_size: int = 100, chunk_duration: float = 1200.0) -> TranscriptionResult:
        with self._lock:
            return self._transcribe_impl(audio, language, temperature, top_p, top_k, repetition_penalty, max_tokens,
                                         repetition_context_size, chunk_duration)

    def transcribe_batch(self, audios: Sequence, language: Optional[str] = None, temperature: float = 0.0,
                         top_p: float = 1.0, top_k: int = 0, repetition_penalty: float = 1.2,
                         max_tokens: Optional[int] = None, repetition_context_size: int = 100,
                         chunk_duration: float = 1200.0) -> list[TranscriptionResult]:
        """Transcribe many utterances together; element i equals ``transcribe(audios[i], ...)``."""
        with self._lock:
            samples = [self._as_samples(a) for a in audios]
            lang = self._resolve_language(language)
            results: list[Optional[TranscriptionResult]] = [None] * len(samples)
            jobs, seg_of = plan_jobs(samples, chunk_duration, max_tokens)
            for i, smp in enumerate(samples):
                if len(smp) == 0:
                    results[i] = TranscriptionResult(text="", language="Unknown", duration=0.0)
            texts = self._run_jobs(jobs, lang, temperature, top_p, top_k, repetition_penalty, repetition_context_size)
            for i, idxs in seg_of.items():
                dur = len(samples[i]) / 16_000
                if dur > chunk_duration:
                    text = " ".join(t for t in (texts[j] for j in idxs) if t)
                else:
                    text = texts[idxs[0]]
                results[i] = TranscriptionResult(text=text, language=lang, duration=dur)
            return results  # type: ignore[return-value]

    def warm_up(self) -> None:
        """Load code objects / build plans with a 0.5 s dummy inference (reference model.py:252-259)."""
        self.transcribe(np.zeros(8000, dtype=np.float32))

    def close(self) -> None:
        """Release weights (reference model.py:261-269)."""
        self._encoder = None  # type: ignore[assignment]
        self._decoder = None  # type: ignore[assignment]
        from .audio import release_staging
        release_staging()                                     # pinned H2D staging buffers
        gc.collect()
        try:
            import torch
            torch.cuda.empty_cache()
        except Exception:
            pass

    def __enter__(self) -> "Qwen3ASR":
        return self

    def __exit__(self, *args) -> None:
        self.close()

    # ------------------------------------------------------------------ internals
    @staticmethod
    def _as_samples(audio) -> np.ndarray:
        if isinstance(audio, (str, Path)):
            return load_audio(audio)
        samples = np.asarray(audio, dtype=np.float32)
        if samples.ndim != 1:
            raise ValueError(f"Audio array must be 1-D (mono), got shape {samples.shape}")
        return samples

    def _transcribe_impl(self, audio, language, temperature, top_p, top_k, repetition_penalty, max_tokens,
                         repetition_context_size, chunk_duration) -> TranscriptionResult:
        samples = self._as_samples(audio)
        if len(samples) == 0:
            return TranscriptionResult(text="", language="Unknown", duration=0.0)
        duration = len(samples) / 16_000
        if max_tokens is None:
            max_tokens = max(256, int(duration * 50))
        lang = self._resolve_language(language)
        if duration > chunk_duration:
            return self._transcribe_chunked(samples, duration, lang, temperature, top_p, top_k, repetition_penalty,
                                            repetition_context_size, chunk_duration)
        text = self._run_jobs([(0, samples, max_tokens)], lang, temperature, top_p, top_k, repetition_penalty,
                              repetition_context_size)[0]
        return TranscriptionResult(text=text, language=lang, duration=duration)

    def _transcribe_chunked(self, samples, duration, lang, temperature, top_p, top_k, repetition_penalty,
                            repetition_context_size, chunk_duration) -> TranscriptionResult:
        """Segments split at low-energy points (model.py:382-447).  They are independent, so they are
        decoded as one batch instead of sequentially."""
        chunk_samples = int(chunk_duration * 16_000)
        points = _find_split_points(samples, chunk_samples, int(5.0 * 16_000))
        jobs, prev = [], 0
        for sp in points + [len(samples)]:
            seg = samples[prev:sp]
            prev = sp
            if len(seg):
                jobs.append((0, seg, max(256, int(len(seg) / 16_000 * 50))))
        texts = self._run_jobs(jobs, lang, temperature, top_p, top_k, repetition_penalty, repetition_context_size)
        return TranscriptionResult(text=" ".join(t for t in texts if t), language=lang, duration=duration)

    def _run_jobs(self, jobs, lang, temperature, top_p, top_k, repetition_penalty, repetition_context_size) -> list[str]:
        """mel -> encoder -> prompt -> generate -> detokenise for a list of (idx, samples, max_tokens)."""
        texts: list[str] = [""] * len(jobs)
        order = sorted(range(len(jobs)), key=lambda j: -len(jobs[j][1]))      # similar lengths share a batch
        if self.continuous and len(jobs) > self.MAX_BATCH:
            return self._run_jobs_continuous(jobs, order, lang, temperature, top_p, top_k, repetition_penalty, repetition_context_size)
        batches = [order[s: s + self.MAX_BATCH] for s in range(0, len(order), self.MAX_BATCH)]
        seeds = [_next_seed() if temperature else 0 for _ in batches]         # fixed before any thread runs: reproducible

        def run(bi: int) -> None:
            sel = batches[bi]
            toks = self.generate_tokens([jobs[j][1] for j in sel], [jobs[j][2] for j in sel], lang, temperature, top_p,
                                        top_k, repetition_penalty, repetition_context_size, seed=seeds[bi])
            for j, t in zip(sel, toks):
                texts[j] = self._decode_output(t)

        self._run_batches(len(batches), run)
        return texts

    def _run_batches(self, n: int, run) -> None:
        """run(bi) for every bi in range(n).  Several batches: `lanes` of them in flight (one decodes, latency-bound, while
        another runs its MFMA-bound encoder and prefill); each worker thread owns a lane of the decoder and a stream, which
        first waits for whatever the caller's stream had queued (device-resident inputs)."""
        n_lanes = min(self.lanes, n)
        dev = self._decoder.device if self._decoder is not None else None
        if n_lanes <= 1 or dev is None or dev.type != "cuda":
            for bi in range(n):
                run(bi)
            return
        import torch
        from .decoder import set_lane
        nxt, errs, lock = iter(range(n)), [], threading.Lock()
        with torch.cuda.device(dev):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())

        def worker(lane: int) -> None:
            try:
                set_lane(lane)
                with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    torch.cuda.current_stream().wait_event(ready)
                    while not errs:
                        with lock:
                            bi = next(nxt, None)
                        if bi is None:
                            break
                        run(bi)
                    torch.cuda.current_stream().synchronize()
            except BaseException as e:                        # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=worker, args=(lane,)) for lane in range(1, n_lanes + 1)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errs:
            raise errs[0]

    def _run_jobs_continuous(self, jobs, order, lang, temperature, top_p, top_k, repetition_penalty, repetition_context_size) -> list[str]:
        """More jobs than decode rows: every lane takes every n-th job (longest first), encodes its share in batches and
        decodes it through MAX_BATCH rows that are refilled as sequences end."""
        import torch
        from .decoder import set_lane
        from .generate import generate_continuous
        from .tokenizer import build_prompt
        texts: list[str] = [""] * len(jobs)
        n_lanes = max(1, min(self.continuous_lanes, len(jobs) // self.MAX_BATCH))
        shares = [order[i::n_lanes] for i in range(n_lanes)]
        seeds = [_next_seed() if temperature else 0 for _ in shares]
        lang_ids = self._tokenizer.encode(f" {lang}")
        dev = self._decoder.device
        self.last_continuous_stats = [dict() for _ in shares]

        def run(li: int) -> None:
            sel = shares[li]
            fe = get_frontend(dev)
            feats, keep = [], []
            for s0 in range(0, len(sel), self.MAX_BATCH):
                part = sel[s0: s0 + self.MAX_BATCH]
                audio, lens_dev, lens = pack_audio([jobs[j][1] for j in part], dev)
                mel, n_frames = fe(audio, lens_dev, lens)
                packed, counts = self._encoder.forward_batch(mel, n_frames)
                keep.append(packed)
                o = 0
                for c in counts:
                    feats.append(packed[o: o + c])
                    o += c
            prompts = [build_prompt(int(f.shape[0]), lang_ids) for f in feats]
            toks = generate_continuous(self._decoder, feats, prompts, [jobs[j][2] for j in sel], self.MAX_BATCH, temperature, top_p,
                                       top_k, repetition_penalty, repetition_context_size, seed=seeds[li],
                                       min_admit=self.min_admit, stats=self.last_continuous_stats[li])
            for j, t in zip(sel, toks):
                texts[j] = self._decode_output(t)

        if n_lanes == 1:
            run(0)
            return texts
        errs: list = []

        def worker(li: int) -> None:
            try:
                set_lane(li + 1)
                with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    run(li)
                    torch.cuda.current_stream().synchronize()
            except BaseException as e:                        # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=worker, args=(li,)) for li in range(n_lanes)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errs:
            raise errs[0]
        return texts

    def generate_tokens(self, batch: Sequence[np.ndarray], max_tokens: Sequence[int], lang: str, temperature=0.0,
                        top_p=1.0, top_k=0, repetition_penalty=1.2, repetition_context_size=100, seed=None) -> list[list[int]]:
        """The device part of the hot path for <= MAX_BATCH utterances (steps 5-8 of model.py:330-352)."""
        if self._encoder is None or self._decoder is None:
            raise TypeError("'NoneType' object is not callable")             # use after close(), as the reference
        dev = self._decoder.device
        audio, lens_dev, lens = pack_audio(batch, dev)
        return self.generate_tokens_device(audio, lens_dev, lens, max_tokens, lang, temperature, top_p, top_k,
                                           repetition_penalty, repetition_context_size, seed)

    def generate_tokens_packed(self, audio, lens: Sequence[int], max_tokens: Sequence[int], lang: str, temperature=0.0,
                               top_p=1.0, top_k=0, repetition_penalty=1.2, repetition_context_size=100, seed=None) -> list[list[int]]:
        """Any number of utterances as ONE padded fp32 tensor [n][Lmax] already on the device (what `dist.run_sharded`
        hands each rank): walked in batches of MAX_BATCH rows (`lanes` of them in flight), no host copy of the audio."""
        import torch
        dev = self._decoder.device if self._decoder is not None else None
        starts = list(range(0, len(lens), self.MAX_BATCH))
        parts: list = [None] * len(starts)

        def run(bi: int) -> None:
            s0 = starts[bi]
            part = [int(n) for n in lens[s0: s0 + self.MAX_BATCH]]
            rows = audio[s0: s0 + len(part)]
            lens_dev = torch.tensor(part, dtype=torch.int32).to(dev, non_blocking=True)
            sd = None if seed is None else int(seed) + s0
            parts[bi] = self.generate_tokens_device(rows, lens_dev, part, list(max_tokens[s0: s0 + len(part)]), lang, temperature, top_p,
                                                    top_k, repetition_penalty, repetition_context_size, sd)

        if temperature and seed is None and len(starts) > 1:
            seed = _next_seed()                               # fixed before any thread runs: the draws do not depend on lane timing
        self._run_batches(len(starts), run)                   # shares beyond MAX_BATCH rows: `lanes` batches in flight
        return [t for p_ in parts for t in p_]

    def generate_tokens_device(self, audio, lens_dev, lens: Sequence[int], max_tokens: Sequence[int], lang: str,
                               temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2,
                               repetition_context_size=100, seed=None) -> list[list[int]]:
        """Same, with the padded fp32 audio [B][Lmax] and int32 lengths already resident in HBM."""
        if self._encoder is None or self._decoder is None:
            raise TypeError("'NoneType' object is not callable")
        import torch
        from .generate import _side_stream
        from .tokenizer import build_prompt
        dev = self._decoder.device
        # The whole request runs on ONE stream, the lane's decode stream (graph capture needs a non-legacy stream anyway): a
        # stream that sits waiting for another one mid-request -- the decode stream for the prefill, the caller's stream for the
        # decode loop -- is a hardware queue blocked on a signal, and that slows every kernel of the queue that is running by
        # ~0.5 us (measured on the decode loop, tools/ab_decode_loop.py).  The only wait is here, for the caller's audio.
        side = _side_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fe = get_frontend(dev)
            mel, n_frames = fe(audio, lens_dev, lens)
            feats, tok_counts = self._encoder.forward_batch(mel, n_frames)
            lang_ids = self._tokenizer.encode(f" {lang}")
            prompts = [build_prompt(n, lang_ids) for n in tok_counts]
            return generate_batch(self._decoder, feats, tok_counts, prompts, list(max_tokens), temperature, top_p, top_k,
                                  repetition_penalty, repetition_context_size,
                                  seed=seed if seed is not None else (_next_seed() if temperature else 0), use_graph=self.use_graph)

    def _resolve_language(self, language: Optional[str]) -> str:
        if language is None or language.lower() in ("auto", ""):
            return "English"
        return LANGUAGE_MAP.get(language.lower(), language)

    def _decode_output(self, tokens: list[int]) -> str:
        tokens = list(tokens)
        while tokens and tokens[-1] in EOS_TOKEN_IDS:
            tokens.pop()
        return self._tokenizer.decode(tokens, skip_special_tokens=True).strip()
