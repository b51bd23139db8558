"""A queue in front of the decode rows ("streaming front door"): requests arrive at any time from
any thread, a worker thread encodes what has arrived, admits it into free rows of a running RowScheduler and resolves each
request's future when its sequence (or, for long audio, all of its segments) has ended.  The reference serialises whole
`transcribe()` calls behind one lock (model.py:239); results here equal `transcribe()` of the same audio and options.

One ContinuousTranscriber = one lane of the decoder (its own decode workspace, KV cache, captured step and streams) and one
set of sampling options (they are baked into the captured decode step)."""
from __future__ import annotations

import queue
import threading
from collections import deque
from concurrent.futures import Future
from typing import Optional

import numpy as np

from .audio import get_frontend, pack_audio
from .generate import RowScheduler
from .model import Qwen3ASR, TranscriptionResult, plan_jobs
from .tokenizer import build_prompt

_STOP = object()


class ContinuousTranscriber:
    def __init__(self, model: Qwen3ASR, rows: int = 32, language: Optional[str] = None, temperature: float = 0.0, top_p: float = 1.0,
                 top_k: int = 0, repetition_penalty: float = 1.2, repetition_context_size: int = 100, seed: int = 0,
                 chunk_duration: float = 30.0, max_new_tokens: int = 1750, lane: int = 7, check_every: int = 16,
                 min_admit: Optional[int] = None, max_wait_rows: int = 64, encode_thread: bool = True, async_admit: bool = False) -> None:
        """`chunk_duration`: audio longer than this is split at low-energy points (as transcribe() does at 1200 s) so that every
        sequence fits the row capacity, which is sized for chunk_duration + 5 s of audio and `max_new_tokens` new tokens.
        `lane`: decoder lane of the worker thread (keep it different from lanes used elsewhere in the process).
        `encode_thread`: mel + encoder of arriving audio run on a second thread and stream (lane + 1), beside the decode rows
        (MFMA-bound work beside latency-bound work) instead of between decode chunks."""
        self.model = model
        self.rows = int(rows)
        self.lang = model._resolve_language(language)
        self.sampling = {"temperature": float(temperature), "top_p": float(top_p), "top_k": int(top_k),
                         "repetition_penalty": float(repetition_penalty),
                         "repetition_context_size": int(repetition_context_size), "seed": int(seed)}
        self.chunk_duration = float(chunk_duration)
        self.max_new = int(max_new_tokens)
        self.lane, self.check_every, self.min_admit = int(lane), int(check_every), min_admit
        self.max_wait_rows = int(max_wait_rows)
        enc = model._encoder
        frames = int((self.chunk_duration + 5.0) * 100) + 1
        n_audio = (frames // enc.chunk_size) * enc.tokens_per_chunk + (enc._conv_output_length(frames % enc.chunk_size) if frames % enc.chunk_size else 0)
        self._lang_ids = model._tokenizer.encode(f" {self.lang}")
        self.max_prompt = len(build_prompt(n_audio, self._lang_ids))
        self._q: "queue.Queue" = queue.Queue()
        self._enc_q: "queue.Queue" = queue.Queue()           # (scheduler items, event on the encoder stream) or _STOP
        self._encode_thread = bool(encode_thread)
        self._async_admit = bool(async_admit)                # admission prefills on a second stream beside the decode steps
        self._saturated = False                              # set by the worker: encoded requests are waiting for a free row
        self.gather_min, self.gather_s = 8, 0.1              # while saturated: collect this many segments / wait this long per encoder batch
        self._closed = False
        self.stats: dict = {}
        self._err: Optional[BaseException] = None
        self._pending: deque = deque()                       # encoded items waiting for a row (owned by the row worker)
        self._th = threading.Thread(target=self._loop, name="q3asr-rows", daemon=True)
        self._th_enc = threading.Thread(target=self._encode_loop, name="q3asr-encode", daemon=True) if self._encode_thread else None
        self._th.start()
        if self._th_enc is not None:
            self._th_enc.start()

    # ------------------------------------------------------------------ client side
    def submit(self, audio, max_tokens: Optional[int] = None) -> Future:
        """Queue one utterance (path or 1-D float array, as transcribe()).  The future resolves to a TranscriptionResult."""
        if self._closed:
            raise RuntimeError("ContinuousTranscriber is closed")
        self._raise_if_dead()
        fut: Future = Future()
        samples = self.model._as_samples(audio)
        if len(samples) == 0:
            fut.set_result(TranscriptionResult(text="", language="Unknown", duration=0.0))
            return fut
        jobs, _ = plan_jobs([samples], self.chunk_duration, max_tokens)
        if any(len(j[1]) < 160 for j in jobs):               # same error transcribe() raises, before anything is queued
            raise ValueError("audio shorter than one hop (160 samples) has no mel frames")
        if any(j[2] > self.max_new for j in jobs):
            raise ValueError(f"max_tokens {max(j[2] for j in jobs)} exceeds this transcriber's max_new_tokens={self.max_new}")
        req = {"future": fut, "texts": [None] * len(jobs), "left": len(jobs), "duration": len(samples) / 16_000,
               "joined": len(samples) / 16_000 > self.chunk_duration}
        self._q.put([(req, k, j[1], int(j[2])) for k, j in enumerate(jobs)])
        if self._err is not None:                            # the service died while this was being queued: nobody reads _q any more
            self._fail_queued(self._err)
        return fut

    def _raise_if_dead(self) -> None:
        """A worker that has failed (or a thread that is gone) can never complete a request: refuse new ones at once."""
        if self._err is not None:
            raise RuntimeError(f"ContinuousTranscriber has failed: {type(self._err).__name__}: {self._err}") from self._err
        if not self._th.is_alive() or (self._th_enc is not None and not self._th_enc.is_alive()):
            raise RuntimeError("ContinuousTranscriber worker is not running")

    def _fail_queued(self, e: BaseException) -> None:
        """Fail every request still waiting in the raw queue or the encoded queue (called once the service is dead)."""
        while True:
            try:
                got = self._q.get_nowait()
            except queue.Empty:
                break
            if got is not _STOP:
                for seg in got:
                    if not seg[0]["future"].done():
                        seg[0]["future"].set_exception(e)
        while True:
            try:
                got = self._enc_q.get_nowait()
            except queue.Empty:
                break
            if got is not _STOP:
                for it in got[0]:
                    if not it[0][0]["future"].done():
                        it[0][0]["future"].set_exception(e)

    def transcribe(self, audio, max_tokens: Optional[int] = None) -> TranscriptionResult:
        return self.submit(audio, max_tokens).result()

    def close(self) -> None:
        """Finish what has been submitted, then stop the worker."""
        if not self._closed:
            self._closed = True
            self._q.put(_STOP)
            if self._th_enc is not None:
                self._th_enc.join()
            self._th.join()
        if self._err is not None:
            raise self._err

    def __enter__(self) -> "ContinuousTranscriber":
        return self

    def __exit__(self, *args) -> None:
        self.close()

    # ------------------------------------------------------------------ worker
    def _drain(self, block: bool):
        """Segments that have arrived: [(req, k, samples, budget)], and whether the stop marker was seen.  While the decode rows
        are saturated (requests are waiting for a row anyway) it keeps collecting for up to `gather_s`, so that the encoder
        runs on a batch: one clip alone costs 4.3 ms of launch-bound GPU work, eight together 0.8 ms each."""
        import time
        items, stop = [], False
        deadline = None
        try:
            got = self._q.get(timeout=0.05) if block else self._q.get_nowait()
            while True:
                if got is _STOP:
                    stop = True
                else:
                    items += got
                if len(items) >= self.max_wait_rows or stop:
                    break
                try:
                    got = self._q.get_nowait()
                except queue.Empty:
                    if not (self._saturated and self._encode_thread and len(items) < self.gather_min):
                        break
                    deadline = deadline or time.perf_counter() + self.gather_s
                    left = deadline - time.perf_counter()
                    if left <= 0:
                        break
                    got = self._q.get(timeout=left)
        except queue.Empty:
            pass
        return items, stop

    def _encode(self, segs: list) -> list:
        """mel + encoder for the arrived segments (batches of <= MAX_BATCH, longest first); returns scheduler items."""
        model, dev = self.model, self.model._decoder.device
        out = []
        order = sorted(range(len(segs)), key=lambda i: -len(segs[i][2]))
        for s0 in range(0, len(order), model.MAX_BATCH):
            part = [segs[i] for i in order[s0: s0 + model.MAX_BATCH]]
            audio, lens_dev, lens = pack_audio([p[2] for p in part], dev)
            mel, n_frames = get_frontend(dev)(audio, lens_dev, lens)
            packed, counts = model._encoder.forward_batch(mel, n_frames)
            o = 0
            for p, c in zip(part, counts):
                out.append(((p[0], p[1]), packed[o: o + c], build_prompt(int(c), self._lang_ids), p[3]))
                o += c
        return out

    def _encode_loop(self) -> None:
        """Second thread: raw segments -> encoder features on its own lane and stream, handed over with an event."""
        import torch
        from .decoder import set_lane
        dev = self.model._decoder.device
        try:
            set_lane(self.lane + 1)
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                while True:
                    segs, stop = self._drain(block=True)
                    if self._err is not None:                 # the row worker has failed: nothing encoded here would ever be decoded
                        for seg in segs:
                            if not seg[0]["future"].done():
                                seg[0]["future"].set_exception(self._err)
                        self._fail_queued(self._err)
                        break
                    if segs:
                        try:
                            items = self._encode(segs)
                        except Exception as e:                # noqa: BLE001  -- these requests fail, the service goes on
                            for seg in segs:
                                if not seg[0]["future"].done():
                                    seg[0]["future"].set_exception(e)
                            if stop:
                                break
                            continue
                        # Hand the batch over when its encoder pass has FINISHED, not when it has been issued: the scheduler's
                        # stream would otherwise wait for this one on the device, and a stream parked on another is a blocked
                        # hardware queue that slows every kernel of the decode steps running beside it (~0.5 us each, DESIGN.md
                        # section 5) -- and the admission that follows would stall the running rows until the features exist.
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream())
                        ev.synchronize()
                        self._enc_q.put((items, ev))
                    if stop:
                        break
                torch.cuda.current_stream().synchronize()
        except BaseException as e:                            # noqa: BLE001
            self._err = self._err or e
            self._fail_queued(self._err)
        finally:
            self._enc_q.put(_STOP)

    def _arrivals(self, block: bool, side):
        """Scheduler items that are ready to be admitted, and whether no more will come."""
        import torch
        # Every item carries a holder of its encoded batch: the feature tensors were allocated on another stream than the one that
        # reads them, so they are kept alive until a synchronisation of the scheduler's stream has passed after their admission
        # (no record_stream: its deferred event queries could land inside the decode-step capture and invalidate it).
        def held(batch):
            holder = {"left": len(batch), "keep": batch}
            return [it + (holder,) for it in batch]

        if not self._encode_thread:
            segs, stop = self._drain(block)
            return (held(self._encode(segs)) if segs else []), stop
        items, stop = [], False
        try:
            got = self._enc_q.get(timeout=0.05) if block else self._enc_q.get_nowait()
            while True:
                if got is _STOP:
                    stop = True
                else:
                    batch, ev = got
                    torch.cuda.current_stream().wait_event(ev)
                    items += held(batch)
                got = self._enc_q.get_nowait()
        except queue.Empty:
            pass
        return items, stop

    def _resolve(self, key, tokens) -> None:
        req, k = key
        if isinstance(tokens, Exception):                     # the row raised the device-side error flag: this request fails alone
            if not req["future"].done():
                req["future"].set_exception(tokens)
            req["left"] -= 1
            return
        if req["future"].done():                              # an earlier segment of this request failed
            req["left"] -= 1
            return
        req["texts"][k] = self.model._decode_output(tokens)
        req["left"] -= 1
        if req["left"] == 0:
            text = " ".join(t for t in req["texts"] if t) if req["joined"] else req["texts"][0]
            req["future"].set_result(TranscriptionResult(text=text, language=self.lang, duration=req["duration"]))

    def _loop(self) -> None:
        import torch
        from .decoder import set_lane
        dev = self.model._decoder.device
        inflight: dict = {}                                  # (id(request), segment) -> key, for failing futures on error
        try:
            set_lane(self.lane)
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                sch = RowScheduler(self.model._decoder, self.rows, self.max_prompt, self.max_new, self.sampling, self.check_every)
                min_admit = max(1, self.rows // 8) if self.min_admit is None else max(1, int(self.min_admit))
                pending: deque = self._pending
                retiring: list = []                          # holders whose last item was admitted since the last synchronisation
                retired: list = []                           # ... and those one synchronisation older: dropped at the next one
                stopping = False
                import time as _t
                tm = {"arrivals_s": 0.0, "harvest_s": 0.0, "admit_s": 0.0, "run_s": 0.0, "admit_calls": 0, "run_calls": 0}
                clock = _t.perf_counter

                def settle(group):                           # every item of an encoded batch admitted -> the batch may go (two
                    for it in group:                         # synchronisations of the decode stream later)
                        it[4]["left"] -= 1
                        if it[4]["left"] == 0:
                            retiring.append(it[4])

                while True:
                    idle = not sch.active() and not pending and not sch.admitting()
                    if idle and stopping:
                        break
                    c0 = clock()
                    items, stop = self._arrivals(idle and not stopping, sch.side)
                    if not idle:
                        tm["arrivals_s"] += clock() - c0
                    stopping = stopping or stop
                    if items:
                        inflight.update({(id(it[0][0]), it[0][1]): it[0] for it in items})
                        pending.extend(items)
                    c0 = clock()
                    ended = sch.harvest()                    # synchronises the scheduler's stream if anything was queued since
                    retired.clear()                          # ... so batches whose every item was admitted before it can go
                    retired, retiring = retiring, retired
                    for key, toks in ended:
                        self._resolve(key, toks)
                        inflight.pop((id(key[0]), key[1]), None)
                    tm["harvest_s"] += clock() - c0          # includes waiting for the decode steps in flight
                    self._saturated = len(pending) > 0 and not sch.free_rows()
                    done_group = sch.poll(wait=not sch.active())         # admission in flight: activate its rows when the prefill is done
                    if done_group:
                        settle(done_group)
                    free = sch.free_rows()
                    # rows freed one at a time wait for company (a packed prefill of 4 costs little more than one of 1), unless
                    # nothing else is running or nothing more can arrive
                    need = min_admit if len(pending) < self.rows // 2 else max(min_admit, self.rows // 4)      # a backlog can wait for a
                    if pending and free and not sch.admitting() and (len(free) >= min(need, len(pending)) or not sch.active()):   # larger prefill
                        c0 = clock()
                        group = [pending.popleft() for _ in range(min(len(free), len(pending)))]
                        if self._async_admit and sch.active():
                            sch.admit_async(group)           # prefill on a second stream beside the running rows' decode steps
                        else:
                            sch.admit(group)
                            settle(group)
                        del group
                        tm["admit_s"] += clock() - c0
                        tm["admit_calls"] += 1
                        if not sch.admitting():
                            continue
                    c0 = clock()
                    sch.run()
                    tm["run_s"] += clock() - c0
                    tm["run_calls"] += 1
                sch.finish()
                torch.cuda.current_stream().synchronize()
                self.stats = dict(sch.stats, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
        except BaseException as e:                            # noqa: BLE001
            self._err = e                                     # from here on submit() refuses and the encode loop stops
            for key in inflight.values():
                if not key[0]["future"].done():
                    key[0]["future"].set_exception(e)
            for it in list(self._pending):                    # encoded, waiting for a row
                if not it[0][0]["future"].done():
                    it[0][0]["future"].set_exception(e)
            self._pending.clear()
            self._fail_queued(e)                              # raw queue and encoded queue
        else:
            if self._err is not None:                         # the encode thread died and sent its stop marker: same duty
                self._fail_queued(self._err)
