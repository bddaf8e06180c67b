"""OpenAI-compatible serving shim over the fishrt hot path: the request / response surface of the reference server
(`server/src/main.rs:60-72`: POST /v1/audio/speech, POST /v1/audio/encoding, GET /v1/voices) with a request SCHEDULER in place of the
reference's global `tokio::Mutex` around the model (`server/lib/state.rs:13`, `handlers/speech.rs:26,77`).

What is mirrored (file:line in the reference):
  * `GenerateRequest` {model, voice, input, response_format, batch_size, speaker_prompt} and the voice lookup with the default-voice
    fallback / "unconditioned" (`handlers/speech.rs:238-297`); system prompt "Speak out the provided text." for Fish 1.5 (:283-289);
  * chunked generation with the conditioning prefix kept in the KV cache between the chunks of a request (`encode_sequence(.., true)`,
    `clear_slow_caches_until(n_conditioning_tokens)`, :40), the one-shot re-roll when a chunk runs into `max_new_tokens` (:41-61), codes - 1
    for Fish <= 1.4 (:63-68), opt-in internal batching over `generate_static_batch` (`batch_size`, :72-96,141-151);
  * WAV body `audio/wav` written as `audio/wav.rs:27-58` (:168-176); errors as HTTP 500 with the message string (`handlers/error.rs:17-31`);
  * `/v1/voices` -> list of names (`handlers/supported_voices.rs`), `/v1/audio/encoding` -> `.npy` of the (8, T) codes, optional `id` +
    `prompt` query registering the voice, duplicate id -> error (`handlers/encode_speech.rs:36-94`).
What is NOT: Opus / Ogg streaming (`audio/opus.rs`; out of scope) -- `response_format: "opus"` answers 501 unless the
caller supplies an encoder; audio decoding other than WAV; the text cleaner is a compact restatement (`preprocess_text`, pluggable).

Scheduler.  One worker thread per LM handle (= per GPU) drains a queue of chunk jobs.  A job whose conditioning prefix equals the one
sitting in the handle's KV cache skips the prefix (the reference's `assume_kv_cache`); jobs of different voices simply invalidate it
(the reference's single mutex serialises requests and would reuse a stale prefix only by construction of one request at a time).  When
several jobs are in flight (or a request asks for `batch_size`) on a `max_batch > 1` handle, they share the static-batch decode step
(the weights are streamed once per step for all rows) through CONTINUOUS batching: the handle's rows are request slots of a session
(`fs_lm_session_*`), a job joins as soon as a slot is free and leaves when it is done -- token-level admission, no lock-step batches.
A lone job takes the batch-1 path with its persistent decode kernels.  `Scheduler(continuous=False)` keeps the lock-step variant
(jobs waiting together go through one `generate_static_batch` call).
"""
import hashlib
import io
import json
import queue
import secrets
import threading
import time
from concurrent.futures import Future

import numpy as np

from . import prompt as fprompt
from . import wav as fwav

FISH15_SYSPROMPT = "Speak out the provided text."


class SamplingArgs:  # sampling/mod.rs:29-34; defaults of server/lib/utils/load.rs:116-125
    def __init__(self, temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.4):
        self.temp, self.top_p, self.top_k, self.repetition_penalty = temp, top_p, top_k, repetition_penalty

    def kw(self):
        return dict(temp=self.temp, top_p=self.top_p, top_k=self.top_k, repetition_penalty=self.repetition_penalty)


# ---- text/clean.rs (compact restatement: symbol map, sentence split, combine short / split long by script-independent Latin thresholds)
_SYMBOLS = {"“": '"', "”": '"', "‘": "'", "’": "'", "…": "...", "«": '"', "»": '"', "​": "", "‌": "", "‍": "", "﻿": "",
            "。": ".", "、": ", ", "！": "!", "？": "?", "「": '"', "」": '"', "『": '"', "』": '"', "・": "", "：": ",", "；": ",",
            "（": "", "）": "", "【": "", "】": ""}


def preprocess_text(text, combine=150, split=400):
    t = text.strip()
    for a, b in _SYMBOLS.items():
        t = t.replace(a, b)
    t = "".join(c for c in t if not (0x1F300 <= ord(c) <= 0x1F9FF))
    t = t.replace(" - ", "—")
    sents, cur = [], ""
    for i, ch in enumerate(t):
        cur += ch
        if ch in ".!?\n" and (i + 1 == len(t) or t[i + 1] not in ".!?") and cur.strip():  # a run of marks ("...", "?!") ends ONE sentence
            sents.append(cur.strip())
            cur = ""
    if cur.strip():
        sents.append(cur.strip())
    chunks, acc = [], ""
    for s in sents:
        while len(s) > split:  # a sentence longer than the split threshold is cut at the last space before it
            cut = s.rfind(" ", 0, split)
            cut = cut if cut > 0 else split
            (chunks.append(acc) if acc else None)
            acc = ""
            chunks.append(s[:cut].strip())
            s = s[cut:].strip()
        if acc and len(acc) + 1 + len(s) > combine:
            chunks.append(acc)
            acc = s
        else:
            acc = (acc + " " + s).strip()
    if acc:
        chunks.append(acc)
    return chunks


class LMState:  # server/lib/state.rs:12-21
    def __init__(self, lm, tokenizer, voices, default_voice, model_type=fprompt.FISH_1_5, default_sampling_args=None, max_new_tokens=1792,
                 max_batch=1, seed_source=None):
        self.lm, self.tokenizer, self.voices, self.default_voice, self.model_type = lm, tokenizer, dict(voices), default_voice, model_type
        self.default_sampling_args = default_sampling_args or SamplingArgs(repetition_penalty=1.4 if model_type == fprompt.FISH_1_5 else 1.2)
        self.max_new_tokens, self.max_batch = max_new_tokens, max_batch
        self.voices_lock = threading.Lock()
        # single_batch.rs:46: every generate call draws a fresh sampler seed (`rand::random::<u64>()`); tests pass a deterministic source
        self.seed_source = seed_source or (lambda: secrets.randbits(64))


class AppState:  # server/lib/state.rs:23-29
    def __init__(self, lm_state, codec, sample_rate=44100, opus_encoder=None, preprocess=preprocess_text, batch_window_s=0.002,
                 continuous=True, auto_batch=False):
        self.lm, self.codec, self.sample_rate, self.opus_encoder, self.preprocess = lm_state, codec, sample_rate, opus_encoder, preprocess
        self.auto_batch = auto_batch  # True: every chunk may join the batching session (batch sampling semantics) without `batch_size`
        self.scheduler = Scheduler(lm_state, batch_window_s, continuous)


class _Job:
    def __init__(self, cond, body, n_cond, allow_batch):
        self.cond, self.body, self.n_cond, self.allow_batch, self.future = cond, body, n_cond, allow_batch, Future()
        self.cond_key = hashlib.sha1(cond.tobytes()).hexdigest() if cond is not None else None

    def full_prompt(self):
        return self.body if self.cond is None else np.ascontiguousarray(np.concatenate([self.cond, self.body], 1))


_STOP = object()  # Scheduler.close() sentinel


class Scheduler:
    """Replaces `state.lm.model.lock().await`: chunk jobs from all requests in one queue, one worker per handle."""

    def __init__(self, lm_state, batch_window_s=0.002, continuous=True, step_frames=8):
        self.s, self.q, self.window = lm_state, queue.Queue(), batch_window_s
        self.continuous, self.step_frames = continuous, step_frames
        self.cached_key = None
        self.stats = dict(jobs=0, single=0, batched_rows=0, batches=0, prefix_hits=0, rerolls=0)
        self._stop = False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def submit(self, cond, body, n_cond, allow_batch):
        j = _Job(cond, body, n_cond, allow_batch)
        self.q.put(j)
        return j.future

    def close(self):
        self._stop = True
        self.q.put(_STOP)
        self.th.join(timeout=10)

    # -- worker
    def _run(self):
        if self.s.max_batch > 1 and self.continuous and hasattr(self.s.lm, "session"):
            return self._run_continuous()
        while True:
            j = self.q.get()
            if j is _STOP or self._stop:
                return
            batch = [j]
            if self.s.max_batch > 1 and j.allow_batch:  # gather what else is waiting (or arrives within the window), up to max_batch rows
                deadline = time.perf_counter() + self.window
                while len(batch) < self.s.max_batch:
                    try:
                        n = self.q.get(timeout=max(0.0, deadline - time.perf_counter()))
                    except queue.Empty:
                        break
                    if n is _STOP:
                        self.q.put(_STOP)
                        break
                    batch.append(n)  # (a job that must not be batched still shares the gather; it runs alone below)
            runs_alone = [b for b in batch if not b.allow_batch]
            together = [b for b in batch if b.allow_batch]
            try:
                if len(together) >= 2:
                    self._batched(together)
                else:
                    runs_alone = together + runs_alone
                for b in runs_alone:
                    self._single(b)
            except BaseException as e:  # every waiting request gets the error (AppError -> HTTP 500)
                for b in batch:
                    if not b.future.done():
                        b.future.set_exception(e)

    def _run_continuous(self):
        """Continuous batching (fishrt.h fs_lm_session_*): the handle's max_batch rows are request slots; a chunk job joins as soon as a slot
        is free (its prompt is prefilled between two decode steps) and leaves when it samples <|im_end|> or runs out of budget -- no job waits
        for a batch to form or for the slowest row of its batch.  A lone job (nothing else live or waiting) takes the batch-1 path instead:
        persistent decode kernels, repetition penalty, conditioning-prefix reuse.  Jobs that must not be batched drain the session first."""
        lm, sess, live, held = self.s.lm, None, {}, None

        def fail_all(e):
            for jb in list(live.values()) + ([held] if held is not None else []):
                if not jb.future.done():
                    jb.future.set_exception(e)
            live.clear()

        stopping = False
        while True:
            try:
                # ---- intake: one job at a time is held until a slot (or the drained handle) is free for it
                if held is None and not stopping:
                    if live:
                        try:
                            held = self.q.get_nowait()
                        except queue.Empty:
                            pass
                    else:
                        if sess is not None:  # idle: give the handle back (its other entry points work between bursts)
                            sess.close()
                            sess = None
                        held = self.q.get()
                    if held is _STOP:
                        held, stopping = None, True
                if stopping and not live:
                    if sess is not None:
                        sess.close()
                    return
                if held is not None:
                    lone = not live and self.q.empty()
                    if not held.allow_batch or lone:
                        if not live:  # drain first; then the batch-1 path
                            if sess is not None:
                                sess.close()
                                sess = None
                            j, held = held, None
                            self._single(j)
                            continue
                    else:
                        if sess is None:
                            sa = self.s.default_sampling_args
                            seed = self.s.seed_source() & (2**64 - 1)
                            # handles with 2 / 4 / 8 slots and the request-row kernels: the slots keep the batch-1 semantics of _single
                            # (repetition penalty, one LogitsProcessor stream per job) while they share every decode launch (FS_SESSION_ROWS)
                            rows = (getattr(lm, "max_batch", 0) in (2, 4, 8) and hasattr(lm, "rows_supported")
                                    and lm.rows_supported(lm.max_batch, **sa.kw()))
                            try:
                                sess = lm.session(temp=sa.temp, top_p=sa.top_p, top_k=sa.top_k, seed=seed, rows=True,
                                                  repetition_penalty=sa.repetition_penalty) if rows else None
                            except Exception:  # (f32 / fp8 / Fish <= 1.4 handles, or the device's persistent kernels are taken)
                                sess = None
                            if sess is None:
                                sess = lm.session(temp=sa.temp, top_p=sa.top_p, top_k=sa.top_k, seed=seed)
                            else:
                                self.stats["row_sessions"] = self.stats.get("row_sessions", 0) + 1
                            self.cached_key = None
                        try:
                            slot = sess.add(held.full_prompt(), self.s.max_new_tokens)
                        except BaseException as e:  # a bad request (prompt longer than max_seq_len, ...) fails ALONE: the live slots go on
                            held.future.set_exception(e)
                            held = None
                            continue
                        if slot is not None:
                            live[slot] = held
                            held = None
                            self.stats["jobs"] += 1
                            self.stats["batched_rows"] += 1
                            self.stats["peak_live"] = max(self.stats.get("peak_live", 0), len(live))
                            continue  # admit more before stepping
                        if not live:  # nothing will ever free a slot / KV pages for it: the batch-1 path (or its error) instead of spinning
                            sess.close()
                            sess = None
                            j, held = held, None
                            self._single(j)
                            continue
                # ---- one scheduling quantum of decode steps for every live slot
                if live:
                    sess.step(self.step_frames)
                    self.stats["batches"] += 1
                    for slot in list(live):
                        n, done = sess.poll(slot, codes=False)
                        if done:
                            codes, _ = sess.poll(slot)
                            sess.release(slot)
                            jb = live.pop(slot)
                            try:  # (jb has left `live`: from here on fail_all no longer sees it, so its future is resolved right here)
                                if codes.shape[1] >= self.s.max_new_tokens:  # speech.rs:41-61 "Failed generation suspected. Rerolling once":
                                    self.stats["rerolls"] += 1               # the re-roll takes the batch-1 path once the session has drained
                                    jb.allow_batch, jb.reroll = False, True
                                    if stopping:  # nothing takes jobs off the queue any more: fail it instead of orphaning its future
                                        raise RuntimeError("the scheduler is shutting down; the re-roll of a failed generation was dropped")
                                    self.q.put(jb)
                                else:
                                    jb.future.set_result(self._codes_out(codes))
                            except BaseException as e:
                                if not jb.future.done():
                                    jb.future.set_exception(e)
            except BaseException as e:  # every in-flight request gets the error (AppError -> HTTP 500); the session is rebuilt
                fail_all(e)
                held = None
                try:
                    if sess is not None:
                        sess.close()
                except BaseException:
                    pass
                sess = None

    def _codes_out(self, codes):
        if self.s.model_type != fprompt.FISH_1_5:  # speech.rs:63-68: Fish <= 1.4 codes are shifted by one
            if (codes == 0).any():  # the reference's u32 subtraction would wrap and its codebook gather then fails: surface it the same way
                raise RuntimeError("Fish <= 1.4 generation produced code 0 (no codebook entry -1)")
            codes = codes - np.uint32(1)
        return codes

    def _single(self, j):
        try:
            lm, sa = self.s.lm, self.s.default_sampling_args
            self.stats["jobs"] += 1
            self.stats["single"] += 1
            if j.cond_key is not None and j.cond_key == self.cached_key and lm.curr_kv_size() == j.n_cond:
                prompt = j.body  # the conditioning prefix is in the KV cache (encode_sequence(.., assume_kv_cache = true))
                self.stats["prefix_hits"] += 1
            else:
                lm.clear_slow_layer_caches()
                prompt = j.full_prompt()
            codes = lm.generate_blocking(prompt, self.s.max_new_tokens, seed=self.s.seed_source(), **sa.kw())
            lm.clear_slow_caches_until(j.n_cond)  # speech.rs:40
            self.cached_key = j.cond_key if j.cond is not None else None
            if codes.shape[1] == self.s.max_new_tokens and getattr(j, "reroll", False):  # this WAS the re-roll of a session job
                raise RuntimeError("Encoded input failed for second time. Bailing")
            if codes.shape[1] == self.s.max_new_tokens:  # speech.rs:41-61: "Failed generation suspected. Rerolling once"
                self.stats["rerolls"] += 1
                lm.clear_slow_layer_caches()
                codes2 = lm.generate_blocking(j.full_prompt(), self.s.max_new_tokens, seed=self.s.seed_source(), **sa.kw())
                lm.clear_slow_caches_until(j.n_cond)
                if codes2.shape[1] == self.s.max_new_tokens:
                    raise RuntimeError("Encoded input failed for second time. Bailing")
                codes = codes2
            j.future.set_result(self._codes_out(codes))
        except BaseException as e:
            self.cached_key = None
            j.future.set_exception(e)

    def _batched(self, jobs):
        lm, sa = self.s.lm, self.s.default_sampling_args
        self.stats["jobs"] += len(jobs)
        self.stats["batches"] += 1
        self.stats["batched_rows"] += len(jobs)
        kw = sa.kw()
        # (only where the C side really takes the row kernels: on any other handle / sampler setting fs_lm_generate_multi would run the jobs
        # one after the other -- N times the latency of ONE lock-step generate_static_batch, which streams the weights once per step)
        if (2 <= len(jobs) <= 8 and len(jobs) <= getattr(lm, "max_batch", 0) and hasattr(lm, "generate_multi")
                and hasattr(lm, "rows_supported") and lm.rows_supported(len(jobs), **kw)):
            # request rows (fishrt.h fs_lm_generate_multi): every job keeps the batch-1 semantics of _single -- its own repetition-penalty
            # window and sampler stream -- while one persistent launch serves all of them
            self.stats["row_batches"] = self.stats.get("row_batches", 0) + 1
            outs = lm.generate_multi([j.full_prompt() for j in jobs], self.s.max_new_tokens, seeds=[self.s.seed_source() & (2**64 - 1) for _ in jobs], **kw)
            lm.clear_slow_layer_caches()
            self.cached_key = None
            for j, o in zip(jobs, outs):
                j.future.set_result(self._codes_out(o))
            return
        kw.pop("repetition_penalty")  # the batch path's repetition penalty is a no-op in the reference (static_batch.rs:204-206)
        outs = lm.generate_static_batch([j.full_prompt() for j in jobs], self.s.max_new_tokens, **kw)
        lm.clear_slow_layer_caches()  # speech.rs:88
        self.cached_key = None
        for j, o in zip(jobs, outs):
            j.future.set_result(self._codes_out(o))


# ---- request handling (framework-free core; `make_app` wraps it in FastAPI)
class AppError(Exception):  # handlers/error.rs:17-31 -> HTTP 500 with the message
    pass


def _prompts_for_request(state, req):
    s = state.lm
    voice = req.get("voice")
    if voice == "unconditioned":
        emb = None
    else:
        with s.voices_lock:
            emb = s.voices.get(voice, s.default_voice)
    chunks = state.preprocess(req.get("input", ""))
    enc = fprompt.PromptEncoder(s.tokenizer, s.lm.cfg["num_codebooks"], s.model_type)
    sysprompt = req.get("speaker_prompt") or (FISH15_SYSPROMPT if s.model_type == fprompt.FISH_1_5 else None)
    n_cond, _ = enc.encode_sequence(chunks, sysprompt, emb, True)
    sys_arr = enc.encode_text("system", sysprompt) if sysprompt is not None else None
    parts = [p for p in (sys_arr, emb) if p is not None]
    cond = np.ascontiguousarray(np.concatenate(parts, 1)) if parts else None
    assistant = enc.encode_vq(None)
    bodies = [np.ascontiguousarray(np.concatenate([enc.encode_text("user", c), assistant], 1)) for c in chunks]
    return n_cond, cond, bodies


def generate_speech(state, req):
    """POST /v1/audio/speech -> (status, content_type, body bytes | iterator of bytes)"""
    for k in ("model", "voice", "input"):
        if k not in req:
            return 422, "application/json", json.dumps({"detail": f"missing field `{k}`"}).encode()
    try:
        n_cond, cond, bodies = _prompts_for_request(state, req)
        fmt = req.get("response_format")
        if fmt == "opus":
            if state.opus_encoder is None:
                return 501, "application/json", json.dumps({"detail": "Opus / Ogg streaming is outside this shim (audio/opus.rs); use the default WAV "
                                                                      "response or response_format 'pcm'"}).encode()
        # like the reference (speech.rs:72-96) chunks are batched only when the request asks for it (`batch_size` > 1): the batch paths sample
        # with BatchedLogitsProcessor semantics and no repetition penalty, so what a client hears must not depend on the server's load.
        # `auto_batch` (server option, off by default) lets every chunk join the continuous-batching session when other work is in flight.
        want_batch = int(req.get("batch_size") or 1) > 1 or getattr(state, "auto_batch", False)
        futs = [state.scheduler.submit(cond, b, n_cond, state.lm.max_batch > 1 and want_batch) for b in bodies]

        def pcm_chunks():
            for f in futs:
                codes = f.result()
                pcm = state.codec.decode(np.ascontiguousarray(codes[None]))[0, 0]  # (an out-of-range code is an error, as in the reference)
                yield pcm

        if fmt == "pcm":  # extension: chunked little-endian s16 PCM at the codec rate, one HTTP chunk per text chunk
            return 200, "audio/pcm", (fwav.pcm_to_i16(p).tobytes() for p in pcm_chunks())
        if fmt == "opus":
            enc = state.opus_encoder
            return 200, "audio/ogg", (b for p in pcm_chunks() for b in enc(p, state.sample_rate))
        all_pcm = np.concatenate(list(pcm_chunks())) if futs else np.zeros(0, np.float32)
        buf = io.BytesIO()
        fwav.write_pcm_as_wav(buf, all_pcm, state.sample_rate)
        return 200, "audio/wav", buf.getvalue()
    except BaseException as e:
        return 500, "text/plain", f"Something went wrong: {e}".encode()


def supported_voices(state):
    with state.lm.voices_lock:
        return list(state.lm.voices.keys())


def _read_wav(data):
    """RIFF/WAVE PCM16 / PCM32 / float32 -> (f32 (channels, n), sample_rate); other containers are outside the shim"""
    import struct
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise AppError("only RIFF/WAVE uploads are decoded by this shim")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise AppError("malformed WAV upload")
    tag, ch, sr, _, _, bits = fmt
    if tag == 1 and bits == 16:
        a = np.frombuffer(pcm, "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        a = np.frombuffer(pcm, "<i4").astype(np.float32) / 2147483648.0
    elif tag == 3 and bits == 32:
        a = np.frombuffer(pcm, "<f4").astype(np.float32)
    else:
        raise AppError(f"unsupported WAV encoding (format {tag}, {bits} bits)")
    n = a.size // ch
    return a[: n * ch].reshape(n, ch).T.copy(), sr


def encode_speaker(state, file_bytes, params):
    """POST /v1/audio/encoding (handlers/encode_speech.rs:36-94) -> (status, content_type, body)"""
    try:
        audio, sr = _read_wav(file_bytes)
        if audio.shape[0] > 1:
            audio = audio.mean(0, keepdims=True)
        if sr != state.sample_rate:
            from math import gcd
            from scipy.signal import resample_poly
            g = gcd(int(sr), int(state.sample_rate))
            audio = resample_poly(audio, state.sample_rate // g, sr // g, axis=1).astype(np.float32)
        codes = state.codec.encode(np.ascontiguousarray(audio[None], np.float32))[0]  # (8, T)
        vid, text = params.get("id"), params.get("prompt")
        if vid is not None and text is not None:
            s = state.lm
            enc = fprompt.PromptEncoder(s.tokenizer, s.lm.cfg["num_codebooks"], s.model_type)
            with s.voices_lock:
                if vid in s.voices:
                    raise AppError(f"ID already exists on server: {vid}")
                s.voices[vid] = enc.encode_conditioning_prompt(text, codes.astype(np.uint32))
        buf = io.BytesIO()
        np.save(buf, codes)
        return 200, "application/x-npy", buf.getvalue()
    except BaseException as e:
        return 500, "text/plain", f"Something went wrong: {e}".encode()


def _multipart_first_file(content_type, body):
    import email.parser
    msg = email.parser.BytesParser().parsebytes(b"Content-Type: " + content_type.encode() + b"\r\n\r\n" + body)
    if not msg.is_multipart():
        raise AppError("No file provided")
    for part in msg.get_payload():
        payload = part.get_payload(decode=True)
        if payload:
            return payload
    raise AppError("No file provided")


def make_app(state):
    """FastAPI application with the reference's three routes (server/src/main.rs:60-72)."""
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse, Response, StreamingResponse

    app = FastAPI(title="fishrt")

    @app.post("/v1/audio/speech")
    async def speech(request: Request):
        import asyncio
        try:
            req = await request.json()
        except Exception:
            return JSONResponse({"detail": "body must be a JSON object"}, status_code=422)
        status, ctype, body = await asyncio.get_event_loop().run_in_executor(None, generate_speech, state, req)
        if isinstance(body, (bytes, bytearray)):
            return Response(content=body, status_code=status, media_type=ctype)
        return StreamingResponse(body, status_code=status, media_type=ctype)

    @app.post("/v1/audio/encoding")
    async def encoding(request: Request):
        import asyncio
        body = await request.body()
        try:
            data = _multipart_first_file(request.headers.get("content-type", ""), body)
        except AppError as e:
            return Response(content=f"Something went wrong: {e}".encode(), status_code=500, media_type="text/plain")
        status, ctype, out = await asyncio.get_event_loop().run_in_executor(None, encode_speaker, state, data, dict(request.query_params))
        return Response(content=out, status_code=status, media_type=ctype)

    @app.get("/v1/voices")
    async def voices():
        return JSONResponse(supported_voices(state))

    return app
