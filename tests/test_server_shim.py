"""CPU: the OpenAI-compatible serving shim (fishrt/server.py) -- request / response schema of the reference handlers
(server/lib/handlers/{speech,encode_speech,supported_voices}.rs), the scheduler that replaces the global model mutex, KV-prefix reuse,
dynamic batching, the re-roll rule and the error mapping.  The LM / codec handles are stand-ins that record how they are driven (the
real ones need an MI355X; tests/test_server_gpu.py runs the same app on the device)."""
import io
import json
import struct
import threading

import numpy as np
import pytest

from fishrt import prompt as fprompt
from fishrt import server


class Tok:  # byte-level stand-in: ids = utf-8 bytes; the two specials the prompt encoder asks for
    def encode(self, text):
        return list(text.encode())

    def token_to_id(self, token):
        return {"<|semantic:0|>": 1000, "<|semantic|>": 5}.get(token)


class FakeLM:
    def __init__(self, max_new_tokens=64, slow=0.0):
        self.cfg = dict(num_codebooks=8)
        self.kv, self.calls, self.lock, self.slow, self.M = 0, [], threading.Lock(), slow, max_new_tokens
        self.frames_for = lambda prompt: 3 + int(prompt[0, -1]) % 5

    def clear_slow_layer_caches(self):
        self.kv = 0

    def clear_slow_caches_until(self, pos):
        self.kv = min(self.kv, pos)

    def curr_kv_size(self):
        return self.kv

    def _gen(self, prompt):
        n = self.frames_for(prompt)
        return np.full((8, n), int(prompt[0, -5]) % 1000, np.uint32)

    def generate_blocking(self, prompt, max_new_tokens, **kw):
        assert self.lock.acquire(blocking=False), "two calls in flight on one handle"
        try:
            import time
            time.sleep(self.slow)
            self.calls.append(("single", prompt.shape[1], self.kv, dict(kw)))
            out = self._gen(prompt)
            self.kv += prompt.shape[1] + out.shape[1] - 1
            return out
        finally:
            self.lock.release()

    def generate_static_batch(self, prompts, max_new_tokens, **kw):
        assert self.lock.acquire(blocking=False), "two calls in flight on one handle"
        try:
            self.calls.append(("batch", [p.shape[1] for p in prompts], dict(kw)))
            self.kv = 99
            return [self._gen(p) for p in prompts]
        finally:
            self.lock.release()


    def session(self, **kw):
        return FakeSession(self, kw)


class FakeSession:
    """stand-in for fishrt.lm.Session: 4 slots; a slot emits one frame per step until its request's length is reached"""

    def __init__(self, lm, kw):
        assert lm.lock.acquire(blocking=False), "session opened while a call is in flight on the handle"
        self.lm, self.slots, self.kw = lm, {}, kw
        lm.calls.append(("session", dict(kw)))

    def close(self):
        self.lm.calls.append(("session_end", len(self.slots)))
        self.lm.lock.release()

    def add(self, prompt, max_new_tokens):
        free = [i for i in range(4) if i not in self.slots]
        if not free:
            return None
        self.slots[free[0]] = [self.lm._gen(prompt), 0]
        self.lm.calls.append(("add", prompt.shape[1], len(self.slots)))
        return free[0]

    def step(self, n):
        live = 0
        for v in self.slots.values():
            v[1] = min(v[0].shape[1], v[1] + n)
            live += v[1] < v[0].shape[1]
        return live

    def poll(self, slot, codes=True):
        full, n = self.slots[slot]
        return (full[:, :n].copy(), n == full.shape[1]) if codes else (n, n == full.shape[1])

    def release(self, slot):
        del self.slots[slot]


class FakeCodec:
    def decode(self, codes):
        b, c, t = codes.shape
        return np.full((b, 1, 2048 * t), 0.25, np.float32)

    def encode(self, pcm):
        return np.full((1, 8, max(1, pcm.shape[2] // 2048)), 7, np.uint32)


def _state(max_batch=1, continuous=True, auto_batch=False, **lm_kw):
    tok = Tok()
    enc = fprompt.PromptEncoder(tok, 8, fprompt.FISH_1_5)
    default = enc.encode_conditioning_prompt("hello there", np.full((8, 4), 3, np.uint32))
    alice = enc.encode_conditioning_prompt("i am alice", np.full((8, 6), 9, np.uint32))
    lm = FakeLM(**lm_kw)
    ls = server.LMState(lm, tok, {"default": default, "alice": alice}, default, max_new_tokens=lm.M, max_batch=max_batch)
    return server.AppState(ls, FakeCodec(), batch_window_s=0.05, continuous=continuous, auto_batch=auto_batch), lm


def _client(state):
    from fastapi.testclient import TestClient
    return TestClient(server.make_app(state))


def test_speech_returns_wav_and_reuses_the_conditioning_prefix():
    state, lm = _state()
    c = _client(state)
    text = "First sentence is here and it is long enough to stand alone as a chunk of text for the model to speak aloud, yes it is. " * 2 + \
           "Second one follows, also long enough to be its own chunk because the combine threshold is one hundred and fifty characters. " * 2
    r = c.post("/v1/audio/speech", json=dict(model="tts-1", voice="alice", input=text))
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    b = r.content
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE" and struct.unpack("<I", b[24:28])[0] == 44100 and struct.unpack("<H", b[34:36])[0] == 16
    n_samples = struct.unpack("<I", b[40:44])[0] // 2
    singles = [k for k in lm.calls if k[0] == "single"]
    assert len(singles) >= 2 and n_samples == 2048 * sum(3 + 0 for _ in []) or n_samples % 2048 == 0
    # chunk 0 carries the conditioning prompt (KV empty); later chunks find it cached and send only the user turn
    assert singles[0][2] == 0 and singles[0][1] > singles[1][1] and singles[1][2] > 0
    assert state.scheduler.stats["prefix_hits"] == len(singles) - 1
    # server defaults (load.rs:116-125): top-k 256, repetition penalty 1.4 for Fish 1.5
    kw = dict(singles[0][3])
    seeds = [k[3]["seed"] for k in singles]  # single_batch.rs:46: a fresh random sampler seed per generate call
    assert kw.pop("seed") in range(2**64) and len(set(seeds)) == len(seeds)
    assert kw == dict(temp=0.7, top_p=0.8, top_k=256, repetition_penalty=1.4)
    assert np.frombuffer(b[44:48], "<i2")[0] == int(0.25 * 32767)
    state.scheduler.close()


def test_voices_unknown_voice_falls_back_and_unconditioned():
    state, lm = _state()
    c = _client(state)
    assert sorted(c.get("/v1/voices").json()) == ["alice", "default"]
    c.post("/v1/audio/speech", json=dict(model="x", voice="nobody", input="Hi."))
    c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Hi."))
    c.post("/v1/audio/speech", json=dict(model="x", voice="unconditioned", input="Hi."))
    a, b, u = [k[1] for k in lm.calls]
    # unknown voice -> the default voice (speech.rs:264-271); the next request with the SAME conditioning finds it cached (prefix reuse across
    # requests: only the user turn is sent); unconditioned -> system prompt only, a different prefix -> full (shorter) prompt
    assert b < a and lm.calls[1][2] == a - b and b < u < a and lm.calls[2][2] == 0
    assert c.post("/v1/audio/speech", json=dict(model="x", input="Hi.")).status_code == 422
    r = c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Hi.", response_format="opus"))
    assert r.status_code == 501
    r = c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Hi.", response_format="pcm"))
    assert r.status_code == 200 and r.headers["content-type"].startswith("audio/pcm") and len(r.content) % 4096 == 0
    state.scheduler.close()


def _fire(c, n, **extra):
    results = {}

    def go(i):
        results[i] = c.post("/v1/audio/speech", json=dict(model="x", voice="alice" if i % 2 else "default", input=f"Request number {i}.", **extra))

    ths = [threading.Thread(target=go, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return results


def test_concurrent_requests_share_a_continuous_batching_session():
    """max_batch > 1: jobs in flight together become slots of one session (fs_lm_session_*); more jobs than slots wait for a release; the
    session is closed when the burst is over and the handle's batch-1 path works again"""
    state, lm = _state(max_batch=8, slow=0.02, auto_batch=True)
    lm.frames_for = lambda prompt: 20 + int(prompt[0, -5]) % 30  # several scheduling quanta per job
    c = _client(state)
    results = _fire(c, 7)
    assert all(r.status_code == 200 for r in results.values())  # FakeLM asserts that no two calls overlap on the handle
    st = state.scheduler.stats
    assert st["jobs"] == 7 and st["batched_rows"] >= 4 and st["peak_live"] >= 2, st
    adds = [k for k in lm.calls if k[0] == "add"]
    assert max(k[2] for k in adds) <= 4 and len(adds) == st["batched_rows"]          # the fake has 4 slots: later jobs joined after a release
    sess = [k for k in lm.calls if k[0] == "session"]
    assert sess and all("repetition_penalty" not in k[1] for k in sess)               # (a no-op on the batch path, static_batch.rs:204-206)
    assert [k[0] for k in lm.calls].count("session") == [k[0] for k in lm.calls].count("session_end")
    # afterwards a lone request takes the batch-1 path again
    n_single = st["single"]
    assert c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Alone.")).status_code == 200
    assert state.scheduler.stats["single"] == n_single + 1 and lm.calls[-1][0] == "single"
    state.scheduler.close()


def test_batching_is_opt_in_like_the_reference():
    """speech.rs:72-96: chunks are batched only when the request passes batch_size; without it concurrent requests keep the full
    single-sequence sampling semantics (repetition penalty, re-roll, prefix reuse) whatever the load"""
    state, lm = _state(max_batch=8, slow=0.01)
    c = _client(state)
    results = _fire(c, 5)
    assert all(r.status_code == 200 for r in results.values())
    assert state.scheduler.stats["single"] == 5 and state.scheduler.stats.get("batched_rows", 0) == 0
    assert all(k[0] == "single" and "repetition_penalty" in k[3] for k in lm.calls)
    results = _fire(c, 5, batch_size=4)
    assert all(r.status_code == 200 for r in results.values())
    assert state.scheduler.stats["batched_rows"] >= 2 and "session" in [k[0] for k in lm.calls]
    state.scheduler.close()


def test_a_bad_request_fails_alone_and_a_request_no_slot_can_ever_hold_takes_the_single_path():
    state, lm = _state(max_batch=8, slow=0.02, auto_batch=True)
    lm.frames_for = lambda prompt: 30
    real_add = FakeSession.add
    seen = {"n": 0}

    def add(self, prompt, max_new_tokens):
        seen["n"] += 1
        if seen["n"] == 3:
            raise RuntimeError("prompt exceeds max_seq_len (dual_ar.rs:623-624)")
        return real_add(self, prompt, max_new_tokens)
    FakeSession.add = add
    try:
        c = _client(state)
        results = _fire(c, 6)
    finally:
        FakeSession.add = real_add
    codes = sorted(r.status_code for r in results.values())
    assert codes == [200] * 5 + [500], codes  # exactly the offending request fails; the live slots finish
    assert any(b"max_seq_len" in r.content for r in results.values())
    # add() == None with nothing live (the KV pool can never hold the request): the job is handed to the batch-1 path, no busy loop
    FakeSession.add = lambda self, prompt, max_new_tokens: None
    try:
        r = c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Too big for any slot."))
    finally:
        FakeSession.add = real_add
    assert r.status_code == 200 and lm.calls[-1][0] == "single"
    state.scheduler.close()


def test_session_jobs_that_hit_max_new_tokens_are_rerolled_on_the_single_path():
    state, lm = _state(max_batch=8, max_new_tokens=6, auto_batch=True, slow=0.01)
    seen = {}

    def frames(prompt):  # the first generation of every request runs into the budget, its re-roll does not
        raw = bytes(int(v) & 0xFF for v in prompt[0])  # the stand-in tokenizer is byte-level: the request's number identifies it,
        k = raw[raw.rindex(b"number ") + 7]           # whether or not the conditioning prefix is part of the prompt
        seen[k] = seen.get(k, 0) + 1
        return 6 if seen[k] == 1 else 4
    lm.frames_for = frames
    c = _client(state)
    results = _fire(c, 3)
    assert all(r.status_code == 200 for r in results.values()), [r.content for r in results.values()]
    assert state.scheduler.stats["rerolls"] == 3 and "add" in [k[0] for k in lm.calls]
    assert all(v == 2 for v in seen.values())  # one re-roll each, session jobs included
    state.scheduler.close()


def test_lock_step_variant_runs_small_batches_as_request_rows():
    """2..8 waiting jobs on a handle with fs_lm_generate_multi: one call, every job keeps its batch-1 semantics (repetition penalty passed,
    one sampler seed per job) instead of the static batch's"""
    state, lm = _state(max_batch=8, continuous=False, slow=0.02, auto_batch=True)
    lm.max_batch = 8

    def generate_multi(prompts, max_new_tokens, seeds=None, **kw):
        assert lm.lock.acquire(blocking=False), "two calls in flight on one handle"
        try:
            lm.calls.append(("multi", [p.shape[1] for p in prompts], list(seeds), dict(kw)))
            return [lm._gen(p) for p in prompts]
        finally:
            lm.lock.release()
    lm.generate_multi = generate_multi
    lm.rows_supported = lambda n, **kw: True
    c = _client(state)
    results = _fire(c, 6)
    assert all(r.status_code == 200 for r in results.values())
    multi = [k for k in lm.calls if k[0] == "multi"]
    assert multi and "batch" not in [k[0] for k in lm.calls]
    assert all("repetition_penalty" in k[3] and len(set(k[2])) == len(k[2]) == len(k[1]) for k in multi)
    assert state.scheduler.stats.get("row_batches", 0) == len(multi)
    state.scheduler.close()


def test_lock_step_variant_keeps_the_static_batch_where_the_row_kernels_do_not_apply():
    """fs_lm_rows_supported == 0 (fp8 / f32 / Fish <= 1.4 handle, sampler outside the in-launch sampler): generate_multi would run the jobs
    one after the other, so the scheduler must keep ONE lock-step generate_static_batch and must not count a row batch"""
    state, lm = _state(max_batch=8, continuous=False, slow=0.02, auto_batch=True)
    lm.max_batch = 8
    asked = []

    def generate_multi(prompts, max_new_tokens, seeds=None, **kw):
        raise AssertionError("generate_multi taken on a handle that cannot run request rows")
    lm.generate_multi = generate_multi
    lm.rows_supported = lambda n, **kw: asked.append((n, dict(kw))) or False
    c = _client(state)
    results = _fire(c, 6)
    assert all(r.status_code == 200 for r in results.values())
    assert asked and all(2 <= n <= 8 and "temp" in kw and "top_k" in kw for n, kw in asked)
    assert "batch" in [k[0] for k in lm.calls] and "multi" not in [k[0] for k in lm.calls]
    assert state.scheduler.stats.get("row_batches", 0) == 0
    state.scheduler.close()


def test_lock_step_variant_batches_waiting_jobs():
    state, lm = _state(max_batch=8, continuous=False, slow=0.02, auto_batch=True)
    c = _client(state)
    results = _fire(c, 6)
    assert all(r.status_code == 200 for r in results.values())
    st = state.scheduler.stats
    assert st["jobs"] == 6 and st["batches"] >= 1 and st["batched_rows"] >= 2, st
    assert "batch" in [k[0] for k in lm.calls] and "session" not in [k[0] for k in lm.calls]
    # the batch path passes no repetition penalty (a no-op in the reference, static_batch.rs:204-206)
    assert all("repetition_penalty" not in k[2] for k in lm.calls if k[0] == "batch")
    state.scheduler.close()


def test_reroll_once_then_error_and_error_mapping():
    state, lm = _state(max_new_tokens=6)
    lm.frames_for = lambda prompt: 6  # every generation runs into max_new_tokens
    c = _client(state)
    r = c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Hi."))
    assert r.status_code == 500 and b"second time" in r.content and state.scheduler.stats["rerolls"] == 1
    n = {"k": 0}

    def flaky(prompt):
        n["k"] += 1
        return 6 if n["k"] == 1 else 4
    lm.frames_for = flaky
    r = c.post("/v1/audio/speech", json=dict(model="x", voice="default", input="Hi."))
    assert r.status_code == 200 and state.scheduler.stats["rerolls"] == 2
    state.scheduler.close()


def test_encode_speaker_returns_npy_and_registers_the_voice():
    state, lm = _state()
    c = _client(state)
    pcm = (np.sin(np.arange(22050 * 2) / 20.0) * 12000).astype("<i2")
    wav = io.BytesIO()
    wav.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 22050, 44100, 2, 16) + b"data" +
              struct.pack("<I", pcm.nbytes) + pcm.tobytes())
    files = {"file": ("ref.wav", wav.getvalue(), "audio/wav")}
    r = c.post("/v1/audio/encoding", files=files, params=dict(id="bob", prompt="this is bob"))
    assert r.status_code == 200 and r.headers["content-type"] == "application/x-npy"
    codes = np.load(io.BytesIO(r.content))
    assert codes.shape == (8, 2 * 44100 // 2048) and codes.dtype == np.uint32  # 2 s resampled 22.05 -> 44.1 kHz
    assert "bob" in c.get("/v1/voices").json()
    r = c.post("/v1/audio/encoding", files=files, params=dict(id="bob", prompt="again"))
    assert r.status_code == 500 and b"ID already exists on server: bob" in r.content
    r = c.post("/v1/audio/encoding", files={"file": ("x.mp3", b"ID3....", "audio/mpeg")})
    assert r.status_code == 500
    state.scheduler.close()


def test_preprocess_text_chunks():
    assert server.preprocess_text("  Hello “world”…  ") == ['Hello "world"...']
    long = ("word " * 100).strip() + "."
    chunks = server.preprocess_text(long + " " + long)
    assert all(len(ch) <= 400 for ch in chunks) and "".join(chunks).replace(" ", "") == (long + long).replace(" ", "")
    with pytest.raises(RuntimeError):
        fprompt.PromptEncoder(Tok(), 8).encode_sequence([], None, None, True)
