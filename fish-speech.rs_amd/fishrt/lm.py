"""Host-side mirror of the reference's LM surface over the C ABI.

`DualARTransformer` mirrors fish_speech_core::lm::DualARTransformer (dual_ar.rs:443-713) +
generate_blocking / generate_static_batch (generate/*.rs); `LM` mirrors the PyO3 class
(fish_speech_python/src/lm.rs:23-199) minus tokenisation (the caller supplies token ids: SURVEY.md §2 row 7)."""
import ctypes as C

import numpy as np

from . import _ffi, config

DTYPES = {"f32": 0, "bf16": 1, "fp8": 2}


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


class DualARTransformer:
    def __init__(self, model_args=None, token_cfg=None, device=0, dtype="bf16", max_batch=1):
        if dtype not in DTYPES:
            raise ValueError(f"Unsupported dtype: {dtype}")  # fish_speech_python/src/utils.rs:25-40
        self.cfg = dict(model_args or config.FISH_1_5)
        self.tok = dict(token_cfg or config.FISH_1_5_TOKENS)
        self.dtype = dtype
        ma = _ffi.ModelArgs(**{k: self.cfg[k] for k, _ in _ffi.ModelArgs._fields_})
        tc = _ffi.TokenCfg(**self.tok)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().fs_lm_create(C.byref(ma), C.byref(tc), int(device), DTYPES[dtype], int(max_batch), C.byref(h)))
        self._h = h
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "_h", None) and _ffi is not None and getattr(_ffi, "lib", None):  # (module globals are gone at interpreter exit)
            _ffi.lib().fs_lm_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- weights
    def load_synthetic(self, seed):
        _ffi.check(_ffi.lib().fs_lm_load_synthetic(self._h, C.c_uint64(seed)))
        return self

    def load_safetensors(self, path):
        _ffi.check(_ffi.lib().fs_lm_load_safetensors(self._h, str(path).encode()))
        return self

    # ---- DualARTransformer methods
    def forward_generate(self, inp, input_pos, want_logits=True):
        """dual_ar.rs:574-635.  inp: u32 (B, C+1, L) or (C+1, L).  Returns (logits (B, V), hidden (B, dim))."""
        inp = _u32(inp)
        if inp.ndim == 2:
            inp = inp[None]
        if inp.ndim != 3 or inp.shape[1] != self.cfg["num_codebooks"] + 1:
            raise ValueError("Input tokens must have num_codebooks + 1 codebooks!")  # dual_ar.rs:535-538
        B, _, L = inp.shape
        logits = np.empty((B, self.cfg["vocab_size"]), np.float32) if want_logits else None
        hidden = np.empty((B, self.cfg["dim"]), np.float32)
        _ffi.check(_ffi.lib().fs_lm_forward_generate(
            self._h, inp.ctypes.data_as(C.POINTER(C.c_uint32)), B, L, int(input_pos),
            logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None,
            hidden.ctypes.data_as(C.POINTER(C.c_float))))
        return logits, hidden

    def forward_generate_fast(self, x, input_pos):
        """dual_ar.rs:638-673.  x: f32 (B, dim).  Returns logits (B, codebook_size)."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.cfg["dim"])
        out = np.empty((x.shape[0], self.cfg["codebook_size"]), np.float32)
        _ffi.check(_ffi.lib().fs_lm_forward_generate_fast(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[0],
                                                          int(input_pos), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def fast_embeddings(self, ids):
        ids = _u32(ids).reshape(-1)
        out = np.empty((ids.size, self.cfg["dim"]), np.float32)
        _ffi.check(_ffi.lib().fs_lm_fast_embed(self._h, ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size,
                                               out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def clear_fast_layer_caches(self):
        _ffi.check(_ffi.lib().fs_lm_clear_fast_layer_caches(self._h))

    def clear_slow_layer_caches(self):
        _ffi.check(_ffi.lib().fs_lm_clear_slow_layer_caches(self._h))

    def clear_slow_caches_until(self, pos):
        _ffi.check(_ffi.lib().fs_lm_clear_slow_caches_until(self._h, int(pos)))

    def curr_kv_size(self):
        n = _ffi.lib().fs_lm_curr_kv_size(self._h)
        if n < 0:
            raise RuntimeError(_ffi.lib().fs_last_error().decode())
        return n

    # ---- generation drivers
    def generate_blocking(self, prompt, max_new_tokens, temp=0.7, top_p=0.9, top_k=50, repetition_penalty=1.2, seed=0,
                          ignore_eos=False, on_frame=None, persistent=True, time_kernels=False):
        """generate/single_batch.rs:308-324.  prompt u32 (C+1, L) -> codes u32 (C, n_frames).
        persistent=False sets FS_GEN_NO_PERSIST (per-node graph path; see include/fishrt.h); time_kernels=True sets FS_GEN_TIME_KERNELS
        (measurement mode: HIP events around each persistent kernel, last_stats()["slow_kernel_us" / "fast_kernel_us"])."""
        prompt = _u32(prompt)
        Cb = self.cfg["num_codebooks"]
        if prompt.ndim != 2 or prompt.shape[0] != Cb + 1:
            raise ValueError("Input tokens must have num_codebooks + 1 codebooks!")
        L = prompt.shape[1]
        cap = max(1, max_new_tokens - L + 2) + 1
        out = np.zeros((Cb, cap), np.uint32)
        n = C.c_size_t(0)
        s = _ffi.Sampling(float(temp), float(top_p), int(top_k), float(repetition_penalty))
        cb = _ffi.FRAME_CB(lambda user, idx, codes: int(bool(on_frame(idx, [codes[i] for i in range(Cb)])))) if on_frame \
            else C.cast(None, _ffi.FRAME_CB)
        _ffi.check(_ffi.lib().fs_lm_generate(self._h, prompt.ctypes.data_as(C.POINTER(C.c_uint32)), L, int(max_new_tokens),
                                             C.byref(s), C.c_uint64(seed), (1 if ignore_eos else 0) | (0 if persistent else 2) | (4 if time_kernels else 0),
                                             out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(cap), C.byref(n), cb, None))
        return out[:, : n.value].copy()

    def generate_blocking_with_hidden(self, prompt, max_new_tokens, collect_hidden_states=True, temp=0.7, top_p=0.9, top_k=50,
                                      repetition_penalty=1.2, seed=0, ignore_eos=False, persistent=True):
        """generate/single_batch.rs:217-306 -> (codes u32 (C, n_frames), hidden f32 (n_iterations, 1, dim) or None)."""
        prompt = _u32(prompt)
        Cb, D = self.cfg["num_codebooks"], self.cfg["dim"]
        if prompt.ndim != 2 or prompt.shape[0] != Cb + 1:
            raise ValueError("Input tokens must have num_codebooks + 1 codebooks!")
        L = prompt.shape[1]
        cap = max(1, max_new_tokens - L + 2) + 1
        out = np.zeros((Cb, cap), np.uint32)
        hid = np.zeros((cap, D), np.float32) if collect_hidden_states else None
        n, nh = C.c_size_t(0), C.c_size_t(0)
        s = _ffi.Sampling(float(temp), float(top_p), int(top_k), float(repetition_penalty))
        _ffi.check(_ffi.lib().fs_lm_generate_with_hidden(
            self._h, prompt.ctypes.data_as(C.POINTER(C.c_uint32)), L, int(max_new_tokens), C.byref(s), C.c_uint64(seed),
            (1 if ignore_eos else 0) | (0 if persistent else 2), out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(cap), C.byref(n),
            hid.ctypes.data_as(C.POINTER(C.c_float)) if collect_hidden_states else None, C.c_size_t(cap), C.byref(nh),
            C.cast(None, _ffi.FRAME_CB), None))
        return out[:, : n.value].copy(), (hid[: nh.value].reshape(nh.value, 1, D).copy() if collect_hidden_states else None)

    def generate_static_batch(self, prompts, max_new_tokens, temp=0.7, top_p=0.9, top_k=50, repetition_penalty=1.2,
                              seed=42, ignore_eos=False, audio_only=True, return_is_audio=False):
        """generate/static_batch.rs:282-390.  prompts: list of u32 (C+1, L_i) -> list of (C, n_i); return_is_audio=True: the reference's full
        return value (codes, is_audio) with is_audio a list of bool arrays (n_i,) -- through fs_lm_generate_static_batch."""
        if return_is_audio or not audio_only:
            return self._generate_static_batch_full(prompts, max_new_tokens, temp, top_p, top_k, repetition_penalty, seed, ignore_eos, audio_only)
        Cb = self.cfg["num_codebooks"]
        ps = [_u32(p) for p in prompts]
        lens = np.array([p.shape[1] for p in ps], np.int32)
        flat = np.concatenate([p.reshape(-1) for p in ps])
        cap = max(1, max_new_tokens - int(lens.max()) + 2) + 1
        out = np.zeros((len(ps), Cb, cap), np.uint32)
        nf = (C.c_size_t * len(ps))()
        s = _ffi.Sampling(float(temp), float(top_p), int(top_k), float(repetition_penalty))
        _ffi.check(_ffi.lib().fs_lm_generate_batch(self._h, flat.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                   lens.ctypes.data_as(C.POINTER(C.c_int)), len(ps), int(max_new_tokens),
                                                   C.byref(s), C.c_uint64(seed), 1 if ignore_eos else 0,
                                                   out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(cap), nf))
        return [out[i, :, : nf[i]].copy() for i in range(len(ps))]

    def _generate_static_batch_full(self, prompts, max_new_tokens, temp, top_p, top_k, repetition_penalty, seed, ignore_eos, audio_only):
        Cb = self.cfg["num_codebooks"]
        ps = [_u32(p) for p in prompts]
        lens = np.array([p.shape[1] for p in ps], np.int32)
        flat = np.concatenate([p.reshape(-1) for p in ps])
        cap = max(1, max_new_tokens - int(lens.max()) + 2) + 1
        out = np.zeros((len(ps), Cb, cap), np.uint32)
        isa = np.zeros((len(ps), cap), np.uint8)
        nf = (C.c_size_t * len(ps))()
        s = _ffi.Sampling(float(temp), float(top_p), int(top_k), float(repetition_penalty))
        _ffi.check(_ffi.lib().fs_lm_generate_static_batch(self._h, flat.ctypes.data_as(C.POINTER(C.c_uint32)), lens.ctypes.data_as(C.POINTER(C.c_int)),
                                                          len(ps), int(max_new_tokens), 1 if audio_only else 0, C.byref(s), C.c_uint64(seed),
                                                          1 if ignore_eos else 0, out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(cap), nf,
                                                          isa.ctypes.data_as(C.POINTER(C.c_uint8))))
        return [out[i, :, : nf[i]].copy() for i in range(len(ps))], [isa[i, : nf[i]].astype(bool) for i in range(len(ps))]

    def generate_multi(self, prompts, max_new_tokens, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, seeds=None,
                       ignore_eos=False, persistent=True):
        """R concurrent generate_blocking requests on this handle (fishrt.h: fs_lm_generate_multi): prompts = list of u32 (C+1, L_i);
        max_new_tokens / temp / top_p / top_k / repetition_penalty: one value for all requests or one per request -> list of (C, n_i)."""
        Cb = self.cfg["num_codebooks"]
        ps = [_u32(p) for p in prompts]
        n = len(ps)
        per = lambda v: list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * n
        mnt, temps, tps, tks, rps = per(max_new_tokens), per(temp), per(top_p), per(top_k), per(repetition_penalty)
        lens = np.array([p.shape[1] for p in ps], np.int32)
        flat = np.concatenate([p.reshape(-1) for p in ps])
        mn = np.array(mnt, np.int32)
        cap = max(max(1, int(m) - int(l) + 2) for m, l in zip(mnt, lens)) + 1
        out = np.zeros((n, Cb, cap), np.uint32)
        nf = (C.c_size_t * n)()
        ss = (_ffi.Sampling * n)(*[_ffi.Sampling(float(temps[i]), float(tps[i]), int(tks[i]), float(rps[i])) for i in range(n)])
        sd = np.array(list(seeds) if seeds is not None else [0] * n, np.uint64)
        _ffi.check(_ffi.lib().fs_lm_generate_multi(self._h, flat.ctypes.data_as(C.POINTER(C.c_uint32)), lens.ctypes.data_as(C.POINTER(C.c_int)), n,
                                                   mn.ctypes.data_as(C.POINTER(C.c_int)), ss, sd.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   (1 if ignore_eos else 0) | (0 if persistent else 2),
                                                   out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(cap), nf))
        return [out[i, :, : nf[i]].copy() for i in range(n)]

    def rows_supported(self, n, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2):
        """fishrt.h fs_lm_rows_supported: would generate_multi serve n requests with these settings on the request-row kernels?"""
        per = lambda v: list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * n
        temps, tps, tks, rps = per(temp), per(top_p), per(top_k), per(repetition_penalty)
        ss = (_ffi.Sampling * n)(*[_ffi.Sampling(float(temps[i]), float(tps[i]), int(tks[i]), float(rps[i])) for i in range(n)])
        ok = C.c_int(0)
        _ffi.check(_ffi.lib().fs_lm_rows_supported(self._h, int(n), ss, C.byref(ok)))
        return bool(ok.value)

    def weights_arena(self):
        """(device pointer, bytes) of the handle's weight arena (fishrt.h: fs_lm_weights_arena) -- for fanout.broadcast_weights"""
        ptr, n = C.c_void_p(), C.c_size_t(0)
        _ffi.check(_ffi.lib().fs_lm_weights_arena(self._h, C.byref(ptr), C.byref(n)))
        return int(ptr.value), int(n.value)

    def adopt_weights(self):
        _ffi.check(_ffi.lib().fs_lm_weights_adopt(self._h))
        return self

    def session(self, temp=0.7, top_p=0.9, top_k=50, seed=42, ignore_eos=False, rows=False, repetition_penalty=1.2):
        """continuous batching over this handle's max_batch slots (fishrt.h: fs_lm_session_*): `with lm.session(...) as s:`"""
        return Session(self, temp, top_p, top_k, seed, ignore_eos, rows, repetition_penalty)

    def last_stats(self):
        st = _ffi.GenStats()
        _ffi.check(_ffi.lib().fs_lm_last_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _ffi.GenStats._fields_}

    def selftest(self, what="persist"):
        """fs_lm_selftest: the persistent decode kernels of this build against the per-node kernels on the loaded weights (raises on a mismatch)"""
        _ffi.check(_ffi.lib().fs_lm_selftest(self._h, what.encode()))

    def debug_capture(self, n_frames):
        """test hook: record what the 9 decisions of each of the first n_frames iterations saw and picked (persistent path only)"""
        _ffi.check(_ffi.lib().fs_lm_debug_capture(self._h, int(n_frames)))

    def debug_read(self, n_frames):
        out = np.zeros((int(n_frames), 9, 2048), np.float32)
        _ffi.check(_ffi.lib().fs_lm_debug_read(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), int(n_frames)))
        return out

    def debug_read_kv(self, layer, t0, n, slot=0):
        """test hook: cached K / V rows [t0, t0 + n) of one slow layer as f32 (n, n_local_heads, head_dim) each"""
        sh = (int(n), self.cfg["n_local_heads"], self.cfg["head_dim"])
        k, v = np.zeros(sh, np.float32), np.zeros(sh, np.float32)
        _ffi.check(_ffi.lib().fs_lm_debug_read_kv(self._h, int(slot), int(layer), int(t0), int(n), k.ctypes.data_as(C.POINTER(C.c_float)),
                                                  v.ctypes.data_as(C.POINTER(C.c_float))))
        return k, v

    def debug_read_row(self, row, n_frames):
        """the capture of request `row` of the last generate_multi call"""
        out = np.zeros((int(n_frames), 9, 2048), np.float32)
        _ffi.check(_ffi.lib().fs_lm_debug_read_row(self._h, int(row), out.ctypes.data_as(C.POINTER(C.c_float)), int(n_frames)))
        return out

    def stream(self):
        return _ffi.lib().fs_lm_stream(self._h)

    def bench_kernel(self, kind, kv_len=495, reps=50):
        """us per launch of one batch-1 decode kernel (0 qkv, 1 attention, 2 wo, 3 ffn_up, 4 ffn_down) as a graph node (measurement hook)."""
        us = C.c_float(0)
        _ffi.check(_ffi.lib().fs_lm_bench_kernel(self._h, int(kind), int(kv_len), int(reps), C.byref(us)))
        return float(us.value)


class Session:
    """Request slots over the static-batch step (no reference counterpart).  add() -> slot or None when full;
    step(n) runs up to n frames for all live slots and returns how many are still generating; poll(slot) -> (codes (C, n), done);
    release(slot) frees the slot.  A slot generates what a one-prompt generate_static_batch would (no repetition penalty)."""

    def __init__(self, lm, temp, top_p, top_k, seed, ignore_eos, rows=False, repetition_penalty=1.2):
        """rows=True: FS_SESSION_ROWS -- the slots run on the request-row persistent kernels with batch-1 semantics (repetition penalty, own
        sampler stream per slot); max_batch <= 8"""
        self.lm, self._open = lm, False
        s = _ffi.Sampling(float(temp), float(top_p), int(top_k), float(repetition_penalty) if rows else 1.0)
        _ffi.check(_ffi.lib().fs_lm_session_begin(lm._h, C.byref(s), C.c_uint64(seed), (1 if ignore_eos else 0) | (8 if rows else 0)))
        self._open = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._open:
            self._open = False
            _ffi.check(_ffi.lib().fs_lm_session_end(self.lm._h))

    def add(self, prompt, max_new_tokens):
        p = _u32(prompt)
        if p.ndim != 2 or p.shape[0] != self.lm.cfg["num_codebooks"] + 1 or p.shape[1] < 1:  # the C side reads (C + 1) * L words
            raise ValueError(f"prompt must be u32 [{self.lm.cfg['num_codebooks'] + 1}, L >= 1], got {p.shape}")
        slot = C.c_int(-1)
        _ffi.check(_ffi.lib().fs_lm_session_add(self.lm._h, p.ctypes.data_as(C.POINTER(C.c_uint32)), int(p.shape[1]), int(max_new_tokens),
                                                C.byref(slot)))
        return None if slot.value < 0 else int(slot.value)

    def step(self, n_frames=8):
        act = C.c_int(0)
        _ffi.check(_ffi.lib().fs_lm_session_step(self.lm._h, int(n_frames), C.byref(act)))
        return int(act.value)

    def poll(self, slot, codes=True):
        n, done = C.c_size_t(0), C.c_int(0)
        _ffi.check(_ffi.lib().fs_lm_session_poll(self.lm._h, int(slot), None, C.c_size_t(0), C.byref(n), C.byref(done)))
        if not codes:
            return int(n.value), bool(done.value)
        cap = max(1, int(n.value))
        out = np.zeros((self.lm.cfg["num_codebooks"], cap), np.uint32)
        _ffi.check(_ffi.lib().fs_lm_session_poll(self.lm._h, int(slot), out.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(cap), C.byref(n),
                                                 C.byref(done)))
        return out[:, : n.value].copy(), bool(done.value)

    def release(self, slot):
        _ffi.check(_ffi.lib().fs_lm_session_release(self.lm._h, int(slot)))


class LM:
    """fish_speech_python `LM` (lm.rs:23-199) with token-id inputs: __call__(list of prompts (C+1, L_i) for successive
    text chunks, speaker_prompt) -> u32 (1, C, sum T).  Conditioning tokens stay cached across chunks exactly as
    lm.rs:94-135: clear cache, generate per chunk, clear_slow_caches_until(n_conditioning_tokens)."""

    def __init__(self, model_args=None, token_cfg=None, device=0, dtype="bf16"):
        self.model = DualARTransformer(model_args, token_cfg, device, dtype)

    def __call__(self, chunk_prompts, n_conditioning_tokens=0, temp=0.7, top_p=0.9, top_k=50, repetition_penalty=1.2,
                 max_new_tokens=1024, seed=0):
        self.model.clear_slow_layer_caches()
        outs = []
        for i, p in enumerate(chunk_prompts):
            outs.append(self.model.generate_blocking(p, max_new_tokens, temp, top_p, top_k, repetition_penalty, seed + i))
            self.model.clear_slow_caches_until(n_conditioning_tokens)
        self.model.clear_slow_layer_caches()
        return np.concatenate(outs, axis=1)[None]
