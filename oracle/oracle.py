"""TEST INFRASTRUCTURE ONLY: ctypes front-end of the CPU restatement (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product path under fish-speech.rs_amd/ never does).  PARITY UNPINNED against the reference
binary -- see oracle/oracle_lm.h for what the restatement is pinned by instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# Field order of the flat integer/float argument vectors (mirrors BaseModelArgs, dual_ar.rs:57-81)
IARGS = ["dim", "n_layer", "n_fast_layer", "n_head", "n_local_heads", "head_dim", "intermediate_size",
         "num_codebooks", "codebook_size", "vocab_size", "max_seq_len"]
FARGS = ["norm_eps", "rope_base"]
TOKS = ["im_end_id", "pad_id", "semantic_start_id", "semantic_end_id", "has_semantic_end"]

FISH15 = dict(dim=1024, n_layer=24, n_fast_layer=4, n_head=16, n_local_heads=2, head_dim=64,
              intermediate_size=4096, num_codebooks=8, codebook_size=1024, vocab_size=102048, max_seq_len=8192,
              norm_eps=1e-6, rope_base=1e6,
              im_end_id=100011, pad_id=5, semantic_start_id=100012, semantic_end_id=101035, has_semantic_end=1)
# tiny config of SURVEY.md §8c: im_end / semantic layout preserved (im_end == semantic_start - 1)
TINY = dict(dim=128, n_layer=2, n_fast_layer=1, n_head=4, n_local_heads=2, head_dim=32,
            intermediate_size=256, num_codebooks=8, codebook_size=64, vocab_size=512, max_seq_len=256,
            norm_eps=1e-6, rope_base=1e6,
            im_end_id=400, pad_id=5, semantic_start_id=401, semantic_end_id=464, has_semantic_end=1)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h")) or f == "Makefile"]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def usable_cpus():
    """CPUs this process may actually use: min(affinity, cgroup quota).  The GPU box exposes 256 hardware threads but
    caps the container at 16 CPUs; OpenMP's default (256 spinning threads) is ~10^4x slower there."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def lib():
    global _LIB
    if _LIB is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_last_error.restype = C.c_char_p
        L.orc_lm_create.restype = C.c_void_p
        L.orc_lm_fast_embeddings.restype = C.POINTER(C.c_float)
        L.orc_lm_freqs.restype = C.POINTER(C.c_float)
        L.orc_lm_tensor.restype = C.POINTER(C.c_float)
        L.orc_lm_force_kv_diff.restype = C.c_float
        L.orc_reppen_create.restype = C.c_void_p
        L.orc_sampler_create.restype = C.c_void_p
        L.orc_sampler_sample.restype = C.c_uint32
        L.orc_codec_create.restype = C.c_void_p
        L.orc_set_num_threads(usable_cpus())
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _chk(rc):
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())


def quant_rows_fp8(w):
    """per-row absmax e4m3fn quantise + dequantise (the FS_FP8 storage format), f32 [rows, cols]"""
    w = np.ascontiguousarray(w, dtype=np.float32).copy()
    lib().orc_quant_rows_fp8(_p(w, C.c_float), C.c_uint64(w.shape[0]), C.c_uint64(w.shape[1]))
    return w


def e4m3_to_f32(b):
    f = lib().orc_e4m3_to_f32
    f.restype = C.c_float
    return float(f(C.c_uint8(int(b))))


def f32_to_e4m3(x):
    f = lib().orc_f32_to_e4m3
    f.restype = C.c_uint8
    return int(f(C.c_float(float(x))))


def synth(name, n, seed, mean=0.0, std=0.02, bf16=False):
    out = np.empty(n, np.float32)
    lib().orc_synth_fill(_p(out, C.c_float), C.c_uint64(n), name.encode(), C.c_uint64(seed), C.c_float(mean),
                         C.c_double(std), int(bf16))
    return out


class OracleLM:
    def __init__(self, cfg):
        self.cfg = dict(cfg)
        ia = np.array([cfg[k] for k in IARGS], np.int32)
        fa = np.array([cfg[k] for k in FARGS], np.float32)
        tk = np.array([cfg[k] for k in TOKS], np.uint32)
        self.h = C.c_void_p(lib().orc_lm_create(_p(ia, C.c_int), _p(fa, C.c_float), _p(tk, C.c_uint32)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_lm_destroy(self.h)
            self.h = None

    def load_synthetic(self, seed, bf16=False, fp8=False):
        # mode 0 f32 / 1 bf16 checkpoint / 2 fp8 Linear weights (FS_FP8 storage of the product)
        _chk(lib().orc_lm_load_synthetic(self.h, C.c_uint64(seed), 2 if fp8 else int(bf16)))
        return self

    def set_kv_round_bf16(self, on):
        lib().orc_lm_set_kv_round_bf16(self.h, int(on))

    def forward_generate(self, toks, input_pos, full_head=True):
        toks = np.ascontiguousarray(toks, np.uint32)
        if toks.ndim == 2:
            toks = toks[None]
        B, _, L = toks.shape
        logits = np.empty((B, self.cfg["vocab_size"]), np.float32)
        hidden = np.empty((B, self.cfg["dim"]), np.float32)
        _chk(lib().orc_lm_forward_generate(self.h, _p(toks, C.c_uint32), B, L, int(input_pos), _p(logits, C.c_float),
                                           _p(hidden, C.c_float), int(full_head)))
        return logits, hidden

    def forward_generate_fast(self, x, pos):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.cfg["dim"])
        out = np.empty((x.shape[0], self.cfg["codebook_size"]), np.float32)
        _chk(lib().orc_lm_forward_generate_fast(self.h, _p(x, C.c_float), x.shape[0], int(pos), _p(out, C.c_float)))
        return out

    def clear_fast(self):
        lib().orc_lm_clear_fast(self.h)

    def clear_slow(self):
        lib().orc_lm_clear_slow(self.h)

    def clear_slow_until(self, pos):
        lib().orc_lm_clear_slow_until(self.h, int(pos))

    def kv_len(self):
        return lib().orc_lm_kv_len(self.h)

    def force_kv(self, layer, k, v):
        """test hook: the NEXT single-token step of `layer` (1000 + l: fast layer l) uses these K / V rows (Hkv, D) instead of its own;
        force_kv_diff(layer) afterwards = how far its own rows were from them, in bf16 ulps"""
        k = np.ascontiguousarray(k, np.float32).reshape(-1); v = np.ascontiguousarray(v, np.float32).reshape(-1)
        if lib().orc_lm_force_kv(self.h, int(layer), _p(k, C.c_float), _p(v, C.c_float)) != 0:
            raise RuntimeError("orc_lm_force_kv: bad layer")

    def force_kv_diff(self, layer):
        return float(lib().orc_lm_force_kv_diff(self.h, int(layer)))

    def set_kv(self, layer, t0, k, v):
        """test hook: overwrite cached K / V rows [t0, t0 + n) of slow layer `layer` (batch 1; 1000 + l: fast decoder layer l); k, v: f32 (n, Hkv, D)"""
        k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        assert k.shape == v.shape and k.ndim == 3
        if lib().orc_lm_set_kv(self.h, int(layer), int(t0), int(k.shape[0]), _p(k, C.c_float), _p(v, C.c_float)) != 0:
            raise RuntimeError("orc_lm_set_kv: rows outside the cache")

    def fast_embeddings(self):
        n = self.cfg["codebook_size"] * self.cfg["dim"]
        return np.ctypeslib.as_array(lib().orc_lm_fast_embeddings(self.h), (n,)).reshape(self.cfg["codebook_size"], -1)

    def freqs(self):
        half = self.cfg["head_dim"] // 2
        n = self.cfg["max_seq_len"] * half
        c = np.ctypeslib.as_array(lib().orc_lm_freqs(self.h, 0), (n,)).reshape(-1, half)
        s = np.ctypeslib.as_array(lib().orc_lm_freqs(self.h, 1), (n,)).reshape(-1, half)
        return c, s

    def tensor(self, name, shape, layer=0):
        ptr = lib().orc_lm_tensor(self.h, name.encode(), int(layer))
        return np.ctypeslib.as_array(ptr, (int(np.prod(shape)),)).reshape(shape)

    def generate(self, prompt, max_new_tokens, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.0, seed=0,
                 ignore_eos=False, max_frames=-1, collect_hidden=False):
        """generate_blocking[_with_hidden] (single_batch.rs:217-324).  collect_hidden: also returns the slow transformer's hidden state
        of EVERY generator iteration, the terminating <|im_end|> one included (:264-266), f32 (n_iter, dim)."""
        prompt = np.ascontiguousarray(prompt, np.uint32)
        Cb = self.cfg["num_codebooks"]
        assert prompt.shape[0] == Cb + 1
        L = prompt.shape[1]
        cap = max_new_tokens + 8
        out = np.zeros(Cb * cap, np.uint32)
        n, nit = C.c_int(0), C.c_int(0)
        pf, dc = C.c_double(0), C.c_double(0)
        margins = np.zeros(cap, np.float32)
        hid = np.zeros((cap, self.cfg["dim"]), np.float32) if collect_hidden else None
        nh = C.c_int(0)
        _chk(lib().orc_lm_generate(self.h, _p(prompt, C.c_uint32), L, int(max_new_tokens), C.c_double(temp),
                                   C.c_double(top_p), C.c_uint64(top_k), C.c_float(repetition_penalty),
                                   C.c_uint64(seed), int(ignore_eos), int(max_frames), _p(out, C.c_uint32), cap,
                                   C.byref(n), C.byref(pf), C.byref(dc), _p(margins, C.c_float), C.byref(nit),
                                   _p(hid, C.c_float) if collect_hidden else None, cap, C.byref(nh)))
        self.last_prefill_s, self.last_decode_s = pf.value, dc.value
        self.last_margins = margins[: nit.value].copy()  # min top-2 margin of the 9 decisions of each iteration
        codes = out[: Cb * n.value].reshape(Cb, n.value).copy()
        return (codes, hid[: nh.value].copy()) if collect_hidden else codes


def _generate_batch(self, prompts, max_new_tokens, temp=0.0, top_p=1.0, top_k=0, seed=42, ignore_eos=False):
    """generate_static_batch (static_batch.rs:282-390): list of (C+1, L_i) -> list of (C, n_i)."""
    Cb = self.cfg["num_codebooks"]
    ps = [np.ascontiguousarray(p, np.uint32) for p in prompts]
    lens = np.array([p.shape[1] for p in ps], np.int32)
    flat = np.concatenate([p.reshape(-1) for p in ps])
    cap = max_new_tokens + 8
    out = np.zeros((len(ps), Cb, cap), np.uint32)
    nf = np.zeros(len(ps), np.int32)
    margins = np.zeros((cap, len(ps)), np.float32)
    nit = C.c_int(0)
    _chk(lib().orc_lm_generate_batch(self.h, _p(flat, C.c_uint32), _p(lens, C.c_int), len(ps), int(max_new_tokens), C.c_double(temp),
                                     C.c_double(top_p), C.c_uint64(top_k), C.c_uint64(seed), int(ignore_eos), _p(out, C.c_uint32), cap,
                                     _p(nf, C.c_int), _p(margins, C.c_float), cap, C.byref(nit)))
    # [iteration, row]: smallest top-2 logit margin among the row's 9 decisions of that iteration (greedy parity tests use it to tell a
    # kernel bug from a legitimate near-tie flip under reduced-precision storage)
    self.last_batch_margins = margins[: nit.value].copy()
    return [out[i, :, : nf[i]].copy() for i in range(len(ps))]


OracleLM.generate_batch = _generate_batch


class OracleCodec:
    def __init__(self, tiny=False):
        self.h = C.c_void_p(lib().orc_codec_create(int(tiny)))
        self.tiny = tiny

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_codec_destroy(self.h)
            self.h = None

    def load_synthetic(self, seed):
        _chk(lib().orc_codec_load_synthetic(self.h, C.c_uint64(seed)))
        return self

    def set_tensors(self, tensors):
        """override decode-side tensors (name -> array, the checkpoint's names) of a codec sized by load_synthetic"""
        for name, v in tensors.items():
            a = np.ascontiguousarray(v, np.float32)
            _chk(lib().orc_codec_set_tensor(self.h, name.encode(), _p(a, C.c_float), C.c_uint64(a.size)))
        return self

    @property
    def hop(self):
        return lib().orc_codec_hop(self.h)

    def fsq_code(self, idx):
        o = np.empty(4, np.float32)
        lib().orc_codec_fsq_code(self.h, C.c_uint32(idx), _p(o, C.c_float))
        return o

    def decode(self, codes, stage=None, stage_size=0):
        codes = np.ascontiguousarray(codes, np.uint32)
        if codes.ndim == 3:
            assert codes.shape[0] == 1, "quantizer reshape is only correct for b=1 (quantizer.rs:138-143)"
            codes = codes[0]
        T = codes.shape[1]
        pcm = np.empty(self.hop * T, np.float32)
        st = np.empty(stage_size, np.float32) if stage is not None else None
        _chk(lib().orc_codec_decode(self.h, _p(codes, C.c_uint32), T, _p(pcm, C.c_float),
                                    int(stage or 0), _p(st, C.c_float) if st is not None else None))
        return (pcm, st) if stage is not None else pcm


def mel_filterbank(sr=44100, n_fft=2048, n_mels=160):
    out = np.zeros((n_fft // 2 + 1, n_mels), np.float32)
    _chk(lib().orc_codec_mel_filterbank(int(sr), int(n_fft), int(n_mels), _p(out, C.c_float)))
    return out


def _codec_log_mel(self, pcm):
    pcm = np.ascontiguousarray(pcm, np.float32).reshape(-1)
    cap = pcm.size // 512 + 8
    mel = np.zeros((160, cap), np.float32)
    flat = np.zeros(160 * cap, np.float32)
    fr = C.c_int(0)
    _chk(lib().orc_codec_log_mel(self.h, _p(pcm, C.c_float), int(pcm.size), _p(flat, C.c_float), cap, C.byref(fr)))
    return flat[: 160 * fr.value].reshape(160, fr.value).copy()


def _codec_encode_mel(self, mel, stage=None, stage_size=0):
    mel = np.ascontiguousarray(mel, np.float32)
    frames = mel.shape[1]
    codes = np.zeros((8, frames), np.uint32)
    L = C.c_int(0)
    st = np.zeros(stage_size, np.float32) if stage is not None else None
    _chk(lib().orc_codec_encode_mel(self.h, _p(mel, C.c_float), int(frames), _p(codes, C.c_uint32), int(frames), C.byref(L),
                                    int(stage or 0), _p(st, C.c_float) if st is not None else None))
    out = codes.reshape(-1)[: 8 * L.value].reshape(8, L.value).copy()
    return (out, st) if stage is not None else out


def _codec_encode(self, pcm):
    pcm = np.ascontiguousarray(pcm, np.float32).reshape(-1)
    cap = pcm.size // 512 + 8
    codes = np.zeros(8 * cap, np.uint32)
    L = C.c_int(0)
    _chk(lib().orc_codec_encode(self.h, _p(pcm, C.c_float), int(pcm.size), _p(codes, C.c_uint32), cap, C.byref(L)))
    return codes[: 8 * L.value].reshape(8, L.value).copy()


OracleCodec.log_mel = _codec_log_mel
OracleCodec.encode_mel = _codec_encode_mel
OracleCodec.encode = _codec_encode


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
