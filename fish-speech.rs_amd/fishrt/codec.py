"""fish_speech_python `FireflyCodec` (codec.rs:18-115) over the C ABI: decode(u32[b,8,T]) -> f32[b,1,2048*T], encode(f32[1,1,n]) -> u32[1,8,L]."""
import ctypes as C

import numpy as np

from . import _ffi


class FireflyCodec:
    PRECISIONS = {"f32": 0, "bf16x3": 1, "f16": 2}

    def __init__(self, device=0, channel_div=1, precision="f16"):
        """precision: arithmetic of the decode path's convs -- "f16" (default: single f16 operands on the matrix cores, f32 accumulation
        and f32 residual stream; PCM 1.6e-5 RMS from the f32 reference at signal rms 0.031, bound 1e-4), "bf16x3" (split-bf16 matrix
        products, 3e-7 RMS) or "f32" (exact f32 products); the encoder always runs exact f32 (fishrt.h: fs_codec_set_precision)"""
        h = C.c_void_p()
        _ffi.check(_ffi.lib().fs_codec_create(int(device), int(channel_div), C.byref(h)))
        self._h = h
        _ffi.check(_ffi.lib().fs_codec_set_precision(self._h, self.PRECISIONS[precision]))
        self.precision = precision

    def close(self):
        if getattr(self, "_h", None) and _ffi is not None and getattr(_ffi, "lib", None):  # (module globals are gone at interpreter exit)
            _ffi.lib().fs_codec_destroy(self._h)
            self._h = None

    __del__ = close

    def load_synthetic(self, seed):
        _ffi.check(_ffi.lib().fs_codec_load_synthetic(self._h, C.c_uint64(seed)))
        return self

    def load_safetensors(self, path):
        _ffi.check(_ffi.lib().fs_codec_load_safetensors(self._h, str(path).encode()))
        return self

    @property
    def sample_rate(self):
        return _ffi.lib().fs_codec_sample_rate(self._h)

    def decode(self, codes):
        if not isinstance(codes, np.ndarray) or not codes.flags["C_CONTIGUOUS"]:
            raise ValueError("Input array must be contiguous")  # codec.rs:97-101
        if codes.ndim != 3 or codes.shape[1] != 8:
            raise ValueError("codes must have shape (b, 8, T)")
        codes = codes.astype(np.uint32, copy=False)
        b, _, T = codes.shape
        pcm = np.empty((b, 1, 2048 * T), np.float32)
        _ffi.check(_ffi.lib().fs_codec_decode(self._h, codes.ctypes.data_as(C.POINTER(C.c_uint32)), b, T,
                                              pcm.ctypes.data_as(C.POINTER(C.c_float))))
        return pcm

    # ---- f16 range guard (fishrt.h fs_codec_set_range_check / fs_codec_range_stats): diagnostic for validating a new checkpoint
    def set_range_check(self, on=True):
        _ffi.check(_ffi.lib().fs_codec_set_range_check(self._h, 1 if on else 0))
        return self

    def range_stats(self):
        out, rms = (C.c_uint64 * 5)(), C.c_double(0.0)
        _ffi.check(_ffi.lib().fs_codec_range_stats(self._h, out, C.byref(rms)))
        return dict(act_saturated=int(out[0]), act_flushed=int(out[1]), weights_saturated=int(out[2]), weights_flushed=int(out[3]),
                    fallbacks=int(out[4]), last_pcm_rms_diff=float(rms.value))

    # ---- stateful streaming (fishrt.h fs_codec_stream_*): chunks of ONE code sequence, left context kept on the device
    STREAM_MIN_FRAMES = 16

    def stream_begin(self):
        _ffi.check(_ffi.lib().fs_codec_stream_begin(self._h))
        return self

    def stream_decode(self, codes):
        """codes u32 (8, T) or (1, 8, T), T >= 16: the next chunk of the stream -> f32 (2048 T,) PCM.  Concatenated over a stream the result
        is bit-identical to decode() of the whole sequence; no frame is decoded twice."""
        codes = np.ascontiguousarray(codes, np.uint32)
        if codes.ndim == 3 and codes.shape[0] == 1:
            codes = codes[0]
        if codes.ndim != 2 or codes.shape[0] != 8:
            raise ValueError("a streamed chunk must have shape (8, T)")
        T = codes.shape[1]
        pcm = np.empty(2048 * T, np.float32)
        _ffi.check(_ffi.lib().fs_codec_stream_decode(self._h, codes.ctypes.data_as(C.POINTER(C.c_uint32)), T, pcm.ctypes.data_as(C.POINTER(C.c_float))))
        return pcm

    def stream_end(self):
        _ffi.check(_ffi.lib().fs_codec_stream_end(self._h))

    def encode(self, pcm_data, lengths=None):
        """codec.rs:73-94 / firefly.rs:36-39: f32 (b, 1, n) mono 44.1 kHz PCM -> u32 (b, 8, L).  Every clip is encoded on its own
        (fishrt.h fs_codec_encode_batch; the reference's front-end would glue a batch into one signal, spectrogram.rs:33).  lengths: optional
        per-clip sample counts <= n (ragged clips, zero-padded rows) -> list of (8, L_i) instead of one array."""
        if not isinstance(pcm_data, np.ndarray) or not pcm_data.flags["C_CONTIGUOUS"]:
            raise ValueError("Data must be a contiguous array")
        if pcm_data.ndim != 3 or pcm_data.shape[1] != 1:
            raise ValueError("pcm_data must be a 3-D array (b, 1, n)")
        b, _, n_max = pcm_data.shape
        pcm = np.ascontiguousarray(pcm_data.astype(np.float32, copy=False).reshape(b, n_max))
        ns = np.array([n_max] * b if lengths is None else [int(v) for v in lengths], np.int32)
        if ns.shape != (b,) or (ns <= 0).any() or (ns > n_max).any():
            raise ValueError("lengths must hold one sample count in (0, n] per clip")
        cap = n_max // 2048 + 4
        codes = np.zeros((b, 8, cap), np.uint32)
        nf = (C.c_size_t * b)()
        _ffi.check(_ffi.lib().fs_codec_encode_batch(self._h, pcm.ctypes.data_as(C.POINTER(C.c_float)), int(b), C.c_size_t(n_max),
                                                    ns.ctypes.data_as(C.POINTER(C.c_int)), codes.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    C.c_size_t(cap), nf))
        if lengths is not None:
            return [codes[i, :, : nf[i]].copy() for i in range(b)]
        assert len(set(nf[i] for i in range(b))) == 1
        return codes[:, :, : nf[0]].copy()

