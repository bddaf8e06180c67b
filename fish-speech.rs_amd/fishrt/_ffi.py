"""ctypes binding of libfishrt.so (include/fishrt.h).  The library is REQUIRED: there is no CPU / PyTorch fallback --
importing works without a GPU (so the symbol table can be checked), but creating any handle without an MI355X fails."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libfishrt.so")


class ModelArgs(C.Structure):  # fs_model_args  <->  BaseModelArgs (dual_ar.rs:57-81)
    _fields_ = [(n, C.c_int32) for n in ("dim", "n_layer", "n_fast_layer", "n_head", "n_local_heads", "head_dim",
                                         "intermediate_size", "num_codebooks", "codebook_size", "vocab_size",
                                         "max_seq_len")] + [("norm_eps", C.c_float), ("rope_base", C.c_float),
                                                            ("tie_word_embeddings", C.c_int32)]


class TokenCfg(C.Structure):  # fs_token_cfg  <->  TokenConfig (dual_ar.rs:17-23)
    _fields_ = [("im_end_id", C.c_uint32), ("pad_id", C.c_uint32), ("semantic_start_id", C.c_uint32),
                ("semantic_end_id", C.c_uint32), ("has_semantic_end", C.c_int32)]


class Sampling(C.Structure):  # fs_sampling  <->  SamplingArgs (sampling/mod.rs:29-34)
    _fields_ = [("temp", C.c_double), ("top_p", C.c_double), ("top_k", C.c_uint64), ("repetition_penalty", C.c_float)]


class GenStats(C.Structure):
    _fields_ = [("prefill_ms", C.c_double), ("decode_ms", C.c_double), ("frames", C.c_uint64),
                ("prompt_tokens", C.c_uint64), ("graph_launches", C.c_uint64), ("kernels_per_frame", C.c_uint64),
                ("slow_kernel_us", C.c_double), ("fast_kernel_us", C.c_double)]


FRAME_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32))

# every symbol include/fishrt.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = ["fs_last_error", "fs_version", "fs_device_count", "fs_lm_create", "fs_lm_destroy", "fs_lm_load_safetensors",
           "fs_lm_load_synthetic", "fs_lm_forward_generate", "fs_lm_forward_generate_fast", "fs_lm_fast_embed",
           "fs_lm_clear_fast_layer_caches", "fs_lm_clear_slow_layer_caches", "fs_lm_clear_slow_caches_until",
           "fs_lm_curr_kv_size", "fs_lm_generate", "fs_lm_generate_with_hidden", "fs_lm_generate_batch", "fs_lm_generate_static_batch", "fs_lm_generate_multi", "fs_lm_rows_supported", "fs_lm_debug_read_row", "fs_lm_debug_read_kv", "fs_lm_last_stats", "fs_lm_stream", "fs_lm_bench_kernel",
           "fs_lm_weights_arena", "fs_lm_weights_adopt", "fs_comm_unique_id", "fs_comm_create", "fs_comm_destroy", "fs_comm_rank", "fs_comm_world", "fs_comm_barrier",
           "fs_comm_all_reduce_f64", "fs_comm_broadcast_weights", "fs_comm_broadcast_prompt_dims", "fs_comm_broadcast_prompts", "fs_comm_all_gather_codes", "fs_lm_session_begin", "fs_lm_session_add", "fs_lm_session_step", "fs_lm_session_poll", "fs_lm_session_release", "fs_lm_session_end",
           "fs_codec_create", "fs_codec_destroy", "fs_codec_load_safetensors", "fs_codec_load_synthetic",
           "fs_codec_decode", "fs_codec_encode", "fs_codec_encode_batch", "fs_codec_sample_rate", "fs_codec_set_precision", "fs_codec_precision", "fs_codec_set_range_check", "fs_codec_range_stats", "fs_codec_stream_begin", "fs_codec_stream_decode", "fs_codec_stream_end", "fs_fp8_quantize_rows", "fs_fp8_decode_table", "fs_selftest", "fs_lm_selftest", "fs_selftest_sample_rows", "fs_lm_debug_capture", "fs_lm_debug_read"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with fish-speech.rs_amd/build.sh (hipcc, gfx950). "
                               "fishrt has no fallback path.")
        L = C.CDLL(LIB_PATH)
        L.fs_last_error.restype = C.c_char_p
        L.fs_version.restype = C.c_char_p
        L.fs_lm_stream.restype = C.c_void_p
        L.fs_lm_destroy.restype = None
        L.fs_codec_destroy.restype = None
        L.fs_comm_destroy.restype = None
        L.fs_comm_destroy.argtypes = [C.c_void_p]
        L.fs_lm_destroy.argtypes = [C.c_void_p]
        L.fs_codec_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().fs_last_error().decode())
