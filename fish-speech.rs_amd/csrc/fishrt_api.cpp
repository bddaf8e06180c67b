// extern "C" surface of libfishrt.so (include/fishrt.h).  Thin: argument checks, exception -> status + message.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/fishrt.h"
#include "codec_engine.h"
#include "fs_comm.h"
#include "fs_common.h"
#include "lm_engine.h"
#include "lm_kernels.h"
#include "lm_persist.h"

#include <cmath>
#include <cstring>
#include <vector>

static thread_local std::string g_err;

struct fs_lm { fs::LMBase* impl; };
struct fs_codec { fs::CodecBase* impl; };
struct fs_comm { fs::Comm* impl; };

#define FS_TRY(body)                                                          \
    try { body; return FS_OK; }                                               \
    catch (const std::exception& e) { g_err = e.what(); return FS_ERR; }      \
    catch (...) { g_err = "unknown error"; return FS_ERR; }
#define FS_ARG(cond, msg) if (!(cond)) { g_err = msg; return FS_ERR; }

extern "C" {

const char* fs_last_error(void) { return g_err.c_str(); }
const char* fs_version(void) { return "fishrt 0.4.0 (gfx950)"; }
int fs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fs_fp8_quantize_rows(int device_id, const float* w, int64_t rows, int64_t cols, uint8_t* q_out, float* scales_out) {
    FS_ARG(w && q_out && scales_out, "null argument");
    FS_TRY(fs::fp8_quantize_rows(device_id, w, rows, cols, q_out, scales_out))
}
int fs_fp8_decode_table(int device_id, float* out) { FS_ARG(out, "null argument"); FS_TRY(fs::fp8_decode_table(device_id, out)) }

static void selftest_pf_reduce(int device) {
    FS_HIP(hipSetDevice(device));
    std::vector<float> in(64 * 32), out(96, 0.f);
    uint32_t st = 12345u;
    for (auto& v : in) { st = st * 1664525u + 1013904223u; v = (float)((st >> 8) & 0xFFFF) / 65536.f - 0.5f; }
    float *din = nullptr, *dout = nullptr;
    FS_HIP(hipMalloc(&din, in.size() * 4));
    FS_HIP(hipMalloc(&dout, out.size() * 4));
    FS_HIP(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    FS_HIP(hipMemset(dout, 0, out.size() * 4));
    fs::launch_pf_reduce_selftest(din, dout, nullptr);
    FS_HIP(hipDeviceSynchronize());
    FS_HIP(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(din); (void)hipFree(dout);
    auto host = [&](int i) { double s = 0; for (int l = 0; l < 64; ++l) s += in[l * 32 + i]; return (float)s; };
    for (int i = 0; i < 32; ++i) if (std::fabs(out[i] - host(i)) > 1e-4f) throw fs::Error("pf_reduce<32> value " + std::to_string(i) + " wrong");
    for (int i = 0; i < 16; ++i) if (std::fabs(out[32 + i] - host(i)) > 1e-4f) throw fs::Error("pf_reduce<16> value " + std::to_string(i) + " wrong");
    for (int i = 0; i < 8; ++i) if (std::fabs(out[48 + i] - host(i)) > 1e-4f) throw fs::Error("pf_reduce<8> value " + std::to_string(i) + " wrong");
    for (int i = 0; i < 4; ++i) if (std::fabs(out[64 + i] - host(i)) > 1e-4f) throw fs::Error("pf_reduce<4> value " + std::to_string(i) + " wrong");
    if (std::fabs(out[68] - host(0)) > 1e-4f) throw fs::Error("pf_wave_sum wrong");
}
int fs_selftest(int device_id, const char* what) {
    FS_ARG(what, "null argument");
    FS_TRY({
        if (std::strcmp(what, "pf_reduce") == 0) selftest_pf_reduce(device_id);
        else throw fs::Error(std::string("unknown self-test: ") + what);
    })
}

int fs_selftest_sample_rows(int device_id, const float* logits, int B, int n, const fs_sampling* s, uint64_t seed, int call_index, uint32_t* out) {
    FS_ARG(logits && s && out, "null argument");
    FS_TRY(fs::debug_sample_rows(device_id, logits, B, n, s->temp, s->top_p, s->top_k, seed, call_index, out))
}

// ---- replica fan-out over RCCL (fs_comm.h)
int fs_comm_unique_id(uint8_t id_out[FS_COMM_ID_BYTES]) { FS_ARG(id_out, "null argument"); FS_TRY(fs::Comm::unique_id(id_out)) }
int fs_comm_create(const uint8_t id[FS_COMM_ID_BYTES], int rank, int world, int device_id, fs_comm_t** out) {
    FS_ARG(id && out, "null argument");
    FS_TRY({ *out = nullptr; fs::Comm* c = new fs::Comm(id, rank, world, device_id); *out = new fs_comm{c}; })
}
void fs_comm_destroy(fs_comm_t* c) {
    if (!c) return;
    delete c->impl;
    delete c;
}
int fs_comm_rank(fs_comm_t* c) { if (!c) { g_err = "null argument"; return -1; } return c->impl->rank(); }
int fs_comm_world(fs_comm_t* c) { if (!c) { g_err = "null argument"; return -1; } return c->impl->world(); }
int fs_comm_barrier(fs_comm_t* c) { FS_ARG(c, "null argument"); FS_TRY(c->impl->barrier()) }
int fs_comm_all_reduce_f64(fs_comm_t* c, double* vals, int n, int op) { FS_ARG(c && vals, "null argument"); FS_TRY(c->impl->all_reduce_f64(vals, n, op)) }
int fs_comm_broadcast_weights(fs_comm_t* c, fs_lm_t* lm, int src, size_t* bytes_moved) {
    FS_ARG(c && lm, "null argument");
    FS_TRY({ const size_t n = c->impl->broadcast_weights(lm->impl, src); if (bytes_moved) *bytes_moved = n; })
}
int fs_comm_broadcast_prompt_dims(fs_comm_t* c, int64_t dims[3], int src) {
    FS_ARG(c && dims, "null argument");
    FS_TRY(c->impl->broadcast_host(dims, 3 * sizeof(int64_t), src))
}
int fs_comm_broadcast_prompts(fs_comm_t* c, uint32_t* packed, int32_t* lens, const int64_t dims[3], int src) {
    FS_ARG(c && packed && lens && dims, "null argument");
    FS_ARG(dims[0] >= 0 && dims[1] >= 1 && dims[2] >= 0 && dims[0] <= (1 << 20) && dims[1] <= 64 && dims[2] <= (1 << 20), "bad prompt batch shape");
    FS_TRY({
        c->impl->broadcast_host(packed, (size_t)dims[0] * (size_t)dims[1] * (size_t)dims[2] * sizeof(uint32_t), src);
        c->impl->broadcast_host(lens, (size_t)dims[0] * sizeof(int32_t), src);
    })
}
int fs_comm_all_gather_codes(fs_comm_t* c, const uint32_t* codes, const int32_t* n_frames, int B, int C, int N, uint32_t* codes_all, int32_t* n_frames_all) {
    FS_ARG(c && codes && n_frames && codes_all && n_frames_all, "null argument");
    FS_ARG(B >= 1 && C >= 1 && N >= 0, "bad code array shape");
    FS_TRY({
        c->impl->all_gather_host(codes, codes_all, (size_t)B * (size_t)C * (size_t)N * sizeof(uint32_t));
        c->impl->all_gather_host(n_frames, n_frames_all, (size_t)B * sizeof(int32_t));
    })
}

int fs_lm_selftest(fs_lm_t* lm, const char* what) { FS_ARG(lm && what, "null argument"); FS_TRY(lm->impl->selftest(what)) }
int fs_lm_debug_capture(fs_lm_t* lm, int n_frames) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->debug_capture(n_frames)) }
int fs_lm_debug_read(fs_lm_t* lm, float* out, int n_frames) { FS_ARG(lm && out, "null argument"); FS_TRY(lm->impl->debug_read(out, n_frames)) }

int fs_lm_create(const fs_model_args* args, const fs_token_cfg* tok, int device_id, fs_dtype dtype, int max_batch, fs_lm_t** out) {
    FS_ARG(args && tok && out, "null argument");
    FS_TRY({ *out = nullptr; fs::LMBase* p = fs::make_lm(*args, *tok, device_id, dtype, max_batch); *out = new fs_lm{p}; })
}
void fs_lm_destroy(fs_lm_t* lm) {
    if (!lm) return;
    delete lm->impl;
    delete lm;
}
int fs_lm_load_safetensors(fs_lm_t* lm, const char* path) { FS_ARG(lm && path, "null argument"); FS_TRY(lm->impl->load_safetensors(path)) }
int fs_lm_load_synthetic(fs_lm_t* lm, uint64_t seed) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->load_synthetic(seed)) }
int fs_lm_forward_generate(fs_lm_t* lm, const uint32_t* toks, int B, int L, int input_pos, float* logits_out, float* hidden_out) {
    FS_ARG(lm && toks, "null argument");
    FS_ARG(input_pos >= 0, "negative input_pos");
    FS_TRY(lm->impl->forward_generate(toks, B, L, input_pos, logits_out, hidden_out))
}
int fs_lm_forward_generate_fast(fs_lm_t* lm, const float* x, int B, int input_pos, float* logits_out) {
    FS_ARG(lm && x && logits_out, "null argument");
    FS_ARG(input_pos >= 0, "negative input_pos");
    FS_TRY(lm->impl->forward_generate_fast(x, B, input_pos, logits_out))
}
int fs_lm_fast_embed(fs_lm_t* lm, const uint32_t* ids, int n, float* out) {
    FS_ARG(lm && ids && out && n > 0, "bad argument");
    FS_TRY(lm->impl->fast_embed(ids, n, out))
}
int fs_lm_clear_fast_layer_caches(fs_lm_t* lm) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->clear_fast()) }
int fs_lm_clear_slow_layer_caches(fs_lm_t* lm) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->clear_slow()) }
int fs_lm_clear_slow_caches_until(fs_lm_t* lm, int pos) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->clear_slow_until(pos)) }
int fs_lm_curr_kv_size(fs_lm_t* lm) {
    if (!lm) { g_err = "null argument"; return -1; }
    try { return lm->impl->kv_len(); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int fs_lm_generate(fs_lm_t* lm, const uint32_t* prompt, int L, int max_new_tokens, const fs_sampling* sampling, uint64_t seed,
                   uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, fs_frame_cb cb, void* cb_user) {
    FS_ARG(lm && prompt && sampling && n_frames, "null argument");
    FS_TRY(lm->impl->generate(prompt, L, max_new_tokens, *sampling, seed, flags, codes_out, cap, n_frames, cb, cb_user))
}
int fs_lm_generate_with_hidden(fs_lm_t* lm, const uint32_t* prompt, int L, int max_new_tokens, const fs_sampling* sampling, uint64_t seed,
                               uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, float* hidden_out, size_t hidden_cap,
                               size_t* n_hidden, fs_frame_cb cb, void* cb_user) {
    FS_ARG(lm && prompt && sampling && n_frames, "null argument");
    FS_ARG(!hidden_out || n_hidden, "hidden_out without n_hidden");
    FS_TRY(lm->impl->generate(prompt, L, max_new_tokens, *sampling, seed, flags, codes_out, cap, n_frames, cb, cb_user, hidden_out,
                              hidden_cap, n_hidden))
}
int fs_lm_generate_batch(fs_lm_t* lm, const uint32_t* prompts, const int* lens, int n, int max_new_tokens, const fs_sampling* sampling,
                         uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames) {
    FS_ARG(lm && prompts && lens && sampling && codes_out && n_frames, "null argument");
    FS_TRY(lm->impl->generate_batch(prompts, lens, n, max_new_tokens, *sampling, seed, flags, codes_out, cap, n_frames))
}
int fs_lm_generate_static_batch(fs_lm_t* lm, const uint32_t* prompts, const int* lens, int n, int max_new_tokens, int audio_only,
                                const fs_sampling* sampling, uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames,
                                uint8_t* is_audio_out) {
    FS_ARG(lm && prompts && lens && sampling && codes_out && n_frames, "null argument");
    FS_ARG(audio_only != 0, "generate_static_batch(audio_only = false) is not implemented: the reference then samples the slow token over the FULL "
                            "vocabulary, never terminates a row and keeps the slow-token row in its outputs (static_batch.rs:132-141,156-173,361-364); "
                            "its server always passes true (server/lib/handlers/speech.rs:80-86)");
    FS_TRY(lm->impl->generate_batch(prompts, lens, n, max_new_tokens, *sampling, seed, flags, codes_out, cap, n_frames, is_audio_out))
}
int fs_lm_generate_multi(fs_lm_t* lm, const uint32_t* prompts, const int* lens, int n, const int* max_new_tokens, const fs_sampling* samplings,
                         const uint64_t* seeds, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames) {
    FS_ARG(lm && prompts && lens && max_new_tokens && samplings && seeds && codes_out && n_frames, "null argument");
    FS_TRY(lm->impl->generate_multi(prompts, lens, n, max_new_tokens, samplings, seeds, flags, codes_out, cap, n_frames))
}
int fs_lm_rows_supported(fs_lm_t* lm, int n, const fs_sampling* samplings, int* supported) {
    FS_ARG(lm && samplings && supported && n >= 1, "bad argument");
    FS_TRY(*supported = lm->impl->rows_supported(n, samplings) ? 1 : 0)
}
int fs_lm_debug_read_row(fs_lm_t* lm, int row, float* out, int n_frames) { FS_ARG(lm && out, "null argument"); FS_TRY(lm->impl->debug_read_row(row, out, n_frames)) }
int fs_lm_debug_read_kv(fs_lm_t* lm, int slot, int layer, int t0, int n, float* k_out, float* v_out) {
    FS_ARG(lm && k_out && v_out, "null argument");
    FS_TRY(lm->impl->debug_read_kv(slot, layer, t0, n, k_out, v_out))
}
int fs_lm_weights_arena(fs_lm_t* lm, void** dev_ptr, size_t* bytes) {
    FS_ARG(lm && dev_ptr && bytes, "null argument");
    FS_TRY(lm->impl->weights_arena(dev_ptr, bytes))
}
int fs_lm_weights_adopt(fs_lm_t* lm) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->weights_adopt()) }
int fs_lm_session_begin(fs_lm_t* lm, const fs_sampling* sampling, uint64_t seed, uint32_t flags) {
    FS_ARG(lm && sampling, "null argument");
    FS_TRY(lm->impl->session_begin(*sampling, seed, flags))
}
int fs_lm_session_add(fs_lm_t* lm, const uint32_t* prompt, int L, int max_new_tokens, int* slot) {
    FS_ARG(lm && prompt && slot, "null argument");
    FS_TRY(*slot = lm->impl->session_add(prompt, L, max_new_tokens))
}
int fs_lm_session_step(fs_lm_t* lm, int n_frames, int* n_active) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->session_step(n_frames, n_active)) }
int fs_lm_session_poll(fs_lm_t* lm, int slot, uint32_t* codes_out, size_t cap, size_t* n_frames, int* done) {
    FS_ARG(lm, "null argument");
    FS_TRY(lm->impl->session_poll(slot, codes_out, cap, n_frames, done))
}
int fs_lm_session_release(fs_lm_t* lm, int slot) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->session_release(slot)) }
int fs_lm_session_end(fs_lm_t* lm) { FS_ARG(lm, "null argument"); FS_TRY(lm->impl->session_end()) }
int fs_lm_last_stats(fs_lm_t* lm, fs_gen_stats* out) { FS_ARG(lm && out, "null argument"); FS_TRY(*out = lm->impl->last_stats()) }
void* fs_lm_stream(fs_lm_t* lm) { return lm ? lm->impl->stream() : nullptr; }
int fs_lm_bench_kernel(fs_lm_t* lm, int kind, int kv_len, int reps, float* us_per_launch) {
    FS_ARG(lm && us_per_launch, "null argument");
    FS_TRY(*us_per_launch = lm->impl->bench_kernel(kind, kv_len, reps))
}

int fs_codec_create(int device_id, int channel_div, fs_codec_t** out) {
    FS_ARG(out, "null argument");
    FS_TRY({ *out = nullptr; fs::CodecBase* p = fs::make_codec(device_id, channel_div); *out = new fs_codec{p}; })
}
void fs_codec_destroy(fs_codec_t* c) {
    if (!c) return;
    delete c->impl;
    delete c;
}
int fs_codec_load_safetensors(fs_codec_t* c, const char* path) { FS_ARG(c && path, "null argument"); FS_TRY(c->impl->load_safetensors(path)) }
int fs_codec_load_synthetic(fs_codec_t* c, uint64_t seed) { FS_ARG(c, "null argument"); FS_TRY(c->impl->load_synthetic(seed)) }
int fs_codec_decode(fs_codec_t* c, const uint32_t* codes, int b, int T, float* pcm_out) {
    FS_ARG(c && codes && pcm_out, "null argument");
    FS_TRY(c->impl->decode(codes, b, T, pcm_out))
}
int fs_codec_encode(fs_codec_t* c, const float* pcm, int n_samples, uint32_t* codes_out, size_t cap, size_t* n_frames) {
    FS_ARG(c && pcm && codes_out && n_frames, "null argument");
    FS_ARG(n_samples > 0, "empty input");
    FS_TRY(c->impl->encode(pcm, n_samples, codes_out, cap, n_frames))
}
int fs_codec_encode_batch(fs_codec_t* c, const float* pcm, int b, size_t stride, const int* n_samples, uint32_t* codes_out, size_t cap, size_t* n_frames) {
    FS_ARG(c && pcm && n_samples && codes_out && n_frames, "null argument");
    FS_ARG(b >= 1, "empty batch");
    for (int i = 0; i < b; ++i) FS_ARG(n_samples[i] > 0 && (size_t)n_samples[i] <= stride, "clip length outside (0, stride]");
    FS_TRY({
        for (int i = 0; i < b; ++i) c->impl->encode(pcm + (size_t)i * stride, n_samples[i], codes_out + (size_t)i * 8 * cap, cap, &n_frames[i]);
    })
}
int fs_codec_sample_rate(fs_codec_t* c) { return c ? c->impl->sample_rate() : -1; }
int fs_codec_stream_begin(fs_codec_t* c) { FS_ARG(c, "null argument"); FS_TRY(c->impl->stream_begin()) }
int fs_codec_stream_decode(fs_codec_t* c, const uint32_t* codes, int T, float* pcm_out) {
    FS_ARG(c && codes && pcm_out, "null argument");
    FS_TRY(c->impl->stream_decode(codes, T, pcm_out))
}
int fs_codec_stream_end(fs_codec_t* c) { FS_ARG(c, "null argument"); FS_TRY(c->impl->stream_end()) }
int fs_codec_set_precision(fs_codec_t* c, int mode) { FS_ARG(c, "null argument"); FS_TRY(c->impl->set_precision(mode)) }
int fs_codec_precision(fs_codec_t* c) { return c ? c->impl->precision() : -1; }
int fs_codec_set_range_check(fs_codec_t* c, int on) { FS_ARG(c, "null argument"); FS_TRY(c->impl->set_range_check(on != 0)) }
int fs_codec_range_stats(fs_codec_t* c, uint64_t* out5, double* last_pcm_rms_diff) {
    FS_ARG(c && out5, "null argument");
    FS_TRY(c->impl->range_stats(out5, last_pcm_rms_diff))
}

}  // extern "C"
