// Host-side engine of the dual-AR LM: owns weights, the paged KV cache, the captured frame graph and the
// generation loop.  One instance == one `DualARTransformer` (dual_ar.rs:443-457) on one GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/fishrt.h"

namespace fs {

class LMBase {
  public:
    virtual ~LMBase() {}
    virtual void load_synthetic(uint64_t seed) = 0;
    virtual void load_safetensors(const std::string& path) = 0;
    // replica start-up over RCCL: the device weight arena as raw bytes, and "the arena now holds a loaded model"
    virtual void weights_arena(void** dev_ptr, size_t* bytes) = 0;
    virtual void weights_adopt() = 0;
    virtual void forward_generate(const uint32_t* toks, int B, int L, int input_pos, float* logits, float* hidden) = 0;
    virtual void forward_generate_fast(const float* x, int B, int input_pos, float* logits) = 0;
    virtual void fast_embed(const uint32_t* ids, int n, float* out) = 0;
    virtual void clear_fast() = 0;
    virtual void clear_slow() = 0;
    virtual void clear_slow_until(int pos) = 0;
    virtual int kv_len() = 0;
    virtual void generate(const uint32_t* prompt, int L, int max_new_tokens, const fs_sampling& s, uint64_t seed,
                          uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, fs_frame_cb cb,
                          void* cb_user, float* hidden_out = nullptr, size_t hidden_cap = 0, size_t* n_hidden = nullptr) = 0;
    // is_audio (nullable): u8 [n][cap], BatchPosition::is_audio of every returned position (generate_static_batch's second return value)
    virtual void generate_batch(const uint32_t* prompts, const int* lens, int n, int max_new_tokens,
                                const fs_sampling& s, uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap,
                                size_t* n_frames, uint8_t* is_audio = nullptr) = 0;
    // R concurrent batch-1 requests (fishrt.h: fs_lm_generate_multi)
    virtual void generate_multi(const uint32_t* prompts, const int* lens, int n, const int* max_new_tokens, const fs_sampling* samplings,
                                const uint64_t* seeds, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames) = 0;
    virtual bool rows_supported(int n, const fs_sampling* samplings) = 0;
    // continuous batching over the static-batch step (fishrt.h: fs_lm_session_*)
    virtual void session_begin(const fs_sampling& s, uint64_t seed, uint32_t flags) = 0;
    virtual int session_add(const uint32_t* prompt, int L, int max_new_tokens) = 0;
    virtual void session_step(int n_frames, int* n_active) = 0;
    virtual void session_poll(int slot, uint32_t* codes_out, size_t cap, size_t* n_frames, int* done) = 0;
    virtual void session_release(int slot) = 0;
    virtual void session_end() = 0;
    virtual void debug_capture(int n_frames) = 0;
    virtual void debug_read(float* out, int n_frames) = 0;
    virtual void debug_read_row(int row, float* out, int n_frames) = 0;
    virtual void debug_read_kv(int slot, int layer, int t0, int n, float* k_out, float* v_out) = 0;
    virtual void selftest(const char* what) = 0;  // fishrt.h: fs_lm_selftest
    virtual fs_gen_stats last_stats() = 0;
    virtual void* stream() = 0;
    // measurement hook: average duration (us) of ONE launch of decode kernel `kind` (0 qkv, 1 attention, 2 wo, 3 ffn_up, 4 ffn_down) as a
    // node of a captured graph that cycles over the slow layers' distinct weights (nothing cache-resident), HIP-event timed on the
    // engine stream at KV length `kv_len`
    virtual float bench_kernel(int kind, int kv_len, int reps) = 0;
};

// FS_FP8 storage format helpers (run on the device; used by offline quantisation tools and the parity tests)
void fp8_quantize_rows(int device, const float* w, int64_t rows, int64_t cols, uint8_t* q_out, float* scales_out);
void fp8_decode_table(int device, float* out /*16 * 256*/);

LMBase* make_lm(const fs_model_args& a, const fs_token_cfg& t, int device, fs_dtype dtype, int max_batch);

}  // namespace fs
