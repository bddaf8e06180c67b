// Host-side engine of the dual-AR LM (see lm_engine.h).  All device work goes through the hand-written kernels
// of lm_kernels.hip on ONE HIP stream; the per-frame inner loop (24 slow blocks + head + sample + 8 x (4 fast blocks
// + head + sample)) is captured once into a hipGraph and replayed with zero host round trips per frame.
//
// Reference call graph implemented here:
//   generate_blocking               fish_speech_core/lib/lm/generate/single_batch.rs:217-324
//   SingleBatchGenerator::{new,next} fish_speech_core/lib/lm/generate/single_batch.rs:31-214
//   DualARTransformer::forward_generate[_fast] / cache lifecycle   fish_speech_core/lib/lm/dual_ar.rs:574-700
#include "lm_engine.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include "fs_common.h"
#include "fs_synth.h"
#include "lm_kernels.h"
#include "lm_persist.h"
#include "lm_persist_rows.h"
#include "safetensors.h"

namespace fs {

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    void alloc(size_t bytes) {
        free();
        n = bytes;
        if (bytes) FS_HIP(hipMalloc(&p, bytes));
    }
    void free() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { free(); }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct TensorDesc {
    std::string name;      // reference tensor name (dual_ar.rs loader)
    int64_t rows, cols;
    bool is_vec;           // f32 norm vector (device f32) vs WT matrix
    bool is_emb;           // embedding table: stored in KVT<WT> (bf16 when the matrices are fp8)
    size_t offset;         // byte offset in the arena (destination base of the possibly interleaved slab)
    size_t scale_off;      // fp8 matrices: byte offset of the per-row f32 scale slab (same row mapping)
    int row_mul, row_off;  // destination row = r * row_mul + row_off
    float mean;
    double stdv;
};

// rand_core SeedableRng::seed_from_u64 (PCG32 expansion) -> ChaCha key
void seed_key(uint64_t state, uint32_t* key) {
    for (int i = 0; i < 8; ++i) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}


// The persistent kernels need all of their 256 workgroups co-resident; two such launches on one GPU -- from two handles of one process
// OR from two processes -- could each hold half of the CUs and wait for the other half forever (until their spin limits).  One call at a
// time may use them per PHYSICAL device: an in-process mutex (threads) plus an advisory flock() on a per-device file keyed by the PCI bus
// id (processes; the kernel drops the lock when its holder dies).  A call that does not get the lock takes the per-node graphs instead;
// FISHRT_PERSIST_WAIT=1 makes it wait for the holder (two ranks sharing one GPU in a rehearsal: both deterministic, one after the other).
class PersistLock {
  public:
    void bind(int device) {
        std::call_once(once_, [&] {
            char bus[64] = {0};
            if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) snprintf(bus, sizeof(bus), "dev%d", device);
            for (char* c = bus; *c; ++c) if (*c == ':' || *c == '/') *c = '_';
            // The file must be visible to EVERY process that can reach the GPU (another user's process holding half the CUs is exactly the
            // case the lock exists for), so it lives in a shared directory.  ADVICE r5: never follow a planted symlink (O_NOFOLLOW), accept
            // only a regular file, and leave its mode alone unless this process created it (fchmod fails for a non-owner anyway).
            for (const char* dir : {"/dev/shm", "/tmp"}) {
                const std::string path = std::string(dir) + "/fishrt-persist-" + bus + ".lock";
                int fd = open(path.c_str(), O_RDWR | O_CLOEXEC | O_NOFOLLOW);
                if (fd < 0 && errno == ENOENT) {
                    fd = open(path.c_str(), O_CREAT | O_EXCL | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
                    if (fd >= 0) (void)fchmod(fd, 0666);
                    else if (errno == EEXIST) fd = open(path.c_str(), O_RDWR | O_CLOEXEC | O_NOFOLLOW);  // (lost the creation race)
                }
                if (fd < 0) continue;
                struct stat st;
                if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); continue; }
                fd_ = fd;
                break;
            }  // (no usable file: the lock stays process-local)
        });
    }
    bool try_lock() {
        {
            std::lock_guard<std::mutex> g(mm_);
            if (held_) return false;
            held_ = true;
        }
        if (fd_ >= 0 && flock(fd_, LOCK_EX | LOCK_NB) != 0) { release_local(); return false; }
        return true;
    }
    // FISHRT_PERSIST_WAIT: wait for the holder -- but not forever (anybody can hold an advisory lock on a shared file): the wait
    // is bounded (the variable's value in seconds; 1 = the default of 600), after which the caller runs on the per-node graphs like a
    // caller that never waited.  Returns whether the lock was taken.
    bool lock_for(double seconds) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
        {
            std::unique_lock<std::mutex> g(mm_);
            if (!cv_.wait_until(g, deadline, [&] { return !held_; })) return false;
            held_ = true;
        }
        if (fd_ < 0) return true;
        for (;;) {
            if (flock(fd_, LOCK_EX | LOCK_NB) == 0) return true;
            if (std::chrono::steady_clock::now() >= deadline) { release_local(); return false; }
            usleep(500);
        }
    }
    void lock() { (void)lock_for(600.0); }
    // (a binary semaphore, not a std::mutex: a row session takes it in session_begin and gives it back in session_end, possibly on another thread)
    void unlock() {
        if (fd_ >= 0) (void)flock(fd_, LOCK_UN);
        release_local();
    }

  private:
    void release_local() {
        { std::lock_guard<std::mutex> g(mm_); held_ = false; }
        cv_.notify_one();
    }
    std::mutex mm_;
    std::condition_variable cv_;
    bool held_ = false;
    std::once_flag once_;
    int fd_ = -1;
};
PersistLock& persist_mutex(int device) {
    static PersistLock m[64];
    m[device & 63].bind(device);
    return m[device & 63];
}
// try (default) or wait (FISHRT_PERSIST_WAIT = seconds; 1 = 600 s) for the device's persistent kernels
std::unique_lock<PersistLock> acquire_persist(int device) {
    PersistLock& pl = persist_mutex(device);
    if (const char* w = getenv("FISHRT_PERSIST_WAIT")) {
        const double secs = atof(w) > 1.0 ? atof(w) : 600.0;
        if (pl.lock_for(secs)) return std::unique_lock<PersistLock>(pl, std::adopt_lock);
        return std::unique_lock<PersistLock>(pl, std::defer_lock);
    }
    return std::unique_lock<PersistLock>(pl, std::try_to_lock);
}

constexpr int kRows = 256;     // static-batch generator: max sequences per step (32-row MFMA panels; <= kPartRows)
constexpr int kRowsCap = 2048; // row capacity of the MFMA row buffers: prompt tokens per prefill pass (one sequence, or a group of
                               // left-padded static-batch prompts)
constexpr int kPartRows = 512; // rows of the chunked-attention partials buffer (static-batch steps; prefill passes when head_dim != 64)

}  // namespace

template <typename WT>
class LM final : public LMBase {
    using KT = KVT<WT>;  // KV cache / embedding storage type
    static constexpr bool kFp8 = std::is_same<WT, fp8_t>::value;

  public:
    LM(const fs_model_args& a, const fs_token_cfg& t, int device, int max_batch) : a_(a), t_(t), device_(device), B_(max_batch) {
        FS_REQUIRE(max_batch >= 1, "max_batch must be >= 1");
        FS_REQUIRE(a.dim == a.n_head * a.head_dim, "dim must equal n_head * head_dim (dual_ar.rs:173,381)");
        FS_REQUIRE(a.n_head % a.n_local_heads == 0, "n_head must be a multiple of n_local_heads");
        FS_REQUIRE(a.head_dim % 8 == 0 && a.num_codebooks + 1 <= 16, "unsupported head_dim / num_codebooks");
        FS_REQUIRE(t.im_end_id < (uint32_t)a.vocab_size, "im_end_id outside the vocabulary");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            throw Error("no HIP device visible: libfishrt has no CPU fallback (MI355X / gfx950 required)");
        FS_REQUIRE(device >= 0 && device < ndev, "device_id out of range");
        FS_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        FS_HIP(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw Error(std::string("libfishrt kernels are built for gfx950 only; device reports ") + prop.gcnArchName);
        FS_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
        d_.dim = a.dim; d_.inter = a.intermediate_size; d_.H = a.n_head; d_.Hk = a.n_local_heads; d_.Dh = a.head_dim;
        d_.n_rep = a.n_head / a.n_local_heads; d_.eps = a.norm_eps;
        legacy_ = !t.has_semantic_end;  // Fish <= 1.4: slow token is a 2-way {pad, im_end} draw (single_batch.rs:104-124)
        // generic DualAR token layout (utils.rs:17-30): <|im_end|> does not directly precede the semantic range -- the slow head's
        // candidates are [im_end] ++ [semantic_start, V) (literally: any control token behind the range, <|im_end|> itself included, stays a
        // candidate), gathered into one head image at load time
        generic_ = !legacy_ && t.im_end_id + 1 != t.semantic_start_id;
        if (!legacy_) FS_REQUIRE(t.semantic_start_id >= 1 && t.semantic_start_id < (uint32_t)a.vocab_size, "semantic_start_id outside the vocabulary");
        n_audio_ = legacy_ ? 2 : (generic_ ? a.vocab_size - (int)t.semantic_start_id + 1 : a.vocab_size - (int)t.im_end_id);
        if (legacy_) FS_REQUIRE(t.pad_id < (uint32_t)a.vocab_size, "pad_id outside the vocabulary");
        plan_tensors();
        alloc_runtime();
        for (auto& e : ev_) FS_HIP(hipEventCreate(&e));
        for (auto& e : ev_batch_) FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    ~LM() override {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(st_);
        for (auto& kvp : batch_graphs_) if (kvp.second) (void)hipGraphExecDestroy(kvp.second);
        for (auto& kvp : multi_graphs_) if (kvp.second) (void)hipGraphExecDestroy(kvp.second);
        for (auto& kvp : graphs_) {
            if (kvp.second.first) (void)hipGraphExecDestroy(kvp.second.first);
            if (kvp.second.second) (void)hipGraphExecDestroy(kvp.second.second);
        }
        for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_batch_) if (e) (void)hipEventDestroy(e);
        if (h_pin_) (void)hipHostFree(h_pin_);
        if (ev_pf_) (void)hipEventDestroy(ev_pf_);
        if (st_pf_) (void)hipStreamDestroy(st_pf_);
        if (st_) (void)hipStreamDestroy(st_);
    }

    // ------------------------------------------------------------------------------------------ weights
    void load_synthetic(uint64_t seed) override {
        use_device();
        // a bf16 (or fp8) checkpoint holds bf16-rounded vectors / embeddings; fp8 matrices are quantised from the raw values
        const int round_bf16 = std::is_same<WT, float>::value ? 0 : 1;
        uint8_t* base = arena_.as<uint8_t>();
        for (const auto& td : tensors_) {
            const uint64_t key = synth_fnv1a64(td.name.c_str()) ^ seed;
            const float sc = synth_scale(td.stdv);
            if (td.is_vec)
                launch_synth_fill<float>((float*)(base + td.offset), key, td.rows, td.cols, 1, 0, td.mean, sc, round_bf16, st_);
            else if (td.is_emb)
                launch_synth_fill<KT>((KT*)(base + td.offset), key, td.rows, td.cols, td.row_mul, td.row_off, td.mean, sc, round_bf16, st_);
            else if constexpr (kFp8)
                launch_synth_quant_fp8(base + td.offset, (float*)(base + td.scale_off), key, td.rows, td.cols, td.row_mul, td.row_off,
                                       td.mean, sc, st_);
            else
                launch_synth_fill<WT>((WT*)(base + td.offset), key, td.rows, td.cols, td.row_mul, td.row_off, td.mean, sc, round_bf16, st_);
        }
        FS_HIP(hipStreamSynchronize(st_));
        loaded_ = true;
        refresh_legacy_head();
        pack_persist();
    }

    void load_safetensors(const std::string& path) override {
        use_device();
        SafeTensors st(path);
        DevBuf stage;
        std::vector<float> host;
        for (const auto& td : tensors_) {
            std::string name = td.name;
            if (name == "output.weight" && a_.tie_word_embeddings) name = "embeddings.weight";  // dual_ar.rs:482-486
            // exact shape, as candle's VarBuilder::get: [rows, cols] for matrices / embeddings, [n] for norm vectors (a transposed tensor
            // of the same element count is an error, not a silent load)
            const StTensor* t = &st.get(name, td.is_vec ? std::vector<int64_t>{td.cols} : std::vector<int64_t>{td.rows, td.cols});
            host.resize((size_t)t->numel());
            SafeTensors::to_f32(*t, host.data());
            if (stage.n < host.size() * 4) stage.alloc(host.size() * 4);
            FS_HIP(hipMemcpyAsync(stage.p, host.data(), host.size() * 4, hipMemcpyHostToDevice, st_));
            uint8_t* base = arena_.as<uint8_t>();
            if (td.is_vec)
                launch_convert_rows<float>((float*)(base + td.offset), stage.as<float>(), td.rows, td.cols, 1, 0, st_);
            else if (td.is_emb)
                launch_convert_rows<KT>((KT*)(base + td.offset), stage.as<float>(), td.rows, td.cols, td.row_mul, td.row_off, st_);
            else if constexpr (kFp8)
                launch_quant_rows_fp8(base + td.offset, (float*)(base + td.scale_off), stage.as<float>(), td.rows, td.cols, td.row_mul,
                                      td.row_off, st_);
            else
                launch_convert_rows<WT>((WT*)(base + td.offset), stage.as<float>(), td.rows, td.cols, td.row_mul, td.row_off, st_);
            FS_HIP(hipStreamSynchronize(st_));
        }
        loaded_ = true;
        refresh_legacy_head();
        pack_persist();
    }

    // The arena is a pure function of (model args, token config, dtype, checkpoint): one rank loads the checkpoint, the others receive
    // the bytes (ncclBroadcast over xGMI instead of N disk reads + N conversions) and derive the load-time extras themselves.
    void weights_arena(void** dev_ptr, size_t* bytes) override {
        use_device();
        FS_HIP(hipStreamSynchronize(st_));
        *dev_ptr = arena_.p; *bytes = arena_bytes_;
    }
    void weights_adopt() override {
        use_device();
        FS_REQUIRE(!sess_active_, "the handle is in session mode (fs_lm_session_end first)");
        FS_HIP(hipDeviceSynchronize());  // the bytes were written by another stream (the communicator's)
        loaded_ = true;
        refresh_legacy_head();
        pack_persist();
    }

    // ------------------------------------------------------------------------------------------ teacher-forced API
    void forward_generate(const uint32_t* toks, int B, int L, int input_pos, float* logits, float* hidden) override {
        use_device();
        require_loaded();
        FS_REQUIRE(!sess_active_, "the handle is in session mode (fs_lm_session_end first)");
        FS_REQUIRE(B >= 1 && B <= B_, "batch size exceeds the handle's max_batch");
        FS_REQUIRE(L >= 1, "empty input");
        const int C1 = a_.num_codebooks + 1;
        for (int b = 1; b < B; ++b) FS_REQUIRE(seq_len_[b] == seq_len_[0], "KV cache length differs across batch rows");
        if (input_pos + L > a_.max_seq_len) throw Error("input_pos + seq_len exceeds max_seq_len (dual_ar.rs:623-624)");
        validate_tokens(toks, (size_t)B * C1 * L, B, L);
        std::vector<float> lg, hd;
        for (int b = 0; b < B; ++b) {
            ensure_capacity(b, seq_len_[b] + L);
            FS_HIP(hipMemcpyAsync(d_prompt_.p, toks + (size_t)b * C1 * L, sizeof(uint32_t) * C1 * L, hipMemcpyHostToDevice, st_));
            SeqState s = {};
            s.pos = seq_len_[b]; s.rope_off = input_pos - seq_len_[b]; s.prompt_L = L;
            FS_HIP(hipMemcpyAsync(state(b), &s, sizeof(s), hipMemcpyHostToDevice, st_));
            prefill_tokens(b, L, /*use_graph=*/false);
            seq_len_[b] += L;
            if (hidden) FS_HIP(hipMemcpyAsync(hidden + (size_t)b * a_.dim, x(b), sizeof(float) * a_.dim, hipMemcpyDeviceToHost, st_));
            if (logits) {
                LmKernels<WT>::head(d_, x(b), norm_w_, out_w_, out_s_, a_.vocab_size, d_logits_slow_.as<float>(), st_);
                FS_HIP(hipMemcpyAsync(logits + (size_t)b * a_.vocab_size, d_logits_slow_.p, sizeof(float) * a_.vocab_size,
                                      hipMemcpyDeviceToHost, st_));
            }
            FS_HIP(hipStreamSynchronize(st_));
        }
    }

    void forward_generate_fast(const float* xin, int B, int input_pos, float* logits) override {
        use_device();
        require_loaded();
        FS_REQUIRE(B >= 1 && B <= B_, "batch size exceeds the handle's max_batch");
        if (input_pos >= a_.max_seq_len) throw Error("input_pos exceeds max_seq_len");
        for (int b = 0; b < B; ++b) {
            FS_REQUIRE(fast_len_[b] < 8, "fast decoder KV holds at most 8 positions per frame (num_codebooks)");
            FS_HIP(hipMemcpyAsync(xf(b), xin + (size_t)b * a_.dim, sizeof(float) * a_.dim, hipMemcpyHostToDevice, st_));
            enqueue_fast_layers(b, fast_len_[b], input_pos);
            LmKernels<WT>::head(d_, xf(b), fast_norm_w_, fast_out_w_, fast_out_s_, a_.codebook_size, d_logits_fast_.as<float>(), st_);
            FS_HIP(hipMemcpyAsync(logits + (size_t)b * a_.codebook_size, d_logits_fast_.p, sizeof(float) * a_.codebook_size,
                                  hipMemcpyDeviceToHost, st_));
            FS_HIP(hipStreamSynchronize(st_));
            fast_len_[b] += 1;
        }
    }

    void fast_embed(const uint32_t* ids, int n, float* out) override {
        use_device();
        require_loaded();
        for (int i = 0; i < n; ++i) FS_REQUIRE(ids[i] < (uint32_t)a_.codebook_size, "fast embedding id out of range");
        DevBuf di, dout;
        di.alloc(sizeof(uint32_t) * n);
        dout.alloc(sizeof(float) * n * a_.dim);
        FS_HIP(hipMemcpyAsync(di.p, ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st_));
        LmKernels<WT>::fast_embed(d_, fast_emb_, di.as<uint32_t>(), n, dout.as<float>(), st_);
        FS_HIP(hipMemcpyAsync(out, dout.p, sizeof(float) * n * a_.dim, hipMemcpyDeviceToHost, st_));
        FS_HIP(hipStreamSynchronize(st_));
    }

    void clear_fast() override { std::fill(fast_len_.begin(), fast_len_.end(), 0); }
    void clear_slow() override { for (int b = 0; b < B_; ++b) truncate(b, 0); }
    void clear_slow_until(int pos) override {
        FS_REQUIRE(pos >= 0, "negative position");
        for (int b = 0; b < B_; ++b) truncate(b, std::min(seq_len_[b], pos));
    }
    int kv_len() override { return seq_len_[0]; }
    // test hook (fs_lm_debug_capture / fs_lm_debug_read): the persistent fast decoder records, for the first n generator iterations of
    // each generate call, the (penalised, masked) logits every one of the 9 decisions of a frame saw and the index it picked
    void debug_capture(int n_frames) override {
        use_device();
        FS_REQUIRE(n_frames >= 0 && n_frames <= 4096, "bad frame count");
        FS_HIP(hipStreamSynchronize(st_));
        cap_frames_ = n_frames;
        if (n_frames) { d_cap_.alloc(sizeof(float) * (size_t)n_frames * 9 * 2048); FS_HIP(hipMemset(d_cap_.p, 0, d_cap_.n)); }
        else d_cap_ = DevBuf();
        for (auto& kv : graphs_) { (void)hipGraphExecDestroy(kv.second.first); (void)hipGraphExecDestroy(kv.second.second); }
        graphs_.clear();  // the captured launches carry the buffer pointer
        for (auto& kv : multi_graphs_) if (kv.second) (void)hipGraphExecDestroy(kv.second);
        multi_graphs_.clear();
        g_frame_ = g_step_ = nullptr;
        drop_batch_graphs();  // (the row-path step graphs include the capture kernels only while the hook is armed)
        if (!n_frames) d_rcap_ = DevBuf();
    }
    void debug_read(float* out, int n_frames) override {
        use_device();
        FS_REQUIRE(n_frames >= 0 && n_frames <= cap_frames_, "more frames than were captured");
        FS_HIP(hipStreamSynchronize(st_));
        FS_HIP(hipMemcpy(out, d_cap_.p, sizeof(float) * (size_t)n_frames * 9 * 2048, hipMemcpyDeviceToHost));
    }
    // test hook (fs_lm_debug_read_kv): the cached K / V rows [t0, t0 + n) of slow layer `layer` of KV slot `slot`, as f32 [n][Hkv][Dh]
    void debug_read_kv(int slot, int layer, int t0, int n, float* k_out, float* v_out) override {
        use_device();
        FS_REQUIRE(slot >= 0 && slot < B_ && layer >= 0 && layer < a_.n_layer, "bad KV slot / layer");
        FS_REQUIRE(t0 >= 0 && n >= 0 && t0 + n <= (int)seq_pages_[slot].size() * KV_PAGE, "rows outside the slot's KV pages");
        FS_HIP(hipStreamSynchronize(st_));
        const int Hk = a_.n_local_heads, Dh = a_.head_dim;
        std::vector<KT> pk(page_elems_), pv(page_elems_);
        const KT* kpool = kv_pool_.as<KT>() + (size_t)layer * 2 * n_pages_ * page_elems_;
        const KT* vpool = kpool + (size_t)n_pages_ * page_elems_;
        int have = -1;
        for (int t = t0; t < t0 + n; ++t) {
            const int pg = seq_pages_[slot][t >> 6];
            if (pg != have) {
                FS_HIP(hipMemcpy(pk.data(), kpool + (size_t)pg * page_elems_, sizeof(KT) * page_elems_, hipMemcpyDeviceToHost));
                FS_HIP(hipMemcpy(pv.data(), vpool + (size_t)pg * page_elems_, sizeof(KT) * page_elems_, hipMemcpyDeviceToHost));
                have = pg;
            }
            for (int g = 0; g < Hk; ++g)
                for (int d = 0; d < Dh; ++d) {
                    const size_t src = ((size_t)g * KV_PAGE + (t & 63)) * Dh + d, dst = ((size_t)(t - t0) * Hk + g) * Dh + d;
                    k_out[dst] = kv_f32(pk[src]); v_out[dst] = kv_f32(pv[src]);
                }
        }
    }
    static float kv_f32(float v) { return v; }
    static float kv_f32(bf16_t v) { return bf16_to_f32_host(v); }
    void debug_read_row(int row, float* out, int n_frames) override {
        use_device();
        FS_REQUIRE(row >= 0 && d_rcap_.p, "no row capture (fs_lm_debug_capture, then fs_lm_generate_multi / fs_lm_generate_batch / a session)");
        FS_REQUIRE(n_frames >= 0 && n_frames <= cap_frames_, "more frames than were captured");
        FS_REQUIRE(sizeof(float) * ((size_t)row + 1) * cap_frames_ * 9 * 2048 <= d_rcap_.n, "row beyond the captured rows");
        FS_HIP(hipStreamSynchronize(st_));
        FS_HIP(hipMemcpy(out, d_rcap_.as<float>() + (size_t)row * cap_frames_ * 9 * 2048, sizeof(float) * (size_t)n_frames * 9 * 2048, hipMemcpyDeviceToHost));
    }
    fs_gen_stats last_stats() override { return stats_; }
    void* stream() override { return (void*)st_; }

    // fs_lm_selftest("persist"): the persistent decode kernels of THIS binary against the per-node kernels, on the loaded
    // weights.  k_slow_persist requests weight slices through loads the compiler's wait-count pass does not see (lm_persist_slow.hip:
    // ps_load16_unseen) and k_fast_persist keeps 168 pinned weight registers per lane -- both depend on how hipcc allocates registers, so a
    // toolchain or flag change must be caught by running, not by reading.  Protocol: a 12-position prompt, 4 greedy frames on the
    // persistent path with the decision capture armed; then ONE teacher-forced per-node step (forward_generate at the position of frame 1,
    // fed frame 0's picks; forward_generate_fast on its hidden state) and a comparison of frame 1's captured logits -- the first slow
    // step k_slow_persist computed, the first codebook pass of k_fast_persist behind it -- with the per-node logits.  Both paths share the
    // prefill's K/V rows and differ in summation order only, so the bound is loose for rounding and tight for a wrong weight register:
    // 2e-2 x max(1, max |logit|).
    void selftest(const char* what) override {
        use_device();
        require_loaded();
        FS_REQUIRE(what && std::string(what) == "persist", "unknown handle self-test (known: \"persist\")");
        FS_REQUIRE(!sess_active_, "the handle is in session mode (fs_lm_session_end first)");
        FS_REQUIRE(cap_frames_ == 0, "decision capture is armed (fs_lm_debug_capture(lm, 0) first)");
        if (!persist_ok_ || !pslow_ok_) throw Error("self-test \"persist\": this handle has no persistent kernels (dtype / geometry / device)");
        const int C = a_.num_codebooks, C1 = C + 1, L = 12;
        const SampleCfg cfg = base_cfg();
        std::vector<uint32_t> prompt((size_t)C1 * L, 0u);
        const uint32_t n_text = std::max<uint32_t>(1u, std::min<uint32_t>(cfg.im_end_id, 1000u));
        for (int i = 0; i < L; ++i) prompt[i] = (uint32_t)((7919ull * (uint64_t)(i + 1) + 13ull) % n_text);
        clear_slow();
        debug_capture(2);
        struct Restore { LM* lm; ~Restore() { try { lm->debug_capture(0); lm->clear_slow(); lm->clear_fast(); } catch (...) {} } } restore{this};
        fs_sampling sa = {};
        sa.temp = 0.0; sa.top_p = 1.0; sa.top_k = 0; sa.repetition_penalty = 1.0f;
        std::vector<uint32_t> codes((size_t)C * 8, 0u);
        size_t nf = 0;
        generate(prompt.data(), L, L + 2, sa, 1, FS_GEN_IGNORE_EOS, codes.data(), 8, &nf, nullptr, nullptr, nullptr, 0, nullptr);
        if (stats_.kernels_per_frame != 2)
            throw Error("self-test \"persist\": the call did not take the persistent kernels (another handle of this GPU holds them?)");
        std::vector<float> cap((size_t)2 * 9 * 2048);
        debug_read(cap.data(), 2);
        std::vector<uint32_t> in(C1);
        in[0] = legacy_ ? cfg.pad_id : audio_tok(cfg, (int)cap[2047]);
        for (int c = 0; c < C; ++c) in[1 + c] = (uint32_t)cap[(size_t)(1 + c) * 2048 + 1024];
        clear_slow_until(L);
        std::vector<float> lg(a_.vocab_size), hid(a_.dim), lf(a_.codebook_size);
        forward_generate(in.data(), 1, 1, L, lg.data(), hid.data());
        clear_fast();
        forward_generate_fast(hid.data(), 1, 0, lf.data());
        const float* c_slow = cap.data() + (size_t)(9 + 0) * 2048;
        const float* c_fast = cap.data() + (size_t)(9 + 1) * 2048;
        float worst = 0.f, scale = 1.f;
        auto cmp = [&](float ref, float got) { scale = std::max(scale, std::fabs(ref)); worst = std::max(worst, std::fabs(ref - got)); };
        if (legacy_) { cmp(lg[cfg.pad_id], c_slow[0]); cmp(lg[cfg.im_end_id], c_slow[1]); }
        else for (int i = 1; i < n_audio_; ++i) cmp(lg[audio_tok(cfg, i)], c_slow[i]);  // (index 0 = <|im_end|>: masked to -inf under FS_GEN_IGNORE_EOS)
        const float worst_slow = worst;
        for (int i = 0; i < a_.codebook_size; ++i) cmp(lf[i], c_fast[i]);
        if (!(worst <= 2e-2f * scale))
            throw Error("self-test \"persist\": persistent kernels disagree with the per-node kernels (max |dlogit| slow " + std::to_string(worst_slow) +
                        ", all " + std::to_string(worst) + " at logit scale " + std::to_string(scale) + "): rebuild with the pinned toolchain or run with FISHRT_NO_PERSIST=1");
    }

    float bench_kernel(int kind, int kv_len, int reps) override {
        use_device();
        require_loaded();
        FS_REQUIRE(kind >= 0 && kind <= 7 && reps >= 1, "bad kernel id");
        FS_REQUIRE(kv_len >= 1 && kv_len < a_.max_seq_len, "bad KV length");
        clear_slow();
        ensure_capacity(0, kv_len);
        SeqState s0 = {};
        s0.pos = kv_len - 1;
        FS_HIP(hipMemcpyAsync(state(0), &s0, sizeof(s0), hipMemcpyHostToDevice, st_));
        set_bucket(kv_len);
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        FS_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
        int nodes = a_.n_layer;
        if (kind == 5) {  // the fast decoder's layers over the 8 codebook positions (4 nodes per layer)
            for (int cbi = 0; cbi < a_.num_codebooks; ++cbi) enqueue_fast_layers(0, cbi, cbi);
            nodes = a_.num_codebooks * a_.n_fast_layer * 4;
        } else if (kind == 6 || kind == 7) {  // fast / slow (audio-range) head GEMV
            nodes = 16;
            for (int i = 0; i < nodes; ++i) {
                if (kind == 6) LmKernels<WT>::head(d_, xf(0), fast_norm_w_, fast_out_w_, fast_out_s_, a_.codebook_size, d_logits_fast_.as<float>(), st_);
                else LmKernels<WT>::head(d_, x(0), norm_w_, slow_head_w(), slow_head_s(), n_audio_, d_logits_slow_.as<float>(), st_);
            }
        } else
        for (int l = 0; l < a_.n_layer; ++l) {
            const LayerW& w = slow_[l];
            KVView kv = slow_kv(l, 0);
            switch (kind) {
                case 0: LmKernels<WT>::qkv(d_, x(0), w, d_cos_.as<float>(), d_sin_.as<float>(), state(0), 0, 0, d_q_.as<float>(), kv, st_); break;
                case 1: LmKernels<WT>::attn_decode(d_, d_q_.as<float>(), kv, state(0), d_part_.as<float>(), n_chunks_, nc_launch_, st_); break;
                case 2: LmKernels<WT>::wo(d_, d_part_.as<float>(), n_chunks_, nc_launch_, state(0), nullptr, kv, 0, w, x(0), st_); break;
                case 3: LmKernels<WT>::ffn_up(d_, x(0), w, d_act_.as<float>(), st_); break;
                default: LmKernels<WT>::ffn_down(d_, d_act_.as<float>(), w, x(0), st_); break;
            }
        }
        FS_HIP(hipStreamEndCapture(st_, &g));
        FS_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        FS_HIP(hipGraphDestroy(g));
        FS_HIP(hipGraphLaunch(ge, st_));
        FS_HIP(hipStreamSynchronize(st_));
        FS_HIP(hipEventRecord(ev_[0], st_));
        for (int r = 0; r < reps; ++r) FS_HIP(hipGraphLaunch(ge, st_));
        FS_HIP(hipEventRecord(ev_[1], st_));
        FS_HIP(hipStreamSynchronize(st_));
        float ms = 0.f;
        FS_HIP(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
        (void)hipGraphExecDestroy(ge);
        clear_slow();
        return ms * 1e3f / (float)(reps * nodes);
    }

    // ------------------------------------------------------------------------------------------ generate_blocking
    void generate(const uint32_t* prompt, int L, int max_new_tokens, const fs_sampling& s, uint64_t seed, uint32_t flags,
                  uint32_t* codes_out, size_t cap, size_t* n_frames, fs_frame_cb cb, void* cb_user, float* hidden_out, size_t hidden_cap,
                  size_t* n_hidden) override {
        use_device();
        require_loaded();
        FS_REQUIRE(!sess_active_, "the handle is in session mode (fs_lm_session_end first)");
        const int C = a_.num_codebooks, C1 = C + 1;
        FS_REQUIRE(L >= 1, "empty prompt");
        FS_REQUIRE(max_new_tokens >= 0, "negative max_new_tokens");
        validate_tokens(prompt, (size_t)C1 * L, 1, L);
        const int n_cached = seq_len_[0];
        if (n_cached + L > a_.max_seq_len) throw Error("prompt exceeds max_seq_len (dual_ar.rs:623-624)");
        // iteration budget (single_batch.rs:61,77,193-197): prefill iteration + one per k with L + k - 1 <= max_new_tokens
        long long n_iter = 1 + std::max<long long>(0, (long long)max_new_tokens - L + 1);
        bool clamped = false;
        const long long room = (long long)a_.max_seq_len - (n_cached + L) + 1;  // iterations that fit the RoPE table / KV
        if (n_iter > room) { n_iter = room; clamped = true; }
        FS_REQUIRE(n_iter <= out_cap_, "generation longer than the output staging buffer");
        ensure_capacity(0, n_cached + L + (int)n_iter - 1);
        // device-side setup
        SampleCfg cfg = base_cfg();
        cfg.temp = (float)s.temp; cfg.top_p = (float)s.top_p; cfg.top_p64 = s.top_p;
        cfg.top_k = (int)std::min<uint64_t>(s.top_k, 1u << 30);
        cfg.rep_pen = s.repetition_penalty; cfg.ignore_eos = (flags & FS_GEN_IGNORE_EOS) ? 1 : 0;
        cfg.batch_rows = batch_rows_; cfg.batch_row = batch_row_; cfg.batch_calls = a_.num_codebooks + 1;  // batch_rows > 0 only inside generate_batch_sequential
        FS_HIP(hipMemcpyAsync(d_cfg_.p, &cfg, sizeof(cfg), hipMemcpyHostToDevice, st_));
        // greedy decoding on a Fish-geometry bf16 handle: the 8 fast-decoder passes of a frame run as ONE persistent launch
        // (lm_persist.hip) instead of 144 graph nodes, if no other handle of this GPU is using it right now
        std::unique_lock<PersistLock> plock;
        use_persist_ = use_pslow_ = false;
        if ((persist_ok_ || pslow_ok_) && !(flags & FS_GEN_NO_PERSIST)) {
            plock = acquire_persist(device_);
            // the fast kernel decides in-launch: greedily (host ArgMax rule), or with the block-parallel top-k / top-p sampler when top_k <= 256
            persist_sampled_ = cfg.temp != 0.f;
            use_persist_ = plock.owns_lock() && persist_ok_ && batch_rows_ == 0 &&
                           (cfg.temp == 0.f || fast_persist_samples(cfg.temp, cfg.top_k, a_.codebook_size));
            use_pslow_ = plock.owns_lock() && pslow_ok_;                          // the slow kernel feeds any sampler
        }
        // generate_blocking_with_hidden: the slow sampler stores the hidden state of every iteration through this pointer cell
        if (hidden_out && !d_hidden_.p) d_hidden_.alloc(sizeof(float) * (size_t)out_cap_ * a_.dim);
        float* hid_dev = hidden_out ? d_hidden_.as<float>() : nullptr;
        FS_HIP(hipMemcpyAsync(d_hid_slot_.p, &hid_dev, sizeof(hid_dev), hipMemcpyHostToDevice, st_));
        RngState rng = {};
        seed_key(seed, rng.key);
        FS_HIP(hipMemcpyAsync(d_rng_.p, &rng, sizeof(rng), hipMemcpyHostToDevice, st_));
        SeqState ss = {};
        ss.pos = n_cached; ss.prompt_L = L;
        FS_HIP(hipMemcpyAsync(state(0), &ss, sizeof(ss), hipMemcpyHostToDevice, st_));
        FS_HIP(hipMemcpyAsync(d_prompt_.p, prompt, sizeof(uint32_t) * C1 * L, hipMemcpyHostToDevice, st_));
        launch_reppen_reset(rp_, C, a_.codebook_size, st_);
        clear_fast();

        stats_ = {};
        FS_HIP(hipEventRecord(ev_[0], st_));
        // prefill: the first L-1 prompt tokens (MFMA chunks of <= 64 tokens for bf16 weights; sequential token steps for
        // f32 parity handles), then the last prompt token runs as the first frame
        prefill_tokens(0, L - 1, /*use_graph=*/true);
        LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(), d_prompt_.as<uint32_t>(), state(0),
                             x(0), st_);
        // frame `it` runs at KV length T = n_cached + L + it: pick the graph captured for that attention chunk bucket
        // FS_GEN_TIME_KERNELS: the frame's two persistent launches one by one, a HIP event in front of / between / behind them
        const bool time_k = (flags & FS_GEN_TIME_KERNELS) && use_persist_ && use_pslow_ && fold_slow_sampler();
        bool b1_rows_fast = false;
        if constexpr (std::is_same<WT, bf16_t>::value) {
            if (getenv("FISHRT_B1_ROWS_FAST") && use_persist_ && use_pslow_ && fold_slow_sampler() && !legacy_) {
                ensure_rows(2);
                b1_rows_fast = true;
                const int big = 1 << 30;
                FS_HIP(hipMemcpyAsync(d_rcfg_.p, &cfg, sizeof(cfg), hipMemcpyHostToDevice, st_));
                FS_HIP(hipMemcpyAsync(d_rbudget_.p, &big, sizeof(int), hipMemcpyHostToDevice, st_));
                FS_HIP(hipMemcpyAsync(d_rrng_.p, &rng, sizeof(rng), hipMemcpyHostToDevice, st_));
                launch_reppen_reset(rows_rp(0), C, a_.codebook_size, st_);
            }
        }
        std::vector<hipEvent_t> kev;
        const bool multi_ok = use_persist_ && use_pslow_ && fold_slow_sampler() && !time_k && !b1_rows_fast;
        auto launch_frame = [&](long long it_) {
            set_bucket(n_cached + L + (int)it_);
            if (b1_rows_fast) {  // experiment hook (FISHRT_B1_ROWS_FAST): the fast decoder of this batch-1 request on k_fast_rows<1>
                launch_slow_persist(pslow_args(), st_);
                RowsFastArgs F = rows_fast_args(0, 1);
                F.slow_logits = d_logits_slow_.as<float>();
                F.cap = nullptr;
                launch_rows_fast(F, 1, persist_sampled_, st_);
                return;
            }
            if (time_k && it_ >= 1) {
                for (int i = 0; i < 3; ++i) { hipEvent_t e; FS_HIP(hipEventCreate(&e)); kev.push_back(e); }
                FS_HIP(hipEventRecord(kev[kev.size() - 3], st_));
                launch_slow_persist(pslow_args(), st_);
                FS_HIP(hipEventRecord(kev[kev.size() - 2], st_));
                launch_fast_persist(persist_args(), persist_sampled_, st_);
                FS_HIP(hipEventRecord(kev[kev.size() - 1], st_));
                return;
            }
            use_graphs_for_bucket();
            FS_HIP(hipGraphLaunch(g_frame_, st_));
        };
        launch_frame(0);
        FS_HIP(hipEventRecord(ev_[1], st_));
        stats_.graph_launches = (uint64_t)L;
        stats_.kernels_per_frame = (uint64_t)((use_pslow_ ? 1 : a_.n_layer * 5 + 1) + (fold_slow_sampler() ? 0 : 1) +
                                              (use_persist_ ? 1 : a_.num_codebooks * (a_.n_fast_layer * 4 + 2)));
        // decode: one graph replay per frame, enqueued in batches of CHUNK.  Behind every batch the stream copies the generator state
        // and the batch's code columns into pinned memory and records an event; the host looks at batch b (done flag, frame callback)
        // while batch b + 1 is already running, so the GPU never waits for the host between batches
        const int CHUNK = cb ? 8 : 32;
        long long it = 1;
        size_t delivered = 0;
        bool stop = false, ended = false;
        SeqState* hs = reinterpret_cast<SeqState*>(h_pin_);  // final state (after the loop)
        auto slot_state = [&](int sl) { return reinterpret_cast<SeqState*>((char*)h_pin_ + 256 + 256 * sl); };
        auto slot_codes = [&](int sl) { return reinterpret_cast<uint32_t*>((char*)h_pin_ + 1024 + 1024 * sl); };  // [C][CHUNK]
        static_assert(sizeof(SeqState) <= 256, "pinned slot size");
        struct Batch { long long first, end; int slot; };
        auto enqueue_batch = [&](int sl) {
            Batch b{it, std::min<long long>(n_iter, it + CHUNK), sl};
            while (it < b.end) {
                // several frames per graph launch where the batch has them and they share an attention chunk bucket (multi_frame_graph)
                const int nf = frames_per_graph();
                if (multi_ok && nf == 0) {
                    set_bucket(n_cached + L + (int)it);
                    launch_slow_persist(pslow_args(), st_);
                    launch_fast_persist(persist_args(), persist_sampled_, st_);
                    ++it; stats_.graph_launches += 1;
                } else if (multi_ok && nf > 1 && b.end - it >= nf && chunk_bucket(n_cached + L + (int)it) == chunk_bucket(n_cached + L + (int)it + nf - 1)) {
                    set_bucket(n_cached + L + (int)it);
                    use_graphs_for_bucket();  // (keeps the single-frame graph of the bucket current as well)
                    FS_HIP(hipGraphLaunch(multi_frame_graph(), st_));
                    it += nf; stats_.graph_launches += (uint64_t)nf;
                } else {
                    launch_frame(it); ++it; stats_.graph_launches += 1;
                }
            }
            FS_HIP(hipMemcpyAsync(slot_state(sl), state(0), sizeof(SeqState), hipMemcpyDeviceToHost, st_));
            if (cb)  // columns [first, end) of every codebook row (frame index == iteration index until <|im_end|>)
                FS_HIP(hipMemcpy2DAsync(slot_codes(sl), sizeof(uint32_t) * CHUNK, d_out_.as<uint32_t>() + b.first, sizeof(uint32_t) * out_cap_,
                                        sizeof(uint32_t) * (size_t)(b.end - b.first), (size_t)C, hipMemcpyDeviceToHost, st_));
            FS_HIP(hipEventRecord(ev_batch_[sl], st_));
            return b;
        };
        auto retire_batch = [&](const Batch& b) {  // true when the generator has terminated or the callback asked to stop
            FS_HIP(hipEventSynchronize(ev_batch_[b.slot]));
            const SeqState* s2 = slot_state(b.slot);
            if (cb) {
                const size_t n = std::min<size_t>((size_t)s2->n_out, (size_t)b.end);
                std::vector<uint32_t> fr(C);
                for (size_t f = std::max<size_t>(delivered, (size_t)b.first); f < n && !stop; ++f) {
                    for (int c = 0; c < C; ++c) fr[c] = slot_codes(b.slot)[(size_t)c * CHUNK + (f - (size_t)b.first)];
                    if (cb(cb_user, f, fr.data())) stop = true;
                    delivered = f + 1;
                }
            }
            if (s2->done != 0) ended = true;
            return ended || stop;
        };
        if (cb) {  // frame 0 (produced by the prefill iteration) is delivered before the decode batches
            FS_HIP(hipMemcpyAsync(slot_state(0), state(0), sizeof(SeqState), hipMemcpyDeviceToHost, st_));
            FS_HIP(hipMemcpy2DAsync(slot_codes(0), sizeof(uint32_t) * CHUNK, d_out_.as<uint32_t>(), sizeof(uint32_t) * out_cap_, sizeof(uint32_t), (size_t)C,
                                    hipMemcpyDeviceToHost, st_));
            FS_HIP(hipEventRecord(ev_batch_[0], st_));
            retire_batch(Batch{0, 1, 0});
        }
        {
            Batch prev{0, 0, 0};
            bool have_prev = false;
            int sl = cb ? 1 : 0;
            while (it < n_iter && !(ended || stop)) {
                const Batch cur = enqueue_batch(sl);
                sl ^= 1;
                if (have_prev) retire_batch(prev);  // batch b is examined while batch b + 1 runs
                prev = cur;
                have_prev = true;
            }
            if (have_prev && !(stop)) retire_batch(prev);
        }
        FS_HIP(hipEventRecord(ev_[2], st_));
        FS_HIP(hipMemcpyAsync(hs, state(0), sizeof(SeqState), hipMemcpyDeviceToHost, st_));
        FS_HIP(hipStreamSynchronize(st_));
        const size_t n = (size_t)hs->n_out;
        seq_len_[0] = hs->pos;
        float ms01 = 0, ms12 = 0;
        FS_HIP(hipEventElapsedTime(&ms01, ev_[0], ev_[1]));
        FS_HIP(hipEventElapsedTime(&ms12, ev_[1], ev_[2]));
        stats_.prefill_ms = ms01; stats_.decode_ms = ms12; stats_.frames = n; stats_.prompt_tokens = (uint64_t)L;
        if (!kev.empty()) {
            double ts = 0, tf = 0;
            for (size_t i = 0; i < kev.size(); i += 3) {
                float a = 0, b = 0;
                FS_HIP(hipEventElapsedTime(&a, kev[i], kev[i + 1]));
                FS_HIP(hipEventElapsedTime(&b, kev[i + 1], kev[i + 2]));
                ts += a; tf += b;
            }
            stats_.slow_kernel_us = ts * 1e3 / (double)(kev.size() / 3);
            stats_.fast_kernel_us = tf * 1e3 / (double)(kev.size() / 3);
            for (hipEvent_t e : kev) (void)hipEventDestroy(e);
        }
        if (use_persist_ && getenv("FISHRT_PERSIST_PROF")) {
            unsigned long long pr[24];
            FS_HIP(hipMemcpy(pr, d_ctl_.as<uint32_t>() + 16, sizeof(pr), hipMemcpyDeviceToHost));
            FS_HIP(hipMemset(d_ctl_.as<uint32_t>() + 16, 0, sizeof(pr)));
            const double f = 0.01 / std::max<double>(1.0, (double)stats_.graph_launches - (double)L + 1);  // us per frame
            fprintf(stderr, "persist prof (us/frame, workgroup 0; wait+work): preload %.1f  S1 %.1f+%.1f  S2 %.1f+%.1f  S3 %.1f+%.1f  S4 %.1f+%.1f  head %.1f+%.1f  decision %.1f+%.1f  tail %.1f | S3 split: gemv+reduce %.1f barrier %.1f epilogue %.1f\n",
                    pr[0] * f, pr[9] * f, pr[1] * f, pr[10] * f, pr[2] * f, pr[11] * f, pr[3] * f, pr[12] * f, pr[4] * f, pr[13] * f, pr[5] * f, pr[14] * f, pr[6] * f, pr[7] * f, pr[8] * f, pr[15] * f, pr[3] * f);
            fprintf(stderr, "persist boundaries (us/frame, workgroup 0): slow kernel's finish -> fast kernel's entry %.2f (%llu frames)  entry -> first stage timer (state, slow-token decision) %.2f  last timer -> finish %.2f\n",
                    pr[20] ? pr[17] * 0.01 / (double)pr[20] : 0.0, pr[20], pr[18] * f, pr[19] * f);
        }
        if (use_pslow_) {
            if (getenv("FISHRT_PERSIST_PROF")) {
                unsigned long long pr[24];
                FS_HIP(hipMemcpy(pr, d_sctl_.as<uint32_t>() + 16, sizeof(pr), hipMemcpyDeviceToHost));
                FS_HIP(hipMemset(d_sctl_.as<uint32_t>() + 16, 0, sizeof(pr)));
                const double f = 0.01 / std::max<double>(1.0, (double)stats_.graph_launches - (double)L + 1);
                fprintf(stderr, "slow persist prof (us/frame, workgroup %d; wait+work): S1 %.1f+%.1f  S2 %.1f+%.1f  S3 %.1f+%.1f  S4 %.1f+%.1f  S5 %.1f+%.1f  head %.1f+%.1f\n",
                        getenv("FISHRT_PERSIST_PROF_WG") ? atoi(getenv("FISHRT_PERSIST_PROF_WG")) : 0, pr[9] * f, pr[1] * f, pr[10] * f, pr[2] * f, pr[11] * f, pr[3] * f,
                        pr[12] * f, pr[4] * f, pr[13] * f, pr[5] * f, pr[14] * f, pr[6] * f);
                fprintf(stderr, "slow persist boundaries (us/frame): fast kernel's finish -> slow kernel's entry %.2f (%llu frames)  entry -> first stage timer (state, first loads) %.2f  last timer -> finish %.2f\n",
                        pr[20] ? pr[17] * 0.01 / (double)pr[20] : 0.0, pr[20], pr[18] * f, pr[19] * f);
            }
            uint32_t ctl[4] = {0, 0, 0, 0};
            FS_HIP(hipMemcpy(ctl, d_sctl_.p, sizeof(ctl), hipMemcpyDeviceToHost));
            if (ctl[1]) {
                FS_HIP(hipMemset(d_sctl_.as<uint32_t>() + 1, 0, 4));
                pslow_ok_ = persist_ok_ = false;  // this handle stays on the per-node graphs from now on: only this request is lost
                throw Error("persistent slow-transformer kernel: a grid-wide wait timed out (are all 256 CUs available to this process?); "
                            "the handle falls back to per-node launches for its next calls");
            }
        }
        if (use_persist_) {
            uint32_t ctl[4] = {0, 0, 0, 0};
            FS_HIP(hipMemcpy(ctl, d_ctl_.p, sizeof(ctl), hipMemcpyDeviceToHost));
            if (ctl[1] || ctl[2]) {
                FS_HIP(hipMemset(d_ctl_.as<uint32_t>() + 1, 0, 8));
                if (ctl[1]) pslow_ok_ = persist_ok_ = false;  // (see above)
                throw Error(ctl[1] ? "persistent fast-decoder kernel: a grid-wide wait timed out (are all 256 CUs available to this process?); "
                                     "the handle falls back to per-node launches for its next calls"
                                   : "persistent fast-decoder kernel launched with a sampling configuration it was not built for");
            }
        }
        if (clamped && !hs->done && !stop)
            throw Error("generation ran past max_seq_len without <|im_end|> (the reference fails at dual_ar.rs:623-624)");
        FS_REQUIRE(n <= cap, "codes_out capacity too small for the generated frames");
        if (codes_out) {
            std::vector<uint32_t> tmp((size_t)C * out_cap_);
            FS_HIP(hipMemcpy(tmp.data(), d_out_.p, sizeof(uint32_t) * C * out_cap_, hipMemcpyDeviceToHost));
            for (int c = 0; c < C; ++c) std::memcpy(codes_out + (size_t)c * cap, tmp.data() + (size_t)c * out_cap_, sizeof(uint32_t) * n);
        }
        if (n_frames) *n_frames = n;
        if (hidden_out) {
            const size_t rows = (size_t)hs->frame;  // iterations executed, the terminating <|im_end|> one included (single_batch.rs:264-266)
            FS_REQUIRE(rows <= hidden_cap, "hidden_out capacity too small for the generator iterations");
            FS_HIP(hipMemcpy(hidden_out, d_hidden_.p, sizeof(float) * rows * a_.dim, hipMemcpyDeviceToHost));
            if (n_hidden) *n_hidden = rows;
        }
    }

    // generate_static_batch (static_batch.rs:282-390).  bf16 / fp8 handles with n <= min(max_batch, kRows): the MFMA row path below
    // (rows = sequences).  Otherwise rows are independent sequences evaluated one after another on KV slot 0 (identical tokens under
    // greedy decoding).  The left padding with <|im_end|>/0 IS applied because the reference never masks it (dual_ar.rs:589-615);
    // repetition penalty is a no-op in the reference's batch path for Fish models (static_batch.rs:204-206 => mask stays 1).
    void generate_batch(const uint32_t* prompts, const int* lens, int n, int max_new_tokens, const fs_sampling& s, uint64_t seed,
                        uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, uint8_t* is_audio) override {
        FS_REQUIRE(n >= 1, "Must have at least one prompt");  // static_batch.rs:69-71
        FS_REQUIRE(!sess_active_, "the handle is in session mode (fs_lm_session_end first)");
        if (LmKernels<WT>::has_mfma_prefill() && n <= kRows && n <= B_ && a_.dim % 128 == 0 && a_.intermediate_size % 128 == 0 &&
            a_.num_codebooks <= 8 && !legacy_) {
            generate_batch_rows(prompts, lens, n, max_new_tokens, s, seed, flags, codes_out, cap, n_frames, is_audio);
            return;
        }
        generate_batch_sequential(prompts, lens, n, max_new_tokens, s, seed, flags, codes_out, cap, n_frames, is_audio);
    }

    // generate_static_batch on the MFMA row path: the B sequences are the rows of every GEMM (weights streamed once per
    // step for the whole batch), per-row paged KV, per-row on-device sampling, one captured graph per (B, chunk bucket).
    void generate_batch_rows(const uint32_t* prompts, const int* lens, int n, int max_new_tokens, const fs_sampling& s, uint64_t seed,
                             uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, uint8_t* is_audio) {
        use_device();
        require_loaded();
        const int C = a_.num_codebooks, C1 = C + 1, B = n;
        FS_REQUIRE(t_.has_semantic_end, "static batches need a Fish 1.5 / DualAR token layout (semantic range)");
        int Lmax = 0;
        for (int i = 0; i < n; ++i) { FS_REQUIRE(lens[i] >= 1, "empty prompt"); Lmax = std::max(Lmax, lens[i]); }
        if (Lmax > a_.max_seq_len) throw Error("prompt exceeds max_seq_len (dual_ar.rs:623-624)");
        long long n_iter = 1 + std::max<long long>(0, (long long)max_new_tokens - Lmax + 1);  // static_batch.rs:122,262-267
        bool clamped = false;
        const long long room = (long long)a_.max_seq_len - Lmax + 1;
        if (n_iter > room) { n_iter = room; clamped = true; }
        FS_REQUIRE(n_iter <= out_cap_, "generation longer than the output staging buffer");
        clear_slow();  // static_batch.rs:118-121
        clear_fast();
        ensure_prefill_buffers();
        ensure_batch_buffers();
        ensure_rows_capture(B);
        SampleCfg cfg = base_cfg();
        cfg.temp = (float)s.temp; cfg.top_p = (float)s.top_p; cfg.top_p64 = s.top_p;
        cfg.top_k = (int)std::min<uint64_t>(s.top_k, 1u << 30);
        cfg.rep_pen = 1.0f; cfg.ignore_eos = (flags & FS_GEN_IGNORE_EOS) ? 1 : 0;
        rows_par_ = rows_par_sampler_ok(s.temp, s.top_k, n_audio_, a_.codebook_size) && !getenv("FISHRT_ROWS_SAMPLER_1024");
        FS_HIP(hipMemcpyAsync(d_cfg_.p, &cfg, sizeof(cfg), hipMemcpyHostToDevice, st_));
        RngState rng = {};
        seed_key(seed, rng.key);  // BatchedLogitsProcessor::new(seed) (the reference passes 42, static_batch.rs:63)
        FS_HIP(hipMemcpyAsync(d_rng_.p, &rng, sizeof(rng), hipMemcpyHostToDevice, st_));
        stats_ = {};
        FS_HIP(hipEventRecord(ev_[0], st_));
        // left-pad with <|im_end|>/0 (static_batch.rs:68-111; the pad mask is built but never applied, dual_ar.rs:589-615)
        const size_t pstride = (size_t)C1 * Lmax;
        std::vector<uint32_t> padded(pstride * B);
        size_t off = 0;
        for (int b = 0; b < B; ++b) {
            const int L = lens[b], pad = Lmax - L;
            uint32_t* pp = padded.data() + pstride * b;
            for (int r = 0; r < C1; ++r) {
                for (int j = 0; j < pad; ++j) pp[(size_t)r * Lmax + j] = r == 0 ? t_.im_end_id : 0u;
                std::memcpy(&pp[(size_t)r * Lmax + pad], prompts + off + (size_t)r * L, sizeof(uint32_t) * L);
            }
            off += (size_t)C1 * L;
            validate_tokens(pp, pstride, 1, Lmax);
            ensure_capacity(b, Lmax + (int)n_iter - 1);
        }
        const int Lp = Lmax - 1;  // prompt tokens run through the slow transformer before the first frame
        const bool no_group = std::getenv("FISHRT_NO_GROUP_PREFILL") != nullptr;  // test hook: one sequence per pass
        if (Lp >= 1 && Lp <= kRowsCap && a_.head_dim == 64 && !no_group) {
            // group prefill: every prompt has Lmax columns and starts at position 0, so the first Lmax - 1 tokens of as many
            // sequences as fit the row buffers go through ONE pass (GEMM rows = sequences x tokens, flash attention per sequence)
            if (d_bprompt_.n < sizeof(uint32_t) * padded.size()) d_bprompt_.alloc(sizeof(uint32_t) * padded.size());
            FS_HIP(hipMemcpyAsync(d_bprompt_.p, padded.data(), sizeof(uint32_t) * padded.size(), hipMemcpyHostToDevice, st_));
            SeqState ss = {};
            ss.prompt_L = Lmax;
            FS_HIP(hipMemcpyAsync(state(0), &ss, sizeof(ss), hipMemcpyHostToDevice, st_));
            const int per_pass = std::max(1, kRowsCap / Lp);
            for (int b0 = 0; b0 < B; b0 += per_pass) {
                const int S = std::min(per_pass, B - b0), M = S * Lp;
                RowsCtx c = rows_ctx(state(0), /*pos_step=*/1, /*pt_stride=*/max_pages_);
                c.seq_rows = Lp;
                LmKernels<WT>::prefill_embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(),
                                             d_bprompt_.as<uint32_t>() + pstride * b0, state(0), M, d_pfx_.as<float>(), st_, Lp, pstride);
                for (int l = 0; l < a_.n_layer; ++l) LmKernels<WT>::rows_layer(d_, M, c, slow_[l], slow_kv(l, b0), l == 0, st_);
            }
            FS_HIP(hipStreamSynchronize(st_));
            for (int b = 0; b < B; ++b) seq_len_[b] = Lp;
        } else {
            for (int b = 0; b < B; ++b) {
                FS_HIP(hipMemcpyAsync(d_prompt_.p, padded.data() + pstride * b, sizeof(uint32_t) * pstride, hipMemcpyHostToDevice, st_));
                SeqState ss = {};
                ss.prompt_L = Lmax;
                FS_HIP(hipMemcpyAsync(state(b), &ss, sizeof(ss), hipMemcpyHostToDevice, st_));
                prefill_tokens(b, Lp, /*use_graph=*/false);
                FS_HIP(hipStreamSynchronize(st_));  // d_prompt_ is reused by the next row
                seq_len_[b] = Lp;
            }
        }
        // first-frame inputs: the rows of d_pfx_ are scratch during prefill, so the last prompt column of every row is
        // embedded only now (through the state's `cur` slots)
        for (int b = 0; b < B; ++b) {
            const int L = lens[b];
            SeqState ss = {};
            ss.pos = Lmax - 1; ss.prompt_L = Lmax; ss.step = Lmax - 1;
            const uint32_t* src = prompts;  // find row b's prompt start
            size_t o2 = 0;
            for (int j = 0; j < b; ++j) o2 += (size_t)C1 * lens[j];
            for (int r = 0; r < C1; ++r) ss.cur[r] = src[o2 + (size_t)r * L + (L - 1)];
            FS_HIP(hipMemcpyAsync(state(b), &ss, sizeof(ss), hipMemcpyHostToDevice, st_));
            LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(), nullptr, state(b),
                                 d_pfx_.as<float>() + (size_t)b * a_.dim, st_);
            FS_HIP(hipStreamSynchronize(st_));
        }
        auto launch_frame = [&](long long it_) {
            set_bucket(Lmax + (int)it_);
            hipGraphExec_t g = batch_graph(B);
            FS_HIP(hipGraphLaunch(g, st_));
        };
        launch_frame(0);
        FS_HIP(hipEventRecord(ev_[1], st_));
        std::vector<SeqState> hs(B);
        auto all_done = [&]() {
            FS_HIP(hipMemcpyAsync(hs.data(), state(0), sizeof(SeqState) * B, hipMemcpyDeviceToHost, st_));
            FS_HIP(hipStreamSynchronize(st_));
            for (int b = 0; b < B; ++b) if (!hs[b].done) return false;
            return true;
        };
        long long it = 1;
        while (it < n_iter) {
            const long long end = std::min<long long>(n_iter, it + 16);
            for (; it < end; ++it) launch_frame(it);
            if (it < n_iter && all_done()) break;  // prompt = None once every row is dead (:255-258)
        }
        FS_HIP(hipEventRecord(ev_[2], st_));
        const bool done_all = all_done();
        check_rows_xchg();
        float ms01 = 0, ms12 = 0;
        FS_HIP(hipEventElapsedTime(&ms01, ev_[0], ev_[1]));
        FS_HIP(hipEventElapsedTime(&ms12, ev_[1], ev_[2]));
        stats_.prefill_ms = ms01; stats_.decode_ms = ms12; stats_.prompt_tokens = (uint64_t)Lmax * B; stats_.graph_launches = (uint64_t)it;
        if (clamped && !done_all)
            throw Error("generation ran past max_seq_len without <|im_end|> on every row (the reference fails at dual_ar.rs:623-624)");
        std::vector<uint32_t> tmp((size_t)B * C * out_cap_);
        FS_HIP(hipMemcpy(tmp.data(), d_out_.p, sizeof(uint32_t) * tmp.size(), hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (int b = 0; b < B; ++b) {
            const size_t nb = (size_t)hs[b].n_out;
            FS_REQUIRE(nb <= cap, "codes_out capacity too small for the generated frames");
            for (int c = 0; c < C; ++c)
                std::memcpy(codes_out + ((size_t)b * C + c) * cap, tmp.data() + ((size_t)b * C + c) * out_cap_, sizeof(uint32_t) * nb);
            n_frames[b] = nb;
            total += nb;
            seq_len_[b] = hs[b].pos;
            // BatchPosition::is_audio (static_batch.rs:229): the first position is returned unconditionally and is not audio when its slow token
            // was <|im_end|> (the row samplers mark that in `step`); every later returned position belongs to a live row, i.e. is a semantic token
            if (is_audio)
                for (size_t f = 0; f < nb; ++f) is_audio[(size_t)b * cap + f] = (f == 0 && hs[b].step == -1) ? 0 : 1;
        }
        stats_.frames = total;
    }

    // ---- continuous batching: the rows of the static-batch step as independent request slots.  No reference counterpart (the reference
    // serialises requests behind one mutex, server/lib/state.rs:12-29, or runs lock-step static batches, static_batch.rs:282-390); a slot
    // behaves exactly like row 0 of a one-prompt generate_static_batch: no left padding (its own positions / KV pages), first frame
    // emitted unconditionally, BatchedLogitsProcessor sampling, no repetition penalty, 1 + max(0, max_new_tokens - L + 1) iterations.
    // Requests join between steps (their prompt is prefilled on the row path while the other slots wait) and leave when done.
    void session_begin(const fs_sampling& s, uint64_t seed, uint32_t flags) override {
        use_device();
        require_loaded();
        FS_REQUIRE(!sess_active_, "a session is already open on this handle");
        FS_REQUIRE(LmKernels<WT>::has_mfma_prefill() && B_ <= kRows && a_.dim % 128 == 0 && a_.intermediate_size % 128 == 0 &&
                       a_.num_codebooks <= 8 && !legacy_ && t_.has_semantic_end,
                   "sessions need the MFMA row path (bf16 / fp8 handle, Fish 1.5 token layout)");
        clear_slow();
        clear_fast();
        ensure_prefill_buffers();
        ensure_batch_buffers();
        // FS_SESSION_ROWS: the slots run on the request-row persistent kernels (lm_persist_rows.hip) and keep BATCH-1 semantics -- every slot is
        // its own generate_blocking (repetition penalty, its own LogitsProcessor stream seeded seed + its admission number) -- instead of the
        // static-batch sampler's.  Needs max_batch <= 8, a bf16 Fish-1.5 handle and a sampler setting the in-launch decisions cover.
        sess_rows_ = false;
        std::unique_lock<PersistLock> rows_lock;
        struct RowsGuard {  // anything thrown below leaves the handle out of row mode (the local lock releases itself)
            bool& flag; bool armed = true;
            ~RowsGuard() { if (armed) flag = false; }
        } rows_guard{sess_rows_};
        if (flags & FS_SESSION_ROWS) {
            const char* why = nullptr;
            if (!rows_predicate(B_, &s, 1, /*for_session=*/true, &why)) throw Error(std::string("FS_SESSION_ROWS needs ") + why);
            FS_REQUIRE(!free_pages_.empty(), "KV page pool exhausted");
            // the device's persistent-kernel lock is held in a LOCAL until nothing below can throw any more (a throw after taking it used to
            // leave sess_plock_ owning the mutex with sess_active_ false: session_end returned early and the device was locked out for good)
            rows_lock = acquire_persist(device_);
            FS_REQUIRE(rows_lock.owns_lock(), "another call on this device holds the persistent kernels");
            sess_R_ = B_;
            ensure_rows(sess_R_);
            sess_rows_ = true;
            sess_sampled_ = s.temp != 0.0;
            sess_seed_ = seed;
            sess_adds_ = 0;
        }
        ensure_rows_capture(sess_rows_ ? sess_R_ : B_);
        SampleCfg cfg = base_cfg();
        cfg.temp = (float)s.temp; cfg.top_p = (float)s.top_p; cfg.top_p64 = s.top_p;
        cfg.top_k = (int)std::min<uint64_t>(s.top_k, 1u << 30);
        cfg.rep_pen = sess_rows_ ? s.repetition_penalty : 1.0f; cfg.ignore_eos = (flags & FS_GEN_IGNORE_EOS) ? 1 : 0;
        cfg.session = sess_rows_ ? 0 : 1;
        if (sess_rows_) {
            std::vector<SampleCfg> cfgs(sess_R_, cfg);
            FS_HIP(hipMemcpyAsync(d_rcfg_.p, cfgs.data(), sizeof(SampleCfg) * sess_R_, hipMemcpyHostToDevice, st_));
            float* nullp = nullptr;
            FS_HIP(hipMemcpyAsync(d_hid_slot_.p, &nullp, sizeof(nullp), hipMemcpyHostToDevice, st_));
            FS_HIP(hipStreamSynchronize(st_));  // (cfgs is a local)
        }
        rows_par_ = rows_par_sampler_ok(s.temp, s.top_k, n_audio_, a_.codebook_size) && !getenv("FISHRT_ROWS_SAMPLER_1024");
        FS_HIP(hipMemcpyAsync(d_cfg_.p, &cfg, sizeof(cfg), hipMemcpyHostToDevice, st_));
        RngState rng = {};
        seed_key(seed, rng.key);
        FS_HIP(hipMemcpyAsync(d_rng_.p, &rng, sizeof(rng), hipMemcpyHostToDevice, st_));
        // empty slots: dead, frame 1 (so the frame-0 rule never fires), position 0 of a scratch page nobody reads
        FS_REQUIRE(!free_pages_.empty(), "KV page pool exhausted");
        sess_scratch_ = free_pages_.back();
        free_pages_.pop_back();
        sess_left_.assign(B_, -1);
        sess_pos_.assign(B_, 0);
        sess_hs_.assign(B_, SeqState{});
        sess_queue_.clear(); sess_flight_.clear();
        for (int b = 0; b < B_; ++b) park_slot(b);
        FS_HIP(hipMemsetAsync(d_pfx_.p, 0, sizeof(float) * (size_t)B_ * a_.dim, st_));
        FS_HIP(hipStreamSynchronize(st_));
        stats_ = {};
        rows_guard.armed = false;
        if (sess_rows_) sess_plock_ = std::move(rows_lock);
        sess_active_ = true;
    }
    void park_slot(int b) {
        SeqState ss = {};
        ss.done = sess_rows_ ? 2 : 1; ss.frame = 1;  // (rows mode: a terminated row is skipped by the row kernels altogether)
        sess_hs_[b] = ss;
        FS_HIP(hipMemcpyAsync(state(b), &sess_hs_[b], sizeof(SeqState), hipMemcpyHostToDevice, st_));
        FS_HIP(hipMemcpyAsync(d_page_table_.as<int>() + (size_t)b * max_pages_, &sess_scratch_, sizeof(int), hipMemcpyHostToDevice, st_));
        FS_HIP(hipStreamSynchronize(st_));
    }
    // Joining requests are prefilled on a second stream with their own row buffers, staging SeqState (index B_) and page-table rows
    // (B_ ..), so the decode steps of the live slots go on underneath; a joining slot stays parked (scratch page, frozen) until its prefill
    // has finished and activate_pending() -- between two steps -- hands it its page-table row, state and first input embedding.
    // session_add only QUEUES the request; flush_pending() (first thing in session_step, or when the row buffers are full) prefills
    // everything queued as GROUP passes: requests sorted by length, each pass = as many of them as fit the 2048 activation rows,
    // right-padded to the longest of the pass (rows = sequences x tokens, RowMap.seq_rows; causal attention: a real token never sees a
    // pad token, and the K/V a pad position writes into the slot's own pages lies beyond its length, where the first decode steps
    // overwrite it before anything reads it).  A burst of 32 requests is admitted in 2-3 passes instead of 32.
    int session_add(const uint32_t* prompt, int L, int max_new_tokens) override {
        use_device();
        FS_REQUIRE(sess_active_, "no open session");
        FS_REQUIRE(L >= 1, "empty prompt");
        if (L > a_.max_seq_len) throw Error("prompt exceeds max_seq_len (dual_ar.rs:623-624)");
        int b = -1;
        for (int i = 0; i < B_; ++i) if (sess_left_[i] == -1) { b = i; break; }
        if (b < 0) return -1;
        const int C1 = a_.num_codebooks + 1;
        validate_tokens(prompt, 0, 1, L);
        long long n_iter = 1 + std::max<long long>(0, (long long)max_new_tokens - L + 1);  // static_batch.rs:122,262-267
        n_iter = std::min<long long>(n_iter, (long long)a_.max_seq_len - L + 1);           // a slot stops at max_seq_len instead of erroring
        FS_REQUIRE(n_iter <= out_cap_, "generation longer than the output staging buffer");
        ensure_prefill2_buffers();
        {   // not enough free KV pages right now: like "all slots busy" (pages come back when slots are released), not an error
            const int need = (L + (int)n_iter - 1 + KV_PAGE - 1) / KV_PAGE;
            if ((int)free_pages_.size() < need - (int)seq_pages_[b].size()) return -1;
        }
        alloc_pages(b, L + (int)n_iter - 1);
        PendingAdd pa;
        pa.slot = b; pa.L = L; pa.n_iter = (int)n_iter; pa.order = sess_adds_++;
        pa.prompt.assign(prompt, prompt + (size_t)C1 * L);
        sess_queue_.push_back(std::move(pa));
        sess_left_[b] = -2;  // reserved: prefilling
        stats_.prompt_tokens += (uint64_t)L;
        return b;
    }
    // launch the prefill of the next group of queued requests (ONE pass on the prefill stream, nothing waited for); no-op while an
    // earlier group is still in flight -- the rest of the queue follows once that one has been activated
    void flush_pending() {
        if (sess_queue_.empty() || !sess_flight_.empty()) return;
        const int C1 = a_.num_codebooks + 1, C = a_.num_codebooks;
        std::stable_sort(sess_queue_.begin(), sess_queue_.end(), [](const PendingAdd& x, const PendingAdd& y) { return x.L > y.L; });
        const bool flash = a_.head_dim == 64;
        size_t i = 0;
        int S = 1;
        {
            const int Lp = sess_queue_[i].L - 1;  // the longest of this pass (sorted): tokens that run through the slow transformer
            if (Lp < 1) {                         // only single-token prompts are left: no pass, they go live at the next activation
                FS_HIP(hipEventRecord(ev_pf_, st_pf_));
                sess_flight_ = std::move(sess_queue_);
                sess_queue_.clear();
                return;
            }
            if (flash && Lp <= kRowsCap) {
                const int fit = std::max(1, std::min(kRowsCap / Lp, B_));
                // pad positions write K/V up to Lp: every member must own pages that far (bounded: they go back with the slot).  The extra
                // pages are counted CUMULATIVELY over the group -- checked per member, several short-budget members could each pass and
                // alloc_pages would then throw in the middle of the flush with the queue half consumed
                auto extra = [&](const PendingAdd& pa) {
                    const int want = (std::max(pa.L + pa.n_iter - 1, Lp) + KV_PAGE - 1) / KV_PAGE;
                    return std::max(0, want - (int)seq_pages_[pa.slot].size());
                };
                int need_sum = extra(sess_queue_[i]);
                while (i + S < sess_queue_.size() && S < fit && sess_queue_[i + S].L - 1 >= 1) {
                    const int need = extra(sess_queue_[i + S]);
                    if (need_sum + need > (int)free_pages_.size()) break;
                    need_sum += need;
                    ++S;
                }
                if (need_sum > (int)free_pages_.size()) S = 1;  // (the first member alone does not fit a group pass either: its own pages were reserved by session_add)
            }
            if (S == 1 || !flash) {  // one sequence, passes of <= kRowsCap rows
                const PendingAdd& pa = sess_queue_[i];
                const auto& pg = seq_pages_[pa.slot];
                FS_HIP(hipMemcpyAsync(d_page_table_.as<int>() + (size_t)B_ * max_pages_, pg.data(), sizeof(int) * pg.size(), hipMemcpyHostToDevice, st_pf_));
                if (d2_prompt_.n < sizeof(uint32_t) * (size_t)C1 * pa.L) d2_prompt_.alloc(sizeof(uint32_t) * (size_t)C1 * pa.L);
                FS_HIP(hipMemcpyAsync(d2_prompt_.p, pa.prompt.data(), sizeof(uint32_t) * (size_t)C1 * pa.L, hipMemcpyHostToDevice, st_pf_));
                sess_stage_ = SeqState{};
                sess_stage_.prompt_L = pa.L;
                FS_HIP(hipMemcpyAsync(state(B_), &sess_stage_, sizeof(SeqState), hipMemcpyHostToDevice, st_pf_));
                RowsCtx c = rows_ctx2(state(B_), /*pos_step=*/1, /*pt_stride=*/0);
                for (int done = 0; done < Lp;) {
                    const int M = std::min(flash ? kRowsCap : kPartRows, Lp - done);
                    c.nc_launch = chunk_bucket(done + M);
                    LmKernels<WT>::prefill_embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(),
                                                 d2_prompt_.as<uint32_t>(), state(B_), M, d2_pfx_.as<float>(), st_pf_);
                    for (int l = 0; l < a_.n_layer; ++l) LmKernels<WT>::rows_layer(d_, M, c, slow_[l], slow_kv(l, B_), l == 0, st_pf_);
                    LmKernels<WT>::rows_finish(d_, M, c, nullptr, st_pf_);
                    launch_advance_n(state(B_), M, st_pf_);
                    done += M;
                }
            } else {
                const size_t pstride = (size_t)C1 * Lp;
                std::vector<uint32_t>& padded = sess_stage_prompts_;  // (members: alive until the pass has been activated)
                std::vector<int>& rows = sess_stage_rows_;
                padded.assign(pstride * S, 0u);
                rows.assign((size_t)S * max_pages_, sess_scratch_);
                for (int s = 0; s < S; ++s) {
                    const PendingAdd& pa = sess_queue_[i + s];
                    alloc_pages(pa.slot, std::max(pa.L + pa.n_iter - 1, Lp));
                    const int n = pa.L - 1;
                    for (int r = 0; r < C1; ++r) std::memcpy(&padded[pstride * s + (size_t)r * Lp], &pa.prompt[(size_t)r * pa.L], sizeof(uint32_t) * n);
                    const auto& pg = seq_pages_[pa.slot];
                    std::copy(pg.begin(), pg.end(), rows.begin() + (size_t)s * max_pages_);
                }
                if (d2_prompt_.n < sizeof(uint32_t) * padded.size()) d2_prompt_.alloc(sizeof(uint32_t) * padded.size());
                FS_HIP(hipMemcpyAsync(d2_prompt_.p, padded.data(), sizeof(uint32_t) * padded.size(), hipMemcpyHostToDevice, st_pf_));
                FS_HIP(hipMemcpyAsync(d_page_table_.as<int>() + (size_t)B_ * max_pages_, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice, st_pf_));
                sess_stage_ = SeqState{};
                sess_stage_.prompt_L = Lp;
                FS_HIP(hipMemcpyAsync(state(B_), &sess_stage_, sizeof(SeqState), hipMemcpyHostToDevice, st_pf_));
                RowsCtx c = rows_ctx2(state(B_), /*pos_step=*/1, /*pt_stride=*/max_pages_);
                c.seq_rows = Lp;
                c.nc_launch = chunk_bucket(Lp);
                const int M = S * Lp;
                LmKernels<WT>::prefill_embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(), d2_prompt_.as<uint32_t>(), state(B_), M,
                                             d2_pfx_.as<float>(), st_pf_, Lp, pstride);
                for (int l = 0; l < a_.n_layer; ++l) LmKernels<WT>::rows_layer(d_, M, c, slow_[l], slow_kv(l, B_), l == 0, st_pf_);
            }
        }
        FS_HIP(hipEventRecord(ev_pf_, st_pf_));
        sess_flight_.assign(std::make_move_iterator(sess_queue_.begin()), std::make_move_iterator(sess_queue_.begin() + S));
        sess_queue_.erase(sess_queue_.begin(), sess_queue_.begin() + S);
    }
    // the requests whose prefill is in flight (if it has finished, or after waiting for it) become live slots; called between steps only
    void activate_pending(bool wait) {
        if (sess_flight_.empty()) return;
        if (!wait) {
            const hipError_t q = hipEventQuery(ev_pf_);
            if (q == hipErrorNotReady) return;
            FS_HIP(q);
        }
        FS_HIP(hipEventSynchronize(ev_pf_));
        const int C1 = a_.num_codebooks + 1;
        for (const PendingAdd& pa : sess_flight_) {
            const int b = pa.slot, L = pa.L, Lp = L - 1;
            const auto& pg = seq_pages_[b];
            FS_HIP(hipMemcpyAsync(d_page_table_.as<int>() + (size_t)b * max_pages_, pg.data(), sizeof(int) * pg.size(), hipMemcpyHostToDevice, st_));
            seq_len_[b] = Lp;
            SeqState ss = {};
            ss.pos = Lp; ss.prompt_L = L; ss.step = Lp;
            for (int r = 0; r < C1; ++r) ss.cur[r] = pa.prompt[(size_t)r * L + (L - 1)];
            sess_hs_[b] = ss;
            FS_HIP(hipMemcpyAsync(state(b), &sess_hs_[b], sizeof(SeqState), hipMemcpyHostToDevice, st_));
            if (sess_rows_) {  // the slot becomes a live row: first-frame input, iteration budget, fresh repetition-penalty window and sampler stream
                LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, a_.num_codebooks, a_.codebook_size, d_cfg_.as<SampleCfg>(), nullptr, state(b), x(b), st_);
                sess_budget_tmp_ = pa.n_iter;
                FS_HIP(hipMemcpyAsync(d_rbudget_.as<int>() + b, &sess_budget_tmp_, sizeof(int), hipMemcpyHostToDevice, st_));
                RngState rng = {};
                seed_key(sess_seed_ + (uint64_t)pa.order, rng.key);
                FS_HIP(hipMemcpyAsync(d_rrng_.as<RngState>() + b, &rng, sizeof(rng), hipMemcpyHostToDevice, st_));
                launch_reppen_reset(rows_rp(b), a_.num_codebooks, a_.codebook_size, st_);
                FS_HIP(hipStreamSynchronize(st_));  // (rng / budget are locals)
            } else
            LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, a_.num_codebooks, a_.codebook_size, d_cfg_.as<SampleCfg>(), nullptr, state(b),
                                 d_pfx_.as<float>() + (size_t)b * a_.dim, st_);
            sess_left_[b] = pa.n_iter;
            sess_pos_[b] = Lp;
        }
        FS_HIP(hipStreamSynchronize(st_));
        sess_flight_.clear();
    }
    void alloc_pages(int b, int n_tokens) {  // ensure_capacity without the upload: the row of a parked slot must keep pointing at the scratch page
        FS_REQUIRE(n_tokens <= a_.max_seq_len, "sequence longer than max_seq_len");
        const int need = (n_tokens + KV_PAGE - 1) / KV_PAGE;
        auto& pg = seq_pages_[b];
        FS_REQUIRE((int)free_pages_.size() >= need - (int)pg.size(), "KV page pool exhausted");
        while ((int)pg.size() < need) { pg.push_back(free_pages_.back()); free_pages_.pop_back(); }
    }
    void ensure_prefill2_buffers() {
        if (d2_pfx_.p) return;
        FS_HIP(hipStreamCreateWithFlags(&st_pf_, hipStreamNonBlocking));
        FS_HIP(hipEventCreateWithFlags(&ev_pf_, hipEventDisableTiming));
        d2_pfx_.alloc(d_pfx_.n); d2_pfq_.alloc(d_pfq_.n); d2_pfslab_.alloc(d_pfslab_.n); d2_pfa_.alloc(d_pfa_.n); d2_pfa2_.alloc(d_pfa2_.n);
        d2_pfss_.alloc(d_pfss_.n); d2_pfc_.alloc(d_pfc_.n); d2_pfpart_.alloc(d_pfpart_.n);
        for (DevBuf* bf : {&d2_pfx_, &d2_pfa_, &d2_pfa2_, &d2_pfss_, &d2_pfc_, &d2_pfpart_}) FS_HIP(hipMemsetAsync(bf->p, 0, bf->n, st_pf_));
        FS_HIP(hipStreamSynchronize(st_pf_));
    }
    RowsCtx rows_ctx2(const SeqState* st, int pos_step, int pt_stride) {
        RowsCtx c = rows_ctx(st, pos_step, pt_stride);
        c.X = d2_pfx_.as<float>(); c.Q = d2_pfq_.as<float>(); c.part = d2_pfpart_.as<float>(); c.P = d2_pfslab_.as<float>();
        c.A = d2_pfa_.as<uint16_t>(); c.C = d2_pfc_.as<uint16_t>(); c.A2 = d2_pfa2_.as<uint16_t>(); c.ss = d2_pfss_.as<float>();
        return c;
    }
    void session_step(int n_frames, int* n_active) override {
        use_device();
        FS_REQUIRE(sess_active_, "no open session");
        FS_HIP(hipEventRecord(ev_[1], st_));
        int launched = 0;
        while (launched < n_frames) {
            flush_pending();                   // queued requests start their (group) prefill on the second stream
            activate_pending(/*wait=*/false);  // a finished prefill joins here, between two replays of the step graph
            int chunk = n_frames - launched, longest = 0, live = 0;
            for (int b = 0; b < B_; ++b)
                if (sess_left_[b] > 0 && !sess_hs_[b].done) { chunk = std::min(chunk, sess_left_[b]); longest = std::max(longest, sess_pos_[b]); ++live; }
            if (!live && (!sess_flight_.empty() || !sess_queue_.empty())) {
                // nothing else to run (a burst into an idle session): admit everything queued, group pass after group pass, then step
                while (!sess_flight_.empty() || !sess_queue_.empty()) { flush_pending(); activate_pending(true); }
                continue;
            }
            if (!live) break;
            if (sess_rows_) {
                PersistFlagsGuard flags_guard{*this};
                use_persist_ = true; use_pslow_ = true; persist_sampled_ = sess_sampled_;
                // the launch group covers slots [0, top): the smallest instantiations that hold the highest live slot (an 8-slot session with two
                // live requests in slots 0, 1 pays the 2-row frame, 0.84 ms, not the 8-row one, 1.67 ms); slots are handed out lowest-first
                int top = 0;
                for (int b = 0; b < B_; ++b)
                    if (sess_left_[b] > 0 && !sess_hs_[b].done) top = b + 1;
                const int Rs = std::min(sess_R_, top <= 2 ? 2 : (top <= 4 ? 4 : 8));
                for (int i = 0; i < chunk; ++i) {
                    set_bucket(longest + i + 1);
                    launch_rows_slow(rows_slow_args(Rs), Rs, st_);
                    for (int r0 = 0; r0 < top; r0 += PR_FAST_ROWS) {
                        const int left = std::min(PR_FAST_ROWS, top - r0), Rf = left >= 3 ? 4 : left;
                        launch_rows_fast(rows_fast_args(r0, Rf), Rf, sess_sampled_, st_);
                    }
                }
            } else
            for (int i = 0; i < chunk; ++i) {
                set_bucket(longest + i + 1);
                FS_HIP(hipGraphLaunch(batch_graph(B_), st_));
            }
            launched += chunk;
            for (int b = 0; b < B_; ++b)
                if (sess_left_[b] > 0 && !sess_hs_[b].done) {
                    sess_left_[b] -= chunk; sess_pos_[b] += chunk;
                    if (sess_left_[b] == 0 && !sess_rows_) {  // iteration budget spent: the slot is finished whatever it sampled (rows mode: the kernels' own budget word terminates the row)
                        // Its last iteration left pos = L + n_iter - 1, one token PAST its page allocation (and == max_seq_len when the
                        // budget was clamped), and a frozen slot still rides the step graphs, whose QKV epilogue writes K/V at pos: park
                        // the write on the scratch page at position 0 (n_out and the codes stay) before the next replay can run.
                        static const int one = 1, zero = 0;
                        FS_HIP(hipMemcpyAsync(&state(b)->done, &one, sizeof(int), hipMemcpyHostToDevice, st_));
                        FS_HIP(hipMemcpyAsync(&state(b)->pos, &zero, sizeof(int), hipMemcpyHostToDevice, st_));
                        FS_HIP(hipMemcpyAsync(d_page_table_.as<int>() + (size_t)b * max_pages_, &sess_scratch_, sizeof(int), hipMemcpyHostToDevice, st_));
                    }
                }
            // (a slot that sampled <|im_end|> inside the chunk froze itself on the device; the host learns it below)
            FS_HIP(hipMemcpyAsync(sess_hs_.data(), state(0), sizeof(SeqState) * B_, hipMemcpyDeviceToHost, st_));
            FS_HIP(hipStreamSynchronize(st_));
            // a spin timeout inside the chunk (the joining prefill shares the CUs the persistent launches need co-resident) means the slots'
            // codes are garbage: raise instead of handing them out; the caller ends the session (the handle is off the row path afterwards)
            if (sess_rows_) rows_check_ctl(/*include_b1_fast=*/false);
            else check_rows_xchg();
        }
        FS_HIP(hipEventRecord(ev_[2], st_));
        FS_HIP(hipEventSynchronize(ev_[2]));
        float ms = 0;
        FS_HIP(hipEventElapsedTime(&ms, ev_[1], ev_[2]));
        stats_.decode_ms += ms;
        stats_.graph_launches += (uint64_t)launched;
        int act = 0;
        uint64_t frames = 0;
        for (int b = 0; b < B_; ++b) {
            if ((sess_left_[b] > 0 && !sess_hs_[b].done) || sess_left_[b] == -2) ++act;  // (-2: still prefilling)
            if (sess_left_[b] >= 0) frames += (uint64_t)sess_hs_[b].n_out;
        }
        stats_.frames = sess_released_frames_ + frames;
        if (n_active) *n_active = act;
    }
    void session_poll(int slot, uint32_t* codes_out, size_t cap, size_t* n_frames, int* done) override {
        use_device();
        FS_REQUIRE(sess_active_ && slot >= 0 && slot < B_ && sess_left_[slot] != -1, "not a live session slot");
        const int C = a_.num_codebooks;
        const size_t nb = sess_left_[slot] == -2 ? 0 : (size_t)sess_hs_[slot].n_out;  // (-2: still prefilling)
        if (codes_out) {
            FS_REQUIRE(nb <= cap, "codes_out capacity too small for the generated frames");
            for (int c = 0; c < C; ++c)
                FS_HIP(hipMemcpyAsync(codes_out + (size_t)c * cap, d_out_.as<uint32_t>() + ((size_t)slot * C + c) * out_cap_, sizeof(uint32_t) * nb,
                                      hipMemcpyDeviceToHost, st_));
            FS_HIP(hipStreamSynchronize(st_));
        }
        if (n_frames) *n_frames = nb;
        if (done) *done = (sess_left_[slot] != -2 && (sess_hs_[slot].done != 0 || sess_left_[slot] == 0)) ? 1 : 0;
    }
    void session_release(int slot) override {
        use_device();
        FS_REQUIRE(sess_active_ && slot >= 0 && slot < B_ && sess_left_[slot] != -1, "not a live session slot");
        while (sess_left_[slot] == -2) { flush_pending(); activate_pending(true); }  // (its prefill is queued or in flight: let it finish first)
        sess_released_frames_ += (uint64_t)sess_hs_[slot].n_out;
        truncate(slot, 0);
        park_slot(slot);
        sess_left_[slot] = -1;
    }
    void session_end() override {
        if (!sess_active_) return;
        use_device();
        if (!sess_flight_.empty()) (void)hipEventSynchronize(ev_pf_);
        sess_flight_.clear(); sess_queue_.clear();
        for (int b = 0; b < B_; ++b) truncate(b, 0);
        free_pages_.push_back(sess_scratch_);
        SampleCfg cfg = base_cfg();
        FS_HIP(hipMemcpyAsync(d_cfg_.p, &cfg, sizeof(cfg), hipMemcpyHostToDevice, st_));
        FS_HIP(hipStreamSynchronize(st_));
        sess_active_ = false;
        sess_released_frames_ = 0;
        if (sess_rows_) {
            uint32_t ctl[4];
            for (DevBuf* cb : {&d_rctl_s_, &d_rctl_f_}) {
                FS_HIP(hipMemcpy(ctl, cb->p, sizeof(ctl), hipMemcpyDeviceToHost));
                if (ctl[1] || ctl[2]) FS_HIP(hipMemset((uint32_t*)cb->p + 1, 0, 8));
            }
            sess_rows_ = false;
        }
        if (sess_plock_.owns_lock()) sess_plock_.unlock();
    }

    void generate_batch_sequential(const uint32_t* prompts, const int* lens, int n, int max_new_tokens, const fs_sampling& s, uint64_t seed,
                                   uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, uint8_t* is_audio) {
        if (legacy_) throw Error("generate_static_batch samples the slow token over the full vocabulary for Fish <= 1.4 (static_batch.rs:132-141); not implemented");
        const int C1 = a_.num_codebooks + 1;
        int Lmax = 0;
        for (int i = 0; i < n; ++i) { FS_REQUIRE(lens[i] >= 1, "empty prompt"); Lmax = std::max(Lmax, lens[i]); }
        fs_sampling sb = s;
        sb.repetition_penalty = 1.0f;
        size_t off = 0;
        std::vector<uint32_t> padded((size_t)C1 * Lmax);
        for (int i = 0; i < n; ++i) {
            const int L = lens[i], pad = Lmax - L;
            for (int r = 0; r < C1; ++r) {
                for (int j = 0; j < pad; ++j) padded[(size_t)r * Lmax + j] = r == 0 ? t_.im_end_id : 0u;
                std::memcpy(&padded[(size_t)r * Lmax + pad], prompts + off + (size_t)r * L, sizeof(uint32_t) * L);
            }
            off += (size_t)C1 * L;
            clear_slow();  // static_batch.rs:118-121
            // BatchedLogitsProcessor semantics on the single-sequence kernels (sampling/mod.rs:77-109): temp <= 1e-7 -> device argmax
            // (FIRST maximal index), else the child StdRng of (sample() call, row i) seeded from the master's u64 number call * n + i --
            // row i generated on its own draws exactly what it draws in lock-step
            batch_rows_ = n; batch_row_ = i;
            try {
                generate(padded.data(), Lmax, max_new_tokens, sb, seed, flags, codes_out + (size_t)i * a_.num_codebooks * cap, cap, &n_frames[i],
                         nullptr, nullptr, nullptr, 0, nullptr);
            } catch (...) { batch_rows_ = 0; batch_row_ = 0; throw; }
            batch_rows_ = 0; batch_row_ = 0;
            if (is_audio) {  // (one iteration executed and its slow token was <|im_end|>: the unconditional first position is not audio)
                const SeqState* hs = reinterpret_cast<const SeqState*>(h_pin_);
                const bool first_not_audio = hs->frame == 1 && hs->cur[0] == t_.im_end_id;
                for (size_t f = 0; f < n_frames[i]; ++f) is_audio[(size_t)i * cap + f] = (f == 0 && first_not_audio) ? 0 : 1;
            }
        }
    }


    // ---- R concurrent batch-1 requests on ONE device (lm_persist_rows.hip; no reference counterpart beyond the lock-step static batch,
    // generate/static_batch.rs:117-274).  Row i is generate_blocking(prompt_i, max_new_tokens_i, sampling_i) on its own KV slot, sampler,
    // repetition-penalty state and RNG stream (single_batch.rs:76-214) -- the same tokens as its own fs_lm_generate call on a cleared
    // cache (other summation order inside the kernels: greedy tokens can differ at near-ties only); every decode frame runs the slow
    // transformer of all rows as ONE persistent launch (weights streamed once for all rows) and the fast decoder as one launch per
    // group of <= 4 rows.  Handles that cannot take that path (f32 / fp8, Fish <= 1.4, sampler settings outside the in-launch sampler,
    // another call holding the device's persistent kernels) run the requests one after the other.
    void generate_multi(const uint32_t* prompts, const int* lens, int n, const int* max_new_tokens, const fs_sampling* samplings,
                        const uint64_t* seeds, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames) override {
        use_device();
        require_loaded();
        FS_REQUIRE(!sess_active_, "the handle is in session mode (fs_lm_session_end first)");
        FS_REQUIRE(n >= 1, "Must have at least one prompt");
        const int C = a_.num_codebooks, C1 = C + 1;
        std::vector<size_t> poff(n);
        {
            size_t off = 0;
            for (int i = 0; i < n; ++i) { FS_REQUIRE(lens[i] >= 1, "empty prompt"); FS_REQUIRE(max_new_tokens[i] >= 0, "negative max_new_tokens"); poff[i] = off; off += (size_t)C1 * lens[i]; }
        }
        bool rows_ok = rows_supported(n, samplings) && !(flags & FS_GEN_NO_PERSIST);
        const bool rows_sampled = samplings[0].temp != 0.0;
        std::unique_lock<PersistLock> plock;
        if (rows_ok) { plock = acquire_persist(device_); rows_ok = plock.owns_lock(); }
        if (!rows_ok) {
            if (plock.owns_lock()) plock.unlock();
            double pf = 0, dc = 0; uint64_t fr = 0, pt = 0, gl = 0;
            for (int i = 0; i < n; ++i) {
                clear_slow();
                generate(prompts + poff[i], lens[i], max_new_tokens[i], samplings[i], seeds[i], flags, codes_out + (size_t)i * C * cap, cap, &n_frames[i],
                         nullptr, nullptr, nullptr, 0, nullptr);
                pf += stats_.prefill_ms; dc += stats_.decode_ms; fr += stats_.frames; pt += stats_.prompt_tokens; gl += stats_.graph_launches;
            }
            stats_.prefill_ms = pf; stats_.decode_ms = dc; stats_.frames = fr; stats_.prompt_tokens = pt; stats_.graph_launches = gl;
            return;
        }
        const int R = n <= 2 ? 2 : (n <= 4 ? 4 : 8);
        ensure_rows(R);
        clear_slow();
        clear_fast();
        std::vector<long long> n_iter(n);
        long long max_iter = 0;
        for (int i = 0; i < n; ++i) {
            validate_tokens(prompts + poff[i], 0, 1, lens[i]);
            if (lens[i] > a_.max_seq_len) throw Error("prompt exceeds max_seq_len (dual_ar.rs:623-624)");
            n_iter[i] = 1 + std::max<long long>(0, (long long)max_new_tokens[i] - lens[i] + 1);  // single_batch.rs:61,77,193-197
            n_iter[i] = std::min<long long>(n_iter[i], (long long)a_.max_seq_len - lens[i] + 1);   // (a row stops at max_seq_len)
            FS_REQUIRE(n_iter[i] <= out_cap_, "generation longer than the output staging buffer");
            max_iter = std::max(max_iter, n_iter[i]);
            ensure_capacity(i, lens[i] + (int)n_iter[i] - 1);
        }
        // per-row device state: sampling configuration, iteration budget, generator state (rows >= n: terminated)
        std::vector<SampleCfg> cfgs(R);
        std::vector<int> budget(R, 0);
        for (int i = 0; i < R; ++i) {
            SampleCfg cfg = base_cfg();
            if (i < n) {
                const fs_sampling& s = samplings[i];
                cfg.temp = (float)s.temp; cfg.top_p = (float)s.top_p; cfg.top_p64 = s.top_p;
                cfg.top_k = (int)std::min<uint64_t>(s.top_k, 1u << 30);
                cfg.rep_pen = s.repetition_penalty; cfg.ignore_eos = (flags & FS_GEN_IGNORE_EOS) ? 1 : 0;
                budget[i] = (int)n_iter[i];
            }
            cfgs[i] = cfg;
        }
        FS_HIP(hipMemcpyAsync(d_rcfg_.p, cfgs.data(), sizeof(SampleCfg) * R, hipMemcpyHostToDevice, st_));
        FS_HIP(hipMemcpyAsync(d_rbudget_.p, budget.data(), sizeof(int) * R, hipMemcpyHostToDevice, st_));
        float* nullp = nullptr;
        FS_HIP(hipMemcpyAsync(d_hid_slot_.p, &nullp, sizeof(nullp), hipMemcpyHostToDevice, st_));
        ensure_rows_capture(R);
        PersistFlagsGuard flags_guard{*this};
        stats_ = {};
        FS_HIP(hipEventRecord(ev_[0], st_));
        for (int i = 0; i < R; ++i) {
            SeqState ss = {};
            if (i >= n) { ss.done = 2; FS_HIP(hipMemcpyAsync(state(i), &ss, sizeof(ss), hipMemcpyHostToDevice, st_)); continue; }
            const int L = lens[i];
            ss.prompt_L = L;
            FS_HIP(hipMemcpyAsync(state(i), &ss, sizeof(ss), hipMemcpyHostToDevice, st_));
            FS_HIP(hipMemcpyAsync(d_prompt_.p, prompts + poff[i], sizeof(uint32_t) * C1 * L, hipMemcpyHostToDevice, st_));
            FS_HIP(hipMemcpyAsync(d_cfg_.p, &cfgs[i], sizeof(SampleCfg), hipMemcpyHostToDevice, st_));  // (embed reads the semantic range from d_cfg_)
            RngState rng = {};
            seed_key(seeds[i], rng.key);
            FS_HIP(hipMemcpyAsync(d_rrng_.as<RngState>() + i, &rng, sizeof(rng), hipMemcpyHostToDevice, st_));
            launch_reppen_reset(rows_rp(i), C, a_.codebook_size, st_);
            prefill_tokens(i, L - 1, /*use_graph=*/false);
            LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(), d_prompt_.as<uint32_t>(), state(i), x(i), st_);
            FS_HIP(hipStreamSynchronize(st_));  // d_prompt_ / d_cfg_ are reused by the next row
            seq_len_[i] = L - 1;
            stats_.prompt_tokens += (uint64_t)L;
        }
        use_persist_ = true; use_pslow_ = true; persist_sampled_ = rows_sampled;
        std::vector<SeqState> hs(R);
        auto all_done = [&]() {
            FS_HIP(hipMemcpyAsync(hs.data(), state(0), sizeof(SeqState) * R, hipMemcpyDeviceToHost, st_));
            FS_HIP(hipStreamSynchronize(st_));
            for (int i = 0; i < n; ++i) if (hs[i].done != 2) return false;
            return true;
        };
        int maxL = 0;
        for (int i = 0; i < n; ++i) maxL = std::max(maxL, lens[i]);
        const bool rows_fast = !getenv("FISHRT_ROWS_FAST_SINGLE");
        auto launch_frame = [&](long long it_) {
            set_bucket(maxL + (int)it_);
            RowsSlowArgs S = rows_slow_args(R);
            launch_rows_slow(S, R, st_);
            if (rows_fast) {
                for (int r0 = 0; r0 < n; r0 += PR_FAST_ROWS) {
                    // the last group takes the smallest instantiation that holds its requests (n = 5: 4 + 1 rows, n = 6: 4 + 2)
                    const int left = std::min(PR_FAST_ROWS, n - r0), Rf = left >= 3 ? 4 : left;
                    launch_rows_fast(rows_fast_args(r0, Rf), Rf, rows_sampled, st_);
                }
            } else {
                static const int two = 2;
                for (int i = 0; i < n; ++i) {
                    if (it_ >= n_iter[i]) { if (it_ == n_iter[i]) FS_HIP(hipMemcpyAsync(&state(i)->done, &two, sizeof(int), hipMemcpyHostToDevice, st_)); continue; }
                    launch_fast_persist(persist_args(i), rows_sampled, st_);
                }
            }
        };
        launch_frame(0);
        FS_HIP(hipEventRecord(ev_[1], st_));
        long long it = 1;
        // (direct launches: 8 frames of these launches in ONE graph were measured at 940.9 vs 937.6 us per 4-row frame, 1575.5 vs 1576.7 at R = 8 --
        // back-to-back launches on a stream already pay what a graph edge does; the 7 us the batch-1 loop saved belong to hipGraphLaunch)
        while (it < max_iter) {
            const long long end = std::min<long long>(max_iter, it + 32);
            for (; it < end; ++it) launch_frame(it);
            if (it < max_iter && all_done()) break;
        }
        if (!rows_fast) {  // rows whose budget ended exactly at max_iter still need their flag
            static const int two = 2;
            for (int i = 0; i < n; ++i) FS_HIP(hipMemcpyAsync(&state(i)->done, &two, sizeof(int), hipMemcpyHostToDevice, st_));
        }
        FS_HIP(hipEventRecord(ev_[2], st_));
        FS_HIP(hipMemcpyAsync(hs.data(), state(0), sizeof(SeqState) * R, hipMemcpyDeviceToHost, st_));
        FS_HIP(hipStreamSynchronize(st_));
        use_persist_ = use_pslow_ = false;
        float ms01 = 0, ms12 = 0;
        FS_HIP(hipEventElapsedTime(&ms01, ev_[0], ev_[1]));
        FS_HIP(hipEventElapsedTime(&ms12, ev_[1], ev_[2]));
        stats_.prefill_ms = ms01; stats_.decode_ms = ms12; stats_.graph_launches = (uint64_t)it;
        stats_.kernels_per_frame = (uint64_t)(1 + (rows_fast ? (n + PR_FAST_ROWS - 1) / PR_FAST_ROWS : n));
        if (getenv("FISHRT_PERSIST_PROF")) {
            unsigned long long pr[16];
            const double f = 0.01 / std::max<double>(1.0, (double)it);  // us per frame (wall_clock64 ticks are 10 ns)
            FS_HIP(hipMemcpy(pr, d_rctl_s_.as<uint32_t>() + 16, sizeof(pr), hipMemcpyDeviceToHost));
            FS_HIP(hipMemset(d_rctl_s_.as<uint32_t>() + 16, 0, sizeof(pr)));
            fprintf(stderr, "rows slow prof R=%d (us/frame, workgroup 0): S1 %.1f  S2 %.1f  S3 sweep %.1f + %.1f  S4 sweep %.1f gemm %.1f barrier %.1f publish %.1f  S5 sweep+gemm %.1f barrier %.1f publish %.1f  head %.1f\n",
                    R, pr[1] * f, pr[2] * f, pr[8] * f, pr[3] * f, pr[9] * f, pr[10] * f, pr[11] * f, pr[4] * f, pr[12] * f, pr[13] * f, pr[5] * f, pr[6] * f);
            FS_HIP(hipMemcpy(pr, d_rctl_f_.as<uint32_t>() + 16, sizeof(pr), hipMemcpyDeviceToHost));
            FS_HIP(hipMemset(d_rctl_f_.as<uint32_t>() + 16, 0, sizeof(pr)));
            fprintf(stderr, "rows fast prof (us/frame summed over the fast launches, workgroup 0; wait+work): preload %.1f  S1 %.1f+%.1f  S2 %.1f+%.1f  S3 %.1f+%.1f  S4 %.1f+%.1f  head %.1f+%.1f  decision %.1f+%.1f (rows' logits -> argmax / draw %.1f, pick -> next input %.1f)  tail %.1f\n",
                    pr[0] * f, pr[9] * f, pr[1] * f, pr[10] * f, pr[2] * f, pr[11] * f, (pr[3] + pr[8]) * f, pr[12] * f, pr[4] * f, pr[13] * f, pr[5] * f, pr[14] * f, (pr[6] + pr[15]) * f, pr[15] * f, pr[6] * f, pr[7] * f);
        }
        rows_check_ctl(/*include_b1_fast=*/!rows_fast);
        std::vector<uint32_t> tmp((size_t)n * C * out_cap_);
        FS_HIP(hipMemcpy(tmp.data(), d_out_.p, sizeof(uint32_t) * tmp.size(), hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (int i = 0; i < n; ++i) {
            const size_t nb = (size_t)hs[i].n_out;
            FS_REQUIRE(nb <= cap, "codes_out capacity too small for the generated frames");
            for (int c = 0; c < C; ++c)
                std::memcpy(codes_out + ((size_t)i * C + c) * cap, tmp.data() + ((size_t)i * C + c) * out_cap_, sizeof(uint32_t) * nb);
            n_frames[i] = nb;
            total += nb;
            seq_len_[i] = hs[i].pos;
        }
        stats_.frames = total;
    }

    // fs_lm_rows_supported: would fs_lm_generate_multi serve these n requests on the request-row kernels (one persistent launch group per
    // frame) rather than one after the other?  Says nothing about whether another call holds the device's persistent kernels right now.
    // ONE predicate for "these requests run on the request-row kernels" (the capability query, fs_lm_generate_multi and
    // fs_lm_session_begin(FS_SESSION_ROWS) used to carry three hand-written copies that had drifted apart).  ns = 1: one sampler setting for
    // all n rows (sessions); ns = n: one per request.  for_session: a row session's slots additionally need the Fish 1.5 token layout (the
    // legacy 2-way slow draw is per-call state of generate_multi, not of a slot) and exactly 2, 4 or 8 slots.  why (nullable): what failed.
    bool rows_predicate(int n, const fs_sampling* ss, int ns, bool for_session, const char** why = nullptr) const {
        const char* w = nullptr;
        if (!(n >= 2 && n <= PR_MAX_ROWS && n <= B_)) w = "2 <= requests <= min(8, max_batch)";
        else if (!(loaded_ && pslow_ok_ && persist_ok_ && n_audio_ <= 2048)) w = "a loaded bf16 / fp8 handle with the Fish geometry on a 256-CU device";
        else if (getenv("FISHRT_NO_ROWS")) w = "FISHRT_NO_ROWS is set";
        else if (!rows_kv_span_ok(n <= 2 ? 2 : (n <= 4 ? 4 : 8))) w = "max_seq_len too long for the row kernels' attention slices at this row count";
        else if (for_session && legacy_) w = "row SESSIONS need the Fish 1.5 token layout (fs_lm_generate_multi serves Fish <= 1.4 handles)";
        else if (for_session && !(n == 2 || n == 4 || n == 8)) w = "row sessions need max_batch 2, 4 or 8";
        else {
            // one instantiation per launch: every request greedy, or every request within the in-launch sampler (temp > 0, 0 < top_k <= 256)
            const bool sampled = ss[0].temp != 0.0;
            for (int i = 0; i < ns && !w; ++i) {
                const int tk = (int)std::min<uint64_t>(ss[i].top_k, 1u << 30);
                if ((ss[i].temp != 0.0) != sampled) w = "every request greedy or every request sampled";
                else if (sampled && !fast_persist_samples((float)ss[i].temp, tk, a_.codebook_size)) w = "sampled requests with 0 < top_k <= 256";
            }
        }
        if (why) *why = w;
        return w == nullptr;
    }
    bool rows_supported(int n, const fs_sampling* samplings) override { return rows_predicate(n, samplings, n, /*for_session=*/false); }

  private:
    void use_device() { FS_HIP(hipSetDevice(device_)); }

    // k_slow_rows stages the page ids of ONE attention slice in 160 LDS slots (lm_persist_rows.hip `s_pages`): a slice is at most
    // ceil(max_seq_len / (16 / R)) tokens (fewer slices only while the cache is shorter than that many 128-token chunks)
    bool rows_kv_span_ok(int R) const {
        const int nslm = std::max(1, 16 / std::max(1, R));
        return ((a_.max_seq_len + nslm - 1) / nslm) / KV_PAGE + 2 <= 160;
    }
    // the request-row kernels' control words: [1] = a grid-wide wait hit its spin bound, [2] = launched with a sampler configuration the
    // instantiation was not built for.  Either way the codes of this call are garbage: reset the words, take the handle off the
    // persistent kernels for its next calls (like generate() does) and raise.
    void rows_check_ctl(bool include_b1_fast) {
        bool timeout = false, bad_cfg = false;
        for (DevBuf* cb : {&d_rctl_s_, &d_rctl_f_, include_b1_fast ? &d_ctl_ : (DevBuf*)nullptr}) {
            if (!cb || !cb->p) continue;
            uint32_t ctl[4] = {0, 0, 0, 0};
            FS_HIP(hipMemcpy(ctl, cb->p, sizeof(ctl), hipMemcpyDeviceToHost));
            if (ctl[1] || ctl[2]) FS_HIP(hipMemset((uint32_t*)cb->p + 1, 0, 8));
            timeout |= ctl[1] != 0; bad_cfg |= ctl[2] != 0;
        }
        if (timeout) {
            pslow_ok_ = persist_ok_ = false;
            throw Error("request-row persistent kernels: a grid-wide wait timed out (are all 256 CUs available to this process?); "
                        "the handle falls back to per-node launches for its next calls");
        }
        if (bad_cfg) throw Error("request-row persistent kernels launched with a sampling configuration they were not built for");
    }
    // use_persist_ / use_pslow_ / persist_sampled_ select graphs and sampler folding for generate(); a row-path call sets them for its own
    // launches and must leave them cleared however it ends
    struct PersistFlagsGuard {
        LM& lm;
        ~PersistFlagsGuard() { lm.use_persist_ = lm.use_pslow_ = false; }
    };
    void require_loaded() { FS_REQUIRE(loaded_, "weights not loaded: call fs_lm_load_safetensors or fs_lm_load_synthetic first"); }

    SampleCfg base_cfg() const {
        SampleCfg c = {};
        c.temp = 0.f; c.top_p = 1.f; c.top_k = 0; c.rep_pen = 1.f; c.ignore_eos = 0;
        c.im_end_id = t_.im_end_id;
        c.audio_base = legacy_ ? t_.im_end_id : t_.semantic_start_id - 1;
        c.sem_lo = t_.semantic_start_id;
        c.sem_hi = t_.has_semantic_end ? t_.semantic_end_id : t_.semantic_start_id;  // dual_ar.rs:554-559
        c.legacy = legacy_ ? 1 : 0;
        c.pad_id = t_.pad_id;
        return c;
    }

    void validate_tokens(const uint32_t* toks, size_t, int B, int L) {
        const int C1 = a_.num_codebooks + 1;
        for (int b = 0; b < B; ++b)
            for (int r = 0; r < C1; ++r)
                for (int l = 0; l < L; ++l) {
                    const uint32_t v = toks[((size_t)b * C1 + r) * L + l];
                    if (r == 0) { if (v >= (uint32_t)a_.vocab_size) throw Error("token id out of vocabulary (index_select out of range)"); }
                    else if (v >= (uint32_t)a_.codebook_size) throw Error("codebook id out of range (index_select out of range)");
                }
    }

    // ---- tensor plan: names/shapes of the reference loader (dual_ar.rs:125-156,219-223,415-419,466-511)
    void plan_tensors() {
        size_t off = 0;
        auto align = [&](size_t v) { return (v + 255) & ~(size_t)255; };
        auto mat = [&](const std::string& name, int64_t rows, int64_t cols, int mul, int roff, std::pair<size_t, size_t> at) {
            tensors_.push_back({name, rows, cols, false, false, at.first, at.second, mul, roff, 0.f, 0.02});  // initializer_range (dual_ar.rs:93)
        };
        auto emb = [&](const std::string& name, int64_t rows, int64_t cols) {
            size_t at = off;
            off = align(off + (size_t)rows * cols * sizeof(KT));
            tensors_.push_back({name, rows, cols, false, true, at, 0, 1, 0, 0.f, 0.02});
            return at;
        };
        // matrix slab (+ per-row f32 scale slab for fp8 weights): {weights offset, scales offset}
        auto slab = [&](int64_t rows, int64_t cols) {
            std::pair<size_t, size_t> at{off, 0};
            off = align(off + (size_t)rows * cols * sizeof(WT));
            if (kFp8) { at.second = off; off = align(off + (size_t)rows * sizeof(float)); }
            return at;
        };
        auto vec = [&](const std::string& name, int64_t n) {
            size_t at = off;
            off = align(off + (size_t)n * sizeof(float));
            tensors_.push_back({name, 1, n, true, false, at, 0, 1, 0, 1.0f, 0.1});
            return at;
        };
        const int64_t D = a_.dim, I = a_.intermediate_size, V = a_.vocab_size;
        const int64_t QKV = (int64_t)(a_.n_head + 2 * a_.n_local_heads) * a_.head_dim;
        o_tok_emb_ = emb("embeddings.weight", V, D);
        o_cb_emb_ = emb("codebook_embeddings.weight", (int64_t)a_.codebook_size * a_.num_codebooks, D);
        auto blocks = [&](int n, const std::string& pre, std::vector<std::array<size_t, 10>>& offs) {
            for (int l = 0; l < n; ++l) {
                const std::string p = pre + std::to_string(l) + ".";
                std::array<size_t, 10> o;
                auto s0 = slab(QKV, D); mat(p + "attention.wqkv.weight", QKV, D, 1, 0, s0);
                auto s1 = slab(D, D); mat(p + "attention.wo.weight", D, D, 1, 0, s1);
                auto s2 = slab(2 * I, D);
                mat(p + "feed_forward.w1.weight", I, D, 2, 0, s2);
                mat(p + "feed_forward.w3.weight", I, D, 2, 1, s2);
                auto s3 = slab(D, I); mat(p + "feed_forward.w2.weight", D, I, 1, 0, s3);
                o[0] = s0.first; o[1] = s1.first; o[2] = s2.first; o[3] = s3.first;
                o[4] = vec(p + "ffn_norm.weight", D);
                o[5] = vec(p + "attention_norm.weight", D);
                o[6] = s0.second; o[7] = s1.second; o[8] = s2.second; o[9] = s3.second;
                offs.push_back(o);
            }
        };
        blocks(a_.n_layer, "layers.", o_slow_);
        o_norm_ = vec("norm.weight", D);
        // tied head (dual_ar.rs:482-486): the embedding table itself, except with fp8 matrices, where the head is the fp8
        // quantisation of the same checkpoint tensor (the table stays bf16 for the lookups)
        if (a_.tie_word_embeddings && !kFp8) o_out_ = {o_tok_emb_, 0};
        else { o_out_ = slab(V, D); mat(a_.tie_word_embeddings ? "embeddings.weight" : "output.weight", V, D, 1, 0, o_out_); }
        o_fast_emb_ = emb("fast_embeddings.weight", a_.codebook_size, D);
        blocks(a_.n_fast_layer, "fast_layers.", o_fast_);
        o_fast_norm_ = vec("fast_norm.weight", D);
        o_fast_out_ = slab(a_.codebook_size, D); mat("fast_output.weight", a_.codebook_size, D, 1, 0, o_fast_out_);
        arena_bytes_ = off;
    }

    void alloc_runtime() {
        arena_.alloc(arena_bytes_);
        uint8_t* base = arena_.as<uint8_t>();
        auto lw = [&](const std::array<size_t, 10>& o) {
            LayerW w;
            w.wqkv = base + o[0]; w.wo = base + o[1]; w.w13 = base + o[2]; w.w2 = base + o[3];
            w.ffn_norm = (const float*)(base + o[4]); w.attn_norm = (const float*)(base + o[5]);
            if (kFp8) {
                w.s_qkv = (const float*)(base + o[6]); w.s_o = (const float*)(base + o[7]);
                w.s_13 = (const float*)(base + o[8]); w.s_2 = (const float*)(base + o[9]);
            }
            return w;
        };
        for (auto& o : o_slow_) slow_.push_back(lw(o));
        for (auto& o : o_fast_) { fast_.push_back(lw(o)); fast_.back().cache_resident = getenv("FISHRT_FAST_NT") == nullptr; }
        tok_emb_ = base + o_tok_emb_; cb_emb_ = base + o_cb_emb_; fast_emb_ = base + o_fast_emb_;
        out_w_ = base + o_out_.first; fast_out_w_ = base + o_fast_out_.first;
        if (kFp8) { out_s_ = (const float*)(base + o_out_.second); fast_out_s_ = (const float*)(base + o_fast_out_.second); }
        norm_w_ = (const float*)(base + o_norm_); fast_norm_w_ = (const float*)(base + o_fast_norm_);
        // RoPE tables on the host exactly as precompute_freqs_cis (dual_ar.rs:168-186): f32 powf / cos / sin
        const int half = a_.head_dim / 2, n_elem = a_.dim / a_.n_head;
        std::vector<float> ct((size_t)a_.max_seq_len * half), sn((size_t)a_.max_seq_len * half), theta(half);
        for (int j = 0; j < half; ++j) theta[j] = 1.f / std::pow(a_.rope_base, (float)(2 * j) / (float)n_elem);
        for (int p = 0; p < a_.max_seq_len; ++p)
            for (int j = 0; j < half; ++j) {
                const float ang = (float)p * theta[j];
                ct[(size_t)p * half + j] = std::cos(ang);
                sn[(size_t)p * half + j] = std::sin(ang);
            }
        d_cos_.alloc(ct.size() * 4); d_sin_.alloc(sn.size() * 4);
        FS_HIP(hipMemcpy(d_cos_.p, ct.data(), ct.size() * 4, hipMemcpyHostToDevice));
        FS_HIP(hipMemcpy(d_sin_.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
        // paged KV: one pool per slow layer, page = KV_PAGE tokens x Hk heads x Dh
        // (the table covers whole attention chunks: k_attn_decode reads the slot of every launched chunk unconditionally)
        n_chunks_ = (a_.max_seq_len + LmKernels<WT>::attn_chunk() - 1) / LmKernels<WT>::attn_chunk();
        FS_REQUIRE(n_chunks_ <= 128, "max_seq_len too large for the attention chunking (128 chunks)");
        max_pages_ = std::max((a_.max_seq_len + KV_PAGE - 1) / KV_PAGE, n_chunks_ * LmKernels<WT>::attn_chunk() / KV_PAGE);
        n_pages_ = max_pages_ * B_;
        page_elems_ = (size_t)a_.n_local_heads * KV_PAGE * a_.head_dim;
        // pools and partial buffers start zeroed: the attention kernels read rows / chunks past the current length
        // unconditionally and mask them afterwards, which needs finite (not uninitialised) contents
        kv_pool_.alloc((size_t)a_.n_layer * 2 * n_pages_ * page_elems_ * sizeof(KT));
        FS_HIP(hipMemset(kv_pool_.p, 0, kv_pool_.n));
        d_page_table_.alloc(sizeof(int) * (size_t)std::max(2 * B_ + 1, PR_MAX_ROWS) * max_pages_);  // + the staging rows of a session's joining requests (one group pass)
        FS_HIP(hipMemset(d_page_table_.p, 0, d_page_table_.n));
        for (int p = n_pages_ - 1; p >= 0; --p) free_pages_.push_back(p);
        seq_pages_.assign(B_, {});
        seq_len_.assign(B_, 0);
        fast_len_.assign(B_, 0);
        // fast-decoder KV: one page per (layer, sequence); its page table is a single zero
        fast_pool_.alloc((size_t)std::max(1, a_.n_fast_layer) * 2 * B_ * page_elems_ * sizeof(KT));
        FS_HIP(hipMemset(fast_pool_.p, 0, fast_pool_.n));
        d_zero_table_.alloc(sizeof(int) * 4);
        FS_HIP(hipMemset(d_zero_table_.p, 0, d_zero_table_.n));
        // activations / state
        // (>= PR_MAX_ROWS rows / states: a request-row launch covers 2, 4 or 8 rows whatever max_batch is -- the padding rows are read, found terminated and skipped)
        d_x_.alloc(sizeof(float) * (size_t)std::max(B_, PR_MAX_ROWS) * a_.dim);
        d_xf_.alloc(sizeof(float) * (size_t)std::max(B_, PR_MAX_ROWS) * a_.dim);
        FS_HIP(hipMemset(d_x_.p, 0, d_x_.n));
        d_q_.alloc(sizeof(float) * a_.dim);
        d_part_.alloc(sizeof(float) * (size_t)a_.n_head * n_chunks_ * (a_.head_dim + 2));
        FS_HIP(hipMemset(d_part_.p, 0, d_part_.n));
        d_act_.alloc(sizeof(float) * a_.intermediate_size);
        d_logits_slow_.alloc(sizeof(float) * a_.vocab_size);
        d_logits_fast_.alloc(sizeof(float) * a_.codebook_size);
        d_hid_slot_.alloc(sizeof(float*));
        FS_HIP(hipMemset(d_hid_slot_.p, 0, sizeof(float*)));
        d_state_.alloc(sizeof(SeqState) * (std::max(B_, PR_MAX_ROWS) + 1));  // + the staging state of a session's joining request (index B_)
        FS_HIP(hipMemset(d_state_.p, 0, d_state_.n));
        d_cfg_.alloc(sizeof(SampleCfg));
        SampleCfg c = base_cfg();
        FS_HIP(hipMemcpy(d_cfg_.p, &c, sizeof(c), hipMemcpyHostToDevice));
        d_rng_.alloc(sizeof(RngState));
        d_prompt_.alloc(sizeof(uint32_t) * (size_t)(a_.num_codebooks + 1) * a_.max_seq_len);
        out_cap_ = a_.max_seq_len + 8;
        d_out_.alloc(sizeof(uint32_t) * (size_t)B_ * a_.num_codebooks * out_cap_);
        const size_t ncb = a_.num_codebooks, cbs = a_.codebook_size;
        d_rp_mask_.alloc(sizeof(float) * ncb * cbs);
        d_rp_seen_.alloc(ncb * cbs);
        d_rp_ring_.alloc(sizeof(int) * ncb * 17);
        d_rp_meta_.alloc(sizeof(int) * ncb * 2);
        rp_.mask = d_rp_mask_.as<float>(); rp_.seen = d_rp_seen_.as<uint8_t>(); rp_.ring = d_rp_ring_.as<int>();
        rp_.ring_meta = d_rp_meta_.as<int>();
        FS_HIP(hipHostMalloc(&h_pin_, 4096, hipHostMallocDefault));
    }

    // rows of output.weight read by the generator's slow head: [im_end, V) for Fish 1.5 (utils.rs:13-16); the gathered
    // {pad_id, im_end_id} pair for Fish <= 1.4
    const void* slow_head_w() {
        if (!legacy_ && !generic_) return (const uint8_t*)out_w_ + (size_t)t_.im_end_id * a_.dim * sizeof(WT);
        if (!d_legacy_head_.p) d_legacy_head_.alloc(gathered_head_bytes() + sizeof(float) * n_audio_ + 256);
        return d_legacy_head_.p;
    }
    size_t gathered_head_bytes() const { return ((sizeof(WT) * (size_t)n_audio_ * a_.dim) + 15) & ~(size_t)15; }
    // per-row scales of the same rows (fp8 weights only)
    const float* slow_head_s() {
        if (!kFp8) return nullptr;
        if (!legacy_ && !generic_) return out_s_ + t_.im_end_id;
        slow_head_w();
        return (const float*)(d_legacy_head_.as<uint8_t>() + gathered_head_bytes());
    }
    void refresh_legacy_head() {
        if (generic_) {  // [W[im_end]; W[semantic_start .. V)]
            const size_t row = sizeof(WT) * (size_t)a_.dim;
            uint8_t* dst = (uint8_t*)const_cast<void*>(slow_head_w());
            FS_HIP(hipMemcpyAsync(dst, (const uint8_t*)out_w_ + t_.im_end_id * row, row, hipMemcpyDeviceToDevice, st_));
            FS_HIP(hipMemcpyAsync(dst + row, (const uint8_t*)out_w_ + t_.semantic_start_id * row, row * (n_audio_ - 1), hipMemcpyDeviceToDevice, st_));
            if (kFp8) {
                float* ds = const_cast<float*>(slow_head_s());
                FS_HIP(hipMemcpyAsync(ds, out_s_ + t_.im_end_id, sizeof(float), hipMemcpyDeviceToDevice, st_));
                FS_HIP(hipMemcpyAsync(ds + 1, out_s_ + t_.semantic_start_id, sizeof(float) * (n_audio_ - 1), hipMemcpyDeviceToDevice, st_));
            }
            FS_HIP(hipStreamSynchronize(st_));
            return;
        }
        if (!legacy_) return;
        launch_gather_rows<WT>(out_w_, a_.dim, t_.pad_id, t_.im_end_id, const_cast<void*>(slow_head_w()), st_);
        if (kFp8) launch_gather_rows<float>(out_s_, 1, t_.pad_id, t_.im_end_id, const_cast<float*>(slow_head_s()), st_);
        FS_HIP(hipStreamSynchronize(st_));
    }
    SeqState* state(int b) { return d_state_.as<SeqState>() + b; }
    float* x(int b) { return d_x_.as<float>() + (size_t)b * a_.dim; }
    float* xf(int b) { return d_xf_.as<float>() + (size_t)b * a_.dim; }
    KVView slow_kv(int layer, int b) {
        KVView v;
        KT* base = kv_pool_.as<KT>() + (size_t)layer * 2 * n_pages_ * page_elems_;
        v.k = base; v.v = base + (size_t)n_pages_ * page_elems_;
        v.page_table = d_page_table_.as<int>() + (size_t)b * max_pages_;
        return v;
    }
    KVView fast_kv(int layer, int b) {
        KVView v;
        KT* base = fast_pool_.as<KT>() + ((size_t)layer * 2 * B_) * page_elems_;
        v.k = base + (size_t)b * page_elems_;
        v.v = base + ((size_t)B_ + b) * page_elems_;
        v.page_table = d_zero_table_.as<int>();
        return v;
    }

    // ---- KV paging (host side; the device only ever sees the page table)
    void ensure_capacity(int b, int n_tokens) {
        FS_REQUIRE(n_tokens <= a_.max_seq_len, "sequence longer than max_seq_len");
        const int need = (n_tokens + KV_PAGE - 1) / KV_PAGE;
        auto& pg = seq_pages_[b];
        if ((int)pg.size() >= need) return;
        while ((int)pg.size() < need) {
            FS_REQUIRE(!free_pages_.empty(), "KV page pool exhausted");
            pg.push_back(free_pages_.back());
            free_pages_.pop_back();
        }
        FS_HIP(hipMemcpyAsync(d_page_table_.as<int>() + (size_t)b * max_pages_, pg.data(), sizeof(int) * pg.size(),
                              hipMemcpyHostToDevice, st_));
        FS_HIP(hipStreamSynchronize(st_));  // pg may be reallocated by the next call
    }
    void truncate(int b, int pos) {  // keep the first `pos` tokens (dual_ar.rs:392-404)
        const int keep = (pos + KV_PAGE - 1) / KV_PAGE;
        auto& pg = seq_pages_[b];
        while ((int)pg.size() > keep) { free_pages_.push_back(pg.back()); pg.pop_back(); }
        seq_len_[b] = pos;
    }

    // Runs the next `n` prompt tokens (columns state->step ..) of the staged prompt through the slow transformer,
    // appending their K/V.  Afterwards x(b) holds the pre-norm hidden state of the last processed token.
    void prefill_tokens(int b, int n, bool use_graph) {
        if (n <= 0) return;
        if (LmKernels<WT>::has_mfma_prefill() && n > 1 && a_.dim % 128 == 0 && a_.intermediate_size % 128 == 0) {
            ensure_prefill_buffers();
            RowsCtx c = rows_ctx(state(b), /*pos_step=*/1, /*pt_stride=*/0);
            for (int done = 0; done < n;) {
                const int M = std::min(a_.head_dim == 64 ? kRowsCap : kPartRows, n - done);
                c.nc_launch = chunk_bucket(seq_len_[b] + done + M);
                LmKernels<WT>::prefill_embed(d_, tok_emb_, cb_emb_, a_.num_codebooks, a_.codebook_size, d_cfg_.as<SampleCfg>(),
                                             d_prompt_.as<uint32_t>(), state(b), M, d_pfx_.as<float>(), st_);
                for (int l = 0; l < a_.n_layer; ++l) LmKernels<WT>::rows_layer(d_, M, c, slow_[l], slow_kv(l, b), l == 0, st_);
                LmKernels<WT>::rows_finish(d_, M, c, nullptr, st_);
                launch_advance_n(state(b), M, st_);
                done += M;
                if (done == n)
                    FS_HIP(hipMemcpyAsync(x(b), d_pfx_.as<float>() + (size_t)(M - 1) * a_.dim, sizeof(float) * a_.dim,
                                          hipMemcpyDeviceToDevice, st_));
            }
            return;
        }
        for (int l = 0; l < n; ++l) {
            set_bucket(seq_len_[b] + l + 1);
            if (use_graph && b == 0) { use_graphs_for_bucket(); FS_HIP(hipGraphLaunch(g_step_, st_)); continue; }
            LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, a_.num_codebooks, a_.codebook_size, d_cfg_.as<SampleCfg>(),
                                 d_prompt_.as<uint32_t>(), state(b), x(b), st_);
            enqueue_slow_layers(b);
            launch_advance(state(b), st_);
        }
    }
    void ensure_prefill_buffers() {
        if (d_pfx_.p) return;
        const int I = a_.intermediate_size;
        down_split_ = I % 1024 == 0 ? I / 1024 : (I % 256 == 0 ? I / 256 : I / 128);
        d_pfx_.alloc(sizeof(float) * kRowsCap * a_.dim);
        d_pfq_.alloc(sizeof(float) * kRowsCap * a_.dim);
        d_pfslab_.alloc(sizeof(float) * down_split_ * kRowsCap * a_.dim);
        d_pfa_.alloc(sizeof(uint16_t) * 2 * kRowsCap * a_.dim);
        d_pfa2_.alloc(sizeof(uint16_t) * 2 * kRowsCap * a_.dim);
        d_pfss_.alloc(sizeof(float) * kRowsCap * (a_.dim / 16));
        FS_HIP(hipMemsetAsync(d_pfa2_.p, 0, d_pfa2_.n, st_));
        FS_HIP(hipMemsetAsync(d_pfss_.p, 0, d_pfss_.n, st_));
        d_pfc_.alloc(sizeof(uint16_t) * 2 * kRowsCap * a_.intermediate_size);
        d_pfpart_.alloc(sizeof(float) * kPartRows * (size_t)a_.n_head * n_chunks_ * (a_.head_dim + 2));
        FS_HIP(hipMemsetAsync(d_pfx_.p, 0, d_pfx_.n, st_));
        FS_HIP(hipMemsetAsync(d_pfpart_.p, 0, d_pfpart_.n, st_));
        FS_HIP(hipMemsetAsync(d_pfa_.p, 0, d_pfa_.n, st_));
        FS_HIP(hipMemsetAsync(d_pfc_.p, 0, d_pfc_.n, st_));
    }
    // the in-launch split-K sums of the folded decode steps give up after ~0.1 s of polling (a tile's blocks were not co-resident): loud failure
    void check_rows_xchg() {
        if (!d_epoch_.p) return;
        uint32_t e[2] = {0, 0};
        FS_HIP(hipMemcpy(e, d_epoch_.p, sizeof(e), hipMemcpyDeviceToHost));
        if (e[1] != 0) {
            const uint32_t z = 0;
            FS_HIP(hipMemcpy(d_epoch_.as<uint32_t>() + 1, &z, sizeof(z), hipMemcpyHostToDevice));
            // this call's tokens are garbage (reported), but the handle recovers: the folded step is off from now on and the captured step
            // graphs are dropped, so the next call runs the slab + k_prep step instead of timing out again
            fold_off_ = true;
            drop_batch_graphs();
            throw Error("static-batch decode: " + std::to_string(e[1]) + " split-K exchange waits timed out (a down projection's blocks were not co-resident); "
                        "this handle falls back to the slab path for its next calls");
        }
    }
    void ensure_batch_buffers() {
        if (d_xfrows_.p) return;
        ld_slow_ = ((n_audio_ + 63) / 64) * 64;
        d_xfrows_.alloc(sizeof(float) * kRows * a_.dim);
        d_lrows_.alloc(sizeof(float) * kRows * ld_slow_);
        d_lfast_.alloc(sizeof(float) * kRows * a_.codebook_size);
        std::vector<SeqState> fs(8);
        for (int i = 0; i < 8; ++i) { fs[i] = SeqState{}; fs[i].pos = i; }
        d_fast_state_.alloc(sizeof(SeqState) * 8);
        FS_HIP(hipMemcpy(d_fast_state_.p, fs.data(), sizeof(SeqState) * 8, hipMemcpyHostToDevice));
        std::vector<int> tb(B_);
        for (int i = 0; i < B_; ++i) tb[i] = i;
        d_rwords_.alloc(sizeof(uint32_t) * 16 * kRows);
        // folded decode steps (k_gemm_down): exchange units of the in-launch split-K sums + {step epoch, timeouts}
        d_xchg_.alloc(rows_xchg_bytes(a_.dim));
        FS_HIP(hipMemset(d_xchg_.p, 0, d_xchg_.n));
        d_epoch_.alloc(sizeof(uint32_t) * 2);
        const uint32_t e0[2] = {1u, 0u};
        FS_HIP(hipMemcpy(d_epoch_.p, e0, sizeof(e0), hipMemcpyHostToDevice));
        d_fast_table_.alloc(sizeof(int) * B_);
        FS_HIP(hipMemcpy(d_fast_table_.p, tb.data(), sizeof(int) * B_, hipMemcpyHostToDevice));
    }
    // one static-batch frame for B rows: x rows (d_pfx_) hold the embedded inputs of position state(0)->pos
    void enqueue_batch_frame(int B) {
        const int C = a_.num_codebooks;
        // lock-step static batch: every row at state(0)->pos (left-padded prompts); session: row m is its own sequence at state(m)->pos
        RowsCtx cs = rows_ctx(state(0), /*pos_step=*/sess_active_ ? -1 : 0, /*pt_stride=*/max_pages_);
        cs.nc_launch = nc_launch_;
        // decode steps of <= 32 rows: the down projections close every layer themselves (in-launch split-K sums: no slabs, no k_prep nodes but the first)
        cs.xchg = d_xchg_.p; cs.epoch = d_epoch_.as<uint32_t>();
        const bool fold = !fold_off_ && LmKernels<WT>::rows_fold_ok(d_, B, cs) && a_.n_layer <= 64 && a_.n_fast_layer <= 8 && C <= 32;
        cs.fold = fold;
        for (int l = 0; l < a_.n_layer; ++l, cs.node_id = (uint32_t)l)
            LmKernels<WT>::rows_layer(d_, B, cs, slow_[l], slow_kv(l, 0), l == 0, st_, fold ? (l + 1 < a_.n_layer ? slow_[l + 1].attn_norm : norm_w_) : nullptr);
        if (!fold) LmKernels<WT>::rows_finish(d_, B, cs, norm_w_, st_);
        LmKernels<WT>::rows_head(d_, B, cs, slow_head_w(), slow_head_s(), n_audio_, d_lrows_.as<float>(), ld_slow_, st_, fold);
        // block-parallel samplers (temp > 1e-7, top_k <= 256): the step's C + 1 StdRng words per row are derived up front
        const uint32_t* words = rows_par_ ? d_rwords_.as<uint32_t>() : nullptr;
        if (rows_par_) SampleKernels<WT>::rows_rng_words(d_rng_.as<RngState>(), B, C + 1, state(0), d_rwords_.as<uint32_t>(), st_);
        const bool capt = cap_frames_ > 0 && d_rcap_.p && d_rcap_.n >= sizeof(float) * (size_t)B * cap_frames_ * 9 * 2048;  // (fs_lm_debug_capture)
        if (capt) launch_cap_rows_logits(d_lrows_.as<float>(), ld_slow_, n_audio_, state(0), d_cfg_.as<SampleCfg>(), B, d_rcap_.as<float>(), cap_frames_, 0, st_);
        // folded steps: the sampler that writes a fast-decoder input row also leaves its first layer's normalised GEMM input (no k_prep node)
        const float* prep_g = fold && a_.dim <= 1024 ? fast_[0].attn_norm : nullptr;
        SampleKernels<WT>::sample_slow_rows(d_, d_lrows_.as<float>(), ld_slow_, n_audio_, d_cfg_.as<SampleCfg>(), d_rng_.as<RngState>(), B,
                                            C + 1, state(0), cs.X, d_xfrows_.as<float>(), st_, words, prep_g, cs.A, fold ? d_epoch_.as<uint32_t>() : nullptr);
        for (int cbi = 0; cbi < C; ++cbi) {
            RowsCtx cf = cs;
            cf.X = d_xfrows_.as<float>();
            cf.state = d_fast_state_.as<SeqState>() + cbi;
            cf.pos_step = 0;  // (the fast decoder's rows always sit at codebook position cbi)
            cf.pt_stride = 1;
            cf.nc_launch = 1;
            cf.small_attn = a_.num_codebooks <= 8;
            cf.first_prepped = prep_g != nullptr;
            cf.identity_pages = true;  // d_fast_table_[i] == i
            cf.attn_t1 = cbi == 0;     // d_fast_state_[0].pos == 0
            // passes 1..: the first fast layer's q / k / v come from the qkv table (row = the code the previous pass picked): no Wqkv node
            const bool tbl = fold && cbi > 0 && d_qkv0_.p && !getenv("FISHRT_ROWS_NO_QKV0") && a_.codebook_size == 1024;
            if (tbl) { cf.qkv0_tbl = d_qkv0_.as<float>(); cf.row_states = state(0); cf.code_slot = cbi; }
            for (int l = 0; l < a_.n_fast_layer; ++l) {
                KVView kv;
                KT* base = fast_pool_.as<KT>() + ((size_t)l * 2 * B_) * page_elems_;
                kv.k = base; kv.v = base + (size_t)B_ * page_elems_; kv.page_table = d_fast_table_.as<int>();
                cf.node_id = 64u + (uint32_t)cbi * 8u + (uint32_t)l;
                LmKernels<WT>::rows_layer(d_, B, cf, fast_[l], kv, l == 0, st_, fold ? (l + 1 < a_.n_fast_layer ? fast_[l + 1].attn_norm : fast_norm_w_) : nullptr);
            }
            if (!fold) LmKernels<WT>::rows_finish(d_, B, cf, fast_norm_w_, st_);
            LmKernels<WT>::rows_head(d_, B, cf, fast_out_w_, kFp8 ? fast_out_s_ : nullptr, a_.codebook_size, d_lfast_.as<float>(), a_.codebook_size, st_, fold);
            if (capt) launch_cap_rows_logits(d_lfast_.as<float>(), a_.codebook_size, a_.codebook_size, state(0), d_cfg_.as<SampleCfg>(), B, d_rcap_.as<float>(),
                                             cap_frames_, 1 + cbi, st_);
            SampleKernels<WT>::sample_fast_rows(d_, d_lfast_.as<float>(), cbi, C, a_.codebook_size, d_cfg_.as<SampleCfg>(),
                                                d_rng_.as<RngState>(), B, state(0), fast_emb_, d_xfrows_.as<float>(), tok_emb_, cb_emb_,
                                                cs.X, d_out_.as<uint32_t>(), out_cap_, st_, words,
                                                // (with the qkv table the next pass's first layer needs no normalised GEMM input from this sampler)
                                                (fold && d_qkv0_.p && !getenv("FISHRT_ROWS_NO_QKV0") && a_.codebook_size == 1024) ? nullptr : prep_g, cs.A);
        }
        if (capt) launch_cap_rows_picks(state(0), d_cfg_.as<SampleCfg>(), B, d_rcap_.as<float>(), cap_frames_, C, st_);
    }
    // the capture record of a row-path call: [B][cap_frames][9][2048] (allocated when fs_lm_debug_capture is armed)
    void ensure_rows_capture(int B) {
        if (!cap_frames_) return;
        FS_REQUIRE((size_t)B * cap_frames_ <= 4096, "decision capture on the row path: rows x frames limited to 4096 (302 MB)");
        d_rcap_.alloc(sizeof(float) * (size_t)B * cap_frames_ * 9 * 2048);
        FS_HIP(hipMemsetAsync(d_rcap_.p, 0, d_rcap_.n, st_));
    }
    void drop_batch_graphs() {
        for (auto& kv : batch_graphs_) if (kv.second) (void)hipGraphExecDestroy(kv.second);
        batch_graphs_.clear();
    }
    hipGraphExec_t batch_graph(int B) {
        const int key = (sess_active_ ? 1 << 24 : 0) + (rows_par_ ? 1 << 25 : 0) + B * 1024 + nc_launch_;
        auto it = batch_graphs_.find(key);
        if (it != batch_graphs_.end()) return it->second;
        {   // the co-residency query of the folded step (rows_fold_ok -> occupancy API) is answered once, OUTSIDE stream capture
            RowsCtx cs = rows_ctx(state(0), 0, max_pages_);
            cs.xchg = d_xchg_.p; cs.epoch = d_epoch_.as<uint32_t>();
            (void)LmKernels<WT>::rows_fold_ok(d_, B, cs);
        }
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        // warm every kernel once outside capture (function attributes are set lazily on first launch)
        if (!batch_warm_) { enqueue_batch_frame_dry(B); batch_warm_ = true; }
        FS_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
        enqueue_batch_frame(B);
        FS_HIP(hipStreamEndCapture(st_, &g));
        FS_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        FS_HIP(hipGraphDestroy(g));
        batch_graphs_[key] = ge;
        return ge;
    }
    // hipFuncSetAttribute (dynamic LDS of the GEMM) is not capturable: issue it before the first capture
    void enqueue_batch_frame_dry(int) { LmKernels<WT>::rows_warmup(); }

    RowsCtx rows_ctx(const SeqState* st, int pos_step, int pt_stride) {
        RowsCtx c;
        c.X = d_pfx_.as<float>(); c.Q = d_pfq_.as<float>(); c.part = d_pfpart_.as<float>(); c.P = d_pfslab_.as<float>();
        c.Mcap = kRowsCap; c.part_rows = kPartRows; c.down_split = down_split_;
        c.A = d_pfa_.as<uint16_t>(); c.C = d_pfc_.as<uint16_t>();
        c.A2 = d_pfa2_.as<uint16_t>(); c.ss = d_pfss_.as<float>();
        c.cos_t = d_cos_.as<float>(); c.sin_t = d_sin_.as<float>();
        c.state = st; c.n_chunks_max = n_chunks_; c.nc_launch = n_chunks_; c.pos_step = pos_step; c.pt_stride = pt_stride;
        return c;
    }

    // naps before the first sweep of each stage kind (64-clock units; tuned on MI355X, profiles/r03_poll_naps.txt); the environment
    // variables ("a,b,c,d,e,f") override them for tuning runs
    static constexpr int kNapsFast[6] = {16, 16, 16, 16, 16, 16}, kNapsSlow[6] = {20, 16, 6, 36, 36, 12};  // (round 6, behind the layer-0 qkv table: 530.0 -> 528.4 us, profiles/r06_tune_naps.txt)
    // (re-tuned in round 5 behind the publishing-wave epilogues and the early W13 request: 558.8 -> 551.1 us on the tuner's workload; again at the end of the round: 543.1 -> 542.2, and behind the matrix-core S4 of the slow kernel: 541.9 -> 540.3)
    // the same coordinate descent on the in-launch-sampler instantiation of k_fast_persist (680 -> 662 us per sampled frame) and on the
    // e4m3 image of k_slow_persist (595 -> 585 us per fp8 frame): their stage arithmetic differs, so the edges complete at other times
    static constexpr int kNapsFastSampled[6] = {12, 16, 12, 20, 12, 16}, kNapsSlowFp8[6] = {20, 4, 28, 24, 24, 0};  // (round 5 re-tune: profiles/r05_tune_naps.txt)
    static void set_naps(int (&naps)[6], const char* env, const int (&dflt)[6]) {
        for (int i = 0; i < 6; ++i) naps[i] = dflt[i];
        if (const char* v = getenv(env)) {
            int i = 0;
            for (const char* p = v; *p && i < 6; ++i) { naps[i] = atoi(p); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        }
    }
    // ---- persistent fast decoder (lm_persist.hip): per-lane weight image, edge buffers, control words
    void pack_persist() {
        persist_ok_ = false;
        if constexpr (std::is_same<WT, bf16_t>::value || std::is_same<WT, fp8_t>::value) {
            constexpr bool FP8 = std::is_same<WT, fp8_t>::value;
            if (getenv("FISHRT_NO_PERSIST")) return;
            if (!fast_persist_supported(d_, a_.n_fast_layer, a_.num_codebooks, a_.codebook_size)) return;
            hipDeviceProp_t prop;
            FS_HIP(hipGetDeviceProperties(&prop, device_));
            if (prop.multiProcessorCount < PF_BLOCKS) return;  // every workgroup needs its own CU
            if (!d_pack_.p) {
                d_pack_.alloc(fast_persist_pack_bytes());
                d_edges_.alloc(fast_persist_edge_bytes());
                FS_HIP(hipMemset(d_edges_.p, 0, d_edges_.n));
                d_ctl_.alloc(320);
                FS_HIP(hipMemset(d_ctl_.p, 0, d_ctl_.n));
                if (FP8) d_fscl_.alloc(sizeof(float) * (size_t)PF_BLOCKS * PF_SCL);
            }
            launch_fast_persist_pack(fast_.data(), fast_out_w_, d_pack_.p, st_, FP8, fast_out_s_, d_fscl_.as<float>());
            // layer-0 qkv table of the codebook passes 1..7 (lm_persist.hip: 7 of a frame's 137 stages become three loads per lane)
            if (!getenv("FISHRT_FAST_NO_QKV0")) {
                if (!d_qkv0_.p) d_qkv0_.alloc(fast_persist_qkv0_bytes());
                launch_fast_persist_qkv0_table(d_pack_.p, FP8 ? d_fscl_.as<float>() : nullptr, fast_[0].attn_norm, fast_emb_, d_.eps, d_qkv0_.as<float>(), st_);
            }
            FS_HIP(hipStreamSynchronize(st_));
            persist_ok_ = true;
            // slow transformer: same geometry, the audio-range head must fit 8 rows per workgroup
            pslow_ok_ = false;
            if (getenv("FISHRT_NO_PERSIST_SLOW") || n_audio_ > 8 * PF_BLOCKS) return;
            if (!d_spack_.p) {
                d_spack_.alloc(slow_persist_pack_bytes(a_.n_layer, FP8));
                d_hpack_.alloc((size_t)PF_BLOCKS * PS_HEAD_IMAGE);
                if (FP8) d_sscl_.alloc(sizeof(float) * slow_persist_scale_floats(a_.n_layer));
                d_snorms_.alloc(sizeof(float) * (size_t)(2 * a_.n_layer + 1) * a_.dim);
                d_sedges_.alloc(slow_persist_edge_bytes());
                FS_HIP(hipMemset(d_sedges_.p, 0, d_sedges_.n));
                d_sctl_.alloc(320);
                FS_HIP(hipMemset(d_sctl_.p, 0, d_sctl_.n));
            }
            std::vector<const float*> np;
            for (int l = 0; l < a_.n_layer; ++l) { np.push_back(slow_[l].attn_norm); np.push_back(slow_[l].ffn_norm); }
            np.push_back(norm_w_);
            if (FP8) launch_slow_persist_pack_fp8(slow_.data(), a_.n_layer, slow_head_w(), slow_head_s(), n_audio_, np.data(), d_spack_.p, d_hpack_.p,
                                                  d_sscl_.as<float>(), d_snorms_.as<float>(), st_);
            else launch_slow_persist_pack(slow_.data(), a_.n_layer, slow_head_w(), n_audio_, np.data(), d_spack_.p, d_hpack_.p, d_snorms_.as<float>(), st_);
            FS_HIP(hipStreamSynchronize(st_));
            pslow_ok_ = true;
        }
    }
    SlowPersistArgs pslow_args() {
        SlowPersistArgs A = {};
        A.wpack = d_spack_.p; A.hpack = d_hpack_.p; A.norms = d_snorms_.as<float>();
        A.scales = d_sscl_.p ? d_sscl_.as<float>() : nullptr;
        A.hscales = d_sscl_.p ? d_sscl_.as<float>() + (size_t)a_.n_layer * PF_BLOCKS * 48 : nullptr;
        A.n_layer = a_.n_layer; A.n_head_rows = n_audio_;
        A.cos_t = d_cos_.as<float>(); A.sin_t = d_sin_.as<float>(); A.eps = d_.eps;
        A.x = x(0); A.logits = d_logits_slow_.as<float>(); A.state = state(0);
        A.kv_pool = kv_pool_.p; A.layer_half = (size_t)n_pages_ * page_elems_;
        A.page_table = d_page_table_.as<int>();
        // Token slices per query head.  8 slices = 128 attention workgroups: the other 128 have no attention item and request their W13 slice
        // three stages ahead (k_slow_persist early_mode bit 0), and S3 merges 8 partials per head.  16 slices halve S2's tokens per slice but give
        // every workgroup an item (W13 arrives as one burst again: S4 waits 31 -> 47-49 us per frame) and double S3's sweep: measured per frame
        // at 8 / 16 slices (profiles/r05_kv_slices.txt): KV 1100 568 / 588 us, 2100 587 / 612, 3100 605 / 611, 4100 624 / 629, 6000 646 / 632
        // (fp8: 1100 576 / 583, 2100 594 / 602, 4100 631 / 620) -- 16 only beyond 4096 cached tokens.
        A.n_sl = std::min(nc_launch_, nc_launch_ > 32 ? 16 : 8);
        if (const char* c = getenv("FISHRT_NSL_MAX")) A.n_sl = std::max(1, std::min(A.n_sl, atoi(c)));  // (debug knobs)
        if (const char* c = getenv("FISHRT_NSL_MIN")) A.n_sl = std::min(16, std::max(A.n_sl, atoi(c)));
        while (A.n_sl & (A.n_sl - 1)) A.n_sl &= A.n_sl - 1;  // the kernel's item mapping needs a power of two (10 / 12 slices: grid-wait timeouts)
        // (measured, round 3: half / a quarter as many slices -> S2 +44 / +115 us per frame, S3 only -10 / -12)
        A.edges = d_sedges_.as<unsigned long long>();
        A.ctl = d_sctl_.as<uint32_t>();
        A.prof = getenv("FISHRT_PERSIST_PROF") ? reinterpret_cast<unsigned long long*>(d_sctl_.as<uint32_t>() + 16) : nullptr;
        A.peer_stamps = (A.prof && d_ctl_.p) ? reinterpret_cast<const unsigned long long*>(d_ctl_.as<uint32_t>() + 16) + 16 : nullptr;
        set_naps(A.naps, "FISHRT_NAPS_SLOW", kFp8 ? kNapsSlowFp8 : kNapsSlow);
        A.prof_wg = getenv("FISHRT_PERSIST_PROF_WG") ? atoi(getenv("FISHRT_PERSIST_PROF_WG")) : 0;
        // k_slow_persist early_mode bits.  bit 3 (round 6, bf16 images): an attention workgroup requests its first K/V tile BEHIND S1's publish instead of
        // in front of the Wqkv rows -- the publishing stores used to queue behind the tile's 32 KB of loads in the CU's vector-memory pipeline (S1 work
        // 1.2 -> 0.7 us on those workgroups; 529.3 -> 526.1 us per frame, 614.5 -> 604.6 at 4100 cached tokens; fp8: no difference, left off)
        A.l2_touch = getenv("FISHRT_SLOW_EARLY") ? atoi(getenv("FISHRT_SLOW_EARLY")) : (kFp8 ? 1 : 9);
        return A;
    }
    // the persistent fast decoder takes the slow-token decision in its prologue (no k_sample_slow node) whenever it runs
    bool fold_slow_sampler() const { return use_persist_ && n_audio_ <= 2048; }  // (Fish <= 1.4 too: the 2-way {pad, im_end} draw is taken in-launch)
    FastPersistArgs persist_args() {
        FastPersistArgs A = {};
        A.wpack = d_pack_.p;
        A.scales = d_fscl_.p ? d_fscl_.as<float>() : nullptr;
        for (int l = 0; l < PF_LAYERS; ++l) { A.norms[2 * l] = fast_[l].attn_norm; A.norms[2 * l + 1] = fast_[l].ffn_norm; }
        A.norms[2 * PF_LAYERS] = fast_norm_w_;
        A.fast_emb = fast_emb_; A.tok_emb = tok_emb_; A.cb_emb = cb_emb_;
        A.qkv0_tbl = d_qkv0_.p ? d_qkv0_.as<float>() : nullptr;
        A.cos_t = d_cos_.as<float>(); A.sin_t = d_sin_.as<float>();
        A.eps = d_.eps;
        const bool fold = fold_slow_sampler();
        A.xf = fold ? x(0) : xf(0); A.x = x(0);
        A.slow_logits = fold ? d_logits_slow_.as<float>() : nullptr; A.n_slow = n_audio_; A.hid_slot = d_hid_slot_.as<float*>();
        A.cap = d_cap_.as<float>(); A.cap_frames = cap_frames_;
        A.state = state(0); A.cfg = d_cfg_.as<SampleCfg>(); A.rp = rp_; A.rng = d_rng_.as<RngState>();
        A.out_codes = d_out_.as<uint32_t>(); A.out_cap = out_cap_;
        A.edges = d_edges_.as<unsigned long long>();
        A.ctl = d_ctl_.as<uint32_t>();
        A.prof = getenv("FISHRT_PERSIST_PROF") ? reinterpret_cast<unsigned long long*>(d_ctl_.as<uint32_t>() + 16) : nullptr;
        A.peer_stamps = (A.prof && d_sctl_.p) ? reinterpret_cast<const unsigned long long*>(d_sctl_.as<uint32_t>() + 16) + 16 : nullptr;
        set_naps(A.naps, "FISHRT_NAPS_FAST", persist_sampled_ ? kNapsFastSampled : kNapsFast);
        return A;
    }


    // ---- request rows (lm_persist_rows.hip): MFMA weight images, row edge buffers, per-row sampler / repetition-penalty state
    void ensure_rows(int R) {
        if constexpr (std::is_same<WT, bf16_t>::value || kFp8) {
            if (!d_rimg_.p) {
                d_rimg_.alloc(slow_persist_pack_bytes(a_.n_layer, false));  // (FS_FP8 too: the row images hold the e4m3 weights widened to bf16)
                d_rhimg_.alloc((size_t)PF_BLOCKS * PS_HEAD_IMAGE);
                if constexpr (kFp8) launch_rows_pack_fp8(slow_.data(), a_.n_layer, slow_head_w(), n_audio_, d_rimg_.p, d_rhimg_.p, st_);
                else launch_rows_pack(slow_.data(), a_.n_layer, slow_head_w(), n_audio_, d_rimg_.p, d_rhimg_.p, st_);
                if constexpr (!kFp8) {  // bf16 handles: the row kernels multiply every K-summed row result by a table entry unconditionally -- ones
                    const size_t n1 = (size_t)a_.n_layer * PF_BLOCKS * 48 + (size_t)PF_BLOCKS * 8 + (size_t)PF_BLOCKS * PF_SCL;
                    std::vector<float> ones(n1, 1.0f);
                    d_rones_.alloc(n1 * sizeof(float));
                    FS_HIP(hipMemcpyAsync(d_rones_.p, ones.data(), n1 * sizeof(float), hipMemcpyHostToDevice, st_));
                    FS_HIP(hipStreamSynchronize(st_));
                }
                d_rpairs_.alloc((size_t)PF_BLOCKS * 40 * PF_THREADS * 4);
                launch_rows_pack_rowpairs(d_pack_.p, d_rpairs_.p, st_);
                d_redges_s_.alloc(rows_slow_edge_bytes(PR_MAX_ROWS));
                FS_HIP(hipMemsetAsync(d_redges_s_.p, 0, d_redges_s_.n, st_));
                d_redges_f_.alloc(rows_fast_edge_bytes(PR_FAST_ROWS));
                FS_HIP(hipMemsetAsync(d_redges_f_.p, 0, d_redges_f_.n, st_));
                d_rctl_s_.alloc(256); d_rctl_f_.alloc(256);
                FS_HIP(hipMemsetAsync(d_rctl_s_.p, 0, 256, st_));
                FS_HIP(hipMemsetAsync(d_rctl_f_.p, 0, 256, st_));
                d_rlogits_.alloc(sizeof(float) * PR_MAX_ROWS * PR_LD);
                d_rcfg_.alloc(sizeof(SampleCfg) * PR_MAX_ROWS);
                d_rrng_.alloc(sizeof(RngState) * PR_MAX_ROWS);
                d_rbudget_.alloc(sizeof(int) * PR_MAX_ROWS);
                const size_t ncb = a_.num_codebooks, cbs = a_.codebook_size;
                d_rrp_mask_.alloc(sizeof(float) * PR_MAX_ROWS * ncb * cbs);
                d_rrp_seen_.alloc((size_t)PR_MAX_ROWS * ncb * cbs);
                d_rrp_ring_.alloc(sizeof(int) * PR_MAX_ROWS * ncb * 17);
                d_rrp_meta_.alloc(sizeof(int) * PR_MAX_ROWS * ncb * 2);
                FS_HIP(hipStreamSynchronize(st_));
            }
            (void)R;
        } else {
            throw Error("request rows need a bf16 or fp8 handle");
        }
    }
    RepPenState rows_rp(int i) {
        const size_t ncb = a_.num_codebooks, cbs = a_.codebook_size;
        RepPenState rp;
        rp.mask = d_rrp_mask_.as<float>() + (size_t)i * ncb * cbs; rp.seen = d_rrp_seen_.as<uint8_t>() + (size_t)i * ncb * cbs;
        rp.ring = d_rrp_ring_.as<int>() + (size_t)i * ncb * 17; rp.ring_meta = d_rrp_meta_.as<int>() + (size_t)i * ncb * 2;
        return rp;
    }
    static constexpr int kNapsRowsSlow[6] = {28, 4, 8, 40, 24, 28}, kNapsRowsFast[6] = {20, 16, 16, 16, 20, 20};  // tools/tune_naps_rows.py at R = 4 (945 -> 933 us per frame; re-tuned after the half-block S2)
    RowsSlowArgs rows_slow_args(int R) {
        RowsSlowArgs A = {};
        A.wimg = d_rimg_.p; A.himg = d_rhimg_.p; A.norms = d_snorms_.as<float>();
        const float* sc0 = kFp8 ? d_sscl_.as<float>() : d_rones_.as<float>();  // FS_FP8: the slow persistent kernel's scale tables (same row -> workgroup mapping); bf16: ones
        A.scales = sc0;
        A.hscales = sc0 + (size_t)a_.n_layer * PF_BLOCKS * 48;
        A.n_layer = a_.n_layer; A.n_head_rows = n_audio_;
        A.cos_t = d_cos_.as<float>(); A.sin_t = d_sin_.as<float>(); A.eps = d_.eps;
        A.x = x(0); A.logits = d_rlogits_.as<float>(); A.state = state(0);
        A.kv_pool = kv_pool_.p; A.layer_half = (size_t)n_pages_ * page_elems_;
        A.page_table = d_page_table_.as<int>(); A.pt_stride = max_pages_;
        A.n_sl = std::max(1, std::min(nc_launch_, 16 / R));
        if (const char* c = getenv("FISHRT_ROWS_NSL_MAX")) A.n_sl = std::max(1, std::min(A.n_sl, atoi(c)));  // experiment: fewer, longer attention slices
        A.edges = d_redges_s_.as<unsigned long long>();
        A.ctl = d_rctl_s_.as<uint32_t>();
        A.prof = getenv("FISHRT_PERSIST_PROF") ? reinterpret_cast<unsigned long long*>(d_rctl_s_.as<uint32_t>() + 16) : nullptr;
        set_naps(A.naps, "FISHRT_NAPS_ROWS_SLOW", kNapsRowsSlow);
        return A;
    }
    RowsFastArgs rows_fast_args(int r0, int Rf) {
        RowsFastArgs A = {};
        (void)Rf;
        A.wpack = d_pack_.p; A.rowpairs = d_rpairs_.as<uint32_t>();
        A.scales = kFp8 ? d_fscl_.as<float>() : d_rones_.as<float>() + (size_t)a_.n_layer * PF_BLOCKS * 48 + (size_t)PF_BLOCKS * 8;
        for (int l = 0; l < PF_LAYERS; ++l) { A.norms[2 * l] = fast_[l].attn_norm; A.norms[2 * l + 1] = fast_[l].ffn_norm; }
        A.norms[2 * PF_LAYERS] = fast_norm_w_;
        A.fast_emb = fast_emb_; A.tok_emb = tok_emb_; A.cb_emb = cb_emb_;
        A.qkv0_tbl = (d_qkv0_.p && !getenv("FISHRT_ROWS_FAST_NO_QKV0")) ? d_qkv0_.as<float>() : nullptr;
        A.cos_t = d_cos_.as<float>(); A.sin_t = d_sin_.as<float>(); A.eps = d_.eps;
        A.xf = x(r0); A.x = x(r0);
        A.slow_logits = d_rlogits_.as<float>() + (size_t)r0 * PR_LD; A.n_slow = n_audio_;
        A.cap = cap_frames_ ? d_rcap_.as<float>() + (size_t)r0 * cap_frames_ * 9 * 2048 : nullptr; A.cap_frames = cap_frames_;
        A.state = state(r0); A.cfg = d_rcfg_.as<SampleCfg>() + r0; A.budget = d_rbudget_.as<int>() + r0; A.rng = d_rrng_.as<RngState>() + r0;
        const RepPenState rp = rows_rp(r0);
        A.rp_mask = rp.mask; A.rp_ring = rp.ring; A.rp_meta = rp.ring_meta;
        A.out_codes = d_out_.as<uint32_t>() + (size_t)r0 * a_.num_codebooks * out_cap_; A.out_cap = out_cap_;
        A.edges = d_redges_f_.as<unsigned long long>();
        A.ctl = d_rctl_f_.as<uint32_t>();
        A.prof = getenv("FISHRT_PERSIST_PROF") ? reinterpret_cast<unsigned long long*>(d_rctl_f_.as<uint32_t>() + 16) : nullptr;
        set_naps(A.naps, "FISHRT_NAPS_ROWS_FAST", kNapsRowsFast);
        A.nap_draw = getenv("FISHRT_NAP_ROWS_DRAW") ? atoi(getenv("FISHRT_NAP_ROWS_DRAW")) : 16;  // flat between 0 and 48 (1075 us per sampled 4-row frame), 1088 at 96, 1111 at 200
        return A;
    }
    // the batch-1 fast kernel on row i of a multi-request call (FISHRT_ROWS_FAST_SINGLE: reference composition for the row fast kernel)
    FastPersistArgs persist_args(int i) {
        FastPersistArgs A = persist_args();
        A.xf = x(i); A.x = x(i);
        A.slow_logits = d_rlogits_.as<float>() + (size_t)i * PR_LD;
        A.cap = cap_frames_ ? d_rcap_.as<float>() + (size_t)i * cap_frames_ * 9 * 2048 : nullptr;
        A.state = state(i); A.cfg = d_rcfg_.as<SampleCfg>() + i; A.rp = rows_rp(i); A.rng = d_rrng_.as<RngState>() + i;
        A.out_codes = d_out_.as<uint32_t>() + (size_t)i * a_.num_codebooks * out_cap_;
        return A;
    }

    // ---- kernel sequences
    void enqueue_slow_layers(int b) {
        for (int l = 0; l < a_.n_layer; ++l) {
            const LayerW& w = slow_[l];
            KVView kv = slow_kv(l, b);
            LmKernels<WT>::qkv(d_, x(b), w, d_cos_.as<float>(), d_sin_.as<float>(), state(b), 0, 0, d_q_.as<float>(), kv, st_);
            LmKernels<WT>::attn_decode(d_, d_q_.as<float>(), kv, state(b), d_part_.as<float>(), n_chunks_, nc_launch_, st_);
            LmKernels<WT>::wo(d_, d_part_.as<float>(), n_chunks_, nc_launch_, state(b), nullptr, kv, 0, w, x(b), st_);
            LmKernels<WT>::ffn_up(d_, x(b), w, d_act_.as<float>(), st_);
            LmKernels<WT>::ffn_down(d_, d_act_.as<float>(), w, x(b), st_);
        }
    }
    void enqueue_fast_layers(int b, int kv_pos, int rope_pos) {
        for (int l = 0; l < a_.n_fast_layer; ++l) {
            const LayerW& w = fast_[l];
            KVView kv = fast_kv(l, b);
            LmKernels<WT>::qkv(d_, xf(b), w, d_cos_.as<float>(), d_sin_.as<float>(), nullptr, kv_pos, rope_pos, d_q_.as<float>(), kv, st_);
            LmKernels<WT>::wo(d_, nullptr, 0, 0, nullptr, d_q_.as<float>(), kv, kv_pos + 1, w, xf(b), st_);
            LmKernels<WT>::ffn_up(d_, xf(b), w, d_act_.as<float>(), st_);
            LmKernels<WT>::ffn_down(d_, d_act_.as<float>(), w, xf(b), st_);
        }
    }

    // smallest power-of-two chunk count covering KV length T (the host knows every frame's position in advance)
    int chunk_bucket(int T) const {
        const int need = (T + LmKernels<WT>::attn_chunk() - 1) / LmKernels<WT>::attn_chunk();
        int b = 1;
        while (b < need) b <<= 1;
        return std::min(b, n_chunks_);
    }
    void set_bucket(int T) { nc_launch_ = chunk_bucket(T); }
    // graphs for the bucket currently in nc_launch_ (captured on first use)
    void use_graphs_for_bucket() {
        const int key = nc_launch_ * 8 + (use_persist_ && persist_sampled_ ? 4 : 0) + (use_persist_ ? 2 : 0) + (use_pslow_ ? 1 : 0);
        auto it = graphs_.find(key);
        if (it == graphs_.end()) {
            g_frame_ = nullptr; g_step_ = nullptr;
            build_graphs();
            graphs_[key] = {g_frame_, g_step_};
        } else {
            g_frame_ = it->second.first; g_step_ = it->second.second;
        }
    }
    void build_graphs() {
        if (g_frame_) return;
        FS_REQUIRE(a_.num_codebooks <= 8, "the fused fast-decoder attention holds at most 8 positions");
        const int C = a_.num_codebooks;
        hipGraph_t g = nullptr;
        // (1) prefill step: embed prompt column state->step, 24 blocks, pos++/step++
        FS_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
        LmKernels<WT>::embed(d_, tok_emb_, cb_emb_, C, a_.codebook_size, d_cfg_.as<SampleCfg>(), d_prompt_.as<uint32_t>(), state(0), x(0), st_);
        enqueue_slow_layers(0);
        launch_advance(state(0), st_);
        FS_HIP(hipStreamEndCapture(st_, &g));
        FS_HIP(hipGraphInstantiate(&g_step_, g, nullptr, nullptr, 0));
        FS_HIP(hipGraphDestroy(g));
        // (2) one audio frame: x holds the embedded input of position state->pos
        FS_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
        if (use_pslow_) launch_slow_persist(pslow_args(), st_);  // 24 blocks + head as ONE persistent launch (lm_persist_slow.hip)
        else {
        enqueue_slow_layers(0);
        // audio-range head: rows [im_end, V) only (constrain_probs_to_audio, utils.rs:13-16)
        LmKernels<WT>::head(d_, x(0), norm_w_, slow_head_w(), slow_head_s(), n_audio_, d_logits_slow_.as<float>(), st_);
        }
        if (!fold_slow_sampler())
            SampleKernels<WT>::sample_slow(d_, d_logits_slow_.as<float>(), n_audio_, d_cfg_.as<SampleCfg>(), d_rng_.as<RngState>(), state(0),
                                           x(0), xf(0), st_, d_hid_slot_.as<float*>());
        if (use_persist_) launch_fast_persist(persist_args(), persist_sampled_, st_);
        else
        for (int cbi = 0; cbi < C; ++cbi) {
            enqueue_fast_layers(0, cbi, cbi);
            LmKernels<WT>::head(d_, xf(0), fast_norm_w_, fast_out_w_, fast_out_s_, a_.codebook_size, d_logits_fast_.as<float>(), st_);
            SampleKernels<WT>::sample_fast(d_, d_logits_fast_.as<float>(), cbi, C, a_.codebook_size, d_cfg_.as<SampleCfg>(),
                                           d_rng_.as<RngState>(), rp_, state(0), fast_emb_, xf(0), tok_emb_, cb_emb_, x(0),
                                           d_out_.as<uint32_t>(), out_cap_, st_);
        }
        FS_HIP(hipStreamEndCapture(st_, &g));
        FS_HIP(hipGraphInstantiate(&g_frame_, g, nullptr, nullptr, 0));
        FS_HIP(hipGraphDestroy(g));
    }

    fs_model_args a_;
    fs_token_cfg t_;
    int device_, B_;
    ModelDims d_;
    int n_audio_ = 0;
    bool generic_ = false;  // DualAR token layout with <|im_end|> away from the semantic range (see the constructor)
    bool legacy_ = false;
    DevBuf d_legacy_head_;
    hipStream_t st_ = nullptr;
    bool loaded_ = false;
    // weights
    std::vector<TensorDesc> tensors_;
    size_t arena_bytes_ = 0, o_tok_emb_ = 0, o_cb_emb_ = 0, o_fast_emb_ = 0, o_norm_ = 0, o_fast_norm_ = 0;
    std::pair<size_t, size_t> o_out_ = {0, 0}, o_fast_out_ = {0, 0};
    std::vector<std::array<size_t, 10>> o_slow_, o_fast_;
    DevBuf arena_;
    std::vector<LayerW> slow_, fast_;
    const void *tok_emb_ = nullptr, *cb_emb_ = nullptr, *fast_emb_ = nullptr, *out_w_ = nullptr, *fast_out_w_ = nullptr;
    const float *norm_w_ = nullptr, *fast_norm_w_ = nullptr, *out_s_ = nullptr, *fast_out_s_ = nullptr;
    DevBuf d_cos_, d_sin_;
    // KV
    int max_pages_ = 0, n_pages_ = 0, out_cap_ = 0, n_chunks_ = 0, nc_launch_ = 1;
    std::map<int, std::pair<hipGraphExec_t, hipGraphExec_t>> graphs_;  // attention chunk bucket -> (frame, prefill-step) graphs
    std::map<int, hipGraphExec_t> multi_graphs_;                       // same key -> kFramesPerGraph frames of the two persistent launches in ONE graph
    // Measured with the launch-boundary stamps of FISHRT_PERSIST_PROF (s_memrealtime, one clock for all kernels): the slow kernel's finish ->
    // the fast kernel's entry inside a graph is 1.9 us, the fast kernel's finish -> the NEXT graph launch's slow kernel 7.2-7.3 us.  A graph
    // that holds several frames pays the 1.9 us edge between them (the kernels read positions / epochs from device memory, so a frame's two
    // launches are the same nodes every time; frames past <|im_end|> are no-ops on the device as in the per-frame replay).
    static int frames_per_graph() {
        static const int v = [] { const char* e = getenv("FISHRT_FRAMES_PER_GRAPH"); const int n = e ? atoi(e) : 8; return n < 0 ? 1 : (n > 32 ? 32 : n); }();  // 0: no graph, the two launches of every frame directly (experiment)
        return v;
    }
    hipGraphExec_t multi_frame_graph() {  // for the bucket currently in nc_launch_; the persistent 2-launch frame only
        const int key = nc_launch_ * 8 + (persist_sampled_ ? 4 : 0);
        auto it = multi_graphs_.find(key);
        if (it != multi_graphs_.end()) return it->second;
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        FS_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
        for (int f = 0; f < frames_per_graph(); ++f) {
            launch_slow_persist(pslow_args(), st_);
            launch_fast_persist(persist_args(), persist_sampled_, st_);
        }
        FS_HIP(hipStreamEndCapture(st_, &g));
        FS_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        FS_HIP(hipGraphDestroy(g));
        multi_graphs_[key] = ge;
        return ge;
    }
    size_t page_elems_ = 0;
    DevBuf kv_pool_, fast_pool_, d_page_table_, d_zero_table_;
    std::vector<int> free_pages_, seq_len_, fast_len_;
    std::vector<std::vector<int>> seq_pages_;
    // activations / state
    DevBuf d_x_, d_xf_, d_q_, d_part_, d_act_, d_logits_slow_, d_logits_fast_, d_state_, d_cfg_, d_rng_, d_prompt_, d_out_;
    DevBuf d_rp_mask_, d_rp_seen_, d_rp_ring_, d_rp_meta_;
    DevBuf d_pack_, d_edges_, d_ctl_, d_qkv0_;  // persistent fast decoder (+ the layer-0 qkv table of the codebook passes 1..7)
    DevBuf d_fscl_, d_sscl_;           // FS_FP8 handles: row scales of the persistent kernels' images
    DevBuf d_cap_;                     // fs_lm_debug_capture
    int cap_frames_ = 0;
    DevBuf d_hidden_, d_hid_slot_;     // generate_blocking_with_hidden: [out_cap][dim] rows + the pointer cell the captured graphs read
    bool persist_ok_ = false, use_persist_ = false, pslow_ok_ = false, use_pslow_ = false, persist_sampled_ = false;
    int batch_rows_ = 0, batch_row_ = 0;  // generate_batch_sequential: batch sampler semantics for the row being generated
    // continuous-batching session: per-slot remaining iterations (-1 = empty), host copy of the slot states, scratch KV page of empty slots
    bool sess_active_ = false, sess_rows_ = false, sess_sampled_ = false;  // sess_rows_: FS_SESSION_ROWS (slots on the request-row kernels)
    int sess_R_ = 0, sess_adds_ = 0, sess_budget_tmp_ = 0;
    uint64_t sess_seed_ = 0;
    std::unique_lock<PersistLock> sess_plock_;
    std::vector<int> sess_left_, sess_pos_;
    std::vector<SeqState> sess_hs_;
    int sess_scratch_ = 0;
    uint64_t sess_released_frames_ = 0;
    struct PendingAdd { int slot = -1, L = 0, n_iter = 0, order = 0; std::vector<uint32_t> prompt; };
    std::vector<uint32_t> sess_stage_prompts_;
    std::vector<int> sess_stage_rows_;
    std::vector<PendingAdd> sess_queue_, sess_flight_;  // joining requests: queued by session_add / prefill launched (flush_pending)
    SeqState sess_stage_ = {};
    hipStream_t st_pf_ = nullptr;
    hipEvent_t ev_pf_ = nullptr;
    DevBuf d2_pfx_, d2_pfq_, d2_pfslab_, d2_pfa_, d2_pfa2_, d2_pfss_, d2_pfc_, d2_pfpart_, d2_prompt_;  // row buffers of the session's prefill stream
    DevBuf d_spack_, d_hpack_, d_snorms_, d_sedges_, d_sctl_;  // persistent slow transformer
    DevBuf d_pfx_, d_pfq_, d_pfslab_, d_pfa_, d_pfa2_, d_pfss_, d_pfc_, d_pfpart_;  // MFMA row-path activations (kRowsCap rows)
    DevBuf d_bprompt_;  // static batch: all left-padded prompts [B][C + 1][Lmax] (group prefill)
    DevBuf d_xfrows_, d_lrows_, d_lfast_, d_fast_state_, d_fast_table_, d_rwords_;  // static-batch generator
    DevBuf d_xchg_, d_epoch_;  // folded decode steps: split-K exchange units, {step epoch, timeouts}
    bool rows_par_ = false;  // this batch / session samples with the block-parallel row samplers
    int ld_slow_ = 0, down_split_ = 4;
    bool fold_off_ = false;  // set after a reported exchange timeout of the folded decode step: the handle keeps the slab path from then on
    bool batch_warm_ = false;
    std::map<int, hipGraphExec_t> batch_graphs_;
    RepPenState rp_ = {};
    DevBuf d_rimg_, d_rhimg_, d_redges_s_, d_redges_f_, d_rctl_s_, d_rctl_f_, d_rlogits_, d_rcfg_, d_rrng_, d_rbudget_;  // request rows (lm_persist_rows.hip)
    DevBuf d_rrp_mask_, d_rrp_seen_, d_rrp_ring_, d_rrp_meta_, d_rcap_, d_rpairs_, d_rones_;
    void* h_pin_ = nullptr;
    hipGraphExec_t g_frame_ = nullptr, g_step_ = nullptr;
    hipEvent_t ev_[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_batch_[2] = {nullptr, nullptr};  // decode batches in flight (generate)
    fs_gen_stats stats_ = {};
};

void fp8_quantize_rows(int device, const float* w, int64_t rows, int64_t cols, uint8_t* q_out, float* scales_out) {
    FS_REQUIRE(rows >= 1 && cols >= 1 && rows < (1ll << 31), "bad matrix shape");
    FS_HIP(hipSetDevice(device));
    DevBuf src, q, sc;
    src.alloc((size_t)rows * cols * 4); q.alloc((size_t)rows * cols); sc.alloc((size_t)rows * 4);
    FS_HIP(hipMemcpy(src.p, w, (size_t)rows * cols * 4, hipMemcpyHostToDevice));
    launch_quant_rows_fp8(q.as<uint8_t>(), sc.as<float>(), src.as<float>(), rows, cols, 1, 0, nullptr);
    FS_HIP(hipDeviceSynchronize());
    FS_HIP(hipMemcpy(q_out, q.p, (size_t)rows * cols, hipMemcpyDeviceToHost));
    FS_HIP(hipMemcpy(scales_out, sc.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
}
void fp8_decode_table(int device, float* out) {
    FS_HIP(hipSetDevice(device));
    DevBuf t;
    t.alloc(16 * 256 * 4);
    launch_fp8_decode_table(t.as<float>(), nullptr);
    FS_HIP(hipDeviceSynchronize());
    FS_HIP(hipMemcpy(out, t.p, 16 * 256 * 4, hipMemcpyDeviceToHost));
}

LMBase* make_lm(const fs_model_args& a, const fs_token_cfg& t, int device, fs_dtype dtype, int max_batch) {
    if (dtype == FS_BF16) return new LM<bf16_t>(a, t, device, max_batch);
    if (dtype == FS_F32) return new LM<float>(a, t, device, max_batch);
    if (dtype == FS_FP8) return new LM<fp8_t>(a, t, device, max_batch);
    throw Error("unknown dtype");
}

}  // namespace fs
