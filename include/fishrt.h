/* fishrt.h -- C ABI of the MI355X-native Fish-Speech hot path (libfishrt.so).
 *
 * Drop-in boundary: the entry points below are what a Rust `fish_speech_core`-shaped shim
 * (or the PyO3 crate, or ctypes) binds INSTEAD of the candle-backed implementation.  Each entry cites the
 * reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions (mirror of the reference's library API):
 *  - every call returns int status, 0 = ok; on failure a thread-local message is available from
 *    fs_last_error()  (== the `Result<_, candle_core::Error>` / PyRuntimeError string, fish_speech_python/src/utils.rs:6-20);
 *  - all I/O buffers are caller-owned HOST memory, C-contiguous, same shapes as the numpy arrays of the PyO3 API
 *    (fish_speech_python/src/lm.rs:72-145, codec.rs:73-114); handles own all device memory;
 *  - a handle is single-threaded (one in-flight call): mirrors `&mut DualARTransformer`
 *    (server/lib/state.rs:13 serialises with a tokio::Mutex); distinct handles (one per GPU) are independent;
 *  - no CPU fallback exists: creating a handle without a usable gfx950 device fails.
 */
#ifndef FISHRT_H
#define FISHRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_OK 0
#define FS_ERR 1

/* weight / KV-cache storage type.  Activations and accumulation are always f32.
 * FS_BF16 == the reference's CUDA dtype choice (fish_speech_core/src/bin/llama_generate.rs:179-182),
 * FS_F32  == its CPU dtype (used here for token-exact parity runs against the f32 oracle),
 * FS_FP8  == no reference counterpart (SURVEY.md §8 configs[4]): the Linear weights (wqkv, wo, w1/w3, w2, output,
 *            fast_output) are quantised at load time to OCP e4m3fn bytes with one f32 absmax scale per output row
 *            (scale = amax/448); embeddings and the KV cache stay bf16, norm vectors f32, all accumulation f32. */
typedef enum { FS_F32 = 0, FS_BF16 = 1, FS_FP8 = 2 } fs_dtype;

/* fish_speech_core/lib/lm/dual_ar.rs:57-81  (BaseModelArgs; training-only fields dropped) */
typedef struct fs_model_args {
    int32_t dim, n_layer, n_fast_layer, n_head, n_local_heads, head_dim;
    int32_t intermediate_size, num_codebooks, codebook_size, vocab_size, max_seq_len;
    float norm_eps, rope_base;
    int32_t tie_word_embeddings;
} fs_model_args;

/* fish_speech_core/lib/lm/dual_ar.rs:17-23  (TokenConfig).  has_semantic_end = 1 for Fish 1.5 (Some(end)), 0 for <= 1.4 */
typedef struct fs_token_cfg {
    uint32_t im_end_id, pad_id, semantic_start_id, semantic_end_id;
    int32_t has_semantic_end;
} fs_token_cfg;

/* fish_speech_core/lib/lm/sampling/mod.rs:29-34  (SamplingArgs) */
typedef struct fs_sampling {
    double temp, top_p;
    uint64_t top_k;
    float repetition_penalty;
} fs_sampling;

typedef struct fs_lm fs_lm_t;
typedef struct fs_codec fs_codec_t;

/* optional per-frame streaming callback (invoked on the calling thread, in frame order):
 * codes[num_codebooks] of frame `frame_idx`.  Return non-zero to stop generation early. */
typedef int (*fs_frame_cb)(void* user, size_t frame_idx, const uint32_t* codes);

const char* fs_last_error(void);
/* library/ABI version and the gfx target the kernels were compiled for ("gfx950") */
const char* fs_version(void);
/* number of visible HIP devices (0 when none: every create call then fails loudly) */
int fs_device_count(void);

/* ---- DualARTransformer ---------------------------------------------------------------------------------- */

/* FS_FP8 storage format (no reference counterpart; SURVEY.md §8 configs[4]).  fs_fp8_quantize_rows runs the loader's device
 * quantiser on a host f32 matrix [rows, cols]: scales_out[r] = amax_r / 448 (1 for an all-zero row), q_out = e4m3fn bytes of
 * w / scale (round-to-nearest-even, saturating; never NaN) -- what fs_lm_load_* stores for every Linear weight of an FS_FP8
 * handle.  fs_fp8_decode_table returns, for each of the 16 byte positions of a lane's 16-byte weight load, the f32 value the
 * GEMV kernels give each of the 256 byte codes (out[slot * 256 + code]). */
int fs_fp8_quantize_rows(int device_id, const float* w, int64_t rows, int64_t cols, uint8_t* q_out, float* scales_out);
int fs_fp8_decode_table(int device_id, float* out);

/* Device self-test of an internal building block, by name (diagnostics; used by the parity tests): "pf_reduce" = the multi-value
 * wave reductions of the persistent fast-decoder kernel (csrc/lm_persist.hip) against host sums.  0 = passed. */
int fs_selftest(int device_id, const char* what);
/* The static-batch sampler (BatchedLogitsProcessor::sample, sampling/mod.rs:77-109) on caller-provided logits f32 [B, n]
 * (n <= 4096): out[b] = token of row b for sample() call number `call_index` of a request seeded with `seed` (row b draws from
 * the child StdRng seeded with master u64 number call_index * B + b; temp <= 1e-7 -> first-max argmax).  Diagnostics / parity tests. */
int fs_selftest_sample_rows(int device_id, const float* logits, int B, int n, const fs_sampling* s, uint64_t seed, int call_index,
                            uint32_t* out);

/* Self-test of a LOADED handle, by name.  "persist": the persistent decode kernels of this binary (csrc/lm_persist.hip, lm_persist_slow.hip:
 * pinned weight registers, loads issued outside the compiler's wait-count bookkeeping) against the per-node kernels on the handle's own
 * weights -- 4 greedy frames on the persistent path with the decision capture armed, then one teacher-forced per-node step
 * (fs_lm_forward_generate / _fast) compared with the captured logits of the first k_slow_persist step and the first k_fast_persist pass
 * (bound 2e-2 x max(1, max |logit|): loose for summation order, tight for a wrong weight register).  A deployment runs it once after a
 * toolchain change; 0 = passed, the message names the remedy otherwise.  Clears the handle's KV caches.  No reference counterpart. */
int fs_lm_selftest(fs_lm_t* lm, const char* what);
/* Decision capture of the persistent batch-1 decode path (diagnostics / parity tests; no reference counterpart).  After
 * fs_lm_debug_capture(lm, n) every fs_lm_generate call that takes the persistent fast decoder records, for its first n generator
 * iterations, what each of the 9 decisions of a frame saw and chose: fs_lm_debug_read copies f32 [n][9][2048]; row 0 = the slow
 * decision (entries [0, V - im_end): logits over the audio range after the <|im_end|> mask; entry 2047: the picked index), rows
 * 1 + c = codebook c (entries [0, 1024): logits after the repetition penalty; entry 1024: the picked code).  n = 0 switches it off.
 * The parity tests replay these rows through the CPU sampler: same logits, same StdRng stream => same picks, token for token. */
int fs_lm_debug_capture(fs_lm_t* lm, int n_frames);
int fs_lm_debug_read(fs_lm_t* lm, float* out, int n_frames);
/* test hook: the cached K / V rows [t0, t0 + n) of slow layer `layer` of KV slot `slot` (0 for batch-1 calls) as f32 [n][n_local_heads][head_dim]
 * -- the parity tests hand the GPU's own cache entries to the CPU oracle (tests/test_kv_forced_gpu.py) */
int fs_lm_debug_read_kv(fs_lm_t* lm, int slot, int layer, int t0, int n, float* k_out, float* v_out);
/* the same record of request `row` of the last fs_lm_generate_multi call (every request row is captured) */
int fs_lm_debug_read_row(fs_lm_t* lm, int row, float* out, int n_frames);

/* DualARTransformer::load (dual_ar.rs:460-529) is split in create + one of the load calls.
 * max_batch: number of independent sequences (KV caches) the handle can hold (1 for the single-batch generator). */
int fs_lm_create(const fs_model_args* args, const fs_token_cfg* tok, int device_id, fs_dtype dtype, int max_batch,
                 fs_lm_t** out);
void fs_lm_destroy(fs_lm_t* lm);
/* tensors by the reference's names (dual_ar.rs:125-156,219-223,415-419,466-511); bf16 or f32 safetensors */
int fs_lm_load_safetensors(fs_lm_t* lm, const char* path);
/* deterministic synthetic weights at the same names/shapes (spec: csrc/fs_synth.h; no checkpoint exists offline) */
int fs_lm_load_synthetic(fs_lm_t* lm, uint64_t seed);

/* DualARTransformer::forward_generate (dual_ar.rs:574-635).  toks: u32 [B, num_codebooks+1, L].
 * logits_out: f32 [B, vocab_size] or NULL; hidden_out: f32 [B, dim] (PRE-norm hidden of the last position) or NULL.
 * The pad mask argument of the reference is accepted nowhere because the reference ignores it (dual_ar.rs:589-615). */
int fs_lm_forward_generate(fs_lm_t* lm, const uint32_t* toks, int B, int L, int input_pos, float* logits_out,
                           float* hidden_out);
/* DualARTransformer::forward_generate_fast (dual_ar.rs:638-673).  x: f32 [B, dim]; logits_out: f32 [B, codebook_size] */
int fs_lm_forward_generate_fast(fs_lm_t* lm, const float* x, int B, int input_pos, float* logits_out);
/* fast_embeddings.forward (dual_ar.rs:447; used at generate/single_batch.rs:176-182): out f32 [n, dim] */
int fs_lm_fast_embed(fs_lm_t* lm, const uint32_t* ids, int n, float* out);
int fs_lm_clear_fast_layer_caches(fs_lm_t* lm);            /* dual_ar.rs:675-679 */
int fs_lm_clear_slow_layer_caches(fs_lm_t* lm);            /* dual_ar.rs:681-685 */
int fs_lm_clear_slow_caches_until(fs_lm_t* lm, int pos);   /* dual_ar.rs:687-693 (NOT inclusive) */
int fs_lm_curr_kv_size(fs_lm_t* lm);                       /* dual_ar.rs:695-700; < 0 on error */

/* generate_blocking (generate/single_batch.rs:217-324): batch-1 generator, audio_only = true.
 * prompt: u32 [num_codebooks+1, L].  codes_out: u32 [num_codebooks, cap] row-major with row stride `cap`;
 * *n_frames receives the number of frames written (<= cap).  The KV cache is NOT cleared first (the caller
 * owns cache lifetime exactly as with the reference: fish_speech_python/src/lm.rs:94,131-135).
 * seed: seeds the sampler RNG (the reference draws rand::random(), single_batch.rs:46).
 * Extensions / restrictions relative to the reference, stated here so that nobody takes them for parity:
 *   - sampling->top_k == 0 means "no top-k" when temp > 0.  The reference has no such setting: `select_nth_unstable_by(0, ..)` yields an
 *     empty candidate set (candle's single path then fails in WeightedIndex::new, the batch path falls to `unwrap_or(0)`,
 *     sampling/mod.rs:57-75).  With temp == 0 top_k is never looked at (argmax first), there as here.
 *   - both branches of constrain_probs_to_audio / rescale_semantic_tokens are implemented (generate/utils.rs:13-30,45-52): with
 *     im_end_id + 1 == semantic_start_id (Fish 1.5) the slow head reads the contiguous rows [im_end, V); otherwise (generic DualAR token
 *     layout) the candidates are [im_end] ++ [semantic_start, V) -- literally, so control tokens behind the range (<|im_end|> itself
 *     included) stay candidates -- gathered into one head image at load time.  All generation paths take either layout.
 * flags: FS_GEN_IGNORE_EOS masks <|im_end|> (bench-only, fixed-length runs: SURVEY.md §8d).
 *        FS_GEN_NO_PERSIST keeps the whole frame on the per-node graph path.  By default a call on a bf16 or fp8 handle with the Fish geometry
 *        runs a frame as TWO persistent launches (csrc/lm_persist_slow.hip: the 24 slow blocks + head; csrc/lm_persist.hip: the slow-token
 *        decision, the 8 codebook passes with weights resident in VGPRs / LDS and their 8 decisions -- greedy, or top-k / top-p sampled
 *        in-launch when 0 < top_k <= 256 -- with in-launch hand-offs); those launches need all 256 CUs of the device, so only one
 *        generate call per GPU uses them at a time (a concurrent call on another handle silently takes the per-node path). */
#define FS_GEN_IGNORE_EOS 1u
#define FS_GEN_NO_PERSIST 2u
/* measurement mode (bench.py `roofline.kernels`): the two persistent kernels of every decode frame are launched one by one with a HIP event
 * in front of, between and behind them instead of as one graph replay; same kernels, same arguments, same tokens -- only the host paces the
 * frames, so decode_ms of such a call is not a throughput figure.  Ignored when the call does not take both persistent kernels. */
#define FS_GEN_TIME_KERNELS 4u
int fs_lm_generate(fs_lm_t* lm, const uint32_t* prompt, int L, int max_new_tokens, const fs_sampling* sampling,
                   uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, fs_frame_cb cb,
                   void* cb_user);
/* generate_blocking_with_hidden (generate/single_batch.rs:217-306; caller: server/lib/handlers/speech.rs:27-48).
 * hidden_out != NULL <=> collect_hidden_states = true: f32 [hidden_cap, dim] receives the slow transformer's pre-norm hidden state
 * (the `hidden_states` of forward_generate, dual_ar.rs:629-634) of EVERY generator iteration in order -- the first frame, every later
 * frame, and the terminating <|im_end|> iteration whose codes are not emitted (:264-266), so *n_hidden is *n_frames or
 * *n_frames + 1; hidden_cap >= max_new_tokens - L + 2 always suffices.  hidden_out == NULL is fs_lm_generate (the reference
 * returns None). */
int fs_lm_generate_with_hidden(fs_lm_t* lm, const uint32_t* prompt, int L, int max_new_tokens, const fs_sampling* sampling,
                               uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap, size_t* n_frames, float* hidden_out,
                               size_t hidden_cap, size_t* n_hidden, fs_frame_cb cb, void* cb_user);

/* generate_static_batch (generate/static_batch.rs:282-390), audio_only = true: n prompts [num_codebooks+1, L_i]
 * (concatenated in `prompts`, lengths in `lens`), left-padded with <|im_end|>/0 as static_batch.rs:68-111,
 * lock-step decode, ragged outputs: codes_out u32 [n, num_codebooks, cap], n_frames[n].
 * bf16 / fp8 handles with n <= min(max_batch, 256): the n sequences are the rows of every GEMM (weights streamed once per step
 * for the whole batch) and the prompts are prefilled as group passes; otherwise (f32 handles, n > 256) the rows are generated
 * one after another on KV slot 0 -- with the BatchedLogitsProcessor semantics of the lock-step path (sampling/mod.rs:77-109: temp <= 1e-7
 * -> first-max argmax; else row b draws sample() call c from the child StdRng seeded with the master's u64 number c * n + b), so the
 * outputs are those of the lock-step batch. */
int fs_lm_generate_batch(fs_lm_t* lm, const uint32_t* prompts, const int* lens, int n, int max_new_tokens,
                         const fs_sampling* sampling, uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap,
                         size_t* n_frames);

/* The same call with generate_static_batch's full signature (generate/static_batch.rs:282-390): `audio_only` as an argument and the second
 * return value, `Vec<Vec<bool>>` of BatchPosition::is_audio (:229: slow token >= semantic_start), as is_audio_out u8 [n, cap] (nullable;
 * is_audio_out[i * cap + f] for f < n_frames[i]).  With audio_only = 1 every returned position is audio except a row's FIRST position when its
 * slow token was <|im_end|> (the first position is returned unconditionally, :305-316, with zero codes, :230-233).  audio_only = 0 -- slow
 * token sampled over the full vocabulary, rows never terminate, outputs keep the slow-token row (:132-141,156-173,361-364) -- is NOT
 * implemented and returns an error; the reference's server always passes true (server/lib/handlers/speech.rs:80-86). */
int fs_lm_generate_static_batch(fs_lm_t* lm, const uint32_t* prompts, const int* lens, int n, int max_new_tokens, int audio_only,
                                const fs_sampling* sampling, uint64_t seed, uint32_t flags, uint32_t* codes_out, size_t cap,
                                size_t* n_frames, uint8_t* is_audio_out);

/* R concurrent batch-1 requests on ONE device (round 4; the reference's only multi-request generator is the lock-step static batch,
 * generate/static_batch.rs:117-274, which changes the sampler semantics; its server serialises requests behind one mutex,
 * server/lib/state.rs:12-29).  Request i IS generate_blocking(prompt_i, max_new_tokens[i], samplings[i]) with sampler seed seeds[i] on a
 * cleared cache (generate/single_batch.rs:76-214: its own KV, repetition-penalty window, RNG stream, <|im_end|> / budget rules) -- the
 * tokens of its own fs_lm_generate call (the kernels sum in another order, so greedy tokens may differ where two candidates are within
 * rounding of each other).  prompts: the n prompts u32 [C+1, L_i] concatenated; codes_out u32 [n, C, cap]; n_frames[n].
 * bf16 handles with the Fish 1.5 geometry and token layout, 2 <= n <= min(8, max_batch), and either every request greedy (temp == 0) or
 * every request inside the in-launch sampler (temp > 0, 0 < top_k <= 256: the server default): every decode frame is ONE persistent
 * launch of the slow transformer for all requests (csrc/lm_persist_rows.hip: the weights are streamed once per frame, the requests are
 * matrix-core columns) plus one fast-decoder launch per group of <= 4 requests, each request deciding on its own logits with its own
 * StdRng stream and repetition-penalty window.  Any other handle / sampler mix / n runs the requests one after the other through
 * fs_lm_generate.  The slow KV cache of every slot is cleared first. */
int fs_lm_generate_multi(fs_lm_t* lm, const uint32_t* prompts, const int* lens, int n, const int* max_new_tokens,
                         const fs_sampling* samplings, const uint64_t* seeds, uint32_t flags, uint32_t* codes_out, size_t cap,
                         size_t* n_frames);

/* Capability query for schedulers (fishrt/server.py): *supported = 1 when fs_lm_generate_multi would serve these n requests with these
 * sampler settings on the request-row kernels (one persistent launch group per frame), 0 when it would run them one after the other
 * through fs_lm_generate (handle dtype / token layout / n / sampler mix outside the row kernels) -- in which case a lock-step
 * fs_lm_generate_batch streams the weights once per step for all of them and is the better multi-request path.  Static property of the
 * handle and the settings: it does not look at whether another call currently holds the device's persistent kernels. */
int fs_lm_rows_supported(fs_lm_t* lm, int n, const fs_sampling* samplings, int* supported);

/* ---- replica start-up (no reference counterpart: the reference has no distributed layer).  The handle's device
 * weight arena -- every checkpoint tensor in the handle's storage type, laid out by its tensor plan -- is a pure function of (model args,
 * token config, dtype, checkpoint), so N replicas need ONE checkpoint read: rank 0 loads, fs_lm_weights_arena() gives every rank the
 * device pointer + size to hand to ncclBroadcast (torch.distributed.broadcast: fishrt/fanout.py broadcast_weights), and the receivers
 * call fs_lm_weights_adopt() (marks the handle loaded and builds the load-time derived data: persistent-kernel weight images etc.). */
int fs_lm_weights_arena(fs_lm_t* lm, void** dev_ptr, size_t* bytes);
int fs_lm_weights_adopt(fs_lm_t* lm);

/* ---- replica fan-out over RCCL / xGMI (BASELINE.json north_star: "batch-sharded across the 8 GPUs of one node with
 * RCCL over xGMI only for multi-request fan-out").  No reference counterpart: the reference serves one model behind one mutex
 * (server/lib/state.rs:12-29) and has no distributed layer.  One process per GPU; each holds its own fs_lm_t (full replica) and ONE
 * communicator.  Request i goes to rank i mod world; nothing on the per-token path crosses GPUs, so these are all the collectives there
 * are: (1) the weight arena from the rank that read the checkpoint (ncclBroadcast in 256 MB pieces, then fs_lm_weights_adopt on the
 * receivers), (2) the packed prompt batch, (3) the end-of-run fan-in of the code arrays (ncclAllGather), plus a barrier and small f64
 * reductions for the job's clock and frame counters (ncclAllReduce).  librccl is bound with dlopen at the first call: single-GPU hosts
 * never load it.  Host buffers are caller-owned, as everywhere in this header; calls block until the collective has completed.
 *   bring-up: rank 0 calls fs_comm_unique_id and ships the 128 bytes to the other ranks over the host's own channel (environment,
 *   file, TCP store); every rank then calls fs_comm_create(id, rank, world, device) -- collective, like ncclCommInitRank. */
#define FS_COMM_ID_BYTES 128
typedef struct fs_comm fs_comm_t;
int fs_comm_unique_id(uint8_t id_out[FS_COMM_ID_BYTES]);
int fs_comm_create(const uint8_t id[FS_COMM_ID_BYTES], int rank, int world, int device_id, fs_comm_t** out);
void fs_comm_destroy(fs_comm_t* comm);
int fs_comm_rank(fs_comm_t* comm);   /* -1 on a null handle */
int fs_comm_world(fs_comm_t* comm);
/* every rank enters; returns when all have (an all-reduce of a rank count + the stream sync behind it) */
int fs_comm_barrier(fs_comm_t* comm);
/* in place on a host array of n <= 4096 doubles; op: 0 sum, 1 max, 2 min (the job's wall time is the MAX over ranks, its frames the SUM) */
int fs_comm_all_reduce_f64(fs_comm_t* comm, double* vals, int n, int op);
/* (1) `lm` on rank `src` is loaded; on every other rank it was created with the same model args / token config / dtype and is NOT loaded:
 * after the call every rank's handle is ready (receivers ran fs_lm_weights_adopt).  Arena sizes are compared across ranks first; a
 * mismatch is an error on EVERY rank before a byte moves.  *bytes_moved (nullable) = the arena size. */
int fs_comm_broadcast_weights(fs_comm_t* comm, fs_lm_t* lm, int src, size_t* bytes_moved);
/* (2) the packed prompt batch: dims = {n_requests, num_codebooks + 1, Lmax}; packed u32 [n_requests][C+1][Lmax] (rows left-aligned,
 * zero-padded), lens i32 [n_requests].  Receivers first learn the shape (fs_comm_broadcast_prompt_dims fills dims from rank src), size
 * their buffers, then every rank calls fs_comm_broadcast_prompts with the same dims. */
int fs_comm_broadcast_prompt_dims(fs_comm_t* comm, int64_t dims[3], int src);
int fs_comm_broadcast_prompts(fs_comm_t* comm, uint32_t* packed, int32_t* lens, const int64_t dims[3], int src);
/* (3) fan-in: this rank's codes u32 [B][C][N] (its requests, padded to N frames) and n_frames i32 [B] -> on EVERY rank
 * codes_all u32 [world][B][C][N] and n_frames_all i32 [world][B].  B, C, N must be the same on every rank (pad the last shard). */
int fs_comm_all_gather_codes(fs_comm_t* comm, const uint32_t* codes, const int32_t* n_frames, int B, int C, int N, uint32_t* codes_all,
                             int32_t* n_frames_all);

/* ---- continuous batching (no reference counterpart: the reference server serialises requests behind one mutex, server/lib/state.rs:12-29,
 * or runs lock-step batches, generate/static_batch.rs:282-390; SURVEY.md section 8 f-4 asks for a scheduler that replaces the mutex).
 * A session turns the max_batch rows of the static-batch decode step into independent request SLOTS: a request's prompt is prefilled on
 * the matrix-core row path on a second stream while the other slots keep stepping, it joins between two steps once that has finished,
 * and leaves when done; every step streams the weights once for all live slots.  A slot behaves exactly like row 0 of a ONE-prompt fs_lm_generate_batch call: no left padding (own positions
 * and KV pages), first frame emitted unconditionally, BatchedLogitsProcessor sampling (sampling/mod.rs:77-109; repetition penalty is
 * ignored like static_batch.rs:204-206), 1 + max(0, max_new_tokens - L + 1) iterations (static_batch.rs:122), stopping early at
 * <|im_end|> or at max_seq_len.  With temp <= 1e-7 a slot's codes are independent of what the other slots do.
 * bf16 / fp8 handles with the Fish 1.5 token layout only; while a session is open the handle's other entry points fail. */
/* FS_SESSION_ROWS (round 4): the slots run on the request-row persistent kernels (csrc/lm_persist_rows.hip, see fs_lm_generate_multi) and keep
 * BATCH-1 semantics: a slot is its own generate_blocking call -- repetition penalty applied, LogitsProcessor sampling on its own StdRng stream
 * seeded `seed + the slot's admission number`, 1 + max(0, max_new_tokens - L + 1) iterations -- while requests join and leave between
 * frames; a decode frame costs one slow launch for all slots + one fast launch per 4 slots instead of the 373-node step.  bf16 Fish-1.5
 * handles with 2 <= max_batch <= 8, greedy or 0 < top_k <= 256; the device's persistent kernels are held for the session's lifetime. */
#define FS_SESSION_ROWS 8u
int fs_lm_session_begin(fs_lm_t* lm, const fs_sampling* sampling, uint64_t seed, uint32_t flags /* FS_GEN_IGNORE_EOS | FS_SESSION_ROWS */);
/* prompt u32 [C+1, L] row-major (copied); *slot = the slot taken, or -1 when all max_batch slots are busy or the KV page pool cannot hold
 * the request right now (not an error: retry after a release).  Returns once the
 * prefill is enqueued (one prefill in flight: a second add first waits for the previous one); the slot starts generating in a later step */
int fs_lm_session_add(fs_lm_t* lm, const uint32_t* prompt, int L, int max_new_tokens, int* slot);
/* run up to n_frames decode steps for all live slots (stops early when none is live); *n_active = slots still generating afterwards */
int fs_lm_session_step(fs_lm_t* lm, int n_frames, int* n_active);
/* frames of `slot` so far: codes_out u32 [C, cap] row-major (may be NULL to query only), *n_frames, *done = 1 once the slot has finished */
int fs_lm_session_poll(fs_lm_t* lm, int slot, uint32_t* codes_out, size_t cap, size_t* n_frames, int* done);
/* give the slot (and its KV pages) back */
int fs_lm_session_release(fs_lm_t* lm, int slot);
int fs_lm_session_end(fs_lm_t* lm);

/* timing of the last generate call, measured with HIP events on the handle's stream (the reference prints the
 * same quantities: single_batch.rs:233-246,291-304) */
typedef struct fs_gen_stats {
    double prefill_ms, decode_ms;     /* decode_ms covers frames 1..n-1 exactly like `start_decode` (:261) */
    uint64_t frames, prompt_tokens, graph_launches;
    uint64_t kernels_per_frame;       /* kernel launches per decode frame of this call: 266 on the per-node path (FS_GEN_NO_PERSIST), 146 with the
                                         persistent slow kernel only (sampler settings outside the in-launch sampler), 2 with both persistent
                                         kernels (greedy, and temp > 0 with 0 < top_k <= 256) */
    double slow_kernel_us, fast_kernel_us;  /* FS_GEN_TIME_KERNELS: average duration of k_slow_persist / k_fast_persist over the decode frames
                                               of this call (HIP events around every launch on the handle's stream); 0 otherwise */
} fs_gen_stats;
int fs_lm_last_stats(fs_lm_t* lm, fs_gen_stats* out);
/* the hipStream_t the handle launches on (for callers that bracket calls with their own HIP events) */
void* fs_lm_stream(fs_lm_t* lm);
/* Measurement hook (bench.py `roofline.dominant_kernel`): average duration in microseconds of ONE launch of a batch-1 decode kernel --
 * kind 0 qkv, 1 attention, 2 wo, 3 ffn_up (RMSNorm + W1||W3 GEMV + SwiGLU), 4 ffn_down -- run as a node of a captured hipGraph that cycles
 * over the slow layers' distinct weights (nothing cache-resident) at KV length kv_len, timed with HIP events on the engine stream;
 * kind 5 = average node of the fast decoder's layers over the 8 codebook positions, 6 = fast head GEMV, 7 = slow audio-range head GEMV.
 * Clears the slow KV caches. */
int fs_lm_bench_kernel(fs_lm_t* lm, int kind, int kv_len, int reps, float* us_per_launch);

/* ---- FireflyCodec ---------------------------------------------------------------------------------------- */

/* FireflyCodec::load (codec/firefly.rs:20-34) with FireflyConfig::get_config_for(1.4 | 1.5) (codec/config.rs:196-202).
 * channel_div = 1 for the real configuration; tests use 8 (channels / 8, same topology). */
int fs_codec_create(int device_id, int channel_div, fs_codec_t** out);
void fs_codec_destroy(fs_codec_t* c);
int fs_codec_load_safetensors(fs_codec_t* c, const char* path);
int fs_codec_load_synthetic(fs_codec_t* c, uint64_t seed);
/* FireflyCodec::decode (codec/firefly.rs:42-48): codes u32 [b, 8, T] (values 0..999) -> pcm f32 [b, 1, 2048*T] */
int fs_codec_decode(fs_codec_t* c, const uint32_t* codes, int b, int T, float* pcm_out);
/* FireflyCodec::encode (codec/firefly.rs:37-40) for one mono 44.1 kHz clip: LogMelSpectrogram::forward (audio/spectrogram.rs:
 * 153-158: streaming STFT n_fft 2048 / hop 512 with edge-repeating reflect padding, 160 slaney mel bins, clamp(1e-5,100).log())
 * -> FireflyEncoder::encode (codec/encoder.rs:38-42: ConvNeXt backbone, downsample x4, grouped FSQ).  codes_out: u32 [8, cap]
 * row-major, *n_frames = L = mel_frames / 4 codes per group.  (channel_div > 1 handles use a reduced backbone depth (1,1,2,1).) */
int fs_codec_encode(fs_codec_t* c, const float* pcm, int n_samples, uint32_t* codes_out, size_t cap, size_t* n_frames);
/* the (b, 1, samples) -> (b, 8, T) signature of FireflyCodec::encode (codec/firefly.rs:36-39, fish_speech_python/src/codec.rs:73-92) for
 * b clips: pcm f32 [b, stride] row-major (clip i = its first n_samples[i] samples), codes_out u32 [b, 8, cap], n_frames[b].  Every clip is
 * encoded ON ITS OWN (per-clip lengths, per-clip padding).  Deviation from the reference, on purpose: its front-end flattens whatever it is
 * given into ONE signal (audio/spectrogram.rs:33 `flatten_all`), so a batch comes back as (1, 8, L) codes of the clips glued together;
 * this entry point returns what the signature promises. */
int fs_codec_encode_batch(fs_codec_t* c, const float* pcm, int b, size_t stride, const int* n_samples, uint32_t* codes_out, size_t cap,
                          size_t* n_frames);
/* Stateful streaming decode (no reference counterpart: the reference vocodes an utterance in one piece, server/lib/handlers/speech.rs:98-129).
 * Every convolution of the 1.4+ / 1.5 codec is causal (codec/utils/mod.rs:53-62,110-122), so the chunks of ONE code sequence can be decoded
 * one after the other with the convolutions' left context carried on the device: fs_codec_stream_begin, then fs_codec_stream_decode per
 * chunk (codes u32 [8, T] row-major, T >= 16 frames; pcm_out f32 [2048 T]), then fs_codec_stream_end.  The concatenated PCM is bit-identical
 * to fs_codec_decode of the whole sequence and no frame is decoded twice.  Needs the plane data flow (precision mode 1 or 2, full-size codec);
 * the precision mode must not change inside a stream; one stream per handle. */
int fs_codec_stream_begin(fs_codec_t* c);
int fs_codec_stream_decode(fs_codec_t* c, const uint32_t* codes, int T, float* pcm_out);
int fs_codec_stream_end(fs_codec_t* c);
/* FireflyCodec.sample_rate (codec/firefly.rs:13) */
int fs_codec_sample_rate(fs_codec_t* c);
/* Arithmetic of the decode path's convolutions (no reference counterpart: the reference runs the codec in f32,
 * server/lib/utils/load.rs:161-164; acceptance bound: PCM within 1e-4 RMS of it).
 * mode 2 (default) = "f16": every matrix operand rounded once to f16 (saturating at 65504), one f16 matrix product per term, f32
 *   accumulation, f32 residual stream and epilogues -- measured 1.6e-5 RMS at signal rms 0.031 (relative 5e-4, i.e. the size of the
 *   16-bit PCM quantisation step the server's WAV output applies anyway);
 * mode 1 = "bf16x3": each f32 operand split into bf16 hi + lo, three bf16 matrix products per term -- 3e-7 RMS, ~1.5x the time of mode 2;
 * mode 0 = exact f32 products on the f32 matrix cores (4e-8 of the oracle, ~4.5x the time of mode 2).  The encoder always uses mode 0. */
int fs_codec_set_precision(fs_codec_t* c, int mode);
int fs_codec_precision(fs_codec_t* c);
/* Range guard of mode 2 (validation tool, off by default; no reference counterpart).  f16 operands saturate beyond +-65504, vanish below
 * 2^-24, and carry a RELATIVE error (2^-11) against an absolute acceptance bound; synthetic N(0, 1 / fan_in) convs at the test signal's level
 * are far from all three, a weight-normed HiFi-GAN checkpoint or loud material has not been seen by this code.  With the check on,
 * fs_codec_decode in mode 2 (a) runs range-counting twins of the conversion kernels, (b) decodes the same codes in mode 1 (bf16x3: f32
 * exponent range, 2^-17 relative) too and takes the RMS difference of the two PCMs, and (c) returns the mode-1 PCM when an activation or
 * weight operand saturated or that difference exceeds 5e-5 (half the 1e-4 bound), else the mode-2 PCM.
 * fs_codec_range_stats: out5 = {activation operands saturated, flushed to zero (cumulative since the check was switched on), f16 weights
 * saturated, flushed (of the loaded checkpoint), decode calls answered from mode 1}; *last_pcm_rms_diff (nullable) = the RMS difference
 * of the last checked call.  Flushed operands are reported, not acted on (SiLU tails put a few hundred activations per decode below 2^-24
 * with any weights, < 6e-8 absolute each).  Streamed chunks (fs_codec_stream_decode) are counted only.  ~3x the cost of a plain call. */
int fs_codec_set_range_check(fs_codec_t* c, int on);
int fs_codec_range_stats(fs_codec_t* c, uint64_t* out5, double* last_pcm_rms_diff);

#ifdef __cplusplus
}
#endif
#endif /* FISHRT_H */
