#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE.json configs[1]:
Fish-Speech-1.5 (synthetic weights at the true shapes), bf16, batch=1, default-voice-sized prompt (368 positions),
256 generated codec frames per request, greedy-free sampling disabled EOS (fixed length).

A "step" is ONE REQUEST through the hot path (prefill of the prompt + 256 decode frames) on each rank; at N GPUs every
rank serves its own independent request stream (replica fan-out, weak scaling, no data-path collective: SURVEY.md §8e).
value = frames produced by all ranks / wall time of the timed region (prefill included), max over ranks.
`python bench.py --gpus N` starts its own ranks (torch.distributed.run) when no launcher environment is present; under the
driver's `python -m torch.distributed.run ... bench.py --gpus N` it uses the given ranks.  `--config 3` runs BASELINE.json
configs[3] instead (256 requests sharded i mod N, static batches of 32 per GPU, strong scaling).  Without a visible MI355X the
ranks exercise the control path only (dry run, value null): libfishrt has no CPU path.

Extra objects on the JSON line:
  roofline     -- HBM roofline of the decode FRAME (the unit of the hot loop, replayed from hipGraphs of 8 frames: the two persistent launches
                  k_slow_persist + k_fast_persist; 266 kernels with --no-persistent):
                  achieved = B_frame(T_avg) / t_frame, t_frame from HIP events recorded on the engine's own stream
                  (fs_lm_last_stats); B_frame is SURVEY.md §8(d)'s algorithmic-bytes formula; frac_min prices the same time against
                  B_min (fast-decoder weights counted once per frame: they stay on chip for the 8 passes).
                  roofline.kernels itemises both launches of the timed region (HIP events around every launch of one more request
                  in FS_GEN_TIME_KERNELS mode): algorithmic bytes per launch, average duration, both fractions; dominant_kernel is
                  the longer of the two.  tools/check_roofline.py recomputes every number from this line + profiles/rNN_*.
  cpu_baseline -- the CPU restatement (oracle/, kind "port": the reference is Rust+candle and cannot be built here) timed
                  on this host on BASELINE.json configs[0] (rank 0, N=1 only), which also yields the greedy golden
                  token stream: the GPU f32 path must reproduce it bit-identically (reported under "parity").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))

import numpy as np

SEED = 0xF15E5EED
HBM_PEAK = 8.0e12  # MI355X HBM3E spec (guide: ~6.3 TB/s achievable by a copy kernel)
FRAME_RATE = 21.535  # generate/single_batch.rs:292-295


def default_voice_prompt(tok):
    """configs[1] prompt layout (prompt.rs:53-104): [sys 12][user 48][assistant 4][VQ span 274][im_end][user 24][assistant 4]."""
    rng = np.random.RandomState(2024)
    codes = np.load(os.path.join(ROOT, "tests", "golden", "default_voice_codes.npy")).astype(np.uint32)  # (8, 274)
    segs = []

    def text(n):
        p = np.zeros((9, n), np.uint32)
        p[0] = rng.randint(0, tok["im_end_id"], n)
        return p

    segs += [text(12), text(48), text(4)]
    vq = np.zeros((9, codes.shape[1]), np.uint32)
    vq[0] = tok["semantic_start_id"] + codes[0]
    vq[1:] = codes
    segs.append(vq)
    e = np.zeros((9, 1), np.uint32)
    e[0, 0] = tok["im_end_id"]
    segs += [e, text(24), text(4)]
    return np.ascontiguousarray(np.concatenate(segs, 1))


def frame_bytes(cfg, tok, T, wbytes=2):
    """algorithmic HBM bytes of one decode frame at KV length T (every dependent pass streams its
    weights once; audio-range head only)."""
    D, I = cfg["dim"], cfg["intermediate_size"]
    qkv = (cfg["n_head"] + 2 * cfg["n_local_heads"]) * cfg["head_dim"]
    block = qkv * D + D * D + 3 * I * D + 2 * D
    n_audio = cfg["vocab_size"] - tok["im_end_id"]
    slow = wbytes * (cfg["n_layer"] * block + D + n_audio * D)
    fast = cfg["num_codebooks"] * wbytes * (cfg["n_fast_layer"] * block + D + cfg["codebook_size"] * D)
    kv_tok = cfg["n_layer"] * 2 * cfg["n_local_heads"] * cfg["head_dim"] * wbytes
    kv_fast = cfg["num_codebooks"] * cfg["n_fast_layer"] * 2 * cfg["n_local_heads"] * cfg["head_dim"] * wbytes * 4
    return slow + fast + kv_tok * T + kv_fast


def frame_bytes_split(cfg, tok, T, wbytes=2):
    """frame_bytes() per launch of the persistent path: (slow step incl. the audio-range head and the slow KV read, the 8 fast passes
    as SURVEY.md §8(d) counts them -- weights streamed by every pass --, the fast decoder with its weights counted ONCE: what a launch
    that keeps them on chip has to read, B_min's fast term)."""
    D, I = cfg["dim"], cfg["intermediate_size"]
    qkv = (cfg["n_head"] + 2 * cfg["n_local_heads"]) * cfg["head_dim"]
    block = qkv * D + D * D + 3 * I * D + 2 * D
    n_audio = cfg["vocab_size"] - tok["im_end_id"]
    slow = wbytes * (cfg["n_layer"] * block + D + n_audio * D) + cfg["n_layer"] * 2 * cfg["n_local_heads"] * cfg["head_dim"] * wbytes * T
    fast_pass = wbytes * (cfg["n_fast_layer"] * block + D + cfg["codebook_size"] * D)
    kv_fast = cfg["num_codebooks"] * cfg["n_fast_layer"] * 2 * cfg["n_local_heads"] * cfg["head_dim"] * wbytes * 4
    return slow, cfg["num_codebooks"] * fast_pass + kv_fast, fast_pass + kv_fast


def config2_prompts(tok, n):
    """SURVEY.md §8d configs[2] / [3]: prompt lengths U{64..384}, seed 77; row 0 random text ids, codebook rows 0."""
    rng = np.random.RandomState(77)
    lens = rng.randint(64, 385, 256)
    prompts = []
    for L in lens:
        p = np.zeros((9, int(L)), np.uint32)
        p[0] = rng.randint(0, tok["im_end_id"], int(L))
        prompts.append(p)
    return prompts[:n]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start one process per GPU ourselves."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def dry_run(args, dist, rank, world):
    """No MI355X visible (CPU box): exercise the launch, request sharding, prompt broadcast and code fan-in with the gloo backend
    on placeholder code arrays -- nothing is computed or timed, `value` is null.  libfishrt has no CPU path to fall back to."""
    from fishrt import config as fcfg, fanout
    tok = fcfg.FISH_1_5_TOKENS
    n_req, B, frames = 256, 32, 4
    packed = lens = None
    if rank == 0:
        prompts = config2_prompts(tok, n_req)
        lens = np.array([p.shape[1] for p in prompts], np.int32)
        packed = np.zeros((n_req, 9, int(lens.max())), np.uint32)
        for i, p in enumerate(prompts):
            packed[i, :, : p.shape[1]] = p
    packed, lens = fanout.broadcast_prompts(dist, packed, lens)
    mine = fanout.shard_requests(n_req, rank, world)
    codes = np.stack([np.full((8, frames), int(packed[i, 0, lens[i] - 1]) % 1000, np.uint32) for i in mine])
    ca, fa, seen = fanout.all_gather_codes(dist, codes, np.full(len(mine), frames, np.int32))
    ok = seen == world and all((ca[r, k] == int(packed[i, 0, lens[i] - 1]) % 1000).all()
                               for r in range(world) for k, i in enumerate(fanout.shard_requests(n_req, r, world)))
    fanout.barrier(dist)
    if rank == 0:
        print(json.dumps({"metric": "codec tokens/sec (frames/s)", "value": None, "unit": "frames/s", "n_gpus": world, "dry_run": True,
                          "reason": "no HIP device visible: control path only (launch, shard, broadcast, all-gather); fishrt has no CPU path",
                          "requests": n_req, "requests_per_rank": [len(fanout.shard_requests(n_req, r, world)) for r in range(world)],
                          "collective_ranks": int(seen), "backend": fanout.backend_name(dist), "fan_in_ok": bool(ok),
                          "batch_per_rank": B, "config": args.config,
                          "static_batches_per_rank": [(len(fanout.shard_requests(n_req, r, world)) + B - 1) // B for r in range(world)]}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def run_config3(args, lm_factory, dist, rank, world, cfg, tok):
    """BASELINE.json configs[3]: 256 requests (configs[2] prompts), request i -> rank i mod N, static batches of 32 per rank
    (generate_static_batch, temp 0.7 / top-p 0.8 / top-k 256, on-device sampling), `frames` frames per request.  The prompt batch is
    broadcast from rank 0 and the codes are all-gathered at the end (RCCL); nothing on the per-token path crosses GPUs."""
    import torch
    from fishrt import fanout
    n_req, B, frames = 256, 32, args.frames
    packed = lens = None
    if rank == 0:
        prompts = config2_prompts(tok, n_req)
        lens = np.array([p.shape[1] for p in prompts], np.int32)
        packed = np.zeros((n_req, 9, int(lens.max())), np.uint32)
        for i, p in enumerate(prompts):
            packed[i, :, : p.shape[1]] = p
    packed, lens = fanout.broadcast_prompts(dist, packed, lens)
    mine = fanout.shard_requests(n_req, rank, world)
    lm = lm_factory(B)

    def one_job():
        codes = np.zeros((len(mine), 8, frames), np.uint32)
        nf = np.zeros(len(mine), np.int32)
        dec_s = pre_s = 0.0
        lmaxes = []
        for b0 in range(0, len(mine), B):
            idx = mine[b0:b0 + B]
            ps = [np.ascontiguousarray(packed[i, :, : lens[i]]) for i in idx]
            Lmax = max(p.shape[1] for p in ps)
            lmaxes.append(Lmax)
            outs = lm.generate_static_batch(ps, frames + Lmax - 2, temp=0.7, top_p=0.8, top_k=256, seed=42, ignore_eos=True)
            st = lm.last_stats()
            dec_s += st["decode_ms"] * 1e-3
            pre_s += st["prefill_ms"] * 1e-3
            for k, o in enumerate(outs):
                codes[b0 + k, :, : o.shape[1]] = o
                nf[b0 + k] = o.shape[1]
        return codes, nf, dec_s, pre_s, float(np.mean(lmaxes))

    def barrier():
        fanout.barrier(dist)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_job()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes, nf, dec_s, pre_s, lmax_avg = one_job()
        ca, fa, seen = fanout.all_gather_codes(dist, codes, nf)  # the job's fan-in is part of the timed region
    barrier()
    dt = fanout.max_over_ranks(dist, time.perf_counter() - t0)
    assert seen == world and fa.shape == (world, len(mine)) and int(fa.sum()) == n_req * frames, (seen, fa.shape, int(fa.sum()))
    frames_total = n_req * frames * args.steps
    n_batches = (len(mine) + B - 1) // B
    step_s = dec_s / (n_batches * (frames - 1))
    # a static batch is left-padded to its longest prompt (static_batch.rs:68-111): every row's KV length is Lmax + the frames so far
    bytes_step = frame_bytes(cfg, tok, 0) + B * 12288 * (lmax_avg + frames / 2)
    res = {
        "metric": "codec tokens/sec (frames/s; 1 frame = 1 slow + 8 codebook tokens = 2048 PCM samples), Fish-1.5 static batches of 32",
        "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (deterministic synthetic weights at Fish-1.5 shapes; configs[2] prompts U{64..384})",
        "config": {"workload": "BASELINE.json configs[3]: 256 requests sharded i mod N, static batches of 32 per GPU, "
                               f"temp 0.7 / top-p 0.8 / top-k 256, {frames} frames per request; one step = the whole 256-request job "
                               "(prefill + decode + code fan-in in the timed region)",
                   "requests": n_req, "batch_per_gpu": B, "frames_per_request": frames,
                   "parallelism": f"request shards x{world}; prompt broadcast + code all-gather over RCCL, no per-token collective"},
        "rtf": round((frames_total / FRAME_RATE) / dt, 2),
        "rccl_ranks": int(seen), "fanout_backend": fanout.backend_name(dist), "frames_per_rank": [int(v) for v in fa.sum(axis=1)],
        "decode_step_us_rank0": round(step_s * 1e6, 1), "prefill_s_per_job_rank0": round(pre_s, 3),
        "roofline": {"bound": "hbm", "kernel": "static-batch decode step (one graph replay, B = 32 rows on the MFMA row path)",
                     "achieved": round(bytes_step / step_s / 1e9, 2),
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(bytes_step / step_s / HBM_PEAK, 4), "algorithmic_bytes_per_step": int(bytes_step),
                     **offline_batch_traffic("static_batch32")},
    }
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--config", type=int, default=1, choices=(1, 3),
                    help="1: BASELINE.json configs[1] (batch-1 request per GPU, the headline metric); 3: configs[3] (256 requests sharded 32/GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed configs[2] / vocoder side measurements")
    ap.add_argument("--no-persistent", action="store_true", help="FS_GEN_NO_PERSIST: fast decoder as 144 graph nodes per frame (A/B)")
    args = ap.parse_args()
    self_launch(args)

    import torch  # before libfishrt: both bring a HIP runtime, and the one loaded first serves the process
    import fishrt
    from fishrt import config as fcfg, fanout
    rank, local_rank, world = fanout.env_rank()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; run `python bench.py --gpus N` (it starts the "
              f"ranks itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`", file=sys.stderr)
        sys.exit(2)
    have_gpu = fishrt.lib().fs_device_count() > 0
    # test hooks for a ONE-GPU box: FISHRT_BENCH_DEVICE pins every rank to one device, FISHRT_BENCH_BACKEND=gloo replaces RCCL (which
    # refuses two ranks on one GPU) -- together they run the whole N > 1 path, weight broadcast included, on device memory
    if "FISHRT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["FISHRT_BENCH_DEVICE"])
    if have_gpu:
        torch.cuda.set_device(local_rank)  # (the barriers' torch.cuda.synchronize() must mean THIS rank's GPU, not device 0 on every rank)
    backend = os.environ.get("FISHRT_BENCH_BACKEND", "rccl") if have_gpu else "gloo"
    # rccl: the fs_comm_* C entry points of libfishrt.so on librccl directly (request fan-out + the job's clock; no torch process group)
    dist = fanout.init(backend if world > 1 else None, device=local_rank)
    if world == 1 and have_gpu and os.environ.get("FISHRT_BENCH_COMM1"):  # test hook: the whole fan-out through a ONE-rank RCCL communicator
        from fishrt.comm import RcclComm
        dist = RcclComm.from_env(device=local_rank)
    if not have_gpu:
        dry_run(args, dist, rank, world)

    cfg, tok = fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS
    if args.config == 3:
        return run_config3(args, lambda B: fishrt.DualARTransformer(cfg, tok, local_rank, "bf16", max_batch=B).load_synthetic(SEED),
                           dist, rank, world, cfg, tok)
    # replica start-up: rank 0 materialises the weights, the other ranks receive the arena over RCCL / xGMI
    lm = fishrt.DualARTransformer(cfg, tok, local_rank, "bf16")
    wbcast = None
    if dist is not None:
        if rank == 0:
            lm.load_synthetic(SEED)
        fanout.barrier(dist)
        t0 = time.perf_counter()
        err = None
        try:
            nbytes = fanout.broadcast_weights(dist, lm, src=0)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"[:300]
        # every rank learns whether EVERY rank succeeded before anyone picks the fallback (a rank that failed alone must not skip a
        # collective the others wait in, and the ranks must not end up with different weight provenance)
        ok_all = fanout.min_over_ranks(dist, 0 if err else 1) == 1
        fanout.barrier(dist)
        dtb = time.perf_counter() - t0
        if ok_all:
            wbcast = {"bytes": int(nbytes), "ms": round(dtb * 1e3, 1), "GBps_per_receiver": round(nbytes / dtb / 1e9, 1),
                      "how": "fs_comm_broadcast_weights: arena -> ncclBroadcast in 256 MB pieces -> fs_lm_weights_adopt" if fanout.backend_name(dist).startswith("rccl")
                             else "fs_lm_weights_arena -> torch.distributed.broadcast (gloo rehearsal) -> fs_lm_weights_adopt"}
        else:  # the replicas do not depend on it: every rank materialises the (deterministic) weights itself -- ALL of them
            wbcast = {"error": err or "another rank failed", "how": "fallback: every rank ran fs_lm_load_synthetic"}
            if rank != 0:
                lm.load_synthetic(SEED)
    else:
        lm.load_synthetic(SEED)
    prompt = default_voice_prompt(tok)
    L = prompt.shape[1]
    M = args.frames + L - 2  # budget counts prompt tokens (single_batch.rs:61,77): frames = M - L + 2
    samp = dict(temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True, persistent=not args.no_persistent)

    def one_request():
        lm.clear_slow_layer_caches()
        out = lm.generate_blocking(prompt, M, **samp)
        assert out.shape == (8, args.frames), out.shape
        return out, lm.last_stats()

    def barrier():
        fanout.barrier(dist)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ref_out, _ = one_request()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        out, st = one_request()
        stats.append(st)
    barrier()
    dt = time.perf_counter() - t0
    if args.warmup:
        assert np.array_equal(out, ref_out), "non-deterministic greedy tokens across requests"
    dt = fanout.max_over_ranks(dist, dt)
    # fan-in of every rank's codes: one all-gather over RCCL after the timed region
    ca, fa, seen = fanout.all_gather_codes(dist, out[None], np.array([out.shape[1]], np.int32))
    assert seen == world and ca.shape == (world, 1, 8, args.frames)
    assert (ca == ca[0]).all(), "replicas disagree on the greedy tokens of the same request (weight broadcast / load mismatch)"

    frames_total = args.frames * args.steps * world
    value = frames_total / dt
    dec_ms = float(np.mean([s["decode_ms"] for s in stats]))
    pre_ms = float(np.mean([s["prefill_ms"] for s in stats]))
    kpf = int(stats[-1]["kernels_per_frame"])
    t_frame = dec_ms * 1e-3 / (args.frames - 1)  # events bracket frames 1..n-1 exactly like `start_decode` (:261)
    T_avg = L + args.frames / 2.0
    bf = frame_bytes(cfg, tok, T_avg)
    achieved = bf / t_frame
    # the kernels of the timed region, one by one: one more request in measurement mode (FS_GEN_TIME_KERNELS: the same two persistent
    # launches per frame, a HIP event in front of / between / behind them on the engine stream); rocprofv3's averages for the same
    # kernels are committed under profiles/ and checked against these by tools/check_roofline.py
    kernels, dominant = {}, None
    b_slow, b_fast, b_fast_min = frame_bytes_split(cfg, tok, T_avg)
    if kpf == 2:
        lm.clear_slow_layer_caches()
        outk = lm.generate_blocking(prompt, M, time_kernels=True, **samp)
        assert np.array_equal(outk, out), "measurement mode changed the tokens"
        stk = lm.last_stats()
        for name, us, nb, nb_min, what in (
                ("k_slow_persist", stk["slow_kernel_us"], b_slow, b_slow, "one launch = forward_generate of one token: 24 blocks over the paged KV cache + norm + audio-range head"),
                ("k_fast_persist", stk["fast_kernel_us"], b_fast, b_fast_min, "one launch = the slow-token decision + 8 x forward_generate_fast + 8 codebook decisions + next-input embedding; "
                                                                            "algorithmic bytes count the weights once per pass (SURVEY.md 8d), bytes_min once per launch (they stay on chip)")):
            kernels[name] = {"what": what, "avg_us": round(us, 2), "launches_per_frame": 1, "algorithmic_bytes_per_launch": int(nb),
                             "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(nb / (us * 1e-6) / HBM_PEAK, 4),
                             "bytes_min_per_launch": int(nb_min), "frac_min": round(nb_min / (us * 1e-6) / HBM_PEAK, 4)}
        dn = max(kernels, key=lambda k: kernels[k]["avg_us"])
        dominant = {"name": dn, **kernels[dn], "share_of_frame_time": round(kernels[dn]["avg_us"] / (t_frame * 1e6), 3),
                    "note": "HIP events on the engine stream around every launch of a 256-frame request (fs_gen_stats.slow_kernel_us / fast_kernel_us)"}
    else:  # per-node path (--no-persistent): the frame's biggest node by bytes, measured as a graph node over distinct layer weights
        up_us = lm.bench_kernel(3, int(T_avg), 50)
        up_bytes = 2 * (2 * cfg["intermediate_size"] * cfg["dim"]) + 8 * cfg["dim"] + 4 * cfg["intermediate_size"]
        dominant = {"name": "k_ffn_up<bf16, 1024> (RMSNorm + W1||W3 GEMV + SwiGLU; 56 launches per frame)", "algorithmic_bytes_per_launch": up_bytes,
                    "avg_us": round(up_us, 2), "achieved": round(up_bytes / up_us / 1e3, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": round(up_bytes / (up_us * 1e-6) / HBM_PEAK, 4), "note": "graph node incl. the launch floor; fs_lm_bench_kernel"}
    b_min = b_slow + b_fast_min
    # the reference's DEFAULT request is sampled (temp 0.7 / top-k 256: llama_generate.rs:114-124, server/lib/utils/load.rs:116-125; top-p 0.8 as
    # BASELINE.json configs[2]); `value` above is the greedy-parity configuration BASELINE.json names.  Same prompt, same 256 frames, same two
    # launches per frame with the block-parallel sampler inside k_fast_persist<true>; outside the timed region.  "flat" = the synthetic weights'
    # near-uniform rows (every draw runs the full top-k / top-p chain: the worst case); "peaked" = temp 0.02 on the same weights (the largest
    # probability alone exceeds top_p on most rows, as on a trained head: the sampler's one-weight shortcut).
    sampled = {}
    for tag, skw in (("flat_rows", dict(temp=0.7, top_p=0.8, top_k=256)), ("peaked_rows", dict(temp=0.02, top_p=0.8, top_k=256))):
        for _ in range(2):
            lm.clear_slow_layer_caches()
            lm.generate_blocking(prompt, M, repetition_penalty=1.2, seed=1, ignore_eos=True, persistent=not args.no_persistent, **skw)
        sst = lm.last_stats()
        s_frame = sst["decode_ms"] * 1e-3 / (args.frames - 1)
        sampled[tag] = {"sampling": skw, "frame_us": round(s_frame * 1e6, 2), "decode_frames_per_s": round(1.0 / s_frame, 1),
                        "achieved": round(bf / s_frame / 1e9, 2), "frac": round(bf / s_frame / HBM_PEAK, 4), "kernels_per_frame": int(sst["kernels_per_frame"])}
    traffic = offline_traffic(kpf)
    res = {
        "metric": "codec tokens/sec (frames/s; 1 frame = 1 slow + 8 codebook tokens = 2048 PCM samples), Fish-1.5 batch=1",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (deterministic synthetic weights at Fish-1.5 shapes; default-voice-shaped prompt)",
        "config": {"workload": "BASELINE.json configs[1]: Fish-1.5 bf16 batch=1, default-voice prompt, 256-frame generation, "
                               "one request per step per GPU (prefill included in the timed region)",
                   "prompt_positions": L, "frames_per_request": args.frames, "requests_per_step": world,
                   "parallelism": f"replicas x{world} (no data-path collective; codes all-gathered over RCCL after the run)"},
        "rtf": round((frames_total / FRAME_RATE) / dt, 2), "rccl_ranks": int(seen), "fanout_backend": fanout.backend_name(dist), "weight_broadcast": wbcast, "frames_per_rank": [int(v) for v in fa.sum(axis=1) * args.steps],
        "decode_frames_per_s_per_gpu": round(1.0 / t_frame, 2), "prefill_ms": round(pre_ms, 3),
        "roofline": {"bound": "hbm", "kernel": (f"decode frame = {kpf} kernels of a hipGraph replay (8 frames per graph launch on the persistent path): " +
                                               ("k_slow_persist (24 slow blocks + head) then k_fast_persist (slow-token decision, 8 codebook passes, 8 decisions)" if kpf == 2 else
                                                "24 slow blocks x 5 + head + sample, then 8 x (4 fast blocks x 4 + head + sample)")
                                               + "; HIP-event timed on the engine stream"),
                     "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK, 4),
                     **traffic,
                     "algorithmic_bytes_per_frame": int(bf), "frame_us": round(t_frame * 1e6, 2), "kv_len_avg": T_avg, "kernels_per_frame": kpf,
                     "bytes_min_per_frame": int(b_min), "frac_min": round(b_min / t_frame / HBM_PEAK, 4),
                     "kernels": kernels, "dominant_kernel": dominant,
                     "sampled": {"what": "the same request under the reference's default sampling (on-device top-k / top-p / WeightedIndex, token-exact StdRng stream) "
                                         "instead of greedy: decode frame time and fraction of the same algorithmic bytes", **sampled}},
    }
    if rank == 0 and world == 1 and not args.no_extras:
        res["extras"] = extras(cfg, tok)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res.update(cpu_baseline_and_parity(cfg, tok))
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def offline_traffic(kernels_per_frame):
    """roofline.traffic: HBM bytes per decode frame from the PMC passes summarised in profiles/rNN_pmc_hbm_traffic.json (written by
    tools/pmc_traffic.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs; counters cannot be read inside this process).
    The summary names the command, the counter corrections and the commit it was taken at; null when no summary matches the path."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):  # the newest round's summary
        with open(path) as f:
            t = json.load(f)
        key = "persistent" if kernels_per_frame < 200 else "per_node"
        if key in t:
            return {"traffic": t[key]["hbm_bytes_per_frame"],
                    "traffic_source": f"offline: profiles/{os.path.basename(path)}[{key}] ({t[key].get('command', '')}; commit {t.get('commit', '?')})"}
    return {"traffic": None, "traffic_source": "no PMC summary for this path under profiles/ (tools/pmc_traffic.sh)"}


def offline_batch_traffic(key):
    """HBM bytes per decode step of the static batch of 32 (key "static_batch32") or per R-row persistent frame ("rows_R4", "rows_R8") from
    the differenced PMC passes summarised in profiles/rNN_pmc_batch_traffic.json (tools/pmc_batch.sh); None when no tracked summary has it."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_batch_traffic.json")), reverse=True):
        with open(path) as f:
            t = json.load(f)
        if key in t:
            v = t[key]
            return {"traffic": v.get("hbm_bytes_per_step", v.get("hbm_bytes_per_frame")),
                    "traffic_algorithmic_bytes_same_window": v.get("algorithmic_bytes_per_step", v.get("algorithmic_bytes_per_frame")),
                    "traffic_source": f"offline: profiles/{os.path.basename(path)}[{key}] ({v.get('command', '')}; commit {t.get('commit', '?')})"}
    return {"traffic": None, "traffic_source": "no PMC summary for this path under profiles/ (tools/pmc_batch.sh)"}


def continuous_vs_lockstep(lmb, prompts):
    """64 requests with ragged lengths (prompts U{64..384}, 64..256 frames each, fixed by ignore_eos) on a 32-row handle: lock-step static
    batches (two batches of 32, each as long as its longest row, generate/static_batch.rs:282-390) vs continuous batching (fs_lm_session_*:
    a request joins when a slot is free and leaves when done).  Whole-job wall time incl. prefill; frames = what the requests asked for."""
    rng = np.random.RandomState(5)
    want = [int(rng.randint(64, 257)) for _ in prompts]
    kw = dict(temp=0.7, top_p=0.8, top_k=256, seed=42, ignore_eos=True)
    t0 = time.perf_counter()
    got_lock, done_lock = 0, []
    for i in range(0, len(prompts), 32):
        ps, w = prompts[i:i + 32], want[i:i + 32]
        Lmax = max(p.shape[1] for p in ps)
        outs = lmb.generate_static_batch(ps, max(w) + Lmax - 2, **kw)  # every row runs to the longest request of its batch
        got_lock += sum(min(o.shape[1], n) for o, n in zip(outs, w))
        done_lock += [time.perf_counter() - t0] * len(ps)           # a lock-step batch hands all its rows back together
    t_lock = time.perf_counter() - t0
    t0 = time.perf_counter()
    got_cont, peak, done_cont = 0, 0, []
    with lmb.session(**kw) as s:
        pending, live = list(range(len(prompts))), {}
        while pending or live:
            while pending:
                i = pending[0]
                slot = s.add(prompts[i], want[i] + prompts[i].shape[1] - 2)
                if slot is None:
                    break
                live[slot] = pending.pop(0)
            peak = max(peak, len(live))
            s.step(8)
            for slot in list(live):
                n, done = s.poll(slot, codes=False)
                if done:
                    assert n == want[live.pop(slot)]
                    got_cont += n
                    done_cont.append(time.perf_counter() - t0)
                    s.release(slot)
    t_cont = time.perf_counter() - t0
    assert got_cont == got_lock == sum(want)
    # what slot refill can buy on THIS workload: lock-step = sum of the batches' longest rows, continuous = FIFO list scheduling of the same
    # requests on 32 slots with free admission (both in decode steps; a step costs the same at any occupancy)
    import heapq
    lock_steps = sum(max(want[i:i + 32]) for i in range(0, len(want), 32))
    slots = [0] * 32
    for w in want:
        heapq.heappush(slots, heapq.heappop(slots) + w)
    bound = lock_steps / max(slots)
    return {"workload": "64 requests, prompts U{64..384}, 64..256 frames each (ignore_eos), 32 slots / rows, bf16, top-k 256 / top-p 0.8 sampling; "
                        "wall time incl. prefill",
            "frames": int(sum(want)), "lockstep_s": round(t_lock, 3), "continuous_s": round(t_cont, 3),
            "lockstep_frames_per_s": round(sum(want) / t_lock, 1), "continuous_frames_per_s": round(sum(want) / t_cont, 1),
            "speedup": round(t_lock / t_cont, 3), "speedup_bound_fifo_steps": round(bound, 3), "peak_live_slots": peak,
            "mean_completion_s": {"lockstep": round(float(np.mean(done_lock)), 3), "continuous": round(float(np.mean(done_cont)), 3)},
            "p50_completion_s": {"lockstep": round(float(np.median(done_lock)), 3), "continuous": round(float(np.median(done_cont)), 3)}}


def extras(cfg, tok):
    """Side measurements outside the timed region (not part of `value`): BASELINE.json configs[2] (static batch of 32 on the
    MFMA row path), the Firefly vocoder on the 256 frames of one request, the encoder on a 10 s clip, and batch-1 decode with the
    on-device top-k/top-p sampler and with fp8 weights."""
    import fishrt
    out = {}
    prompts_all = config2_prompts(tok, 256)  # SURVEY.md §8d configs[2]: prompt lengths U{64..384}, seed 77 (first 32 = the B=32 batch)
    for name, dtype, wb, B, frames in (("static_batch32", "bf16", 2, 32, 256), ("static_batch32_fp8", "fp8", 1, 32, 256),
                                       ("static_batch256", "bf16", 2, 256, 256)):
        prompts = prompts_all[:B]
        Lmax = max(p.shape[1] for p in prompts)
        lmb = fishrt.DualARTransformer(cfg, tok, 0, dtype, max_batch=B).load_synthetic(SEED)
        outs = lmb.generate_static_batch(prompts, frames + Lmax - 2, temp=0.7, top_p=0.8, top_k=256, seed=42, ignore_eos=True)
        st = lmb.last_stats()
        step_s = st["decode_ms"] * 1e-3 / (frames - 1)
        bytes_step = frame_bytes(cfg, tok, 0, wb) + B * 12288 * (Lmax + frames / 2)
        out[name] = {"workload": f"BASELINE.json configs[2] shape: B={B}, {dtype} weights, temp 0.7 / top-p 0.8 / top-k 256, prompts U{{64..384}}, "
                                 f"{frames} frames (decode steps HIP-event timed; prefill = one group pass per <= 2048 prompt rows, timed separately)",
                     "decode_frames_per_s": round(B / step_s, 1), "step_us": round(step_s * 1e6, 1),
                     "roofline_frac": round(bytes_step / step_s / HBM_PEAK, 4),
                     **({"note": "fp8 is a FOOTPRINT format here (0.64 GB instead of 1.28 GB per replica): the step is latency-bound (300 dependent graph nodes), so "
                                 "halving the streamed bytes does not shorten it and roofline_frac -- half the algorithmic bytes over the same time -- reads lower, not slower"}
                        if dtype == "fp8" else {}),
                     "frames_out": int(sum(o.shape[1] for o in outs)), "prefill_ms_all_rows": round(st["prefill_ms"], 1),
                     "prefill_tokens_per_s": round(B * (Lmax - 1) / (st["prefill_ms"] * 1e-3), 0),
                     "algorithmic_bytes_per_step": int(bytes_step)}
        if name == "static_batch32":
            out[name].update(offline_batch_traffic("static_batch32"))
            out["continuous_batching32"] = continuous_vs_lockstep(lmb, prompts_all[:64])
        lmb.close()
    codes = np.random.RandomState(1).randint(0, 1000, (1, 8, 256)).astype(np.uint32)
    voc_ms, voc_pcm, voc_med = {}, {}, {}
    for mode in ("f16", "bf16x3", "f32"):
        codec = fishrt.FireflyCodec(0, precision=mode).load_synthetic(0xC0DEC)
        codec.decode(codes)
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            voc_pcm[mode] = codec.decode(codes)
            runs.append(time.perf_counter() - t0)
        voc_ms[mode] = min(runs) * 1e3
        voc_med[mode] = float(np.median(runs)) * 1e3
        if mode != "f32":
            codec.close()
    ref64 = voc_pcm["f32"].astype(np.float64)
    dt = voc_ms["f16"] / 1e3
    out["vocoder"] = {"workload": "FireflyCodec.decode of 256 frames (11.9 s of 44.1 kHz audio), default f16 precision mode (single f16 matrix "
                                  "operands, f32 accumulation and residual stream; bound: PCM within 1e-4 RMS of the f32 oracle), host buffers "
                                  "in/out, best of 3",
                      "ms": round(voc_ms["f16"], 2), "ms_median_of_3": round(voc_med["f16"], 2), "rtf": round((256 / FRAME_RATE) / dt, 1), "tflops_equiv": round(2.65e9 * 256 / dt / 1e12, 2),
                      "pcm_finite": bool(np.isfinite(voc_pcm["f16"]).all()),
                      "pcm_rms_vs_f32_mode": float(np.sqrt(np.mean((voc_pcm["f16"] - ref64) ** 2))),
                      "ms_bf16x3_mode": round(voc_ms["bf16x3"], 2),
                      "pcm_rms_bf16x3_vs_f32_mode": float(np.sqrt(np.mean((voc_pcm["bf16x3"] - ref64) ** 2))),
                      "ms_f32_mode": round(voc_ms["f32"], 2), "signal_rms": float(np.sqrt(np.mean(ref64 ** 2)))}
    # FireflyCodec.encode of a 10 s clip (mel front-end + ConvNeXt encoder + FSQ), host buffers in/out
    t = np.arange(441000) / 44100.0
    clip = (0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.05 * np.random.RandomState(2).randn(t.size)).astype(np.float32)[None, None]
    codec.encode(clip)
    t0 = time.perf_counter()
    enc = codec.encode(clip)
    dt = time.perf_counter() - t0
    out["encoder"] = {"workload": "FireflyCodec.encode of 10 s of 44.1 kHz audio (log-mel + ConvNeXt encoder + grouped FSQ), f32",
                      "ms": round(dt * 1e3, 2), "rtf": round(10.0 / dt, 1), "codes_shape": list(enc.shape)}
    codec.close()
    # batch-1 decode under the server's default sampling (temp 0.7 / top-p 0.8 / top-k 256, on-device sampler) and with fp8 weights
    tokp = default_voice_prompt(tok)
    # sampled_b1_peaked: the same sampler on PEAKED rows (temp 0.02 sharpens the flat logits of the synthetic weights until, on most rows,
    # the largest probability alone exceeds top_p -- what a trained model's rows mostly look like): the sampler's one-weight shortcut
    # (1.2-1.6 us) instead of its full top-k / top-p chain (5.7 us on flat rows; tools/ubench_bsample.hip)
    for name, dtype, kw in (("sampled_b1", "bf16", dict(temp=0.7, top_p=0.8, top_k=256)), ("sampled_b1_peaked", "bf16", dict(temp=0.02, top_p=0.8, top_k=256)),
                            ("fp8_b1_greedy", "fp8", dict(temp=0.0, top_p=1.0, top_k=0))):
        lm1 = fishrt.DualARTransformer(cfg, tok, 0, dtype).load_synthetic(SEED)
        for _ in range(2):
            lm1.clear_slow_layer_caches()
            lm1.generate_blocking(tokp, 128 + tokp.shape[1] - 2, repetition_penalty=1.2, seed=1, ignore_eos=True, **kw)
        st1 = lm1.last_stats()
        out[name] = {"workload": f"configs[1] prompt, 128 frames, {dtype} weights, {kw}", "frame_us": round(st1["decode_ms"] * 1e3 / 127, 1),
                     "decode_frames_per_s": round(127 / (st1["decode_ms"] * 1e-3), 1), "prefill_ms": round(st1["prefill_ms"], 2)}
        lm1.close()
    # R concurrent configs[1] requests through the request-row persistent kernels (fs_lm_generate_multi: each request IS its own
    # generate_blocking call -- own KV, repetition-penalty window, sampler stream; every decode frame = one slow launch for all rows + one
    # fast launch per <= 4 rows)
    lm8 = fishrt.DualARTransformer(cfg, tok, 0, "bf16", max_batch=8).load_synthetic(SEED)
    Lp, Fr = tokp.shape[1], 256
    one_us = 1e9
    for _ in range(2):  # (the first call captures the frame graphs inside its decode region)
        lm8.clear_slow_layer_caches()
        ref1 = lm8.generate_blocking(tokp, Fr + Lp - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
        one_us = min(one_us, lm8.last_stats()["decode_ms"] * 1e3 / (Fr - 1))
    rows = {"workload": "R concurrent BASELINE.json configs[1] requests (default-voice prompt, 256 frames each) on ONE GPU through "
                        "fs_lm_generate_multi; frame_us = one decode frame of all R requests (HIP events), decode_frames_per_s = R / frame; "
                        "whole_job_frames_per_s includes the R prefills (host wall clock)",
            "one_request_frame_us": round(one_us, 1)}
    for R, kw in ((2, dict(temp=0.0, top_p=1.0, top_k=0)), (4, dict(temp=0.0, top_p=1.0, top_k=0)), (8, dict(temp=0.0, top_p=1.0, top_k=0)),
                  (4, dict(temp=0.7, top_p=0.8, top_k=256))):
        best, wall, outs = 1e9, 1e9, None
        for _ in range(2):
            t0 = time.perf_counter()
            outs = lm8.generate_multi([tokp] * R, Fr + Lp - 2, repetition_penalty=1.2, seeds=list(range(1, R + 1)), ignore_eos=True, **kw)
            wall = min(wall, time.perf_counter() - t0)
            stR = lm8.last_stats()
            best = min(best, stR["decode_ms"] * 1e3 / (Fr - 1))
        assert all(o.shape == (8, Fr) for o in outs) and stR["kernels_per_frame"] == 1 + (R + 3) // 4
        T_avg = Lp + Fr / 2.0
        bytes_frame = frame_bytes(cfg, tok, 0) + R * 12288 * T_avg  # SURVEY.md 8d "batch B": weights once per step, KV term x B
        key = f"R{R}" + ("" if kw["temp"] == 0.0 else "_sampled")
        rows[key] = {"frame_us": round(best, 1), "decode_frames_per_s": round(R * 1e6 / best, 1), "whole_job_frames_per_s": round(R * Fr / wall, 1),
                     "speedup_vs_one_request_at_a_time": round(R * one_us / best, 2), "launches_per_frame": int(stR["kernels_per_frame"]),
                     "roofline_frac": round(bytes_frame / (best * 1e-6) / HBM_PEAK, 4), "algorithmic_bytes_per_frame": int(bytes_frame),
                     **(offline_batch_traffic(f"rows_R{R}") if kw["temp"] == 0.0 and R in (4, 8) else {}),
                     # (the R requests are the same request: identical rows; they part from the batch-1 kernels' tokens at the first bf16
                     # near-tie -- other summation order -- like the batch-1 persistent path parts from the per-node path)
                     "rows_identical_to_each_other": (bool(all(np.array_equal(o, outs[0]) for o in outs)) if kw["temp"] == 0.0 else None),
                     "frames_identical_to_the_batch1_call": (int(np.argmax((outs[0] != ref1).any(0))) if kw["temp"] == 0.0 and (outs[0] != ref1).any() else
                                                             (Fr if kw["temp"] == 0.0 else None))}
    # round 5: the same on an FS_FP8 handle (e4m3 weights widened into the row kernels' MFMA images, quantiser row scales in the publishing lanes)
    lm4f = fishrt.DualARTransformer(cfg, tok, 0, "fp8", max_batch=4).load_synthetic(SEED)
    bestf, outsf = 1e9, None
    for _ in range(2):
        outsf = lm4f.generate_multi([tokp] * 4, Fr + Lp - 2, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, seeds=[1, 2, 3, 4], ignore_eos=True)
        stf = lm4f.last_stats()
        bestf = min(bestf, stf["decode_ms"] * 1e3 / (Fr - 1))
    assert all(o.shape == (8, Fr) for o in outsf) and stf["kernels_per_frame"] == 2
    lm4f.close()
    bytes_f = frame_bytes(cfg, tok, 0, 1) + 4 * 12288 * (Lp + Fr / 2.0)
    rows["R4_fp8"] = {"frame_us": round(bestf, 1), "decode_frames_per_s": round(4 * 1e6 / bestf, 1), "launches_per_frame": 2,
                      "roofline_frac": round(bytes_f / (bestf * 1e-6) / HBM_PEAK, 4), "algorithmic_bytes_per_frame": int(bytes_f),
                      "note": "fp8 weight STORAGE (0.64 GB per replica); the row images stream the widened bf16 values, so the frame costs what the bf16 rows cost"}
    # continuous batching on the row kernels (FS_SESSION_ROWS): 24 ragged sampled requests through 8 slots vs one request at a time on the
    # batch-1 persistent kernels (what the reference's mutex-serialised server does, server/lib/state.rs:12-29) -- same handle, same requests
    rngc = np.random.RandomState(5)
    reqs = prompts_all[:24]
    wantf = [int(rngc.randint(64, 257)) for _ in reqs]
    kwc = dict(temp=0.7, top_p=0.8, top_k=256)
    t0 = time.perf_counter()
    done_seq = []
    for q, w in zip(reqs, wantf):
        lm8.clear_slow_layer_caches()
        o = lm8.generate_blocking(q, w + q.shape[1] - 2, repetition_penalty=1.2, seed=3, ignore_eos=True, **kwc)
        assert o.shape[1] == w
        done_seq.append(time.perf_counter() - t0)
    t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    done_rows = []
    with lm8.session(seed=3, ignore_eos=True, rows=True, repetition_penalty=1.2, **kwc) as sess:
        pending, live = list(range(len(reqs))), {}
        while pending or live:
            while pending:
                i = pending[0]
                slot = sess.add(reqs[i], wantf[i] + reqs[i].shape[1] - 2)
                if slot is None:
                    break
                live[slot] = pending.pop(0)
            sess.step(8)
            for slot in list(live):
                nfr, dn = sess.poll(slot, codes=False)
                if dn:
                    assert nfr == wantf[live.pop(slot)]
                    done_rows.append(time.perf_counter() - t0)
                    sess.release(slot)
    t_rows = time.perf_counter() - t0
    rows["session8_sampled"] = {"workload": "24 requests, prompts U{64..384}, 64..256 frames each (ignore_eos), temp 0.7 / top-p 0.8 / top-k 256, repetition "
                                            "penalty 1.2: 8 request-row slots (FS_SESSION_ROWS, batch-1 semantics per slot) vs one request at a time on the "
                                            "batch-1 persistent kernels; wall time incl. prefill",
                                "frames": int(sum(wantf)), "one_at_a_time_s": round(t_seq, 3), "row_session_s": round(t_rows, 3),
                                "speedup": round(t_seq / t_rows, 2), "frames_per_s": round(sum(wantf) / t_rows, 1),
                                "mean_completion_s": {"one_at_a_time": round(float(np.mean(done_seq)), 3), "row_session": round(float(np.mean(done_rows)), 3)}}
    lm8.close()
    if "static_batch32" in out and "R8" in rows:
        rows["vs_static_batch32"] = {"thirty_two_requests_as_4_launch_groups_of_8_us": round(4 * rows["R8"]["frame_us"], 1),
                                     "static_batch32_step_us": out["static_batch32"]["step_us"],
                                     "note": "the stage chain of the row kernels is latency-bound per launch group, so 32 rows as 4 x R = 8 cost 4 chains: the "
                                             "MFMA row path (one chain of ~300 nodes for all 32 rows) stays the B = 32 path; the row kernels are the 2..8-request path"}
    out["persistent_rows"] = rows
    # N independent batch-1 request streams on this ONE GPU (own handle, HIP stream and host thread each; no lock-step batching):
    # the frame is a chain of dependent graph nodes whose launch gaps leave the chip idle, so independent chains interleave
    import threading
    nmax = 8
    lms = [fishrt.DualARTransformer(cfg, tok, 0, "bf16").load_synthetic(SEED) for _ in range(nmax)]
    Mreq = 256 + tokp.shape[1] - 2
    kw = dict(temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    # one generate call per device holds the persistent decode kernels; concurrent calls take the per-node graph (same arithmetic up to bf16
    # near-ties: tests/test_persist_gpu.py), so a stream's tokens must equal one of the two single-stream references
    ref = None
    for lmx in lms:
        lmx.clear_slow_layer_caches()
        o = lmx.generate_blocking(tokp, Mreq, **kw)
        ref = o if ref is None else ref
        assert np.array_equal(o, ref)
    lms[0].clear_slow_layer_caches()
    ref_nodes = lms[0].generate_blocking(tokp, Mreq, persistent=False, **kw)
    conc, bad = {}, []
    for n in (1, 2, 8):
        def work(lmx):
            for _ in range(2):
                lmx.clear_slow_layer_caches()
                # the persistent kernels spin on every CU: right for a lone stream, wrong next to other streams (measured: 2 streams 0.95 x,
                # 8 streams 1.3 x of one stream with them vs 1.6 x / 2.2 x without), so concurrent streams take the per-node graphs
                o = lmx.generate_blocking(tokp, Mreq, persistent=(n == 1), **kw)
                if not np.array_equal(o, ref if n == 1 else ref_nodes):
                    bad.append(n)
        ths = [threading.Thread(target=work, args=(lms[i],)) for i in range(n)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        conc[str(n)] = {"frames_per_s": round(n * 2 * 256 / dt, 1), "ms_per_request": round(dt / 2 * 1e3, 1)}
    for lmx in lms:
        lmx.close()
    assert not bad, f"a concurrent stream's tokens match neither single-stream reference (N = {bad})"
    out["concurrent_b1_streams_one_gpu"] = {"workload": "configs[1] requests (prefill + 256 frames, greedy), N independent batch-1 streams in flight "
                                                        "on one GPU (N = 1: persistent decode kernels; N > 1: per-node graphs on every stream -- the persistent kernels "
                                                        "occupy all CUs); every stream's tokens equal the single-stream run of its path", **conc}
    return out


def cpu_baseline_and_parity(cfg, tok):
    """the Candle-CPU stand-in on configs[0] (f32, prompt (9,16), greedy, rep-pen 1.2, 242 frames) --
    timed on this host, and its token stream checked bit-for-bit against the GPU f32 path."""
    from oracle import oracle as orc
    import fishrt
    rng = np.random.RandomState(1234)
    p = np.zeros((9, 16), np.uint32)
    p[0] = rng.randint(0, tok["im_end_id"], 16)
    o = orc.OracleLM(orc.FISH15).load_synthetic(SEED, bf16=False)
    t0 = time.perf_counter()
    exp = o.generate(p, 256, temp=0.0, repetition_penalty=1.2, ignore_eos=True)
    wall = time.perf_counter() - t0
    n = exp.shape[1]
    cpu_fps = (n - 1) / o.last_decode_s
    min_margin = float(o.last_margins.min())
    del o
    lm32 = fishrt.DualARTransformer(cfg, tok, 0, "f32").load_synthetic(SEED)
    got = lm32.generate_blocking(p, 256, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    st = lm32.last_stats()
    lm32.close()
    return {
        "cpu_baseline": {"value": round(cpu_fps, 3), "unit": "frames/s", "cores": orc.usable_cpus(), "kind": "port",
                         "sample": f"BASELINE.json configs[0] in full: f32, prompt (9,16), greedy, rep-pen 1.2, {n} frames "
                                   f"({wall:.1f} s wall incl. prefill; OpenMP over {orc.usable_cpus()} usable CPUs)",
                         "rtf": round(((n - 1) / FRAME_RATE) / o_decode(wall, cpu_fps, n), 3)},
        "parity": {"config0_greedy_f32_tokens_identical": bool(np.array_equal(got, exp)), "frames": int(n),
                   "oracle_min_top2_margin": min_margin,
                   "gpu_f32_decode_frames_per_s": round((st["frames"] - 1) / (st["decode_ms"] * 1e-3), 1)},
    }


def o_decode(wall, fps, n):
    return (n - 1) / fps


if __name__ == "__main__":
    main()
