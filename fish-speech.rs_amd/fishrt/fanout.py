"""Multi-GPU request fan-out: the reference has no distributed layer, requests are independent
sequences, so the only multi-GPU mode is REPLICAS -- one process per GPU, each with its own fishrt handle; request i goes
to rank i mod world.  Nothing on the per-token path crosses GPUs; what does cross is the start-up weight broadcast, the prompt
broadcast, the fan-in of the (KB-sized) code arrays, and the barrier / max-reduce of the timed region.

Two carriers, one function set (every function takes the `dist` object init() returned, or None at world 1):
 * fishrt.comm.RcclComm -- the fs_comm_* C entry points of libfishrt.so on librccl DIRECTLY (ncclBroadcast / ncclAllGather /
   ncclAllReduce over xGMI): the GPU path, and what a Rust host binds.  torch is used only for the launcher's key-value store that carries
   the 128-byte communicator id; no torch process group exists.
 * the torch.distributed module with the gloo backend -- CPU tests (world 2 without a GPU) and the one-GPU rehearsal of N > 1
   (RCCL refuses two ranks on one device)."""
import os

from .comm import RcclComm, MAX, MIN, SUM


def _is_rccl(dist):
    return isinstance(dist, RcclComm)


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_requests(n_requests, rank, world):
    """Indices of the requests served by `rank` (request i -> rank i mod world)."""
    return list(range(rank, n_requests, world))


def init(backend=None, device=None):
    """world 1 -> None.  backend "nccl" / "rccl" (default when a HIP device is visible) -> an RcclComm created from the torchrun environment
    (fs_comm_create on device LOCAL_RANK, or `device`); "gloo" -> the initialised torch.distributed module."""
    rank, local_rank, world = env_rank()
    if world == 1:
        return None
    if backend is None:
        from . import _ffi
        backend = "rccl" if _ffi.lib().fs_device_count() > 0 else "gloo"
    dev = local_rank if device is None else device
    if backend in ("nccl", "rccl"):
        try:
            return RcclComm.from_env(device=dev)
        except Exception as e:  # noqa: BLE001 -- agreed by every rank (RcclComm.from_env votes through the store): all fall back together
            import sys
            print(f"fishrt.fanout: fs_comm_* bring-up failed on rank {rank} ({e}); falling back to torch.distributed's RCCL backend", file=sys.stderr, flush=True)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        return dist
    import torch.distributed as dist
    dist.init_process_group(backend)
    return dist


def backend_name(dist):
    if dist is None:
        return None
    return "rccl (fs_comm_* C ABI on librccl)" if _is_rccl(dist) else dist.get_backend()


def barrier(dist):
    if dist is not None:
        dist.barrier()


def _reduce(dist, value, op):
    if dist is None:
        return float(value)
    if _is_rccl(dist):
        return float(dist.all_reduce([float(value)], op)[0])
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_dev(dist))
    dist.all_reduce(t, op={MAX: dist.ReduceOp.MAX, MIN: dist.ReduceOp.MIN, SUM: dist.ReduceOp.SUM}[op])
    return float(t.item())


def max_over_ranks(dist, value):
    """MAX-reduce a python float over all ranks (the timed region of the whole job is the slowest rank's)."""
    return _reduce(dist, value, MAX)


def min_over_ranks(dist, value):
    """MIN-reduce a python number over all ranks (e.g. a success flag: 1 only if every rank succeeded)."""
    return _reduce(dist, value, MIN)


def sum_over_ranks(dist, value):
    return _reduce(dist, value, SUM)


def gather_results(dist, n_requests, local_results):
    """Fan-in: `local_results` maps request index -> numpy codes (C, n_i) for this rank's shard.  Rank 0 gets the full
    list ordered by request index (others get None).  (RCCL carrier: padded to the longest result and all-gathered.)"""
    if dist is None:
        return [local_results[i] for i in range(n_requests)]
    if _is_rccl(dist):
        import numpy as np
        per = (n_requests + dist.world - 1) // dist.world
        C = next(iter(local_results.values())).shape[0] if local_results else 1
        nmax = int(max_over_ranks(dist, max([v.shape[1] for v in local_results.values()], default=0)))
        C = int(max_over_ranks(dist, C))
        codes = np.zeros((per, C, nmax), np.uint32)
        nf = np.full(per, -1, np.int32)
        for k, i in enumerate(shard_requests(n_requests, dist.rank, dist.world)):
            codes[k, :, : local_results[i].shape[1]] = local_results[i]
            nf[k] = local_results[i].shape[1]
        ca, fa, _ = dist.all_gather_codes(codes, nf)
        if dist.rank != 0:
            return None
        out = [None] * n_requests
        for r in range(dist.world):
            for k, i in enumerate(shard_requests(n_requests, r, dist.world)):
                out[i] = ca[r, k, :, : fa[r, k]].copy()
        return out
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local_results, out, dst=0)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for d in out:
        merged.update(d)
    return [merged[i] for i in range(n_requests)]


def _dev(dist):
    return "cuda" if dist.get_backend() == "nccl" else "cpu"  # (gloo: host tensors; "nccl" only as the fallback carrier of init())


class _DeviceBytes:
    """a raw device allocation as a CUDA-array-interface object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_weights(dist, lm, src=0, chunk_bytes=256 << 20):
    """rank `src` has loaded the checkpoint; every other rank's handle (same model args / dtype, not loaded) receives the
    weight arena over the communicator (RCCL broadcast over xGMI: one checkpoint read + conversion instead of N) and adopts it.  `lm` is a
    DualARTransformer, or any object with weights_arena() -> (ptr, nbytes) / adopt_weights() (tests pass a host-memory double:
    weights_host() -> numpy u8).  Returns the bytes moved."""
    if dist is None:
        return 0
    if _is_rccl(dist):
        return dist.broadcast_weights(lm, src)
    import torch
    me = dist.get_rank()
    if hasattr(lm, "weights_host"):  # host-memory double (gloo tests)
        t = torch.from_numpy(lm.weights_host())
    else:  # gloo with real handles (the one-GPU rehearsal): the arena as a CUDA tensor, gloo stages it through the host
        ptr, n = lm.weights_arena()
        t = torch.as_tensor(_DeviceBytes(ptr, n), device="cuda")
    n = int(t.numel())
    sizes = torch.tensor([n, -n], dtype=torch.int64, device=t.device)  # max and -min in one reduction: every rank sees a mismatch
    dist.all_reduce(sizes, op=dist.ReduceOp.MAX)
    hi, lo = int(sizes[0].item()), -int(sizes[1].item())
    if hi != lo:
        raise RuntimeError(f"weight arenas differ across ranks ({lo}..{hi} bytes, {n} here): same model args and dtype on every rank")
    for o in range(0, n, chunk_bytes):  # chunks: bounded communicator staging, overlappable launches
        dist.broadcast(t[o:o + chunk_bytes], src)
    if t.is_cuda:
        torch.cuda.synchronize()
    if me != src:
        lm.adopt_weights()
    return n


def broadcast_prompts(dist, packed=None, lens=None, src=0):
    """the packed prompt batch -- u32 [n_req, C+1, Lmax] (left-aligned rows) + lengths [n_req] -- goes from
    `src` to every rank with two broadcasts (<= 3.5 MB for 256 x 9 x 384); every rank then serves shard_requests() of it."""
    import numpy as np
    if dist is None:
        return np.ascontiguousarray(packed, np.uint32), np.asarray(lens, np.int32)
    if _is_rccl(dist):
        return dist.broadcast_prompts(packed, lens, src)
    import torch
    dev, me = _dev(dist), dist.get_rank()
    shape = torch.tensor(list(packed.shape) if me == src else [0, 0, 0], dtype=torch.int64, device=dev)
    dist.broadcast(shape, src)
    n, c1, lmax = (int(v) for v in shape.tolist())
    if me == src:
        buf = torch.from_numpy(np.ascontiguousarray(packed, np.uint32).view(np.int32)).to(dev)
        ln = torch.from_numpy(np.asarray(lens, np.int32)).to(dev)
    else:
        buf = torch.empty((n, c1, lmax), dtype=torch.int32, device=dev)
        ln = torch.empty((n,), dtype=torch.int32, device=dev)
    dist.broadcast(buf, src)
    dist.broadcast(ln, src)
    return buf.cpu().numpy().view(np.uint32), ln.cpu().numpy()


def all_gather_codes(dist, codes, n_frames):
    """end-of-run fan-in.  codes u32 [B, C, N] (this rank's requests, padded to N frames), n_frames i32 [B]
    -> (codes_all [world, B, C, N], n_frames_all [world, B], ranks_seen) on EVERY rank: one all-gather each over RCCL (gloo in the
    CPU tests).  ranks_seen = the communicator's world size, reported by the bench."""
    import numpy as np
    codes = np.ascontiguousarray(codes, np.uint32)
    n_frames = np.ascontiguousarray(n_frames, np.int32)
    if dist is None:
        return codes[None], n_frames[None], 1
    if _is_rccl(dist):
        return dist.all_gather_codes(codes, n_frames)
    import torch
    dev, world = _dev(dist), dist.get_world_size()
    c = torch.from_numpy(codes.view(np.int32)).to(dev)
    f = torch.from_numpy(n_frames).to(dev)
    co = [torch.empty_like(c) for _ in range(world)]
    fo = [torch.empty_like(f) for _ in range(world)]
    dist.all_gather(co, c)
    dist.all_gather(fo, f)
    return torch.stack(co).cpu().numpy().view(np.uint32), torch.stack(fo).cpu().numpy(), world
