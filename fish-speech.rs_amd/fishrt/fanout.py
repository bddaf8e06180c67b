"""Multi-GPU request fan-out: the reference has no distributed layer, requests are independent
sequences, so the only multi-GPU mode is REPLICAS -- one process per GPU, each with its own fishrt handle; request i goes
to rank i mod world.  torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests) is used only for control:
barrier, max-reduce of the timed region, the start-up weight broadcast, the prompt broadcast and the fan-in of the (KB-sized)
code arrays.  Nothing on the per-token path crosses GPUs."""
import os


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_requests(n_requests, rank, world):
    """Indices of the requests served by `rank` (request i -> rank i mod world)."""
    return list(range(rank, n_requests, world))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns the module (or None when world == 1)."""
    rank, local_rank, world = env_rank()
    if world == 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value):
    """MAX-reduce a python float over all ranks (the timed region of the whole job is the slowest rank's)."""
    if dist is None:
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(dist, value):
    """MIN-reduce a python number over all ranks (e.g. a success flag: 1 only if every rank succeeded)."""
    if dist is None:
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def sum_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_results(dist, n_requests, local_results):
    """Fan-in: `local_results` maps request index -> numpy codes (C, n_i) for this rank's shard.  Rank 0 gets the full
    list ordered by request index (others get None)."""
    if dist is None:
        return [local_results[i] for i in range(n_requests)]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local_results, out, dst=0)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for d in out:
        merged.update(d)
    return [merged[i] for i in range(n_requests)]


def _dev(dist):
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


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
    import torch
    me = dist.get_rank()
    if hasattr(lm, "weights_host"):  # host-memory double (gloo tests)
        t = torch.from_numpy(lm.weights_host())
    else:
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
    import torch
    dev, world = _dev(dist), dist.get_world_size()
    c = torch.from_numpy(codes.view(np.int32)).to(dev)
    f = torch.from_numpy(n_frames).to(dev)
    co = [torch.empty_like(c) for _ in range(world)]
    fo = [torch.empty_like(f) for _ in range(world)]
    dist.all_gather(co, c)
    dist.all_gather(fo, f)
    return torch.stack(co).cpu().numpy().view(np.uint32), torch.stack(fo).cpu().numpy(), world
