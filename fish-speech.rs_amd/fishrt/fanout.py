"""Multi-GPU request fan-out: the reference has no distributed layer, requests are independent
sequences, so the only multi-GPU mode is REPLICAS -- one process per GPU, each with its own fishrt handle; request i goes
to rank i mod world.  torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests) is used only for control:
barrier, max-reduce of the timed region, and the fan-in of the (KB-sized) code arrays.  Nothing on the per-token path
crosses GPUs."""
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
