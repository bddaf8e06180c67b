"""Worker for tests/test_fanout.py: run under torch.distributed.run with the gloo backend (CPU, no GPU needed)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
from fishrt import fanout

dist = fanout.init("gloo")
rank, _, world = fanout.env_rank()
assert dist is not None and world == int(sys.argv[1])
n_req = 7
mine = fanout.shard_requests(n_req, rank, world)
assert mine == list(range(rank, n_req, world))
# every request is served exactly once across ranks
served = fanout.sum_over_ranks(dist, len(mine))
assert served == n_req, served
# the job's timed region is the slowest rank's
assert fanout.max_over_ranks(dist, 1.0 + rank) == float(world)
fanout.barrier(dist)
# fan-in of ragged per-request code arrays (C, n_i)
local = {i: np.full((8, 3 + i), i, np.uint32) for i in mine}
allr = fanout.gather_results(dist, n_req, local)
if rank == 0:
    assert len(allr) == n_req
    for i, a in enumerate(allr):
        assert a.shape == (8, 3 + i) and (a == i).all()
    print("FANOUT_OK", world, flush=True)
else:
    assert allr is None
# SURVEY.md 8e (2) + (3): prompt broadcast, per-rank shard, all-gather of the padded code arrays
packed = (np.arange(6 * 9 * 5, dtype=np.uint32).reshape(6, 9, 5) * 7 + 3) if rank == 0 else None
lens = np.array([5, 4, 3, 5, 2, 1], np.int32) if rank == 0 else None
pk, ln = fanout.broadcast_prompts(dist, packed, lens)
assert pk.shape == (6, 9, 5) and pk.dtype == np.uint32 and int(pk[5, 8, 4]) == (6 * 9 * 5 - 1) * 7 + 3 and ln.tolist() == [5, 4, 3, 5, 2, 1]
mine = fanout.shard_requests(6, rank, world)
codes = np.stack([np.full((8, 4), 100 * i + 1, np.uint32) for i in mine])
nf = np.array([1 + (i % 4) for i in mine], np.int32)
ca, fa, seen = fanout.all_gather_codes(dist, codes, nf)
assert seen == world and ca.shape == (world, len(mine), 8, 4) and fa.shape == (world, len(mine))
for r in range(world):
    for k, i in enumerate(fanout.shard_requests(6, r, world)):
        assert (ca[r, k] == 100 * i + 1).all() and fa[r, k] == 1 + (i % 4)
if rank == 0:
    print("FANIN_OK", world, flush=True)


# weight-arena broadcast -- rank 0 "loaded the checkpoint", the others receive the bytes and adopt them
class HostArena:
    def __init__(self, n, loaded):
        self.buf = (np.arange(n, dtype=np.uint64) * 2654435761 % 251).astype(np.uint8) if loaded else np.zeros(n, np.uint8)
        self.adopted = False

    def weights_host(self):
        return self.buf

    def adopt_weights(self):
        self.adopted = True


h = HostArena(3_000_001, rank == 0)
moved = fanout.broadcast_weights(dist, h, src=0, chunk_bytes=1 << 20)  # 3 chunks, the last one ragged
ref = HostArena(3_000_001, True).buf
assert moved == 3_000_001 and np.array_equal(h.buf, ref) and h.adopted == (rank != 0)
try:  # arenas of different sizes (different model args on some rank) are an error on every rank, not a hang
    fanout.broadcast_weights(dist, HostArena(1000 + rank, True))
    raise SystemExit("size mismatch not detected")
except RuntimeError as e:
    assert "differ across ranks" in str(e)
if rank == 0:
    print("WEIGHTS_OK", world, flush=True)
dist.destroy_process_group()
