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
dist.destroy_process_group()
