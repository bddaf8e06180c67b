"""CPU: the N > 1 (replica fan-out) control path of bench.py / fishrt.fanout under torch.distributed with the gloo
backend, world_size 2 (one process per 'GPU').  Nothing on the per-token path is distributed."""
import os
import subprocess
import sys

from conftest import ROOT

from fishrt import fanout


def test_shard_requests_partition():
    for world in (1, 2, 8):
        seen = sorted(i for r in range(world) for i in fanout.shard_requests(256, r, world))
        assert seen == list(range(256))
        assert all(len(fanout.shard_requests(256, r, world)) == 256 // world for r in range(world))
    assert fanout.max_over_ranks(None, 2.5) == 2.5 and fanout.gather_results(None, 2, {0: "a", 1: "b"}) == ["a", "b"]


def test_gloo_world2_fanout():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py"), "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "FANOUT_OK 2" in p.stdout
