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
    assert "FANOUT_OK 2" in p.stdout and "FANIN_OK 2" in p.stdout and "WEIGHTS_OK 2" in p.stdout


def test_comm_id_travels_through_the_launcher_store():
    """fishrt.comm.share_id: the one step of the RCCL bring-up outside the C ABI (rank 0's fs_comm_unique_id bytes -> every rank) through the
    launcher's key-value store, world 2 under torch.distributed.run, no process group -- what RcclComm.from_env does on the GPU box"""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "tests", "comm_store_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "COMM_STORE_OK rank 0" in p.stdout and "COMM_STORE_OK rank 1" in p.stdout


def test_bench_gpus2_self_launch_dry_run():
    """`python bench.py --gpus 2` without torchrun starts its own ranks; on a box without an MI355X the ranks run the control path
    only (shard, prompt broadcast, code all-gather over gloo), print a JSON line flagged dry_run and exit 0."""
    import json
    import fishrt
    if fishrt.lib().fs_device_count() > 0:  # a GPU is visible: the ranks would run the real bench -- that IS tests/test_bench_rehearsal_gpu.py
        import test_bench_rehearsal_gpu as reh
        return reh.test_bench_config1_two_ranks_on_one_gpu()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["collective_ranks"] == 2 and j["fan_in_ok"] and j["requests_per_rank"] == [128, 128]
    # a mismatching launcher environment is a usage error, not an assert
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env2)
    assert p2.returncode == 2 and "WORLD_SIZE=1" in p2.stderr


def test_bench_gpus8_config3_dry_run():
    """BASELINE.json configs[3] as the driver would launch it on a node (`--gpus 8 --config 3`): 256 requests -> 32 per rank = one static
    batch of 32 per GPU; on a CPU box the 8 ranks run the control path (shard, prompt broadcast, code all-gather over gloo)."""
    import json
    import fishrt
    if fishrt.lib().fs_device_count() > 0:  # one visible GPU cannot host 8 real ranks: the 2-rank rehearsal of the same code path instead
        import test_bench_rehearsal_gpu as reh
        return reh.test_bench_config3_two_ranks_on_one_gpu()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["dry_run"] is True and j["config"] == 3 and j["n_gpus"] == 8 and j["collective_ranks"] == 8 and j["fan_in_ok"]
    assert j["requests_per_rank"] == [32] * 8 and j["static_batches_per_rank"] == [1] * 8 and j["batch_per_rank"] == 32
