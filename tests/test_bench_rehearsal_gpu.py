"""GPU (one MI355X): rehearsal of the bench lines the driver takes on an 8-GPU node -- `bench.py --config 3` (BASELINE.json configs[3]) and the
N > 1 launch of both configs -- on the ONE device a test box has.  FISHRT_BENCH_DEVICE=0 pins every rank to device 0 and
FISHRT_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU), so everything else is the real path: bench.py's own self-launch
(torch.distributed.run), the weight-arena broadcast into device memory + fs_lm_weights_adopt, the prompt broadcast, request sharding, the
timed region with its barriers / max-over-ranks, the code all-gather and the JSON line.  FISHRT_PERSIST_WAIT=1: the ranks share the device's
persistent kernels one request at a time (csrc/lm_engine.hip PersistLock) instead of one of them dropping to the per-node graphs, so the
replicas' greedy tokens are comparable bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(argv, extra_env=None, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}"
    return json.loads(lines[0])


ONE_GPU = {"FISHRT_BENCH_DEVICE": "0", "FISHRT_BENCH_BACKEND": "gloo", "FISHRT_PERSIST_WAIT": "1"}


def check_common(j, n_gpus, steps, warmup):
    assert j.get("dry_run") is None and j["n_gpus"] == n_gpus and j["steps"] == steps and j["warmup"] == warmup
    assert j["unit"] == "frames/s" and j["higher_is_better"] is True and j["dtype"] == "bf16" and j["vs_baseline"] is None
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["rccl_ranks"] == n_gpus
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def check_config3(j, n_gpus, frames, steps=1, warmup=0):
    check_common(j, n_gpus, steps, warmup)
    assert j["scaling"] == "strong" and "configs[3]" in j["config"]["workload"]
    assert j["config"]["requests"] == 256 and j["config"]["batch_per_gpu"] == 32 and j["config"]["frames_per_request"] == frames
    assert j["frames_per_rank"] == [256 // n_gpus * frames] * n_gpus  # every request served exactly once, every frame fanned in
    # value is whole-job throughput: all requests' frames over the max-over-ranks wall time of the timed region
    assert abs(j["value"] - 256 * frames * steps / (j["ms_per_step"] * 1e-3 * steps)) / j["value"] < 1e-3
    assert j["decode_step_us_rank0"] > 0 and j["roofline"]["algorithmic_bytes_per_step"] > 1.6e9


def check_config1(j, n_gpus, frames, steps, warmup):
    check_common(j, n_gpus, steps, warmup)
    assert j["scaling"] == "weak" and "configs[1]" in j["config"]["workload"] and j["config"]["requests_per_step"] == n_gpus
    assert j["frames_per_rank"] == [frames * steps] * n_gpus
    assert abs(j["value"] - frames * steps * n_gpus / (j["ms_per_step"] * 1e-3 * steps)) / j["value"] < 1e-3
    assert j["roofline"]["kernels_per_frame"] == 2, "a rank fell off the persistent kernels"
    if n_gpus > 1:  # replica start-up over the communicator: the whole bf16 arena, adopted by the receivers (tokens compared inside bench.py)
        wb = j["weight_broadcast"]
        assert "error" not in wb and wb["bytes"] > 1.2e9 and "fs_lm_weights_adopt" in wb["how"]
    else:
        assert j["weight_broadcast"] is None


def test_bench_config3_one_gpu():
    """`bench.py --config 3 --gpus 1`: the 256 requests as 8 static batches of 32 on one device (what each of 8 ranks does once)"""
    check_config3(run_bench(["--config", "3", "--gpus", "1", "--frames", "16", "--steps", "1", "--warmup", "0"]), 1, 16)


def test_bench_config3_two_ranks_on_one_gpu():
    """`bench.py --config 3 --gpus 2` (self-launch): prompts broadcast from rank 0, 128 requests per rank, codes all-gathered"""
    check_config3(run_bench(["--config", "3", "--gpus", "2", "--frames", "8", "--steps", "1", "--warmup", "0"], ONE_GPU), 2, 8)


def test_bench_config1_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` (the headline configuration, replicas): rank 0 loads, rank 1 receives the arena and adopts it; bench.py itself
    asserts that the replicas' greedy tokens are identical and deterministic across requests"""
    check_config1(run_bench(["--gpus", "2", "--frames", "48", "--steps", "2", "--warmup", "1"], ONE_GPU), 2, 48, 2, 1)


def test_two_processes_one_gpu_share_the_persistent_kernels():
    """the persistent kernels need every CU of the device: a SECOND PROCESS that asks for them while a call of the first holds them takes the
    per-node graphs (kernels_per_frame 266) instead of timing out in a half-resident launch -- and with FISHRT_PERSIST_WAIT it waits its turn"""
    code = r'''
import sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/fish-speech.rs_amd")
import numpy as np, fishrt
from fishrt import config as fcfg
lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
p = np.zeros((9, 32), np.uint32); p[0] = np.random.RandomState(3).randint(0, 100000, 32)
open(sys.argv[2] + ".ready", "w").close()
while not all(__import__("os").path.exists(f + ".ready") for f in sys.argv[3:]): time.sleep(0.01)
seen = set()
for _ in range(6):
    lm.clear_slow_layer_caches()
    o = lm.generate_blocking(p, 32 + 190, temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    seen.add(int(lm.last_stats()["kernels_per_frame"]))
print("KPF", sorted(seen), "SUM", int(o.astype(np.int64).sum()), flush=True)
'''
    import tempfile
    for wait in (False, True):
        with tempfile.TemporaryDirectory() as td:
            tags = [os.path.join(td, f"p{i}") for i in range(2)]
            env = {k: v for k, v in os.environ.items() if k != "FISHRT_PERSIST_WAIT"}
            if wait:
                env["FISHRT_PERSIST_WAIT"] = "1"
            ps = [subprocess.Popen([sys.executable, "-c", code, ROOT, tags[i]] + tags, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
                  for i in range(2)]
            outs = [p.communicate(timeout=600) for p in ps]
        assert all(p.returncode == 0 for p in ps), [o[1][-1500:] for o in outs]
        kpf = [set(json.loads(o[0].split("KPF ")[1].split(" SUM")[0])) for o in outs]
        print("wait" if wait else "try", "kernels per frame seen by the two processes:", kpf)
        if wait:
            assert kpf == [{2}, {2}], kpf
            sums = [o[0].split("SUM ")[1].split()[0] for o in outs]
            assert sums[0] == sums[1], "two processes on the same path disagree on greedy tokens"
        else:
            assert all(k <= {2, 266} for k in kpf) and any(2 in k for k in kpf), kpf
