"""GPU (one MI355X): the fs_comm_* C entry points (include/fishrt.h; csrc/fs_comm.cpp on librccl directly) through a ONE-rank RCCL
communicator -- a legal communicator, and the only one a one-GPU box can build (RCCL refuses two ranks on one device; the 2-rank
rehearsal of the same call sequence runs on gloo in tests/test_bench_rehearsal_gpu.py and tests/test_fanout.py).  Every collective the
replica fan-out uses is issued for real (ncclCommInitRank, ncclAllReduce, ncclBroadcast, ncclAllGather) and checked for its world-1 value;
then `bench.py --gpus 1` and `--config 3` run their whole fan-out -- weight broadcast + adopt, prompt broadcast, barriers, max-over-ranks,
code all-gather -- through such a communicator (FISHRT_BENCH_COMM1)."""
import numpy as np
import pytest

import fishrt
from fishrt import config as fcfg, fanout
from fishrt.comm import RcclComm, unique_id, SUM, MAX, MIN

from test_bench_rehearsal_gpu import run_bench, check_config1, check_config3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    uid = unique_id()
    assert len(uid) == 128 and any(uid)
    c = RcclComm(uid, 0, 1, 0)
    yield c
    c.close()


def test_one_rank_collectives(comm):
    assert comm.rank == 0 and comm.world == 1 and fanout.backend_name(comm).startswith("rccl")
    comm.barrier()
    v = np.array([1.5, -2.0, 7.25])
    for op in (SUM, MAX, MIN):
        assert np.array_equal(comm.all_reduce(v, op), v)
    assert fanout.max_over_ranks(comm, 3.5) == 3.5 and fanout.sum_over_ranks(comm, 4) == 4.0 and fanout.min_over_ranks(comm, 1) == 1.0
    rng = np.random.RandomState(3)
    packed = rng.randint(0, 1 << 30, (37, 9, 211)).astype(np.uint32)
    lens = rng.randint(1, 212, 37).astype(np.int32)
    pk, ln = fanout.broadcast_prompts(comm, packed, lens)
    assert np.array_equal(pk, packed) and np.array_equal(ln, lens)
    codes = rng.randint(0, 1024, (5, 8, 33)).astype(np.uint32)
    nf = rng.randint(1, 34, 5).astype(np.int32)
    ca, fa, seen = fanout.all_gather_codes(comm, codes, nf)
    assert seen == 1 and ca.shape == (1, 5, 8, 33) and np.array_equal(ca[0], codes) and np.array_equal(fa[0], nf)
    res = fanout.gather_results(comm, 3, {i: codes[i, :, : nf[i]] for i in range(3)})
    assert all(np.array_equal(res[i], codes[i, :, : nf[i]]) for i in range(3))


def test_weight_broadcast_through_the_c_abi(comm):
    """fs_comm_broadcast_weights at world 1: sizes agreed through ncclAllReduce, the arena goes through ncclBroadcast in place, the source
    handle is untouched -- same greedy tokens before and after"""
    lm = fishrt.DualARTransformer(fcfg.FISH_1_5, fcfg.FISH_1_5_TOKENS, 0, "bf16").load_synthetic(0xF15E5EED)
    p = np.zeros((9, 12), np.uint32)
    p[0] = np.arange(12) * 37 % 1000
    kw = dict(temp=0.0, top_p=1.0, top_k=0, repetition_penalty=1.2, ignore_eos=True)
    a = lm.generate_blocking(p, 12 + 6, **kw)
    moved = fanout.broadcast_weights(comm, lm, src=0)
    assert moved == lm.weights_arena()[1] and moved > 1.2e9
    lm.clear_slow_layer_caches()
    assert np.array_equal(a, lm.generate_blocking(p, 12 + 6, **kw))


def test_bad_arguments_are_errors(comm):
    from fishrt import _ffi
    import ctypes as C
    L = _ffi.lib()
    assert L.fs_comm_rank(None) == -1 and L.fs_comm_barrier(None) != 0
    h = C.c_void_p()
    assert L.fs_comm_create((C.c_uint8 * 128)(), 3, 2, 0, C.byref(h)) != 0 and b"rank" in L.fs_last_error()
    v = (C.c_double * 1)(1.0)
    assert L.fs_comm_all_reduce_f64(comm._h, v, 1, 7) != 0


def test_bench_config1_through_a_one_rank_communicator():
    j = run_bench(["--gpus", "1", "--frames", "32", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], {"FISHRT_BENCH_COMM1": "1"})
    assert j["fanout_backend"].startswith("rccl") and j["rccl_ranks"] == 1
    wb = j["weight_broadcast"]
    assert "error" not in wb and wb["bytes"] > 1.2e9 and wb["how"].startswith("fs_comm_broadcast_weights")
    j["weight_broadcast"] = None
    check_config1(j, 1, 32, 1, 1)


def test_bench_config3_through_a_one_rank_communicator():
    j = run_bench(["--config", "3", "--gpus", "1", "--frames", "8", "--steps", "1", "--warmup", "0"], {"FISHRT_BENCH_COMM1": "1"})
    assert j["fanout_backend"].startswith("rccl")
    check_config3(j, 1, 8)
