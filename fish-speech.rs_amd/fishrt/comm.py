"""ctypes mirror of the fs_comm_* entry points (include/fishrt.h): the replica fan-out on librccl directly -- what a Rust host binds
(INTEGRATION.md section 4).  The only thing NOT in the C ABI is how the 128-byte communicator id travels from rank 0 to the other
ranks: that is the host's own channel; `from_env()` uses the launcher's rendezvous key-value store (torch.distributed's TCPStore, the
one torchrun's MASTER_ADDR / MASTER_PORT point at) and creates no torch process group."""
import ctypes as C
import os

import numpy as np

from . import _ffi

ID_BYTES = 128
SUM, MAX, MIN = 0, 1, 2


def unique_id():
    buf = (C.c_uint8 * ID_BYTES)()
    _ffi.check(_ffi.lib().fs_comm_unique_id(buf))
    return bytes(buf)


def share_id(make_id, key="fishrt_comm_id"):
    """The one step of the bring-up that is NOT in the C ABI: rank 0 calls make_id() (fs_comm_unique_id) and the 128 bytes reach every rank
    through the launcher's key-value store (torchrun's MASTER_ADDR / MASTER_PORT; under torchrun the agent hosts it).  Returns
    (id bytes, rank, world, store).  No torch process group is created.  (tests/test_fanout.py runs this with world 2 on CPU.)"""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world == 1:
        return make_id(), 0, 1, None
    from torch.distributed import rendezvous
    store, rank, world = next(rendezvous("env://", rank, world))
    if rank == 0:
        store.set(key, make_id())
    return bytes(store.get(key)), rank, world, store


class RcclComm:
    """one RCCL communicator of this process's GPU (fs_comm_t)"""

    def __init__(self, uid, rank, world, device):
        assert len(uid) == ID_BYTES
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().fs_comm_create((C.c_uint8 * ID_BYTES).from_buffer_copy(uid), int(rank), int(world), int(device), C.byref(self._h)))
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        assert _ffi.lib().fs_comm_rank(self._h) == self.rank and _ffi.lib().fs_comm_world(self._h) == self.world

    @classmethod
    def from_env(cls, device=None, key="fishrt_comm_id"):
        """ranks started by torchrun (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT): rank 0 publishes the id in the launcher's store"""
        dev = int(os.environ.get("LOCAL_RANK", 0)) if device is None else int(device)
        uid, rank, world, store = share_id(unique_id, key)
        c, err = None, None
        try:
            c = cls(uid, rank, world, dev)
        except Exception as e:  # noqa: BLE001 -- voted on below: a rank that failed alone must not leave the others inside a collective
            err = e
        if store is not None:  # every rank learns whether EVERY rank has its communicator before anybody uses (or abandons) it
            store.set(f"{key}_ok_{rank}", b"1" if c is not None else b"0")
            ok = all(bytes(store.get(f"{key}_ok_{r}")) == b"1" for r in range(world))
            if not ok:
                if c is not None:
                    c.close()
                raise RuntimeError(f"fs_comm_create failed on at least one rank (this rank: {err or 'ok'})")
        elif c is None:
            raise err
        c._store = store  # (keeps the store's server on rank 0 alive for the slower ranks)
        return c

    def close(self):
        if self._h:
            _ffi.lib().fs_comm_destroy(self._h)
            self._h = C.c_void_p()

    destroy_process_group = close  # (so that callers can treat it like a torch.distributed module)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        _ffi.check(_ffi.lib().fs_comm_barrier(self._h))

    def all_reduce(self, values, op):
        v = np.ascontiguousarray(values, np.float64).copy()
        _ffi.check(_ffi.lib().fs_comm_all_reduce_f64(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size), int(op)))
        return v

    def broadcast_weights(self, lm, src=0):
        n = C.c_size_t(0)
        _ffi.check(_ffi.lib().fs_comm_broadcast_weights(self._h, lm._h, int(src), C.byref(n)))
        return int(n.value)

    def broadcast_prompts(self, packed=None, lens=None, src=0):
        dims = np.zeros(3, np.int64)
        if self.rank == src:
            packed = np.ascontiguousarray(packed, np.uint32)
            lens = np.ascontiguousarray(lens, np.int32)
            dims[:] = packed.shape
        _ffi.check(_ffi.lib().fs_comm_broadcast_prompt_dims(self._h, dims.ctypes.data_as(C.POINTER(C.c_int64)), int(src)))
        if self.rank != src:
            packed = np.zeros(tuple(int(d) for d in dims), np.uint32)
            lens = np.zeros(int(dims[0]), np.int32)
        _ffi.check(_ffi.lib().fs_comm_broadcast_prompts(self._h, packed.ctypes.data_as(C.POINTER(C.c_uint32)), lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       dims.ctypes.data_as(C.POINTER(C.c_int64)), int(src)))
        return packed, lens

    def all_gather_codes(self, codes, n_frames):
        codes = np.ascontiguousarray(codes, np.uint32)
        n_frames = np.ascontiguousarray(n_frames, np.int32)
        B, Cc, N = codes.shape
        ca = np.zeros((self.world, B, Cc, N), np.uint32)
        fa = np.zeros((self.world, B), np.int32)
        _ffi.check(_ffi.lib().fs_comm_all_gather_codes(self._h, codes.ctypes.data_as(C.POINTER(C.c_uint32)), n_frames.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      B, Cc, N, ca.ctypes.data_as(C.POINTER(C.c_uint32)), fa.ctypes.data_as(C.POINTER(C.c_int32))))
        return ca, fa, self.world
