"""Worker of tests/test_fanout.py::test_comm_id_travels_through_the_launcher_store: the id exchange of fishrt.comm.RcclComm.from_env (the only
step of the RCCL bring-up outside the C ABI) under torch.distributed.run with world 2, on CPU -- no process group, no communicator."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fish-speech.rs_amd"))
from fishrt import comm

calls = []


def make_id():
    calls.append(1)
    return bytes((7 * i + 3) % 256 for i in range(comm.ID_BYTES))


uid, rank, world, store = comm.share_id(make_id, key="test_comm_id")
assert world == 2 and len(uid) == comm.ID_BYTES and uid == bytes((7 * i + 3) % 256 for i in range(comm.ID_BYTES))
assert len(calls) == (1 if rank == 0 else 0), "only rank 0 makes the id"
# a second communicator in the same job uses another key
uid2, _, _, _ = comm.share_id(lambda: b"\x01" * comm.ID_BYTES, key="test_comm_id_2")
assert uid2 == b"\x01" * comm.ID_BYTES
store.add("done", 1)
while int(store.add("done", 0)) < world:  # (rank 0 hosts the store under a bare launcher: do not leave before everybody has read)
    pass
print(f"COMM_STORE_OK rank {rank}")
