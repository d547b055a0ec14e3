"""The multi-GPU entry points of the C ABI that need no GPU: shard arithmetic and argument checking
(include/kfpos.h: kfpos_shard_range, kfpos_comm_*). The collective itself runs in tests/test_comm_gpu.py."""
import ctypes as C

import pytest

from roskfpos_amd import capi
from roskfpos_amd.dist import shard_range, shard_sizes


@pytest.mark.parametrize("total", [8, 9, 97, 65536 * 8, 1048576, 1000003])
@pytest.mark.parametrize("world", [1, 2, 3, 7, 8])
def test_c_abi_and_python_cut_the_batch_alike(total, world):
    spans = [capi.shard_range(total, world, r) for r in range(world)]
    assert spans == [shard_range(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    sizes = shard_sizes(total, world)
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)  # the larger shards come first


def test_shard_range_rejects_nonsense():
    lib = capi.load()
    lo, hi = C.c_int64(), C.c_int64()
    assert lib.kfpos_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) == 1      # KFPOS_ERR_ARG
    assert lib.kfpos_shard_range(10, 2, 2, C.byref(lo), C.byref(hi)) == 1
    assert lib.kfpos_shard_range(10, 2, 0, None, C.byref(hi)) == 1


def test_comm_argument_errors_without_touching_rccl():
    lib = capi.load()
    c = C.c_void_p()
    uid = b"\0" * capi.COMM_ID_BYTES
    assert lib.kfpos_comm_create(0, 0, uid, 0, C.byref(c)) == 1
    assert lib.kfpos_comm_create(2, 2, uid, 0, C.byref(c)) == 1
    assert lib.kfpos_comm_create(2, 0, None, 0, C.byref(c)) == 1
    assert lib.kfpos_comm_create(2, 0, uid, 0, None) == 1
    assert lib.kfpos_comm_destroy(None) == 1 and lib.kfpos_comm_sync(None) == 1 and lib.kfpos_comm_wait(None, None) == 1
    assert lib.kfpos_comm_world(None) == 0 and lib.kfpos_comm_rank(None) == -1
    assert lib.kfpos_allgather_poses(None, None, None, 3, None, None) == 1
    assert lib.kfpos_comm_set_algorithm(None, 1) == 1 and lib.kfpos_comm_algorithm(None) == -1
    assert lib.kfpos_strerror(6).decode() == "RCCL error"


def test_unique_id_comes_from_rccl_when_it_can_be_opened():
    """librccl is opened on first use. Where it can be, kfpos_comm_unique_id hands out 128 bytes that differ from call
    to call; where it cannot, the error says what to set."""
    lib = capi.load()
    if lib.kfpos_comm_backend_version() == 0:
        with pytest.raises(capi.KfposError, match="KFPOS_RCCL_PATH"):
            capi.comm_unique_id()
        return
    assert lib.kfpos_comm_backend_version() >= 20000
    a, b = capi.comm_unique_id(), capi.comm_unique_id()
    assert len(a) == len(b) == capi.COMM_ID_BYTES == 128 and a != b
