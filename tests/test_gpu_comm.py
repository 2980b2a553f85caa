"""The RCCL communicator entry points of the C ABI (miseg_comm_*, miseg_allreduce_async): the gradient
sum of data-parallel training (/root/reference model_train.py:272-301).  One GPU per test box, so the
communicator has one rank: the all-reduce must leave the buffer as it is, asynchronously on the
given stream, and every call must report success.  (Two ranks on one device are refused by RCCL;
the multi-rank arithmetic is covered with gloo in tests/test_dist_cpu.py and tests/test_gpu_fit.py.)"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_comm_single_rank_allreduce():
    import miseg_amd
    from miseg_amd import lib as L
    lib = L.load()
    torch.cuda.set_device(0)
    L.check(lib.miseg_device_check())
    uid = (C.c_char * 128)()
    L.check(lib.miseg_comm_unique_id(uid))
    assert any(uid.raw), "empty unique id"
    comm = C.c_void_p()
    L.check(lib.miseg_comm_init(C.byref(comm), 0, 1, uid))
    assert comm.value
    g = torch.randn(3 * 1024 * 1024 + 17, device="cuda:0")
    want = g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    L.check(lib.miseg_allreduce_async(comm, g.data_ptr(), g.numel(), side.cuda_stream))
    L.check(lib.miseg_allreduce_async(comm, g.data_ptr(), 0, side.cuda_stream))      # empty bucket
    side.synchronize()
    assert torch.equal(g, want)
    assert lib.miseg_comm_init(C.byref(comm), 2, 1, uid) == L.E_INVALID_ARG        # rank outside the world
    L.check(lib.miseg_comm_destroy(comm))
    L.check(lib.miseg_comm_destroy(None))
