"""GPU: the C-ABI collectives (include/iamx.h iamx_comm_*: RCCL bound at run time) on a
one-rank communicator -- what a 1-GPU box can run; the exchanges themselves are exercised by the
world-size-2 gloo tests through torch.distributed."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_roundtrip():
    import torch
    from imageanalysis_amd import _lib
    from imageanalysis_amd._lib import check, stream_ptr
    L = _lib.lib()
    dev = _lib.require_gpu()
    uid = ctypes.create_string_buffer(128)
    check(L.iamx_comm_unique_id(uid), 'iamx_comm_unique_id')
    assert any(b != 0 for b in uid.raw)
    comm = ctypes.c_void_p()
    check(L.iamx_comm_init(1, 0, uid, ctypes.byref(comm)), 'iamx_comm_init')
    assert comm.value
    # descriptor-store shaped all-gather, out of place and in place
    src = torch.randint(-128, 128, (3 * 4096, 128), dtype=torch.int8, device=dev)
    dst = torch.zeros_like(src)
    nbytes = src.numel()
    check(L.iamx_comm_allgather(comm, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()),
                                nbytes, stream_ptr()), 'iamx_comm_allgather')
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    check(L.iamx_comm_allgather(comm, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(dst.data_ptr()),
                                nbytes, stream_ptr()), 'iamx_comm_allgather')
    # camera part of J^T u + one scalar: 7 C + 1 doubles
    buf = torch.arange(7 * 2812 + 1, dtype=torch.float64, device=dev) * 0.5
    want = buf.clone()
    check(L.iamx_comm_allreduce_f64(comm, ctypes.c_void_p(buf.data_ptr()), buf.numel(), stream_ptr()),
          'iamx_comm_allreduce_f64')
    torch.cuda.synchronize()
    assert torch.equal(buf, want) and torch.equal(src, dst)
    check(L.iamx_comm_destroy(comm), 'iamx_comm_destroy')


def test_comm_argument_checks():
    from imageanalysis_amd import _lib
    L = _lib.lib()
    assert L.iamx_comm_init(2, 5, ctypes.create_string_buffer(128), ctypes.byref(ctypes.c_void_p())) == -1
    assert b'bad rank' in L.iamx_last_error()
    assert L.iamx_comm_allreduce_f64(None, None, 4, None) == -1
