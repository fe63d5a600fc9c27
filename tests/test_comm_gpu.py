"""GPU: the native communicator (RCCL bound by libsemcode_hip at run time, include/semcode_hip.h "communicator") on a world of
one -- the only world a 1-GPU box can host (RCCL wants one rank per device).  The N > 1 logic (shard ranges, exchange order,
merge tie rule, sharded IVF) is covered on CPU over gloo (tests/test_sharded_gloo.py) and by the two-logical-shard GPU tests;
no N > 1 hardware run of this path exists yet."""
import numpy as np
import pytest
import torch

from oracle import sc_oracle as orc
from semcode_amd import _native
from semcode_amd.storage.sharded import ShardedSearcher, make_comm

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_world_of_one_collectives_and_sharded_calls(rt):
    uid = _native.comm_unique_id()
    assert len(uid) == _native.COMM_ID_BYTES
    comm = make_comm(rt, rank=0, world=1, unique_id=uid)
    try:
        assert comm.allreduce_max(3.25) == 3.25
        X = orc.synth(20_000, 96, seed=1)
        ix = _native.Index(rt, 96, metric="L2", kind="IVF_FLAT", nlist=16, row_base=1000)
        try:
            ix.add(X)
            s = ShardedSearcher(ix, "L2", comm=comm)
            for nq in (3, 40):  # exact path / batched path
                Q = orc.synth(nq, 96, seed=2)
                d, r = s.search(Q, k=10)
                od, orow = orc.search(X, Q, 10, "L2", row_base=1000)
                assert np.array_equal(r, orow) and np.array_equal(bits(d), bits(od))
            # device-resident variant: this rank's result lands in its slot of the gathered arrays
            Q = orc.synth(5, 96, seed=3)
            q = torch.from_numpy(Q).cuda()
            all_d = torch.zeros((1, 5, 10), dtype=torch.float32, device="cuda")
            all_r = torch.zeros((1, 5, 10), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            s.search_dev(q.data_ptr(), 5, 10, all_d.data_ptr(), all_r.data_ptr())
            rt.synchronize()
            od, orow = orc.search(X, Q, 10, "L2", row_base=1000)
            assert np.array_equal(all_r[0].cpu().numpy(), orow) and np.array_equal(bits(all_d[0].cpu().numpy()), bits(od))
            # broadcast of a device buffer from the only rank leaves it as it is; sharded training on one rank = training
            buf = torch.arange(1000, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            comm.broadcast(buf.data_ptr(), buf.numel() * 4, root=0)
            rt.synchronize()
            assert torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.float32))
            s.train(niter=3)
            assert ix.ivf_info()["nlist"] == 16
        finally:
            ix.close()
    finally:
        comm.close()
    with pytest.raises(_native.ScError):
        _native.Comm(rt, 2, 2, uid[:64] + uid[:64])  # rank outside the world
