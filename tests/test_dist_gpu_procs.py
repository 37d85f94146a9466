"""The sharded four-step NTT with the REAL HIP engine in separate rank processes under torch.distributed.

This box has one GPU, so the ranks share it and the exchange goes through gloo with host staging
(ronkathon_amd/dist.py); everything else -- one process per rank, rank-specific phase plans, the send/recv block
layout, input scatter and output placement -- is what runs on an 8-GPU node with RCCL.  Result vs the oracle."""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

from test_dist_gloo import _free_port

GP, GG = 0xFFFFFFFF00000001, 7


def _worker(rank, world, port, log2n, inverse, q, chunks=1):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as orc
        from conftest import splitmix_field
        from ronkathon_amd import dist as rdist
        torch.cuda.set_device(0)
        x = splitmix_field(0xD15C + log2n, 1 << log2n)
        fs = rdist.FourStepNTT(log2n, inverse=inverse, chunks=chunks)  # default engine: HipEngine
        loc = torch.from_numpy(rdist.scatter_input(x, rank, world).view(np.int64).copy()).cuda()
        out = fs.transform(loc)
        torch.cuda.synchronize()
        got = np.zeros(1 << log2n, dtype=np.uint64)
        rdist.place_output(got, out.cpu().numpy().view(np.uint64), rank, world)
        tot = torch.from_numpy(got.view(np.int64).copy())
        dist.all_reduce(tot)                                           # disjoint blocks: the sum assembles X
        ref = orc.ifft(GP, GG, x) if inverse else orc.fft(GP, GG, x)
        if rank == 0:
            q.put(bool(np.array_equal(tot.numpy().view(np.uint64), ref)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("log2n,world,inverse,chunks", [(16, 2, False, 1), (18, 4, True, 1), (20, 8, False, 1), (20, 4, False, 4)])
def test_fourstep_hip_engine_rank_processes(log2n, world, inverse, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, log2n, inverse, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank process failed"
    assert q.get(timeout=5) is True
