"""world_size-2 (and 4) gloo test of ronkathon_amd.dist.FourStepNTT on CPU.

The product local engine is HIP-only (HipEngine); here a CHECKER engine built on the oracle is
injected so that the exchange logic -- block layout of the send/recv buffers, all_to_all_single,
input scatter and output placement maps -- runs under torch.distributed exactly as on GPUs."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GP, GG = 0xFFFFFFFF00000001, 7


class OracleEngine:
    """phase semantics of csrc/plan.h build_dist_phase1/2 restated with the oracle (test only)"""

    def __init__(self, log2n, inverse, rank, world):
        import oracle as orc
        from ronkathon_amd.dist import shape
        self.orc, self.inv, self.rank, self.world, self.log2n = orc, inverse, rank, world, log2n
        self.R, self.C, self.Rw, self.Cw = shape(log2n, world)
        self.per = (1 << log2n) // world

    def _view(self, ptr):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(self.per,))

    def _ntt(self, v):
        return self.orc.ifft(GP, GG, v) if self.inv else self.orc.fft(GP, GG, v)

    def _unscale(self, v, n):
        # the oracle's ifft scales by n^-1; the distributed plan applies the single 1/N at the end
        return self.orc.vec_mul(GP, v, np.full(v.size, n % GP, dtype=np.uint64)) if self.inv else v

    def phase1(self, in_ptr, send_ptr, stream=0):
        orc, n = self.orc, 1 << self.log2n
        x = self._view(in_ptr).reshape(self.R, self.Cw)
        out = self._view(send_ptr).reshape(self.R, self.Cw)
        w = orc.primitive_root_of_unity(GP, GG, n)
        if self.inv:
            w = orc.inverse(GP, w)
        for cl in range(self.Cw):
            col = self._unscale(self._ntt(np.ascontiguousarray(x[:, cl])), self.R)
            c = self.rank * self.Cw + cl
            tw = np.array([orc.pow_(GP, w, (c * k1) % n) for k1 in range(self.R)], dtype=np.uint64)
            out[:, cl] = orc.vec_mul(GP, col, tw)

    chunks = 1

    def phase1_chunk(self, j, in_ptr, send_ptr, stream=0):
        """column chunk j: columns [j*Cwc, (j+1)*Cwc) -> the contiguous piece [R][Cwc] at send[j*R*Cwc ..)"""
        orc, n = self.orc, 1 << self.log2n
        cwc = self.Cw // self.chunks
        x = self._view(in_ptr).reshape(self.R, self.Cw)
        piece = self._view(send_ptr)[j * self.R * cwc:(j + 1) * self.R * cwc].reshape(self.R, cwc)
        w = orc.primitive_root_of_unity(GP, GG, n)
        if self.inv:
            w = orc.inverse(GP, w)
        for cc in range(cwc):
            cl = j * cwc + cc
            col = self._unscale(self._ntt(np.ascontiguousarray(x[:, cl])), self.R)
            c = self.rank * self.Cw + cl
            tw = np.array([orc.pow_(GP, w, (c * k1) % n) for k1 in range(self.R)], dtype=np.uint64)
            piece[:, cc] = orc.vec_mul(GP, col, tw)

    def phase2(self, recv_ptr, out_ptr, stream=0):
        n = 1 << self.log2n
        cwc = self.Cw // self.chunks
        # block (g, j) = [Rw][Cwc]; global column c = g*Cw + j*Cwc + cc
        r = self._view(recv_ptr).reshape(self.world, self.chunks, self.Rw, cwc).transpose(0, 2, 1, 3).reshape(self.world, self.Rw, self.Cw)
        out = self._view(out_ptr).reshape(self.C, self.Rw)
        for k1l in range(self.Rw):
            row = np.ascontiguousarray(r[:, k1l, :]).reshape(-1)          # c = g*Cw + cl
            y = self._unscale(self._ntt(row), self.C)
            if self.inv:
                y = self.orc.vec_mul(GP, y, np.full(y.size, self.orc.inverse(GP, n % GP), dtype=np.uint64))
            out[:, k1l] = y


def _worker(rank, world, port, log2n, inverse, q, chunks=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as orc
        from conftest import splitmix_field
        from ronkathon_amd import dist as rdist
        x = splitmix_field(0xD157 + log2n, 1 << log2n)
        fs = rdist.FourStepNTT(log2n, inverse=inverse, engine=OracleEngine(log2n, inverse, rank, world), chunks=chunks)
        loc = torch.from_numpy(rdist.scatter_input(x, rank, world).view(np.int64).copy())
        out = fs.transform(loc)
        got = np.zeros(1 << log2n, dtype=np.uint64)
        rdist.place_output(got, out.numpy().view(np.uint64), rank, world)
        tot = torch.from_numpy(got.view(np.int64).copy())
        dist.all_reduce(tot)                                   # disjoint blocks: the sum assembles X
        ref = orc.ifft(GP, GG, x) if inverse else orc.fft(GP, GG, x)
        ok = bool(np.array_equal(tot.numpy().view(np.uint64), ref))
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("log2n,world,inverse,chunks", [(10, 2, False, 1), (11, 2, True, 1), (12, 4, False, 1),
                                                        (12, 2, False, 2), (14, 2, True, 4)])
def test_fourstep_gloo(log2n, world, inverse, chunks):
    """chunks > 1: the column-chunked exchange (one all-to-all per chunk, receive layout [source rank][chunk])"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, log2n, inverse, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "rank process failed"
    assert q.get(timeout=5) is True


def test_scatter_place_roundtrip():
    from ronkathon_amd import dist as rdist
    n, world = 1 << 12, 4
    R, Cc, Rw, Cw = rdist.shape(12, world)
    x = np.arange(n, dtype=np.uint64)
    for g in range(world):
        loc = rdist.scatter_input(x, g, world).reshape(R, Cw)
        assert loc[5, 3] == 5 * Cc + g * Cw + 3
    out = np.zeros(n, dtype=np.uint64)
    for g in range(world):
        blk = np.arange(n // world, dtype=np.uint64) + np.uint64(g * 10**6)
        rdist.place_output(out, blk, g, world)
    k1, k2 = 2 * Rw + 7, 11
    assert out[k1 + R * k2] == 2 * 10**6 + k2 * Rw + 7
