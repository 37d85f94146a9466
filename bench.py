#!/usr/bin/env python3
"""bench.py -- headline benchmark: forward NTTs/s at degree 2^22 over the 64-bit Goldilocks prime.

python bench.py --gpus N --steps K --warmup W     (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one forward 2^22-point NTT (natural order in/out, device resident, synthetic uniform
coefficients).  With N ranks every rank transforms its own polynomials (the path shards by
polynomial, no data-path collective): weak scaling, value = N*K / max-over-ranks time.
Prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
Other workloads (--workload): batch16 (1024 x 2^16), mul22 (polynomial multiply, NTT size 2^22),
roundtrip16 (fwd+inv 2^16), fourstep (sharded four-step NTT with an RCCL all-to-all, N >= 1),
open22 (kzg::open's quotient: 2^22 coefficients / (x - z)), eval22 (Polynomial::evaluate, 2^22 coefficients),
rs16 (batched Reed-Solomon encode: 1024 messages of 2^15 symbols -> 2^16-point codewords),
vecmul24 / vecadd24 (element-wise Field Mul / Add over 2^24-element arrays: 24 bytes per element).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / cross-process device buffers fail with the legacy mode); the
# environment normally carries it already -- set before anything initialises the HIP runtime, inherited by self-launched ranks
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# workload -> (log2n, polynomials per step, algorithmic bytes per step as a multiple of n, unit, dominant kernel)
# Algorithmic bytes: an n-point NTT reads n and writes n 8-byte elements = 16*n.
WORKLOADS = {
    "ntt22":       (22, 1,    16.0,          "NTT/s", None),   # the headline: one forward transform per step
    "batch16":     (16, 1024, 16.0 * 1024,   "NTT/s", None),   # config 4: 1024 x 2^16 in one launch pair
    "mul22":       (22, 1,    48.0,          "op/s",  None),   # config 3: 3 transforms of size 2^22, pad/pointwise/truncate fused
    "roundtrip16": (16, 1,    32.0,          "op/s",  None),   # config 2: forward + inverse
    "open22":      (22, 1,    16.0,          "op/s",  "lindiv_one_kernel (csrc/lindiv_kernels.h: one launch up to 2^23 coefficients)"),
    "eval22":      (22, 1,    8.0,           "op/s",  "eval_onepass_kernel (csrc/scan_kernels.h)"),
    "rs16":        (16, 1024, 12.0 * 1024,   "op/s",  None),   # reads n/2, writes n coefficients per codeword
    "vecmul24":    (24, 1,    24.0,          "op/s",  "vec_binary2_kernel<GlOps, VEC_MUL>"),
    "vecadd24":    (24, 1,    24.0,          "op/s",  "vec_binary2_kernel<GlOps, VEC_ADD>"),
}


MODULUS = 0xFFFFFFFF00000001   # the field of the run: Goldilocks unless --prime says otherwise (main)
GENERATOR = 7


def synth(n, seed):
    """i.i.d. uniform on [0, p), as int64-viewable uint64: rejection sampling from 64 random bits (primes
    near 2^64), from the bits the prime has (small primes)"""
    rng = np.random.default_rng(seed)
    P = np.uint64(MODULUS)
    top = 1 << MODULUS.bit_length()
    x = rng.integers(0, top, size=n, dtype=np.uint64)
    bad = x >= P
    while bad.any():
        x[bad] = rng.integers(0, top, size=int(bad.sum()), dtype=np.uint64)
        bad = x >= P
    return x


KERNEL_SOURCES = ("ronkathon_amd/csrc/ntt_tile.h", "ronkathon_amd/csrc/gl64.h", "ronkathon_amd/csrc/plan.h",
                  "ronkathon_amd/csrc/tile_kernels.hip", "ronkathon_amd/csrc/tile_kernels_cfg.hip",
                  "ronkathon_amd/csrc/tile_kernel_def.h", "ronkathon_amd/csrc/tile_cfg_table.h",
                  "ronkathon_amd/csrc/tile_kernels_half.hip", "ronkathon_amd/csrc/tile_kernels_feat.hip",
                  "ronkathon_amd/csrc/field_policy.h", "ronkathon_amd/csrc/mont64.h", "ronkathon_amd/csrc/tile_kernels_mont.hip",
                  "ronkathon_amd/csrc/tile_kernels_mont_feat.hip", "ronkathon_amd/csrc/ntt_tile_wl.h",
                  "ronkathon_amd/csrc/tile_kernels_wl.hip")
VALU_PEAK_LANE_OPS = 52.5e12   # full-rate 32-bit VALU lane-instructions/s measured on MI355X (profiles/r01_instr_rate_gfx950.txt)


# the scan workloads run kernels of scan_kernels.h / lindiv_kernels.h only: their counters stay valid while THOSE files
# are unchanged
SCAN_SOURCES = ("ronkathon_amd/csrc/scan_kernels.h",)
SOURCES_BY_WORKLOAD = {"open22": ("ronkathon_amd/csrc/lindiv_kernels.h", "ronkathon_amd/csrc/gl64.h"), "eval22": SCAN_SOURCES}


def kernel_source_hash(files=KERNEL_SOURCES):
    """sha256 over the kernel sources a measurement depends on (default: the tile-kernel sources): PMC / census files
    record it, so stale counters are detected"""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_if_current(path, workload=None):
    """a committed measurement file (profiles/latest_*.json) -- only if it was taken on the current kernel sources: the
    tile-kernel sources (`kernel_source_hash`), or, for a workload with its own source list, those (`workload_source_hash`)"""
    try:
        with open(os.path.join(ROOT, path)) as f:
            d = json.load(f)
    except Exception:  # noqa: BLE001
        return None, "missing"
    if d.get("kernel_source_hash") == kernel_source_hash():
        return d, None
    own = SOURCES_BY_WORKLOAD.get(workload)
    if own and d.get("workload_source_hash") == kernel_source_hash(own):
        return d, None
    return None, "stale (kernel sources changed since %s was taken)" % path


def cpu_baseline_mul(log2n, budget_s=10.0):
    """(a) the fair comparator: NTT - pointwise - iNTT with the reference's recursive fft/ifft restated
    in C, one thread; (b) the reference's own schoolbook Mul (polynomial/arithmetic.rs:97-119) timed at small D and
    EXTRAPOLATED ~ D^2 to the benchmark size (flagged as an extrapolation)."""
    import oracle as orc
    P, G = MODULUS, GENERATOR
    n = 1 << log2n
    a = np.concatenate([synth(n // 2, 7), np.zeros(n // 2, dtype=np.uint64)])
    b = np.concatenate([synth(n // 2, 8), np.zeros(n // 2, dtype=np.uint64)])
    t0 = time.perf_counter()
    fa, fb = orc.fft(P, G, a), orc.fft(P, G, b)
    prod = orc.ifft(P, G, orc.vec_mul(P, fa, fb))
    t_ntt = time.perf_counter() - t0
    del prod
    sb = []
    for d in (1 << 11, 1 << 12, 1 << 13):
        u, v = synth(d, 70 + d), synth(d, 71 + d)
        t0 = time.perf_counter()
        orc.poly_mul(P, u, v)
        sb.append((d, time.perf_counter() - t0))
        if sum(t for _, t in sb) > budget_s:
            break
    d_last, t_last = sb[-1]
    extrap = t_last * ((n // 2) / d_last) ** 2
    return {"value": 1.0 / t_ntt, "unit": "op/s", "cores": 1, "kind": "port",
            "sample": "1 product of two 2^%d-coefficient polynomials by NTT size 2^%d (2 fft + pointwise + ifft, the "
                      "reference's recursive algorithm restated in C), %.2f s, 1 thread" % (log2n - 1, log2n, t_ntt),
            "schoolbook_reference_algorithm": {"measured": [{"D": d, "seconds": t} for d, t in sb],
                                               "extrapolated_seconds_at_benchmark_size": extrap,
                                               "note": "EXTRAPOLATION ~ D^2 from D = %d (polynomial/arithmetic.rs:97-119 is "
                                                       "O(D*D2) and never uses the FFT)" % d_last},
            "host_cores_available": os.cpu_count()}


def cpu_baseline_roundtrip(log2n, budget_s=8.0):
    """forward + inverse (polynomial/mod.rs:295-323, :430-453: same recursion + n^-1 scaling), one thread"""
    import oracle as orc
    P, G = MODULUS, GENERATOR
    x = synth(1 << log2n, 98)
    reps, t = 0, 0.0
    while t < budget_s and reps < 64:
        t0 = time.perf_counter()
        y = orc.fft(P, G, x)
        z = orc.ifft(P, G, y)
        t += time.perf_counter() - t0
        reps += 1
    assert np.array_equal(z, x)
    return {"value": reps / t, "unit": "op/s", "cores": 1, "kind": "port",
            "sample": "%d forward+inverse 2^%d round trips (root lookup included), recursive algorithm of the reference "
                      "restated in C, 1 thread" % (reps, log2n), "host_cores_available": os.cpu_count()}


def cpu_baseline(log2n, budget_s=12.0, p=None, g=None):
    """oracle (port of the reference's recursive fft, polynomial/mod.rs:295-323) on one host core"""
    import oracle as orc
    p, g = p or orc.GOLDILOCKS_P, g or orc.GOLDILOCKS_G
    n = 1 << log2n
    x = synth(n, 99)
    w = orc.primitive_root_of_unity(p, g, n)  # root excluded from the timed region
    reps, t = 0, 0.0
    while True:
        v = x.copy()
        t0 = time.perf_counter()
        orc.fft_recursive_inplace(p, v, w)
        t += time.perf_counter() - t0
        reps += 1
        if t > budget_s or reps >= 8:
            break
    return {"value": reps / t, "unit": "NTT/s", "cores": 1, "kind": "port",
            "sample": "%d forward 2^%d NTTs, recursive even/odd algorithm of the reference restated in C, 1 thread" % (reps, log2n),
            "host_cores_available": os.cpu_count()}


def cpu_baseline_batched(log2n, budget_s=10.0):
    """the batched shape on ALL host cores, one polynomial per thread (the oracle's C code
    runs outside the GIL); the reference itself is single-threaded, so `cores` says what was used"""
    import concurrent.futures as cf
    import oracle as orc
    n = 1 << log2n
    cores = os.cpu_count() or 1
    w = orc.primitive_root_of_unity(orc.GOLDILOCKS_P, orc.GOLDILOCKS_G, n)
    polys = [synth(n, 1000 + i) for i in range(cores)]

    def work(v):
        t_end = time.perf_counter() + budget_s
        done = 0
        while time.perf_counter() < t_end:
            orc.fft_recursive_inplace(orc.GOLDILOCKS_P, v, w)   # in place: the input of the next repetition
            done += 1
        return done

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        total = sum(ex.map(work, polys))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "NTT/s", "cores": cores, "kind": "port",
            "sample": "%d forward 2^%d NTTs in %.1f s, one polynomial per thread on %d threads, recursive algorithm of the "
                      "reference restated in C" % (total, log2n, dt, cores)}


def self_launch_command(gpus, environ, argv):
    """The command line `python bench.py --gpus N ...` re-executes itself under when no launcher set WORLD_SIZE: one rank per
    GPU through torch.distributed.run on 127.0.0.1 (the contract's own form).  None when nothing is to be launched: N = 1,
    or the ranks exist already.  The in-library `sharded` workload drives every GPU from ONE process and never re-launches."""
    if gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    if "--workload" in argv and argv[argv.index("--workload") + 1: argv.index("--workload") + 2] == ["sharded"]:
        return None
    if any(a == "--workload=sharded" for a in argv):
        return None
    port = environ.get("MASTER_PORT")
    if not port:            # a free port, asked of the kernel (a fixed formula can collide with another job or a stale rendezvous)
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(argv[0])] + list(argv[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks = GPUs of the job (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="ntt22")
    ap.add_argument("--log2n", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="polynomials per launch (default: 1, batch16: 1024)")
    ap.add_argument("--streams", type=int, default=2, help="independent transforms in flight (HIP streams)")
    ap.add_argument("--tile-logc", type=int, default=-2, dest="tile_logc",
                    help="log2 columns per tile of the timed plans (-2 = bench default: 2 when several streams, else library default)")
    ap.add_argument("--twf", type=int, default=-1, help="largest full inter-pass twiddle matrix, log2 entries (-1 = library default)")
    ap.add_argument("--ranks", type=int, default=0, help="sharded: logical ranks (default: one per visible GPU)")
    ap.add_argument("--chunks", type=int, default=0, help="sharded: column chunks of the exchange (0 = default, up to 4)")
    ap.add_argument("--samples", type=int, default=5, help="timed regions of --steps steps each (median reported)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed plans")
    ap.add_argument("--mode", default="auto", choices=["auto", "many", "batch", "streams"],
                    help="ntt22: how the K transforms of a region reach the library -- many: ONE plan handle (in_flight = 2), "
                         "ronk_ntt_forward_many_dev with --group arrays per call; batch: ONE plan of batch --group, K/group calls; "
                         "streams: one plan per caller stream (--streams), the round-1/2 protocol; auto = many")
    ap.add_argument("--group", type=int, default=0, help="many / batch: polynomials per library call (default 32, at most K)")
    ap.add_argument("--rotate", type=int, default=-1,
                    help="distinct input AND output buffers the steps cycle through (HBM-cold protocol; default 8 for ntt22 "
                         "= 512 MiB touched between two uses of a buffer, else 1 = the same buffers every step)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "mesh", "host"],
                    help="fourstep / sharded: how the transpose travels -- rccl: torch.distributed all_to_all_single on the nccl "
                         "backend (fourstep, one process per GPU) or the library's dlopen'ed ncclGroup Send/Recv (sharded, one "
                         "process); mesh: the library's hipMemcpyPeerAsync mesh (sharded); host: staged through host memory "
                         "(gloo smoke runs); auto = rccl for fourstep, mesh for sharded")
    ap.add_argument("--prime", type=lambda v: int(v, 0), default=0,
                    help="ntt22 / batch16 / roundtrip16 / mul22: another odd 64-bit prime than Goldilocks (the tile kernels over "
                         "Montgomery arithmetic, csrc/field_policy.h), e.g. 0xFFFFFFFC00000001")
    ap.add_argument("--generator", type=lambda v: int(v, 0), default=0,
                    help="primitive element for --prime (default: the smallest quadratic non-residue)")
    args = ap.parse_args()
    if args.gpus <= 0:      # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world is the job
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the contract's
    # torch.distributed.run command line), then fall through as rank RANK of WORLD_SIZE.  Under a launcher the two must agree.
    cmd = self_launch_command(args.gpus, os.environ, sys.argv)
    if cmd:
        os.execv(cmd[0], cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and args.workload != "sharded":
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus (or run plain "
                 "`python bench.py --gpus N`, which starts the ranks itself)" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # One rank per GPU.  RONK_BENCH_BACKEND=gloo + fewer GPUs than ranks is a control-flow smoke test only (ranks then
    # share devices); the driver's runs use the default: nccl (= RCCL) with local_rank < device_count.
    backend = os.environ.get("RONK_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import ronkathon_amd as R
    from ronkathon_amd import _lib as L
    P, G = R.GOLDILOCKS_P, R.GOLDILOCKS_G
    wl = args.workload
    mont = False
    if args.prime and (args.prime, args.generator or 7) != (P, 7):
        if wl not in ("ntt22", "batch16", "roundtrip16", "mul22"):
            sys.exit("bench.py: --prime applies to the transform workloads (ntt22, batch16, roundtrip16, mul22)")
        global MODULUS, GENERATOR
        P = MODULUS = args.prime
        G = args.generator
        if not G:      # any quadratic non-residue gives omega_n = g^((p-1)/n) its exact order for every power of two n
            G = next(c for c in range(2, 1000) if pow(c, (P - 1) // 2, P) == P - 1)
        GENERATOR = G
        mont = True

    if wl == "fourstep":
        from ronkathon_amd import dist as rdist
        # the exchange: RCCL all_to_all_single on the nccl backend (one process per GPU, the product path) or, on a backend
        # without device collectives (gloo smoke runs on fewer GPUs than ranks), staged through host memory
        want = args.exchange if args.exchange != "auto" else ("rccl" if backend == "nccl" else "host")
        if (want == "rccl") != (backend == "nccl") or want == "mesh":
            sys.exit("bench.py --workload fourstep: --exchange %s needs %s" % (
                want, {"rccl": "the nccl backend (unset RONK_BENCH_BACKEND)", "host": "RONK_BENCH_BACKEND=gloo",
                       "mesh": "--workload sharded (the one-process in-library form)"}[want]))
        # fail fast, BEFORE anything is timed: every rank the launcher promised took part in this collective
        one0_ = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(one0_)
        assert int(one0_.item()) == args.gpus, "ranks_seen %d != --gpus %d (before timing)" % (int(one0_.item()), args.gpus)
        res = rdist.bench_fourstep(args.log2n or 26, args.steps, args.warmup, chunks=args.chunks or None)
        res["config"]["exchange"] = want
        # what the all-to-all rides on: can this rank's GPU reach the others peer-to-peer (xGMI)?  RCCL falls back to host
        # staging where it cannot; reported so that a slow exchange is never mistaken for a result
        nd_ = torch.cuda.device_count()
        me_ = torch.cuda.current_device()
        res["config"]["peer_access"] = {"device": me_, "visible_devices": nd_,
                                        "can_access": [bool(torch.cuda.can_device_access_peer(me_, d_)) for d_ in range(nd_) if d_ != me_]}
        one_ = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(one_)
        res["ranks_seen"] = int(one_.item())
        assert res["ranks_seen"] == args.gpus, "ranks_seen %d != --gpus %d" % (res["ranks_seen"], args.gpus)
        if rank == 0:
            print(json.dumps(res))
        if world > 1:
            dist.destroy_process_group()
        return

    if wl == "msm20":
        # row N4: kzg::commit as a bucket-method MSM over BN254 G1 (csrc/msm_kernels.h); points = multiples of G built by the
        # oracle (outside the timed region), random 254-bit scalars; every step is one whole MSM incl. its host tail
        from oracle import bn254 as ob
        lg = args.log2n or 20
        nn = 1 << lg
        m = min(nn, 1 << 11)
        mult = ob.multiples(m)
        m64 = (1 << 64) - 1
        pw1 = np.array([[(pt[0] >> (64 * j)) & m64 for j in range(4)] + [(pt[1] >> (64 * j)) & m64 for j in range(4)]
                        for pt in mult], dtype=np.uint64)
        pw = np.tile(pw1, (nn // m, 1))
        rng = np.random.default_rng(0x5EED4000 + rank)
        sw = rng.integers(0, 2**63, size=(nn, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(nn, 4), dtype=np.uint64)
        sw[:, 3] &= np.uint64((1 << 61) - 1)                                   # < 2^253 < r
        dp = torch.from_numpy(pw.view(np.int64)).cuda(); ds = torch.from_numpy(sw.view(np.int64)).cuda()
        out = np.zeros(8, dtype=np.uint64)

        def one():
            L.check(L.lib.ronk_msm_bn254_dev(dp.data_ptr(), ds.data_ptr(), nn, L.ptr(out), 0))
        verified = None
        if not args.no_verify:
            one()
            a = np.tile(np.arange(1, m + 1, dtype=object), nn // m)
            ks = sw[:, 0].astype(object) + (sw[:, 1].astype(object) << 64) + (sw[:, 2].astype(object) << 128) + (sw[:, 3].astype(object) << 192)
            want = ob.mul(int((ks * a).sum() % ob.R), ob.G)
            got = (sum(int(out[i]) << (64 * i) for i in range(4)), sum(int(out[4 + i]) << (64 * i) for i in range(4)))
            verified = bool(got == want)
            assert verified, "MSM differs from the oracle"
        steps = min(args.steps, 20)
        for _ in range(min(args.warmup, 3)):
            one()
        dts = []
        for _ in range(max(1, args.samples)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize()
            dts.append((time.perf_counter() - t0) / steps)
        dt = float(np.median(dts))
        cpu = None
        if not args.no_cpu:
            # the oracle's fold (affine add, Python integers) on a bounded sample
            k = 256
            kk = [int(sw[i, 0]) | (int(sw[i, 1]) << 64) | (int(sw[i, 2]) << 128) | (int(sw[i, 3]) << 192) for i in range(k)]
            t0 = time.perf_counter()
            ob.msm(mult[:k], kk)
            tc = time.perf_counter() - t0
            cpu = {"value": k / tc, "unit": "points/s", "cores": 1, "kind": "port",
                   "sample": "the first %d points and (253-bit) scalars of the same workload through oracle/bn254.py (the "
                             "reference's fold of AffinePoint Mul / Add, on Python integers), %.2f s" % (k, tc)}
        res = {"metric": "MSM points/s, BN254 G1, 2^%d points (kzg::commit)" % lg, "value": nn / dt, "unit": "points/s",
               "n_gpus": 1, "steps": steps, "warmup": min(args.warmup, 3), "ms_per_step": dt * 1e3,
               "min_ms_per_step": min(dts) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u32 limbs (254-bit Montgomery)", "data": "synthetic", "verified": verified,
               "config": {"workload": "bucket-method MSM over BN254 G1, 2^%d points (multiples of G), random 253-bit scalars, "
                                      "device-resident inputs, host tail included" % lg},
               "roofline": {"bound": "hbm", "achieved": nn * 96.0 / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": nn * 96.0 / dt / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "note": "algorithmic bytes = 96 per point (64-byte point + 32-byte scalar); the work is VALU-bound "
                                    "254-bit modular arithmetic, not memory-bound: see valu",
                            "valu": {"bucket_additions": int(nn) * 18, "valu_per_addition": 4080,
                                     "issue_bound_ms": nn * 18 * 4080 / 64.0 / 1024.0 * 1.95e-6,
                                     "frac_of_valu_peak": (nn * 18 * 4080 / 64.0 / 1024.0 * 1.95e-6) / (dt * 1e3),
                                     "note": "mixed additions of the accumulate kernel alone (n x ~18 windows at 2^20 points): "
                                             "SQ_INSTS_VALU = 1.204e9 wave-instructions per launch = 4 080 lane-instructions per "
                                             "addition incl. divergence (profiles/r02_rocprof_msm20_pmc.txt; static count of the "
                                             "8M + 2S path: ~3 500), at the measured 1.95 ns per wave-instruction per SIMD "
                                             "(profiles/r02_clock_probe.txt); the sort, the bucket reduction and the host tail are on top"}}}
        if cpu:
            res["cpu_baseline"] = cpu
        print(json.dumps(res))
        return

    if wl == "e2e22":
        # The drop-in path's own rate: HOST buffers in, HOST buffers out (what `Polynomial<_, Goldilocks, D>` with its inline
        # `[F; D]` gets through the Rust shim), PCIe both ways inside the timed region.  (a) one polynomial per call through the
        # one-shot `ronk_fft` (cached plan); (b) `--group` polynomials per call through a batched plan, which the library
        # pipelines over its three internal streams (upload i+1 | transform i | download i-1).  Bound: the link, not HBM.
        PCIE_GBS = 63.0                                  # PCIe Gen5 x16 spec, one direction (MI355X_MICROARCH.md)
        lg = args.log2n or 22
        nn = 1 << lg
        grp = args.group or 8
        xh = synth(nn * grp, 0x5EED5000 + rank)
        yh = np.empty_like(xh)
        x1, y1 = xh[:nn].copy(), np.empty(nn, dtype=np.uint64)

        def single(count):
            for _ in range(count):
                L.check(L.lib.ronk_fft(P, G, L.ptr(x1), L.ptr(y1), None, nn))
        bplan = L.Plan(P, G, lg, grp, local_rank)

        def batched(calls):
            for _ in range(calls):
                L.check(L.lib.ronk_ntt_forward(bplan.h, L.ptr(xh), L.ptr(yh), None))
        verified = None
        if not args.no_verify:
            import oracle as orc
            single(1); batched(1)
            want = orc.fft(P, G, x1)
            verified = bool(np.array_equal(y1, want) and np.array_equal(yh[:nn], want)
                            and np.array_equal(yh[(grp - 1) * nn:], orc.fft(P, G, xh[(grp - 1) * nn:])))
            assert verified, "host-pointer transform differs from the oracle"
        steps = max(grp, args.steps - args.steps % grp)
        single(min(args.warmup, 5)); batched(2)

        def med(f, arg, units):
            ts = []
            for _ in range(max(1, args.samples)):
                t0 = time.perf_counter(); f(arg); ts.append(time.perf_counter() - t0)
            return units / float(np.median(ts)), float(np.median(ts)) / units * 1e3
        v_b, ms_b = med(batched, steps // grp, steps)
        v_s, ms_s = med(single, min(steps, 64), min(steps, 64))
        link = 16.0 * nn                                  # bytes over the link per transform: 8n in + 8n out
        res = {"metric": "forward NTTs/s end to end (host buffers in and out, PCIe inside the timed region), degree 2^%d" % lg,
               "value": v_b * world, "unit": "NTT/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": ms_b,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "verified": verified,
               "config": {"workload": "forward NTT n = 2^%d through the HOST-pointer entry points: value = ronk_ntt_forward on a "
                                      "plan of batch %d (pipelined over the batch inside the library); single = one polynomial per "
                                      "ronk_fft call; pageable numpy buffers" % (lg, grp), "log2n": lg, "group": grp},
               "single": {"value": v_s, "ms_per_step": ms_s, "entry_point": "ronk_fft (one-shot, cached plan)",
                          "frac_of_serial_link_floor": (link / (PCIE_GBS * 1e9)) / (ms_s / 1e3)},
               "roofline": {"bound": "pcie", "achieved": link / (ms_b / 1e3) / 1e9, "peak": 2 * PCIE_GBS, "unit": "GB/s",
                            "frac": link / (ms_b / 1e3) / 1e9 / (2 * PCIE_GBS), "traffic": link,
                            "note": "bytes over the link per transform = 16*n (8n up, 8n down); peak = 63 GB/s each way, full "
                                    "duplex (the pipelined batch keeps both directions busy); one polynomial per call uses the "
                                    "link one way at a time: its floor is 16n / 63 GB/s (single.frac_of_serial_link_floor)"}}
        if rank == 0:
            print(json.dumps(res))
        bplan.close()
        return

    if wl == "sharded":
        # the in-library sharded transform (ronk_sharded_*): ONE process drives every visible GPU (or --ranks logical ranks
        # on the GPUs there are), peer-copy exchange in column chunks; device-resident blocks, K transforms pipelined
        import ctypes as C
        ndev = torch.cuda.device_count()
        W = args.ranks or ndev
        lg = args.log2n or 26
        ex = {"auto": L.EXCHANGE_MESH, "mesh": L.EXCHANGE_MESH, "rccl": L.EXCHANGE_RCCL}.get(args.exchange)
        if ex is None:
            sys.exit("bench.py --workload sharded: --exchange mesh or rccl")
        sp = L.ShardedPlan(lg, [g % ndev for g in range(W)], chunks=args.chunks, exchange=ex)
        per = sp.per_rank
        din, dout = [], []
        for g in range(W):
            torch.cuda.set_device(g % ndev)
            a_ = torch.from_numpy(synth(per, 0x5EED3000 + g).view(np.int64)).cuda()
            din.append(a_); dout.append(torch.empty_like(a_))
        ip, op = [t.data_ptr() for t in din], [t.data_ptr() for t in dout]
        for _ in range(args.warmup):
            sp.transform_dev(ip, op)
        sp.sync()
        dts = []
        for _ in range(max(1, args.samples)):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sp.transform_dev(ip, op)
            sp.sync()
            dts.append(time.perf_counter() - t0)
        dt = float(np.median(dts))
        nn = 1 << lg
        # where the time of ONE transform goes when its stages do not overlap (ronk_sharded_time_stages): phase 1, the whole
        # exchange, phase 2, and the achieved rate per directed link -- so that a first run on a real node can be read without a
        # second lease (a link rate far below xGMI's, or staged pairs above, says the exchange is not travelling peer-to-peer)
        stage_runs = [sp.time_stages(ip, op) for _ in range(3)]
        stages = min(stage_runs, key=lambda d_: d_["phase1_ms"] + d_["exchange_ms"] + d_["phase2_ms"])
        stages["overlap_gain_ms"] = stages["phase1_ms"] + stages["exchange_ms"] + stages["phase2_ms"] - dt / args.steps * 1e3
        print(json.dumps({"metric": "sharded four-step forward NTTs/s (in-library, single process), degree 2^%d" % lg,
                          "value": args.steps / dt, "unit": "NTT/s", "n_gpus": ndev, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "min_ms_per_step": min(dts) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                          "config": {"workload": "four-step NTT n = 2^%d over %d rank(s) on %d GPU(s), %s, "
                                                 "%d column chunk(s)" % (lg, W, ndev, "ncclGroup Send/Recv (dlopen'ed RCCL)" if ex == L.EXCHANGE_RCCL
                                                                         else "hipMemcpyPeerAsync mesh", sp.chunks),
                                     "ranks": W, "chunks": sp.chunks, "exchange": "rccl" if ex == L.EXCHANGE_RCCL else "mesh",
                                     # how blocks travel between ranks (ronk_sharded_plan_peer_access): a staged pair means the
                                     # runtime refused peer access and copies go through host memory -- the number below would
                                     # then measure THAT, not xGMI
                                     "peer_access": dict(zip(("matrix", "staged_pairs"), sp.peer_access())),
                                     "stages_serialised": stages,
                                     "ranks_per_gpu": W / float(ndev)},
                          "roofline": {"bound": "hbm", "achieved": 16.0 * nn / ndev / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": 16.0 * nn / ndev / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None}}))
        sp.close()
        return

    wl_log2n, wl_batch, wl_bytes_per_n, wl_unit, wl_kernel = WORKLOADS[wl]
    log2n = args.log2n or wl_log2n
    batch = args.batch or wl_batch
    n = 1 << log2n
    if wl in ("batch16", "rs16"):
        args.streams = 1
    # S independent transforms in flight: each stream has its own plan (scratch), input and output, so
    # the load/store phases of one transform overlap the VALU-bound butterflies of another.
    mode = args.mode if wl == "ntt22" else "streams"
    if mode == "auto":
        mode = "many"
    group = 1
    if mode in ("many", "batch"):
        group = max(1, min(args.group or 32, args.steps))
        if mode == "batch":
            while args.steps % group:          # a region is exactly K transforms = K/group calls
                group -= 1
        args.streams = 1
    S = max(1, args.streams) if wl in ("ntt22", "batch16") else 1
    R_cold = args.rotate if args.rotate >= 1 else (8 if wl == "ntt22" else 1)
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream() for _ in range(S - 1)]
    pb = batch * (group if mode == "batch" else 1)          # polynomials per buffer
    xs = [torch.from_numpy(synth(n * pb, 0x5EED0000 + rank * 16 + i).view(np.int64)).cuda() for i in range(S)]
    ys = [torch.empty_like(xs[0]) for _ in range(S)]
    # rotation pool (per stream): buffer r > 0 is buffer 0 rolled by r*977 elements -- distinct canonical data, and every
    # step reads and writes memory that was last touched R steps ago
    xpool = [[xs[k]] + [torch.roll(xs[k], 977 * r) for r in range(1, R_cold)] for k in range(S)]
    ypool = [[ys[k]] + [torch.empty_like(ys[k]) for _ in range(1, R_cold)] for k in range(S)]
    # several transforms in flight -> narrower tiles (two workgroups per CU); one at a time -> default plan
    tile_lc = 2 if (S > 1 and log2n >= 20) else -1
    if args.tile_logc >= -1 and args.tile_logc != -2:
        tile_lc = args.tile_logc
    if mode == "many":      # ONE handle; the library keeps two transforms in flight (ronk_plan_opts::in_flight)
        plans = [L.Plan(P, G, log2n, batch, local_rank, tile_log2_columns=args.tile_logc if args.tile_logc >= -1 else -1,
                        twiddle_matrix_log2_max=args.twf, in_flight=2)]
    elif mode == "batch":   # ONE handle, `group` polynomials per call; in_flight left to the library (automatic = 1)
        plans = [L.Plan(P, G, log2n, pb, local_rank, tile_log2_columns=args.tile_logc if args.tile_logc >= -1 else -1,
                        twiddle_matrix_log2_max=args.twf)]
    else:
        plans = [L.Plan(P, G, log2n, batch, local_rank, tile_log2_columns=tile_lc, twiddle_matrix_log2_max=args.twf,
                        in_flight=1 if wl == "ntt22" else -1) for _ in range(S)]
    # latency plan: one transform at a time on one stream, library defaults
    lat_plan = (L.Plan(P, G, log2n, batch, local_rank, twiddle_matrix_log2_max=args.twf, in_flight=1)
                if (wl == "ntt22" and (tile_lc >= 0 or mode != "streams")) else plans[0])
    if wl in ("ntt22", "batch16", "roundtrip16"):
        assert lat_plan.path() == (2 if mont else 1), "the timed plan does not run the tile kernels (ronk_plan_path)"
    x, y, plan, stream = xs[0], ys[0], plans[0], main_stream.cuda_stream
    rot = {"R": R_cold, "lat": False}     # what run() cycles over: set before each measurement
    if wl == "mul22":
        b = torch.from_numpy(synth(n // 2, 0x5EED1000 + rank).view(np.int64)).cuda()
        a = x[: n // 2].contiguous()
        out = torch.empty(n - 1, dtype=torch.int64, device="cuda")
    if wl in ("vecmul24", "vecadd24"):
        x2 = torch.from_numpy(synth(n, 0x5EED2000 + rank).view(np.int64)).cuda()
    if wl in ("open22", "eval22"):
        zpt = 0x123456789ABCDEF1 % P                       # evaluation point z; divisor is x - z = [-z, 1]
        scal = torch.zeros(1, dtype=torch.int64, device="cuda")

    def step(i):
        if wl == "mul22":
            L.check(L.lib.ronk_poly_mul_dev(P, G, a.data_ptr(), n // 2, b.data_ptr(), n // 2, out.data_ptr(), stream))
        elif wl == "open22":
            L.check(L.lib.ronk_poly_div_linear_dev(P, x.data_ptr(), n, P - zpt, 1, y.data_ptr(), scal.data_ptr(), stream))
        elif wl == "eval22":
            L.check(L.lib.ronk_poly_eval_dev(P, x.data_ptr(), n, zpt, scal.data_ptr(), stream))
        elif wl == "vecmul24":
            L.check(L.lib.ronk_vec_mul_dev(P, x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, stream))
        elif wl == "vecadd24":
            L.check(L.lib.ronk_vec_add_dev(P, x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, stream))
        elif wl == "rs16":
            plan.rs_encode_batch_dev(x.data_ptr(), n // 2, y.data_ptr(), stream)   # x: [batch][n/2] compact messages
        elif wl == "roundtrip16":
            plan.forward_dev(x.data_ptr(), y.data_ptr(), stream)
            plan.inverse_dev(y.data_ptr(), y.data_ptr(), stream)
        else:
            k = i % S
            r = (i // S) % rot["R"]
            plans[k].forward_dev(xpool[k][r].data_ptr(), ypool[k][r].data_ptr(), streams[k].cuda_stream)

    def run(count):
        if wl == "ntt22" and rot["lat"]:               # one transform at a time, default plan, one stream
            for i in range(count):
                r = i % rot["R"]
                lat_plan.forward_dev(xpool[0][r].data_ptr(), ypool[0][r].data_ptr(), stream)
        elif wl == "ntt22" and mode == "many":         # `count` transforms as calls of `group` arrays
            i = 0
            while i < count:
                g_ = min(group, count - i)
                idx = [(i + j) % rot["R"] for j in range(g_)]
                plans[0].forward_many_dev([xpool[0][r].data_ptr() for r in idx], [ypool[0][r].data_ptr() for r in idx], stream)
                i += g_
        elif wl == "ntt22" and mode == "batch":        # count/group calls of `group` polynomials each
            for c_ in range(count // group):
                r = c_ % rot["R"]
                plans[0].forward_dev(xpool[0][r].data_ptr(), ypool[0][r].data_ptr(), stream)
        else:
            for i in range(count):
                step(i)

    # ---- verification (outside every timed region; the oracle is the CHECKER here, never the thing measured): every
    # plan that is timed below transforms one input and is compared with the oracle's restatement of
    # Polynomial::fft (reference src/polynomial/mod.rs:273-323); the other workloads are checked through the oracle too.
    def to_np(t):
        return t.cpu().numpy().view(np.uint64)

    def verify():
        import oracle as orc
        checked = []
        if wl in ("ntt22", "batch16"):
            xh = to_np(xs[0])
            rows = sorted({0, pb - 1, pb // 2})
            refs = {r: orc.fft(P, G, xh[r * n:(r + 1) * n]) for r in rows}
            seen = []
            for pl_ in plans + [lat_plan]:
                if any(pl_ is q for q in seen):
                    continue
                if pl_.batch != pb:        # the latency plan of the batch mode transforms one polynomial of the buffer
                    rows_, nb_ = [0], 1
                else:
                    rows_, nb_ = rows, pb
                seen.append(pl_)
                ys[0].zero_()
                pl_.forward_dev(xs[0].data_ptr(), ys[0].data_ptr(), stream)
                torch.cuda.synchronize()
                yh = to_np(ys[0])
                for r in rows_:
                    if not np.array_equal(yh[r * n:(r + 1) * n], refs[r]):
                        raise SystemExit("bench.py: plan output differs from the oracle (polynomial %d) -- refusing to time it" % r)
            checked.append("%d timed plan(s) x %d polynomial(s) == oracle.fft" % (len(seen), len(rows)))
            if wl == "ntt22" and mode == "many":
                # the timed entry point itself: three arrays in one call (both lanes, the second one twice), all three
                # outputs against the oracle
                kk = min(3, R_cold)
                for r in range(kk):
                    ypool[0][r].zero_()
                plans[0].forward_many_dev([xpool[0][r].data_ptr() for r in range(kk)], [ypool[0][r].data_ptr() for r in range(kk)], stream)
                torch.cuda.synchronize()
                for r in range(kk):
                    want = refs[0] if r == 0 else orc.fft(P, G, to_np(xpool[0][r]))
                    if not np.array_equal(to_np(ypool[0][r]), want):
                        raise SystemExit("bench.py: ronk_ntt_forward_many_dev array %d differs from the oracle" % r)
                checked.append("ronk_ntt_forward_many_dev: %d arrays in one call == oracle.fft" % kk)
        elif wl == "mul22":
            step(0); torch.cuda.synchronize()
            ah, bh, oh = to_np(a), to_np(b), to_np(out)
            # every one of the d + d2 - 1 coefficients against the oracle's product: ifft(fft(a) * fft(b)) of the zero-padded
            # operands, three oracle transforms (the schoolbook Mul of arithmetic.rs:97-119 is O(d^2): hours at this size)
            zpad = np.zeros(n - ah.size, dtype=np.uint64)
            want = orc.ifft(P, G, orc.vec_mul(P, orc.fft(P, G, np.concatenate([ah, zpad])),
                                              orc.fft(P, G, np.concatenate([bh, np.zeros(n - bh.size, dtype=np.uint64)]))))
            if int(want[-1]) != 0 or not np.array_equal(oh, want[: oh.size]):
                raise SystemExit("bench.py: product differs from the oracle's")
            checked.append("all %d product coefficients == oracle ifft(fft(a) * fft(b))" % oh.size)
        elif wl in ("open22", "eval22"):
            step(0); torch.cuda.synchronize()
            xh = to_np(x)
            val = orc.poly_eval(P, xh, zpt)
            if int(to_np(scal)[0]) != val:
                raise SystemExit("bench.py: evaluation / remainder differs from the oracle")
            if wl == "open22":
                # the whole quotient: q[n-1] = 0 and q[j-1] = p[j] + z q[j] for every j (division by the monic x - z,
                # src/kzg/setup.rs:63-78): n - 1 equations, one per coefficient
                qh = to_np(y)
                rhs = orc.vec_add(P, xh[1:], orc.vec_mul(P, qh[1:], np.full(n - 1, zpt, dtype=np.uint64)))
                if int(qh[n - 1]) != 0 or not np.array_equal(qh[:-1], rhs):
                    raise SystemExit("bench.py: quotient differs from the division recurrence")
            checked.append("remainder / value == oracle.poly_eval" +
                           ("; all %d quotient coefficients satisfy q[j-1] = p[j] + z q[j]" % n if wl == "open22" else ""))
        elif wl == "roundtrip16":
            plan.forward_dev(x.data_ptr(), y.data_ptr(), stream); torch.cuda.synchronize()
            if not np.array_equal(to_np(y), orc.fft(P, G, to_np(x))):
                raise SystemExit("bench.py: forward differs from the oracle")
            plan.inverse_dev(y.data_ptr(), y.data_ptr(), stream); torch.cuda.synchronize()
            if not np.array_equal(to_np(y), to_np(x)):
                raise SystemExit("bench.py: round trip is not the identity")
            checked.append("forward == oracle.fft, inverse(forward(x)) == x")
        elif wl == "rs16":
            step(0); torch.cuda.synchronize()
            xh, yh = to_np(x), to_np(y)
            for r in (0, batch - 1):
                msg = np.concatenate([xh[r * (n // 2):(r + 1) * (n // 2)], np.zeros(n // 2, dtype=np.uint64)])
                if not np.array_equal(yh[r * n:(r + 1) * n], orc.fft(P, G, msg)):
                    raise SystemExit("bench.py: codeword %d differs from the oracle" % r)
            checked.append("2 codewords == oracle.fft(zero-padded message)")
        elif wl in ("vecmul24", "vecadd24"):
            step(0); torch.cuda.synchronize()
            m = 1 << 16
            f = orc.vec_mul if wl == "vecmul24" else orc.vec_add
            if not np.array_equal(to_np(y)[:m], f(P, to_np(x)[:m], to_np(x2)[:m])):
                raise SystemExit("bench.py: element-wise result differs from the oracle")
            checked.append("first 2^16 elements == oracle")
        return checked

    verified_what = verify() if not args.no_verify else None

    # ---- timing: W untimed warmup steps, a spin-up of untimed steps until the device has been busy >= 50 ms (clocks),
    # then SAMPLES regions of exactly K steps each, every one bracketed by barrier + synchronize on both sides and
    # max-reduced over ranks; `value` / `ms_per_step` come from the MEDIAN region, the minimum is reported beside it.
    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    SAMPLES = max(1, args.samples)

    def timed_regions():
        """SAMPLES regions of exactly K steps, each bracketed by barrier + synchronize on both sides: wall seconds per region
        (max over ranks) and, for the SAME regions, the device time between two HIP events recorded on the launch stream
        right after the opening and right before the closing synchronisation (milliseconds, this rank)"""
        out_, dev_ = [], []
        for _ in range(SAMPLES):
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync_all()
            t0_ = time.perf_counter()
            e0_.record()
            run(args.steps)
            e1_.record()
            sync_all()
            out_.append(time.perf_counter() - t0_)
            dev_.append(e0_.elapsed_time(e1_))
        if world > 1:
            tt = torch.tensor(out_, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            out_ = [float(v) for v in tt.tolist()]
        return out_, dev_

    def device_regions():
        """the same K steps, SAMPLES back-to-back regions delimited by HIP events on the launch stream (torch's current
        stream IS the launch stream; the side lanes join it before the event), ONE synchronisation at the end: the
        device never idles between regions.  Milliseconds per region."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(SAMPLES + 1)]
        evs[0].record()
        for i in range(SAMPLES):
            run(args.steps)
            evs[i + 1].record()
        torch.cuda.synchronize()
        return [evs[i].elapsed_time(evs[i + 1]) for i in range(SAMPLES)]

    rot["R"], rot["lat"] = R_cold, False
    run(args.warmup)
    torch.cuda.synchronize()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:
        run(max(1, args.steps))
        torch.cuda.synchronize()
    dts, dts_dev = timed_regions()                        # the protocol of `value`: throughput mode, R_cold buffers
    dt = float(np.median(dts))
    dt_min = float(min(dts))
    value = world * args.steps * batch / dt               # whole-job units per second (polynomials, products, ...)
    # same regime on the device clock (events); with several CALLER streams an event on one of them does not bracket the
    # others' work: the wall time of the region stands in
    thr_dev_ms = float(np.median(dts_dev)) if S == 1 else dt * 1e3
    warm = None
    # warm = the same buffers every step (Infinity-Cache resident inputs): one pair per transform in flight (the two lanes of
    # the many mode must not write ONE output buffer at the same time)
    R_warm = 2 if (wl == "ntt22" and mode == "many") else 1
    if R_cold > R_warm:
        rot["R"] = R_warm
        run(max(10, args.steps))
        dts_w, _ = timed_regions()
        warm = {"value": world * args.steps * batch / float(np.median(dts_w)), "ms_per_step": float(np.median(dts_w)) / args.steps * 1e3,
                "rotate": R_warm, "note": "the round-1/2 protocol: every step transforms the same buffers (inputs stay in the "
                                          "256 MiB Infinity Cache); `value` is the HBM-cold protocol (rotate = %d)" % R_cold}
    # Latency regime: the same K steps one at a time on ONE stream with the default plan (kernel durations add up)
    S_saved, S = S, 1
    rot["R"], rot["lat"] = R_cold, wl == "ntt22"
    if not rot["lat"]:
        plan0, plans[0] = plans[0], lat_plan
    run(max(10, args.steps))
    dev_ms_samples = device_regions()
    dev_ms = float(np.median(dev_ms_samples))
    lat_warm_ms = None
    if R_cold > 1:
        rot["R"] = 1
        run(max(10, args.steps))
        lat_warm_ms = float(np.median(device_regions()))
    rot["R"], rot["lat"] = R_cold, False
    S = S_saved
    if wl != "ntt22":
        plans[0] = plan0

    # per-kernel device time (hipEvents on the launch stream) for the roofline of the dominant kernel
    pass_ms = None
    if wl in ("ntt22", "batch16"):
        pass_ms = lat_plan.time_passes(x.data_ptr(), y.data_ptr(), inverse=False, iters=50, stream=stream)
    alg_bytes_step = wl_bytes_per_n * (batch / wl_batch) * n
    step_s = (dev_ms / 1e3) / args.steps                             # device time per step on the launch stream
    achieved = alg_bytes_step / step_s / 1e9              # latency regime
    thr_step_s = (thr_dev_ms / 1e3) / args.steps                     # throughput regime (device clock, same regime as value)
    achieved_thr = alg_bytes_step / thr_step_s / 1e9
    # HBM bytes per launch from the PMC passes of the same command under rocprofv3 (tools/profile.sh ->
    # tools/rocprof_summary.py; FETCH_SIZE x2 gfx950 correction, calibrated on this kernel's known byte count).
    # bench.py cannot read PMCs itself: it reports the committed measurement of the dominant kernel IF that file was
    # taken on the current kernel sources (hash stored in the file), else null.
    # Which file: the counters of the kernels `value` RAN.  The headline's regime is two lanes on the 4-column kernels
    # (ntt_tile_kernel<11, false, 2, *>): profiles/latest_pmc_ntt22.json is taken from exactly this command line
    # (tools/profile_all.sh, "many"); the one-transform-at-a-time kernels (8-column tiles) have their own file, reported as
    # roofline.latency_regime.  A Montgomery prime runs other kernels again: latest_pmc_ntt22_mont.json.
    traffic, traffic_note, valu = None, None, None
    pmc_key = wl + ("_mont" if mont else "") + ("_1stream" if (wl == "ntt22" and mode == "streams" and S == 1) else "")
    pmc, why = load_if_current("profiles/latest_pmc_%s.json" % pmc_key, wl)
    if pmc and log2n == wl_log2n and batch == wl_batch:
        # every kernel of the step that moved data (the passes of a transform; the two scan kernels of a division): the
        # step's traffic and VALU count are SUMS over them, each counted once per step
        # (the library's kernels only: under the profiler the same process also runs torch's roll / fill kernels while it sets
        # its buffers up, once, not per step)
        kerns = [kname for kname, c in pmc.get("counters", {}).items() if "_hbm_bytes_per_launch" in c and "ronk::" in kname]
        if wl in ("ntt22", "batch16"):
            # the profiled command also runs the OTHER regime's plan once per sample (the one-at-a-time measurement inside this
            # script: 8-column kernels next to the two-lane regime's 4-column ones): a step is the `num_passes` kernels that were
            # launched most often (the trace's call counts)
            calls = {k_["name"]: k_["calls"] for k_ in pmc.get("kernels", [])}
            kerns.sort(key=lambda kname: -calls.get(kname, 0))
            kerns = kerns[:plan.num_passes()]
        kerns.sort(key=lambda kname: -pmc["counters"][kname].get("_avg_us", 0))
        if kerns:
            cs = [pmc["counters"][kname] for kname in kerns]
            traffic = sum(c["_hbm_bytes_per_launch"]["total"] for c in cs)
            traffic_note = ("fabric-side bytes per STEP = sum over the %d kernel(s) of a step (%s) of FETCH_SIZE x2 (gfx950 "
                            "correction) + WRITE_SIZE per launch (rocprofv3 PMC, profiles/%s); these counters include "
                            "Infinity-Cache hits (MI355X_MICROARCH.md), so this is traffic at the L2's memory side, an upper "
                            "bound on HBM bytes" % (len(kerns), "; ".join("%s: %.0f" % (k_[:60], c["_hbm_bytes_per_launch"]["total"])
                                                                         for k_, c in zip(kerns, cs)),
                                                  pmc.get("source", "latest_pmc_%s.json" % pmc_key)))
            if wl == "ntt22":
                traffic_note += ("; algorithmic bytes per step = %d (16*n); a two-pass plan moves 2x that by construction, "
                                 "anything above is re-reads / split lines" % (16 * n))
            if all("SQ_INSTS_VALU" in c for c in cs):
                valu = {"insts_per_coeff": sum(c["SQ_INSTS_VALU"]["avg"] for c in cs) * 64.0 / (n * batch),
                        "source": "sum over the step's %d kernel(s) of SQ_INSTS_VALU x 64 lanes / coefficients (PMC, same file)" % len(kerns)}
    else:
        traffic_note = "no current PMC file for this workload/kernel (%s): re-run tools/profile.sh" % why
    # Secondary ceiling: 64-bit modular arithmetic is VALU-issue bound.  Static census of the executed
    # path (tools/census.py: instructions and issue slots per coefficient, weights from tools/instr_rate.hip).
    if wl == "ntt22" and not mont:
        cen, why_c = load_if_current("profiles/latest_census.json")
        if cen:
            valu = valu or {}
            issue_us = cen["slots_per_coeff"] * n / VALU_PEAK_LANE_OPS * 1e6
            valu.update({"static_insts_per_coeff": cen["valu_per_coeff"], "issue_slots_per_coeff": cen["slots_per_coeff"],
                         "issue_bound_us": issue_us, "frac_of_valu_peak": issue_us / (step_s * 1e6),
                         "valu_peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                         "note": "issue_bound_us = issue slots per coefficient x n / measured full-rate VALU throughput "
                                 "(profiles/r01_instr_rate_gfx950.txt); frac_of_valu_peak = issue_bound_us / device time per "
                                 "transform: how close the kernel is to its own arithmetic ceiling"})
        elif valu is None:
            valu = {"note": "census " + why_c}
    # `achieved` / `frac` follow `value`: the throughput regime, R_cold buffers, device clock over the timed region.  The
    # latency regime (one transform at a time: kernel durations add up) and the warm variants are reported beside it.
    # `achieved` / `frac` are computed from the SAME clock as `ms_per_step` (the wall time of the median region, max over
    # ranks): frac == algorithmic_bytes_per_step / (ms_per_step / 1e3) / 8e12, recomputable from the line itself.  The HIP-event
    # figure over the same regions (device clock on the launch stream) is reported beside it as frac_device.
    achieved_wall = alg_bytes_step / (dt / args.steps) / 1e9
    lat_pmc = None
    if wl == "ntt22" and not mont and pmc_key == "ntt22":
        lp, _ = load_if_current("profiles/latest_pmc_ntt22_1stream.json", wl)
        if lp:
            lk = [c for kname, c in lp.get("counters", {}).items() if "_hbm_bytes_per_launch" in c and "ronk::" in kname]
            if lk:
                lat_pmc = {"traffic": sum(c["_hbm_bytes_per_launch"]["total"] for c in lk),
                           "valu_insts_per_coeff": (sum(c["SQ_INSTS_VALU"]["avg"] for c in lk) * 64.0 / n
                                                    if all("SQ_INSTS_VALU" in c for c in lk) else None),
                           "source": lp.get("source"),
                           "note": "counters of the kernels frac_latency times (one transform at a time, default plan: 8-column tiles)"}
    roofline = {"bound": "hbm", "achieved": achieved_wall, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_wall / HBM_PEAK_GBS, "frac_device": achieved_thr / HBM_PEAK_GBS,
                "frac_throughput": achieved_thr / HBM_PEAK_GBS,
                "frac_latency": achieved / HBM_PEAK_GBS, "latency_regime": lat_pmc,
                "frac_throughput_warm": (alg_bytes_step * warm["value"] / world / batch / 1e9 / HBM_PEAK_GBS) if warm else None,
                "frac_latency_warm": (alg_bytes_step / (lat_warm_ms / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS) if lat_warm_ms else None,
                "rotate": R_cold,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "kernel": wl_kernel or (
                    "ntt_tile_wl_col_kernel + ntt_tile_wl_row_kernel (2^11-row x 4-column passes, csrc/ntt_tile_wl.h), 2 launches per NTT"
                    if (wl == "ntt22" and log2n == 22 and mode in ("many", "streams") and os.environ.get("RONK_WL", "1") != "0"
                        and (mode == "many" or args.tile_logc == 2))
                    else "ntt_tile_kernel<%d> x %d launches per NTT" % ((log2n + 1) // 2 if log2n > 12 else log2n, plan.num_passes())),
                "algorithmic_bytes_per_step": alg_bytes_step, "device_us_per_step": step_s * 1e6,
                "device_us_per_step_min": min(dev_ms_samples) * 1e3 / args.steps,
                "note": "achieved / frac: the regime of `value` (%s) on the clock of ms_per_step (wall, median region); frac_device / "
                        "frac_throughput: HIP events on the launch stream over the same "
                        "timed regions; inputs and outputs cycling over %d buffer pairs (%d MiB touched between two uses of a "
                        "buffer); frac_latency: one transform at a time on ONE stream, default plan (kernel durations add up), "
                        "same rotation; *_warm: the same buffers every step (inputs stay in the 256 MiB Infinity Cache); median of "
                        "%d regions of %d steps" % (
                            {"many": "ONE plan handle, in_flight = 2, ronk_ntt_forward_many_dev, %d arrays per call" % group,
                             "batch": "ONE plan handle of batch %d" % group,
                             "streams": "%d caller streams, one plan each" % S}[mode] if wl == "ntt22" else "one stream",
                            R_cold, R_cold * 2 * pb * n * 8 >> 20, SAMPLES, args.steps),
                "throughput_GBs": alg_bytes_step * args.steps / dt / 1e9,
                "pass_us": [m * 1e3 for m in pass_ms] if pass_ms else None,
                "valu": valu}

    # how many ranks really took part (an all-reduce of 1 over the job): the driver can check it against --gpus
    ranks_seen = 1
    if world > 1:
        one_ = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(one_)
        ranks_seen = int(one_.item())
    if rank == 0:
        pdesc = "p = 2^64 - 2^32 + 1" if not mont else "p = %#x (g = %d; Montgomery tile kernels)" % (P, G)
        res = {"metric": ("forward NTTs/s, degree 2^%d, 64-bit %s" % (log2n, "Goldilocks prime" if not mont else "prime %#x" % P))
                         if wl == "ntt22" else wl,
               "value": value, "unit": wl_unit,
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "min_ms_per_step": dt_min / args.steps * 1e3,
               "samples_ms_per_step": [t / args.steps * 1e3 for t in dts],
               "timing": "median of %d regions of exactly %d steps after %d warmup steps + >= 50 ms spin-up; each region "
                         "bracketed by barrier + torch.cuda.synchronize, max over ranks" % (SAMPLES, args.steps, args.warmup),
               "verified": bool(verified_what), "verified_how": verified_what,
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "forward NTT, n = 2^%d, batch %d, %s, natural order in/out, device resident"
                          % (log2n, batch, pdesc) if wl in ("ntt22", "batch16") else (wl + (", " + pdesc if mont else "")),
                          "log2n": log2n, "batch": batch, "streams": S, "parallelism": "independent polynomials per GPU (x%d)" % world},
               "roofline": roofline}
        if warm:
            res["warm"] = warm
        res["config"].update({"mode": mode, "group": group, "rotate": R_cold, "plan_handles": len(plans),
                              "in_flight": plans[0].in_flight() if hasattr(L.lib, "ronk_plan_in_flight") else None})
        res["ranks_seen"] = ranks_seen
        assert ranks_seen == args.gpus, "ranks_seen %d != --gpus %d" % (ranks_seen, args.gpus)
        if not args.no_cpu and world == 1 and wl == "ntt22":       # reported at N = 1 only (bench contract)
            res["cpu_baseline"] = cpu_baseline(log2n, p=P, g=G)
        if not args.no_cpu and world == 1 and wl in ("batch16", "rs16"):
            res["cpu_baseline"] = cpu_baseline_batched(log2n)
        if not args.no_cpu and world == 1 and wl == "mul22":
            res["cpu_baseline"] = cpu_baseline_mul(log2n)
        if not args.no_cpu and world == 1 and wl == "roundtrip16":
            res["cpu_baseline"] = cpu_baseline_roundtrip(log2n)
        print(json.dumps(res), flush=True)
    for p_ in plans + [lat_plan]:
        p_.close()
    # Multi-GPU extra (never part of the JSON line above, which is already out): the sharded four-step NTT of
    # BASELINE config 5 (2^26 over the ranks, one RCCL all-to-all over xGMI per transform).  Result goes to
    # stderr and gpurun_out/; any failure here is reported, not raised.  Opt-in (RONK_BENCH_FOURSTEP=1) so that the
    # scaling run of the default benchmark never depends on a collective that was not asked for.
    if world > 1 and wl == "ntt22" and os.environ.get("RONK_BENCH_FOURSTEP", "0") == "1":
        try:
            from ronkathon_amd import dist as rdist
            fs = rdist.bench_fourstep(26, 20, 3)
            if rank == 0:
                sys.stderr.write("fourstep: " + json.dumps(fs) + "\n")
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "fourstep_w%d.json" % world), "w") as f:
                    json.dump(fs, f)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("fourstep extra failed: %r\n" % (e,))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
