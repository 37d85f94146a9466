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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# workload -> (log2n, polynomials per step, algorithmic bytes per step as a multiple of n, unit, dominant kernel)
# Algorithmic bytes: an n-point NTT reads n and writes n 8-byte elements = 16*n.
WORKLOADS = {
    "ntt22":       (22, 1,    16.0,          "NTT/s", None),   # the headline: one forward transform per step
    "batch16":     (16, 1024, 16.0 * 1024,   "NTT/s", None),   # config 4: 1024 x 2^16 in one launch pair
    "mul22":       (22, 1,    48.0,          "op/s",  None),   # config 3: 3 transforms of size 2^22, pad/pointwise/truncate fused
    "roundtrip16": (16, 1,    32.0,          "op/s",  None),   # config 2: forward + inverse
    "open22":      (22, 1,    16.0,          "op/s",  "chunk_horner_kernel + chunk_carry_kernel + lindiv_apply_kernel (csrc/scan_kernels.h)"),
    "eval22":      (22, 1,    8.0,           "op/s",  "chunk_horner_kernel + chunk_carry_kernel (csrc/scan_kernels.h)"),
    "rs16":        (16, 1024, 12.0 * 1024,   "op/s",  None),   # reads n/2, writes n coefficients per codeword
    "vecmul24":    (24, 1,    24.0,          "op/s",  "vec_binary2_kernel<GlOps, VEC_MUL>"),
    "vecadd24":    (24, 1,    24.0,          "op/s",  "vec_binary2_kernel<GlOps, VEC_ADD>"),
}


def synth(n, seed):
    """i.i.d. uniform on [0, p), as int64-viewable uint64"""
    rng = np.random.default_rng(seed)
    P = np.uint64(0xFFFFFFFF00000001)
    x = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    bad = x >= P
    while bad.any():
        x[bad] = rng.integers(0, 2**64, size=int(bad.sum()), dtype=np.uint64)
        bad = x >= P
    return x


def cpu_baseline(log2n, budget_s=12.0):
    """oracle (port of the reference's recursive fft, polynomial/mod.rs:295-323) on one host core"""
    import oracle as orc
    n = 1 << log2n
    x = synth(n, 99)
    w = orc.primitive_root_of_unity(orc.GOLDILOCKS_P, orc.GOLDILOCKS_G, n)  # root excluded from the timed region
    reps, t = 0, 0.0
    while True:
        v = x.copy()
        t0 = time.perf_counter()
        orc.fft_recursive_inplace(orc.GOLDILOCKS_P, v, w)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="ntt22")
    ap.add_argument("--log2n", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="polynomials per launch (default: 1, batch16: 1024)")
    ap.add_argument("--streams", type=int, default=2, help="independent transforms in flight (HIP streams)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    if wl == "fourstep":
        from ronkathon_amd import dist as rdist
        res = rdist.bench_fourstep(args.log2n or 26, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(res))
        if world > 1:
            dist.destroy_process_group()
        return

    wl_log2n, wl_batch, wl_bytes_per_n, wl_unit, wl_kernel = WORKLOADS[wl]
    log2n = args.log2n or wl_log2n
    batch = args.batch or wl_batch
    n = 1 << log2n
    if wl in ("batch16", "rs16"):
        args.streams = 1
    # S independent transforms in flight: each stream has its own plan (scratch), input and output, so
    # the load/store phases of one transform overlap the VALU-bound butterflies of another.
    S = max(1, args.streams) if wl in ("ntt22", "batch16") else 1
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream() for _ in range(S - 1)]
    xs = [torch.from_numpy(synth(n * batch, 0x5EED0000 + rank * 16 + i).view(np.int64)).cuda() for i in range(S)]
    ys = [torch.empty_like(xs[0]) for _ in range(S)]
    # several transforms in flight -> narrower tiles (two workgroups per CU); one at a time -> default plan
    tile_lc = 2 if (S > 1 and log2n >= 20) else -1
    plans = [L.Plan(P, G, log2n, batch, local_rank, tile_log2_columns=tile_lc) for _ in range(S)]
    lat_plan = L.Plan(P, G, log2n, batch, local_rank) if tile_lc >= 0 else plans[0]
    x, y, plan, stream = xs[0], ys[0], plans[0], main_stream.cuda_stream
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
            plans[k].forward_dev(xs[k].data_ptr(), ys[k].data_ptr(), streams[k].cuda_stream)

    def run(count):
        for i in range(count):
            step(i)

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    value = world * args.steps * batch / dt          # whole-job units per second (polynomials, products, ...)

    # Roofline of the dominant kernel: the same K steps again, one at a time on ONE stream with the default plan,
    # bracketed by HIP events recorded on the launch stream (torch's current stream IS the launch stream here).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    S_saved, S, plan0 = S, 1, plans[0]
    plans[0] = lat_plan
    run(10)
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    S = S_saved
    plans[0] = plan0
    dev_ms = ev0.elapsed_time(ev1)

    # per-kernel device time (hipEvents on the launch stream) for the roofline of the dominant kernel
    pass_ms = None
    if wl in ("ntt22", "batch16"):
        pass_ms = lat_plan.time_passes(x.data_ptr(), y.data_ptr(), inverse=False, iters=50, stream=stream)
    alg_bytes_step = wl_bytes_per_n * (batch / wl_batch) * n
    step_s = (dev_ms / 1e3) / args.steps                             # device time per step on the launch stream
    achieved = alg_bytes_step / step_s / 1e9
    # HBM bytes per launch from the PMC passes of the same command under rocprofv3 (tools/rocprof_summary.py;
    # FETCH_SIZE x2 gfx950 correction, calibrated on this kernel's known byte count); bench.py cannot
    # read PMCs itself, so it reports the committed measurement of the dominant kernel, or null.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc_ntt22.json")) as f:
            for kname, c in json.load(f)["counters"].items():
                if "ntt_tile_kernel<11, false>" in kname and "_hbm_bytes_per_launch" in c and wl == "ntt22":
                    traffic = c["_hbm_bytes_per_launch"]["total"]
    except Exception:
        pass
    traffic_note = None
    if traffic:
        traffic_note = ("HBM bytes per kernel launch (rocprofv3 PMC, profiles/latest_pmc_ntt22.json); "
                        "algorithmic bytes = %d per transform (16*n), i.e. %d per launch of the 2-launch plan; each launch reads and "
                        "writes the whole vector once, so measured traffic per launch is ~2x the per-launch algorithmic share "
                        "(inherent to a two-pass transform), with no wasted re-reads" % (16 * n, 8 * n))
    elif wl != "ntt22":
        try:   # the other workloads: the dominant (longest) kernel of the committed PMC run of the same command
            with open(os.path.join(ROOT, "profiles", "latest_pmc_workloads.json")) as f:
                ks = json.load(f).get(wl, {})
            if ks and log2n == wl_log2n and batch == wl_batch:
                kname, kv = max(ks.items(), key=lambda kv_: kv_[1].get("avg_us") or 0.0)
                traffic = kv["hbm_bytes_per_launch"]
                traffic_note = ("HBM bytes per launch of %s (rocprofv3 PMC, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, "
                                "profiles/r01_rocprof_%s.txt); all kernels of one step move %d bytes against %d algorithmic"
                                % (kname, wl, sum(v["hbm_bytes_per_launch"] for v in ks.values()) *
                                   (2 if wl in ("batch16", "rs16") else 1), int(wl_bytes_per_n * n)))
        except Exception:
            pass
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_note": traffic_note,
                "kernel": wl_kernel or "ntt_tile_kernel<%d> x %d launches per NTT" % ((log2n + 1) // 2 if log2n > 12 else log2n,
                                                                                         plan.num_passes()),
                "algorithmic_bytes_per_step": alg_bytes_step, "device_us_per_step": step_s * 1e6,
                "note": "achieved/frac: one transform at a time on ONE stream, default plan (kernel durations); "
                        "value: %d streams%s" % (S, ", plans tuned for concurrency (tile_log2_columns=2)" if tile_lc >= 0 else ""),
                "throughput_GBs": alg_bytes_step * args.steps / dt / 1e9,
                "pass_us": [m * 1e3 for m in pass_ms] if pass_ms else None}

    if rank == 0:
        res = {"metric": "forward NTTs/s, degree 2^%d, 64-bit Goldilocks prime" % log2n if wl == "ntt22" else wl,
               "value": value, "unit": wl_unit,
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "forward NTT, n = 2^%d, batch %d, p = 2^64 - 2^32 + 1, natural order in/out, device resident"
                          % (log2n, batch) if wl in ("ntt22", "batch16") else wl,
                          "log2n": log2n, "batch": batch, "streams": S, "parallelism": "independent polynomials per GPU (x%d)" % world},
               "roofline": roofline}
        if not args.no_cpu and world == 1 and wl == "ntt22":       # reported at N = 1 only (bench contract)
            res["cpu_baseline"] = cpu_baseline(log2n)
        if not args.no_cpu and world == 1 and wl in ("batch16", "rs16"):
            res["cpu_baseline"] = cpu_baseline_batched(log2n)
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
