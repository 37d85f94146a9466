"""ctypes binding of ronkathon_amd/libronk_ntt.so (the C ABI in include/ronk_ntt.h).

There is no CPU implementation behind this module: if the shared library is missing it
raises at import, and without a HIP device every compute call raises RonkPanic(-10).
"""
import ctypes as C
import os

import numpy as np

# PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  In a process that uses both, torch's copy has to be
# mapped BEFORE the system runtime this library links against (the other order leaves torch with "No HIP GPUs
# are available" -- measured on ROCm 7.2 + torch 2.10/rocm7.0).  So: import torch first when it is installed.
# The library itself never calls into torch; callers hand it raw device pointers and hipStream_t values.
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001  (no torch: nothing to order)
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
# RONK_LIB_PATH: developer A/B runs against another build of the SAME C ABI (e.g. last round's .so on the same box)
LIB_PATH = os.environ.get("RONK_LIB_PATH") or os.path.join(_HERE, "libronk_ntt.so")

GOLDILOCKS_P = 0xFFFFFFFF00000001
GOLDILOCKS_G = 7

OK, ERR_NO_ROOT, ERR_ZERO_INVERSE, ERR_NOT_POW2, ERR_NOT_PRIME = 0, -1, -2, -3, -4
ERR_NO_GENERATOR, ERR_INDEX, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = -5, -6, -7, -8, -9, -10
ERR_NOT_ON_CURVE = -11
ERR_RCCL = -12
ERR_NOT_RESIDUE = -13
EXCHANGE_MESH, EXCHANGE_RCCL = 0, 1


class RonkPanic(Exception):
    """A non-zero return code: what the reference reports by panicking (same message text)."""

    def __init__(self, code, detail=""):
        msg = lib.ronk_strerror(code).decode()
        if code in (ERR_HIP, ERR_RCCL):
            msg += ": " + lib.ronk_last_hip_error().decode()
        super().__init__(msg + (" " + detail if detail else ""))
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "ronkathon_amd: %s not found -- build the HIP extension first (`make` at the repo root or "
        "`python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_u64, _sz, _pu, _vp, _int = C.c_uint64, C.c_size_t, C.POINTER(C.c_uint64), C.c_void_p, C.c_int
_SIG = {
    "ronk_strerror": (C.c_char_p, [_int]),
    "ronk_last_hip_error": (C.c_char_p, []),
    "ronk_device_count": (_int, [C.POINTER(_int)]),
    "ronk_primitive_element": (_int, [_u64, _pu]),
    "ronk_root_of_unity": (_int, [_u64, _u64, _u64, _pu]),
    "ronk_check_prime": (_int, [_u64]),
    "ronk_vec_add": (_int, [_u64, _vp, _vp, _vp, _sz]),
    "ronk_vec_sub": (_int, [_u64, _vp, _vp, _vp, _sz]),
    "ronk_vec_mul": (_int, [_u64, _vp, _vp, _vp, _sz]),
    "ronk_vec_neg": (_int, [_u64, _vp, _vp, _sz]),
    "ronk_vec_inv": (_int, [_u64, _vp, _vp, _sz]),
    "ronk_vec_pow": (_int, [_u64, _vp, _u64, _vp, _sz]),
    "ronk_vec_euler": (_int, [_u64, _vp, _vp, _sz]),
    "ronk_vec_sqrt": (_int, [_u64, _vp, _vp, _vp, _sz]),
    "ronk_vec_euler_dev": (_int, [_u64, _vp, _vp, _sz, _vp]),
    "ronk_vec_sqrt_dev": (_int, [_u64, _vp, _vp, _vp, _sz, _vp, _vp]),
    "ronk_vec_add_dev": (_int, [_u64, _vp, _vp, _vp, _sz, _vp]),
    "ronk_vec_sub_dev": (_int, [_u64, _vp, _vp, _vp, _sz, _vp]),
    "ronk_vec_mul_dev": (_int, [_u64, _vp, _vp, _vp, _sz, _vp]),
    "ronk_plan_create": (_int, [C.POINTER(_vp), _u64, _u64, C.c_uint32, _u64, _int]),
    "ronk_plan_create_tuned": (_int, [C.POINTER(_vp), _u64, _u64, C.c_uint32, _u64, _int, _int, _int]),
    "ronk_plan_create_opts": (_int, [C.POINTER(_vp), _u64, _u64, C.c_uint32, _u64, _int, _vp]),
    "ronk_plan_in_flight": (_int, [_vp]),
    "ronk_ntt_forward_many_dev": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _sz, _vp]),
    "ronk_ntt_inverse_many_dev": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _sz, _vp]),
    "ronk_plan_destroy": (_int, [_vp]),
    "ronk_plan_path": (_int, [_vp]),
    "ronk_ntt_forward": (_int, [_vp, _vp, _vp, _vp]),
    "ronk_ntt_inverse": (_int, [_vp, _vp, _vp]),
    "ronk_ntt_forward_dev": (_int, [_vp, _vp, _vp, _vp]),
    "ronk_ntt_inverse_dev": (_int, [_vp, _vp, _vp, _vp]),
    "ronk_lagrange_nodes": (_int, [_u64, _u64, _vp, _sz]),
    "ronk_fft": (_int, [_u64, _u64, _vp, _vp, _vp, _sz]),
    "ronk_ifft": (_int, [_u64, _u64, _vp, _vp, _sz]),
    "ronk_dft": (_int, [_u64, _u64, _vp, _vp, _sz]),
    "ronk_plan_num_passes": (_int, [_vp]),
    "ronk_plan_time_passes": (_int, [_vp, _vp, _vp, _int, _int, C.POINTER(C.c_float), _vp]),
    "ronk_poly_mul": (_int, [_u64, _u64, _vp, _sz, _vp, _sz, _vp]),
    "ronk_poly_mul_dev": (_int, [_u64, _u64, _vp, _sz, _vp, _sz, _vp, _vp]),
    "ronk_poly_add": (_int, [_u64, _vp, _sz, _vp, _sz, _vp]),
    "ronk_poly_sub": (_int, [_u64, _vp, _sz, _vp, _sz, _vp]),
    "ronk_poly_eval": (_int, [_u64, _vp, _sz, _u64, _pu]),
    "ronk_poly_eval_dev": (_int, [_u64, _vp, _sz, _u64, _vp, _vp]),
    "ronk_poly_div_linear_dev": (_int, [_u64, _vp, _sz, _u64, _u64, _vp, _vp, _vp]),
    "ronk_lagrange_eval": (_int, [_u64, _vp, _vp, _sz, _u64, _pu]),
    "ronk_poly_divrem": (_int, [_u64, _vp, _sz, _vp, _sz, _vp, _vp]),
    "ronk_rs_encode": (_int, [_u64, _u64, _vp, _sz, _sz, _vp, _vp]),
    "ronk_rs_decode": (_int, [_u64, _vp, _vp, _sz, _vp]),
    "ronk_rs_encode_batch_dev": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "ronk_lde_batch_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _vp]),
    "ronk_curve_msm": (_int, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "ronk_msm_bn254": (_int, [_vp, _vp, _sz, _vp]),
    "ronk_msm_bn254_dev": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "ronk_poly_div_linear_bn254_dev": (_int, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "ronk_kzg_open_bn254_dev": (_int, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ronk_kzg_open_bn254": (_int, [_vp, _sz, _vp, _vp, _sz, _vp, _vp]),
    "ronk_dist_plan_create": (_int, [C.POINTER(_vp), C.c_uint32, _int, _int, _int, _int]),
    "ronk_dist_plan_destroy": (_int, [_vp]),
    "ronk_dist_plan_create_chunked": (_int, [C.POINTER(_vp), C.c_uint32, _int, _int, _int, _int, _int]),
    "ronk_dist_plan_create_p": (_int, [C.POINTER(_vp), _u64, _u64, C.c_uint32, _int, _int, _int, _int, _int]),
    "ronk_dist_phase1_chunk_dev": (_int, [_vp, _int, _vp, _vp, _vp]),
    "ronk_dist_phase1_dev": (_int, [_vp, _vp, _vp, _vp]),
    "ronk_dist_phase2_dev": (_int, [_vp, _vp, _vp, _vp]),
    "ronk_vec_neg_dev": (_int, [_u64, _vp, _vp, _sz, _vp]),
    "ronk_vec_pow_dev": (_int, [_u64, _vp, _u64, _vp, _sz, _vp]),
    "ronk_vec_inv_dev": (_int, [_u64, _vp, _vp, _sz, _vp, _vp]),
    "ronk_dft_dev": (_int, [_u64, _u64, _vp, _vp, _sz, _vp]),
    "ronk_lagrange_eval_dev": (_int, [_u64, _vp, _vp, _sz, _u64, _vp, _vp, _vp]),
    "ronk_poly_divrem_dev": (_int, [_u64, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ronk_poly_divrem_full_dev": (_int, [_u64, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ronk_rs_decode_dev": (_int, [_u64, _vp, _vp, _sz, _vp, _vp, _vp]),
    "ronk_curve_msm_dev": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "ronk_sharded_plan_create": (_int, [C.POINTER(_vp), C.c_uint32, _int, C.POINTER(_int), _int, _int]),
    "ronk_sharded_plan_create_ex": (_int, [C.POINTER(_vp), C.c_uint32, _int, C.POINTER(_int), _int, _int, _int]),
    "ronk_sharded_time_stages": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_float)]),
    "ronk_sharded_plan_create_p": (_int, [C.POINTER(_vp), _u64, _u64, C.c_uint32, _int, C.POINTER(_int), _int, _int, _int]),
    "ronk_sharded_plan_exchange": (_int, [_vp]),
    "ronk_sharded_plan_peer_access": (_int, [_vp, C.POINTER(_int), _int]),
    "ronk_sharded_plan_destroy": (_int, [_vp]),
    "ronk_sharded_plan_info": (_int, [_vp, _pu, _pu, _pu, C.POINTER(_int)]),
    "ronk_ntt_sharded_dev": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "ronk_sharded_sync": (_int, [_vp]),
    "ronk_ntt_sharded": (_int, [_vp, _vp, _vp]),
    "ronk_dev_alloc": (_int, [C.POINTER(_vp), _sz]),
    "ronk_dev_free": (_int, [_vp]),
    "ronk_memcpy_h2d": (_int, [_vp, _vp, _sz]),
    "ronk_memcpy_d2h": (_int, [_vp, _vp, _sz]),
    "ronk_dev_sync": (_int, []),
    "ronk_set_device": (_int, [_int]),
    "ronk_get_device": (_int, [C.POINTER(_int)]),
    "ronk_trim_workspace": (_int, []),
}
for _name, (_res, _args) in _SIG.items():
    if os.environ.get("RONK_LIB_PATH") and not hasattr(lib, _name):
        continue              # an older build in an A/B run may lack the newest entry points
    _f = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _f.restype, _f.argtypes = _res, _args

EXPORTS = sorted(_SIG)


def check(rc, detail=""):
    if rc != 0:
        raise RonkPanic(rc, detail)


def device_count():
    n = _int(0)
    check(lib.ronk_device_count(C.byref(n)))
    return n.value


def arr(x):
    """canonical residues as a contiguous numpy uint64 array"""
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def ptr(a):
    return a.ctypes.data_as(_vp)


def out_scalar(fn, *args):
    o = _u64(0)
    check(fn(*args, C.byref(o)))
    return o.value


class PlanOpts(C.Structure):
    """ronk_plan_opts (include/ronk_ntt.h)"""
    _fields_ = [("tile_log2_columns", _int), ("twiddle_matrix_log2_max", _int), ("in_flight", _int), ("split_log2_rows", _int),
                ("three_pass_from_log2", _int), ("reserved", _int * 3)]

    def __init__(self, tile_log2_columns=-1, twiddle_matrix_log2_max=-1, in_flight=-1):
        super().__init__(tile_log2_columns, twiddle_matrix_log2_max, in_flight)


class Plan:
    """RAII wrapper of ronk_plan: (p, g, n = 2^log2n, batch) on one device."""

    def __init__(self, p, g, log2n, batch=1, device=-1, tile_log2_columns=-1, twiddle_matrix_log2_max=-1, in_flight=-1):
        self.h = None
        h = _vp()
        if in_flight == -1 and not hasattr(lib, "ronk_plan_create_opts"):   # an older build in an A/B run
            check(lib.ronk_plan_create_tuned(C.byref(h), p, g, log2n, batch, device, tile_log2_columns,
                                             twiddle_matrix_log2_max))
        else:
            opts = PlanOpts(tile_log2_columns, twiddle_matrix_log2_max, in_flight)
            check(lib.ronk_plan_create_opts(C.byref(h), p, g, log2n, batch, device, C.byref(opts)))
        self.h, self.p, self.g, self.log2n, self.n, self.batch = h, p, g, log2n, 1 << log2n, batch

    def in_flight(self):
        """lanes the plan keeps in flight (ronk_plan_opts::in_flight resolved): 1 or 2"""
        return lib.ronk_plan_in_flight(self.h)

    def forward_many_dev(self, d_ins, d_outs, stream=0, inverse=False):
        """`len(d_ins)` independent [batch][n] device arrays in one call (ronk_ntt_forward_many_dev)"""
        k = len(d_ins)
        a = (_vp * k)(*d_ins)
        b = (_vp * k)(*d_outs)
        f = lib.ronk_ntt_inverse_many_dev if inverse else lib.ronk_ntt_forward_many_dev
        check(f(self.h, a, b, k, stream))

    def close(self):
        if getattr(self, "h", None):
            lib.ronk_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        if lib is not None:      # module globals are already torn down at interpreter exit
            self.close()

    def num_passes(self):
        return lib.ronk_plan_num_passes(self.h)

    def path(self):
        """1 = tiled Goldilocks kernels, 2 = the tile kernels over a Montgomery prime, 0 = generic radix-2 path (ronk_plan_path)"""
        return lib.ronk_plan_path(self.h)

    def forward(self, x, nodes=False):
        x = arr(x)
        assert x.size == self.n * self.batch
        out = np.empty_like(x)
        nd = np.empty(self.n, dtype=np.uint64) if nodes else None
        check(lib.ronk_ntt_forward(self.h, ptr(x), ptr(out), ptr(nd) if nodes else None))
        return (out, nd) if nodes else out

    def inverse(self, x):
        x = arr(x)
        assert x.size == self.n * self.batch
        out = np.empty_like(x)
        check(lib.ronk_ntt_inverse(self.h, ptr(x), ptr(out)))
        return out

    def forward_dev(self, d_in, d_out, stream=0):
        check(lib.ronk_ntt_forward_dev(self.h, d_in, d_out, stream))

    def inverse_dev(self, d_in, d_out, stream=0):
        check(lib.ronk_ntt_inverse_dev(self.h, d_in, d_out, stream))

    def rs_encode_batch_dev(self, d_msgs, k, d_ys, stream=0):
        """batched Message::encode::<N> (codes/reed_solomon.rs:42-52): [batch][k] messages -> [batch][n] y-coordinates"""
        check(lib.ronk_rs_encode_batch_dev(self.h, d_msgs, k, d_ys, stream))

    def time_passes(self, d_in, d_out, inverse=False, iters=20, stream=0):
        np_ = self.num_passes()
        ms = (C.c_float * np_)()
        check(lib.ronk_plan_time_passes(self.h, d_in, d_out, int(inverse), iters, ms, stream))
        return [float(v) for v in ms]


class ShardedPlan:
    """ronk_sharded_plan: the four-step NTT sharded over `devices` inside the library (single process; peer copies over
    xGMI in column chunks).  Rank g = devices[g]; the same ordinal may appear several times (logical ranks sharing a GPU:
    how the single-GPU tests drive this path)."""

    def __init__(self, log2n, devices, inverse=False, chunks=0, exchange=EXCHANGE_MESH, p=None, g=None):
        """p, g: any odd prime with 2^log2n | p - 1 and a primitive element of it (ronk_sharded_plan_create_p); default Goldilocks"""
        self.h = None
        h = _vp()
        devs = (_int * len(devices))(*devices)
        if p is None:
            check(lib.ronk_sharded_plan_create_ex(C.byref(h), log2n, int(inverse), devs, len(devices), chunks, exchange))
        else:
            check(lib.ronk_sharded_plan_create_p(C.byref(h), int(p), int(g), log2n, int(inverse), devs, len(devices), chunks, exchange))
        self.exchange = exchange
        self.h, self.n, self.ndev = h, 1 << log2n, len(devices)
        r, c, per, ch = _u64(0), _u64(0), _u64(0), _int(0)
        check(lib.ronk_sharded_plan_info(h, C.byref(r), C.byref(c), C.byref(per), C.byref(ch)))
        self.R, self.C, self.per_rank, self.chunks = r.value, c.value, per.value, ch.value

    def peer_access(self):
        """(matrix, staged): matrix[g][h] in {0 same device, 1 direct peer access, 2 staged through the host}; staged = the
        number of rank pairs whose blocks do NOT travel peer-to-peer (ronk_sharded_plan_peer_access)"""
        w = self.ndev
        m = (_int * (w * w))()
        staged = lib.ronk_sharded_plan_peer_access(self.h, m, w * w)
        if staged < 0:
            check(staged)
        return [[m[g * w + h] for h in range(w)] for g in range(w)], staged

    def transform(self, x):
        """host natural-order vector -> natural-order result (ronk_ntt_sharded)"""
        x = arr(x)
        assert x.size == self.n
        out = np.empty_like(x)
        check(lib.ronk_ntt_sharded(self.h, ptr(x), ptr(out)))
        return out

    def time_stages(self, d_in, d_out):
        """one transform in three serialised stages (ronk_sharded_time_stages): [phase 1, exchange, phase 2] in ms, and the
        achieved GB/s per directed link of the exchange"""
        a = (_vp * self.ndev)(*d_in)
        b = (_vp * self.ndev)(*d_out)
        ms = (C.c_float * 3)()
        check(lib.ronk_sharded_time_stages(self.h, a, b, ms))
        t = [float(v) for v in ms]
        per_link = self.n * 8 / (self.ndev ** 2)
        return {"phase1_ms": t[0], "exchange_ms": t[1], "phase2_ms": t[2],
                "bytes_per_directed_link": per_link, "GBs_per_directed_link": per_link / (t[1] * 1e-3) / 1e9 if t[1] > 0 else None}

    def transform_dev(self, d_in, d_out):
        """device pointers per rank (lists of ints); asynchronous, see sync()"""
        a = (_vp * self.ndev)(*d_in)
        b = (_vp * self.ndev)(*d_out)
        check(lib.ronk_ntt_sharded_dev(self.h, a, b))

    def sync(self):
        check(lib.ronk_sharded_sync(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib.ronk_sharded_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        if lib is not None:
            self.close()
