// ronk_plan.hip -- C ABI of libronk_ntt.so, part 2: plan construction (plan.h), twiddle upload, transform launches
// (tile_kernels.hip; generic primes: field_kernels.h), staging for the host-pointer entry points, the plan cache and
// what is built on it (fft / ifft / dft, polynomial multiply, batched Reed-Solomon encode).
#include <map>

#include "runtime.h"
#include "ntt_mul.h"
#include "ntt_aux_kernels.h"

// ------------------------------------------------------------------------------------- plans



extern "C" int ronk_plan_destroy(ronk_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  pl->fwd.release();
  pl->inv.release();
  if (pl->d_tmp) (void)hipFree(pl->d_tmp);
  if (pl->d_stage_in) (void)hipFree(pl->d_stage_in);
  if (pl->d_stage_out) (void)hipFree(pl->d_stage_out);
  if (pl->d_wtab_f) (void)hipFree(pl->d_wtab_f);
  if (pl->d_wtab_i) (void)hipFree(pl->d_wtab_i);
  if (pl->scratch_ev) (void)hipEventDestroy(pl->scratch_ev);
  if (pl->d_tmp2) (void)hipFree(pl->d_tmp2);
  if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
  if (pl->ev_join) (void)hipEventDestroy(pl->ev_join);
  if (pl->side) (void)hipStreamDestroy(pl->side);
  for (auto e : pl->st_ev) (void)hipEventDestroy(e);
  if (pl->st_h2d) (void)hipStreamDestroy(pl->st_h2d);
  if (pl->st_d2h) (void)hipStreamDestroy(pl->st_d2h);
  if (pl->st_exec) (void)hipStreamDestroy(pl->st_exec);
  delete pl;
  return RONK_OK;
}

extern "C" int ronk_plan_create(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device) {
  return ronk_plan_create_tuned(out, p, g, log2n, batch, device, -1, -1);
}

extern "C" int ronk_plan_path(const ronk_plan* pl) { return pl ? (pl->fast ? (pl->mont_tiled ? 2 : 1) : 0) : RONK_ERR_INVALID; }

// would a plan for n = 2^k over (p, g) run the tile kernels?  (the rule of ronk_plan_create_opts, without building anything)
static bool tiled_plan_exists(u64 p, u64 g, int k) {
  if (k < 4 || k > 30 || p < 3 || !(p & 1) || (p - 1) % ((u64)1 << k) != 0) return false;
  if (p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G) return true;
  static const bool no_mont_tiles = getenv("RONK_NO_MONT_TILES") != nullptr;
  return !no_mont_tiles && h_powmod(g % p, (p - 1) / 2, p) == p - 1;
}

extern "C" int ronk_plan_create_tuned(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device,
                                      int tile_log2_columns, int twiddle_matrix_log2_max) {
  ronk_plan_opts o;
  memset(&o, 0, sizeof o);
  o.tile_log2_columns = tile_log2_columns;
  o.twiddle_matrix_log2_max = twiddle_matrix_log2_max;
  o.in_flight = -1;
  return ronk_plan_create_opts(out, p, g, log2n, batch, device, &o);
}

extern "C" int ronk_plan_in_flight(const ronk_plan* pl) { return pl ? pl->in_flight : RONK_ERR_INVALID; }

extern "C" int ronk_plan_create_opts(ronk_plan** out, uint64_t p, uint64_t g, uint32_t log2n, uint64_t batch, int device,
                                     const ronk_plan_opts* opts) {
  const int tile_log2_columns = opts ? opts->tile_log2_columns : -1;
  const int twiddle_matrix_log2_max = opts ? opts->twiddle_matrix_log2_max : -1;
  int in_flight = opts ? opts->in_flight : -1;
  const int split_ka = opts ? opts->split_log2_rows : 0;
  if (!out || batch == 0 || log2n > 36) return RONK_ERR_INVALID;
  if (in_flight < -1 || in_flight == 0 || in_flight > 2) return RONK_ERR_INVALID;
  *out = nullptr;
  RCHK(ronk_check_prime(p));                                   // PrimeField::new -> is_prime
  if (p < 2) return RONK_ERR_INVALID;
  const u64 n = (u64)1 << log2n;
  if ((p - 1) % n != 0) return RONK_ERR_NO_ROOT;               // field/mod.rs:72, polynomial/mod.rs:361
  RCHK(need_device());
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  ronk_plan* pl = new ronk_plan();
  pl->p = p; pl->g = g % p; pl->log2n = log2n; pl->n = n; pl->batch = batch; pl->device = device;
  int rc = make_field(p, &pl->field);
  if (rc) { delete pl; return rc; }
  // tile path: 16 <= n <= 2^30 (32-bit lane offsets inside a tile, grids below 2^31 workgroups), over
  //   Goldilocks with the reference generator 7 -- the shift-twiddle kernels (gl64.h), or
  //   any other odd prime (and Goldilocks with another generator) whose g is a quadratic non-residue, i.e. omega_n =
  //   g^((p-1)/n) has order exactly n for every power of two n | p - 1 -- the same kernels over Montgomery arithmetic
  //   (field_policy.h MontField, tile_kernels_mont.hip).  RONK_NO_MONT_TILES=1: the pre-round-5 behaviour (A/B).
  // Anything else (a g that generates no full 2-power subgroup: the reference's recursion is then NOT the DFT and the
  // radix-2 path restates it stage by stage; n < 16; n > 2^30) takes the generic radix-2 path.
  const bool size_ok = log2n >= 4 && log2n <= 30 && batch < ((u64)1 << 31) && (double)batch * (double)n / 2048.0 < 2.0e9;
  const bool gl_tiled = p == RONK_GOLDILOCKS_P && pl->g == RONK_GOLDILOCKS_G;
  static const bool no_mont_tiles = getenv("RONK_NO_MONT_TILES") != nullptr;
  pl->mont_tiled = !gl_tiled && !no_mont_tiles && size_ok && (p & 1) && p > 2 && h_powmod(pl->g, (p - 1) / 2, p) == p - 1;
  pl->fast = size_ok && (gl_tiled || pl->mont_tiled);
  const HostField hf = pl->mont_tiled ? HostField::montgomery(p, pl->g) : HostField::goldilocks();
  // generic radix-2 path: 32-bit bit reversal and element indices (field_kernels.h bitrev32)
  if (!pl->fast && log2n > 32) { delete pl; return RONK_ERR_UNSUPPORTED; }
  if (pl->fast) {
    // columns per tile = 2^max_logc at most (4 = 128-byte segments, 1 workgroup per CU at 2^11 rows; 2 = 32-byte
    // segments but two workgroups per CU, better when several transforms are in flight); RONK_MAX_LOGC overrides
    // Single-pass plans (n <= 2^12, the batch is the column axis: a column is a whole polynomial, n elements away from
    // its neighbour) default to the narrowest tile that still fills 256 work-items, C = 2^(12 - log2n): a wave then
    // covers 64/C consecutive rows of each polynomial -- 2^12: 0.509 -> 0.431 ms, 2^11: 0.500 -> 0.414, 2^10: 0.512 ->
    // 0.429 per 2^26 coefficients against C = 16.
    int max_logc = log2n <= 12 ? 0 : 4;
    bool auto_tiles = true;   // per-pass tile preferences of plan.h apply unless a width was asked for
    // Two lanes (in_flight = 2): a second stream with its own scratch for ronk_ntt_forward_many_dev, and the second half of
    // a batched call on it.  At 2^19 .. 2^22 both passes then get 4-column tiles -- two workgroups per CU -- unless a width
    // was asked for: the configuration in which two independent kernels overlap.  Automatic = 1: measured
    // (round 3, HBM-cold) the lanes pay for independent arrays (many_dev: 17.6 k -> 22.9 k NTT/s at 2^22) but not for the
    // halves of ONE batched launch pair, whose workgroups drift apart by themselves (2^22 x 16: 20.9 k without, 20.1 k
    // with; multiply 2^22: 157 vs 169 us).  RONK_IN_FLIGHT=1 / 2 overrides automatic.
    if (in_flight < 0) {
      in_flight = 1;
      if (const char* e = getenv("RONK_IN_FLIGHT")) { int v = atoi(e); if (v == 1 || v == 2) in_flight = v; }
    }
    if (in_flight == 2 && log2n >= 19 && log2n <= 22 && tile_log2_columns < 0 && !getenv("RONK_MAX_LOGC")) {
      max_logc = 2; auto_tiles = false;
    }
    if (const char* e = getenv("RONK_MAX_LOGC")) { int v = atoi(e); if (v >= 0 && v <= 8) { max_logc = v; auto_tiles = false; } }
    if (tile_log2_columns >= 0 && tile_log2_columns <= 8) { max_logc = tile_log2_columns; auto_tiles = false; }
    if (getenv("RONK_WG_FLOOR_LOG")) auto_tiles = false;
    // Full inter-pass twiddle matrix (one coalesced load + one multiply instead of two gathers + two
    // multiplies) while it stays L2-resident: up to 2^18 entries = 2 MiB.  Larger matrices would add an
    // n-element HBM read per transform (measured +4 % speed at 2^22 for +25 % traffic): left to RONK_TWF_MAX_LOG.
    // Round 3 (specialised kernels, HBM-cold buffers, same-box A/B/A/B: profiles/r03_twf_ab.txt): at 2^21 and 2^22 the
    // matrix is worth its n-element read after all -- pass 1 executes 139 instead of 158 VALU per coefficient and drops two
    // table gathers per coefficient: one 2^21 transform 42.3 -> 36.8 us, one 2^22 transform 56.5 -> 52.9 us, 2^22 with two
    // lanes 23.1 k -> 23.8 k NTT/s (2^20: 27.9 -> 28.7 us, slower: left alone).  64 MiB of tables per 2^22 plan (both
    // directions) out of 288 GB.
    int twf_max_log = (log2n == 21 || log2n == 22) ? (int)log2n : 18;
    if (const char* e = getenv("RONK_TWF_MAX_LOG")) { int v = atoi(e); if (v >= 0 && v <= 26) twf_max_log = v; }
    if (twiddle_matrix_log2_max >= 0 && twiddle_matrix_log2_max <= 26) twf_max_log = twiddle_matrix_log2_max;
    // Two passes up to 2^22; from 2^23 three passes are faster although they move 1.5x the bytes: a two-pass plan
    // would need 2^12-row tiles of only 4 columns (32-byte row segments) -- 2^24: 0.331 -> 0.291 ms, 2^23 x 8: 1.135 ->
    // 1.027 ms; at 2^22 two passes win (58.9 vs 66.6 us).  RONK_THREE_PASS_FROM overrides.
    // Round 3: with the specialised kernels two passes (2^12 x 2^11) win again at 2^23 for ONE transform (117.7 -> 114.7 us with a
    // generic first pass, 110 with the (12, 2, 1) shape); batches of 2^23 keep three passes (measured in round 2).
    int three_from = (log2n == 23 && batch == 1) ? 24 : 23;
    if (const char* e = getenv("RONK_THREE_PASS_FROM")) { int v = atoi(e); if (v >= 13 && v <= 25) three_from = v; }
    if (opts && opts->three_pass_from_log2 >= 13 && opts->three_pass_from_log2 <= 25) three_from = opts->three_pass_from_log2;
    // The full twiddle matrix of a 2^11-row x 4-column column pass, transposed (plan.h twf_transposed), where ntt_tile_wl.h's
    // kernels run it.  EXPERIMENT, measured SLOWER over Goldilocks and kept off (round 6, same box, profiles/r06_wl_ab.txt: two
    // lanes at 2^22 24.1 k -> 22.3 k NTT/s, 2^21 38.0 k -> 35.2 k; the Montgomery prime +1 %): whole 128-byte lines per load
    // instruction lose against 32-byte pieces of lines that four neighbouring tiles pull through the same L2 at the same time.
    // RONK_TWF_T = 0 (default) never, 1 the automatic two-lane plans, 2 every plan with such a pass.
    static const int twf_t_mode = [] { const char* e = getenv("RONK_TWF_T"); return e ? atoi(e) : 0; }();
    bool wl_half_ = false;
    const bool twf_t = tile_wl_wanted(3, &wl_half_) && (twf_t_mode == 2 || (twf_t_mode == 1 && in_flight == 2 && tile_log2_columns < 0 && max_logc == 2));
    rc = pl->fwd.compile(build_plan((int)log2n, batch, false, max_logc, twf_max_log, three_from, auto_tiles, split_ka, hf, twf_t));
    if (!rc) rc = pl->inv.compile(build_plan((int)log2n, batch, true, max_logc, twf_max_log, three_from, auto_tiles, split_ka, hf, twf_t));
    for (auto& ps : pl->fwd.pd.passes)  // grid must fit the launch API
      if (!rc && (u64)ps.args.tiles * ps.args.nb1 * ps.args.nb2 > 0x7FFFFFFFull) rc = RONK_ERR_UNSUPPORTED;
    // the tables were uploaded by null-stream copies; the plan may be used from any (non-blocking) stream the moment this
    // returns, so the null stream is drained here once (a legacy copy may return before its DMA has finished)
    if (!rc && hipStreamSynchronize(0) != hipSuccess) rc = RONK_ERR_HIP;
  } else {
    pl->w_f = h_powmod(pl->g, (p - 1) / n, p);
    pl->w_i = h_powmod(pl->w_f, p - 2, p);                     // root.inverse().unwrap(), mod.rs:433
    pl->n_inv = h_powmod(n % p, p - 2, p);                     // F::from(D).inverse().unwrap(), mod.rs:442
    if (n % p == 0) { ronk_plan_destroy(pl); return RONK_ERR_ZERO_INVERSE; }
    if (log2n >= 1) {
      size_t half = n / 2;
      hipError_t e = hipMalloc((void**)&pl->d_wtab_f, half * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&pl->d_wtab_i, half * 8);
      if (e != hipSuccess) { ronk_plan_destroy(pl); return hip_fail(e, "hipMalloc"); }
      FIELD_DISPATCH(pl->field, {
        hipLaunchKernelGGL((power_table_kernel<decltype(ops)>), dim3(grid_for(half)), dim3(256), 0, 0, ops, pl->w_f,
                           pl->d_wtab_f, half);
        hipLaunchKernelGGL((power_table_kernel<decltype(ops)>), dim3(grid_for(half)), dim3(256), 0, 0, ops, pl->w_i,
                           pl->d_wtab_i, half);
      });
      if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = RONK_ERR_HIP;
    }
  }
  if (!rc) {
    hipError_t e = hipMalloc((void**)&pl->d_tmp, n * batch * 8);
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(scratch)");
  }
  // the side lane exists for multi-pass tile plans only (single-pass plans have nothing to overlap with themselves)
  if (!rc && in_flight == 2 && pl->fast && pl->fwd.pd.passes.size() >= 2) {
    hipError_t e = hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) rc = hip_fail(e, "side stream");
    else pl->in_flight = 2;
  }
  if (rc) { ronk_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}

extern "C" int ronk_plan_num_passes(const ronk_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  return pl->fast ? (int)pl->fwd.pd.passes.size() : (int)pl->log2n + 1;
}

// generic power-of-two transform: bit-reversal copy + log2(n) radix-2 DIT stages
static int generic_transform(ronk_plan* pl, bool inverse, const u64* in, u64* out, hipStream_t s) {
  const size_t total = pl->n * pl->batch;
  u64* work = out;
  if (in == out) work = pl->d_tmp;  // bit-reversal is not in-place safe
  FIELD_DISPATCH(pl->field, {
    hipLaunchKernelGGL((bitrev_copy_kernel<decltype(ops)>), dim3(grid_for(total)), dim3(256), 0, s, ops, in, work,
                       (int)pl->log2n, total);
    for (int st = 0; st < (int)pl->log2n; st++) {
      const bool last = st == (int)pl->log2n - 1;
      hipLaunchKernelGGL((radix2_stage_kernel<decltype(ops)>), dim3(grid_for(total / 2)), dim3(256), 0, s, ops, work,
                         inverse ? pl->d_wtab_i : pl->d_wtab_f, (int)pl->log2n, st, total / 2,
                         (inverse && last) ? pl->n_inv : (u64)1);
    }
  });
  HIPCHK(hipGetLastError());
  if (work != out) HIPCHK(hipMemcpyAsync(out, work, total * 8, hipMemcpyDeviceToDevice, s));
  return RONK_OK;
}

// order `s` behind the previous user of the plan's scratch (see ronk_plan::scratch_ev); caller holds stream_mu.
// While a plan stays on ONE stream nothing is recorded: an event per call is a marker packet in the stream between the
// launches, measured at +3 us per forward + inverse pair of a 2^16 transform (21.5 -> 27.4 us, round 4) -- a quarter of
// BASELINE config 2.  The first call that arrives on a second stream therefore finds no event behind the previous call, and
// the previous stream's HANDLE cannot be used to place one (the caller may have destroyed it: recording on a dead handle
// crashes -- tests/test_gpu_parity.py::test_plan_seen_on_several_streams does exactly that).  What is left is to drain the
// plan's DEVICE once (not the caller's current one); from then on every call leaves its event.  Not reached for a
// capturing stream (transform_dev skips the guard there: a captured graph owns its plan).
static void scratch_acquire(ronk_plan* pl, hipStream_t s) {
  if (!pl->scratch_used || pl->scratch_stream == s) return;
  if (!pl->scratch_multi || !pl->scratch_ev_valid || hipStreamWaitEvent(s, pl->scratch_ev, 0) != hipSuccess) {
    (void)hipGetLastError();
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != pl->device) (void)hipSetDevice(pl->device);
    (void)hipDeviceSynchronize();
    if (cur >= 0 && cur != pl->device) (void)hipSetDevice(cur);
  }
  pl->scratch_multi = true;
}
static void scratch_release(ronk_plan* pl, hipStream_t s) {
  pl->scratch_stream = s; pl->scratch_used = true;
  if (!pl->scratch_multi) return;
  hipError_t e = hipSuccess;
  if (!pl->scratch_ev) e = hipEventCreateWithFlags(&pl->scratch_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(pl->scratch_ev, s);
  pl->scratch_ev_valid = e == hipSuccess;
  if (e != hipSuccess) (void)hipGetLastError();   // the next cross-stream acquire drains the device instead
}

int transform_dev(ronk_plan* pl, bool inverse, const u64* in, const u64* in2, u64* out, hipStream_t s, u64 in_valid,
                  u64 out_valid, u64 in_poly_stride, u64 in_valid1, u64* tmp_override) {
  if (!pl || !in || !out) return RONK_ERR_INVALID;
  // The plan's scratch buffer is shared by every call on the plan.  Calls on ONE stream are ordered by the stream; a call
  // that arrives on a different stream than the previous one waits for the event that call left behind (scratch_acquire),
  // so two streams can never interleave inside the scratch.  The lock is held across the enqueue: two host threads
  // cannot interleave their launches either.  Costs nothing while a plan stays on one stream.
  // (a capturing stream cannot wait for an event recorded outside its capture: a captured graph owns its plan)
  const CompiledPlan* cp = pl->fast ? &(inverse ? pl->inv : pl->fwd) : nullptr;
  u64* const tmp = tmp_override ? tmp_override : pl->d_tmp;
  const bool uses_tmp = tmp && (cp ? cp->pd.needs_tmp : true);
  std::unique_lock<std::mutex> lk(pl->stream_mu, std::defer_lock);
  bool guard = false;
  if (uses_tmp && !tmp_override) {
    lk.lock();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    guard = cap == hipStreamCaptureStatusNone;
    if (guard) scratch_acquire(pl, s);
  }
  int rc;
  if (cp) {
    // in == out is safe: a single-pass plan rewrites exactly the tile it read; multi-pass plans
    // read BUF_IN only in pass 1 and write BUF_OUT only in the last pass.
    if (pl->in_flight == 2 && cp->sliceable() && !tmp_override) {
      // two halves of the batch, the second on the side stream: fork after everything already queued on `s`, join before
      // anything queued later (stream order as seen by the caller is unchanged)
      const u32 h = (u32)((pl->batch + 1) / 2), h2 = (u32)(pl->batch - h);
      hipError_t e = hipEventRecord(pl->ev_fork, s);
      if (e == hipSuccess) e = hipStreamWaitEvent(pl->side, pl->ev_fork, 0);
      if (e != hipSuccess) return hip_fail(e, "fork");
      rc = cp->run_slice(h, h2, in, in2, out, tmp, pl->side, in_valid, out_valid, in_poly_stride, in_valid1);
      if (!rc) rc = cp->run_slice(0, h, in, in2, out, tmp, s, in_valid, out_valid, in_poly_stride, in_valid1);
      e = hipEventRecord(pl->ev_join, pl->side);
      if (e == hipSuccess) e = hipStreamWaitEvent(s, pl->ev_join, 0);
      if (e != hipSuccess) { (void)hipDeviceSynchronize(); if (!rc) rc = hip_fail(e, "join"); }
    } else {
      rc = cp->run(in, in2, out, tmp, s, in_valid, out_valid, in_poly_stride, 0, in_valid1);
    }
  } else if (in2 || in_valid != ~(u64)0 || out_valid != ~(u64)0 || in_poly_stride) {
    rc = RONK_ERR_UNSUPPORTED;
  } else if (pl->log2n == 0) {  // n = 1: fft/ifft are the identity (the recursion returns at n <= 1)
    rc = RONK_OK;
    if (in != out) {
      hipError_t e = hipMemcpyAsync(out, in, pl->batch * 8, hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync");
    }
  } else {
    rc = generic_transform(pl, inverse, in, out, s);
  }
  if (guard) scratch_release(pl, s);
  return rc;
}

// K independent [batch][n] arrays in ONE call (a batch-1 plan: K polynomials that live in unrelated buffers): with
// in_flight = 2 the arrays go round-robin over the caller's stream and the plan's side stream (which has its own scratch),
// each lane running its transforms back to back; fork / join on `stream` as above.
static int many_dev(ronk_plan* pl, bool inverse, const uint64_t* const* d_in, uint64_t* const* d_out, size_t count, hipStream_t s) {
  if (!pl || !d_in || !d_out) return RONK_ERR_INVALID;
  if (count == 0) return RONK_OK;
  for (size_t i = 0; i < count; i++) if (!d_in[i] || !d_out[i]) return RONK_ERR_INVALID;
  const CompiledPlan* cp = pl->fast ? &(inverse ? pl->inv : pl->fwd) : nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  if (pl->in_flight != 2 || !cp || count < 2 || cap != hipStreamCaptureStatusNone) {
    for (size_t i = 0; i < count; i++) RCHK(transform_dev(pl, inverse, d_in[i], nullptr, d_out[i], s));
    return RONK_OK;
  }
  std::lock_guard<std::mutex> lk(pl->stream_mu);
  if (!pl->d_tmp2) {   // the side lane's scratch lives on the PLAN's device, whatever the caller's current device is
    int cur = -1;
    HIPCHK(hipGetDevice(&cur));
    if (cur != pl->device) HIPCHK(hipSetDevice(pl->device));
    hipError_t em = hipMalloc((void**)&pl->d_tmp2, pl->n * pl->batch * 8);
    if (cur != pl->device) (void)hipSetDevice(cur);
    if (em != hipSuccess) return hip_fail(em, "hipMalloc(side scratch)");
  }
  scratch_acquire(pl, s);
  HIPCHK(hipEventRecord(pl->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(pl->side, pl->ev_fork, 0));
  int rc = RONK_OK;
  // (Round 6 measured a software prefetch of the array a lane transforms next -- one 4-byte load per 128-byte line as the last
  //  instructions of a pass -- and removed it: 7 % slower, the same bytes moved earlier cost more than the warm input saves;
  //  profiles/r06_occupancy_ab.txt.)
  for (size_t i = 0; i < count && !rc; i++) {
    const bool lane1 = (i & 1) != 0;
    rc = cp->run(d_in[i], nullptr, d_out[i], lane1 ? pl->d_tmp2 : pl->d_tmp, lane1 ? pl->side : s);
  }
  hipError_t e = hipEventRecord(pl->ev_join, pl->side);
  if (e == hipSuccess) e = hipStreamWaitEvent(s, pl->ev_join, 0);
  if (e != hipSuccess) { (void)hipDeviceSynchronize(); if (!rc) rc = hip_fail(e, "join"); }
  scratch_release(pl, s);
  return rc;
}
extern "C" int ronk_ntt_forward_many_dev(ronk_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out, size_t count,
                                         void* st) {
  return many_dev(pl, false, d_in, d_out, count, (hipStream_t)st);
}
extern "C" int ronk_ntt_inverse_many_dev(ronk_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out, size_t count,
                                         void* st) {
  return many_dev(pl, true, d_in, d_out, count, (hipStream_t)st);
}
// Batched Message::encode::<N> on device (codes/reed_solomon.rs:42-52): the y-coordinates of `batch` codewords,
// ys[b][i] = message_b(omega_N^i), from compact messages msgs[b][0..k).  Multi-pass Goldilocks plans read the
// messages in place with implicit zero padding; other plans pad into d_ys first and transform in place.
extern "C" int ronk_rs_encode_batch_dev(ronk_plan* pl, const uint64_t* d_msgs, size_t k, uint64_t* d_ys, void* st) {
  if (!pl || !d_msgs || !d_ys || k == 0) return RONK_ERR_INVALID;
  if (k > pl->n) return RONK_ERR_INDEX;   // assert_ge::<N, K>()
  hipStream_t s = (hipStream_t)st;
  if (pl->fast && pl->fwd.pd.passes.size() > 1)
    return transform_dev(pl, false, d_msgs, nullptr, d_ys, s, (u64)k, ~(u64)0, (u64)k);
  const size_t total = pl->n * pl->batch;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(total)), dim3(256), 0, s, d_msgs, k, d_ys, (size_t)pl->n, total);
  HIPCHK(hipGetLastError());
  return transform_dev(pl, false, d_ys, nullptr, d_ys, s);
}
// Low-degree extension ("iNTT -> zero-pad -> NTT"): a batch of polynomials given by their values on the
// K-point domain {omega_K^i} -> their values on shift * {omega_N^i}, N >= K.  In the reference's terms:
// `Message::encode::<N>` of `lagrange_poly.ifft()` (src/polynomial/mod.rs:430-453, src/codes/reed_solomon.rs:42-52), with the
// coefficients scaled by shift^i first when a coset is asked for (shift = 1: the plain extension).
template <class Ops>
__global__ void __launch_bounds__(256) lde_coset_scale_kernel(Ops ops, u64* __restrict__ c, size_t k, size_t total, u64 shift) {
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x)
    c[t] = ops.mul(c[t], ops.pow(shift, (u64)(t % k)));
}
extern "C" int ronk_lde_batch_dev(ronk_plan* plan_k, ronk_plan* plan_n, const uint64_t* d_evals, uint64_t* d_coeffs,
                                  uint64_t* d_out, uint64_t coset_shift, void* st) {
  if (!plan_k || !plan_n || !d_evals || !d_coeffs || !d_out) return RONK_ERR_INVALID;
  if (plan_k->p != plan_n->p || plan_k->batch != plan_n->batch || plan_k->n > plan_n->n) return RONK_ERR_INVALID;
  hipStream_t s = (hipStream_t)st;
  RCHK(transform_dev(plan_k, true, d_evals, nullptr, d_coeffs, s));            // [batch][K] coefficients
  coset_shift %= plan_k->p;
  if (coset_shift != 1) {   // c_i <- c_i * shift^i (any field: the element-wise operators of field_kernels.h)
    if (coset_shift == 0) return RONK_ERR_UNSUPPORTED;
    const size_t total = plan_k->n * plan_k->batch;
    FIELD_DISPATCH(plan_k->field, { hipLaunchKernelGGL((lde_coset_scale_kernel<decltype(ops)>), dim3(grid_for(total)), dim3(256), 0, s, ops,
                                                       d_coeffs, (size_t)plan_k->n, total, coset_shift); });
    HIPCHK(hipGetLastError());
  }
  return ronk_rs_encode_batch_dev(plan_n, d_coeffs, plan_k->n, d_out, st);
}
extern "C" int ronk_ntt_forward_dev(ronk_plan* pl, const uint64_t* in, uint64_t* out, void* st) {
  return transform_dev(pl, false, in, nullptr, out, (hipStream_t)st);
}
extern "C" int ronk_ntt_inverse_dev(ronk_plan* pl, const uint64_t* in, uint64_t* out, void* st) {
  return transform_dev(pl, true, in, nullptr, out, (hipStream_t)st);
}

// Page-locking of the caller's buffers for the duration of a pipelined host call.  Registrations are shared: two threads that
// transform from the SAME input buffer (on different plans) hold one registration between them (reference count), memory
// that is page-locked already -- hipHostMalloc, the caller's own hipHostRegister -- is left alone, and only a range that
// OVERLAPS a foreign registration partially is refused by the runtime: the copies of that call then run synchronously
// (correct, un-overlapped).  The lock covers the runtime calls only.
static std::mutex g_pin_mu;
static std::map<uintptr_t, std::pair<size_t, int>> g_pins;   // base -> (bytes, users), disjoint intervals
// Returns the base of OUR registration that now covers [p, p + bytes) (the caller hands it back to pin_release), or 0 when
// the copies of this call have to run synchronously / the memory is page-locked by somebody else already.  Lookup is by
// INTERVAL: a sub-range of a range this library registered for another thread shares that registration (reference count on
// the owning entry), so the owner's release cannot unregister memory a second thread's copies are still reading; a range
// that only partially overlaps one of ours is left alone (synchronous copies).
static uintptr_t pin_acquire(const void* p, size_t bytes) {
  const uintptr_t a = (uintptr_t)p, b = a + bytes;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pins.upper_bound(a);                 // first entry that starts after a
  if (it != g_pins.begin()) {
    auto prev = std::prev(it);
    const uintptr_t pa = prev->first, pb = pa + prev->second.first;
    if (a >= pa && b <= pb) { prev->second.second++; return pa; }   // contained in one of ours
    if (a < pb) return 0;                                           // partial overlap with ours
  }
  if (it != g_pins.end() && it->first < b) return 0;                // runs into the next one of ours
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) return 0;   // page-locked by the caller
  (void)hipGetLastError();
  if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 0; }
  g_pins[a] = {bytes, 1};
  return a;
}
static void pin_release(uintptr_t base) {
  if (!base) return;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pins.find(base);
  if (it == g_pins.end() || --it->second.second > 0) return;
  (void)hipHostUnregister((void*)base);
  g_pins.erase(it);
}

static int ensure_stage(ronk_plan* pl) {
  const size_t bytes = pl->n * pl->batch * 8;
  if (!pl->d_stage_in) HIPCHK(hipMalloc((void**)&pl->d_stage_in, bytes));
  if (!pl->d_stage_out) HIPCHK(hipMalloc((void**)&pl->d_stage_out, bytes));
  return RONK_OK;
}
// Host-pointer transforms.  One polynomial: upload, transform, download -- the link is used one way at a time and the
// transform itself (0.06 ms at 2^22) hides nowhere: 2 x 32 MiB at the measured 56 GB/s (pageable and pinned host memory copy
// at the same rate on this platform, tools/pcie_probe.hip) is 1.19 ms of the 1.25.  A BATCH is pipelined over its
// polynomials on three internal streams -- upload of slice i+1, transform of slice i and download of slice i-1 overlap
// (PCIe is full duplex: 48 GB/s each way when both directions are busy), so the cost per polynomial drops from
// upload + download to the slower of the two.  Slices are whole polynomials, at least 4 MiB, at most 64 per call.
static int transform_host_pipelined(ronk_plan* pl, bool inverse, const u64* in, u64* out) {
  const CompiledPlan& cp = inverse ? pl->inv : pl->fwd;
  const u64 n = pl->n;
  u64 per = ((u64)4 << 20) / (n * 8);                      // polynomials per slice
  if (per < 1) per = 1;
  if ((pl->batch + per - 1) / per > 64) per = (pl->batch + 63) / 64;
  const u32 nsl = (u32)((pl->batch + per - 1) / per);
  if (!pl->st_h2d) HIPCHK(hipStreamCreateWithFlags(&pl->st_h2d, hipStreamNonBlocking));
  if (!pl->st_exec) HIPCHK(hipStreamCreateWithFlags(&pl->st_exec, hipStreamNonBlocking));
  if (!pl->st_d2h) HIPCHK(hipStreamCreateWithFlags(&pl->st_d2h, hipStreamNonBlocking));
  while (pl->st_ev.size() < 2 * (size_t)nsl) {
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    pl->st_ev.push_back(e);
  }
  HIPCHK(hipStreamSynchronize(0));                          // earlier null-stream users of the staging buffers / the scratch
  // Asynchronous copies need page-locked host memory: a hipMemcpyAsync from pageable memory returns only when the copy
  // is done (measured: the three streams then run strictly one after the other, 1.30 ms per 2^22 polynomial).  The caller's
  // buffers are registered for the duration of the call (cheap on this platform: tools/pcie_probe.hip) and released
  // again; buffers that are already page-locked (hipHostMalloc, an earlier registration) are used as they are, and if a
  // registration is refused the copies simply fall back to their synchronous behaviour.
  const size_t total_bytes = (size_t)(pl->batch * n * 8);
  const uintptr_t reg_in = pin_acquire(in, total_bytes);
  const uintptr_t reg_out = pin_acquire(out, total_bytes);
  int rc = RONK_OK;
  for (u32 i = 0; i < nsl && !rc; i++) {
    const u64 b0 = (u64)i * per, cnt = pl->batch - b0 < per ? pl->batch - b0 : per;
    const size_t off = (size_t)(b0 * n), bytes = (size_t)(cnt * n * 8);
    hipEvent_t ev_in = pl->st_ev[2 * i], ev_done = pl->st_ev[2 * i + 1];
    hipError_t e = hipMemcpyAsync(pl->d_stage_in + off, in + off, bytes, hipMemcpyHostToDevice, pl->st_h2d);
    if (e == hipSuccess) e = hipEventRecord(ev_in, pl->st_h2d);
    if (e == hipSuccess) e = hipStreamWaitEvent(pl->st_exec, ev_in, 0);
    if (e != hipSuccess) { rc = hip_fail(e, "upload"); break; }
    rc = cp.run_slice((u32)b0, (u32)cnt, pl->d_stage_in, nullptr, pl->d_stage_out, pl->d_tmp, pl->st_exec);
    if (rc) break;
    e = hipEventRecord(ev_done, pl->st_exec);
    if (e == hipSuccess) e = hipStreamWaitEvent(pl->st_d2h, ev_done, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(out + off, pl->d_stage_out + off, bytes, hipMemcpyDeviceToHost, pl->st_d2h);
    if (e != hipSuccess) rc = hip_fail(e, "download");
  }
  // drain all three before returning (also on errors: nothing may still touch the caller's buffers)
  hipError_t e1 = hipStreamSynchronize(pl->st_h2d), e2 = hipStreamSynchronize(pl->st_exec), e3 = hipStreamSynchronize(pl->st_d2h);
  if (!rc && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess))
    rc = hip_fail(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3, "pipeline drain");
  pin_release(reg_in);
  pin_release(reg_out);
  return rc;
}

static int transform_host(ronk_plan* pl, bool inverse, const u64* in, u64* out) {
  if (!pl || !in || !out) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  HIPCHK(hipSetDevice(pl->device));
  RCHK(ensure_stage(pl));
  const size_t bytes = pl->n * pl->batch * 8;
  static const bool no_pipe = getenv("RONK_HOST_NO_PIPELINE") != nullptr;   // A/B
  if (!no_pipe && pl->fast && (inverse ? pl->inv : pl->fwd).sliceable() && bytes >= ((size_t)8 << 20)) {
    // the pipeline owns the scratch for the duration of the call: order it against other streams' users like any transform
    std::lock_guard<std::mutex> lk2(pl->stream_mu);
    if (pl->scratch_used) (void)hipDeviceSynchronize();
    const int rc = transform_host_pipelined(pl, inverse, in, out);
    pl->scratch_used = false;                               // everything has drained
    return rc;
  }
  HIPCHK(hipMemcpy(pl->d_stage_in, in, bytes, hipMemcpyHostToDevice));
  RCHK(transform_dev(pl, inverse, pl->d_stage_in, nullptr, pl->d_stage_out, 0));
  HIPCHK(hipMemcpy(out, pl->d_stage_out, bytes, hipMemcpyDeviceToHost));
  return RONK_OK;
}

static int lagrange_nodes_dev(const FieldCtx& f, u64 w, u64* d_nodes, size_t n, hipStream_t s) {
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((power_table_kernel<decltype(ops)>), dim3(grid_for(n)), dim3(256), 0, s, ops, w,
                                        d_nodes, n); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}
extern "C" int ronk_lagrange_nodes(uint64_t p, uint64_t g, uint64_t* nodes, size_t n) {
  if (!nodes || n == 0) return RONK_ERR_INVALID;
  u64 w;
  RCHK(ronk_root_of_unity(p, g % p, n, &w));   // assert_eq!((F::ORDER - 1) % n, 0), mod.rs:361
  RCHK(need_device());
  FieldCtx f;
  RCHK(make_field(p, &f));
  DevBuf d;
  RCHK(d.alloc(n * 8));
  RCHK(lagrange_nodes_dev(f, w, d.u(), n, 0));
  HIPCHK(hipMemcpy(nodes, d.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

extern "C" int ronk_ntt_forward(ronk_plan* pl, const uint64_t* in, uint64_t* out, uint64_t* nodes) {
  RCHK(transform_host(pl, false, in, out));
  if (nodes) RCHK(ronk_lagrange_nodes(pl->p, pl->g, nodes, pl->n));
  return RONK_OK;
}
extern "C" int ronk_ntt_inverse(ronk_plan* pl, const uint64_t* in, uint64_t* out) {
  return transform_host(pl, true, in, out);
}

extern "C" int ronk_plan_time_passes(ronk_plan* pl, const uint64_t* d_in, uint64_t* d_out, int inverse, int iters,
                                     float* ms, void* st) {
  if (!pl || !d_in || !d_out || !ms || iters < 1) return RONK_ERR_INVALID;
  if (!pl->fast) return RONK_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)st;
  const CompiledPlan& cp = inverse ? pl->inv : pl->fwd;
  const size_t np = cp.pd.passes.size();
  std::vector<hipEvent_t> ev(np + 1);
  for (auto& e : ev) HIPCHK(hipEventCreate(&e));
  std::vector<double> acc(np, 0.0);
  for (int it = 0; it < iters; it++) {
    HIPCHK(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < np; i++) {
      RCHK(cp.launch(i, d_in, nullptr, d_out, pl->d_tmp, s));
      HIPCHK(hipEventRecord(ev[i + 1], s));
    }
    HIPCHK(hipEventSynchronize(ev[np]));
    for (size_t i = 0; i < np; i++) {
      float t = 0;
      HIPCHK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      acc[i] += t;
    }
  }
  for (size_t i = 0; i < np; i++) ms[i] = (float)(acc[i] / iters);
  for (auto& e : ev) (void)hipEventDestroy(e);
  return RONK_OK;
}

// ------------------------------------------------------------------------------ plan cache
// One-shot entry points (ronk_fft/ronk_ifft/ronk_dft/ronk_poly_mul*) reuse plans -- twiddle tables and
// scratch stay resident in HBM -- through a small LRU cache guarded by one lock (the reference is
// stateless; `cargo test` calls in from many threads).  An entry also owns two padded operand
// buffers for the multiply; an event orders successive uses of an entry across streams.

struct CacheEntry {
  ronk_plan* pli = nullptr;          // the multiply's own inverse plan when it wants other planner options than `pl` (lazy)
  ronk_plan* pl = nullptr;
  u64 *fa = nullptr, *fb = nullptr;  // poly_mul operands, n elements each (lazy)
  ronk_plan* pl2 = nullptr;          // the same size with batch 2: both operands of a multiply in ONE pair of launches (lazy)
  ronk_plan* plf = nullptr;          // the fused multiply's inverse plan: the pair plan's tile width (4 columns) (lazy)
  u64* fab = nullptr;                // its output: [2][n]
  hipEvent_t done = nullptr;
  uint64_t stamp = 0;
  int pins = 0;                      // users outside g_cache_mu (never evicted while pinned)
};
static std::mutex g_cache_mu;
static std::vector<CacheEntry*> g_cache;   // heap entries: addresses stay valid while the vector changes
static uint64_t g_cache_clock = 0;

// Cross-stream guard of a cache entry's scratch buffers: every use waits for the event the previous use left behind and leaves
// its own.  A CAPTURING stream can neither wait for an event recorded outside its capture nor lend the entry an event recorded
// inside it (it would never complete for anybody else): a captured call skips both -- the graph then owns the entry's
// scratch whenever it is replayed, like a captured transform owns its plan (transform_dev), and replays must not overlap other
// users of the same product / transform size (include/ronk_ntt.h).
static bool entry_capturing(hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cap != hipStreamCaptureStatusNone;
}
static hipError_t entry_wait(hipStream_t s, hipEvent_t ev) { return entry_capturing(s) ? hipSuccess : hipStreamWaitEvent(s, ev, 0); }
static hipError_t entry_record(hipEvent_t ev, hipStream_t s) { return entry_capturing(s) ? hipSuccess : hipEventRecord(ev, s); }

static void cache_entry_free(CacheEntry* e) {
  if (e->pl) ronk_plan_destroy(e->pl);
  if (e->fa) (void)hipFree(e->fa);
  if (e->fb) (void)hipFree(e->fb);
  if (e->pl2) ronk_plan_destroy(e->pl2);
  if (e->pli) ronk_plan_destroy(e->pli);
  if (e->plf) ronk_plan_destroy(e->plf);
  if (e->fab) (void)hipFree(e->fab);
  if (e->done) (void)hipEventDestroy(e->done);
  delete e;
}
// caller holds g_cache_mu
static int cache_get(u64 p, u64 g, u32 log2n, CacheEntry** out) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  for (CacheEntry* e : g_cache)
    if (e->pl && e->pl->p == p && e->pl->g == g % p && e->pl->log2n == log2n && e->pl->batch == 1 && e->pl->device == dev) {
      e->stamp = ++g_cache_clock;
      *out = e;
      return RONK_OK;
    }
  // 24 entries: a Newton division (ronk_callers.hip) walks a ladder of ~2 log2(n) product sizes -- 8 entries made every
  // large division rebuild all of its plans
  if (g_cache.size() >= 24) {  // evict the least recently used entry nobody holds (its work must have drained)
    size_t lru = g_cache.size();
    for (size_t i = 0; i < g_cache.size(); i++)
      if (g_cache[i]->pins == 0 && (lru == g_cache.size() || g_cache[i]->stamp < g_cache[lru]->stamp)) lru = i;
    if (lru < g_cache.size()) {
      if (g_cache[lru]->done) (void)hipEventSynchronize(g_cache[lru]->done);
      cache_entry_free(g_cache[lru]);
      g_cache.erase(g_cache.begin() + lru);
    }
  }
  CacheEntry* e = new CacheEntry();
  int rc = ronk_plan_create(&e->pl, p, g, log2n, 1, -1);
  if (rc) { cache_entry_free(e); return rc; }
  hipError_t he = hipEventCreateWithFlags(&e->done, hipEventDisableTiming);
  if (he != hipSuccess) { cache_entry_free(e); return hip_fail(he, "hipEventCreate"); }
  e->stamp = ++g_cache_clock;
  g_cache.push_back(e);
  *out = e;
  return RONK_OK;
}

// Polynomial::fft / ifft one-shot forms (polynomial/mod.rs:273-292, :430-453) on host pointers
static int fft_oneshot(bool inverse, u64 p, u64 g, const u64* in, u64* out, u64* nodes, size_t n) {
  if (!in || !out || n == 0) return RONK_ERR_INVALID;
  if (!is_pow2(n)) return RONK_ERR_NOT_POW2;  // `[(); D.is_power_of_two() as usize - 1]:`
  RCHK(ronk_check_prime(p));
  if (p < 2) return RONK_ERR_INVALID;
  if ((p - 1) % n != 0) return RONK_ERR_NO_ROOT;
  RCHK(need_device());
  // The cache lock covers the lookup only: the entry is pinned, the PCIe copies and the transform run under the plan's
  // own lock (transform_host), so threads using different sizes (`cargo test`) do not serialise on one mutex.
  CacheEntry* e = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    RCHK(cache_get(p, g, (u32)ilog2(n), &e));
    e->pins++;
  }
  const int rc = inverse ? ronk_ntt_inverse(e->pl, in, out) : ronk_ntt_forward(e->pl, in, out, nullptr);
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    e->pins--;
  }
  RCHK(rc);
  if (nodes) RCHK(ronk_lagrange_nodes(p, g, nodes, n));
  return RONK_OK;
}
extern "C" int ronk_fft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, uint64_t* nodes, size_t n) {
  return fft_oneshot(false, p, g, in, out, nodes, n);
}
extern "C" int ronk_ifft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  return fft_oneshot(true, p, g, in, out, nullptr, n);
}

// ------------------------------------------------------------------------------ dft (any n | p-1)
static int conv_dev(u64 p, u64 g, int k, const u64* d_a, size_t d, const u64* d_b, size_t d2, u64* d_out, size_t out_len,
                    hipStream_t s);

// Bluestein's chirp-z for n that is not a power of two (Goldilocks): with C(m) = m(m-1)/2, j*k = C(j+k) - C(j) - C(k), so
//   X_k = w^-C(k) * sum_j (x_j w^-C(j)) * w^C(j+k)
// is a correlation, computed as ONE cyclic convolution of size 2^ceil(log2(2n-1)) on the fast NTT path.  Only w = omega_n
// itself is needed (no square root of it).  Same values as Polynomial::dft (polynomial/mod.rs:240-258), O(n log n).
static int bluestein_dev(u64 p, u64 g, u64 w, const u64* d_x, u64* d_out, size_t n, hipStream_t s) {
  FieldCtx f;
  RCHK(make_field(p, &f));
  const size_t lb = 2 * n - 1;
  int k = ilog2(lb);
  if (k < 4) k = 4;
  DevBuf dT, da, db, dc;
  RCHK(dT.alloc(n * 8)); RCHK(da.alloc(n * 8)); RCHK(db.alloc(lb * 8)); RCHK(dc.alloc(lb * 8));
  RCHK(lagrange_nodes_dev(f, w, dT.u(), n, s));
  hipLaunchKernelGGL(bluestein_pre_kernel, dim3(grid_for(lb)), dim3(256), 0, s, d_x, dT.u(), n, da.u(), db.u());
  HIPCHK(hipGetLastError());
  RCHK(conv_dev(p, g, k, da.u(), n, db.u(), lb, dc.u(), lb, s));   // cyclic wrap-around only reaches indices < n-1
  hipLaunchKernelGGL(bluestein_post_kernel, dim3(grid_for(n)), dim3(256), 0, s, dc.u(), dT.u(), n, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));   // the temporaries above are freed on return
  return RONK_OK;
}

// Polynomial::dft on device-resident data (any n | p-1): powers of two over Goldilocks run the cached NTT plan, other
// n >= 512 over Goldilocks Bluestein on the NTT path (synchronises `stream` once: its temporaries), the rest the direct
// O(n^2) kernel (n <= 2^16).  d_in == d_out allowed for the power-of-two path only.
extern "C" int ronk_dft_dev(uint64_t p, uint64_t g, const uint64_t* d_in, uint64_t* d_out, size_t n, void* st) {
  if (!d_in || !d_out || n == 0) return RONK_ERR_INVALID;
  u64 w;
  RCHK(ronk_root_of_unity(p, g % p, n, &w));
  RCHK(need_device());
  hipStream_t s = (hipStream_t)st;
  const bool gl = p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G;
  // (a tiled plan implies omega_n of order exactly n, so Polynomial::fft == Polynomial::dft there)
  if (is_pow2(n) && n >= 16 && n <= ((size_t)1 << 30) && ronk_check_prime(p) == RONK_OK && tiled_plan_exists(p, g, ilog2(n))) {
    CacheEntry* e = nullptr;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    RCHK(cache_get(p, g, (u32)ilog2(n), &e));
    HIPCHK(entry_wait(s, e->done));
    RCHK(transform_dev(e->pl, false, d_in, nullptr, d_out, s));
    HIPCHK(entry_record(e->done, s));
    return RONK_OK;
  }
  if (d_in == d_out) return RONK_ERR_INVALID;
  const bool chirp = gl && n >= 512 && n <= ((size_t)1 << 29);
  if (chirp) return bluestein_dev(p, g % p, w, d_in, d_out, n, s);
  if (n > ((size_t)1 << 16)) return RONK_ERR_UNSUPPORTED;
  FieldCtx f;
  RCHK(make_field(p, &f));
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((dft_naive_kernel<decltype(ops)>), dim3((u32)((n + 255) / 256)), dim3(256), 0, s,
                                        ops, d_in, d_out, n, w); });
  HIPCHK(hipGetLastError());
  return RONK_OK;
}

extern "C" int ronk_dft(uint64_t p, uint64_t g, const uint64_t* in, uint64_t* out, size_t n) {
  if (!in || !out || n == 0) return RONK_ERR_INVALID;
  u64 w;
  RCHK(ronk_root_of_unity(p, g % p, n, &w));
  RCHK(need_device());
  if (is_pow2(n) && n >= 16 && n <= ((size_t)1 << 30) && ronk_check_prime(p) == RONK_OK && tiled_plan_exists(p, g, ilog2(n)))
    return fft_oneshot(false, p, g, in, out, nullptr, n);
  const bool chirp = p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G && n >= 512 && n <= ((size_t)1 << 29);
  if (!chirp && n > ((size_t)1 << 16)) return RONK_ERR_UNSUPPORTED;
  FieldCtx f;
  RCHK(make_field(p, &f));
  DevBuf di, dout;
  RCHK(di.alloc(n * 8)); RCHK(dout.alloc(n * 8));
  HIPCHK(hipMemcpy(di.p, in, n * 8, hipMemcpyHostToDevice));
  if (chirp) {
    RCHK(bluestein_dev(p, g % p, w, di.u(), dout.u(), n, 0));
    HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    return RONK_OK;
  }
  FIELD_DISPATCH(f, { hipLaunchKernelGGL((dft_naive_kernel<decltype(ops)>), dim3((u32)((n + 255) / 256)), dim3(256), 0, 0,
                                        ops, di.u(), dout.u(), n, w); });
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

// ------------------------------------------------------------------------------ polynomial multiply
// Goldilocks: pad both to N = 2^k >= d + d2 - 1, NTT(a), then the inverse plan's first pass loads
// NTT(a) * NTT(b) (fused pointwise product) -- 3 transforms, 48*N algorithmic bytes.
extern "C" int ronk_poly_mul_dev(uint64_t p, uint64_t g, const uint64_t* d_a, size_t d, const uint64_t* d_b, size_t d2,
                                 uint64_t* d_out, void* st) {
  if (!d_a || !d_b || !d_out || d == 0 || d2 == 0) return RONK_ERR_INVALID;
  hipStream_t s = (hipStream_t)st;
  const size_t m = d + d2 - 1;
  FieldCtx f;
  RCHK(make_field(p, &f));
  // the NTT path: Goldilocks, or any prime that has a tiled plan of the product's size (N | p - 1, g a non-residue) --
  // the product does not depend on g, only the transform's existence does
  int k = ilog2(m);
  if (k < 4) k = 4;
  const bool fast = m > 64 && ((p == RONK_GOLDILOCKS_P && g % p == RONK_GOLDILOCKS_G) ||
                               (ronk_check_prime(p) == RONK_OK && tiled_plan_exists(p, g, k)));
  if (!fast) {
    if ((double)d * (double)d2 > 1.2e12) return RONK_ERR_UNSUPPORTED;
    FIELD_DISPATCH(f, { hipLaunchKernelGGL((poly_mul_schoolbook_kernel<decltype(ops)>), dim3(grid_for(m)), dim3(256), 0, s,
                                          ops, d_a, d, d_b, d2, d_out); });
    HIPCHK(hipGetLastError());
    return RONK_OK;
  }
  return conv_dev(p, g, k, d_a, d, d_b, d2, d_out, m, s);
}
// cyclic convolution of size N = 2^k of a (d entries) and b (d2 entries), the first out_len entries stored
static int conv_dev(u64 p, u64 g, int k, const u64* d_a, size_t d, const u64* d_b, size_t d2, u64* d_out, size_t out_len,
                    hipStream_t s) {
  const size_t N = (size_t)1 << k;
  const size_t m = out_len;
  RCHK(need_device());
  std::lock_guard<std::mutex> lk(g_cache_mu);
  CacheEntry* e = nullptr;
  RCHK(cache_get(p, g, (u32)k, &e));
  ronk_plan* pl = e->pl;
  static const bool no_pair = getenv("RONK_MUL_NO_PAIR") != nullptr;   // A/B: the two forward transforms one after the other
  if (pl->fast && pl->fwd.pd.passes.size() > 1 && !no_pair && d_a != d_b) {   // (a square passes the same buffer twice: stride 0 is the "unchanged" sentinel)
    // Both operands as ONE batch of two (batch stride = the distance between the caller's buffers, each with its own
    // zero-padding limit): 512 instead of 2 x 256 tiles per pass at 2^22, so the load / store phases of one operand's
    // tiles run under the arithmetic of the other's -- what two streams do for independent transforms.
    // (from 2^20 on with 4-column tiles, two workgroups per CU: the configuration that lets two transforms overlap)
    // (2^19 .. 2^22: that plan runs its two polynomials on two streams, ronk_plan::in_flight)
    // Inter-pass twiddles of the multiply's plans: A/B knobs RONK_MUL_FWD_TWF / RONK_MUL_INV_TWF (log2 of the largest full
    // matrix; default: see below)
    static const int fwd_twf = [] { const char* e_ = getenv("RONK_MUL_FWD_TWF"); return e_ ? atoi(e_) : -2; }();
    static const int inv_twf = [] { const char* e_ = getenv("RONK_MUL_INV_TWF"); return e_ ? atoi(e_) : -2; }();
    // measured (round 3, 2^22, same box: profiles/r03_mul_twf_ab.txt): matrix for both the forward pair and the inverse 167 us,
    // for either one alone or for neither 152 us (two 32 MiB matrices plus the operands do not stay cached together); at
    // 2^21 all four within 1 %.  Default: the forward pair keeps the two-level tables at 2^22, the inverse runs the cached
    // plan as it is.
    const int ftw = fwd_twf != -2 ? fwd_twf : (k == 22 ? 18 : -1), itw = inv_twf != -2 ? inv_twf : -1;
    static const bool fused23 = [] { const char* e_ = getenv("RONK_MUL_FUSED23"); return !e_ || atoi(e_) != 0; }();
    if (!e->pl2) {
      if (k >= 20) {
        // N = 2^23 (round 5): the pair as a TWO-pass plan (2^12 x 2^11, which the planner gives only to a single transform of
        // that size) so that its row pass -- 2^11 rows, 4-column tiles -- feeds the fused middle like at 2^21 / 2^22
        ronk_plan_opts o = RONK_PLAN_OPTS_DEFAULT;
        o.tile_log2_columns = 2;
        o.twiddle_matrix_log2_max = ftw;
        if (k == 23 && fused23) o.three_pass_from_log2 = 24;
        RCHK(ronk_plan_create_opts(&e->pl2, p, g, (u32)k, 2, pl->device, &o));
      } else {
        RCHK(ronk_plan_create(&e->pl2, p, g, (u32)k, 2, pl->device));
      }
    }
    if (itw >= 0 && !e->pli) RCHK(ronk_plan_create_tuned(&e->pli, p, g, (u32)k, 1, pl->device, -1, itw));
    const u64 stride = (u64)(d_b - d_a);   // element stride, modulo 2^64 (a negative distance wraps back in the address arithmetic)
    // Fused middle (ntt_mul.h): forward row pass of both operands + pointwise product + inverse column pass in ONE launch --
    // three launches per product, NTT(a) / NTT(b) never written.  Needs the same tile width on both sides (a dedicated
    // inverse plan with the pair plan's 4-column tiles) and two-pass plans of 2^10 / 2^11-row passes; fused for NTT sizes 2^21,
    // 2^22 and (round 5) 2^23 (2^20 measured slower, below).  RONK_MUL_FUSED=0: the
    // four-launch form (A/B); RONK_MUL_INV_TWF picks the inverse's twiddle form as before (default there: two-level tables).
    static const bool fused_on = [] { const char* e_ = getenv("RONK_MUL_FUSED"); return !e_ || atoi(e_) != 0; }();
    // Measured (round 4, same box, us per product): 2^22 158.6 -> 132.6, 2^21 88.0 -> 80.6, but 2^20 63.1 -> 68.1 -- there a
    // pass has 256 tiles of four wavefronts, one wavefront per SIMD, and three transforms in sequence inside a workgroup are
    // three times one wavefront's dependent instruction stream; the four-launch form spreads them over twice the tiles.
    // (instantiated for Goldilocks -- tile_kernels_mul.hip -- and for Montgomery primes -- tile_kernels_mont_mul.hip)
    if (fused_on && (k == 21 || k == 22 || (k == 23 && fused23))) {
      if (!e->plf) {
        // the inverse whose COLUMN pass has the rows of the pair plan's ROW pass: the balanced split for even k, the other
        // split of an odd one (2^21: pair plan 2^11 x 2^10, inverse 2^10 x 2^11)
        ronk_plan_opts o = RONK_PLAN_OPTS_DEFAULT;
        o.tile_log2_columns = 2;
        o.twiddle_matrix_log2_max = inv_twf != -2 ? inv_twf : 18;
        o.split_log2_rows = k / 2;          // 2^21: 2^10 x 2^11, 2^22: 2^11 x 2^11, 2^23: 2^11 x 2^12
        if (k == 23) o.three_pass_from_log2 = 24;    // two passes
        RCHK(ronk_plan_create_opts(&e->plf, p, g, (u32)k, 1, pl->device, &o));
      }
      const CompiledPlan& F = e->pl2->fwd;
      const CompiledPlan& I = e->plf->inv;
      if (F.pd.passes.size() == 2 && I.pd.passes.size() == 2 && !F.pd.passes[1].small && !I.pd.passes[0].small) {
        TileArgs fa = F.bound(1, nullptr, nullptr, nullptr, e->pl2->d_tmp);
        TileArgs ia = I.bound(0, nullptr, nullptr, nullptr, e->plf->d_tmp);
        const PassDesc& fp = F.pd.passes[1];
        const int kindi = ia.tw_full ? 3 : 1;
        // (lookup first: without an instantiation nothing is enqueued here and the four-launch form below runs alone.
        //  pl2->d_tmp and plf->d_tmp are used directly, without stream_mu / scratch_acquire / scratch_release: both plans are
        //  private to this cache entry, and every use of the entry is serialised by g_cache_mu (held here) + e->done -- that
        //  pair is the ONLY guard of these two scratch buffers.)
        const bool mont = pl->mont_tiled;
        if (mul_mid_matches(fa, ia, fp.logr, (int)fa.logc, kindi) &&
            (mont ? mul_mid_available_mont(fp.logr, (int)fa.logc, kindi) : mul_mid_available(fp.logr, (int)fa.logc, kindi))) {
          HIPCHK(entry_wait(s, e->done));
          RCHK(F.launch(0, d_a, nullptr, nullptr, e->pl2->d_tmp, s, (u64)d, ~(u64)0, stride, 0, 0, 0, (u64)d2));
          bool found = false;
          hipError_t he = mont ? launch_mul_mid_mont(fp.logr, kindi, fa, ia, fa.tiles, fp.block, fp.lds_bytes, s, &found)
                               : launch_mul_mid(fp.logr, kindi, fa, ia, fa.tiles, fp.block, fp.lds_bytes, s, &found);
          if (he != hipSuccess) return hip_fail(he, "launch_mul_mid");
          if (found) {
            RCHK(I.launch(1, nullptr, nullptr, d_out, e->plf->d_tmp, s, ~(u64)0, (u64)m));
            HIPCHK(entry_record(e->done, s));
            return RONK_OK;
          }
          // (not reached: mul_mid_available said yes)
        }
      }
    }
    ronk_plan* const plinv = e->pli ? e->pli : pl;
    if (!e->fab) HIPCHK(hipMalloc((void**)&e->fab, 2 * N * 8));
    HIPCHK(entry_wait(s, e->done));
    RCHK(transform_dev(e->pl2, false, d_a, nullptr, e->fab, s, (u64)d, ~(u64)0, stride, (u64)d2));
    RCHK(transform_dev(plinv, true, e->fab, e->fab + N, d_out, s, ~(u64)0, (u64)m));
    HIPCHK(entry_record(e->done, s));
    return RONK_OK;
  }
  if (!e->fa) HIPCHK(hipMalloc((void**)&e->fa, N * 8));
  if (!e->fb) HIPCHK(hipMalloc((void**)&e->fb, N * 8));
  HIPCHK(entry_wait(s, e->done));                                // previous use of this entry's scratch
  // From<[F;N]> zero padding (mod.rs:503-515) is implicit: the forward transforms read the operands in place and
  // treat indices >= d (d2) as ZERO; the inverse loads NTT(a)*NTT(b) (pointwise product fused into the load) and
  // stores only the d + d2 - 1 product coefficients, straight into the caller's buffer.  No memset, no copy.
  RCHK(transform_dev(pl, false, d_a, nullptr, e->fa, s, (u64)d));
  RCHK(transform_dev(pl, false, d_b, nullptr, e->fb, s, (u64)d2));
  RCHK(transform_dev(pl, true, e->fa, e->fb, d_out, s, ~(u64)0, (u64)m));
  HIPCHK(entry_record(e->done, s));
  return RONK_OK;
}
extern "C" int ronk_poly_mul(uint64_t p, uint64_t g, const uint64_t* a, size_t d, const uint64_t* b, size_t d2,
                             uint64_t* out) {
  if (!a || !b || !out || d == 0 || d2 == 0) return RONK_ERR_INVALID;
  RCHK(need_device());
  const size_t m = d + d2 - 1;
  DevBuf da, db, dout;
  RCHK(da.alloc(d * 8)); RCHK(db.alloc(d2 * 8)); RCHK(dout.alloc(m * 8));
  HIPCHK(hipMemcpy(da.p, a, d * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, d2 * 8, hipMemcpyHostToDevice));
  RCHK(ronk_poly_mul_dev(p, g, da.u(), d, db.u(), d2, dout.u(), 0));
  HIPCHK(hipMemcpy(out, dout.p, m * 8, hipMemcpyDeviceToHost));
  return RONK_OK;
}

