// ronk_dist.hip -- C ABI of libronk_ntt.so, part 4: the multi-GPU four-step transform (plan.h build_dist_phase1/2).
//   * ronk_dist_*      one rank's local phases; the exchange between them is the caller's all-to-all (one process per
//                      GPU: ronkathon_amd/dist.py issues it through torch.distributed = RCCL over xGMI)
//   * ronk_sharded_*   the whole sharded transform inside the library for a single-process host (the Rust host of
//                      BASELINE config 5): one compute and one copy stream per device, the exchange as a mesh of
//                      hipMemcpyPeerAsync copies over xGMI, column chunks so that chunk j is on the links while chunk
//                      j+1 is still being computed (SURVEY.md 8e "overlap")
#include "runtime.h"

// ------------------------------------------------------------------------------ multi-GPU four-step
struct ronk_dist_plan {
  DistShape sh;
  CompiledPlan p1, p2;
  u64* d_tmp = nullptr;  // n / world elements
  int device;
  int chunks = 1;        // column chunks of phase 1 (plan.h); p1 is compiled for chunk 0
};
extern "C" int ronk_dist_plan_destroy(ronk_dist_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  pl->p1.release(); pl->p2.release();
  if (pl->d_tmp) (void)hipFree(pl->d_tmp);
  delete pl;
  return RONK_OK;
}
extern "C" int ronk_dist_plan_create_chunked(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world,
                                             int device, int chunks);
extern "C" int ronk_dist_plan_create(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world, int device) {
  return ronk_dist_plan_create_chunked(out, log2n, inverse, rank, world, device, 1);
}
extern "C" int ronk_dist_plan_create_chunked(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world,
                                             int device, int chunks) {
  if (!out || world < 1 || rank < 0 || rank >= world) return RONK_ERR_INVALID;
  *out = nullptr;
  if (log2n > 32) return RONK_ERR_NO_ROOT;  // 2-adicity of p - 1 is 32
  DistShape sh;
  if (!dist_shape((int)log2n, world, &sh)) return RONK_ERR_UNSUPPORTED;
  if (!dist_chunks_ok(sh, chunks)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  ronk_dist_plan* pl = new ronk_dist_plan();
  pl->sh = sh; pl->device = device; pl->chunks = chunks;
  int rc = pl->p1.compile(build_dist_phase1((int)log2n, inverse != 0, rank, world, 4, 0, 0, chunks));
  if (!rc) rc = pl->p2.compile(build_dist_phase2((int)log2n, inverse != 0, rank, world, 4, 0, chunks));
  if (!rc) {
    hipError_t e = hipMalloc((void**)&pl->d_tmp, (sh.n / sh.W) * 8);
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(scratch)");
  }
  if (rc) { ronk_dist_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}
// one column chunk of phase 1: reads columns [chunk*Cwc, (chunk+1)*Cwc) of d_in ([R][C/world]) and writes the contiguous
// piece d_send[chunk*R*Cwc ..) = `world` blocks [R/world][Cwc]
extern "C" int ronk_dist_phase1_chunk_dev(ronk_dist_plan* pl, int chunk, const uint64_t* d_in, uint64_t* d_send, void* st) {
  if (!pl || !d_in || !d_send || d_in == d_send || chunk < 0 || chunk >= pl->chunks) return RONK_ERR_INVALID;
  const u64 Cwc = pl->sh.Cw / (u64)pl->chunks;
  return pl->p1.run(d_in + (u64)chunk * Cwc, nullptr, d_send + (u64)chunk * pl->sh.R * Cwc, pl->d_tmp, (hipStream_t)st,
                    ~(u64)0, ~(u64)0, 0, (u64)chunk * Cwc);
}
extern "C" int ronk_dist_phase1_dev(ronk_dist_plan* pl, const uint64_t* d_in, uint64_t* d_send, void* st) {
  if (!pl || !d_in || !d_send || d_in == d_send) return RONK_ERR_INVALID;
  for (int j = 0; j < pl->chunks; j++) RCHK(ronk_dist_phase1_chunk_dev(pl, j, d_in, d_send, st));
  return RONK_OK;
}
extern "C" int ronk_dist_phase2_dev(ronk_dist_plan* pl, const uint64_t* d_recv, uint64_t* d_out, void* st) {
  if (!pl || !d_recv || !d_out || d_recv == d_out) return RONK_ERR_INVALID;
  return pl->p2.run(d_recv, nullptr, d_out, pl->d_tmp, (hipStream_t)st);
}


// ------------------------------------------------------------------------------ sharded transform, single process
// Rank g = devices[g].  Per rank: phase-1 plan (compiled for chunk 0; chunk j shifts the twiddle's column offset),
// phase-2 plan, scratch, send and receive buffers (n/W elements each), a compute stream and a copy stream.
//   compute_g :  phase 1 chunk 0 | chunk 1 | ... | (wait: every rank's copies have landed) phase 2
//   copy_g    :        (wait chunk 0) W copies | (wait chunk 1) W copies | ...
// Buffers are reused by the next call: the copies of call k+1 wait for every phase 2 of call k (recv), and phase 1 of
// call k+1 waits for the copies of call k on its own device (send) -- all by events, nothing blocks the host.
struct ShardRank {
  int device = 0;
  CompiledPlan p1, p2;
  u64 *tmp = nullptr, *send = nullptr, *recv = nullptr;
  u64 *stage_in = nullptr, *stage_out = nullptr;   // host-pointer entry point only (lazy)
  hipStream_t compute = nullptr, copy = nullptr;
  std::vector<hipEvent_t> chunk_done;
  hipEvent_t sent = nullptr, done = nullptr;
  bool used = false;
};
struct ronk_sharded_plan {
  DistShape sh;
  int ndev = 0, chunks = 1;
  bool inverse = false;
  std::vector<ShardRank> r;
  std::mutex mu;
};

static int on_device(int dev) {
  HIPCHK(hipSetDevice(dev));
  return RONK_OK;
}

extern "C" int ronk_sharded_plan_destroy(ronk_sharded_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  for (auto& k : pl->r) {
    (void)hipSetDevice(k.device);
    if (k.compute) (void)hipStreamSynchronize(k.compute);
    if (k.copy) (void)hipStreamSynchronize(k.copy);
    k.p1.release(); k.p2.release();
    for (u64* q : {k.tmp, k.send, k.recv, k.stage_in, k.stage_out}) if (q) (void)hipFree(q);
    for (auto e : k.chunk_done) (void)hipEventDestroy(e);
    if (k.sent) (void)hipEventDestroy(k.sent);
    if (k.done) (void)hipEventDestroy(k.done);
    if (k.compute) (void)hipStreamDestroy(k.compute);
    if (k.copy) (void)hipStreamDestroy(k.copy);
  }
  delete pl;
  return RONK_OK;
}

extern "C" int ronk_sharded_plan_create(ronk_sharded_plan** out, uint32_t log2n, int inverse, const int* devices, int ndev,
                                        int chunks) {
  if (!out || !devices || ndev < 1) return RONK_ERR_INVALID;
  *out = nullptr;
  if (log2n > 32) return RONK_ERR_NO_ROOT;  // 2-adicity of p - 1 is 32
  DistShape sh;
  if (!dist_shape((int)log2n, ndev, &sh)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  int ndevices = 0;
  HIPCHK(hipGetDeviceCount(&ndevices));
  for (int g = 0; g < ndev; g++)
    if (devices[g] < 0 || devices[g] >= ndevices) return RONK_ERR_INVALID;
  if (chunks <= 0) {   // default: up to 4 chunks (the exchange of chunk j hides under chunks j+1 ..)
    chunks = 4;
    while (chunks > 1 && !dist_chunks_ok(sh, chunks)) chunks >>= 1;
  }
  if (!dist_chunks_ok(sh, chunks)) return RONK_ERR_UNSUPPORTED;
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  ronk_sharded_plan* pl = new ronk_sharded_plan();
  pl->sh = sh; pl->ndev = ndev; pl->chunks = chunks; pl->inverse = inverse != 0;
  pl->r.resize(ndev);
  const size_t per = (size_t)(sh.n / sh.W);
  int rc = RONK_OK;
  for (int g = 0; g < ndev && !rc; g++) {
    ShardRank& k = pl->r[g];
    k.device = devices[g];
    rc = on_device(k.device);
    // peer access to every other device of the plan (xGMI); "already enabled" is fine, a refusal falls back to
    // staged copies inside hipMemcpyPeerAsync
    for (int h = 0; h < ndev && !rc; h++)
      if (devices[h] != k.device) { (void)hipDeviceEnablePeerAccess(devices[h], 0); (void)hipGetLastError(); }
    if (!rc) rc = k.p1.compile(build_dist_phase1((int)log2n, inverse != 0, g, ndev, 4, 0, 0, chunks));
    if (!rc) rc = k.p2.compile(build_dist_phase2((int)log2n, inverse != 0, g, ndev, 4, 0, chunks));
    if (!rc && (k.p1.pd.passes.empty() || k.p2.pd.passes.empty())) rc = RONK_ERR_UNSUPPORTED;
    hipError_t e = hipSuccess;
    if (!rc) e = hipMalloc((void**)&k.tmp, per * 8);
    if (!rc && e == hipSuccess) e = hipMalloc((void**)&k.send, per * 8);
    if (!rc && e == hipSuccess) e = hipMalloc((void**)&k.recv, per * 8);
    if (!rc && e == hipSuccess) e = hipStreamCreateWithFlags(&k.compute, hipStreamNonBlocking);
    if (!rc && e == hipSuccess) e = hipStreamCreateWithFlags(&k.copy, hipStreamNonBlocking);
    k.chunk_done.resize(chunks, nullptr);
    for (int j = 0; j < chunks && !rc && e == hipSuccess; j++) e = hipEventCreateWithFlags(&k.chunk_done[j], hipEventDisableTiming);
    if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&k.sent, hipEventDisableTiming);
    if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&k.done, hipEventDisableTiming);
    if (!rc && e != hipSuccess) rc = hip_fail(e, "sharded plan resources");
  }
  (void)hipSetDevice(prev);
  if (rc) { ronk_sharded_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}

extern "C" int ronk_sharded_plan_info(const ronk_sharded_plan* pl, uint64_t* rows, uint64_t* cols, uint64_t* per_rank,
                                      int* chunks) {
  if (!pl) return RONK_ERR_INVALID;
  if (rows) *rows = pl->sh.R;
  if (cols) *cols = pl->sh.C;
  if (per_rank) *per_rank = pl->sh.n / pl->sh.W;
  if (chunks) *chunks = pl->chunks;
  return RONK_OK;
}

static int sharded_enqueue(ronk_sharded_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out) {
  const DistShape& sh = pl->sh;
  const int W = pl->ndev, chunks = pl->chunks;
  const u64 Cwc = sh.Cw / (u64)chunks, blk = sh.Rw * Cwc;
  // phase 1 + exchange
  for (int g = 0; g < W; g++) {
    ShardRank& k = pl->r[g];
    RCHK(on_device(k.device));
    if (k.used) {
      HIPCHK(hipStreamWaitEvent(k.compute, k.sent, 0));                 // send buffer: the previous call's copies are out
      for (int h = 0; h < W; h++) HIPCHK(hipStreamWaitEvent(k.copy, pl->r[h].done, 0));   // recv buffers: phase 2 has read them
    }
    for (int j = 0; j < chunks; j++) {
      u64* piece = k.send + (u64)j * sh.R * Cwc;
      RCHK(k.p1.run(d_in[g] + (u64)j * Cwc, nullptr, piece, k.tmp, k.compute, ~(u64)0, ~(u64)0, 0, (u64)j * Cwc));
      HIPCHK(hipEventRecord(k.chunk_done[j], k.compute));
      HIPCHK(hipStreamWaitEvent(k.copy, k.chunk_done[j], 0));
      for (int hh = 0; hh < W; hh++) {
        const int h = (g + hh) % W;                                     // start with the local block, then ring order
        u64* dst = pl->r[h].recv + ((u64)g * chunks + j) * blk;
        const u64* src = piece + (u64)h * blk;
        if (pl->r[h].device == k.device) HIPCHK(hipMemcpyAsync(dst, src, blk * 8, hipMemcpyDeviceToDevice, k.copy));
        else HIPCHK(hipMemcpyPeerAsync(dst, pl->r[h].device, src, k.device, blk * 8, k.copy));
      }
    }
    HIPCHK(hipEventRecord(k.sent, k.copy));
  }
  // phase 2: every rank waits for every sender
  for (int h = 0; h < W; h++) {
    ShardRank& k = pl->r[h];
    RCHK(on_device(k.device));
    for (int g = 0; g < W; g++) HIPCHK(hipStreamWaitEvent(k.compute, pl->r[g].sent, 0));
    RCHK(k.p2.run(k.recv, nullptr, d_out[h], k.tmp, k.compute));
    HIPCHK(hipEventRecord(k.done, k.compute));
    k.used = true;
  }
  return RONK_OK;
}

extern "C" int ronk_ntt_sharded_dev(ronk_sharded_plan* pl, const uint64_t* const* d_in, uint64_t* const* d_out) {
  if (!pl || !d_in || !d_out) return RONK_ERR_INVALID;
  for (int g = 0; g < pl->ndev; g++)
    if (!d_in[g] || !d_out[g] || d_in[g] == d_out[g]) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  int rc = sharded_enqueue(pl, d_in, d_out);
  (void)hipSetDevice(prev);
  return rc;
}

extern "C" int ronk_sharded_sync(ronk_sharded_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  std::lock_guard<std::mutex> lk(pl->mu);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  int rc = RONK_OK;
  for (auto& k : pl->r) {
    if (hipSetDevice(k.device) != hipSuccess || hipStreamSynchronize(k.copy) != hipSuccess ||
        hipStreamSynchronize(k.compute) != hipSuccess) { rc = hip_fail(hipGetLastError(), "ronk_sharded_sync"); break; }
  }
  (void)hipSetDevice(prev);
  return rc;
}

// Host natural-order vector in, natural-order vector out (Polynomial::fft / ifft of src/polynomial/mod.rs:273-323,
// :430-484 at a size sharded over the node): scatter the column blocks (strided H2D copies), transform, gather
// X[k1 + R*k2] from the ranks' [C][R/W] blocks.
extern "C" int ronk_ntt_sharded(ronk_sharded_plan* pl, const uint64_t* in, uint64_t* out) {
  if (!pl || !in || !out) return RONK_ERR_INVALID;
  const DistShape& sh = pl->sh;
  const size_t per = (size_t)(sh.n / sh.W);
  int prev = 0;
  HIPCHK(hipGetDevice(&prev));
  std::vector<const uint64_t*> din(pl->ndev);
  std::vector<uint64_t*> dout(pl->ndev);
  int rc = RONK_OK;
  {
    std::lock_guard<std::mutex> lk(pl->mu);
    for (int g = 0; g < pl->ndev && !rc; g++) {
      ShardRank& k = pl->r[g];
      rc = on_device(k.device);
      hipError_t e = hipSuccess;
      if (!rc && !k.stage_in) e = hipMalloc((void**)&k.stage_in, per * 8);
      if (!rc && e == hipSuccess && !k.stage_out) e = hipMalloc((void**)&k.stage_out, per * 8);
      // rank g's [R][C/W] column block of the R x C input
      if (!rc && e == hipSuccess)
        e = hipMemcpy2DAsync(k.stage_in, sh.Cw * 8, in + (size_t)g * sh.Cw, sh.C * 8, sh.Cw * 8, sh.R, hipMemcpyHostToDevice, k.compute);
      if (!rc && e != hipSuccess) rc = hip_fail(e, "scatter");
      din[g] = k.stage_in; dout[g] = k.stage_out;
    }
    if (!rc) rc = sharded_enqueue(pl, din.data(), dout.data());
    for (int h = 0; h < pl->ndev && !rc; h++) {
      ShardRank& k = pl->r[h];
      rc = on_device(k.device);
      // rank h's [C][R/W] block holds X[(h*R/W + k1l) + R*k2]
      hipError_t e = hipSuccess;
      if (!rc) e = hipMemcpy2DAsync(out + (size_t)h * sh.Rw, sh.R * 8, k.stage_out, sh.Rw * 8, sh.Rw * 8, sh.C, hipMemcpyDeviceToHost, k.compute);
      if (!rc && e == hipSuccess) e = hipStreamSynchronize(k.compute);
      if (!rc && e != hipSuccess) rc = hip_fail(e, "gather");
    }
  }
  (void)hipSetDevice(prev);
  return rc;
}
