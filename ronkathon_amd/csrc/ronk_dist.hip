// ronk_dist.hip -- C ABI of libronk_ntt.so, part 4: the local phases of the multi-GPU four-step transform
// (plan.h build_dist_phase1/2); the exchange between them is the host side's all-to-all (ronkathon_amd/dist.py).
#include "runtime.h"

// ------------------------------------------------------------------------------ multi-GPU four-step
struct ronk_dist_plan {
  DistShape sh;
  CompiledPlan p1, p2;
  u64* d_tmp = nullptr;  // n / world elements
  int device;
};
extern "C" int ronk_dist_plan_destroy(ronk_dist_plan* pl) {
  if (!pl) return RONK_ERR_INVALID;
  pl->p1.release(); pl->p2.release();
  if (pl->d_tmp) (void)hipFree(pl->d_tmp);
  delete pl;
  return RONK_OK;
}
extern "C" int ronk_dist_plan_create(ronk_dist_plan** out, uint32_t log2n, int inverse, int rank, int world, int device) {
  if (!out || world < 1 || rank < 0 || rank >= world) return RONK_ERR_INVALID;
  *out = nullptr;
  if (log2n > 32) return RONK_ERR_NO_ROOT;  // 2-adicity of p - 1 is 32
  DistShape sh;
  if (!dist_shape((int)log2n, world, &sh)) return RONK_ERR_UNSUPPORTED;
  RCHK(need_device());
  if (device >= 0) HIPCHK(hipSetDevice(device));
  else HIPCHK(hipGetDevice(&device));
  ronk_dist_plan* pl = new ronk_dist_plan();
  pl->sh = sh; pl->device = device;
  int rc = pl->p1.compile(build_dist_phase1((int)log2n, inverse != 0, rank, world));
  if (!rc) rc = pl->p2.compile(build_dist_phase2((int)log2n, inverse != 0, rank, world));
  if (!rc) {
    hipError_t e = hipMalloc((void**)&pl->d_tmp, (sh.n / sh.W) * 8);
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(scratch)");
  }
  if (rc) { ronk_dist_plan_destroy(pl); return rc; }
  *out = pl;
  return RONK_OK;
}
extern "C" int ronk_dist_phase1_dev(ronk_dist_plan* pl, const uint64_t* d_in, uint64_t* d_send, void* st) {
  if (!pl || !d_in || !d_send || d_in == d_send) return RONK_ERR_INVALID;
  return pl->p1.run(d_in, nullptr, d_send, pl->d_tmp, (hipStream_t)st);
}
extern "C" int ronk_dist_phase2_dev(ronk_dist_plan* pl, const uint64_t* d_recv, uint64_t* d_out, void* st) {
  if (!pl || !d_recv || !d_out || d_recv == d_out) return RONK_ERR_INVALID;
  return pl->p2.run(d_recv, nullptr, d_out, pl->d_tmp, (hipStream_t)st);
}

