// tools/timeline.hip -- developer tool (not part of the product library): a per-wavefront phase timeline of the tile
// kernels.  rocprofv3's thread trace has no decoder in this image, so the kernel stamps itself: every wavefront records
// the constant 100 MHz clock (s_memrealtime, one time base for the whole chip) at kernel entry, when its 16 global loads
// have landed, on both sides of every workgroup barrier, when its last store is issued and when its stores are
// acknowledged.  tools/timeline_report.py turns the records into "where does a wavefront's life go" and "how many
// wavefronts of a CU compute at any time".
//
//   timeline lat   [iters]     one 2^22 transform at a time, default plan (8-column tiles), HBM-cold rotation
//   timeline lanes [iters]     two streams, 4-column tiles, transforms back to back on each (the `many` regime)
//   timeline batch [polys] [half]   one plan of `polys` polynomials per launch (half = 1: half-image kernels)
// Every scenario is run twice: plain (the production instruction stream + stamps) and FORCE (an s_waitcnt vmcnt(0)
// after the loads, so that load wait and round-1 arithmetic are separated).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/timeline.hip -o build/timeline
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../ronkathon_amd/csrc/plan.h"
#include "experiments/ntt_tile_w.h"

using namespace ronk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NS = 24;   // u64 slots per wavefront record: [0] = count | hw_id << 8, [1] = xcc | block << 32, [2..] stamps

template <int LOGR, int LOGC, int KIND, bool HALF, bool FORCE>
__global__ void __launch_bounds__(1024, HALF ? 8 : 4) stamp_kernel(const TileArgs a, u64* rec) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  typedef TileCfg<LOGC, KIND, false, HALF> CFG;
  u64 st[NS];
  int k = 2;
  st[k++] = wall_clock64();
  const TileCtx cx = tile_ctx<LOGR, CFG>(a, threadIdx.x, bid);
  u64 x[16];
  auto bar = [&] { st[k++] = wall_clock64(); __syncthreads(); st[k++] = wall_clock64(); };
  tile_load<LOGR, false, 0, CFG>(cx, lds, threadIdx.x, x, bar);
  if (FORCE) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("" : "+v"(x[i]));   // the loads are issued before the wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  st[k++] = wall_clock64();
  tile_compute<LOGR, false, 0, CFG>(cx, lds, threadIdx.x, x, bar);
  st[k++] = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st[k++] = wall_clock64();
  if (rec && (threadIdx.x & 63) == 0) {
    const u32 hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID
    const u32 xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);    // HW_REG_XCC_ID[3:0]
    u64* o = rec + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NS;
    o[0] = (u64)k | ((u64)hw << 8);
    o[1] = (u64)xcc | ((u64)blockIdx.x << 32);
#pragma unroll
    for (int i = 2; i < NS; i++) o[i] = i < k ? st[i] : 0;
  }
}

// the wave-local body (ntt_tile_w.h) with stamps: entry, loads issued (FORCE: landed), round 1 parked, before / after the one
// workgroup barrier, last store issued, stores acknowledged -> the report sees ONE barrier pair
template <int KIND, bool FORCE>
__global__ void __launch_bounds__(512, 2) stamp_kernel_w(const TileArgs a, u64* rec) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  u64 st[NS];
  int k = 2;
  st[k++] = wall_clock64();
  tile_body_w<false, KIND>(a, lds, threadIdx.x, bid, [] { __syncthreads(); }, [] { __builtin_amdgcn_wave_barrier(); },
                           [&](int what) {
                             if (what == 0) { if (FORCE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[k++] = wall_clock64(); }
                             else if (what == 2 || what == 3 || what == 4) st[k++] = wall_clock64();
                           });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st[k++] = wall_clock64();
  if (rec && (threadIdx.x & 63) == 0) {
    const u32 hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    const u32 xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    u64* o = rec + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NS;
    o[0] = (u64)k | ((u64)hw << 8);
    o[1] = (u64)xcc | ((u64)blockIdx.x << 32);
#pragma unroll
    for (int i = 2; i < NS; i++) o[i] = i < k ? st[i] : 0;
  }
}
template <int KIND, bool FORCE>
static void launch_kw(const PassDesc& ps, const TileArgs& a, u64* rec, hipStream_t s) {
  static bool done = false;
  auto fn = stamp_kernel_w<KIND, FORCE>;
  if (!done) { CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
  hipLaunchKernelGGL(fn, dim3(ps.grid), dim3(512), TW_LDS_BYTES, s, a, rec);
}

struct DevPlan {
  PlanDesc pd;
  std::vector<u64*> wr;
  std::vector<std::pair<u64*, u64*>> tw;
  std::vector<u64*> twf;
  u64* tmp = nullptr;
};
static DevPlan upload(int log2n, u64 batch, int max_logc, int twf_log) {
  DevPlan d;
  d.pd = build_plan(log2n, batch, false, max_logc, twf_log);
  for (auto& t : d.pd.wr) { u64* p; CK(hipMalloc(&p, t.size() * 8)); CK(hipMemcpy(p, t.data(), t.size() * 8, hipMemcpyHostToDevice)); d.wr.push_back(p); }
  for (auto& t : d.pd.tw) {
    u64 *lo, *hi;
    CK(hipMalloc(&lo, t.lo.size() * 8)); CK(hipMemcpy(lo, t.lo.data(), t.lo.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&hi, t.hi.size() * 8)); CK(hipMemcpy(hi, t.hi.data(), t.hi.size() * 8, hipMemcpyHostToDevice));
    d.tw.push_back({lo, hi});
  }
  for (auto& t : d.pd.twf) { u64* p; CK(hipMalloc(&p, t.size() * 8)); CK(hipMemcpy(p, t.data(), t.size() * 8, hipMemcpyHostToDevice)); d.twf.push_back(p); }
  CK(hipMalloc(&d.tmp, ((size_t)8 << log2n) * batch));
  return d;
}
static TileArgs bind(const DevPlan& d, int pass, const u64* in, u64* out, u64* tmp) {
  const PassDesc& ps = d.pd.passes[pass];
  TileArgs a = ps.args;
  a.in = ps.in_buf == BUF_IN ? in : tmp;
  a.out = ps.out_buf == BUF_OUT ? out : tmp;
  a.wr = d.wr[ps.wr_id];
  if (ps.tw_id >= 0) { a.tw_lo = d.tw[ps.tw_id].first; a.tw_hi = d.tw[ps.tw_id].second; }
  if (ps.twf_id >= 0) a.tw_full = d.twf[ps.twf_id];
  return a;
}

template <int LOGC, int KIND, bool HALF, bool FORCE, int LOGR = 11>
static void launch_k(const PassDesc& ps, const TileArgs& a, u64* rec, hipStream_t s) {
  static bool done = false;
  auto fn = stamp_kernel<LOGR, LOGC, KIND, HALF, FORCE>;
  if (!done) { CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
  hipLaunchKernelGGL(fn, dim3(ps.grid), dim3(ps.block), HALF ? ps.lds_bytes / 2 : ps.lds_bytes, s, a, rec);
}
// pass -> kernel: the shapes of the 2^22 plans (11 rows; 8- or 4-column tiles; KIND 3 / 1 column pass, KIND 2 row pass)
static void launch_pass(const DevPlan& d, int pass, const TileArgs& a, bool half, bool force, u64* rec, hipStream_t s) {
  const PassDesc& ps = d.pd.passes[pass];
  const int lc = (int)a.logc;
  const int kind = pass == 1 ? 2 : (a.tw_full ? 3 : 1);
  if (!tile_cfg_matches(a, 11, lc, kind)) { printf("pass %d does not match kind %d\n", pass, kind); exit(1); }
#define CASE(LC, KD)                                                                               \
  if (lc == LC && kind == KD) {                                                                    \
    if (half) { if (force) launch_k<LC, KD, true, true>(ps, a, rec, s); else launch_k<LC, KD, true, false>(ps, a, rec, s); } \
    else { if (force) launch_k<LC, KD, false, true>(ps, a, rec, s); else launch_k<LC, KD, false, false>(ps, a, rec, s); }   \
    return;                                                                                        \
  }
  CASE(3, 3) CASE(3, 2) CASE(2, 3) CASE(2, 2) CASE(3, 1) CASE(2, 1)
#undef CASE
  printf("no stamp kernel for logc %d kind %d\n", lc, kind);
  exit(1);
}

static FILE* g_out;
static void dump(const char* scenario, const char* label, const std::vector<u64>& h, size_t waves, double region_us, int launches) {
  // text header line + binary payload (`launches` kernel launches of waves / launches wavefronts each)
  fprintf(g_out, "REC %s %s %zu %d %.3f %d\n", scenario, label, waves, NS, region_us, launches);
  fwrite(h.data(), 8, waves * NS, g_out);
  fprintf(g_out, "\n");
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "lat";
  const char* outp = getenv("TIMELINE_OUT") ? getenv("TIMELINE_OUT") : "gpurun_out/timeline.bin";
  g_out = fopen(outp, "ab");
  if (!g_out) { printf("cannot open %s\n", outp); return 1; }
  const size_t n = (size_t)1 << 22;
  const int ROT = 8;
  const int twf = getenv("TIMELINE_TWF") ? atoi(getenv("TIMELINE_TWF")) : 22;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<u64> h(n);
  u64 s = 4242;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl64::P; }

  if (!strcmp(what, "lat")) {
    const int iters = argc > 2 ? atoi(argv[2]) : 24;
    DevPlan d = upload(22, 1, 4, twf);
    printf("lat: logc %u / %u, grid %u x %u\n", d.pd.passes[0].args.logc, d.pd.passes[1].args.logc, d.pd.passes[0].grid, d.pd.passes[0].block);
    std::vector<u64*> in(ROT), out(ROT);
    for (int r = 0; r < ROT; r++) { CK(hipMalloc(&in[r], n * 8)); CK(hipMalloc(&out[r], n * 8)); CK(hipMemcpy(in[r], h.data(), n * 8, hipMemcpyHostToDevice)); }
    const size_t waves = (size_t)d.pd.passes[0].grid * (d.pd.passes[0].block / 64);
    u64* rec; CK(hipMalloc(&rec, 2 * waves * NS * 8));
    for (int force = 0; force < 2; force++) {
      for (int stamped = 0; stamped < 2; stamped++) {
        CK(hipMemset(rec, 0, 2 * waves * NS * 8));
        CK(hipDeviceSynchronize());
        float ms = 0;
        for (int it = 0; it < iters; it++) {
          if (it == iters / 2) CK(hipEventRecord(e0, 0));
          const int r = it % ROT;
          launch_pass(d, 0, bind(d, 0, in[r], out[r], d.tmp), false, force, stamped ? rec : nullptr, 0);
          launch_pass(d, 1, bind(d, 1, in[r], out[r], d.tmp), false, force, stamped ? rec + waves * NS : nullptr, 0);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (iters - iters / 2);
        printf("lat force %d stamped %d: %.2f us per transform\n", force, stamped, us);
        if (stamped) {
          std::vector<u64> hr(2 * waves * NS);
          CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
          dump("lat", force ? "force" : "plain", hr, 2 * waves, us, 2);
        }
      }
    }
  } else if (!strcmp(what, "lanes")) {
    const int iters = argc > 2 ? atoi(argv[2]) : 16;
    const int lc = argc > 3 ? atoi(argv[3]) : 2;
    const int half = argc > 4 ? atoi(argv[4]) : 0;
    DevPlan d[2] = {upload(22, 1, lc, twf), upload(22, 1, lc, twf)};
    printf("lanes: logc %u / %u, grid %u x %u, half %d\n", d[0].pd.passes[0].args.logc, d[0].pd.passes[1].args.logc, d[0].pd.passes[0].grid, d[0].pd.passes[0].block, half);
    hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    std::vector<u64*> in[2], out[2];
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < ROT; r++) {
        u64 *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
        in[l].push_back(a); out[l].push_back(b);
      }
    const size_t waves = (size_t)d[0].pd.passes[0].grid * (d[0].pd.passes[0].block / 64);
    // slots: [lane][last two transforms][pass]
    u64* rec; CK(hipMalloc(&rec, 8 * waves * NS * 8));
    for (int force = 0; force < 2; force++) {
      for (int stamped = 0; stamped < 2; stamped++) {
        CK(hipMemset(rec, 0, 8 * waves * NS * 8));
        CK(hipDeviceSynchronize());
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        CK(hipEventRecord(e0, st[0])); CK(hipEventRecord(f0, st[1]));
        for (int it = 0; it < iters; it++)
          for (int l = 0; l < 2; l++) {
            const int r = it % ROT;
            const bool rec_it = stamped && it >= iters - 4 && it < iters - 2;   // two transforms from the steady state
            u64* base = rec_it ? rec + ((size_t)(l * 2 + (it - (iters - 4))) * 2) * waves * NS : nullptr;
            launch_pass(d[l], 0, bind(d[l], 0, in[l][r], out[l][r], d[l].tmp), half, force, base, st[l]);
            launch_pass(d[l], 1, bind(d[l], 1, in[l][r], out[l][r], d[l].tmp), half, force, base ? base + waves * NS : nullptr, st[l]);
          }
        CK(hipEventRecord(e1, st[0])); CK(hipEventRecord(f1, st[1]));
        CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
        float m0 = 0, m1 = 0; CK(hipEventElapsedTime(&m0, e0, e1)); CK(hipEventElapsedTime(&m1, f0, f1));
        const double us = (m0 > m1 ? m0 : m1) * 1e3 / (2 * iters);
        printf("lanes force %d stamped %d: %.2f us per transform (%.0f NTT/s)\n", force, stamped, us, 1e6 / us);
        if (stamped) {
          std::vector<u64> hr(8 * waves * NS);
          CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
          dump(half ? "lanes_half" : (lc == 2 ? "lanes_c4" : "lanes_c8"), force ? "force" : "plain", hr, 8 * waves, us, 8);
        }
      }
    }
  } else if (!strcmp(what, "b16")) {
    // timeline b16 [half]: BASELINE config 4, 1024 x 2^16 in one launch pair (2^8-row passes, 16-column tiles), plain stamps
    const int half = argc > 2 ? atoi(argv[2]) : 1;
    const size_t n16 = (size_t)1 << 16, total = n16 * 1024;
    DevPlan d = upload(16, 1024, 4, 18);
    const PassDesc &p0 = d.pd.passes[0], &p1 = d.pd.passes[1];
    printf("b16: pass 0 logr %d logc %u grid %u x %u, pass 1 logr %d logc %u, half %d\n", p0.logr, p0.args.logc, p0.grid, p0.block, p1.logr, p1.args.logc, half);
    if (p0.logr != 8 || p0.args.logc != 4 || p1.logr != 8 || p1.args.logc != 4) { printf("unexpected shape\n"); return 1; }
    u64 *in, *out; CK(hipMalloc(&in, total * 8)); CK(hipMalloc(&out, total * 8));
    for (size_t off = 0; off < total; off += n) CK(hipMemcpy(in + off, h.data(), (total - off < n ? total - off : n) * 8, hipMemcpyHostToDevice));
    const size_t waves = (size_t)p0.grid * (p0.block / 64);
    u64* rec; CK(hipMalloc(&rec, 2 * waves * NS * 8));
    for (int stamped = 0; stamped < 2; stamped++) {
      CK(hipMemset(rec, 0, 2 * waves * NS * 8));
      CK(hipDeviceSynchronize());
      float ms = 0;
      const int iters = 4;
      for (int it = 0; it < iters; it++) {
        if (it == 1) CK(hipEventRecord(e0, 0));
        const TileArgs a0 = bind(d, 0, in, out, d.tmp), a1 = bind(d, 1, in, out, d.tmp);
        if (!tile_cfg_matches(a0, 8, 4, 3) || !tile_cfg_matches(a1, 8, 4, 2)) { printf("passes are not (8,4,3) / (8,4,2)\n"); return 1; }
        u64* r0 = stamped ? rec : nullptr; u64* r1 = stamped ? rec + waves * NS : nullptr;
        if (half) { launch_k<4, 3, true, false, 8>(p0, a0, r0, 0); launch_k<4, 2, true, false, 8>(p1, a1, r1, 0); }
        else { launch_k<4, 3, false, false, 8>(p0, a0, r0, 0); launch_k<4, 2, false, false, 8>(p1, a1, r1, 0); }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / (iters - 1);
      printf("b16 half %d stamped %d: %.1f us per batch\n", half, stamped, us);
      if (stamped) {
        std::vector<u64> hr(2 * waves * NS);
        CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
        dump(half ? "b16_half" : "b16_full", "plain", hr, 2 * waves, us, 2);
      }
    }
  } else if (!strcmp(what, "wlat")) {
    // timeline wlat: one 2^22 transform at a time with 4-column tiles (two workgroups of 8 wavefronts per CU), HBM-cold:
    // ntt_tile.h's body ("c4") against the wave-local first exchange of ntt_tile_w.h ("c4w"), FORCE stamps (loads landed)
    const int iters = 24;
    DevPlan d = upload(22, 1, 2, twf);
    std::vector<u64*> in(ROT), out(ROT);
    for (int r = 0; r < ROT; r++) { CK(hipMalloc(&in[r], n * 8)); CK(hipMalloc(&out[r], n * 8)); CK(hipMemcpy(in[r], h.data(), n * 8, hipMemcpyHostToDevice)); }
    const size_t waves = (size_t)d.pd.passes[0].grid * (d.pd.passes[0].block / 64);
    u64* rec; CK(hipMalloc(&rec, 2 * waves * NS * 8));
    for (int wl = 0; wl < 2; wl++)
      for (int stamped = 0; stamped < 2; stamped++) {
        CK(hipMemset(rec, 0, 2 * waves * NS * 8));
        CK(hipDeviceSynchronize());
        float ms = 0;
        for (int it = 0; it < iters; it++) {
          if (it == iters / 2) CK(hipEventRecord(e0, 0));
          const int r = it % ROT;
          const TileArgs a0 = bind(d, 0, in[r], out[r], d.tmp), a1 = bind(d, 1, in[r], out[r], d.tmp);
          if (wl) {
            if (a0.tw_full) launch_kw<3, true>(d.pd.passes[0], a0, stamped ? rec : nullptr, 0);
            else launch_kw<1, true>(d.pd.passes[0], a0, stamped ? rec : nullptr, 0);
            launch_kw<2, true>(d.pd.passes[1], a1, stamped ? rec + waves * NS : nullptr, 0);
          } else {
            launch_pass(d, 0, a0, false, true, stamped ? rec : nullptr, 0);
            launch_pass(d, 1, a1, false, true, stamped ? rec + waves * NS : nullptr, 0);
          }
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (iters - iters / 2);
        printf("wlat wave-local %d stamped %d: %.2f us per transform\n", wl, stamped, us);
        if (stamped) {
          std::vector<u64> hr(2 * waves * NS);
          CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
          dump(wl ? "lat_c4w" : "lat_c4", "force", hr, 2 * waves, us, 2);
        }
      }
  } else if (!strcmp(what, "p24")) {
    // timeline p24: the strided FIRST pass of the three-pass 2^24 plan (2^9 rows x 16 columns per tile, row stride 256 KiB on
    // both sides), one stream, HBM-cold rotation; plain and FORCE
    const size_t n24 = (size_t)1 << 24;
    DevPlan d = upload(24, 1, 4, 18);
    // (upload() builds with three_pass_from = 25: rebuild with 23 so that 2^24 is three passes like the library's plan)
    d.pd = build_plan(24, 1, false, 4, 18, 23);
    d.wr.clear(); d.tw.clear(); d.twf.clear();
    for (auto& t : d.pd.wr) { u64* p; CK(hipMalloc(&p, t.size() * 8)); CK(hipMemcpy(p, t.data(), t.size() * 8, hipMemcpyHostToDevice)); d.wr.push_back(p); }
    for (auto& t : d.pd.tw) {
      u64 *lo, *hi;
      CK(hipMalloc(&lo, t.lo.size() * 8)); CK(hipMemcpy(lo, t.lo.data(), t.lo.size() * 8, hipMemcpyHostToDevice));
      CK(hipMalloc(&hi, t.hi.size() * 8)); CK(hipMemcpy(hi, t.hi.data(), t.hi.size() * 8, hipMemcpyHostToDevice));
      d.tw.push_back({lo, hi});
    }
    for (auto& t : d.pd.twf) { u64* p; CK(hipMalloc(&p, t.size() * 8)); CK(hipMemcpy(p, t.data(), t.size() * 8, hipMemcpyHostToDevice)); d.twf.push_back(p); }
    const PassDesc& ps = d.pd.passes[0];
    printf("p24: %zu passes; pass 0 logr %d logc %u grid %u x %u lds %zu\n", d.pd.passes.size(), ps.logr, ps.args.logc, ps.grid, ps.block, ps.lds_bytes);
    if (ps.logr != 9 || ps.args.logc != 4) { printf("unexpected shape\n"); return 1; }
    const int R3 = 3;
    std::vector<u64*> in(R3), out(R3);
    std::vector<u64> h24(n24);
    for (auto& v : h24) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl64::P; }
    for (int r = 0; r < R3; r++) { CK(hipMalloc(&in[r], n24 * 8)); CK(hipMalloc(&out[r], n24 * 8)); CK(hipMemcpy(in[r], h24.data(), n24 * 8, hipMemcpyHostToDevice)); }
    const size_t waves = (size_t)ps.grid * (ps.block / 64);
    u64* rec; CK(hipMalloc(&rec, waves * NS * 8));
    const int iters = 9;
    for (int force = 0; force < 2; force++)
      for (int stamped = 0; stamped < 2; stamped++) {
        CK(hipMemset(rec, 0, waves * NS * 8));
        CK(hipDeviceSynchronize());
        float ms = 0;
        for (int it = 0; it < iters; it++) {
          if (it == 3) CK(hipEventRecord(e0, 0));
          const TileArgs a = bind(d, 0, in[it % R3], out[it % R3], out[it % R3]);   // pass 0: in -> tmp (here: the out buffer)
          if (!tile_cfg_matches(a, 9, 4, 1)) { printf("pass 0 is not KIND 1\n"); return 1; }
          if (force) launch_k<4, 1, false, true, 9>(ps, a, stamped ? rec : nullptr, 0);
          else launch_k<4, 1, false, false, 9>(ps, a, stamped ? rec : nullptr, 0);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (iters - 3);
        printf("p24 pass 0 force %d stamped %d: %.2f us per launch\n", force, stamped, us);
        if (stamped) {
          std::vector<u64> hr(waves * NS);
          CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
          dump("p24_pass0", force ? "force" : "plain", hr, waves, us, 1);
        }
      }
  } else if (!strcmp(what, "nlanes")) {
    // timeline nlanes <streams> <logc> <half> [iters]: throughput only (no stamps): N streams, transforms back to back on each
    const int NL = argc > 2 ? atoi(argv[2]) : 4;
    const int lc = argc > 3 ? atoi(argv[3]) : 2;
    const int half = argc > 4 ? atoi(argv[4]) : 1;
    const int iters = argc > 5 ? atoi(argv[5]) : 24;
    std::vector<DevPlan> d;
    std::vector<hipStream_t> st(NL);
    std::vector<std::vector<u64*>> in(NL), out(NL);
    const int ROTN = getenv("TIMELINE_ROT") ? atoi(getenv("TIMELINE_ROT")) : 4;
    for (int l = 0; l < NL; l++) {
      d.push_back(upload(22, 1, lc, twf));
      CK(hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking));
      for (int r = 0; r < ROTN; r++) {
        u64 *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
        in[l].push_back(a); out[l].push_back(b);
      }
    }
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      std::vector<hipEvent_t> b0(NL), b1(NL);
      for (int l = 0; l < NL; l++) { CK(hipEventCreate(&b0[l])); CK(hipEventCreate(&b1[l])); CK(hipEventRecord(b0[l], st[l])); }
      for (int it = 0; it < iters; it++)
        for (int l = 0; l < NL; l++) {
          const int r = it % ROTN;
          launch_pass(d[l], 0, bind(d[l], 0, in[l][r], out[l][r], d[l].tmp), half, false, nullptr, st[l]);
          launch_pass(d[l], 1, bind(d[l], 1, in[l][r], out[l][r], d[l].tmp), half, false, nullptr, st[l]);
        }
      float mx = 0;
      for (int l = 0; l < NL; l++) { CK(hipEventRecord(b1[l], st[l])); }
      for (int l = 0; l < NL; l++) { CK(hipEventSynchronize(b1[l])); float m = 0; CK(hipEventElapsedTime(&m, b0[0], b1[l])); if (m > mx) mx = m; }
      const double us = mx * 1e3 / (NL * iters);
      printf("nlanes %d logc %d half %d rep %d: %.2f us per transform (%.0f NTT/s)\n", NL, lc, half, rep, us, 1e6 / us);
    }
  } else if (!strcmp(what, "batch")) {
    const int polys = argc > 2 ? atoi(argv[2]) : 8;
    const int half = argc > 3 ? atoi(argv[3]) : 0;
    const int lc = argc > 4 ? atoi(argv[4]) : 3;
    DevPlan d = upload(22, (u64)polys, lc, twf);
    printf("batch %d: logc %u / %u, grid %u x %u, half %d\n", polys, d.pd.passes[0].args.logc, d.pd.passes[1].args.logc, d.pd.passes[0].grid, d.pd.passes[0].block, half);
    const int ROTB = 3;
    std::vector<u64*> in(ROTB), out(ROTB);
    for (int r = 0; r < ROTB; r++) {
      CK(hipMalloc(&in[r], n * 8 * polys)); CK(hipMalloc(&out[r], n * 8 * polys));
      for (int p = 0; p < polys; p++) CK(hipMemcpy(in[r] + (size_t)p * n, h.data(), n * 8, hipMemcpyHostToDevice));
    }
    const size_t waves = (size_t)d.pd.passes[0].grid * (d.pd.passes[0].block / 64);
    u64* rec; CK(hipMalloc(&rec, 2 * waves * NS * 8));
    const int iters = 6;
    for (int force = 0; force < 2; force++) {
      for (int stamped = 0; stamped < 2; stamped++) {
        CK(hipMemset(rec, 0, 2 * waves * NS * 8));
        CK(hipDeviceSynchronize());
        float ms = 0;
        for (int it = 0; it < iters; it++) {
          if (it == iters / 2) CK(hipEventRecord(e0, 0));
          const int r = it % ROTB;
          launch_pass(d, 0, bind(d, 0, in[r], out[r], d.tmp), half, force, stamped ? rec : nullptr, 0);
          launch_pass(d, 1, bind(d, 1, in[r], out[r], d.tmp), half, force, stamped ? rec + waves * NS : nullptr, 0);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (iters - iters / 2) / polys;
        printf("batch force %d stamped %d: %.2f us per transform (%.0f NTT/s)\n", force, stamped, us, 1e6 / us);
        if (stamped) {
          std::vector<u64> hr(2 * waves * NS);
          CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
          std::string sc = std::string("batch") + std::to_string(polys) + (half ? "_half" : "") + (lc == 2 ? "_c4" : "");
          dump(sc.c_str(), force ? "force" : "plain", hr, 2 * waves, us, 2);
        }
      }
    }
  } else {
    printf("usage: timeline lat|lanes|batch ...\n");
    return 1;
  }
  fclose(g_out);
  return 0;
}
