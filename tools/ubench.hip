// tools/ubench.hip -- developer micro-benchmarks (not part of the product library).
//   * ablation timings of the 2^22 plan's two tile passes (ntt_tile.h ABL masks)
//   * raw field-op issue rates (mul / add / sub / shift-mul) and memory-pattern floors
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/ubench.hip -o build/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <vector>

#include "../ronkathon_amd/csrc/plan.h"

using namespace ronk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// LOGC / KIND: the compile-time pass shape (TileCfg, ntt_tile.h); -1 / 0 = the generic body
template <int LOGR, bool INV, int ABL, int LOGC = -1, int KIND = 0>
__global__ void __launch_bounds__(1024) abl_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tile_body<LOGR, INV, ABL, TileCfg<LOGC, KIND, cfg_ldstw(LOGR, LOGC, KIND)>>(a, lds, threadIdx.x, bid, [] { __syncthreads(); });
}

template <int K>
__global__ void __launch_bounds__(256) mul_rate_kernel(u64* out, u64 w, int iters) {
  u64 x[K];
  for (int k = 0; k < K; k++) x[k] = threadIdx.x * 7919 + blockIdx.x + k + 1;
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int k = 0; k < K; k++) x[k] = gl64::mul(x[k], w);
  u64 acc = 0;
  for (int k = 0; k < K; k++) acc ^= x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int K, int MODE>
__global__ void __launch_bounds__(256) op_rate_kernel(u64* out, u64 w, int iters) {
  u64 x[K];
  for (int k = 0; k < K; k++) x[k] = (threadIdx.x * 7919 + blockIdx.x + k + 1) % gl64::P;
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int k = 0; k < K; k++) {
      if (MODE == 0) x[k] = gl64::add(x[k], w);
      else if (MODE == 1) x[k] = gl64::sub(x[k], w);
      else if (MODE == 2) x[k] = gl64::mul_2exp<60>(x[k]);
      else if (MODE == 3) x[k] = gl64::mul_2exp<24>(x[k]);
      else if (MODE == 4) x[k] = gl64::mul_2exp<84>(x[k]);
    }
  u64 acc = 0;
  for (int k = 0; k < K; k++) acc ^= x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) copy16_kernel(const ulonglong2* in, ulonglong2* out, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) copy8_kernel(const u64* in, u64* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

static float time_launch(std::function<void()> f, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; i++) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3f / iters;  // us
}

template <int ABL, int LOGR, int LOGC, int KIND>
static float run_abl_k(const PassDesc& ps, const TileArgs& a) {
  CK(hipFuncSetAttribute((const void*)abl_kernel<LOGR, false, ABL, LOGC, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t ldsb = ps.lds_bytes + (cfg_ldstw(LOGR, LOGC, KIND) ? ((size_t)8 << LOGR) : 0);
  return time_launch([&] { hipLaunchKernelGGL((abl_kernel<LOGR, false, ABL, LOGC, KIND>), dim3(ps.grid), dim3(ps.block), ldsb, 0, a); }, 50);
}
static bool g_generic = false;   // ubench ... with RONK_NO_CFG_KERNELS set: generic bodies everywhere
template <int ABL, int LOGR = 11>
static void run_abl(const PlanDesc& pd, int pass, TileArgs a, const char* label) {
  const PassDesc& ps = pd.passes[pass];
  float us;
  const char* k = "generic";
  // the shapes the library specialises at 2^22 (C = 8 and C = 4) and 2^16 (C = 16); ABL & 1 keeps the column pass generic
  if (!g_generic && LOGR == 11 && a.logc == 3 && tile_cfg_matches(a, 11, 3, 1)) { us = run_abl_k<ABL, LOGR == 11 ? 11 : 11, 3, 1>(ps, a); k = "cfg1"; }
  else if (!g_generic && LOGR == 11 && a.logc == 3 && tile_cfg_matches(a, 11, 3, 2)) { us = run_abl_k<ABL, 11, 3, 2>(ps, a); k = "cfg2"; }
  else if (!g_generic && LOGR == 11 && a.logc == 2 && tile_cfg_matches(a, 11, 2, 1)) { us = run_abl_k<ABL, 11, 2, 1>(ps, a); k = "cfg1"; }
  else if (!g_generic && LOGR == 11 && a.logc == 2 && tile_cfg_matches(a, 11, 2, 2)) { us = run_abl_k<ABL, 11, 2, 2>(ps, a); k = "cfg2"; }
  else if (!g_generic && LOGR == 8 && a.logc == 4 && tile_cfg_matches(a, 8, 4, 3)) { us = run_abl_k<ABL, 8, 4, 3>(ps, a); k = "cfg3"; }
  else if (!g_generic && LOGR == 8 && a.logc == 4 && tile_cfg_matches(a, 8, 4, 2)) { us = run_abl_k<ABL, 8, 4, 2>(ps, a); k = "cfg2"; }
  else us = run_abl_k<ABL, LOGR, -1, 0>(ps, a);
  printf("  pass %d  ABL=%3d  %-42s %8.2f us  [%s]\n", pass, ABL, label, us, k);
}

// single-pass batched transforms (n = 2^LOGR, the batch is the column axis): the same ablation
template <int LOGR>
static void single_pass_ablation(u64 batch, int max_logc) {
  const size_t n = (size_t)1 << LOGR, total = n * batch;
  PlanDesc pd = build_plan(LOGR, batch, false, max_logc);   // same tile rule as the library: pass max_logc = 0
  std::vector<u64> h(total);
  u64 s = 777;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl64::P; }
  u64 *d_in, *d_out, *d_wr;
  CK(hipMalloc(&d_in, total * 8)); CK(hipMalloc(&d_out, total * 8));
  CK(hipMemcpy(d_in, h.data(), total * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_wr, pd.wr[0].size() * 8)); CK(hipMemcpy(d_wr, pd.wr[0].data(), pd.wr[0].size() * 8, hipMemcpyHostToDevice));
  const PassDesc& ps = pd.passes[0];
  TileArgs a = ps.args;
  a.in = d_in; a.out = d_out; a.wr = d_wr;
  printf("== single-pass ablation, n = 2^%d x %llu polynomials, logc = %u, grid %u x %u threads, lds %zu B\n", LOGR,
         (unsigned long long)batch, ps.args.logc, ps.grid, ps.block, ps.lds_bytes);
  run_abl<0, LOGR>(pd, 0, a, "full");
  run_abl<2, LOGR>(pd, 0, a, "- round twiddles");
  run_abl<4, LOGR>(pd, 0, a, "- butterflies");
  run_abl<7, LOGR>(pd, 0, a, "- all math (loads, LDS, stores only)");
  run_abl<8, LOGR>(pd, 0, a, "- LDS exchange");
  run_abl<15, LOGR>(pd, 0, a, "- math - LDS (global loads+stores only)");
  run_abl<16, LOGR>(pd, 0, a, "- global loads");
  run_abl<32, LOGR>(pd, 0, a, "- global stores");
  run_abl<48, LOGR>(pd, 0, a, "- global loads - stores (compute+LDS only)");
  run_abl<47, LOGR>(pd, 0, a, "loads only");
  run_abl<31, LOGR>(pd, 0, a, "stores only");
  run_abl<56, LOGR>(pd, 0, a, "math only (no mem, no LDS)");
  CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(d_wr));
}

// two-pass ablation of a [batch][2^log2n] transform whose passes both have 2^LOGR rows
template <int LOGR>
static void two_pass_ablation(int log2n, u64 batch, int max_logc, u64** keep_in, u64** keep_out) {
  const size_t n = ((size_t)1 << log2n) * batch;
  PlanDesc pd = build_plan(log2n, batch, false, max_logc, 18);
  std::vector<u64> h(n);
  u64 s = 12345;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl64::P; }
  u64 *d_in, *d_tmp, *d_out;
  CK(hipMalloc(&d_in, n * 8)); CK(hipMalloc(&d_tmp, n * 8)); CK(hipMalloc(&d_out, n * 8));
  CK(hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tmp, h.data(), n * 8, hipMemcpyHostToDevice));
  std::vector<u64*> d_wr;
  for (auto& t : pd.wr) { u64* d; CK(hipMalloc(&d, t.size() * 8)); CK(hipMemcpy(d, t.data(), t.size() * 8, hipMemcpyHostToDevice)); d_wr.push_back(d); }
  std::vector<std::pair<u64*, u64*>> d_tw;
  for (auto& t : pd.tw) {
    u64 *lo, *hi;
    CK(hipMalloc(&lo, t.lo.size() * 8)); CK(hipMemcpy(lo, t.lo.data(), t.lo.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&hi, t.hi.size() * 8)); CK(hipMemcpy(hi, t.hi.data(), t.hi.size() * 8, hipMemcpyHostToDevice));
    d_tw.push_back({lo, hi});
  }
  printf("== ablation, n = 2^%d x %llu, logc = %u / %u, grid %u x %u threads, lds %zu B\n", log2n, (unsigned long long)batch, pd.passes[0].args.logc,
         pd.passes[1].args.logc, pd.passes[0].grid, pd.passes[0].block, pd.passes[0].lds_bytes);
  for (int pass = 0; pass < 2; pass++) {
    const PassDesc& ps = pd.passes[pass];
    TileArgs a = ps.args;
    a.in = pass == 0 ? d_in : d_tmp;
    a.out = pass == 0 ? d_tmp : d_out;
    a.wr = d_wr[ps.wr_id];
    if (ps.tw_id >= 0) { a.tw_lo = d_tw[ps.tw_id].first; a.tw_hi = d_tw[ps.tw_id].second; }
    if (ps.twf_id >= 0) {
      u64* d; CK(hipMalloc(&d, pd.twf[ps.twf_id].size() * 8));
      CK(hipMemcpy(d, pd.twf[ps.twf_id].data(), pd.twf[ps.twf_id].size() * 8, hipMemcpyHostToDevice));
      a.tw_full = d;
    }
    run_abl<0, LOGR>(pd, pass, a, "full");
    run_abl<1, LOGR>(pd, pass, a, "- inter-pass twiddle");
    run_abl<2, LOGR>(pd, pass, a, "- round twiddles");
    run_abl<3, LOGR>(pd, pass, a, "- all twiddles");
    run_abl<4, LOGR>(pd, pass, a, "- butterflies");
    run_abl<7, LOGR>(pd, pass, a, "- all math (loads, LDS, stores only)");
    run_abl<8, LOGR>(pd, pass, a, "- LDS exchange");
    run_abl<15, LOGR>(pd, pass, a, "- math - LDS (global loads+stores only)");
    run_abl<16, LOGR>(pd, pass, a, "- global loads");
    run_abl<32, LOGR>(pd, pass, a, "- global stores");
    run_abl<48, LOGR>(pd, pass, a, "- global loads - stores (compute+LDS only)");
    run_abl<47, LOGR>(pd, pass, a, "loads only");
    run_abl<31, LOGR>(pd, pass, a, "stores only");
    run_abl<56, LOGR>(pd, pass, a, "math only (no mem, no LDS)");
    run_abl<64, LOGR>(pd, pass, a, "full, twiddle values without table loads");
    run_abl<120, LOGR>(pd, pass, a, "math only, no twiddle loads");
  }
  *keep_in = d_in; *keep_out = d_out;
}

// staggered start: the second workgroup of a CU (HW_ID.tg_id odd) sleeps `delay` x 64 cycles before it touches memory, so
// that its load/store phases fall under the other workgroup's arithmetic (C = 4 tiles: two workgroups per CU)
template <int LOGR, int LOGC, int KIND>
__global__ void __launch_bounds__(1024) stagger_kernel(const TileArgs a, int delay, u32* hist) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 nb = gridDim.x, b = blockIdx.x;
  const u32 q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  const u32 bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const u32 tg = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4);   // HW_REG_HW_ID[19:16] = workgroup slot on the CU
  if (hist && threadIdx.x == 0) atomicAdd(&hist[tg & 15], 1u);
  if ((tg & 1) && delay > 0)
    for (int i = 0; i < delay; i++) __builtin_amdgcn_s_sleep(1);
  tile_body<LOGR, false, 0, TileCfg<LOGC, KIND>>(a, lds, threadIdx.x, bid, [] { __syncthreads(); });
}
static void stagger_sweep() {
  u64 *d_in = nullptr, *d_out = nullptr;
  PlanDesc pd = build_plan(22, 1, false, 2, 18);
  const size_t n = (size_t)1 << 22;
  u64 *din, *dtmp, *dout; CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&dtmp, n * 8)); CK(hipMalloc(&dout, n * 8));
  CK(hipMemset(din, 1, n * 8)); CK(hipMemset(dtmp, 1, n * 8));
  std::vector<u64*> d_wr;
  for (auto& t : pd.wr) { u64* d; CK(hipMalloc(&d, t.size() * 8)); CK(hipMemcpy(d, t.data(), t.size() * 8, hipMemcpyHostToDevice)); d_wr.push_back(d); }
  u64 *lo, *hi;
  CK(hipMalloc(&lo, pd.tw[0].lo.size() * 8)); CK(hipMemcpy(lo, pd.tw[0].lo.data(), pd.tw[0].lo.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&hi, pd.tw[0].hi.size() * 8)); CK(hipMemcpy(hi, pd.tw[0].hi.data(), pd.tw[0].hi.size() * 8, hipMemcpyHostToDevice));
  u32* hist; CK(hipMalloc(&hist, 64)); CK(hipMemset(hist, 0, 64));
  CK(hipFuncSetAttribute((const void*)stagger_kernel<11, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)stagger_kernel<11, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  TileArgs a0 = pd.passes[0].args, a1 = pd.passes[1].args;
  a0.in = din; a0.out = dtmp; a0.wr = d_wr[pd.passes[0].wr_id]; a0.tw_lo = lo; a0.tw_hi = hi;
  a1.in = dtmp; a1.out = dout; a1.wr = d_wr[pd.passes[1].wr_id];
  printf("== staggered start, 2^22, C = 4 tiles (grid %u x %u threads, %zu B LDS): delay in units of 64 cycles\n", pd.passes[0].grid, pd.passes[0].block, pd.passes[0].lds_bytes);
  hipLaunchKernelGGL((stagger_kernel<11, 2, 1>), dim3(pd.passes[0].grid), dim3(pd.passes[0].block), pd.passes[0].lds_bytes, 0, a0, 0, hist);
  CK(hipDeviceSynchronize());
  u32 h[16]; CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
  printf("  tg_id histogram:"); for (int i = 0; i < 16; i++) printf(" %u", h[i]); printf("\n");
  for (int delay : {0, 20, 40, 60, 80, 100, 140, 180}) {
    float p0 = time_launch([&] { hipLaunchKernelGGL((stagger_kernel<11, 2, 1>), dim3(pd.passes[0].grid), dim3(pd.passes[0].block), pd.passes[0].lds_bytes, 0, a0, delay, (u32*)nullptr); }, 50);
    float p1 = time_launch([&] { hipLaunchKernelGGL((stagger_kernel<11, 2, 2>), dim3(pd.passes[1].grid), dim3(pd.passes[1].block), pd.passes[1].lds_bytes, 0, a1, delay, (u32*)nullptr); }, 50);
    float both = time_launch([&] {
      hipLaunchKernelGGL((stagger_kernel<11, 2, 1>), dim3(pd.passes[0].grid), dim3(pd.passes[0].block), pd.passes[0].lds_bytes, 0, a0, delay, (u32*)nullptr);
      hipLaunchKernelGGL((stagger_kernel<11, 2, 2>), dim3(pd.passes[1].grid), dim3(pd.passes[1].block), pd.passes[1].lds_bytes, 0, a1, delay, (u32*)nullptr); }, 50);
    printf("  delay %3d (%.2f us)  pass 0 %7.2f  pass 1 %7.2f  transform %7.2f us\n", delay, delay * 64 / 2400.0, p0, p1, both);
  }
  (void)d_in; (void)d_out;
}

// per-wave issue rate: the math-only body (ABL = 56) with 1 / 2 / 4 waves per SIMD -- one workgroup per CU (LDS request
// forced to 136 KiB), 256 workgroups; every lane does the same work, so time ~ waves/SIMD when the VALU is saturated and
// constant when a wave is latency-bound.
template <int ABL>
static void occupancy_sweep(const char* what) {
  PlanDesc pd = build_plan(22, 1, false, 3, 18);
  u64* d; CK(hipMalloc(&d, (size_t)8 << 22)); CK(hipMemset(d, 1, (size_t)8 << 22));
  CK(hipFuncSetAttribute((const void*)abl_kernel<11, false, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int pass = 0; pass < 2; pass++)
    for (int logc = 1; logc <= 3; logc++) {
      TileArgs a = pd.passes[pass].args;
      a.in = d; a.out = d; a.wr = d; a.tw_lo = d; a.tw_hi = d;
      a.logc = (u32)logc; a.tiles = 256;
      const u32 block = (2048u << logc) / 16;
      float us = time_launch([&] { hipLaunchKernelGGL((abl_kernel<11, false, ABL>), dim3(256), dim3(block), 139264, 0, a); }, 20);
      printf("  %s pass %d: %d wave(s)/SIMD (block %4u)  %8.2f us  -> %.2f us per wave/SIMD\n", what, pass, (int)block / 256, block, us, us / (block / 256));
    }
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  g_generic = getenv("RONK_NO_CFG_KERNELS") != nullptr;
  int max_logc = argc > 1 ? atoi(argv[1]) : 4;
  if (argc == 2 && atoi(argv[1]) == -2) { stagger_sweep(); return 0; }
  if (argc == 2 && atoi(argv[1]) == -1) { occupancy_sweep<56>("math only"); occupancy_sweep<120>("math only, no table loads"); return 0; }
  if (argc == 3) {   // ubench <max_logc> <log2n in {8, 10, 12}>: single-pass batched ablation over 2^24 coefficients
    const int k = atoi(argv[2]);
    if (k == 4) single_pass_ablation<4>(1048576, max_logc);
    else if (k == 6) single_pass_ablation<6>(262144, max_logc);
    else if (k == 12) single_pass_ablation<12>(4096, max_logc);
    else if (k == 10) single_pass_ablation<10>(16384, max_logc);
    else single_pass_ablation<8>(65536, max_logc);
    return 0;
  }
  u64 *d_in = nullptr, *d_out = nullptr;
  if (argc > 3) {   // ubench <max_logc> 16 1024: the two LOGR = 8 passes of the batched 2^16 transform (config 4)
    two_pass_ablation<8>(16, (u64)atoll(argv[3]), max_logc, &d_in, &d_out);
    return 0;
  }
  const size_t n = (size_t)1 << 22;
  two_pass_ablation<11>(22, 1, max_logc, &d_in, &d_out);
  // ---- raw op rates
  u64* d_scr;
  CK(hipMalloc(&d_scr, 256 * 2048 * 8));
  const int iters = 512;
  const double lanes = 256.0 * 2048;
  auto rate = [&](const char* name, float us, int K) {
    double ops = lanes * K * iters;
    printf("  %-28s %8.2f us  %8.1f Gop/s  (%.2f cycles/op/lane-slot at 2.4 GHz x 256 CU x 128 lanes/clk)\n", name, us, ops / us * 1e-3,
           256.0 * 128 * 2.4e9 / (ops / (us * 1e-6)));
  };
  printf("== op rates (8 independent chains per work-item)\n");
  rate("gl64::mul", time_launch([&] { hipLaunchKernelGGL((mul_rate_kernel<8>), dim3(2048), dim3(256), 0, 0, d_scr, (u64)0x123456789abcdefull, iters); }, 10), 8);
  rate("gl64::add", time_launch([&] { hipLaunchKernelGGL((op_rate_kernel<8, 0>), dim3(2048), dim3(256), 0, 0, d_scr, (u64)0x123456789abcdefull, iters); }, 10), 8);
  rate("gl64::sub", time_launch([&] { hipLaunchKernelGGL((op_rate_kernel<8, 1>), dim3(2048), dim3(256), 0, 0, d_scr, (u64)0x123456789abcdefull, iters); }, 10), 8);
  rate("mul_2exp<60>", time_launch([&] { hipLaunchKernelGGL((op_rate_kernel<8, 2>), dim3(2048), dim3(256), 0, 0, d_scr, (u64)1, iters); }, 10), 8);
  rate("mul_2exp<24>", time_launch([&] { hipLaunchKernelGGL((op_rate_kernel<8, 3>), dim3(2048), dim3(256), 0, 0, d_scr, (u64)1, iters); }, 10), 8);
  rate("mul_2exp<84>", time_launch([&] { hipLaunchKernelGGL((op_rate_kernel<8, 4>), dim3(2048), dim3(256), 0, 0, d_scr, (u64)1, iters); }, 10), 8);
  // ---- memory floors (32 MiB in -> 32 MiB out, cache-resident after the first iteration)
  printf("== copy 32 MiB -> 32 MiB\n");
  for (int g : {1024, 2048, 4096, 8192}) {
    float us = time_launch([&] { hipLaunchKernelGGL(copy16_kernel, dim3(g), dim3(256), 0, 0, (const ulonglong2*)d_in, (ulonglong2*)d_out, n / 2); }, 20);
    printf("  copy16 grid %5d  %8.2f us  %7.1f GB/s\n", g, us, 2.0 * n * 8 / us * 1e-3);
    us = time_launch([&] { hipLaunchKernelGGL(copy8_kernel, dim3(g), dim3(256), 0, 0, d_in, d_out, n); }, 20);
    printf("  copy8  grid %5d  %8.2f us  %7.1f GB/s\n", g, us, 2.0 * n * 8 / us * 1e-3);
  }
  return 0;
}
