// tools/gridbar_probe.hip -- developer probe (round 4): what does a grid-wide barrier cost against a kernel boundary?
// Question behind it: a 2^16 transform is two launches of 64 workgroups, ~5.3 us each in a dependent chain (config 2); would
// ONE launch with a spin barrier between the passes be faster, and does it matter whether the 64 workgroups sit on one XCD
// (exchange through that XCD's L2) or on all eight?
//   chain     N dependent launches of a 64-workgroup kernel that touches 8 bytes per lane           -> us per launch
//   bar_all   one launch, 64 workgroups over all XCDs, K barriers (agent-scope atomic + spin)        -> us per barrier
//   bar_xcd   one launch of 512 workgroups of which the 64 with blockIdx % 8 == 0 take part (XCD 0)  -> us per barrier
// Every variant also moves 8 bytes per lane between the barriers (write own slot, read a partner's) and checks the value.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint64_t u64; typedef uint32_t u32;

__global__ void __launch_bounds__(256) chain_kernel(u64* buf, u32 step) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  const u32 j = (i + 4099u * step) & (64 * 256 - 1);   // partner written by the previous launch
  const u64 v = buf[(step & 1) * 16384 + j];
  buf[((step + 1) & 1) * 16384 + i] = v + 1;
}

__device__ __forceinline__ u32 xcc_id() { u32 v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }

// MODE 0: all workgroups take part; MODE 1: only blockIdx % 8 == 0 (rank = blockIdx / 8)
template <int MODE>
__global__ void __launch_bounds__(256) bar_kernel(u64* buf, u32* ctr, u32 base, u32 nbar, u32* xcc_seen, u64* stamps) {
  if (MODE == 1 && (blockIdx.x & 7)) return;
  const u32 rank = MODE == 1 ? blockIdx.x >> 3 : blockIdx.x;
  const u32 nwg = 64;
  const u32 i = rank * 256 + threadIdx.x;
  if (threadIdx.x == 0) xcc_seen[rank] = xcc_id();
  u64 v = buf[i];
  u64 t0 = 0;
  if (threadIdx.x == 0) t0 = __builtin_readcyclecounter();
  for (u32 b = 0; b < nbar; b++) {
    // publish, barrier, read the partner's value
    __hip_atomic_store(&buf[((b + 1) & 1) * 16384 + i], v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const u32 target = base + nwg * (b + 1);
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const u32 j = (i + 4099u * (b + 1)) & (64 * 256 - 1);
    v = __hip_atomic_load(&buf[((b + 1) & 1) * 16384 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) stamps[rank] = __builtin_readcyclecounter() - t0;
  buf[2 * 16384 + i] = v;
}

int main(int argc, char** argv) {
  const int nbar = argc > 1 ? atoi(argv[1]) : 64;
  u64* buf; u32 *ctr, *xcc; u64* stamps;
  CK(hipMalloc(&buf, 3 * 16384 * 8)); CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&xcc, 64 * 4)); CK(hipMalloc(&stamps, 64 * 8));
  CK(hipMemset(buf, 0, 3 * 16384 * 8)); CK(hipMemset(ctr, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  // chain
  for (int rep = 0; rep < 3; rep++) {
    const int N = 2000;
    CK(hipMemset(buf, 0, 3 * 16384 * 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int s = 0; s < N; s++) hipLaunchKernelGGL(chain_kernel, dim3(64), dim3(256), 0, 0, buf, (u32)s);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<u64> h(16384); CK(hipMemcpy(h.data(), buf + (N & 1) * 16384, 16384 * 8, hipMemcpyDeviceToHost));
    bool ok = true; for (u64 x : h) ok = ok && x == (u64)N;
    printf("chain: %d dependent launches of 64 workgroups: %.2f us per launch (%s)\n", N, ms * 1e3 / N, ok ? "values ok" : "VALUES WRONG");
  }
  u32 base = 0;
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemset(buf, 0, 3 * 16384 * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(64), dim3(256), 0, 0, buf, ctr, base, (u32)nbar, xcc, stamps);
      else hipLaunchKernelGGL(bar_kernel<1>, dim3(512), dim3(256), 0, 0, buf, ctr, base, (u32)nbar, xcc, stamps);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      base += 64 * nbar;
      std::vector<u64> h(16384); CK(hipMemcpy(h.data(), buf + 2 * 16384, 16384 * 8, hipMemcpyDeviceToHost));
      bool ok = true; for (u64 x : h) ok = ok && x == (u64)nbar;
      std::vector<u32> hx(64); CK(hipMemcpy(hx.data(), xcc, 256, hipMemcpyDeviceToHost));
      std::vector<u64> hs(64); CK(hipMemcpy(hs.data(), stamps, 512, hipMemcpyDeviceToHost));
      int cnt[16] = {0}; for (u32 x : hx) cnt[x & 15]++;
      u64 mx = 0; for (u64 x : hs) mx = x > mx ? x : mx;
      printf("%s: %d barriers in one launch: %.2f us per barrier incl. launch (%.2f us whole launch), %.0f shader cycles per barrier; XCDs used:",
             mode ? "bar_xcd" : "bar_all", nbar, ms * 1e3 / nbar, ms * 1e3, (double)mx / nbar);
      for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
      printf(" (%s)\n", ok ? "values ok" : "VALUES WRONG");
    }
  }
  return 0;
}
