// tools/stride_probe.hip -- developer tool: bandwidth of the column-pass access pattern as a function of the row stride.
// A workgroup of 1024 lanes reads a tile of ROWS rows x 128 bytes (16 lanes x 8 bytes per row; the 64 rows per wave-instruction
// group are spread like the tile kernel's: lane = (row_in_group, column)), rows `stride` bytes apart, tiles 128 bytes apart along a
// row -- the read side of pass 1 of a three-pass NTT plan -- and writes the tile either with the same pattern or contiguously.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int MODE>   // 0: strided read + strided write, 1: strided read only, 2: strided write only, 3: contiguous read + write
__global__ void __launch_bounds__(1024) probe(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                              size_t stride_e, int rows, size_t tiles_per_row) {
  const unsigned tid = threadIdx.x, c = tid & 15, m = tid >> 4;          // 64 row-lanes x 16 columns
  const size_t t = blockIdx.x % tiles_per_row, slab = blockIdx.x / tiles_per_row;
  const size_t base = slab * (size_t)rows * stride_e + t * 16 + c;
  unsigned long long x[16];
  const int per = rows / 64;                                               // loads per lane (rows = 64 * per)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (i < per) {
      const size_t row = (size_t)i * 64 + m;
      if (MODE == 3) x[i] = in[(size_t)blockIdx.x * rows * 16 + (size_t)i * 1024 + tid];
      else if (MODE == 2) x[i] = tid + i;
      else x[i] = in[base + row * stride_e];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (i < per) {
      const size_t row = (size_t)i * 64 + m;
      x[i] = x[i] * 3 + 1;
      if (MODE == 3 || MODE == 1) out[(size_t)blockIdx.x * rows * 16 + (size_t)i * 1024 + tid] = x[i];
      else out[base + row * stride_e] = x[i];
    }
  }
}
int main(int argc, char** argv) {
  const size_t total_e = (size_t)1 << 24;                                  // 128 MiB in, 128 MiB out
  unsigned long long *in, *out;
  CK(hipMalloc((void**)&in, total_e * 24)); CK(hipMalloc((void**)&out, total_e * 24));
  CK(hipMemset(in, 1, total_e * 8)); CK(hipMemset(out, 0, total_e * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rows = 512;
  printf("rows %d, 128-byte row segments, 2^24 elements each way; GB/s counts read + written bytes of the mode\n", rows);
  for (size_t row_elems : {(size_t)2048, (size_t)8192, (size_t)32768, (size_t)32768 + 16, (size_t)32768 + 32, (size_t)32768 + 512, (size_t)65536, (size_t)65536 + 16}) {
    // matrix [rows][row_elems] slabs; tiles of 16 columns; use as many slabs as fit 2^24 elements
    const size_t tiles_per_row = (row_elems >= 32768 ? 32768 : row_elems) / 16;   // touch 32768 columns of a row at most (the padded variants leave gaps)
    size_t slabs = total_e / ((size_t)rows * row_elems);
    if (slabs < 1) slabs = 1;
    const size_t grid = tiles_per_row * slabs;
    const double bytes1 = (double)grid * rows * 128;
    for (int mode = 0; mode < 4; mode++) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(1024), 0, 0, in, out, row_elems, rows, tiles_per_row);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(1024), 0, 0, in, out, row_elems, rows, tiles_per_row);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(1024), 0, 0, in, out, row_elems, rows, tiles_per_row);
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(1024), 0, 0, in, out, row_elems, rows, tiles_per_row);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double moved = bytes1 * ((mode == 0 || mode == 3) ? 2 : (mode == 1 ? 2 : 1));
      printf("row stride %8zu B  %-28s grid %6zu  %8.1f us  %7.1f GB/s\n", row_elems * 8,
             mode == 0 ? "strided read + strided write" : mode == 1 ? "strided read, contiguous write" : mode == 2 ? "strided write only" : "contiguous read + write",
             grid, best * 1e3, moved / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
