// tools/pcie_probe.hip -- developer tool: what the host-pointer entry points can expect from the PCIe link on this box.
// Pageable vs pinned copies, in-place registration, CPU staging copies, duplex, strided (2-D) copies.  32 MiB = one 2^22 polynomial.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double best(int reps, F f) { double b = 1e9; for (int i = 0; i < reps; i++) { double t = now(); f(); t = now() - t; if (t < b) b = t; } return b; }
int main() {
  const size_t N = 32u << 20;
  char* pg = (char*)malloc(N); char* pg2 = (char*)malloc(N);
  memset(pg, 1, N); memset(pg2, 2, N);
  char *pin, *pin2; CK(hipHostMalloc((void**)&pin, N)); CK(hipHostMalloc((void**)&pin2, N));
  memset(pin, 3, N); memset(pin2, 4, N);
  char *d, *d2; CK(hipMalloc((void**)&d, N)); CK(hipMalloc((void**)&d2, N));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  auto gb = [&](double t) { return N / t / 1e9; };
  double t;
  t = best(5, [&] { CK(hipMemcpy(d, pg, N, hipMemcpyHostToDevice)); });  printf("pageable H2D            %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  t = best(5, [&] { CK(hipMemcpy(pg2, d, N, hipMemcpyDeviceToHost)); }); printf("pageable D2H            %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  t = best(5, [&] { CK(hipMemcpy(d, pin, N, hipMemcpyHostToDevice)); }); printf("pinned   H2D            %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  t = best(5, [&] { CK(hipMemcpy(pin2, d, N, hipMemcpyDeviceToHost)); });printf("pinned   D2H            %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  t = best(5, [&] { CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin2, d2, N, hipMemcpyDeviceToHost, s2));
                    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
  printf("pinned   H2D || D2H     %7.3f ms  %6.1f GB/s each way\n", t * 1e3, gb(t));
  t = best(5, [&] { CK(hipHostRegister(pg, N, hipHostRegisterDefault)); CK(hipHostUnregister(pg)); });
  printf("register + unregister   %7.3f ms\n", t * 1e3);
  CK(hipHostRegister(pg, N, hipHostRegisterDefault));
  t = best(5, [&] { CK(hipMemcpy(d, pg, N, hipMemcpyHostToDevice)); }); printf("registered H2D          %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  CK(hipHostUnregister(pg));
  t = best(5, [&] { memcpy(pin, pg, N); }); printf("memcpy pageable->pinned, 1 thread %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  for (int nt : {2, 4, 8}) {
    t = best(5, [&] { std::vector<std::thread> th; for (int i = 0; i < nt; i++) th.emplace_back([&, i] { memcpy(pin + N / nt * i, pg + N / nt * i, N / nt); }); for (auto& x : th) x.join(); });
    printf("memcpy pageable->pinned, %d threads %7.3f ms  %6.1f GB/s\n", nt, t * 1e3, gb(t));
  }
  // chunked: 8 x 4 MiB, CPU copy of chunk i+1 while chunk i travels (1 thread)
  t = best(5, [&] { const size_t C = N / 8; for (int i = 0; i < 8; i++) { memcpy(pin + C * i, pg + C * i, C); CK(hipMemcpyAsync(d + C * i, pin + C * i, C, hipMemcpyHostToDevice, s1)); } CK(hipStreamSynchronize(s1)); });
  printf("staged H2D, 8 chunks, 1 thread   %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  t = best(5, [&] { const size_t C = N / 8; for (int i = 0; i < 8; i++) { std::thread a([&] { memcpy(pin + C * i, pg + C * i, C / 2); }); memcpy(pin + C * i + C / 2, pg + C * i + C / 2, C / 2); a.join();
                    CK(hipMemcpyAsync(d + C * i, pin + C * i, C, hipMemcpyHostToDevice, s1)); } CK(hipStreamSynchronize(s1)); });
  printf("staged H2D, 8 chunks, 2 threads  %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  // strided: 2048 rows x 2 KiB segments out of a [2048][16 KiB] matrix (one column chunk of the 2^22 transform's [A][B] view)
  t = best(5, [&] { CK(hipMemcpy2DAsync(d, 2048, pin, 16384, 2048, 2048, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
  printf("pinned 2-D H2D 2048 x 2 KiB (4 MiB) %7.3f ms  %6.1f GB/s\n", t * 1e3, 4194304.0 / t / 1e9);
  t = best(5, [&] { CK(hipMemcpy2DAsync(d, 16384, pin, 16384, 16384, 2048, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
  printf("pinned 2-D H2D 2048 x 16 KiB (32 MiB, contiguous) %7.3f ms  %6.1f GB/s\n", t * 1e3, gb(t));
  t = best(5, [&] { CK(hipMemcpyAsync(d, pin, 4 << 20, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
  printf("pinned 1-D H2D 4 MiB     %7.3f ms  %6.1f GB/s\n", t * 1e3, 4194304.0 / t / 1e9);
  t = best(5, [&] { CK(hipMemcpyAsync(d, pin, 1 << 20, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
  printf("pinned 1-D H2D 1 MiB     %7.3f ms  %6.1f GB/s\n", t * 1e3, 1048576.0 / t / 1e9);
  t = best(5, [&] { CK(hipHostRegister(pg, 4 << 20, hipHostRegisterDefault)); CK(hipHostUnregister(pg)); });
  printf("register + unregister 4 MiB  %7.3f ms\n", t * 1e3);
  return 0;
}
