// Host check of ronkathon_amd/csrc/gl64.h (the device field arithmetic, compiled for the host) against
// 128-bit reference arithmetic, on edge values chosen to hit every carry / borrow / wrap branch, and on
// random values.  TEST INFRASTRUCTURE; built by tests/test_emu_kernel.py with g++ (portable fallback path)
// and, when present, ROCm's clang++ (the __builtin_subc path the device build uses).
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../ronkathon_amd/csrc/gl64.h"

typedef unsigned __int128 u128;
using gl64::P;
using gl64::u64;
static int fails = 0;
#define EXPECT(got, want, what, a, b)                                                              \
  do {                                                                                             \
    u64 g_ = (got), w_ = (want);                                                                   \
    if (g_ != w_ && fails++ < 10) printf("FAIL %s a=%llx b=%llx got=%llx want=%llx\n", what,        \
                                         (unsigned long long)(a), (unsigned long long)(b),         \
                                         (unsigned long long)g_, (unsigned long long)w_);          \
  } while (0)

static u64 ref_mul(u64 a, u64 b) { return (u64)(((u128)a * b) % P); }
static u64 ref_pow2(int k) { u64 r = 1; for (int i = 0; i < k; i++) r = (u64)(((u128)r * 2) % P); return r; }

template <int K>
static void check_shift(const std::vector<u64>& vals, const std::vector<u64>& any64) {
  const u64 c = ref_pow2(K);
  for (u64 x : vals) EXPECT(gl64::mul_2exp<K>(x), ref_mul(x, c), "mul_2exp", x, K);
  if (K > 0)  // K == 0 is the identity and keeps whatever representative it is given
    for (u64 x : any64) EXPECT(gl64::mul_2exp<K>(x), ref_mul(x % P, c), "mul_2exp(noncanonical)", x, K);
}
template <int... Ks>
static void check_all_shifts(std::integer_sequence<int, Ks...>, const std::vector<u64>& v, const std::vector<u64>& w) {
  (check_shift<Ks>(v, w), ...);
}

int main() {
  std::vector<u64> e = {0, 1, 2, 3, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0x100000000ull,
                        0x100000001ull, 0x1FFFFFFFFull, 0xFFFFFFFF00000000ull >> 1, 0x7FFFFFFFFFFFFFFFull,
                        0x8000000000000000ull, 0xFFFFFFFE00000000ull, 0xFFFFFFFE00000001ull, 0xFFFFFFFEFFFFFFFFull,
                        0xFFFFFFFF00000000ull, P - 1, P - 2, P - 0xFFFFFFFFull, P - 0x100000000ull, P / 2, P / 2 + 1,
                        0x0000FFFF00000001ull, 0xFFFF0000FFFF0000ull, 0x00000001FFFFFFFFull, 0xFFFFFFFDFFFFFFFFull};
  std::vector<u64> any64 = {P, P + 1, P + 0xFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF80000000ull};
  u64 s = 88172645463325252ull;
  std::vector<u64> rnd;
  for (int i = 0; i < 2000; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; rnd.push_back(s % P); }
  std::vector<u64> vals = e;
  vals.insert(vals.end(), rnd.begin(), rnd.begin() + 200);
  for (u64 a : vals)
    for (u64 b : vals) {
      EXPECT(gl64::add(a, b), (u64)(((u128)a + b) % P), "add", a, b);
      EXPECT(gl64::sub(a, b), (u64)(((u128)a + P - b) % P), "sub", a, b);
      EXPECT(gl64::mul(a, b), ref_mul(a, b), "mul", a, b);
    }
  for (u64 a : vals) {
    EXPECT(gl64::neg(a), (P - a) % P, "neg", a, 0);
    EXPECT(gl64::canon(a), a, "canon", a, 0);
    if (a) EXPECT(gl64::mul(a, gl64::inv(a)), 1, "inv", a, 0);
  }
  for (u64 x : any64) EXPECT(gl64::canon(x), x % P, "canon(any)", x, 0);
  // sub() with an arbitrary 64-bit minuend and a subtrahend <= p returns SOME representative
  for (u64 a : any64) for (u64 b : e) EXPECT(gl64::sub(a, b) % P, (u64)(((u128)(a % P) + P - b) % P), "sub(any)", a, b);
  // reduce128 on extreme 128-bit inputs
  for (u64 lo : e) for (u64 hi : e) EXPECT(gl64::reduce128(lo, hi), (u64)((((u128)hi << 64) | lo) % P), "reduce128", lo, hi);
  for (u64 lo : any64) for (u64 hi : any64) EXPECT(gl64::reduce128(lo, hi), (u64)((((u128)hi << 64) | lo) % P), "reduce128", lo, hi);
  check_all_shifts(std::make_integer_sequence<int, 96>{}, vals, any64);
  EXPECT(gl64::pow(7, (P - 1) / 64), ref_pow2(39), "omega_64 == 2^39", 0, 0);
  printf(fails ? "FAILED %d\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
