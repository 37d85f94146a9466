// field_policy.h -- the field a tile pass computes in, as a compile-time policy of the NTT bodies (ntt_tile.h, ntt_small.h).
//
// ronkathon's field and transform are generic over the modulus: `PrimeField<const P: usize>`
// (src/algebra/field/prime/mod.rs:39-52), `Polynomial<B, F, D>::fft / ifft` for any `F: FiniteField`
// (src/polynomial/mod.rs:273-323, :430-484).  The same tile kernels therefore exist for two arithmetic families:
//
//   GlField    Goldilocks, p = 2^64 - 2^32 + 1 (gl64.h): reduction by 2^64 = 2^32 - 1, and -- with the reference's root
//              convention omega_64 = 7^((p-1)/64) = 2^39 -- every twiddle inside a 16-point register round is a shift.
//   MontField  ANY odd prime p < 2^64 whose multiplicative group has the power-of-two roots the transform needs
//              (Montgomery form, R = 2^64, mont64.h).  Data stay CANONICAL in HBM, LDS and registers; every table entry
//              (round twiddles, inter-pass twiddles, the folded n^-1, the eight roots of a 16-point round) is stored as
//              w * R mod p, so that mmul(x, wR) = x * w mod p goes canonical -> canonical with ONE Montgomery product and
//              no conversion pass on either side of a transform ("Montgomery without conversion
//              passes").  p > 2^63 is allowed: sums carry into a 65th bit that add / sub / redc fold explicitly.
//
// A policy object is built from TileArgs::fc (kernel arguments: wave-uniform, SGPRs).  `mul(x, w)`: w in TABLE form
// (canonical for Goldilocks, w * R for Montgomery).  `mul_plain(x, y)`: both canonical (the fused pointwise product).
// Plain C++, so the host emulator under tests/emu runs the same code.
#pragma once
#include "gl64.h"
#include "mont64.h"

namespace ronk {

typedef uint64_t u64;
typedef uint32_t u32;

// What a Montgomery pass knows about its prime.  p == 0: Goldilocks (nothing else is read).
struct FieldConst {
  u64 p;        // odd modulus
  u64 pinv;     // -p^-1 mod 2^64
  u64 r2;       // 2^128 mod p
  u64 w16[8];   // omega_16^j * 2^64 mod p, j = 0..7, in the direction of the pass (inverse plans: omega_16^-j)
};

// exponent E with omega_N^j == 2^E (mod Goldilocks), N | 64; inverse direction uses omega^-1
constexpr int root_exp(int n, int j, bool inv) {
  int e = (39 * (64 / n) * j) % 192;
  return inv ? (192 - e) % 192 : e;
}

struct GlField {
  static constexpr bool MONT = false;
  RONK_HD GlField() {}
  RONK_HD explicit GlField(const FieldConst&) {}
  RONK_HD u64 add(u64 a, u64 b) const { return gl64::add(a, b); }
  RONK_HD u64 add_lazy(u64 a, u64 b) const { return gl64::add_lazy(a, b); }
  RONK_HD u64 sub(u64 a, u64 b) const { return gl64::sub(a, b); }
  RONK_HD u64 mul(u64 x, u64 w) const { return gl64::mul(x, w); }
  RONK_HD u64 mul_plain(u64 x, u64 y) const { return gl64::mul(x, y); }
  // (a - b) * omega_N^J
  template <int N, int J, bool INV>
  RONK_HD u64 sub_mul_root(u64 a, u64 b) const {
    constexpr int E = root_exp(N, J, INV);
    if constexpr (E >= 96) {
      return gl64::mul_2exp<E - 96>(gl64::sub(b, a));  // omega = -2^(E-96)
    } else {
      return gl64::mul_2exp<E>(gl64::sub(a, b));
    }
  }
};

struct MontField {
  static constexpr bool MONT = true;
  mont64::Field f;
  u64 w16[8];
  RONK_HD explicit MontField(const FieldConst& c) {
    f.p = c.p; f.pinv = c.pinv; f.r2 = c.r2; f.one = c.w16[0];
#pragma unroll
    for (int j = 0; j < 8; j++) w16[j] = c.w16[j];
  }
  RONK_HD u64 add(u64 a, u64 b) const { return mont64::add(f, a, b); }
  // (a "lazy" sum -- only the wrap folded back, as for Goldilocks -- costs the same six instructions as the canonical one here)
  RONK_HD u64 add_lazy(u64 a, u64 b) const { return mont64::add(f, a, b); }
  RONK_HD u64 sub(u64 a, u64 b) const { return mont64::sub(f, a, b); }
  RONK_HD u64 mul(u64 x, u64 w) const { return mont64::mmul(f, x, w); }
  RONK_HD u64 mul_plain(u64 x, u64 y) const { return mont64::mmul(f, mont64::mmul(f, x, y), f.r2); }
  template <int N, int J, bool INV>
  RONK_HD u64 sub_mul_root(u64 a, u64 b) const {
    const u64 d = mont64::sub(f, a, b);
    if constexpr (J == 0) return d;
    else return mont64::mmul(f, d, w16[J * (16 / N)]);   // the table carries the direction
  }
};

}  // namespace ronk
