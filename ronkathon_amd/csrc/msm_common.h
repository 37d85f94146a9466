// msm_common.h -- the parts of the bucket-method MSM (msm_kernels.h) that are plain C++: the window shape, the signed-digit
// recoding of a scalar and the host tail.  Shared by the kernels, by ronk_msm.hip and by the host-compiled check of the
// algorithm (tests/emu/bn254_host.cpp, test infrastructure).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "bn254.h"

namespace ronk {

typedef uint64_t u64;
typedef uint32_t u32;

struct MsmShape {
  u32 n;        // points
  u32 c;        // window bits
  u32 W;        // windows, W*c >= 257
  u32 NB;       // buckets per window = 2^(c-1), weights 1 .. NB
};

// signed digit w of a 256-bit scalar (4 x u64 little endian) given the carry from the digit below; updates the carry
RONK_HD int msm_digit(const u64* k, u32 w, u32 c, u32* carry) {
  const u32 bit = w * c;
  u32 raw = 0;
  if (bit < 256) {
    const u32 word = bit >> 6, off = bit & 63;
    u64 v = k[word] >> off;
    if (off + c > 64 && word + 1 < 4) v |= k[word + 1] << (64 - off);
    raw = (u32)(v & ((1u << c) - 1));
  }
  raw += *carry;
  const u32 half = 1u << (c - 1);
  if (raw > half) { *carry = 1; return (int)raw - (int)(1u << c); }
  *carry = 0;
  return (int)raw;
}

// host tail: rows[w*c + k] = Q_k of window w (the sum of the buckets whose weight has bit k set)
//   ->  sum_w 2^(c w) sum_k 2^k Q_k, affine standard form (8 words; all zero = infinity)
inline void msm_host_tail(const MsmShape& sh, const bn254::Xyzz* rows, u64 out[8]) {
  bn254::Xyzz total = bn254::xyzz_inf();
  for (int w = (int)sh.W - 1; w >= 0; w--) {
    for (u32 i = 0; i < sh.c; i++) total = bn254::xyzz_dbl(total);
    bn254::Xyzz win = bn254::xyzz_inf();
    for (int k = (int)sh.c - 1; k >= 0; k--) {
      win = bn254::xyzz_dbl(win);
      win = bn254::xyzz_add(win, rows[(size_t)w * sh.c + k]);
    }
    total = bn254::xyzz_add(total, win);
  }
  bn254::xyzz_store_affine(total, out);
}

}  // namespace ronk
