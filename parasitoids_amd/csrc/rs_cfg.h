// Compile-time configuration of the register-resident kernels per size (shared by the two
// translation units that instantiate them).
#pragma once
#include "fft_rs_kernels.h"
#include "fft_rs_sizes.h"

template <int R2, int R3>
struct RsCfg {
  using S = Rs<16, R2, R3>;
  static constexpr int W = S::NTHR / 64;
  // row pairs per workgroup: up to 12 waves (3 per SIMD at the ~165 registers of a radix-18 stage)
  static constexpr int NP = 12 / W < 1 ? 1 : (12 / W > 4 ? 4 : 12 / W);
  static constexpr size_t LDS = RsInvLds<16, R2, R3>::bytes(NP);
  // full-column pass: one column per workgroup; the state column is parked in LDS next to the
  // exchange buffer when both fit (k_colfull CHAIN)
  static constexpr size_t LDSC1 = RsInvLds<16, R2, R3>::bytes(1);
  static constexpr bool CHAIN = LDSC1 + (size_t)S::L * sizeof(cplx) <= (size_t)160 * 1024;
  static constexpr size_t LDSC = CHAIN ? LDSC1 + (size_t)S::L * sizeof(cplx) : LDSC1;
  // single-day pass: complex exchange words (rs_tail_c) when the doubled buffer fits
  // (The library is built with -ffp-contract=on: multiply-adds are contracted where the SOURCE
  // writes them in one expression, not where the optimiser finds them.  With the default `fast`
  // this pass and the chained pass -- real exchange words, no LDS for complex ones -- came out one
  // bit apart, and a chain rolling back to single-day passes after a flag must stay bit-identical
  // to the non-speculative chain: test_flag_speculation_is_exact_in_the_full_column_pipeline.)
  static constexpr bool CEX = 2 * LDSC1 <= (size_t)160 * 1024;
  static constexpr size_t LDSD = CEX ? 2 * LDSC1 : LDSC1;
  // single-day pass that transforms a flagged day's truncated column itself (k_colfull_day ALT): R3
  // more complex values stay in registers across the kernel column's transform -- the sizes where
  // that still compiles without scratch
  static constexpr bool ALT = S::NTHR <= 512 && R2 * R3 <= 450 && !(R2 == 21 && R3 == 21);
};
