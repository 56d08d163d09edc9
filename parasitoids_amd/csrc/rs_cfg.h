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
  // OFF: with it the single-day pass is 3.5 % (5600) to 13 % (8400) faster, but the compiler then
  // contracts the transform's multiply-adds differently from the chained pass (which has no LDS for
  // complex words): the two differ in the last bit, and a chain that rolls back to single-day
  // passes after a flag would no longer be bit-identical to the non-speculative chain
  // (test_flag_speculation_is_exact_in_the_full_column_pipeline).
  static constexpr bool CEX = false && 2 * LDSC1 <= (size_t)160 * 1024;
  static constexpr size_t LDSD = CEX ? 2 * LDSC1 : LDSC1;
};
