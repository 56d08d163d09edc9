// Register-resident row transforms for three-stage sizes L = 16 * R2 * R3 (5184 = 16*18*18).
//
// The LDS-resident row passes of fft_kernels.h need the whole row pair in LDS (83 KB at
// L = 5184), i.e. one workgroup per CU: load, transform and store of a row pair cannot
// overlap with anything.  Here the data lives in REGISTERS (one radix-R butterfly per
// thread, Stockham auto-sort indexing) and LDS is only the exchange medium between stages,
// real parts first, then imaginary parts, so a workgroup needs L*8 B (+1/16 padding) =
// 44 KB instead of 83 KB.  The first stage takes its inputs straight from HBM (thread j reads
// elements j + q*L/R1: coalesced), the last stage feeds the epilogue from registers (thread
// j holds outputs j + q*L/R3: coalesced stores), so a row pair makes 4 LDS passes instead of 8.
//
// Stockham stage (radix r, Ns = product of the earlier radices, T = L/r), thread j < T:
//   k = j mod Ns;  x_q = in[j + q T] * w_{Ns r}^{k q};  y = DFT_r(x);
//   out[(j - k) r + k + q' Ns] = y_q'            -> natural order after the last stage.
// LDS word index i lives at i + (i >> 4): with R1 = 16 every exchange access below is a
// compile-time offset from one per-thread base and (nearly) bank-conflict free.
//
// Plain C++ index helpers are shared with tests/host/fft_emul.cpp.
#pragma once
#include "fft_core.h"

template <int R1, int R2, int R3>
struct Rs {
  static_assert(R1 == 16, "the exchange padding assumes a radix-16 first stage");
  static constexpr int L = R1 * R2 * R3;
  static constexpr int T1 = L / R1, T2 = L / R2, T3 = L / R3;
  static constexpr int TMAX = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
  static constexpr int NTHR = (TMAX + 63) / 64 * 64;
  static constexpr int RMAX = R1 > R2 ? (R1 > R3 ? R1 : R3) : (R2 > R3 ? R2 : R3);
  static constexpr int XWORDS = L + L / 16;          // padded exchange buffer (doubles)
  static constexpr int X1_RS = T2 + T2 / 16;         // read stride of exchange 1 (T2 % 16 == 0)
  static constexpr int X2_RS = T3 + T3 / 16;         // read stride of exchange 2 (T3 % 16 == 0)
  // exchange 1: stage-1 thread j writes y_q' at x1_w(j) + q'; stage-2 thread j reads x_q at
  // x_r(j) + q * X1_RS
  static PS_HD int x1_w(int j) { return 17 * j; }
  static PS_HD int x_r(int j) { return j + (j >> 4); }
  // exchange 2: stage-2 thread j (k = j & 15) writes y_q' at x2_w(j) + 17 q'; stage-3 thread j
  // reads x_q at x_r(j) + q * X2_RS
  static PS_HD int x2_w(int j) {
    const int k = j & 15;
    const int b = (j - k) * R2 + k;
    return b + (b >> 4);
  }
  // twiddle exponents (of w_L) of thread j in stages 2 and 3
  static PS_HD int tw2(int j) { return (j & 15) * R3; }
  static PS_HD int tw3(int j) { return j; }
};

template <int R, int PART>
PS_HD void rs_put(double* ex, int base, int stride, const cplx* x) {
#pragma unroll
  for (int q = 0; q < R; ++q) ex[base + q * stride] = PART ? x[q].y : x[q].x;
}
template <int R, int PART>
PS_HD void rs_get(const double* ex, int base, int stride, cplx* x) {
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (PART) x[q].y = ex[base + q * stride];
    else x[q].x = ex[base + q * stride];
  }
}

// input twiddles w1^q (w1 = forward table value, conjugated for the inverse) + butterfly.
// w^q = w^(4a) * w^b (q = 4a + b) from w, w^2, w^3 and w^4, w^8, w^12, w^16: at most four
// chained products per twiddle and ~30 live registers instead of a full w[R] array.
template <int R, int DIR>
PS_HD void rs_stage(cplx* x, cplx w1, bool tw) {
  static_assert(R <= 28, "twiddle generation covers q < 28");
  if (tw) {
    cplx lo[4], hi[7];
    lo[1] = w1;
    lo[2] = cmul(w1, w1);
    lo[3] = cmul(lo[2], w1);
    hi[1] = cmul(lo[2], lo[2]);
    hi[2] = cmul(hi[1], hi[1]);
    hi[3] = cmul(hi[2], hi[1]);
    hi[4] = cmul(hi[2], hi[2]);
    if (R > 20) {
      hi[5] = cmul(hi[4], hi[1]);
      hi[6] = cmul(hi[3], hi[3]);
    }
#pragma unroll
    for (int q = 1; q < R; ++q) {
      const int a = q >> 2, b = q & 3;
      const cplx w = a == 0 ? lo[b] : (b == 0 ? hi[a] : cmul(hi[a], lo[b]));
      // inverse: x * conj(w) written as cmul with the conjugated twiddle (not cmulc): every
      // multiply-add then rounds like the forward transform's on conjugated data, so
      // inverse(x) == conj(forward(conj(x))) BIT FOR BIT -- the two-role chained pass
      // (k_colfull_dual) runs its inverse transforms through the forward code that way
      x[q] = DIR == PS_FWD ? cmul(x[q], w) : cmul(x[q], cconj(w));
    }
  }
  bfly<R, DIR>(x);
}

#if defined(__HIPCC__)
// Workgroup barrier for the LDS exchanges WITHOUT the memory-model fence of __syncthreads(): the
// fence makes the compiler wait for every outstanding vector-memory operation, which would drain
// the asynchronous HBM -> LDS prefetch of the persistent kernels (k_row_inv_rsp) at the first
// exchange.  LDS traffic of this wave is complete (lgkmcnt 0) before the barrier, which is all the
// exchanges need.
#define PS_BAR_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PS_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
template <bool RAW>
__device__ __forceinline__ void rs_bar() {
  if (RAW) PS_BAR_LDS();
  else __syncthreads();
}

// stages 2 and 3 with both exchanges; on entry x holds the stage-1 OUTPUT of thread j,
// on exit the stage-3 output (natural index j + q' T3) of thread j < T3.
struct RsNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `hook` runs in the second stage's butterfly interval, the longest stretch without a barrier:
// threads beyond T2 have nothing to do there (the persistent row kernel's last wave finishes the
// previous pair's statistics in it)
template <class S, int R1, int R2, int R3, int DIR, bool RAW = false, class Hook = RsNoHook>
__device__ __forceinline__ void rs_tail(cplx* x, double* ex, const int j, const cplx w2, const cplx w3,
                                        const Hook& hook = Hook()) {
  if (j < S::T1) rs_put<R1, 0>(ex, S::x1_w(j), 1, x);
  rs_bar<RAW>();
  if (j < S::T2) rs_get<R2, 0>(ex, S::x_r(j), S::X1_RS, x);
  rs_bar<RAW>();
  if (j < S::T1) rs_put<R1, 1>(ex, S::x1_w(j), 1, x);
  rs_bar<RAW>();
  if (j < S::T2) rs_get<R2, 1>(ex, S::x_r(j), S::X1_RS, x);
  hook();
  if (j < S::T2) rs_stage<R2, DIR>(x, w2, true);
  rs_bar<RAW>();
  if (j < S::T2) rs_put<R2, 0>(ex, S::x2_w(j), 17, x);
  rs_bar<RAW>();
  if (j < S::T3) rs_get<R3, 0>(ex, S::x_r(j), S::X2_RS, x);
  rs_bar<RAW>();
  if (j < S::T2) rs_put<R2, 1>(ex, S::x2_w(j), 17, x);
  rs_bar<RAW>();
  if (j < S::T3) rs_get<R3, 1>(ex, S::x_r(j), S::X2_RS, x);
  if (j < S::T3) rs_stage<R3, DIR>(x, w3, true);
}


// The same two exchanges with COMPLEX words (16 bytes: buffer 2 x XWORDS doubles): real and
// imaginary parts travel together, so an exchange is put -> barrier -> get and the transform has 3
// barriers instead of 8 and half the LDS instructions.  Same index maps: with 16-byte words the
// i + (i >> 4) padding spreads 16 consecutive words over all 64 banks.  For the kernels that own
// their CU's LDS anyway (the buffer is 17 L bytes: 88 KB at 5184).
template <int R>
__device__ __forceinline__ void rs_putc(cplx* ex, int base, int stride, const cplx* x) {
#pragma unroll
  for (int q = 0; q < R; ++q) ex[base + q * stride] = x[q];
}
template <int R>
__device__ __forceinline__ void rs_getc(const cplx* ex, int base, int stride, cplx* x) {
#pragma unroll
  for (int q = 0; q < R; ++q) x[q] = ex[base + q * stride];
}
template <class S, int R1, int R2, int R3, int DIR, bool RAW = false>
__device__ __forceinline__ void rs_tail_c(cplx* x, cplx* ex, const int j, const cplx w2, const cplx w3) {
  if (j < S::T1) rs_putc<R1>(ex, S::x1_w(j), 1, x);
  rs_bar<RAW>();
  if (j < S::T2) rs_getc<R2>(ex, S::x_r(j), S::X1_RS, x);
  if (j < S::T2) rs_stage<R2, DIR>(x, w2, true);
  rs_bar<RAW>();
  if (j < S::T2) rs_putc<R2>(ex, S::x2_w(j), 17, x);
  rs_bar<RAW>();
  if (j < S::T3) rs_getc<R3>(ex, S::x_r(j), S::X2_RS, x);
  if (j < S::T3) rs_stage<R3, DIR>(x, w3, true);
}

#endif
