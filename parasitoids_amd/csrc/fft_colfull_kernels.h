// Full-column passes on the register-resident three-stage transform of fft_rs.h -- the
// "column-major spectrum" pipeline for FFT sizes L = 16 * R2 * R3.
//
// The tiled column passes of fft_kernels.h split a length-L column transform into two sub-passes
// (L = L1 * L2, a [L2 x W] tile per workgroup) because a [L x 8] complex128 tile does not fit in
// LDS: every 2-D transform streams the half spectrum three times.  Here the half spectrum is
// kept COLUMN-MAJOR ("T layout": [H columns][L], a column is 16 L contiguous bytes) and one
// workgroup transforms ONE whole column with the data in registers (LDS is only the exchange
// medium between the three stages, 8.5 L bytes), so a column transform is a single pass:
//
//   k_colfull mode 0 (day step):  kernel column (row-pass output, live rows only) -> forward
//       FFT -> x state column (product stored back: CalcSol.py:66) -> inverse FFT -> the
//       spatial-row intermediate the inverse row pass reads, row-major [L][ld]
//   mode 1 (state transform):     row-pass output column -> forward FFT -> state column
//   mode 2 (get_cursol):          state column -> inverse FFT -> row-major intermediate
//   mode 3 (fftconv2):            state column *= FFT(kernel column)
//
// Per day step the spectrum-sized traffic is: state in, state out, intermediate out (3 S) instead
// of kernel-intermediate in/out + state in/out + two intermediates (6-7 S) -- the second inverse
// column sub-pass and the kernel's first forward column sub-pass do not exist.
//
// The row-major intermediate is written 16 bytes per row and column.  Eight adjacent columns
// complete a 128-byte line; their workgroups are consecutive blocks of ONE XCD (blocks go to the
// eight XCDs round-robin), so the line is assembled in that XCD's L2 before it goes to HBM
// (measured with scripts/microbench/scatter_write.hip: 3.4 TB/s against 1.4 TB/s for the naive
// block order; a tiled pass writing full lines with the same row stride reaches 2.9-3.3).
//
// Measured and not kept (N = 4097, L = 5184; the day-step pass takes 255 us for 650 MB): two
// columns per workgroup in lockstep, the way the row kernels pair rows -- 12 resident waves per
// CU instead of the 5.2 a 6-wave workgroup at 162 registers gets (its second copy does not fit
// the SIMDs' wave slots) -- 294 us: the columns then load, transform and store in the same
// phase; touching every line of the state column (one dword) behind the kernel loads so that
// HBM delivers it while the forward transform runs: 266 us.  The phases of a workgroup add up
// (PMC: VALU busy 21 %, 75 % of the wave cycles waiting); what would overlap them is a second
// INDEPENDENT workgroup per CU, i.e. <= 128 registers (a radix <= 12 four-stage plan).
#pragma once
#include "fft_rs_kernels.h"

struct ColFullArgs {
  const cplx* src;       // T layout [H][L]: row-pass output of the kernel (modes 0, 3) / of the state (mode 1)
  int64_t src_bstride;   // per blockIdx.y
  cplx* state;           // T layout [H][L]
  int64_t state_bstride;
  cplx* dst;             // row-major [L][ld] (modes 0, 2)
  int64_t dst_bstride;
  int ld, ncols, mode, store_prod;
  int nd;                // mode 0: consecutive days in this pass
  int dst_t;             // layout of dst (the inverse row pass reads it that way): 0 row-major [P][ld], 1 column-major
                         // [H][L], 2 row pairs interleaved [P/2][ld][2] -- as the four coefficients below
  int dst_jb, dst_cc, dst_rst;   // colfull_set_layout
  int64_t src_dstride, dst_dstride;   // per day
  RowLive live;          // rows of src that were never written (known zero); range advances 2 ints per day
  const unsigned long long* pred;
  // k_colfull_dual: sum of |x|^2 over the pad-only spatial rows (r >= pad_row0) of this column's
  // intermediate, per day: pad_energy[(blockIdx.y + day) * ncols + c]; nullptr = not wanted
  double* pad_energy;
  int pad_row0, ncols_total;
  int col0;              // first column of this launch (blocks map to columns col0, col0 + 1, ...)
  // k_colfull_day (mode 0): when *alt_pred is above the flag threshold -- the previous day raised the
  // boundary flag (CalcSol.py:200-201) -- the state column is not state[c] but the forward transform of
  // column c of alt_src, the row-pass output of that day's truncated field (rows as alt_live says)
  const cplx* alt_src;
  const unsigned long long* alt_pred;
  RowLive alt_live;
  FftProg prog;          // the length-L row plan (its two-level twiddle table)
};

// Where thread j of the last inverse stage puts its spatial rows j + q T3 of column c: `rst` elements apart per
// T3 rows in every layout (T3 = 16 R2 is even, so a row keeps its parity and its pair moves by T3 / 2).
//   0  row-major [P][ld]: 16-byte stores, every lane in another 128-byte line; the eight columns of a line are
//      eight workgroups of one XCD and meet in its L2
//   1  column-major [H][L] (PS_TINV): contiguous stores, but the row pass then gathers
//   2  row pairs interleaved [P/2][ld][2]: element (r, c) at ((r >> 1) ld + c) 2 + (r & 1).  Lanes 2i and 2i + 1
//      store 32 contiguous bytes -- half as many memory requests per store instruction, four columns to a line
//      (30-day two-role launch at 5184: 4.36 -> 4.00 ms; eight rows to a block instead of two: 3.92, not worth
//      what it does to the row pass) -- and the row pass, which transforms rows 2p and 2p + 1 as ONE complex row,
//      finds both in one contiguous stream (fft_rs_kernels.h: pair_src)
// One linear form for all three, its coefficients set by the host (colfull_set_layout): element offset of row j
// of column c = j dst_rst + (j & 1) dst_jb + c dst_cc, T3 more rows = T3 dst_rst more elements -- a 32-bit lane
// offset on top of uniform bases, one address register for all R3 stores, no layout test in the kernel (a
// three-way select at this point of k_colfull_dual cost 23 spilled registers).
inline void colfull_set_layout(ColFullArgs& a, int lay, int L) {
  a.dst_t = lay;
  a.dst_jb = lay == 2 ? 1 - a.ld : 0;
  a.dst_cc = lay == 1 ? L : (lay == 2 ? 2 : 1);
  a.dst_rst = lay == 1 ? 1 : a.ld;
}
#define PS_COLFULL_OFF(a, c, j) ((unsigned)((j) * (a).dst_rst + ((j) & 1) * (a).dst_jb + (c) * (a).dst_cc))

// One block per day: total pad-row energy of the day's intermediate (fixed summation order) ->
// quiet[day] = 1 when even the total passes the per-pair Parseval test of the inverse row pass
// (k_row_inv_rs: sqrt(2 P e) * scale < pad_floor), i.e. every pad-only row pair would be skipped.
static __global__ void k_pad_quiet(const double* energy, int ncols, double P, double scale, double pad_floor, int* quiet) {
  __shared__ double red[256];
  const double* e = energy + (int64_t)blockIdx.x * ncols;
  double acc = 0.0;
  for (int c = threadIdx.x; c < ncols; c += 256) acc += e[c];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  // NaN-safe: anything that is not provably quiet is read
  if (threadIdx.x == 0) quiet[blockIdx.x] = (sqrt(2.0 * P * red[0]) * scale < pad_floor) ? 1 : 0;
}

// Single day step (mode 0 with nd == 1), the state column straight from HBM: the form the
// compiler turns into 162 registers and no scratch (255 us per day at L = 5184).  The templated
// kernel below serves the other modes (114-128 registers) and the chained groups of days; its
// one-day instance costs 246-256 registers and is 15-45 % slower than this one.
// ALT: the instance that can take the state column from alt_src (see ColFullArgs); launched only
// when a re-transform is pending, so the plain instance keeps its 162 registers
template <int R1, int R2, int R3, bool CEX, bool ALT>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR)) k_colfull_day(ColFullArgs a) {
  using S = Rs<R1, R2, R3>;
  constexpr int L = S::L;
  if (pred_skip(a.pred)) return;
  // block -> column: blocks b, b+8, ..., b+56 (one XCD, dispatched back to back) own the eight
  // columns of one 128-byte line of the row-major output
  const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
  const int c = a.col0 + ((((qq >> 3) << 3) + xcd) << 3) + (qq & 7);
  if (c >= a.ncols) return;   // whole workgroup: no barrier is pending
  // CEX: complex exchange words (rs_tail_c: 3 barriers per transform, 8 for the whole day step
  // instead of 21); the buffer is then 17 L bytes, which sizes beyond 9400 do not have
  cplx* exc = reinterpret_cast<cplx*>(ps_lds_raw);
  double* ex = reinterpret_cast<double*>(ps_lds_raw);
  const int j = threadIdx.x;
  const FftProg& P = a.prog;
  const cplx w2 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j));
  const cplx w3 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j < S::T3 ? S::tw3(j) : 0);
  cplx x[S::RMAX];
  cplx* st = a.state + (int64_t)blockIdx.y * a.state_bstride + (int64_t)c * L;
  if (a.mode != 2) {
    // the re-transform of a flagged day's field, column half: done here instead of a pass of its own
    // (one spectrum less read, one less written per flagged day); the values stay in registers
    // while the kernel column is transformed
    const bool alt = ALT && a.alt_pred && !pred_skip(a.alt_pred);
    cplx sv[R3];
    if (ALT && alt) {
      const cplx* ac = a.alt_src + (int64_t)c * L;
      if (j < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) {
          const int n = j + q * S::T1;
          x[q] = make_double2(0.0, 0.0);
          if (row_live(a.alt_live, n, 0)) x[q] = ac[n];
        }
        bfly<R1, PS_FWD>(x);
      }
      if constexpr (CEX) rs_tail_c<S, R1, R2, R3, PS_FWD>(x, exc, j, w2, w3);
      else rs_tail<S, R1, R2, R3, PS_FWD>(x, ex, j, w2, w3);
      if (j < S::T3) {
#pragma unroll
        for (int q = 0; q < R3; ++q) sv[q] = x[q];
      }
      __syncthreads();   // the exchange buffer goes to the kernel column's transform
    }
    const cplx* sc = a.src + (int64_t)blockIdx.y * a.src_bstride + (int64_t)c * L;
    if (j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) {
        const int n = j + q * S::T1;
        x[q] = make_double2(0.0, 0.0);
        if (row_live(a.live, n, blockIdx.y)) x[q] = sc[n];
      }
      bfly<R1, PS_FWD>(x);
    }
    if constexpr (CEX) rs_tail_c<S, R1, R2, R3, PS_FWD>(x, exc, j, w2, w3);   // thread j < T3: X[j + q T3]
    else rs_tail<S, R1, R2, R3, PS_FWD>(x, ex, j, w2, w3);
    if (a.mode == 1) {
      if (j < S::T3) {
#pragma unroll
        for (int q = 0; q < R3; ++q) st[j + q * S::T3] = x[q];
      }
      return;
    }
    if (j < S::T3) {
      if (!alt) {
#pragma unroll
        for (int q = 0; q < R3; ++q) sv[q] = st[j + q * S::T3];
      }
#pragma unroll
      for (int q = 0; q < R3; ++q) {
        x[q] = cmul(sv[q], x[q]);
        if (a.store_prod) st[j + q * S::T3] = x[q];
      }
    }
    if (a.mode == 3) return;
    // natural order (j + q T3) -> first-stage input order (j + q T1) through the exchange buffer
    __syncthreads();
    if constexpr (CEX) {
      if (j < S::T3) {
#pragma unroll
        for (int q = 0; q < R3; ++q) exc[j + q * S::T3] = x[q];
      }
      __syncthreads();
      if (j < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q] = exc[j + q * S::T1];
      }
    } else {   // real parts, then imaginary parts
      if (j < S::T3) {
#pragma unroll
        for (int q = 0; q < R3; ++q) ex[j + q * S::T3] = x[q].x;
      }
      __syncthreads();
      if (j < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q].x = ex[j + q * S::T1];
      }
      __syncthreads();
      if (j < S::T3) {
#pragma unroll
        for (int q = 0; q < R3; ++q) ex[j + q * S::T3] = x[q].y;
      }
      __syncthreads();
      if (j < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q].y = ex[j + q * S::T1];
      }
    }
    __syncthreads();
  } else if (j < S::T1) {
#pragma unroll
    for (int q = 0; q < R1; ++q) x[q] = st[j + q * S::T1];
  }
  if (j < S::T1) bfly<R1, PS_INV>(x);
  if constexpr (CEX) rs_tail_c<S, R1, R2, R3, PS_INV>(x, exc, j, w2, w3);     // thread j < T3: spatial rows j + q T3
  else rs_tail<S, R1, R2, R3, PS_INV>(x, ex, j, w2, w3);
  if (j < S::T3) {
    const int64_t rst = a.dst_rst;
    const unsigned off = PS_COLFULL_OFF(a, c, j);
    cplx* d = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
#pragma unroll
    for (int q = 0; q < R3; ++q) (d + (int64_t)(q * S::T3) * rst)[off] = x[q];
  }
}

// A 6-wave workgroup at ~160 registers is alone on its CU anyway (see above), so the 160 KB of
// LDS are its own: with CHAIN the state column is parked in LDS next to the exchange buffer
// (every thread keeps the elements k = j + q T3 it multiplies -- private slots, no barrier)
// from the first load to the last store.  That puts the column's HBM round trip behind the first
// forward transform and lets ONE pass take `nd` consecutive un-flagged days -- A_{d+1} = A_d K_d
// chained on chip, the same products in the same order as nd single-day passes -- with the
// state read and written once instead of nd times (mode 0; kernels at src + day * src_dstride,
// outputs at dst + day * dst_dstride, live ranges at live.range + 2 * day).  Sizes whose column
// does not fit next to the exchange buffer (L > 6400) run CHAIN = false, one day per pass.
// (Holding the column in registers instead: 256 VGPRs and 1.2 KB of scratch per thread, 3x slower.)
template <int R1, int R2, int R3, bool CHAIN, int MODE>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR)) k_colfull(ColFullArgs a) {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  constexpr int L = S::L;
  if (pred_skip(a.pred)) return;
  // block -> column: blocks b, b+8, ..., b+56 (one XCD, dispatched back to back) own the eight
  // columns of one 128-byte line of the row-major output
  const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
  const int c = a.col0 + ((((qq >> 3) << 3) + xcd) << 3) + (qq & 7);
  if (c >= a.ncols) return;   // whole workgroup: no barrier is pending
  double* ex = reinterpret_cast<double*>(ps_lds_raw);
  cplx* sst = reinterpret_cast<cplx*>(ex + Y::XW + Y::RED);   // CHAIN: the state column, [L]
  const int j = threadIdx.x;
  const FftProg& P = a.prog;
  const cplx w2c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j));
  const cplx w3c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j < S::T3 ? S::tw3(j) : 0);
  cplx* st = a.state + (int64_t)blockIdx.y * a.state_bstride + (int64_t)c * L;
  if constexpr (MODE == 2) {           // inverse of the state itself
    cplx x[S::RMAX];
    if (j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) x[q] = st[j + q * S::T1];
      bfly<R1, PS_INV>(x);
    }
    rs_tail<S, R1, R2, R3, PS_INV>(x, ex, j, w2c, w3c);
    if (j < S::T3) {
      const int64_t rst = a.dst_rst;
      const unsigned off = PS_COLFULL_OFF(a, c, j);
      cplx* d = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
#pragma unroll
      for (int q = 0; q < R3; ++q) (d + (int64_t)(q * S::T3) * rst)[off] = x[q];
    }
    if (a.pad_energy) {   // as k_colfull_dual: |x|^2 over the pad-only rows of this column and day
      double pe = 0.0;
      if (j < S::T3) {
        const int qlo = a.pad_row0 / S::T3;
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          if (q < qlo) continue;
          if (q > qlo || j + q * S::T3 >= a.pad_row0) pe += x[q].x * x[q].x + x[q].y * x[q].y;
        }
      }
      pe = ps_wave_sum(pe);
      double* red = ex + Y::XW;
      if ((j & 63) == 63) red[j >> 6] = pe;
      __syncthreads();
      if (j == 0) {
        double e = 0.0;
        for (int w = 0; w < S::NTHR / 64; ++w) e += red[w];
        a.pad_energy[(int64_t)blockIdx.y * a.ncols_total + c] = e;
      }
    }
  } else if constexpr (MODE == 1) {    // forward: row-pass output column -> state column
    cplx x[S::RMAX];
    const cplx* sc = a.src + (int64_t)blockIdx.y * a.src_bstride + (int64_t)c * L;
    if (j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) {
        const int n = j + q * S::T1;
        x[q] = make_double2(0.0, 0.0);
        if (row_live(a.live, n, (int)blockIdx.y)) x[q] = sc[n];
      }
      bfly<R1, PS_FWD>(x);
    }
    rs_tail<S, R1, R2, R3, PS_FWD>(x, ex, j, w2c, w3c);
    if (j < S::T3) {
#pragma unroll
      for (int q = 0; q < R3; ++q) st[j + q * S::T3] = x[q];
    }
  } else {                             // MODE 0: day steps; MODE 3: product only
    constexpr bool chain = CHAIN;
    if (chain && j < S::T3) {   // natural order k = j + q T3: the forward transform's output order
#pragma unroll
      for (int q = 0; q < R3; ++q) sst[j + q * S::T3] = st[j + q * S::T3];
    }
    const int nd = (CHAIN && MODE == 0) ? a.nd : 1;
    // CHAIN: the kernel column of the NEXT day is fetched into xn while this day's inverse
    // transform runs (64 registers; the loop body needs 128).  Without it the fetch waits behind
    // the previous day's 66 KB of output stores in the CU's memory pipeline and nothing else can
    // run meanwhile (one workgroup per CU): 52 of 221 us per day at 5184, another 50 for the stores.
    cplx xn[R1];
    auto fetch = [&](int day, int jt) {
      const cplx* sc = a.src + (int64_t)blockIdx.y * a.src_bstride + (int64_t)day * a.src_dstride + (int64_t)c * L;
      if (jt < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) {
          const int n = jt + q * S::T1;
          xn[q] = make_double2(0.0, 0.0);
          if (row_live(a.live, n, (int)blockIdx.y + day)) xn[q] = sc[n];
        }
      }
    };
    if (CHAIN && MODE == 0) fetch(0, j);
    for (int day = 0; day < nd; ++day) {
      cplx x[S::RMAX];
      // opaque copy of the thread index: keeps the compiler from hoisting the body's ~100
      // loop-invariant addresses out of the day loop (256 registers and scratch otherwise)
      int jv = threadIdx.x;
      asm volatile("" : "+v"(jv));
      // ... and of the stage twiddles: their powers (rs_stage) are loop-invariant too, 40 complex
      // values that would stay in registers across the whole loop (246 registers at 5184, scratch
      // from 5376 up); recomputing them per transform costs ~20 complex products
      cplx w2 = w2c, w3 = w3c;
      asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
      if (!(CHAIN && MODE == 0)) fetch(day, jv);
      if (jv < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q] = xn[q];
        bfly<R1, PS_FWD>(x);
      }
      rs_tail<S, R1, R2, R3, PS_FWD>(x, ex, jv, w2, w3);   // thread jv < T3: X[jv + q T3]
      if (jv < S::T3) {
        if (chain) {
#pragma unroll
          for (int q = 0; q < R3; ++q) {
            x[q] = cmul(sst[jv + q * S::T3], x[q]);
            sst[jv + q * S::T3] = x[q];
          }
        } else {
          cplx sv[R3];
#pragma unroll
          for (int q = 0; q < R3; ++q) sv[q] = st[jv + q * S::T3];
#pragma unroll
          for (int q = 0; q < R3; ++q) {
            x[q] = cmul(sv[q], x[q]);
            if (a.store_prod) st[jv + q * S::T3] = x[q];
          }
        }
      }
      if constexpr (MODE == 0 && CHAIN) {
        // natural order (jv + q T3) -> first-stage input order (jv + q T1): the product was just
        // parked in the LDS state column, which IS the inverse transform's input -- one barrier and
        // 16-byte reads instead of a real/imaginary round trip through the exchange buffer
        __syncthreads();
        if (jv < S::T1) {
#pragma unroll
          for (int q = 0; q < R1; ++q) x[q] = sst[jv + q * S::T1];
        }
        if (day + 1 < nd) fetch(day + 1, jv);   // in flight during the inverse transform below
      }
      if constexpr (MODE == 0 && !CHAIN) {
        // natural order (jv + q T3) -> first-stage input order (jv + q T1), real parts then imaginary
        __syncthreads();
        if (jv < S::T3) {
#pragma unroll
          for (int q = 0; q < R3; ++q) ex[jv + q * S::T3] = x[q].x;
        }
        __syncthreads();
        if (jv < S::T1) {
#pragma unroll
          for (int q = 0; q < R1; ++q) x[q].x = ex[jv + q * S::T1];
        }
        __syncthreads();
        if (jv < S::T3) {
#pragma unroll
          for (int q = 0; q < R3; ++q) ex[jv + q * S::T3] = x[q].y;
        }
        __syncthreads();
        if (jv < S::T1) {
#pragma unroll
          for (int q = 0; q < R1; ++q) x[q].y = ex[jv + q * S::T1];
        }
        __syncthreads();
      }
      if constexpr (MODE == 0) {
        if (CHAIN) asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
        if (jv < S::T1) bfly<R1, PS_INV>(x);
        rs_tail<S, R1, R2, R3, PS_INV>(x, ex, jv, w2, w3);     // thread jv < T3: spatial rows jv + q T3
        if (jv < S::T3) {
          const int64_t rst = a.dst_rst;
          const unsigned off = PS_COLFULL_OFF(a, c, jv);
          cplx* d = a.dst + (int64_t)blockIdx.y * a.dst_bstride + (int64_t)day * a.dst_dstride;
#pragma unroll
          for (int q = 0; q < R3; ++q) (d + (int64_t)(q * S::T3) * rst)[off] = x[q];
        }
        if (day + 1 < nd) __syncthreads();   // the exchange buffer is reused by the next day's transform
      }
    }
    if (chain && a.store_prod && j < S::T3) {
#pragma unroll
      for (int q = 0; q < R3; ++q) st[j + q * S::T3] = sst[j + q * S::T3];
    }
  }
}


// Chained days of a FEW columns, spread over the chip: the spectra K_d of their kernel columns are
// already in `khat` ([day][column][L], written by the forward pass, mode 1); every element runs
// the chain A_{d+1} = A_d K_d (CalcSol.py:66; the same products in the same order as the chained
// passes), leaves the products in place of the spectra (the inverse pass, mode 2, reads them) and
// the last one in the state.  See conv_inv_multi: the columns that would make a thin extra round
// of the chained pass (2593 columns = 10 rounds of 256 CUs + 33).
static __global__ void k_prefix_cols(cplx* state, cplx* khat, int L, int ncols, int nd) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= L) return;
  const int64_t col = (int64_t)blockIdx.y * L + k;
  cplx p = state[col];
  for (int d = 0; d < nd; ++d) {
    cplx* e = khat + ((int64_t)d * ncols) * L + col;
    p = cmul(p, *e);
    *e = p;
  }
  state[col] = p;
}

// ------------------------------------------------------------------ chained days, two roles
// What bounds the chained pass above, measured without HBM in the loop (scripts/microbench/
// fftcore.hip, 5184): a day step costs 11.7 us on the CU = 7.3 us of butterflies + 3.5 us of LDS
// exchange + barriers, and they ADD UP -- six waves sit 2-2-1-1 on the four SIMDs (wave slots are
// handed out round-robin from SIMD 0, which is also why a second 6-wave workgroup is never
// co-resident above 128 registers), so the VALU time is that of TWO waves per stage, and everybody
// is in the same phase.  Two transforms side by side in one 12-wave workgroup (3-2-3-2 instead of
// twice 2-2-1-1, twice the waves to hide LDS latency) take 8.8 us.
//
// The two transforms of a chained day step that CAN run side by side are the inverse of day d's
// product and the forward transform of day d+1's kernel column.  So the workgroup has two roles
// of NTHR threads each, in lockstep (shared barriers, all at top level):
//   role 0 ("A"): slot s >= 1: product of day s-1 out of the LDS state column (first-stage input
//                 order) -> inverse transform -> the row-major intermediate of day s-1
//   role 1 ("B"): slot s < nd: kernel column of day s -> forward transform -> state column *= it
// nd + 1 slots for nd days (A idles in the first, B in the last: the state column is loaded /
// stored there, off the other role's path), which is why only long groups come here (a solver whose previous run raised no
// flag opens with windows of up to 16 days).  Same butterflies, twiddles and products in the same
// order as the single-role pass: results are bit-identical.
//
// LDS: the state column (16 L bytes) leaves no room for two whole exchange buffers (8.5 L each),
// so each role's exchange runs in two half passes through a half-size buffer (the writers of the
// first half are the threads below a split, the readers' first half the inputs q < ceil(R/2); 15
// barriers per transform instead of 8 -- measured +5 %).  What is left (32 KB at 5184) stages the
// NEXT day's kernel column, copied HBM -> LDS by the load unit (global_load_lds_dwordx4) while this
// slot's transforms run: role B has no registers to spare for a prefetch (the whole kernel sits at
// the 168 registers a 12-wave workgroup may use), and a fetch at the top of the slot would stall
// both roles.  Live rows beyond the staging capacity are loaded directly.
template <int R1, int R2, int R3>
struct RsDual {
  using S = Rs<R1, R2, R3>;
  static constexpr int L = S::L;
  static constexpr int QA2 = (R2 + 1) / 2, QA3 = (R3 + 1) / 2;
  static constexpr int IA = QA2 * S::T2, IB = QA3 * S::T3;          // elements of the first half pass (exchange 1, 2)
  static constexpr int JA1 = IA / 16, JA2 = 16 * QA3;               // its writer threads: j < JA
  static constexpr int OFF1 = IA + IA / 16, OFF2 = IB + IB / 16;    // padded words ahead of the second half
  static constexpr int mx(int a, int b) { return a > b ? a : b; }
  static constexpr int MAXEL = mx(mx(IA, L - IA), mx(IB, L - IB));
  static constexpr int XH = (MAXEL + MAXEL / 16 + 32 + 15) & ~15;   // doubles per role (>= 32 slack words at the end: wave sums)
  static constexpr size_t fixed_bytes = 2 * (size_t)XH * sizeof(double) + (size_t)L * sizeof(cplx);
  static constexpr long spare = (long)160 * 1024 - (long)fixed_bytes;
  static constexpr int CAP = spare > 0 ? (int)(spare / (64 * (long)sizeof(cplx))) * 64 : 0;   // staged elements: whole 1 KB chunks
  // (radix 21 and up -- and 20 x 15, 20 x 18 -- do not fit the 168 registers of a 12-wave workgroup: 8 to
  // 100 spilled registers, which cost more than the second role gains; 20 x 14 and 20 x 16 spill 1-4,
  // measured still ahead of the single-role pass: 8.32 -> 7.78 ms per 30-day stack at 5120)
  static constexpr bool ok = S::NTHR <= 384 && CAP >= 512 &&
                             (S::RMAX <= 18 || (S::RMAX == 20 && (R2 * R3 == 20 * 14 || R2 * R3 == 20 * 16)));   // (<= 6 waves per role: their sums fit the slack words)
  static constexpr size_t bytes = fixed_bytes + (size_t)CAP * sizeof(cplx);
  static_assert(S::T2 % 16 == 0 && S::T3 % 16 == 0, "half-pass offsets assume whole padding groups");
};

template <int R1, int R2, int R3>
__global__ void __launch_bounds__((2 * Rs<R1, R2, R3>::NTHR)) k_colfull_dual(ColFullArgs a) {
  using S = Rs<R1, R2, R3>;
  using D = RsDual<R1, R2, R3>;
  constexpr int L = S::L;
  constexpr int NW = S::NTHR / 64;
  if (pred_skip(a.pred)) return;
  const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
  const int c = a.col0 + ((((qq >> 3) << 3) + xcd) << 3) + (qq & 7);   // see k_colfull
  if (c >= a.ncols) return;
  const int role = threadIdx.x / S::NTHR;                      // wave-uniform
  const int j0 = threadIdx.x - role * S::NTHR;
  double* exh = reinterpret_cast<double*>(ps_lds_raw) + role * D::XH;
  cplx* sst = reinterpret_cast<cplx*>(reinterpret_cast<double*>(ps_lds_raw) + 2 * D::XH);
  cplx* stage = sst + L;
  const FftProg& P = a.prog;
  const cplx w2c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j0));
  const cplx w3c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j0 < S::T3 ? S::tw3(j0) : 0);
  cplx* st = a.state + (int64_t)blockIdx.y * a.state_bstride + (int64_t)c * L;
  const int nd = a.nd;
  const int lane = j0 & 63, wave = j0 >> 6;

  // live rows of day `day`'s kernel column: two torus intervals [a1, b1) and [a2, b2) (RowLive:
  // the two pieces of the source map, clipped to the day's source-row range)
  auto live_of = [&](int day, int& a1, int& b1, int& a2, int& b2) {
    if (!a.live.on) { a1 = 0; b1 = L; a2 = L; b2 = L; return; }
    int lo = 0, hi = 0x3fffffff;
    if (a.live.range) { lo = a.live.range[2 * ((int)blockIdx.y + day)]; hi = a.live.range[2 * ((int)blockIdx.y + day) + 1]; }
    const SrcMap& m = a.live.map;
    a1 = max(0, lo - m.off1); b1 = min(m.n1, hi - m.off1 + 1);
    a2 = m.lo2 + max(0, lo - m.off2); b2 = min(L, m.lo2 + hi - m.off2 + 1);
    if (b1 < a1) b1 = a1;
    if (b2 < a2) b2 = a2;
    if (a2 < b1) a2 = b1;      // (maps of this code never overlap; keeps the staging order well defined)
  };
  // role B: copy the live rows of day `day`'s kernel column into the staging area, row order,
  // 1 KB per instruction; every wave takes the chunks wave, wave + NW, ...
  auto stage_column = [&](int day) {
    int a1, b1, a2, b2;
    live_of(day, a1, b1, a2, b2);
    const int len1 = b1 - a1, tot = min(len1 + (b2 - a2), D::CAP);
    const cplx* sc = a.src + (int64_t)blockIdx.y * a.src_bstride + (int64_t)day * a.src_dstride + (int64_t)c * L;
    for (int ch = wave; ch * 64 < tot; ch += NW) {
      int e = ch * 64 + lane;
      e = e < tot ? e : tot - 1;                       // the last chunk: stay inside the live rows
      const int n = e < len1 ? a1 + e : a2 + (e - len1);
      __builtin_amdgcn_global_load_lds(sc + n, (ps_lds_ptr)(stage + ch * 64), 16, 0, 0);
    }
  };

  // prologue: kernel column of day 0 -> staging (role B); role A brings the state column in during
  // slot 0, where it has no transform to run (role B needs it only for the product at the slot's end)
  if (role == 1) {
    stage_column(0);
    PS_WAIT_VM0();
  }
  PS_BAR_LDS();

  for (int slot = 0; slot <= nd; ++slot) {
    const bool act = role == 0 ? slot >= 1 : slot < nd;        // wave-uniform
    cplx x[S::RMAX];
    int j = j0;
    cplx w2 = w2c, w3 = w3c;
    // opaque per-slot copies (see k_colfull): no loop-invariant twiddle powers or addresses in registers
    asm volatile("" : "+v"(j), "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
    // ---- first-stage inputs.  Role A's inverse transform runs through the FORWARD code on
    // conjugated data (inverse(x) == conj(forward(conj(x))) bit for bit, see rs_stage): one
    // transform body for both roles, one set of data registers.
    if (role == 0 && slot == 0) {        // state column -> LDS, coalesced
#pragma unroll 7
      for (int k = j; k < L; k += S::NTHR) sst[k] = st[k];
    }
    if (act && j < S::T1) {
      if (role == 0) {
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q] = cconj(sst[j + q * S::T1]);
      } else {
        int a1, b1, a2, b2;
        live_of(slot, a1, b1, a2, b2);
        const int len1 = b1 - a1, len2 = b2 - a2;
        if (len1 + len2 <= D::CAP) {     // uniform; the usual case: every live row was staged
#pragma unroll
          for (int q = 0; q < R1; ++q) {
            const int n = j + q * S::T1;
            const unsigned d1 = (unsigned)(n - a1), d2 = (unsigned)(n - a2);
            const bool in1 = d1 < (unsigned)len1, in2 = d2 < (unsigned)len2;
            const cplx v = stage[in1 ? d1 : (in2 ? len1 + d2 : 0u)];
            x[q] = (in1 || in2) ? v : make_double2(0.0, 0.0);
          }
        } else {
          const cplx* sc = a.src + (int64_t)blockIdx.y * a.src_bstride + (int64_t)slot * a.src_dstride + (int64_t)c * L;
#pragma unroll
          for (int q = 0; q < R1; ++q) {
            const int n = j + q * S::T1;
            x[q] = make_double2(0.0, 0.0);
            int e = -1;
            if (n >= a1 && n < b1) e = n - a1;
            else if (n >= a2 && n < b2) e = len1 + (n - a2);
            if (e >= 0) x[q] = e < D::CAP ? stage[e] : sc[n];
          }
        }
      }
      bfly<R1, PS_FWD>(x);
    }
    // ---- exchange 1 in two half passes per part.  A thread of the second half pass still holds
    // un-written outputs when the first half's inputs arrive: those wait in `keep` until its put.
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      double keep[D::QA2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (act && j < S::T1 && (j < D::JA1) == (h == 0)) {
          if (part) rs_put<R1, 1>(exh, S::x1_w(j) - h * D::OFF1, 1, x);
          else rs_put<R1, 0>(exh, S::x1_w(j) - h * D::OFF1, 1, x);
        }
        PS_BAR_LDS();
        if (part == 0 && h == 0 && role == 1 && slot + 1 < nd) stage_column(slot + 1);   // the staging area was read above
        if (act && j < S::T2) {
          if (h == 0) {
#pragma unroll
            for (int q = 0; q < D::QA2; ++q) keep[q] = exh[S::x_r(j) + q * S::X1_RS];
          } else {
#pragma unroll
            for (int q = D::QA2; q < R2; ++q) {
              const double v = exh[S::x_r(j) + q * S::X1_RS - D::OFF1];
              if (part) x[q].y = v;
              else x[q].x = v;
            }
#pragma unroll
            for (int q = 0; q < D::QA2; ++q) {
              if (part) x[q].y = keep[q];
              else x[q].x = keep[q];
            }
          }
        }
        if (!(part == 1 && h == 1)) PS_BAR_LDS();
      }
    }
    // (twiddle powers are formed here, not hoisted above the exchange where they would cost 36 registers)
    asm volatile("" : "+v"(w2.x), "+v"(w2.y));
    if (act && j < S::T2) rs_stage<R2, PS_FWD>(x, w2, true);
    PS_BAR_LDS();
    // ---- exchange 2
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      double keep[D::QA3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (act && j < S::T2 && (j < D::JA2) == (h == 0)) {
          if (part) rs_put<R2, 1>(exh, S::x2_w(j) - h * D::OFF2, 17, x);
          else rs_put<R2, 0>(exh, S::x2_w(j) - h * D::OFF2, 17, x);
        }
        PS_BAR_LDS();
        if (act && j < S::T3) {
          if (h == 0) {
#pragma unroll
            for (int q = 0; q < D::QA3; ++q) keep[q] = exh[S::x_r(j) + q * S::X2_RS];
          } else {
#pragma unroll
            for (int q = D::QA3; q < R3; ++q) {
              const double v = exh[S::x_r(j) + q * S::X2_RS - D::OFF2];
              if (part) x[q].y = v;
              else x[q].x = v;
            }
#pragma unroll
            for (int q = 0; q < D::QA3; ++q) {
              if (part) x[q].y = keep[q];
              else x[q].x = keep[q];
            }
          }
        }
        if (!(part == 1 && h == 1)) PS_BAR_LDS();
      }
    }
    asm volatile("" : "+v"(w3.x), "+v"(w3.y));
    if (act && j < S::T3) rs_stage<R3, PS_FWD>(x, w3, true);
    if (act && j < S::T3) {
      if (role == 0) {                                     // conj: spatial rows j + q T3 of day slot - 1
        const int64_t rst = a.dst_rst;
        cplx* d = a.dst + (int64_t)blockIdx.y * a.dst_bstride + (int64_t)(slot - 1) * a.dst_dstride + PS_COLFULL_OFF(a, c, j);
#pragma unroll
        for (int q = 0; q < R3; ++q) d[(int64_t)(q * S::T3) * rst] = cconj(x[q]);
      } else {                                             // X[j + q T3] of day slot's kernel column
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          const cplx p = cmul(sst[j + q * S::T3], x[q]);   // CalcSol.py:66
          sst[j + q * S::T3] = p;
        }
      }
    }
    // the idle role of the last slot writes the final state back (it is complete since the end of
    // slot nd - 1) while role A runs the last inverse transform
    if (role == 1 && slot == nd && a.store_prod) {
      for (int k = j; k < L; k += S::NTHR) st[k] = sst[k];
    }
    if (role == 0 && a.pad_energy) {                       // wave sums -> the slack words behind role A's exchange buffer
      double pe = 0.0;                                     // (after the stores: x is all that is live here)
      if (act && j < S::T3) {
        const int qlo = a.pad_row0 / S::T3;                // uniform: only the last few q hold pad-only rows
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          if (q < qlo) continue;
          if (q > qlo || j + q * S::T3 >= a.pad_row0) pe += x[q].x * x[q].x + x[q].y * x[q].y;
        }
      }
      pe = ps_wave_sum(pe);
      if (lane == 63) exh[D::XH - 8 + wave] = pe;
    }
    if (role == 1) PS_WAIT_VM0();                          // this wave's share of the next kernel column has landed
    PS_BAR_LDS();                                          // product and staging visible; exchange buffers free
    if (role == 0 && a.pad_energy && slot >= 1 && j0 == 0) {
      double e = 0.0;
      for (int w = 0; w < NW; ++w) e += exh[D::XH - 8 + w];
      a.pad_energy[((int64_t)blockIdx.y + (slot - 1)) * a.ncols_total + c] = e;
    }
  }
}
