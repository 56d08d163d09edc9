// Row-pass kernels on the register-resident three-stage transform of fft_rs.h.
// Same arguments, same results (to round-off) and the same fused epilogue as k_row_inv /
// k_row_fwd of fft_kernels.h; one row pair per workgroup.
#pragma once
#include "fft_kernels.h"
#include "fft_rs.h"
#include <type_traits>


// ------------------------------------------------------------ inverse rows
// Half-spectrum rows (ra, rb) -> Z = A + i B (Hermitian-extended while loading) -> inverse
// FFT -> a = Re z, b = Im z, with the epilogue of CalcSol.ifft2 / r_small_vals
// (CalcSol.py:35-41, :126-135) applied to the last stage's registers.
// NP row pairs per workgroup, each on its own NTHR threads and its own exchange buffer, in
// lockstep (shared barriers): twice the loads in flight and twice the waves to hide VALU/LDS
// latency per CU without needing a second resident workgroup (158 registers allow 3 waves per
// SIMD = 12 per CU, and two 6-wave workgroups are not co-scheduled).
template <int R1, int R2, int R3>
struct RsInvLds {
  using S = Rs<R1, R2, R3>;
  static constexpr int XW = (S::XWORDS + 15) & ~15;
  static constexpr int RED = 16 * (S::NTHR / 64);   // two sets of 5 wave partials (k_row_inv_rsp defers the finalisation) + energy
  static constexpr size_t bytes(int np) { return (size_t)np * (XW + RED) * sizeof(double); }
};

template <int R1, int R2, int R3, int NP>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR * NP)) k_row_inv_rs(RowInvArgs a) {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  constexpr int L = S::L;
  constexpr int NW = S::NTHR / 64;
  const int half = threadIdx.x / S::NTHR;
  // Reverse dispatch order: the cheap pad-only pairs (top rows, mostly skipped) and the odd
  // last domain row go first, so the equally long domain workgroups that follow end together
  // instead of leaving one straggler for an extra round.
  int unit = (int)blockIdx.x;
  if (a.tstride) {
    // column-major input: a 128-byte line holds 8 rows = 4 / NP workgroups' worth; they are
    // consecutive blocks of ONE XCD, so the line comes from HBM once and the other readers
    // find it in that L2 (gridDim.x is a multiple of 8 * (4 / NP))
    constexpr int KL = NP >= 4 ? 1 : 4 / NP;
    const int xcd = unit & 7, q = unit >> 3;
    unit = ((q / KL) * 8 + xcd) * KL + (q % KL);
  }
  const int pair = ((int)gridDim.x - 1 - unit) * NP + half;
  if (2 * pair >= a.P || pair < 0) return;   // ended waves do not take part in the barriers below
  double* ex = reinterpret_cast<double*>(ps_lds_raw) + half * (Y::XW + Y::RED);
  double* red = ex + Y::XW;      // 5 * NW doubles
  const int j = threadIdx.x - half * S::NTHR, lane = j & 63, wave = j >> 6;
  const cplx* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  const int ra = 2 * pair, rb = ra + 1;
  const bool hasb = rb < a.P;
  // element k of row r: row-major at r * ld + k, column-major at k * tstride + r, row pairs interleaved
  // (pair_src; P even) at ((r >> 1) ld + k) 2 + (r & 1) -- row 2p starts where it does in the row-major layout
  const int64_t kst = a.tstride ? (int64_t)a.tstride : (a.pair_src ? 2 : 1);
  const cplx* pa = src + (a.tstride ? (int64_t)ra : (int64_t)ra * a.ld);
  const cplx* pb = a.pair_src ? pa + 1
                              : src + (a.tstride ? (int64_t)(hasb ? rb : ra) : (int64_t)(hasb ? rb : ra) * a.ld);
  const bool pad_only = ra >= a.N;
  const FftProg& P = a.prog;
  const cplx w2 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j));
  const cplx w3 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j < S::T3 ? S::tw3(j) : 0);
  cplx x[S::RMAX];
  double energy = 0.0;
  if (j < S::T1) {
    // A straight into x; B in two halves, so at most 3/4 of the 2 x 16 loads hold registers
    auto idx = [&](int q) -> unsigned {
      const bool direct = (q < R1 / 2) || (q == R1 / 2 && j == 0);   // j + q T1 <= L/2
      const unsigned i = (unsigned)j + (unsigned)(q * S::T1);
      return direct ? i : (unsigned)L - i;
    };
#pragma unroll
    for (int q = 0; q < R1; ++q) x[q] = pa[idx(q) * kst];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cplx B[R1 / 2];
#pragma unroll
      for (int u = 0; u < R1 / 2; ++u) B[u] = pb[idx(h * (R1 / 2) + u) * kst];   // pb aliases row a when there is no row b
#pragma unroll
      for (int u = 0; u < R1 / 2; ++u) {
        const int q = h * (R1 / 2) + u;
        const bool direct = (q < R1 / 2) || (q == R1 / 2 && j == 0);
        const cplx A = x[q];
        if (!hasb) B[u] = make_double2(0.0, 0.0);
        if (direct) energy += A.x * A.x + A.y * A.y + B[u].x * B[u].x + B[u].y * B[u].y;
        x[q] = direct ? make_double2(A.x - B[u].y, A.y + B[u].x)
                      : make_double2(A.x + B[u].y, B[u].x - A.y);
      }
    }
  }
  // Pad-only row pairs feed nothing but the boundary flag; Parseval bounds their largest
  // value (see k_row_inv) and the transform is skipped when that cannot raise the flag.
  // The barrier is unconditional: the NP row pairs of a workgroup share it, and a pair that
  // returns here simply stops counting for the later ones.
  if (pad_only) {   // uniform per row pair
    energy = ps_wave_sum(energy);
    if (lane == 63) red[wave] = energy;
  }
  __syncthreads();
  if (pad_only) {
    double e = 0.0;
    for (int w = 0; w < NW; ++w) e += red[w];
    if (sqrt(2.0 * (double)a.P * e) * a.scale < a.pad_floor) return;
  }
  if (j < S::T1) bfly<R1, PS_INV>(x);
  rs_tail<S, R1, R2, R3, PS_INV>(x, ex, j, w2, w3);

  double* rec = a.rec + (int64_t)blockIdx.y * a.rec_bstride;
  double sa = 0.0, sb = 0.0, pmax = 0.0;
  int ca = 0, cb = 0;
  if (j < S::T3) {
    // uniform row bases + 32-bit lane offsets: one address register for all 2 x R3 stores
    double* reca = rec + (int64_t)ra * a.N;
    double* recb = rec + (int64_t)rb * a.N;
    const bool dom_a = ra < a.N, dom_b = rb < a.N;
    const unsigned uj = (unsigned)j, uN = (unsigned)a.N;
    const bool dom_ab = dom_a && dom_b;
#pragma unroll
    for (int q = 0; q < R3; ++q) {
      const unsigned i = uj + (unsigned)(q * S::T3);
      const double va = x[q].x * a.scale, vb = x[q].y * a.scale;
      const double ta = va * a.stat_scale, tb = vb * a.stat_scale;
      if (dom_ab && (unsigned)((q + 1) * S::T3) <= uN) {
        // uniform fast path (almost every element): both rows and the whole chunk of
        // columns lie inside the domain -- no pad bookkeeping, unpredicated stores
        reca[i] = va;
        recb[i] = vb;
        const bool ka = !(ta < a.negval), kb = !(tb < a.negval);
        sa += ka ? ta : 0.0;
        sb += kb ? tb : 0.0;
        ca += __popcll(__ballot(ka));     // per wave, on the scalar unit
        cb += __popcll(__ballot(kb));
      } else {
        const bool in = i < uN;
        const bool ina = in && dom_a, inb = in && dom_b;
        // branch-free statistics; only the two stores are predicated
        const bool ka = ina && !(ta < a.negval), kb = inb && !(tb < a.negval);
        sa += ka ? ta : 0.0;
        sb += kb ? tb : 0.0;
        ca += __popcll(__ballot(ka));
        cb += __popcll(__ballot(kb));
        pmax = fmax(pmax, ina ? 0.0 : va);
        pmax = fmax(pmax, (inb || !hasb) ? 0.0 : vb);
        if (ina) reca[i] = va;
        if (inb) recb[i] = vb;
      }
    }
  }
  // deterministic block reduction (fixed DPP tree, then waves in order)
  sa = ps_wave_sum(sa);
  sb = ps_wave_sum(sb);
  ca = __builtin_amdgcn_readfirstlane(ca);   // the same in every lane that took part (lane 0 did if any did)
  cb = __builtin_amdgcn_readfirstlane(cb);
  pmax = ps_wave_max0(pmax);
  double* redm = red + 4 * NW;
  if (lane == 63) {
    red[wave * 4 + 0] = sa;
    red[wave * 4 + 1] = sb;
    red[wave * 4 + 2] = (double)ca;
    red[wave * 4 + 3] = (double)cb;
    redm[wave] = pmax;
  }
  __syncthreads();
  if (j == 0) {
    double ta = 0, tb = 0, na = 0, nbb = 0, m = 0;
    for (int w = 0; w < NW; ++w) {
      ta += red[w * 4 + 0];
      tb += red[w * 4 + 1];
      na += red[w * 4 + 2];
      nbb += red[w * 4 + 3];
      m = fmax(m, redm[w]);
    }
    double* rowsum = a.rowsum + (int64_t)blockIdx.y * a.stat_bstride;
    long long* rowcnt = a.rowcnt + (int64_t)blockIdx.y * a.stat_bstride;
    if (ra < a.N) { rowsum[ra] = ta; rowcnt[ra] = (long long)na; }
    if (rb < a.N) { rowsum[rb] = tb; rowcnt[rb] = (long long)nbb; }
    unsigned long long* pm = a.padmax + blockIdx.y;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    // only maxima that can matter for the flag (> 1e-8) are published: the read-check costs a
    // global round trip that would otherwise end every workgroup
    if (m > a.pad_floor && bits > __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(pm, bits);
  }
}


// ------------------------------------------------------------ inverse rows + fold (PS_MODE_FOLD)
// The inverse row pass of the full linear-convolution field (M x M, y in [-m, P + m) with negative y
// at M + y) and its fold back onto the reference's P-torus (k_fold: c[x] = lin[x] + lin[x + P] for
// x < m, + lin[x - P + M] for x >= P - m, per dimension) in one kernel: the field itself never goes to
// HBM.  What makes that possible is the pairing of the rows: the two real rows that travel through one
// complex transform are a torus row's own two sources (r, r + P) or (r, r - P + M) wherever it has two
// -- after the transform their sum is Re + Im of the same register -- and two neighbours (r, r + 1)
// in the middle of the torus where a row has one source.  The fold along the row needs values of other
// threads: one trip through the exchange buffer per output row.  Needs P >= 2 m (a row folds at most
// once).  Writes what k_fold writes: the P x P torus (next day's state), the raw N x N record, the row
// statistics of r_small_vals (CalcSol.py:126-135) and the maximum over the pad region (CalcSol.py:36-37).
// Sums are formed rows first, then columns (k_fold: one element after the other): equal to round-off.
struct RowFoldArgs {
  const cplx* src;   // column-pass output, row-major [M][ld]
  int ld, M, P, N, m;
  double scale, negval, stat_scale;
  double* torus;     // [P][P]
  double* rec;       // [N][N]
  double* rowsum;
  long long* rowcnt;
  unsigned long long* padmax;
  FftProg prog;
};

template <int R1, int R2, int R3>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR)) k_row_inv_fold(RowFoldArgs a) {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  constexpr int L = S::L;
  constexpr int NW = S::NTHR / 64;
  double* ex = reinterpret_cast<double*>(ps_lds_raw);
  double* red = ex + Y::XW;
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  // unit -> rows (uniform): [0, m) rows with a partner above, [m, 2m) rows with a partner below (stored at
  // the top of the M-torus), then the middle rows two by two
  const int u = (int)blockIdx.x;
  int ra, rb;
  bool summed, hasb = true;
  if (u < a.m) { ra = u; rb = u + a.P; summed = true; }
  else if (u < 2 * a.m) { ra = a.P - a.m + (u - a.m); rb = ra - a.P + a.M; summed = true; }
  else { ra = a.m + 2 * (u - 2 * a.m); rb = ra + 1; summed = false; hasb = rb < a.P - a.m; }
  const cplx* pa = a.src + (int64_t)ra * a.ld;
  const cplx* pb = a.src + (int64_t)(hasb ? rb : ra) * a.ld;
  const FftProg& P = a.prog;
  const cplx w2 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j));
  const cplx w3 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j < S::T3 ? S::tw3(j) : 0);
  cplx x[S::RMAX];
  if (j < S::T1) {   // Z = A + i B, Hermitian-extended while loading (as k_row_inv_rs)
    auto idx = [&](int q) -> unsigned {
      const bool direct = (q < R1 / 2) || (q == R1 / 2 && j == 0);
      const unsigned i = (unsigned)j + (unsigned)(q * S::T1);
      return direct ? i : (unsigned)L - i;
    };
#pragma unroll
    for (int q = 0; q < R1; ++q) x[q] = pa[idx(q)];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cplx B[R1 / 2];
#pragma unroll
      for (int v = 0; v < R1 / 2; ++v) B[v] = pb[idx(h * (R1 / 2) + v)];
#pragma unroll
      for (int v = 0; v < R1 / 2; ++v) {
        const int q = h * (R1 / 2) + v;
        const bool direct = (q < R1 / 2) || (q == R1 / 2 && j == 0);
        const cplx A = x[q];
        if (!hasb) B[v] = make_double2(0.0, 0.0);
        x[q] = direct ? make_double2(A.x - B[v].y, A.y + B[v].x)
                      : make_double2(A.x + B[v].y, B[v].x - A.y);
      }
    }
    bfly<R1, PS_INV>(x);
  }
  rs_tail<S, R1, R2, R3, PS_INV>(x, ex, j, w2, w3);   // thread j < T3: columns j + q T3 of rows ra (Re), rb (Im)

  double pmax = 0.0;
  // one output row: its values along the row through the exchange buffer, folded, written, counted
  auto emit = [&](const double (&w)[R3], int r, double& sum, int& cnt) {
    __syncthreads();   // the buffer is free (last transform stage / previous row read)
    if (j < S::T3) {
#pragma unroll
      for (int q = 0; q < R3; ++q) ex[j + q * S::T3] = w[q];
    }
    __syncthreads();
    if (j < S::T3) {
      double* trow = a.torus + (int64_t)r * a.P;
      double* rrow = a.rec + (int64_t)r * a.N;
      const bool dom_r = r < a.N;
#pragma unroll
      for (int q = 0; q < R3; ++q) {
        const int c = j + q * S::T3;
        if (c < a.P) {
          double t = w[q];
          if (c < a.m) t += ex[c + a.P];
          if (c >= a.P - a.m) t += ex[c - a.P + a.M];
          trow[c] = t;
          if (dom_r && c < a.N) {
            rrow[c] = t;
            const double tt = t * a.stat_scale;
            if (!(tt < a.negval)) { sum += tt; ++cnt; }
          } else {
            pmax = fmax(pmax, t);
          }
        }
      }
    }
  };
  double sa = 0.0, sb = 0.0;
  int ca = 0, cb = 0;
  {
    double w[R3];
    if (summed) {
#pragma unroll
      for (int q = 0; q < R3; ++q) w[q] = x[q].x * a.scale + x[q].y * a.scale;
      emit(w, ra, sa, ca);
    } else {
#pragma unroll
      for (int q = 0; q < R3; ++q) w[q] = x[q].x * a.scale;
      emit(w, ra, sa, ca);
      if (hasb) {   // uniform
#pragma unroll
        for (int q = 0; q < R3; ++q) w[q] = x[q].y * a.scale;
        emit(w, rb, sb, cb);
      }
    }
  }
  // deterministic block reduction (fixed DPP tree, then waves in order)
  sa = ps_wave_sum(sa);
  sb = ps_wave_sum(sb);
  ca = ps_wave_sum_i32(ca);
  cb = ps_wave_sum_i32(cb);
  pmax = ps_wave_max0(pmax);
  double* redm = red + 4 * NW;
  if (lane == 63) {
    red[wave * 4 + 0] = sa;
    red[wave * 4 + 1] = sb;
    red[wave * 4 + 2] = (double)ca;
    red[wave * 4 + 3] = (double)cb;
    redm[wave] = pmax;
  }
  __syncthreads();
  if (j == 0) {
    double ta = 0, tb = 0, na = 0, nbb = 0, mx = 0;
    for (int w = 0; w < NW; ++w) {
      ta += red[w * 4 + 0];
      tb += red[w * 4 + 1];
      na += red[w * 4 + 2];
      nbb += red[w * 4 + 3];
      mx = fmax(mx, redm[w]);
    }
    if (ra < a.N) { a.rowsum[ra] = ta; a.rowcnt[ra] = (long long)na; }
    if (!summed && hasb && rb < a.N) { a.rowsum[rb] = tb; a.rowcnt[rb] = (long long)nbb; }
    const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
    if (mx > 0.0 && bits > __hip_atomic_load(a.padmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(a.padmax, bits);
  }
}


// ------------------------------------------------------------ inverse rows, persistent + prefetch
// Same transform and epilogue as k_row_inv_rs, organised for ONE workgroup per CU that walks its
// share of the row pairs.  What k_row_inv_rs cannot do with one resident workgroup -- fetch the
// next rows while this pair is being transformed -- is done by the load unit itself: the two
// half-spectrum rows of the NEXT pair are copied HBM -> LDS by global_load_lds_dwordx4 (no
// registers, no waiting wave) while the three stages and both exchanges of the current pair run,
// and the first stage then takes its Hermitian-extended inputs from LDS.  It also shrinks the
// scheduling quantum from a workgroup of two pairs to one pair: the 2049 domain pairs of the
// 4097^2 stack are 8.004 rounds of 256 CUs, which cost 5 rounds of two-pair workgroups (4.002
// needed) but 9 of these half-length ones.
// LDS: exchange buffer (L * 8.5 B) + two staged rows (2 * (L/2 + 1) * 16 B) = 24.5 L bytes, so
// sizes up to L = 6400 (5184: 128 KB); larger ones keep k_row_inv_rs.  Row-major input only: gathering
// a column-major intermediate (PS_TINV) 16 bytes per line costs the load unit 122 us more per day
// than the contiguous stores save the column pass (25 us), asynchronous or not.
typedef __attribute__((address_space(3))) void* ps_lds_ptr;

template <int R1, int R2, int R3>
struct RsPLds {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  static constexpr int HS = (S::L / 2 + 1 + 63) & ~63;   // a staged row: whole 64-element chunks
  static constexpr size_t bytes = (size_t)(Y::XW + Y::RED) * sizeof(double) + 2 * (size_t)HS * sizeof(cplx);
  static constexpr bool fits = bytes <= (size_t)160 * 1024;
};

template <int R1, int R2, int R3>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR)) k_row_inv_rsp(RowInvArgs a, int npairs, int units) {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  using Z = RsPLds<R1, R2, R3>;
  constexpr int L = S::L;
  constexpr int NW = S::NTHR / 64;
  constexpr int NCH = Z::HS / 64;              // 1 KB chunks per staged row
  constexpr int H = L / 2 + 1;
  double* ex = reinterpret_cast<double*>(ps_lds_raw);
  double* red = ex + Y::XW;      // two sets of 5 * NW wave partials, then NW energy sums
  double* ered = red + 10 * NW;
  cplx* st = reinterpret_cast<cplx*>(ex + Y::XW + Y::RED);   // two staged rows, 32 elements of A, 32 of B, ... (prefetch)
  const int j0 = threadIdx.x, lane = j0 & 63, wave = j0 >> 6;
  const FftProg& P = a.prog;
  const cplx w2c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j0));
  const cplx w3c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j0 < S::T3 ? S::tw3(j0) : 0);

  // rows of unit u -> staging; every wave copies the chunks c = wave, wave + NW, ...  A chunk is 32 elements of
  // row A (lanes 0-31) and the same 32 of row B (lanes 32-63): st[64 c + l].  From a row-major source that is
  // two runs of 512 bytes; from the pair-interleaved one (pair_src: A[k], B[k] side by side from where row ra
  // starts) ONE run of 1 KB -- the copy takes the rows apart, and the first stage reads the same staging
  // either way.  (Staging whole rows one after the other, as this kernel did first, makes a lane stride of
  // 32 bytes out of the interleaved source: every line fetched twice, 0.27 ms more per 30-day launch at 5184.)
  auto prefetch = [&](int u) {
    const int b = u / npairs, pair = u - b * npairs;
    if (a.pad_quiet && 2 * pair >= a.N && a.pad_quiet[b]) return;   // never read: see the round below
    const cplx* src = a.src + (int64_t)b * a.src_bstride;
    const int ra = 2 * pair, rb = (ra + 1 < a.P) ? ra + 1 : ra;
    const cplx* pa = src + (int64_t)ra * a.ld;
    const int ksh = a.pair_src ? 1 : 0;
    const int bofs = (lane >> 5) ? (a.pair_src ? 1 : (rb - ra) * a.ld) : 0;
#pragma unroll
    for (int t = 0; t < (2 * NCH + NW - 1) / NW; ++t) {
      const int c = wave + t * NW;             // wave-uniform
      if (c < 2 * NCH) {
        int k = c * 32 + (lane & 31);
        k = k < H ? k : H - 1;                 // the last chunk: stay inside the row
        const cplx* g = pa + ((k << ksh) + bofs);
        cplx* d = st + c * 64;                 // + lane * 16 B by the instruction
        __builtin_amdgcn_global_load_lds(g, (ps_lds_ptr)d, 16, 0, 0);
      }
    }
  };

  // The row statistics of a pair (sums of the wave partials, two stores, the pre-checked atomicMax of
  // the pad maximum) used to end its round on one thread while everybody waited at the next barrier.
  // They are now written by the last wave's last lane -- idle in the first stage -- one round LATER,
  // behind that round's first barrier, out of the other waves' way (partials double-buffered).
  auto finalize = [&](int fb, int fra, const double* part) {
    const double* partm = part + 4 * NW;
    double ta = 0, tb = 0, na = 0, nbb = 0, m = 0;
    for (int w = 0; w < NW; ++w) {
      ta += part[w * 4 + 0];
      tb += part[w * 4 + 1];
      na += part[w * 4 + 2];
      nbb += part[w * 4 + 3];
      m = fmax(m, partm[w]);
    }
    const int frb = fra + 1;
    double* rowsum = a.rowsum + (int64_t)fb * a.stat_bstride;
    long long* rowcnt = a.rowcnt + (int64_t)fb * a.stat_bstride;
    if (fra < a.N) { rowsum[fra] = ta; rowcnt[fra] = (long long)na; }
    if (frb < a.N) { rowsum[frb] = tb; rowcnt[frb] = (long long)nbb; }
    unsigned long long* pm = a.padmax + fb;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    if (m > a.pad_floor && bits > __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(pm, bits);
  };
  bool pend = false;             // uniform: a finished pair whose partials wait in red[(par ^ 1) * 5 NW ...]
  int pend_b = 0, pend_ra = 0, par = 0;
  const bool finalizer = j0 == S::NTHR - 1;

  // Pad-only pairs of a day whose pad_quiet entry is set are not units of work at all: the column pass
  // summed |x|^2 over ALL pad-only rows of that day and found even that total below what one pair needs
  // to reach pad_floor, so every pair's own Parseval test (below) would skip it.  The round-robin walks
  // past them -- a skipped unit in the loop would start the next pair's prefetch and then wait for it
  // with nothing to do meanwhile, once per day and workgroup.
  auto next_unit = [&](int v) {   // uniform
    while (v < units && a.pad_quiet) {
      const int vb = v / npairs, vp = v - vb * npairs;
      if (2 * vp < a.N || !a.pad_quiet[vb]) break;
      v += (int)gridDim.x;
    }
    return v;
  };
  int u = next_unit((int)blockIdx.x), nu = 0;
  if (u < units) prefetch(u);
  PS_WAIT_VM0();
  for (; u < units; u = nu) {
    nu = next_unit(u + (int)gridDim.x);
    const int b = u / npairs, pair = u - b * npairs;
    const int ra = 2 * pair, rb = ra + 1;
    const bool hasb = rb < a.P;
    const bool pad_only = ra >= a.N;
    // opaque per-round copies: nothing derived from the thread index or the stage twiddles is
    // loop-invariant for the compiler, which would otherwise hoist ~100 registers of twiddle
    // powers and LDS addresses out of the loop (and spill)
    int j = j0;
    cplx w2 = w2c, w3 = w3c;
    asm volatile("" : "+v"(j), "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
    PS_BAR_LDS();                               // every wave's chunks of this unit have landed
    cplx x[S::RMAX];
    double energy = 0.0;
    if (j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) {
        const bool direct = (q < R1 / 2) || (q == R1 / 2 && j == 0);   // j + q T1 <= L/2
        const int i = j + q * S::T1;
        const int k = direct ? i : L - i;
        const int slot = k + (k & ~31);          // element k of row A; row B's 32 behind
        const cplx A = st[slot];
        cplx B = st[slot + 32];
        if (!hasb) B = make_double2(0.0, 0.0);
        if (direct) energy += A.x * A.x + A.y * A.y + B.x * B.x + B.y * B.y;
        x[q] = direct ? make_double2(A.x - B.y, A.y + B.x) : make_double2(A.x + B.y, B.x - A.y);
      }
    }
    if (pad_only) {   // uniform per workgroup
      energy = ps_wave_sum(energy);
      if (lane == 63) ered[wave] = energy;
    }
    PS_BAR_LDS();                               // staging is free again
    if (nu < units) prefetch(nu);
    if (pad_only) {
      // Parseval bound on the largest value of a pad-only pair (see k_row_inv)
      double e = 0.0;
      for (int w = 0; w < NW; ++w) e += ered[w];
      if (sqrt(2.0 * (double)a.P * e) * a.scale < a.pad_floor) {
        PS_WAIT_VM0();
        continue;
      }
    }
    if (j < S::T1) bfly<R1, PS_INV>(x);
    // (the previous pair's partials were complete before this round's first barrier)
    const bool fin = pend && finalizer;
    const int fb = pend_b, fra = pend_ra;
    const double* fpart = red + (par ^ 1) * 5 * NW;
    auto hook = [&]() {
      if (fin) finalize(fb, fra, fpart);
    };
    rs_tail<S, R1, R2, R3, PS_INV, true>(x, ex, j, w2, w3, hook);
    pend = false;
    // the prefetch has had the whole transform to land; waiting for it HERE (before this pair's
    // stores are issued) keeps the stores out of the wait at the top of the next round
    PS_WAIT_VM0();

    double* rec = a.nrec > 0 ? a.rec_multi[b] : a.rec + (int64_t)b * a.rec_bstride;
    double sa = 0.0, sb = 0.0, pmax = 0.0;
    int ca = 0, cb = 0;
    if (j < S::T3) {
      double* reca = rec + (int64_t)ra * a.N;
      double* recb = rec + (int64_t)rb * a.N;
      const bool dom_a = ra < a.N, dom_b = rb < a.N;
      const unsigned uj = (unsigned)j, uN = (unsigned)a.N;
      const bool dom_ab = dom_a && dom_b;
#pragma unroll
      for (int q = 0; q < R3; ++q) {
        const unsigned i = uj + (unsigned)(q * S::T3);
        const double va = x[q].x * a.scale, vb = x[q].y * a.scale;
        const double ta = va * a.stat_scale, tb = vb * a.stat_scale;
        if (dom_ab && (unsigned)((q + 1) * S::T3) <= uN) {
          reca[i] = va;
          recb[i] = vb;
          const bool ka = !(ta < a.negval), kb = !(tb < a.negval);
          sa += ka ? ta : 0.0;
          sb += kb ? tb : 0.0;
          ca += __popcll(__ballot(ka));     // per wave, on the scalar unit
          cb += __popcll(__ballot(kb));
        } else {
          const bool in = i < uN;
          const bool ina = in && dom_a, inb = in && dom_b;
          const bool ka = ina && !(ta < a.negval), kb = inb && !(tb < a.negval);
          sa += ka ? ta : 0.0;
          sb += kb ? tb : 0.0;
          ca += __popcll(__ballot(ka));
          cb += __popcll(__ballot(kb));
          pmax = fmax(pmax, ina ? 0.0 : va);
          pmax = fmax(pmax, (inb || !hasb) ? 0.0 : vb);
          if (ina) reca[i] = va;
          if (inb) recb[i] = vb;
        }
      }
    }
    sa = ps_wave_sum(sa);
    sb = ps_wave_sum(sb);
    ca = __builtin_amdgcn_readfirstlane(ca);   // the same in every lane that took part (lane 0 did if any did)
    cb = __builtin_amdgcn_readfirstlane(cb);
    pmax = ps_wave_max0(pmax);
    double* part = red + par * 5 * NW;
    if (lane == 63) {
      part[wave * 4 + 0] = sa;
      part[wave * 4 + 1] = sb;
      part[wave * 4 + 2] = (double)ca;
      part[wave * 4 + 3] = (double)cb;
      part[4 * NW + wave] = pmax;
    }
    pend = true;
    pend_b = b;
    pend_ra = ra;
    par ^= 1;
  }
  PS_BAR_LDS();
  if (pend && finalizer) finalize(pend_b, pend_ra, red + (par ^ 1) * 5 * NW);
}



// ------------------------------------------------------------ inverse rows, persistent, TWO ROLES in anti-phase
// k_row_inv_rsp runs ONE row pair per workgroup and CU at a time: its six waves sit 2-2-1-1 on the SIMDs
// (324 / 288 / 288 butterflies per stage at 5184 are 5.06 / 4.5 / 4.5 waves), every phase of a pair --
// first-stage reads, three stages with eight exchange barriers, scaling / threshold / statistics / stores --
// waits for the one before, and only the HBM traffic (prefetch by the load unit, stores) overlaps anything:
// 11.4 us per pair where the transform alone takes 5.9 and the memory side about 6.
// Here a workgroup has two roles of NTHR threads each, every role walking its own row pairs, half a period
// apart: while role X transforms its pair (the "T" half-period), role Y finishes the pair it transformed
// before -- scale, threshold statistics, stores -- and starts the copy of its next pair's two rows HBM -> LDS
// (the "S" half-period); then they swap.  The staging area is shared in time: the T role has read its rows
// out of it before the S role's copy starts, and that copy has the whole transform of the other role to
// land.  Same LDS as k_row_inv_rsp (exchange buffer + two staged rows), twelve waves 3-3-3-3.
// Barriers are shared (`s_barrier` counts arrivals per workgroup, whatever the program counter): a
// half-period is nine of them in either kind, and each role runs its own straight-line loop -- as ONE loop
// body with a wave-uniform branch between the barriers the register allocator, which is not
// path-sensitive, kept one role's values alive through the other role's code.
// Same loads, same transform code (rs_stage / bfly), same epilogue arithmetic in the same order as
// k_row_inv_rsp / k_row_inv_rs: bit-identical records, statistics and pad maxima
// (tests/test_solver_gpu.py::test_persistent_row_kernel_is_bit_identical).
template <int R1, int R2, int R3>
struct Rs2Lds {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  static constexpr int NW = S::NTHR / 64;
  static constexpr int RED = 8 * NW;                      // per role: 5 NW partials, NW energy sums (+ slack)
  static constexpr int HS = (S::L / 2 + 1 + 63) & ~63;    // a staged row: whole 64-element chunks
  static constexpr size_t bytes = (size_t)(Y::XW + 2 * RED) * sizeof(double) + 2 * (size_t)HS * sizeof(cplx);
  // two roles of NTHR threads in one workgroup; radices above 18 do not fit the 168 registers of 12 waves
  static constexpr bool ok = 2 * S::NTHR <= 1024 && S::RMAX <= 18 && bytes <= (size_t)160 * 1024;
};

template <int R1, int R2, int R3>
__global__ void __launch_bounds__((2 * Rs<R1, R2, R3>::NTHR)) k_row_inv_rs2(RowInvArgs a, int npairs, int units) {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  using Z = Rs2Lds<R1, R2, R3>;
  constexpr int L = S::L;
  constexpr int NW = S::NTHR / 64;
  constexpr int NCH = Z::HS / 64;              // 1 KB chunks per staged row
  constexpr int H = L / 2 + 1;
  double* ex = reinterpret_cast<double*>(ps_lds_raw);
  // (the scalar copy tells the compiler what it cannot see: everything derived from the role -- the unit a
  // role holds, its rows, the domain tests of the epilogue -- is uniform and lives in scalar registers)
  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x / S::NTHR);
  const int j0 = threadIdx.x - role * S::NTHR, lane = j0 & 63, wave = __builtin_amdgcn_readfirstlane(j0 >> 6);
  double* red = ex + Y::XW + role * Z::RED;                 // this role's 5 NW partials
  double* ered = red + 5 * NW;                              // ... and NW energy sums
  cplx* st = reinterpret_cast<cplx*>(ex + Y::XW + 2 * Z::RED);   // two staged rows in chunks of 32 + 32 (k_row_inv_rsp)
  const FftProg& P = a.prog;
  const cplx w2c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j0));
  const cplx w3c = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j0 < S::T3 ? S::tw3(j0) : 0);
  const int G2 = 2 * (int)gridDim.x;

  // a role's units: (2 b + role), + 2 G, ...; pad-only pairs of a day the column pass found quiet are no
  // units at all (see k_row_inv_rsp)
  auto next_unit = [&](int v) {   // uniform
    while (v < units && a.pad_quiet) {
      const int vb = v / npairs, vp = v - vb * npairs;
      if (2 * vp < a.N || !a.pad_quiet[vb]) break;
      v += G2;
    }
    return v;
  };
  auto count_units = [&](int r) {
    int n = 0;
    for (int v = next_unit(2 * (int)blockIdx.x + r); v < units; v = next_unit(v + G2)) ++n;
    return n;
  };
  // half-periods of the workgroup: role 0 transforms in the even ones (its last epilogue in 2 n0 - 1), role 1
  // in the odd ones (first copy in 0, last epilogue in 2 n1): both roles run the same number of barriers
  const int n0 = count_units(0), n1 = count_units(1);
  const int HP = max(2 * n0, n1 > 0 ? 2 * n1 + 1 : 0);

  cplx x[S::RMAX];
  int cur = next_unit(2 * (int)blockIdx.x + role);          // the unit this role is fetching / holds
  bool loaded = false;        // the copy of unit cur's rows is on its way to (or in) the staging area
  bool done = false;          // x holds unit cur's transformed rows: the epilogue is due

  // rows of unit u -> staging; every wave of the role copies the chunks c = wave, wave + NW, ... (32 elements
  // of row A and of row B each, see k_row_inv_rsp)
  auto prefetch = [&](int u) {
    const int b = u / npairs, pair = u - b * npairs;
    const cplx* src = a.src + (int64_t)b * a.src_bstride;
    const int ra = 2 * pair, rb = (ra + 1 < a.P) ? ra + 1 : ra;
    const cplx* pa = src + (int64_t)ra * a.ld;
    const int ksh = a.pair_src ? 1 : 0;
    const int bo = a.pair_src ? 1 : (rb - ra) * a.ld;          // uniform: from an element of row A to row B's
    // (opaque lane: the per-lane element indices are loop-invariant, and at this kernel's register limit the
    // compiler kept them in scratch -- a scratch reload and a wait for ALL outstanding copies in front of
    // every copy instruction, i.e. one copy at a time)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int bofs = (ln >> 5) ? bo : 0;
    ln &= 31;
#pragma unroll
    for (int t = 0; t < (2 * NCH + NW - 1) / NW; ++t) {
      const int c = wave + t * NW;             // wave-uniform
      if (c < 2 * NCH) {
        int k = c * 32 + ln;
        k = k < H ? k : H - 1;                 // the last chunk: stay inside the row
        const cplx* g = pa + ((k << ksh) + bofs);
        cplx* d = st + c * 64;                 // + lane * 16 B by the instruction
        __builtin_amdgcn_global_load_lds(g, (ps_lds_ptr)d, 16, 0, 0);
      }
    }
  };

  // the epilogue of a finished pair (as k_row_inv_rsp) runs in three pieces, in the shadow of the other
  // role's three butterfly intervals; its sums travel in registers, its statistics are written by one lane
  // in this role's next transform half-period (`pend`)
  int eb = 0, era = 0;          // day and first row of the pair whose epilogue is running
  double e_sa = 0.0, e_sb = 0.0, e_pmax = 0.0;
  int e_ca = 0, e_cb = 0;
  bool pend = false;            // partials of a finished epilogue wait in `red`
  int pend_b = 0, pend_ra = 0;
  auto epi_piece = [&](int j, auto q0c, auto q1c) {
    constexpr int Q0 = decltype(q0c)::value, Q1 = decltype(q1c)::value;
    const int ra = era, rb = era + 1;
    double* rec = a.nrec > 0 ? a.rec_multi[eb] : a.rec + (int64_t)eb * a.rec_bstride;
    if (j < S::T3) {
      double* reca = rec + (int64_t)ra * a.N;
      double* recb = rec + (int64_t)rb * a.N;
      const bool dom_a = ra < a.N, dom_b = rb < a.N;
      const unsigned uj = (unsigned)j, uN = (unsigned)a.N;
      const bool dom_ab = dom_a && dom_b;
      const bool hb = rb < a.P;
#pragma unroll
      for (int q = Q0; q < Q1; ++q) {
        const unsigned i = uj + (unsigned)(q * S::T3);
        const double va = x[q].x * a.scale, vb = x[q].y * a.scale;
        const double ta = va * a.stat_scale, tb = vb * a.stat_scale;
        if (dom_ab && (unsigned)((q + 1) * S::T3) <= uN) {
          reca[i] = va;
          recb[i] = vb;
          const bool ka = !(ta < a.negval), kb = !(tb < a.negval);
          e_sa += ka ? ta : 0.0;
          e_sb += kb ? tb : 0.0;
          e_ca += __popcll(__ballot(ka));
          e_cb += __popcll(__ballot(kb));
        } else {
          const bool in = i < uN;
          const bool ina = in && dom_a, inb = in && dom_b;
          const bool ka = ina && !(ta < a.negval), kb = inb && !(tb < a.negval);
          e_sa += ka ? ta : 0.0;
          e_sb += kb ? tb : 0.0;
          e_ca += __popcll(__ballot(ka));
          e_cb += __popcll(__ballot(kb));
          e_pmax = fmax(e_pmax, ina ? 0.0 : va);
          e_pmax = fmax(e_pmax, (inb || !hb) ? 0.0 : vb);
          if (ina) reca[i] = va;
          if (inb) recb[i] = vb;
        }
      }
    }
  };
  auto finalize = [&]() {       // one lane: the statistics of the pair whose partials are in `red`
    double ta = 0, tb = 0, na = 0, nbb = 0, m = 0;
    for (int w = 0; w < NW; ++w) {
      ta += red[w * 4 + 0];
      tb += red[w * 4 + 1];
      na += red[w * 4 + 2];
      nbb += red[w * 4 + 3];
      m = fmax(m, red[4 * NW + w]);
    }
    const int frb = pend_ra + 1;
    double* rowsum = a.rowsum + (int64_t)pend_b * a.stat_bstride;
    long long* rowcnt = a.rowcnt + (int64_t)pend_b * a.stat_bstride;
    if (pend_ra < a.N) { rowsum[pend_ra] = ta; rowcnt[pend_ra] = (long long)na; }
    if (frb < a.N) { rowsum[frb] = tb; rowcnt[frb] = (long long)nbb; }
    unsigned long long* pm = a.padmax + pend_b;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    if (m > a.pad_floor && bits > __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(pm, bits);
  };
  // ---- the two kinds of half-period, nine barriers each
  auto T_half = [&]() {
    int j = j0;
    cplx w2 = w2c, w3 = w3c;
    asm volatile("" : "+v"(j), "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));   // see k_row_inv_rsp
    bool act = loaded;          // this role has a pair to transform
    const int pair = cur % npairs;
    const bool pad_only = 2 * pair >= a.N;
    const bool hasb = 2 * pair + 1 < a.P;
    // (this wave's chunks of the staged rows have landed: it waited for them in its S half-period, BEFORE it
    // issued the last piece of its epilogue's stores -- a wait here would be a wait for those stores)
    PS_BAR_LDS();                                           // B0: ... and everybody's (and the last epilogue's partials)
    if (pend && j0 == S::NTHR - 1) finalize();
    pend = false;
    double energy = 0.0;
    if (act && j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) {
        const bool direct = (q < R1 / 2) || (q == R1 / 2 && j == 0);   // j + q T1 <= L/2
        const int i = j + q * S::T1;
        const int k = direct ? i : L - i;
        const int slot = k + (k & ~31);          // element k of row A; row B's 32 behind
        const cplx A = st[slot];
        cplx B = st[slot + 32];
        if (!hasb) B = make_double2(0.0, 0.0);
        if (direct) energy += A.x * A.x + A.y * A.y + B.x * B.x + B.y * B.y;
        x[q] = direct ? make_double2(A.x - B.y, A.y + B.x) : make_double2(A.x + B.y, B.x - A.y);
      }
    }
    if (act && pad_only) {      // Parseval bound on the largest value of a pad-only pair, see k_row_inv
      energy = ps_wave_sum(energy);
      if (lane == 63) ered[wave] = energy;
    }
    PS_BAR_LDS();                                           // B1: staging free for the other role's copy
    if (act && pad_only) {
      double e = 0.0;
      for (int w = 0; w < NW; ++w) e += ered[w];
      if (sqrt(2.0 * (double)a.P * e) * a.scale < a.pad_floor) act = false;   // cannot raise the flag: skipped
    }
    if (act && j < S::T1) {
      bfly<R1, PS_INV>(x);
      rs_put<R1, 0>(ex, S::x1_w(j), 1, x);
    }
    PS_BAR_LDS();                                           // B2
    if (act && j < S::T2) rs_get<R2, 0>(ex, S::x_r(j), S::X1_RS, x);
    PS_BAR_LDS();                                           // B3
    if (act && j < S::T1) rs_put<R1, 1>(ex, S::x1_w(j), 1, x);
    PS_BAR_LDS();                                           // B4
    if (act && j < S::T2) rs_get<R2, 1>(ex, S::x_r(j), S::X1_RS, x);
    // (twiddle powers are formed here, not hoisted above the exchange where they would cost 36 registers)
    asm volatile("" : "+v"(w2.x), "+v"(w2.y));
    if (act && j < S::T2) rs_stage<R2, PS_INV>(x, w2, true);
    PS_BAR_LDS();                                           // B5
    if (act && j < S::T2) rs_put<R2, 0>(ex, S::x2_w(j), 17, x);
    PS_BAR_LDS();                                           // B6
    if (act && j < S::T3) rs_get<R3, 0>(ex, S::x_r(j), S::X2_RS, x);
    PS_BAR_LDS();                                           // B7
    if (act && j < S::T2) rs_put<R2, 1>(ex, S::x2_w(j), 17, x);
    PS_BAR_LDS();                                           // B8
    if (act && j < S::T3) rs_get<R3, 1>(ex, S::x_r(j), S::X2_RS, x);
    asm volatile("" : "+v"(w3.x), "+v"(w3.y));
    if (act && j < S::T3) rs_stage<R3, PS_INV>(x, w3, true);
    // what this role does in its next half-period: the epilogue of this pair (or, skipped: the next pair)
    if (loaded) {
      loaded = false;
      if (act) done = true;
      else cur = next_unit(cur + G2);
    }
  };
  // piece k of 6 of the epilogue: q in [k R3 / 6, (k + 1) R3 / 6)
#define PS_EPI_PIECE(k) \
  if (epi) epi_piece(j, std::integral_constant<int, ((k) * R3) / 6>(), std::integral_constant<int, (((k) + 1) * R3) / 6>())
  auto S_half = [&]() {
    int j = j0;
    asm volatile("" : "+v"(j));
    const bool epi = done;
    if (epi) {
      eb = cur / npairs;
      era = 2 * (cur - eb * npairs);
      e_sa = e_sb = e_pmax = 0.0;
      e_ca = e_cb = 0;
      done = false;
      cur = next_unit(cur + G2);
    }
    PS_BAR_LDS();                                           // B0
    PS_BAR_LDS();                                           // B1: the other role has read the staging area
    loaded = false;
    if (cur < units) {                                      // the next pair of this role: its copy has the whole
      prefetch(cur);                                        // transform of the other role to land
      loaded = true;
    }
    PS_BAR_LDS();                                           // B2
    // the epilogue in six pieces, one per exchange interval of the other role's transform: a piece has to
    // be shorter than the interval it hides in, or the transform waits for it at the next barrier
    PS_EPI_PIECE(0);
    PS_BAR_LDS();                                           // B3
    PS_EPI_PIECE(1);
    PS_BAR_LDS();                                           // B4
    PS_EPI_PIECE(2);
    PS_BAR_LDS();                                           // B5
    PS_EPI_PIECE(3);
    PS_BAR_LDS();                                           // B6
    PS_EPI_PIECE(4);
    PS_BAR_LDS();                                           // B7
    PS_EPI_PIECE(5);
    PS_BAR_LDS();                                           // B8
    if (epi) {
      const double sa = ps_wave_sum(e_sa), sb = ps_wave_sum(e_sb);
      const int ca = __builtin_amdgcn_readfirstlane(e_ca), cb = __builtin_amdgcn_readfirstlane(e_cb);
      const double pmax = ps_wave_max0(e_pmax);
      if (lane == 63) {
        red[wave * 4 + 0] = sa;
        red[wave * 4 + 1] = sb;
        red[wave * 4 + 2] = (double)ca;
        red[wave * 4 + 3] = (double)cb;
        red[4 * NW + wave] = pmax;
      }
      pend = true;
      pend_b = eb;
      pend_ra = era;
    }
    PS_WAIT_VM0();                                          // the copy issued behind B1 (and the stores, long issued)
  };
#undef PS_EPI_PIECE
  auto idle_half = [&]() {
#pragma unroll
    for (int i = 0; i < 9; ++i) PS_BAR_LDS();
  };

  // role 0 transforms in the even half-periods, role 1 in the odd ones; each role runs its own loop
  if (role == 0) {
    if (cur < units) {          // prologue: its first pair comes in here
      prefetch(cur);
      PS_WAIT_VM0();
      loaded = true;
    }
    for (int k = 0; k < n0; ++k) {
      T_half();
      S_half();
    }
    for (int h = 2 * n0; h < HP; ++h) idle_half();
  } else {
    for (int k = 0; k < n1; ++k) {
      S_half();
      T_half();
    }
    int h = 2 * n1;
    if (n1 > 0) {
      S_half();
      ++h;
    }
    for (; h < HP; ++h) idle_half();
  }
  PS_BAR_LDS();                                             // the last epilogue's partials
  if (pend && j0 == S::NTHR - 1) finalize();
}


// ------------------------------------------------------------ forward rows
// Two real rows (ra, rb) -> z = a + i b, loaded by the first stage straight from the source
// (zero outside the row/column maps) -> three register stages -> Z in natural order, thread j
// holding k = j + q T3.  A_k = (Z_k + conj Z_{L-k})/2 and B_k = (Z_k - conj Z_{L-k})/(2i) need
// the mirrored element, which lives in another thread: one more exchange through LDS (real
// parts, then imaginary parts), then coalesced 16-byte stores of the half spectra (k <= L/2).
// Row pairs without a live source row are neither transformed nor written (the column pass
// is told the same live-row window), exactly like k_row_fwd.
template <int R1, int R2, int R3, int NP>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR * NP)) k_row_fwd_rs(RowFwdArgs a) {
  using S = Rs<R1, R2, R3>;
  using Y = RsInvLds<R1, R2, R3>;
  constexpr int L = S::L;
  if (pred_skip(a.pred)) return;
  const int half = threadIdx.x / S::NTHR;
  int unit = (int)blockIdx.x;
  if (a.tstride) {
    // column-major output: a 128-byte line holds 8 rows = 4 row pairs = 4 / NP workgroups, which
    // are consecutive blocks of ONE XCD (blocks go to the XCDs round-robin) so that the line is
    // assembled in that L2 (see k_colfull); gridDim.x is a multiple of 8 * (4 / NP)
    constexpr int KL = NP >= 4 ? 1 : 4 / NP;
    const int xcd = unit & 7, q = unit >> 3;
    unit = ((q / KL) * 8 + xcd) * KL + (q % KL);
  }
  const int pair = ((int)gridDim.x - 1 - unit) * NP + half;   // cheap (dead) pairs first, see k_row_inv_rs
  const int ra = 2 * pair, rb = ra + 1;
  if (ra >= a.P || pair < 0) return;   // ended waves do not take part in the barriers below
  double* ex = reinterpret_cast<double*>(ps_lds_raw) + half * (Y::XW + Y::RED);
  const int j = threadIdx.x - half * S::NTHR;
  const double* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  cplx* dst = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
  const int rlo = a.rowrange ? a.rowrange[2 * blockIdx.y] : 0;
  const int rhi = a.rowrange ? a.rowrange[2 * blockIdx.y + 1] : 0x7fffffff;
  SrcMap rmap = a.rmap, cmap = a.cmap;
  const bool trunc = a.trunc_pred && !pred_skip(a.trunc_pred);
  if (trunc) {
    rmap.n1 = rmap.n1 < a.trunc_n ? rmap.n1 : a.trunc_n;
    cmap.n1 = cmap.n1 < a.trunc_n ? cmap.n1 : a.trunc_n;
  }
  auto srow = [&](int r) {
    const int sr = r < a.P ? src_map(rmap, r) : -1;
    return (sr < rlo || sr > rhi) ? -1 : sr;
  };
  const int sa = srow(ra), sb = srow(rb);
  // row-major: element k of row r at r * ld + k; column-major (T layout): at k * tstride + r
  const int64_t kst = a.tstride ? (int64_t)a.tstride : 1;
  cplx* da = dst + (a.tstride ? (int64_t)ra : (int64_t)ra * a.ld);
  cplx* db = dst + (a.tstride ? (int64_t)rb : (int64_t)rb * a.ld);
  const bool hasb = rb < a.P;
  if (sa < 0 && sb < 0) {   // uniform per row pair
    const bool cut = trunc && ((ra < a.P && src_map(a.rmap, ra) >= 0) || (rb < a.P && src_map(a.rmap, rb) >= 0));
    if (a.skip_zero && !cut) return;
    const cplx z = make_double2(0.0, 0.0);
    for (int k = j; k < a.H; k += S::NTHR) {
      da[k * kst] = z;
      if (hasb) db[k * kst] = z;
    }
    return;
  }
  const FftProg& P = a.prog;
  const cplx w2 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, S::tw2(j));
  const cplx w3 = tw_lookup(P.tw_lo, P.tw_hi, P.tw_shift, j < S::T3 ? S::tw3(j) : 0);
  cplx x[S::RMAX];
  if (j < S::T1) {
    const double* pa = src + (int64_t)(sa >= 0 ? sa : 0) * a.src_ld;
    const double* pb = src + (int64_t)(sb >= 0 ? sb : 0) * a.src_ld;
#pragma unroll
    for (int q = 0; q < R1; ++q) {
      const int sc = src_map(cmap, j + q * S::T1);
      const bool ok = sc >= 0;
      const int c = ok ? sc : 0;
      const double va = pa[c], vb = pb[c];          // always in bounds; masked below
      x[q] = make_double2((ok && sa >= 0) ? va : 0.0, (ok && sb >= 0) ? vb : 0.0);
    }
    bfly<R1, PS_FWD>(x);
  }
  rs_tail<S, R1, R2, R3, PS_FWD>(x, ex, j, w2, w3);
  // mirrored elements through LDS: Z_k of thread j sits at word k = j + q T3
  cplx zm[R3 / 2 + 1];
  constexpr int NQ = R3 / 2 + 1;   // slots q with j + q T3 <= L/2 for some j
  __syncthreads();
  if (j < S::T3) {
#pragma unroll
    for (int q = 0; q < R3; ++q) ex[j + q * S::T3] = x[q].x;
  }
  __syncthreads();
  if (j < S::T3) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int k = j + q * S::T3;
      zm[q].x = ex[k ? L - k : 0];
    }
  }
  __syncthreads();
  if (j < S::T3) {
#pragma unroll
    for (int q = 0; q < R3; ++q) ex[j + q * S::T3] = x[q].y;
  }
  __syncthreads();
  if (j < S::T3) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int k = j + q * S::T3;
      zm[q].y = ex[k ? L - k : 0];
      if (k <= L / 2) {
        const cplx zk = x[q];
        da[k * kst] = make_double2(0.5 * (zk.x + zm[q].x), 0.5 * (zk.y - zm[q].y));
        if (hasb) db[k * kst] = make_double2(0.5 * (zk.y + zm[q].y), -0.5 * (zk.x - zm[q].x));
      }
    }
  }
}
