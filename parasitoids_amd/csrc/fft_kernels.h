// HIP pass kernels of the 2-D real FFT used by the day-chain solver.
//
// A 2-D real transform of the P x P torus is
//   forward : row pass (R2C, two real rows packed into one complex FFT)
//             -> column pass(es) over the half spectrum (H = P/2+1 columns)
//   inverse : column pass(es) (optionally fused with the spectral product
//             A_hat * B_hat of CalcSol.py:66) -> row pass (C2R) with the fused
//             epilogue of CalcSol.ifft2 / r_small_vals (CalcSol.py:35-41,:126-135).
// A column transform whose [P x W] tile does not fit LDS is split 4-step style
// into two sub-passes (P = L1 * L2) that each stream the spectrum once.
// Every pass stages its tile in LDS and runs the FftProg of fft_core.h there;
// HBM is touched with >= 64 B (normally 256-512 B) contiguous segments only.
//
// Latency structure: global loads are issued in unrolled batches of PS_UNROLL per
// thread before any of them is consumed, so a tile costs ~one HBM round trip, not one
// per loop iteration; index tables and the 4-step twiddles of the tile are staged in
// LDS once per workgroup.
#pragma once
#include "fft_core.h"

#define PS_UNROLL 5

// ---------------------------------------------------------------- arguments
struct SrcMap {  // torus index i -> source index, or -1 (zero)
  int n1, off1, lo2, off2;
};
// Flag-conditional passes (CalcSol.py:200-201) read the day's pad maximum (bit pattern of
// a non-negative double, written by the inverse row pass) and return at once unless it
// exceeds 1e-8 -- the boundary flag never travels to the host.
__device__ __forceinline__ bool pred_skip(const unsigned long long* pred) {
  return pred && !(__longlong_as_double((long long)*pred) > 1e-8);
}
__device__ __forceinline__ int src_map(const SrcMap& m, int i) {
  if (i < m.n1) return i + m.off1;
  if (i >= m.lo2) return i - m.lo2 + m.off2;
  return -1;
}

// Which input rows of a pass are known to be zero (and were never written by the pass
// before): row r is live iff it maps into the source (SrcMap) and, when a per-batch source
// row range is given, falls inside it (day kernels are small blobs inside a K x K box).
struct RowLive {
  int on;              // 0: every row is live
  SrcMap map;
  const int* range;    // device [batch][2] inclusive source-row range, or nullptr
};
__device__ __forceinline__ bool row_live(const RowLive& v, int r, int batch) {
  if (!v.on) return true;
  const int sr = src_map(v.map, r);
  if (sr < 0) return false;
  if (!v.range) return true;
  return sr >= v.range[2 * batch] && sr <= v.range[2 * batch + 1];
}

struct RowFwdArgs {
  const double* src;
  int64_t src_bstride;
  int src_ld;
  SrcMap rmap, cmap;
  cplx* dst;
  int64_t dst_bstride;
  int H, ld, P;  // dst is [P][ld], H valid columns
  int tstride;   // != 0 (register-resident kernels only): dst is column-major [H][tstride] (fft_colfull_kernels.h)
  int rp;        // row pairs per block
  int skip_zero;  // all-zero row pairs are not written (the next pass knows they are zero)
  const int* rowrange;  // device [batch][2] inclusive live source-row range, or nullptr
  int nblocks;          // work items along x (grid.x may be smaller: the kernel loops)
  const unsigned long long* pred;
  // k_row_fwd_rs: when *trunc_pred is above the flag threshold the source counts as truncated to its
  // first trunc_n rows and columns (PS_MODE_FOLD: the torus of a flagged day, CalcSol.py:200-201); rows
  // that are live only without the truncation are written as zeros (the column pass still reads them)
  const unsigned long long* trunc_pred;
  int trunc_n;
  FftProg prog;
};

struct ColArgs {
  const cplx* src;
  const cplx* src2;  // optional: multiply (spectral product)
  cplx* prod_dst;    // optional: store the product (new A_hat)
  cplx* dst;
  int64_t src_bstride, src2_bstride, prod_bstride, dst_bstride;
  int ld, ncols, wsh, n_outer;
  int in_base_mul, in_stride, out_base_mul, out_stride;
  int tw_mode;  // 0 none, 1 post-multiply w_P^(o k) (forward), 2 pre-multiply conj (inverse)
  const cplx* tp_lo;
  const cplx* tp_hi;
  int tp_shift;
  RowLive live;  // input rows known to be zero are not read
  int nblocks;   // work items along x (grid.x may be smaller: the kernel loops)
  const unsigned long long* pred;
  FftProg prog;
};

struct RowInvArgs {
  const cplx* src;
  int64_t src_bstride;
  int H, ld, P, N;
  int tstride;   // != 0 (register-resident kernel only): src is column-major [H][tstride] (full-column pipeline)
  int pair_src;  // != 0 (register-resident kernels, P even): rows 2p and 2p + 1 interleaved element by element,
                 // [P/2][ld][2] (ColFullArgs::dst_t == 2, fft_colfull_kernels.h: colfull_dst)
  int rp;
  double scale;  // 1 / Pfft^2
  double* rec;   // [N][N] raw real solution
  int64_t rec_bstride;
  double negval, stat_scale;  // stats on v * stat_scale >= negval
  double* rowsum;             // [N]
  long long* rowcnt;          // [N]
  unsigned long long* padmax; // bits of max(pad, 0)
  double pad_floor;           // pad maxima at or below this are not published (0.5e-8: only what can raise the flag)
  int64_t stat_bstride;       // per-batch stride of rowsum/rowcnt (padmax: 1)
  // k_row_inv_rsp (persistent, prefetching; register-resident sizes up to 6400): chosen by the
  // launcher when the input does not come straight out of the Infinity Cache (full-column pipeline)
  int persistent;
  // per batch entry: 1 = the column pass found the pad-only row pairs of this day too quiet to raise
  // the flag (k_pad_quiet: the sum of the per-pair Parseval tests below) -- k_row_inv_rsp then neither
  // fetches nor tests them; nullptr = unknown
  const int* pad_quiet;
  int nrec;                   // > 0: batch entry b writes rec_multi[b] (the days of a chained group)
  double* rec_multi[32];      // = PS_MAX_GROUP_DAYS (ps_solver.hip)
  FftProg prog;
};

// -------------------------------------------------------------- LDS helpers
// twiddle block of a program in LDS: lo | hi | gen (contiguous in HBM too)
__device__ __forceinline__ int tw_count(const FftProg& P) { return P.n_lo + P.n_hi + P.n_gen; }
__device__ __forceinline__ void load_tw(cplx* tlo, cplx* thi, const FftProg& P) {
  const int n = tw_count(P);
  for (int t = threadIdx.x; t < n; t += blockDim.x) tlo[t] = P.tw_lo[t];
}

template <int DIR, bool GEN, bool BIG>
__device__ __forceinline__ void run_stage_sel(cplx* data, const cplx* tlo, const cplx* thi,
                                              const FftProg& P, int s, int mode, int nb, int wsh,
                                              int bs) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  switch (P.radix[s]) {
    case 2: run_stage_r<2, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 3: run_stage_r<3, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 4: run_stage_r<4, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 5: run_stage_r<5, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 7: run_stage_r<7, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 8: run_stage_r<8, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 9: run_stage_r<9, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 16: if (BIG) run_stage_r<16, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 18: if (BIG) run_stage_r<18, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    default:
      if (GEN) {
        if (DIR == PS_INV && P.m[s] > 1) {   // uniform: input twiddles once per element
          gen_pretwiddle<DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr);
          __syncthreads();
        }
        run_stage_generic<DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr);
      }
      break;
  }
}

template <int DIR, bool GEN, bool BIG = false>
__device__ __forceinline__ void lds_fft(cplx* data, const cplx* tlo, const cplx* thi,
                                        const FftProg& P, int mode, int nb, int wsh, int bs) {
  if (DIR == PS_FWD) {
    for (int s = 0; s < P.ns; ++s) {
      run_stage_sel<DIR, GEN, BIG>(data, tlo, thi, P, s, mode, nb, wsh, bs);
      __syncthreads();
    }
  } else {
    for (int s = P.ns - 1; s >= 0; --s) {
      run_stage_sel<DIR, GEN, BIG>(data, tlo, thi, P, s, mode, nb, wsh, bs);
      __syncthreads();
    }
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char ps_lds_raw[];

// ------------------------------------------------------------ forward rows
// Two real rows (ra, rb) -> z = a + i b -> complex FFT -> A = (Z_k + conj Z_{L-k})/2,
// B = (Z_k - conj Z_{L-k})/(2i).  Rows outside the source map are zero.
template <bool GEN, bool BIG>
__device__ __forceinline__ void row_fwd_block(const RowFwdArgs& a, const int bx) {
  const FftProg& P = a.prog;
  const int L = P.L;
  const int pitch = row_pitch(P);
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + (size_t)a.rp * pitch;
  cplx* thi = tlo + P.n_lo;
  const double* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  cplx* dst = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
  const int pair0 = bx * a.rp;
  const int nthr = blockDim.x;
  // any non-zero source row in this block?
  const int rlo = a.rowrange ? a.rowrange[2 * blockIdx.y] : 0;
  const int rhi = a.rowrange ? a.rowrange[2 * blockIdx.y + 1] : 0x7fffffff;
  auto srow = [&](int r) {
    const int sr = r < a.P ? src_map(a.rmap, r) : -1;
    return (sr < rlo || sr > rhi) ? -1 : sr;
  };
  bool any = false;
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    if (srow(ra) >= 0) any = true;
    if (srow(rb) >= 0) any = true;
  }
  if (!any) {
    if (a.skip_zero) return;
    const cplx z = make_double2(0.0, 0.0);
    for (int b = 0; b < a.rp; ++b) {
      const int ra = 2 * (pair0 + b), rb = ra + 1;
      for (int k = threadIdx.x; k < a.H; k += nthr) {
        if (ra < a.P) dst[(int64_t)ra * a.ld + k] = z;
        if (rb < a.P) dst[(int64_t)rb * a.ld + k] = z;
      }
    }
    return;
  }
  load_tw(tlo, thi, P);
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    const int sa = srow(ra);
    const int sb = srow(rb);
    const double* pa = src + (int64_t)sa * a.src_ld;
    const double* pb = src + (int64_t)sb * a.src_ld;
    for (int i0 = threadIdx.x; i0 < L; i0 += nthr * PS_UNROLL) {
      double va[PS_UNROLL], vb[PS_UNROLL];
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) {
        const int i = i0 + u * nthr;
        va[u] = 0.0;
        vb[u] = 0.0;
        if (i < L) {
          const int sc = src_map(a.cmap, i);
          if (sc >= 0) {
            if (sa >= 0) va[u] = pa[sc];
            if (sb >= 0) vb[u] = pb[sc];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) {
        const int i = i0 + u * nthr;
        if (i < L) data[b * pitch + row_phys(P, i)] = make_double2(va[u], vb[u]);
      }
    }
  }
  __syncthreads();
  lds_fft<PS_FWD, GEN, BIG>(data, tlo, thi, P, PS_MODE_ROW, a.rp, 0, pitch);
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    if (ra >= a.P) break;
    const cplx* d = data + b * pitch;
    for (int k0 = threadIdx.x; k0 < a.H; k0 += nthr * PS_UNROLL) {
      unsigned pk[PS_UNROLL], pm[PS_UNROLL];
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) {
        const int k = k0 + u * nthr;
        pk[u] = 0;
        pm[u] = 0;
        if (k < a.H) {
          pk[u] = P.pos_phys[k];
          pm[u] = P.pos_phys[k ? L - k : 0];
        }
      }
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) {
        const int k = k0 + u * nthr;
        if (k < a.H) {
          const cplx zk = d[pk[u]];
          const cplx zm = d[pm[u]];
          const cplx A = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
          const cplx B = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
          dst[(int64_t)ra * a.ld + k] = A;
          if (rb < a.P) dst[(int64_t)rb * a.ld + k] = B;
        }
      }
    }
  }
}

// Grid: a.nblocks work items; flag-conditional launches use a small grid that loops, so an
// un-flagged day costs a ~2 us launch instead of dispatching thousands of empty workgroups.
template <bool GEN, bool BIG>
__global__ void __launch_bounds__(BIG ? 512 : 1024) k_row_fwd(RowFwdArgs a) {
  if (pred_skip(a.pred)) return;
  for (int bx = blockIdx.x; bx < a.nblocks; bx += gridDim.x) {
    row_fwd_block<GEN, BIG>(a, bx);
    if (bx + (int)gridDim.x < a.nblocks) __syncthreads();
  }
}

// ---------------------------------------------------------------- columns
template <int DIR, bool GEN>
__device__ __forceinline__ void col_block(const ColArgs& a, const int bx) {
  const FftProg& P = a.prog;
  const int L = P.L;
  const int W = 1 << a.wsh;
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + ((size_t)L << a.wsh);
  cplx* thi = tlo + P.n_lo;
  cplx* stw = tlo + tw_count(P);                     // [L] 4-step twiddle of tile row
  int* spos = reinterpret_cast<int*>(stw + L);       // [L] digit-reversed LDS row
  const int ntiles = (a.ncols + W - 1) >> a.wsh;
  const int tile = bx % ntiles;
  const int o = bx / ntiles;
  const int c0 = tile << a.wsh;
  const int nthr = blockDim.x;
  const cplx* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  const cplx* src2 = a.src2 ? a.src2 + (int64_t)blockIdx.y * a.src2_bstride : nullptr;
  cplx* prod = a.prod_dst ? a.prod_dst + (int64_t)blockIdx.y * a.prod_bstride : nullptr;
  cplx* dst = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
  const bool tw_on = a.tw_mode != 0 && o != 0;
  const int tot = L << a.wsh;
  const int in_base = o * a.in_base_mul, out_base = o * a.out_base_mul;
  cplx v[PS_UNROLL], v2[PS_UNROLL];
  auto load_batch = [&](int idx0) {
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      const int row = idx >> a.wsh, col = c0 + (idx & (W - 1));
      v[u] = make_double2(0.0, 0.0);
      v2[u] = make_double2(1.0, 0.0);
      const int grow = in_base + row * a.in_stride;
      if (idx < tot && col < a.ncols && row_live(a.live, grow, blockIdx.y)) {
        const int64_t g = (int64_t)grow * a.ld + col;
        v[u] = src[g];
        if (src2) v2[u] = src2[g];
      }
    }
  };
  auto store_batch = [&](int idx0) {
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      if (idx >= tot) continue;
      const int row = idx >> a.wsh, c = idx & (W - 1);
      cplx x = v[u];
      if (src2) {
        x = cmul(x, v2[u]);
        if (prod && c0 + c < a.ncols) prod[(int64_t)(in_base + row * a.in_stride) * a.ld + c0 + c] = x;
      }
      if (a.tw_mode == 2 && tw_on) x = cmulc(x, stw[row]);
      const int lrow = DIR == PS_INV ? spos[row] : row;
      data[(lrow << a.wsh) + c] = x;
    }
  };
  // the first batch of tile loads is in flight while the tables are fetched: one HBM round
  // trip for both instead of two
  load_batch(threadIdx.x);
  load_tw(tlo, thi, P);
  for (int r = threadIdx.x; r < L; r += nthr) {
    spos[r] = (int)P.pos[r];
    if (tw_on) stw[r] = tw_lookup(a.tp_lo, a.tp_hi, a.tp_shift, o * r);
  }
  __syncthreads();
  store_batch(threadIdx.x);
  for (int idx0 = threadIdx.x + nthr * PS_UNROLL; idx0 < tot; idx0 += nthr * PS_UNROLL) {
    load_batch(idx0);
    store_batch(idx0);
  }
  __syncthreads();
  lds_fft<DIR, GEN>(data, tlo, thi, P, PS_MODE_COL, W, a.wsh, 0);
  for (int idx = threadIdx.x; idx < tot; idx += nthr) {
    const int row = idx >> a.wsh, c = idx & (W - 1);
    const int col = c0 + c;
    if (col >= a.ncols) continue;
    const int lrow = DIR == PS_FWD ? spos[row] : row;
    cplx x = data[(lrow << a.wsh) + c];
    if (a.tw_mode == 1 && tw_on) x = cmul(x, stw[row]);
    dst[(int64_t)(out_base + row * a.out_stride) * a.ld + col] = x;
  }
}

template <int DIR, bool GEN>
__global__ void __launch_bounds__(256, 4) k_col(ColArgs a) {
  if (pred_skip(a.pred)) return;
  for (int bx = blockIdx.x; bx < a.nblocks; bx += gridDim.x) {
    col_block<DIR, GEN>(a, bx);
    if (bx + (int)gridDim.x < a.nblocks) __syncthreads();
  }
}

// ------------------------------------------------- fused convolution column pass
// The last forward sub-pass of the day kernel, the spectral product with the state
// (CalcSol.py:66) and the first inverse sub-pass work on the SAME tile (fixed outer
// index o = k1, columns c0..c0+W): forward FFT over r2 leaves B_hat[k1 + L1 k2] at LDS
// row pos[k2], which is exactly where the inverse program wants its input.  So B_hat
// never goes to HBM:  load kernel tile -> FFT -> x *= A_hat (store new A_hat) -> iFFT ->
// store.  Rows: kernel/out rows o*L2 + i, state rows o + L1*k.  Single-pass columns are
// the case L1 = 1.
struct ColFusedArgs {
  const cplx* src;   // kernel after row pass (+ first column sub-pass when split)
  cplx* state;       // A_hat, read; overwritten with the product when store_prod
  cplx* dst;
  int64_t src_bstride;
  int64_t dst_dstride; // MULTI: distance between the days' outputs
  int ld, ncols, wsh, L1, L2, store_prod;
  RowLive live;      // kernel rows known to be zero are not read
  RowLive live2;     // DUAL: state rows known to be zero
  // direct != 0 (split columns, sparse day kernels): `src` is the ROW-pass output and the
  // kernel's first column sub-pass is evaluated as a direct sum over its few live rows
  // (kt_direct_fill); tp_* = two-level twiddle table of w_N, N = L1 L2; mgL2 = ps_magic(L2)
  int direct;
  int no_conj;       // A/B knob PS_NO_CONJ: direct fill of a conjugate pair with two twiddle tables
  const cplx* tp_lo;
  const cplx* tp_hi;
  int tp_shift;
  uint32_t mgL2;
  FftProg prog;      // length L2
};

// First column sub-pass of a sparse day kernel without the HBM round trip.  With row index
// n = i + L2 j (i < L2, j < L1) the sub-pass output the fused kernels consume is
//   Y[o L2 + i] = w_N^{i o} * sum_j R[i + L2 j] w_L1^{j o}            (R = row-pass output),
// an L1-point DFT per i of which only the live rows (the kernel's support, two intervals of
// the wrapped K x K box: RowLive) are non-zero -- a handful of terms per i for the compact
// kernels prob_mass produces.  The sums are evaluated right where the tile is needed, so the
// intermediate spectrum (one write + one read of P^2/2 complex per day) never exists and the
// separate k_col launch disappears; the live rows are re-read by every o, out of L2.
// Fills data[(i << psh) + (day << wsh) + c]; stw[i] = w_N^{i o}, wj[j] = w_L1^{j o} in LDS.
__device__ __forceinline__ void kt_direct_tables(const ColFusedArgs& a, cplx* stw, cplx* wj, int o) {
  for (int r = threadIdx.x; r < a.L2; r += blockDim.x) stw[r] = tw_lookup(a.tp_lo, a.tp_hi, a.tp_shift, o * r);
  for (int j = threadIdx.x; j < a.L1; j += blockDim.x)
    wj[j] = tw_lookup(a.tp_lo, a.tp_hi, a.tp_shift, ((j * o) % a.L1) * a.L2);
}
// G outer indices o .. o+G-1 per workgroup share every row load: column group (g * ND + day)
// of the tile, tables stw[g * L2 + i], wj[g * L1 + j]
template <int G>
__device__ __forceinline__ void kt_direct_fill(const ColFusedArgs& a, cplx* data, const cplx* stw, const cplx* wj,
                                               int c0, int psh, int ndsh, const int* rng = nullptr) {
  constexpr int JB = 8, EL = 2;   // up to EL * JB row loads in flight per thread
  const int L = a.L2, W = 1 << a.wsh, WN = W << ndsh;
  const int totn = L << (a.wsh + ndsh);
  const int N = a.L1 * a.L2;
  const SrcMap& m = a.live.map;
  const int nthr = blockDim.x;
  for (int idx0 = threadIdx.x; idx0 < totn; idx0 += EL * nthr) {
    int row[EL], nA[EL], jA[EL], jB[EL], T[EL], slot[EL];
    const cplx* src[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int idx = idx0 + e * nthr;
      const int i = idx >> (a.wsh + ndsh), cc = idx & (WN - 1);
      const int day = cc >> a.wsh, c = cc & (W - 1), col = c0 + c;
      row[e] = i;
      slot[e] = (i << psh) + (day << a.wsh) + c;
      nA[e] = 0; jA[e] = 0; jB[e] = 0; T[e] = 0;
      src[e] = a.src;
      if (idx < totn && col < a.ncols) {
        int x0[2] = {0, 1}, x1[2] = {N - 1, 0};
        if (a.live.on) {
          int lo = -(1 << 30), hi = 1 << 30;
          if (rng) {   // LDS copy (fused_prologue): no global round trip ahead of the row loads
            lo = rng[2 * day];
            hi = rng[2 * day + 1];
          } else if (a.live.range) {
            lo = a.live.range[2 * day];
            hi = a.live.range[2 * day + 1];
          }
          x0[0] = max(0, lo - m.off1);
          x1[0] = min(m.n1 - 1, hi - m.off1);
          x0[1] = max(m.lo2, lo - m.off2 + m.lo2);
          x1[1] = min(N - 1, hi - m.off2 + m.lo2);
        }
        int j0[2], cnt[2];
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
          j0[seg] = x0[seg] <= i ? 0 : ps_div(x0[seg] - i + L - 1, a.mgL2);
          const int j1 = x1[seg] < i ? -1 : ps_div(x1[seg] - i, a.mgL2);
          cnt[seg] = max(0, j1 - j0[seg] + 1);
        }
        jA[e] = j0[0]; nA[e] = cnt[0]; jB[e] = j0[1]; T[e] = cnt[0] + cnt[1];
        src[e] = a.src + day * a.src_bstride + col;
      }
      if (idx >= totn) slot[e] = -1;
    }
    int tmax = 0;
#pragma unroll
    for (int e = 0; e < EL; ++e) tmax = max(tmax, T[e]);
    cplx acc[EL][G];
#pragma unroll
    for (int e = 0; e < EL; ++e)
#pragma unroll
      for (int g = 0; g < G; ++g) acc[e][g] = make_double2(0.0, 0.0);
    for (int tb = 0; tb < tmax; tb += JB) {
      cplx v[EL][JB];
      int jj[EL][JB];
#pragma unroll
      for (int e = 0; e < EL; ++e)
#pragma unroll
        for (int t = 0; t < JB; ++t) {
          const int tt = tb + t;
          v[e][t] = make_double2(0.0, 0.0);
          jj[e][t] = 0;
          if (tt < T[e]) {
            const int j = tt < nA[e] ? jA[e] + tt : jB[e] + tt - nA[e];
            jj[e][t] = j;
            v[e][t] = src[e][(int64_t)(row[e] + L * j) * a.ld];
          }
        }
#pragma unroll
      for (int e = 0; e < EL; ++e)
#pragma unroll
        for (int t = 0; t < JB; ++t)
#pragma unroll
          for (int g = 0; g < G; ++g) acc[e][g] = cadd(acc[e][g], cmul(v[e][t], wj[g * a.L1 + jj[e][t]]));
    }
#pragma unroll
    for (int e = 0; e < EL; ++e)
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (slot[e] >= 0) data[slot[e] + (g << (a.wsh + ndsh))] = cmul(acc[e][g], stw[g * L + row[e]]);
  }
}

// ---- paired direct fill (k_col_fused_multi, G == 2) -------------------------------------------
// The two outer indices of a workgroup are a CONJUGATE pair (o, L1 - o): w_L1^{j (L1 - o)} =
// conj(w_L1^{j o}), so with v = R[i + L2 j] and w = w_L1^{j o} the four real sums
//   S1 = sum vx wx,  S2 = sum vy wy,  S3 = sum vx wy,  S4 = sum vy wx
// give BOTH outputs:  Y_o = (S1 - S2, S3 + S4),  Y_{L1-o} = (S1 + S2, S4 - S3)  --  four FMAs
// per term instead of two complex multiply-adds (twelve operations).  The self-conjugate indices
// 0 and L1/2 form the pair oq == 0, which takes the generic two-table route.
// `ent[i * ND + day]` (LDS, built once per workgroup by fused_prologue) holds the live terms of
// tile row i for that day's kernel: jA | nA << 8 | jB << 16 | nB << 24 -- terms j = jA .. jA+nA-1
// (rows below the kernel centre, torus rows [0, n1)) and jB .. jB+nB-1 (wrapped rows [lo2, N)).
__device__ __forceinline__ int pair_o(int oq, int g, int L1) { return g == 0 ? oq : (oq == 0 ? (L1 >> 1) : L1 - oq); }

__device__ __forceinline__ unsigned kt_direct_entry(const ColFusedArgs& a, int i, int lo, int hi) {
  const int L = a.L2, N = a.L1 * a.L2;
  const SrcMap& m = a.live.map;
  int x0[2] = {0, 1}, x1[2] = {N - 1, 0};
  if (a.live.on) {
    x0[0] = max(0, lo - m.off1);
    x1[0] = min(m.n1 - 1, hi - m.off1);
    x0[1] = max(m.lo2, lo - m.off2 + m.lo2);
    x1[1] = min(N - 1, hi - m.off2 + m.lo2);
  }
  unsigned e = 0;
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const int j0 = x0[seg] <= i ? 0 : ps_div(x0[seg] - i + L - 1, a.mgL2);
    const int j1 = x1[seg] < i ? -1 : ps_div(x1[seg] - i, a.mgL2);
    const int cnt = max(0, j1 - j0 + 1);
    e |= ((unsigned)(j0 & 255) | ((unsigned)cnt << 8)) << (16 * seg);
  }
  return e;
}

template <int ND>
__device__ __forceinline__ void kt_direct_fill_pair(const ColFusedArgs& a, cplx* data, const cplx* stw, const cplx* wj,
                                                    int c0, int psh, const unsigned* ent, bool conj_pair) {
  constexpr int NDSH = ND == 1 ? 0 : (ND == 2 ? 1 : (ND == 4 ? 2 : 3));
  constexpr int JB = 8, EL = 2;   // up to EL * JB row loads in flight per thread
  const int L = a.L2, W = 1 << a.wsh, WN = W << NDSH;
  const int totn = L << (a.wsh + NDSH);
  const int nthr = blockDim.x;
  const unsigned strideJ = (unsigned)__umul24(L, a.ld);   // elements between consecutive terms
  const int gstep = 1 << (a.wsh + NDSH);                   // tile columns between the pair's halves
  for (int idx0 = threadIdx.x; idx0 < totn; idx0 += EL * nthr) {
    int row[EL], nA[EL], jA[EL], jBp[EL], T[EL], slot[EL];
    unsigned baseA[EL], baseB[EL];
    const cplx* src[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int idx = idx0 + e * nthr;
      const int i = idx >> (a.wsh + NDSH), cc = idx & (WN - 1);
      const int day = cc >> a.wsh, c = cc & (W - 1), col = c0 + c;
      row[e] = i;
      slot[e] = idx < totn ? (i << psh) + (day << a.wsh) + c : -1;
      nA[e] = 0; jA[e] = 0; jBp[e] = 0; T[e] = 0; baseA[e] = 0; baseB[e] = 0;
      src[e] = a.src;
      if (idx < totn && col < a.ncols) {
        const unsigned en = ent[i * ND + day];
        jA[e] = (int)(en & 255u);
        nA[e] = (int)((en >> 8) & 255u);
        const int jB = (int)((en >> 16) & 255u);
        T[e] = nA[e] + (int)(en >> 24);
        jBp[e] = jB - nA[e];
        baseA[e] = (unsigned)__umul24(i + __umul24(L, jA[e]), a.ld);
        baseB[e] = (unsigned)__umul24(i + __umul24(L, jB), a.ld) - (unsigned)nA[e] * strideJ;
        src[e] = a.src + day * a.src_bstride + col;
      }
    }
    int tmax = 0;
#pragma unroll
    for (int e = 0; e < EL; ++e) tmax = max(tmax, T[e]);
    if (conj_pair) {
      double S[EL][4];
#pragma unroll
      for (int e = 0; e < EL; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) S[e][q] = 0.0;
      for (int tb = 0; tb < tmax; tb += JB) {
        cplx v[EL][JB];
        int jj[EL][JB];
#pragma unroll
        for (int e = 0; e < EL; ++e)
#pragma unroll
          for (int t = 0; t < JB; ++t) {
            const int tt = tb + t;
            v[e][t] = make_double2(0.0, 0.0);
            const bool inA = tt < nA[e];
            jj[e][t] = (inA ? jA[e] : jBp[e]) + tt;
            if (tt < T[e]) v[e][t] = src[e][(inA ? baseA[e] : baseB[e]) + (unsigned)tt * strideJ];
            else jj[e][t] = 0;
          }
#pragma unroll
        for (int e = 0; e < EL; ++e)
#pragma unroll
          for (int t = 0; t < JB; ++t) {
            const cplx w = wj[jj[e][t]];
            S[e][0] = fma(v[e][t].x, w.x, S[e][0]);
            S[e][1] = fma(v[e][t].y, w.y, S[e][1]);
            S[e][2] = fma(v[e][t].x, w.y, S[e][2]);
            S[e][3] = fma(v[e][t].y, w.x, S[e][3]);
          }
      }
#pragma unroll
      for (int e = 0; e < EL; ++e)
        if (slot[e] >= 0) {
          data[slot[e]] = cmul(make_double2(S[e][0] - S[e][1], S[e][2] + S[e][3]), stw[row[e]]);
          data[slot[e] + gstep] = cmul(make_double2(S[e][0] + S[e][1], S[e][3] - S[e][2]), stw[L + row[e]]);
        }
    } else {
      cplx acc[EL][2];
#pragma unroll
      for (int e = 0; e < EL; ++e) acc[e][0] = acc[e][1] = make_double2(0.0, 0.0);
      for (int tb = 0; tb < tmax; tb += JB) {
        cplx v[EL][JB];
        int jj[EL][JB];
#pragma unroll
        for (int e = 0; e < EL; ++e)
#pragma unroll
          for (int t = 0; t < JB; ++t) {
            const int tt = tb + t;
            v[e][t] = make_double2(0.0, 0.0);
            const bool inA = tt < nA[e];
            jj[e][t] = (inA ? jA[e] : jBp[e]) + tt;
            if (tt < T[e]) v[e][t] = src[e][(inA ? baseA[e] : baseB[e]) + (unsigned)tt * strideJ];
            else jj[e][t] = 0;
          }
#pragma unroll
        for (int e = 0; e < EL; ++e)
#pragma unroll
          for (int t = 0; t < JB; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[e][g] = cadd(acc[e][g], cmul(v[e][t], wj[g * a.L1 + jj[e][t]]));
      }
#pragma unroll
      for (int e = 0; e < EL; ++e)
        if (slot[e] >= 0) {
          data[slot[e]] = cmul(acc[e][0], stw[row[e]]);
          data[slot[e] + gstep] = cmul(acc[e][1], stw[L + row[e]]);
        }
    }
  }
}

// Workgroup prologue of the fused kernels: FFT twiddle block, digit-reversal table and (direct
// mode) the stw / wj tables of G outer indices, with ALL global loads of a round issued before
// the first LDS store -- one memory round trip instead of one per table (they used to be six
// of a workgroup's ~25 us).
template <int G>
__device__ __forceinline__ void fused_prologue(const ColFusedArgs& a, cplx* tlo, int* spos, cplx* stw, cplx* wj,
                                               int oq, unsigned* ent, int nd) {
  const FftProg& P = a.prog;
  const int n = tw_count(P), L = P.L;
  // outer indices of the workgroup: o0 + g (G == 1: just oq), or the conjugate pair (pair_o)
  const int og1 = G == 2 ? pair_o(oq, 1, a.L1) : oq;
  // per (tile row, day) live terms of the kernels (direct mode) -> LDS (kt_direct_entry)
  if (a.direct)
    for (int t = threadIdx.x; t < a.L2 * nd; t += blockDim.x) {
      const int i = t / nd, day = t - i * nd;
      int lo = -(1 << 30), hi = 1 << 30;
      if (a.live.range) { lo = a.live.range[2 * day]; hi = a.live.range[2 * day + 1]; }
      ent[t] = kt_direct_entry(a, i, lo, hi);
    }
  const int n_stw = a.direct ? G * a.L2 : 0, n_dir = a.direct ? G * (a.L2 + a.L1) : 0;
  const int nmax = max(max(n, L), n_dir);
  const int mask = (1 << a.tp_shift) - 1;
  for (int t = threadIdx.x; t < nmax; t += blockDim.x) {
    cplx tv = make_double2(0.0, 0.0), hi = tv, lo = tv;
    int sp = 0, dst = 0;
    if (t < n) tv = P.tw_lo[t];
    if (t < L) sp = (int)P.pos[t];
    if (t < n_dir) {
      int e;
      if (t < n_stw) {                       // stw[g * L2 + r] = w_N^{o_g r}
        const int g = t / a.L2, r = t - g * a.L2;
        e = (g ? og1 : oq) * r;
        dst = t;
      } else {                               // wj[g * L1 + j] = w_L1^{j o_g}
        const int u = t - n_stw;
        const int g = u / a.L1, j = u - g * a.L1;
        e = ((j * (g ? og1 : oq)) % a.L1) * a.L2;
        dst = u;
      }
      hi = a.tp_hi[e >> a.tp_shift];
      lo = a.tp_lo[e & mask];
    }
    if (t < n) tlo[t] = tv;
    if (t < L) spos[t] = sp;
    if (t < n_dir) (t < n_stw ? stw : wj)[dst] = cmul(hi, lo);
  }
}

// blockIdx -> (column tile, o).  Direct mode: the L1 workgroups that share one column tile
// (= the same live-row slice of the kernel) are consecutive on ONE XCD (blocks go to the 8
// XCDs round-robin), so the slice is fetched from HBM once and re-read out of that L2.
__device__ __forceinline__ bool fused_tile_map(const ColFusedArgs& a, int ntiles, int* tile, int* o, int G = 1) {
  const int nouter = a.L1 / G;   // workgroups per column tile
  const int bx = blockIdx.x;
  if (!a.direct) {
    *tile = bx % ntiles;
    *o = bx / ntiles;
    return true;
  }
  const int xcd = bx & 7, q = bx >> 3;
  const int grp = q / nouter;
  *o = q - grp * nouter;
  *tile = grp * 8 + xcd;
  return *tile < ntiles;
}

template <bool GEN>
__global__ void k_col_fused(ColFusedArgs a) {
  const FftProg& P = a.prog;
  const int L = P.L;
  const int W = 1 << a.wsh;
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + ((size_t)L << a.wsh);
  cplx* thi = tlo + P.n_lo;
  int* spos = reinterpret_cast<int*>(tlo + tw_count(P));
  const int ntiles = (a.ncols + W - 1) >> a.wsh;
  int tile, o;
  if (!fused_tile_map(a, ntiles, &tile, &o)) return;
  const int c0 = tile << a.wsh;
  const int nthr = blockDim.x;
  const cplx* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  load_tw(tlo, thi, P);
  for (int r = threadIdx.x; r < L; r += nthr) spos[r] = (int)P.pos[r];
  const int tot = L << a.wsh;
  const int64_t base = (int64_t)o * a.L2;
  if (a.direct) {
    cplx* stw = reinterpret_cast<cplx*>(spos + ((L + 3) & ~3));
    cplx* wj = stw + L;
    kt_direct_tables(a, stw, wj, o);
    __syncthreads();
    kt_direct_fill<1>(a, data, stw, wj, c0, a.wsh, 0);
  } else
  for (int idx0 = threadIdx.x; idx0 < tot; idx0 += nthr * PS_UNROLL) {
    cplx v[PS_UNROLL];
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      const int row = idx >> a.wsh, col = c0 + (idx & (W - 1));
      v[u] = make_double2(0.0, 0.0);
      if (idx < tot && col < a.ncols && row_live(a.live, (int)(base + row), 0))
        v[u] = src[(base + row) * a.ld + col];
    }
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      if (idx < tot) data[idx] = v[u];
    }
  }
  __syncthreads();
  lds_fft<PS_FWD, GEN>(data, tlo, thi, P, PS_MODE_COL, W, a.wsh, 0);
  for (int idx0 = threadIdx.x; idx0 < tot; idx0 += nthr * PS_UNROLL) {
    cplx v[PS_UNROLL];
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      const int k = idx >> a.wsh, col = c0 + (idx & (W - 1));
      v[u] = make_double2(0.0, 0.0);
      if (idx < tot && col < a.ncols) v[u] = a.state[((int64_t)o + (int64_t)a.L1 * k) * a.ld + col];
    }
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      if (idx >= tot) continue;
      const int k = idx >> a.wsh, c = idx & (W - 1);
      const int l = (spos[k] << a.wsh) + c;
      const cplx x = cmul(v[u], data[l]);
      data[l] = x;
      if (a.store_prod && c0 + c < a.ncols)
        a.state[((int64_t)o + (int64_t)a.L1 * k) * a.ld + c0 + c] = x;
    }
  }
  __syncthreads();
  lds_fft<PS_INV, GEN>(data, tlo, thi, P, PS_MODE_COL, W, a.wsh, 0);
  for (int idx = threadIdx.x; idx < tot; idx += nthr) {
    const int row = idx >> a.wsh, col = c0 + (idx & (W - 1));
    if (col < a.ncols) a.dst[(base + row) * a.ld + col] = data[idx];
  }
}

// ND consecutive days in one pass (ND = 2, 4 or 8): the kernels of days d .. d+ND-1 sit side by
// side in one [L x ND*W] LDS tile, ONE forward FFT finishes all of them, the state is chained
// through them (A_{d+1} = A_d K_d, A_{d+2} = A_{d+1} K_{d+1}, ... -- the same products in the
// same order as ND single-day passes, so the results are bit-identical), only the LAST
// spectrum goes back to HBM, one inverse FFT, ND outputs.  HBM traffic per day drops from 4
// spectra to (2 + 2 ND) / ND.  Valid while no day in the group raises the boundary flag: the
// caller only groups days inside a speculation window (ps_chain_run).
template <bool GEN, int ND, int G>
__global__ void k_col_fused_multi(ColFusedArgs a) {
  constexpr int NDSH = ND == 1 ? 0 : (ND == 2 ? 1 : (ND == 4 ? 2 : 3));
  constexpr int GSH = G == 1 ? 0 : 1;
  static_assert(ND == 1 || ND == 2 || ND == 4 || ND == 8, "ND must be 1, 2, 4 or 8");
  static_assert(G == 1 || G == 2, "G must be 1 or 2");
  const FftProg& P = a.prog;
  const int L = P.L;
  // tile columns: (g * ND + day) * W + c
  const int W = 1 << a.wsh, wshd = a.wsh + NDSH, wshn = wshd + GSH, WN = W << (NDSH + GSH);
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + ((size_t)L << wshn);
  cplx* thi = tlo + P.n_lo;
  int* spos = reinterpret_cast<int*>(thi + P.n_hi + P.n_gen);
  const int ntiles = (a.ncols + W - 1) >> a.wsh;
  int tile, oq;
  if (!fused_tile_map(a, ntiles, &tile, &oq, G)) return;
  // outer indices of this workgroup: G == 1: oq; G == 2: the conjugate pair (oq, L1 - oq), with
  // (0, L1/2) as the pair oq == 0 (kt_direct_fill_pair)
  const int oA = oq, oB = G == 2 ? pair_o(oq, 1, a.L1) : oq;
  const int c0 = tile << a.wsh;
  const int nthr = blockDim.x;
  cplx* stw = reinterpret_cast<cplx*>(spos + ((L + 3) & ~3));
  cplx* wj = stw + G * L;
  unsigned* ent = reinterpret_cast<unsigned*>(wj + G * a.L1);   // [L2][ND]
  fused_prologue<G>(a, tlo, spos, stw, wj, oq, ent, ND);
  const int totg = L << (a.wsh + GSH), totn = L << wshn;
  if (a.direct) {
    __syncthreads();
    if (G == 2) kt_direct_fill_pair<ND>(a, data, stw, wj, c0, wshn, ent, oq != 0 && !a.no_conj);
    else kt_direct_fill<1>(a, data, stw, wj, c0, wshn, NDSH, nullptr);
  } else
  for (int idx0 = threadIdx.x; idx0 < totn; idx0 += nthr * PS_UNROLL) {
    cplx v[PS_UNROLL];
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      const int row = idx >> wshn, cc = idx & (WN - 1);
      const int g = cc >> wshd, day = (cc >> a.wsh) & (ND - 1), col = c0 + (cc & (W - 1));
      const int64_t grow = (int64_t)(g ? oB : oA) * a.L2 + row;
      v[u] = make_double2(0.0, 0.0);
      if (idx < totn && col < a.ncols && row_live(a.live, (int)grow, day))
        v[u] = a.src[day * a.src_bstride + grow * a.ld + col];
    }
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      if (idx < totn) data[idx] = v[u];
    }
  }
  __syncthreads();
  lds_fft<PS_FWD, GEN>(data, tlo, thi, P, PS_MODE_COL, WN, wshn, 0);
  // elements (k, g, c): state row o_g + L1 k
  for (int idx0 = threadIdx.x; idx0 < totg; idx0 += nthr * PS_UNROLL) {
    cplx v[PS_UNROLL];
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      const int k = idx >> (a.wsh + GSH), g = (idx >> a.wsh) & (G - 1), col = c0 + (idx & (W - 1));
      v[u] = make_double2(0.0, 0.0);
      if (idx < totg && col < a.ncols) v[u] = a.state[((int64_t)(g ? oB : oA) + (int64_t)a.L1 * k) * a.ld + col];
    }
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      if (idx >= totg) continue;
      const int k = idx >> (a.wsh + GSH), g = (idx >> a.wsh) & (G - 1), c = idx & (W - 1);
      const int l = (spos[k] << wshn) + (g << wshd) + c;
      cplx x = v[u];
#pragma unroll
      for (int day = 0; day < ND; ++day) {
        x = cmul(x, data[l + (day << a.wsh)]);
        data[l + (day << a.wsh)] = x;
      }
      if (a.store_prod && c0 + c < a.ncols)
        a.state[((int64_t)(g ? oB : oA) + (int64_t)a.L1 * k) * a.ld + c0 + c] = x;
    }
  }
  __syncthreads();
  lds_fft<PS_INV, GEN>(data, tlo, thi, P, PS_MODE_COL, WN, wshn, 0);
  for (int idx = threadIdx.x; idx < totn; idx += nthr) {
    const int row = idx >> wshn, cc = idx & (WN - 1);
    const int g = cc >> wshd, day = (cc >> a.wsh) & (ND - 1), col = c0 + (cc & (W - 1));
    if (col < a.ncols) a.dst[day * a.dst_dstride + ((int64_t)(g ? oB : oA) * a.L2 + row) * a.ld + col] = data[idx];
  }
}

// PS_MODE_FOLD variant: the state arrives like the kernel, one column sub-pass short of its
// spectrum (it is re-transformed from space every day).  Kernel tile and state tile sit side by
// side in one [L x 2W] LDS tile, ONE forward FFT finishes both, the product goes into the left
// half, the inverse sub-pass follows -- the state's last forward
// sub-pass never touches HBM.  `state` is read-only here, same row layout as `src`.
template <bool GEN>
__global__ void k_col_fused_dual(ColFusedArgs a) {
  const FftProg& P = a.prog;
  const int L = P.L;
  const int W = 1 << a.wsh, wsh2 = a.wsh + 1;
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + ((size_t)L << wsh2);
  cplx* thi = tlo + P.n_lo;
  const int ntiles = (a.ncols + W - 1) >> a.wsh;
  int tile, o;
  if (!fused_tile_map(a, ntiles, &tile, &o)) return;
  const int c0 = tile << a.wsh;
  const int nthr = blockDim.x;
  load_tw(tlo, thi, P);
  const int tot = L << a.wsh;
  const int64_t base = (int64_t)o * a.L2;
  for (int idx0 = threadIdx.x; idx0 < tot; idx0 += nthr * PS_UNROLL) {
    cplx v[PS_UNROLL], w[PS_UNROLL];
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      const int row = idx >> a.wsh, col = c0 + (idx & (W - 1));
      v[u] = make_double2(0.0, 0.0);
      w[u] = v[u];
      if (idx < tot && col < a.ncols) {
        if (row_live(a.live, (int)(base + row), 0)) v[u] = a.src[(base + row) * a.ld + col];
        if (row_live(a.live2, (int)(base + row), 0)) w[u] = a.state[(base + row) * a.ld + col];
      }
    }
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const int idx = idx0 + u * nthr;
      if (idx < tot) {
        const int row = idx >> a.wsh, c = idx & (W - 1);
        data[(row << wsh2) + c] = v[u];
        data[(row << wsh2) + W + c] = w[u];
      }
    }
  }
  __syncthreads();
  lds_fft<PS_FWD, GEN>(data, tlo, thi, P, PS_MODE_COL, 2 * W, wsh2, 0);
  // product of the two halves (same LDS row = same spectral index) into the left half, which
  // the inverse sub-pass then transforms at the tile's pitch of 2W
  for (int idx = threadIdx.x; idx < tot; idx += nthr) {
    const int e = ((idx >> a.wsh) << wsh2) + (idx & (W - 1));
    data[e] = cmul(data[e], data[e + W]);
  }
  __syncthreads();
  lds_fft<PS_INV, GEN>(data, tlo, thi, P, PS_MODE_COL, W, a.wsh, wsh2);
  for (int idx = threadIdx.x; idx < tot; idx += nthr) {
    const int row = idx >> a.wsh, c = idx & (W - 1), col = c0 + c;
    if (col < a.ncols) a.dst[(base + row) * a.ld + col] = data[(row << wsh2) + c];
  }
}

// ------------------------------------------------------------ inverse rows
// Half-spectrum rows (ra, rb) -> Z = A + i B (Hermitian-extended) -> inverse FFT ->
// a = Re z, b = Im z.  Fused epilogue: scale by 1/P^2, write the raw domain
// field (CalcSol.py:41), per-row threshold statistics (CalcSol.py:126-135) and
// the max over the pad region (CalcSol.py:36-37).
template <bool GEN, bool BIG>
__global__ void __launch_bounds__(BIG ? 512 : 1024) k_row_inv(RowInvArgs a) {
  const FftProg& P = a.prog;
  const int L = P.L;
  const int pitch = row_pitch(P);
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + (size_t)a.rp * pitch;
  cplx* thi = tlo + P.n_lo;
  double* red = reinterpret_cast<double*>(tlo + tw_count(P));  // 4 * (blockDim/64) doubles
  const cplx* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  const int pair0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * a.rp;   // cheap pad-only pairs first: no straggler round
  const int nthr = blockDim.x;
  load_tw(tlo, thi, P);
  // Workgroups whose rows all lie in the pad region (row >= N) only feed the boundary flag.
  // Parseval bounds their largest value: max|x| <= sqrt(2 P sum_k |Y_k|^2) / P^2 over the
  // stored half spectra of the rows.  If that bound is below the flag threshold the rows
  // cannot raise the flag and the transform is skipped (the reported pad maximum is then a
  // lower bound, exact whenever it matters, i.e. above 1e-8).
  const bool pad_only = 2 * pair0 >= a.N;
  double energy = 0.0;
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    const bool hasa = ra < a.P, hasb = rb < a.P;
    const cplx* pa = src + (int64_t)ra * a.ld;
    const cplx* pb = src + (int64_t)rb * a.ld;
    for (int k0 = threadIdx.x; k0 < L; k0 += nthr * PS_UNROLL) {
      cplx A[PS_UNROLL], B[PS_UNROLL];
      unsigned pp[PS_UNROLL];
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) {
        const int k = k0 + u * nthr;
        A[u] = make_double2(0.0, 0.0);
        B[u] = A[u];
        pp[u] = 0;
        if (k < L) {
          const int kk = k < a.H ? k : L - k;
          if (hasa) A[u] = pa[kk];
          if (hasb) B[u] = pb[kk];
          pp[u] = P.pos_phys[k];
        }
      }
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) {
        const int k = k0 + u * nthr;
        if (k < L) {
          const cplx z = k < a.H ? make_double2(A[u].x - B[u].y, A[u].y + B[u].x)
                                 : make_double2(A[u].x + B[u].y, B[u].x - A[u].y);
          data[b * pitch + pp[u]] = z;
          if (pad_only && k < a.H)
            energy += A[u].x * A[u].x + A[u].y * A[u].y + B[u].x * B[u].x + B[u].y * B[u].y;
        }
      }
    }
  }
  if (pad_only) {   // uniform across the workgroup
    for (int off = 32; off > 0; off >>= 1) energy += __shfl_down(energy, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = energy;
    __syncthreads();
    double e = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) e += red[w];
    if (sqrt(2.0 * (double)a.P * e) * a.scale < a.pad_floor) return;
  }
  __syncthreads();
  lds_fft<PS_INV, GEN, BIG>(data, tlo, thi, P, PS_MODE_ROW, a.rp, 0, pitch);
  double* rec = a.rec + (int64_t)blockIdx.y * a.rec_bstride;
  double* rowsum = a.rowsum + (int64_t)blockIdx.y * a.stat_bstride;
  long long* rowcnt = a.rowcnt + (int64_t)blockIdx.y * a.stat_bstride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nthr >> 6;
  double pmax = 0.0;
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    if (ra >= a.P) break;
    const cplx* d = data + b * pitch;
    double sa = 0.0, sb = 0.0;
    int ca = 0, cb = 0;
    for (int i = threadIdx.x; i < a.P; i += nthr) {
      const cplx z = d[row_phys(P, i)];
      const double va = z.x * a.scale, vb = z.y * a.scale;
      if (i < a.N) {
        if (ra < a.N) {
          rec[(int64_t)ra * a.N + i] = va;
          const double t = va * a.stat_scale;
          if (!(t < a.negval)) { sa += t; ++ca; }
        } else {
          pmax = fmax(pmax, va);
        }
        if (rb < a.N) {
          rec[(int64_t)rb * a.N + i] = vb;
          const double t = vb * a.stat_scale;
          if (!(t < a.negval)) { sb += t; ++cb; }
        } else if (rb < a.P) {
          pmax = fmax(pmax, vb);
        }
      } else {
        pmax = fmax(pmax, va);
        if (rb < a.P) pmax = fmax(pmax, vb);
      }
    }
    // deterministic block reduction (fixed shuffle tree, then waves in order)
    for (int off = 32; off > 0; off >>= 1) {
      sa += __shfl_down(sa, off);
      sb += __shfl_down(sb, off);
      ca += __shfl_down(ca, off);
      cb += __shfl_down(cb, off);
    }
    __syncthreads();
    if (lane == 0) {
      red[wave * 4 + 0] = sa;
      red[wave * 4 + 1] = sb;
      red[wave * 4 + 2] = (double)ca;
      red[wave * 4 + 3] = (double)cb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ta = 0, tb = 0, na = 0, nbb = 0;
      for (int w = 0; w < nw; ++w) {
        ta += red[w * 4 + 0];
        tb += red[w * 4 + 1];
        na += red[w * 4 + 2];
        nbb += red[w * 4 + 3];
      }
      if (ra < a.N) { rowsum[ra] = ta; rowcnt[ra] = (long long)na; }
      if (rb < a.N) { rowsum[rb] = tb; rowcnt[rb] = (long long)nbb; }
    }
  }
  // one atomic per workgroup at most, and only when it can raise the maximum: every
  // workgroup hitting the same word costs ~12 ns each, serialised
  for (int off = 32; off > 0; off >>= 1) pmax = fmax(pmax, __shfl_down(pmax, off));
  __syncthreads();
  if (lane == 0) red[wave] = pmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = 0.0;
    for (int w = 0; w < nw; ++w) m = fmax(m, red[w]);
    unsigned long long* pm = a.padmax + blockIdx.y;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    // only maxima that can matter for the flag (> 1e-8) are published: the read-check is a
    // global round trip at the end of every workgroup otherwise
    if (m > a.pad_floor && bits > __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(pm, bits);
  }
}
