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
#pragma once
#include "fft_core.h"

// ---------------------------------------------------------------- arguments
struct SrcMap {  // torus index i -> source index, or -1 (zero)
  int n1, off1, lo2, off2;
};
__device__ __forceinline__ int src_map(const SrcMap& m, int i) {
  if (i < m.n1) return i + m.off1;
  if (i >= m.lo2) return i - m.lo2 + m.off2;
  return -1;
}

struct RowFwdArgs {
  const double* src;
  int64_t src_bstride;
  int src_ld;
  SrcMap rmap, cmap;
  cplx* dst;
  int64_t dst_bstride;
  int H, ld, P;  // dst is [P][ld], H valid columns
  int rp;        // row pairs per block
  const int* pred;
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
  const int* pred;
  FftProg prog;
};

struct RowInvArgs {
  const cplx* src;
  int64_t src_bstride;
  int H, ld, P, N;
  int rp;
  double scale;  // 1 / Pfft^2
  double* rec;   // [N][N] raw real solution
  int64_t rec_bstride;
  double negval, stat_scale;  // stats on v * stat_scale >= negval
  double* rowsum;             // [N]
  long long* rowcnt;          // [N]
  unsigned long long* padmax; // bits of max(pad, 0)
  int64_t stat_bstride;       // per-batch stride of rowsum/rowcnt (padmax: 1)
  FftProg prog;
};

// -------------------------------------------------------------- LDS helpers
__device__ __forceinline__ void load_tw(cplx* tlo, cplx* thi, const FftProg& P) {
  for (int t = threadIdx.x; t < P.n_lo; t += blockDim.x) tlo[t] = P.tw_lo[t];
  for (int t = threadIdx.x; t < P.n_hi; t += blockDim.x) thi[t] = P.tw_hi[t];
}

template <int DIR, bool GEN>
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
    default:
      if (GEN) run_stage_generic<DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr);
      break;
  }
}

template <int DIR, bool GEN>
__device__ __forceinline__ void lds_fft(cplx* data, const cplx* tlo, const cplx* thi,
                                        const FftProg& P, int mode, int nb, int wsh, int bs) {
  if (DIR == PS_FWD) {
    for (int s = 0; s < P.ns; ++s) {
      run_stage_sel<DIR, GEN>(data, tlo, thi, P, s, mode, nb, wsh, bs);
      __syncthreads();
    }
  } else {
    for (int s = P.ns - 1; s >= 0; --s) {
      run_stage_sel<DIR, GEN>(data, tlo, thi, P, s, mode, nb, wsh, bs);
      __syncthreads();
    }
  }
}

extern __shared__ __attribute__((aligned(16))) unsigned char ps_lds_raw[];

// ------------------------------------------------------------ forward rows
// Two real rows (ra, rb) -> z = a + i b -> complex FFT -> A = (Z_k + conj Z_{L-k})/2,
// B = (Z_k - conj Z_{L-k})/(2i).  Rows outside the source map are zero.
template <bool GEN>
__global__ void k_row_fwd(RowFwdArgs a) {
  if (a.pred && *a.pred == 0) return;
  const FftProg& P = a.prog;
  const int L = P.L;
  const int pitch = row_pitch(P);
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + (size_t)a.rp * pitch;
  cplx* thi = tlo + P.n_lo;
  const double* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  cplx* dst = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
  const int pair0 = blockIdx.x * a.rp;
  // any non-zero source row in this block?
  bool any = false;
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    if (ra < a.P && src_map(a.rmap, ra) >= 0) any = true;
    if (rb < a.P && src_map(a.rmap, rb) >= 0) any = true;
  }
  if (!any) {
    const cplx z = make_double2(0.0, 0.0);
    for (int b = 0; b < a.rp; ++b) {
      const int ra = 2 * (pair0 + b), rb = ra + 1;
      for (int k = threadIdx.x; k < a.H; k += blockDim.x) {
        if (ra < a.P) dst[(int64_t)ra * a.ld + k] = z;
        if (rb < a.P) dst[(int64_t)rb * a.ld + k] = z;
      }
    }
    return;
  }
  load_tw(tlo, thi, P);
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    const int sa = ra < a.P ? src_map(a.rmap, ra) : -1;
    const int sb = rb < a.P ? src_map(a.rmap, rb) : -1;
    const double* pa = src + (int64_t)sa * a.src_ld;
    const double* pb = src + (int64_t)sb * a.src_ld;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      const int sc = src_map(a.cmap, i);
      double va = 0.0, vb = 0.0;
      if (sc >= 0) {
        if (sa >= 0) va = pa[sc];
        if (sb >= 0) vb = pb[sc];
      }
      data[b * pitch + row_phys(P, i)] = make_double2(va, vb);
    }
  }
  __syncthreads();
  lds_fft<PS_FWD, GEN>(data, tlo, thi, P, PS_MODE_ROW, a.rp, 0, pitch);
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    if (ra >= a.P) break;
    const cplx* d = data + b * pitch;
    for (int k = threadIdx.x; k < a.H; k += blockDim.x) {
      const cplx zk = d[P.pos_phys[k]];
      const cplx zm = d[P.pos_phys[k ? L - k : 0]];
      const cplx A = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
      const cplx B = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
      dst[(int64_t)ra * a.ld + k] = A;
      if (rb < a.P) dst[(int64_t)rb * a.ld + k] = B;
    }
  }
}

// ---------------------------------------------------------------- columns
template <int DIR, bool GEN>
__global__ void k_col(ColArgs a) {
  if (a.pred && *a.pred == 0) return;
  const FftProg& P = a.prog;
  const int L = P.L;
  const int W = 1 << a.wsh;
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + ((size_t)L << a.wsh);
  cplx* thi = tlo + P.n_lo;
  const int ntiles = (a.ncols + W - 1) >> a.wsh;
  const int tile = blockIdx.x % ntiles;
  const int o = blockIdx.x / ntiles;
  const int c0 = tile << a.wsh;
  const cplx* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  const cplx* src2 = a.src2 ? a.src2 + (int64_t)blockIdx.y * a.src2_bstride : nullptr;
  cplx* prod = a.prod_dst ? a.prod_dst + (int64_t)blockIdx.y * a.prod_bstride : nullptr;
  cplx* dst = a.dst + (int64_t)blockIdx.y * a.dst_bstride;
  load_tw(tlo, thi, P);
  const int tot = L << a.wsh;
  const int in_base = o * a.in_base_mul, out_base = o * a.out_base_mul;
  for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
    const int row = idx >> a.wsh, c = idx & (W - 1);
    const int col = c0 + c;
    cplx v = make_double2(0.0, 0.0);
    if (col < a.ncols) {
      const int64_t g = (int64_t)(in_base + row * a.in_stride) * a.ld + col;
      v = src[g];
      if (src2) {
        v = cmul(v, src2[g]);
        if (prod) prod[g] = v;
      }
      if (a.tw_mode == 2 && o != 0 && row != 0)
        v = cmulc(v, tw_lookup(a.tp_lo, a.tp_hi, a.tp_shift, o * row));
    }
    const int lrow = DIR == PS_INV ? (int)P.pos[row] : row;
    data[(lrow << a.wsh) + c] = v;
  }
  __syncthreads();
  lds_fft<DIR, GEN>(data, tlo, thi, P, PS_MODE_COL, W, a.wsh, 0);
  for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
    const int row = idx >> a.wsh, c = idx & (W - 1);
    const int col = c0 + c;
    if (col >= a.ncols) continue;
    const int lrow = DIR == PS_FWD ? (int)P.pos[row] : row;
    cplx v = data[(lrow << a.wsh) + c];
    if (a.tw_mode == 1 && o != 0 && row != 0)
      v = cmul(v, tw_lookup(a.tp_lo, a.tp_hi, a.tp_shift, o * row));
    dst[(int64_t)(out_base + row * a.out_stride) * a.ld + col] = v;
  }
}

// ------------------------------------------------------------ inverse rows
// Half-spectrum rows (ra, rb) -> Z = A + i B (Hermitian-extended) -> inverse FFT ->
// a = Re z, b = Im z.  Fused epilogue: scale by 1/P^2, write the raw domain
// field (CalcSol.py:41), per-row threshold statistics (CalcSol.py:126-135) and
// the max over the pad region (CalcSol.py:36-37).
template <bool GEN>
__global__ void k_row_inv(RowInvArgs a) {
  const FftProg& P = a.prog;
  const int L = P.L;
  const int pitch = row_pitch(P);
  cplx* data = reinterpret_cast<cplx*>(ps_lds_raw);
  cplx* tlo = data + (size_t)a.rp * pitch;
  cplx* thi = tlo + P.n_lo;
  double* red = reinterpret_cast<double*>(thi + P.n_hi);  // 3 * (blockDim/64) doubles
  const cplx* src = a.src + (int64_t)blockIdx.y * a.src_bstride;
  const int pair0 = blockIdx.x * a.rp;
  load_tw(tlo, thi, P);
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    const bool hasa = ra < a.P, hasb = rb < a.P;
    const cplx* pa = src + (int64_t)ra * a.ld;
    const cplx* pb = src + (int64_t)rb * a.ld;
    for (int k = threadIdx.x; k < L; k += blockDim.x) {
      const int kk = k < a.H ? k : L - k;
      cplx A = make_double2(0.0, 0.0), B = A;
      if (hasa) A = pa[kk];
      if (hasb) B = pb[kk];
      cplx z = k < a.H ? make_double2(A.x - B.y, A.y + B.x) : make_double2(A.x + B.y, B.x - A.y);
      data[b * pitch + P.pos_phys[k]] = z;
    }
  }
  __syncthreads();
  lds_fft<PS_INV, GEN>(data, tlo, thi, P, PS_MODE_ROW, a.rp, 0, pitch);
  double* rec = a.rec + (int64_t)blockIdx.y * a.rec_bstride;
  double* rowsum = a.rowsum + (int64_t)blockIdx.y * a.stat_bstride;
  long long* rowcnt = a.rowcnt + (int64_t)blockIdx.y * a.stat_bstride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double pmax = 0.0;
  for (int b = 0; b < a.rp; ++b) {
    const int ra = 2 * (pair0 + b), rb = ra + 1;
    if (ra >= a.P) break;
    const cplx* d = data + b * pitch;
    double sa = 0.0, sb = 0.0;
    int ca = 0, cb = 0;
    for (int i = threadIdx.x; i < a.P; i += blockDim.x) {
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
  for (int off = 32; off > 0; off >>= 1) pmax = fmax(pmax, __shfl_down(pmax, off));
  if (lane == 0 && pmax > 0.0)
    atomicMax(a.padmax + blockIdx.y, (unsigned long long)__double_as_longlong(pmax));
}
