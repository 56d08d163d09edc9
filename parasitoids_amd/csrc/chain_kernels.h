// Non-FFT kernels of the day chain: COO scatter (CalcSol.py:23,:61-64), per-day
// statistics/flag (CalcSol.py:36-40,:126-135), ordered stream compaction back to
// COO, and the release-day weighted sum of CalcSol.get_populations (:322-323).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// dst[(row+roff)*ld + col+coff] += val   (duplicates sum, as coo.toarray() does)
static __global__ void k_scatter_coo(const int* __restrict__ row, const int* __restrict__ col,
                              const double* __restrict__ val, int64_t nnz, double* dst, int ld,
                              int roff, int coff) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nnz;
       i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&dst[(int64_t)(row[i] + roff) * ld + col[i] + coff], val[i]);
}

// zero rows [r0, r1] of each of gridDim.y consecutive K x K blocks (the band of staging rows
// some kernel of the chunk writes: the row pass reads nothing else)
static __global__ void k_zero_band(double* dst, int K, int r0, int r1) {
  double* base = dst + (int64_t)blockIdx.y * K * K + (int64_t)r0 * K;
  const int64_t n = (int64_t)(r1 - r0 + 1) * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    base[i] = 0.0;
}

// batched form: day d = blockIdx.y scatters its triplets [koff[d], koff[d+1]) into the d-th
// K x K staging block, centred (offset M - kshape[d]/2): one launch per chunk of kernels
static __global__ void k_scatter_coo_batch(const int* __restrict__ row, const int* __restrict__ col,
                                           const double* __restrict__ val, const long long* __restrict__ koff,
                                           const int* __restrict__ kshape, int first, double* dst, int K) {
  const int d = first + blockIdx.y;
  const long long lo = koff[d], hi = koff[d + 1];
  const int off = K / 2 - kshape[d] / 2;
  double* out = dst + (int64_t)blockIdx.y * K * K;
  for (long long i = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < hi;
       i += (long long)gridDim.x * blockDim.x)
    atomicAdd(&out[(int64_t)(row[i] + off) * K + col[i] + off], val[i]);
}

struct DayStats {
  long long nnz;      // entries with v*scale >= negval
  double sum;         // their sum
  double delta;       // (1 - sum)/nnz  (prob model renormalisation), else 0
  double padmax;      // max over the pad region (>= 0)
  int flag;           // padmax > 1e-8
  int pad_;
};

// one block per day: reduce the per-row statistics in a fixed order
static __global__ void k_day_finalize(const double* rowsum_, const long long* rowcnt_,
                               const unsigned long long* padmax_, int N, int renorm,
                               DayStats* out_) {
  const double* rowsum = rowsum_ + (int64_t)blockIdx.x * N;
  const long long* rowcnt = rowcnt_ + (int64_t)blockIdx.x * N;
  const unsigned long long* padmax = padmax_ + blockIdx.x;
  DayStats* out = out_ + blockIdx.x;
  __shared__ double ssum[256];
  __shared__ long long scnt[256];
  double s = 0.0;
  long long c = 0;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    s += rowsum[r];
    c += rowcnt[r];
  }
  ssum[threadIdx.x] = s;
  scnt[threadIdx.x] = c;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
      scnt[threadIdx.x] += scnt[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    DayStats d;
    d.nnz = scnt[0];
    d.sum = ssum[0];
    d.delta = (renorm && scnt[0] > 0) ? (1.0 - ssum[0]) / (double)scnt[0] : 0.0;
    d.padmax = __longlong_as_double((long long)*padmax);
    d.flag = d.padmax > 1e-8 ? 1 : 0;
    d.pad_ = 0;
    *out = d;
  }
}

// per-row statistics of a dense N x N field (used when a field was not produced
// by the fused inverse-row epilogue, e.g. weighted sums and the first day)
static __global__ void k_row_stats(const double* __restrict__ rec, int N, double scale, double negval,
                            double* rowsum, long long* rowcnt) {
  __shared__ double ssum[256];
  __shared__ int scnt[256];
  const int r = blockIdx.x;
  double s = 0.0;
  int c = 0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const double t = rec[(int64_t)r * N + i] * scale;
    if (t != 0.0 && !(t < negval)) { s += t; ++c; }
  }
  ssum[threadIdx.x] = s;
  scnt[threadIdx.x] = c;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
      scnt[threadIdx.x] += scnt[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { rowsum[r] = ssum[0]; rowcnt[r] = scnt[0]; }
}

// PS_MODE_FOLD: linear convolution result on the M-torus (index y in [-m, P+m), negative y at
// M + y) -> circular result on the reference's P-torus: c[x] = lin[x] + lin[x+P] (x < m) +
// lin[x-P] (x >= P-m) per dimension.  One block per torus row.  Writes the P x P torus field
// (next day's state), the raw N x N record, the row statistics of r_small_vals
// (CalcSol.py:126-135) and the maximum over the pad region (CalcSol.py:36-37).
static __global__ void k_fold(const double* __restrict__ lin, int M, int P, int N, int m, double* torus,
                              double* rec, double negval, double stat_scale, double* rowsum,
                              long long* rowcnt, unsigned long long* padmax) {
  const int r = blockIdx.x;
  int yr[3], nr = 0;
  yr[nr++] = r;
  if (r < m) yr[nr++] = r + P;
  if (r >= P - m) yr[nr++] = r - P + M;
  __shared__ double ssum[256];
  __shared__ long long scnt[256];
  __shared__ double smax[256];
  double s = 0.0, pm = 0.0;
  long long cnt = 0;
  for (int c = threadIdx.x; c < P; c += blockDim.x) {
    int yc[3], nc = 0;
    yc[nc++] = c;
    if (c < m) yc[nc++] = c + P;
    if (c >= P - m) yc[nc++] = c - P + M;
    double v = 0.0;
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) v += lin[(int64_t)yr[i] * M + yc[j]];
    torus[(int64_t)r * P + c] = v;
    if (r < N && c < N) {
      rec[(int64_t)r * N + c] = v;
      const double t = v * stat_scale;
      if (!(t < negval)) { s += t; ++cnt; }
    } else {
      pm = fmax(pm, v);
    }
  }
  ssum[threadIdx.x] = s;
  scnt[threadIdx.x] = cnt;
  smax[threadIdx.x] = pm;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
      scnt[threadIdx.x] += scnt[threadIdx.x + off];
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (r < N) { rowsum[r] = ssum[0]; rowcnt[r] = scnt[0]; }
    const double mx = smax[0];
    const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
    if (mx > 0.0 && bits > __hip_atomic_load(padmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(padmax, bits);
  }
}

// PS_MODE_FOLD: truncate the torus field to the domain when the day raised the flag
// (CalcSol.py:200-201: the next transform starts from coo(A[:N,:N]))
static __global__ void k_truncate_if_flag(double* torus, int P, int N, const unsigned long long* padmax) {
  if (!(__longlong_as_double((long long)*padmax) > 1e-8)) return;
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < P; c += blockDim.x)
    if (r >= N || c >= N) torus[(int64_t)r * P + c] = 0.0;
}

// exclusive scan of per-row counts (one block per record: blockIdx.x-th array of N counts)
static __global__ void k_scan_rows(const long long* rowcnt_, int N, long long* rowoff_) {
  const long long* rowcnt = rowcnt_ + (int64_t)blockIdx.x * N;
  long long* rowoff = rowoff_ + (int64_t)blockIdx.x * N;
  __shared__ long long part[1024];
  const int T = blockDim.x;
  const int per = (N + T - 1) / T;
  const int lo = threadIdx.x * per, hi = min(N, lo + per);
  long long s = 0;
  {
    int r = lo;
    for (; r + 8 <= hi; r += 8) {   // eight independent loads per step
      long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = rowcnt[r + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < hi; ++r) s += rowcnt[r];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of the per-thread partial sums (integers: any order is exact)
  for (int off = 1; off < T; off <<= 1) {
    const long long v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  long long acc = part[threadIdx.x] - s;
  int r = lo;
  for (; r + 8 <= hi; r += 8) {
    long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = rowcnt[r + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) { rowoff[r + u] = acc; acc += v[u]; }
  }
  for (; r < hi; ++r) { rowoff[r] = acc; acc += rowcnt[r]; }
}

// ordered compaction: one wave per row, row-major COO order like coo_matrix(dense)
static __global__ void k_compact_rows(const double* __restrict__ rec, int N, double scale, double negval,
                               double delta, double post_scale, const long long* rowoff,
                               int* orow, int* ocol, double* oval) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= N) return;
  long long base = rowoff[wave];
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int c = c0 + lane;
    double t = 0.0;
    bool keep = false;
    if (c < N) {
      t = rec[(int64_t)wave * N + c] * scale;
      keep = (t != 0.0) && !(t < negval);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int o = __popcll(m & ((1ull << lane) - 1ull));
      orow[base + o] = wave;
      ocol[base + o] = c;
      oval[base + o] = (t + delta) * post_scale;
    }
    base += __popcll(m);
  }
}

// out = sum_d w[d] * rec_d   (release-day weighted population, CalcSol.py:322)
static __global__ void k_weighted_sum(const double* const* recs, const double* w, int nrec, int64_t n,
                               double* out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int d = 0; d < nrec; ++d) acc += recs[d][i] * w[d];
    out[i] = acc;
  }
}

// Hermitian expansion of the half spectrum to a full P x P complex array
// (only for the function-level CalcSol.fft2/fftconv2 mirrors)
static __global__ void k_expand_spectrum(const double2* __restrict__ half, int P, int H, int ld,
                                  double2* full) {
  const int64_t tot = (int64_t)P * P;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < tot;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx / P), c = (int)(idx % P);
    double2 v;
    if (c < H) {
      v = half[(int64_t)r * ld + c];
    } else {
      const int rr = r ? P - r : 0, cc = P - c;
      v = half[(int64_t)rr * ld + cc];
      v.y = -v.y;
    }
    full[idx] = v;
  }
}

static __global__ void k_take_half_spectrum(const double2* __restrict__ full, int P, int H, int ld,
                                     double2* half) {
  const int64_t tot = (int64_t)P * H;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < tot;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx / H), c = (int)(idx % H);
    half[(int64_t)r * ld + c] = full[(int64_t)r * P + c];
  }
}

// out[i] = rec[rows[i]][cols[i]] * scale, zeroed below negval (thresholded-solution lookup)
static __global__ void k_gather_points_multi(const double* const* __restrict__ recs, int N, const int* rows,
                                             const int* cols, int64_t n, double scale, double negval, double* out) {
  const double* rec = recs[blockIdx.y];
  double* o = out + (int64_t)blockIdx.y * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double t = rec[(int64_t)rows[i] * N + cols[i]] * scale;
    o[i] = (t < negval) ? 0.0 : t;
  }
}

static __global__ void k_gather_points(const double* __restrict__ rec, int N, const int* rows,
                                const int* cols, int64_t n, double scale, double negval, double* out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double t = rec[(int64_t)rows[i] * N + cols[i]] * scale;
    out[i] = (t < negval) ? 0.0 : t;
  }
}

// two ints by value into device memory (a live-row range the host knows): no staging buffer, no sync
static __global__ void k_set_int2(int* dst, int a, int b) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { dst[0] = a; dst[1] = b; }
}

