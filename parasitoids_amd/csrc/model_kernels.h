// HIP kernels for the per-day probability-mass kernel construction
// (ParasitoidModel.py:231-613).  fp64 VALU / transcendental bound, not HBM bound:
// the work is ~(2H+2)^2 bivariate-normal corner evaluations per period, 1440
// periods per day.
//
// Layout: one dense N x N fp64 pmf per day in HBM, cut into 16 x 16 tiles.  Per tile the
// ORDERED list of periods whose stamp window touches it is built first (k_tile_count ->
// scan -> k_tile_fill: only real (tile, period) pairs are stored).  k_pair_masses gives
// every pair its own wave: the Phi factors of the needed column/row edges, the BVU corner
// values of the part of the tile under the window, the 256 cell masses times hprob[t]
// (explicit __dmul_rn) -> a 2 KB record.  k_tile_accumulate adds a tile's records with
// __dadd_rn in ascending period order (ParasitoidModel.py:539-540 -- the reference's
// unfused accumulation order), so results are bit-identical to a sequential loop and
// reproducible run to run: no atomics, no chain of barriers through a tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_TS 16
#define PM_MAX_NODES 10

// Genz BVU quadrature data, precomputed on the host for the day's correlation
struct BvuRule {
  double r;                  // correlation
  int lg;                    // number of node pairs (3, 6, 10)
  int high;                  // |r| >= 0.925 branch
  double w[PM_MAX_NODES];    // Gauss-Legendre weights
  double x[PM_MAX_NODES];    // abscissae
  double sn1[PM_MAX_NODES];  // sin(asr (x+1)/2)
  double sn2[PM_MAX_NODES];  // sin(asr (-x+1)/2)
  double iv1[PM_MAX_NODES];  // 1 / (1 - sn1^2)
  double iv2[PM_MAX_NODES];  // 1 / (1 - sn2^2)
  double asr;                // asin(r)
};

struct ModelParams {
  double hp[7];     // lam, aw, bw, a1, b1, a2, b2
  double sdx, sdy;  // flight diffusion std devs (sqrt of Dmat diagonal)
  BvuRule rule;     // flight diffusion
  double lsdx, lsdy;
  BvuRule lrule;    // out-of-flow diffusion
  double mu_r, rad_dist, cell;
  int n_periods, rad_res, N, T, test_run, ndays_wind;
};

struct PeriodInfo {  // one per (day, period)
  double mux, muy;   // residual mean, metres (ParasitoidModel.py:485)
  double hprob;
  int rc, cc, H;     // window centre (row, col) and half width
  int skip;          // 1: t < start_indx
};

struct DayInfo {
  double loss, pmfsum, total, delta, pmfmin, ksum;
  long long nnz;
  int rad, warned, status, start_indx, Hl, day_idx;
  int r0, r1, c0, c1;  // bounding box of all windows, clipped to the domain
};

__device__ __forceinline__ double pm_phi(double z) {  // standard normal cdf
  return 0.5 * erfc(-z * 0.70710678118654752440);
}

// P(X > h, Y > k), restating Genz's BVU (MVNDST) -- see oracle/model.py:bvu
__device__ __forceinline__ double pm_bvu(const BvuRule& R, double h, double k) {
  const double TWOPI = 6.283185307179586;
  double hk = h * k;
  double bvn = 0.0;
  if (!R.high) {
    const double hs = (h * h + k * k) / 2;
    for (int i = 0; i < R.lg; ++i) {
      // (sn hk - hs)/(1 - sn^2) with the reciprocal precomputed on the host (<= 1 ulp
      // difference in the exponent's argument)
      bvn = bvn + R.w[i] * exp((R.sn1[i] * hk - hs) * R.iv1[i]);
      bvn = bvn + R.w[i] * exp((R.sn2[i] * hk - hs) * R.iv2[i]);
    }
    return bvn * R.asr / (2 * TWOPI) + pm_phi(-h) * pm_phi(-k);
  }
  const double r = R.r;
  if (r < 0) {
    k = -k;
    hk = -hk;
  }
  if (fabs(r) < 1) {
    const double as = (1 - r) * (1 + r);
    double a = sqrt(as);
    const double bs = (h - k) * (h - k);
    const double c = (4 - hk) / 8;
    const double d = (12 - hk) / 16;
    bvn = a * exp(-(bs / as + hk) / 2) * (1 - c * (bs - as) * (1 - d * bs / 5) / 3 + c * d * as * as / 5);
    if (hk > -160) {
      const double b = sqrt(bs);
      bvn = bvn - exp(-hk / 2) * sqrt(TWOPI) * pm_phi(-b / a) * b * (1 - c * bs * (1 - d * bs / 5) / 3);
    }
    a = a / 2;
    for (int i = 0; i < R.lg; ++i) {
      double xs = (a + a * R.x[i]) * (a + a * R.x[i]);
      double rs = sqrt(1 - xs);
      bvn = bvn + a * R.w[i] * (exp(-bs / (2 * xs) - hk / (1 + rs)) / rs -
                                exp(-(bs / xs + hk) / 2) * (1 + c * xs * (1 + d * xs)));
      xs = as * (-R.x[i] + 1) * (-R.x[i] + 1) / 4;
      rs = sqrt(1 - xs);
      bvn = bvn + a * R.w[i] * exp(-(bs / xs + hk) / 2) *
                      (exp(-hk * (1 - rs) / (2 * (1 + rs))) / rs - (1 + c * xs * (1 + d * xs)));
    }
    bvn = -bvn / TWOPI;
  }
  if (r > 0) bvn = bvn + pm_phi(-fmax(h, k));
  if (r < 0) bvn = -bvn + fmax(0.0, pm_phi(-h) - pm_phi(-k));
  return bvn;
}

// pm_bvu with Phi(-h), Phi(-k) supplied (they are shared by a whole row / column of the
// corner grid); identical arithmetic to pm_bvu
__device__ __forceinline__ double pm_bvu_phi(const BvuRule& R, double h, double k, double ph, double pk) {
  if (R.high) return pm_bvu(R, h, k);
  const double TWOPI = 6.283185307179586;
  const double hk = h * k;
  const double hs = (h * h + k * k) / 2;
  double bvn = 0.0;
  for (int i = 0; i < R.lg; ++i) {
    bvn = bvn + R.w[i] * exp((R.sn1[i] * hk - hs) * R.iv1[i]);
    bvn = bvn + R.w[i] * exp((R.sn2[i] * hk - hs) * R.iv2[i]);
  }
  return bvn * R.asr / (2 * TWOPI) + ph * pk;
}

// the |rho| < 0.925 branch of pm_bvu_phi on its own (identical arithmetic)
__device__ __forceinline__ double pm_bvu_low_phi(const BvuRule& R, double h, double k, double ph, double pk) {
  const double TWOPI = 6.283185307179586;
  const double hk = h * k;
  const double hs = (h * h + k * k) / 2;
  double bvn = 0.0;
  for (int i = 0; i < R.lg; ++i) {
    bvn = bvn + R.w[i] * exp((R.sn1[i] * hk - hs) * R.iv1[i]);
    bvn = bvn + R.w[i] * exp((R.sn2[i] * hk - hs) * R.iv2[i]);
  }
  return bvn * R.asr / (2 * TWOPI) + ph * pk;
}

// ... with the number of Gauss-Legendre node pairs a compile-time constant, and the 2 LG exponentials of
// a corner evaluated SIDE BY SIDE.  The run-time-length loop above re-loads the rule's constants from
// the kernel arguments every iteration (a scalar wait each time) and leaves every wave on one
// 17-operation dependent chain per exponential: an in-order wave sits out each link, and four waves
// per SIMD do not cover it (VALU busy 42 %).  Here the device library's exp() -- reduction by
// ln 2 in two parts, degree-11 polynomial, ldexp, the two range clamps; constants and operation order
// read off the compiler's own expansion -- is written out stage by stage over all 2 LG arguments, so
// consecutive instructions of a wave are independent.  Same operations on the same operands, same
// order of the weighted additions: bit-identical to pm_bvu_low_phi
// (tests/test_model_gpu.py::test_unrolled_pair_masses_are_bit_identical).
template <int M>
__device__ __forceinline__ void pm_exp_many(double (&x)[M]) {
  const double LOG2E = __longlong_as_double(0x3ff71547652b82feLL);
  const double NLN2H = __longlong_as_double((long long)0xbfe62e42fefa39efULL);
  const double NLN2L = __longlong_as_double((long long)0xbc7abc9e3b39803fULL);
  const double C[10] = {__longlong_as_double(0x3e5ade156a5dcb37LL), __longlong_as_double(0x3e928af3fca7ab0cLL),
                        __longlong_as_double(0x3ec71dee623fde64LL), __longlong_as_double(0x3efa01997c89e6b0LL),
                        __longlong_as_double(0x3f2a01a014761f6eLL), __longlong_as_double(0x3f56c16c1852b7b0LL),
                        __longlong_as_double(0x3f81111111122322LL), __longlong_as_double(0x3fa55555555502a1LL),
                        __longlong_as_double(0x3fc5555555555511LL), __longlong_as_double(0x3fe000000000000bLL)};
  double n[M], r[M], p[M];
  // the scheduling barriers keep the stages apart: left alone, the machine scheduler strings the M chains
  // back together one after the other to save registers (seen in the ISA), which is the dependent-chain
  // stall this function exists to remove
#define PM_STAGE(stmt)                \
  _Pragma("unroll") for (int i = 0; i < M; ++i) { stmt; } \
  __builtin_amdgcn_sched_barrier(0)
  PM_STAGE(n[i] = rint(x[i] * LOG2E));
  PM_STAGE(r[i] = __builtin_fma(NLN2H, n[i], x[i]));
  PM_STAGE(r[i] = __builtin_fma(NLN2L, n[i], r[i]));
  PM_STAGE(p[i] = __builtin_fma(C[0], r[i], C[1]));
#pragma unroll
  for (int c = 2; c < 10; ++c) {
    PM_STAGE(p[i] = __builtin_fma(r[i], p[i], C[c]));
  }
  PM_STAGE(p[i] = __builtin_fma(r[i], p[i], 1.0));
  PM_STAGE(p[i] = __builtin_fma(r[i], p[i], 1.0));
#undef PM_STAGE
#pragma unroll
  for (int i = 0; i < M; ++i) {
    // (the library's other clamp, x > 1024 -> inf, cannot fire: the BVU exponents are <= 0 up to round-off)
    const double v = ldexp(p[i], (int)n[i]);
    x[i] = x[i] < -1075.0 ? 0.0 : v;
  }
}

template <int LG>
__device__ __forceinline__ double pm_bvu_low_phi_t(const BvuRule& R, double h, double k, double ph, double pk) {
  const double TWOPI = 6.283185307179586;
  const double hk = h * k;
  const double hs = (h * h + k * k) / 2;
  double e[2 * LG];
#pragma unroll
  for (int i = 0; i < LG; ++i) {
    e[2 * i] = (R.sn1[i] * hk - hs) * R.iv1[i];
    e[2 * i + 1] = (R.sn2[i] * hk - hs) * R.iv2[i];
  }
  if (LG <= 3) {
    pm_exp_many<2 * LG>(e);
  } else {   // six chains at a time: twelve side by side do not fit the registers of three waves per SIMD
    static_assert(LG <= 3 || LG % 3 == 0, "groups of three node pairs");
#pragma unroll
    for (int g0 = 0; g0 < 2 * LG; g0 += 6) {
      double t[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i] = e[g0 + i];
      pm_exp_many<6>(t);
#pragma unroll
      for (int i = 0; i < 6; ++i) e[g0 + i] = t[i];
    }
  }
  double bvn = 0.0;
#pragma unroll
  for (int i = 0; i < LG; ++i) {
    bvn = bvn + R.w[i] * e[2 * i];
    bvn = bvn + R.w[i] * e[2 * i + 1];
  }
  return bvn * R.asr / (2 * TWOPI) + ph * pk;
}

// rectangle probability of N(mu, S) on [xl,xu] x [yl,yu] as mvnun computes it
__device__ __forceinline__ double pm_rect(const BvuRule& R, double sdx, double sdy, double mux,
                                          double muy, double xl, double xu, double yl, double yu) {
  const double l0 = (xl - mux) / sdx, u0 = (xu - mux) / sdx;
  const double l1 = (yl - muy) / sdy, u1 = (yu - muy) / sdy;
  return pm_bvu(R, l0, l1) - pm_bvu(R, u0, l1) - pm_bvu(R, l0, u1) + pm_bvu(R, u0, u1);
}

// support half width of get_mvn_cdf_values (ParasitoidModel.py:329,:348): smallest h
// with 1 - P(square of half width (h + 1/2) cell) < 1e-3.  One wave, lanes try
// h = base + lane.  The square probability equals the reference's running sum of
// cell masses up to round-off ("support rule").
__device__ __forceinline__ int pm_support(const BvuRule& R, double sdx, double sdy, double mux,
                                          double muy, double cell, int hmax) {
  const int lane = threadIdx.x & 63;
  for (int base = 0; base <= hmax; base += 64) {
    const int h = base + lane;
    const double e = (h + 0.5) * cell;
    const double p = pm_rect(R, sdx, sdy, mux, muy, -e, e, -e, e);
    const bool ok = (h <= hmax) && (1 - p < 0.001);
    const unsigned long long m = __ballot(ok);
    if (m) return base + __ffsll((long long)m) - 1;
  }
  return hmax;
}

// The same search for FOUR windows per wave: lane group g = lane / 16 looks for its own window's h
// among base + (lane & 15), 16 candidates a step (a typical support is 15-30 cells: 64 candidates per
// window and step mostly evaluated squares nobody needed).  Same result as pm_support: the smallest h
// that passes, hmax if none does.  Groups with `active` false take no part and get 0.
__device__ __forceinline__ int pm_support16(const BvuRule& R, double sdx, double sdy, double mux,
                                            double muy, double cell, int hmax, bool active) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  int found = active ? -1 : 0;
  for (int base = 0; base <= hmax; base += 16) {
    const int h = base + sub;
    bool ok = false;
    if (found < 0) {
      const double e = (h + 0.5) * cell;
      const double p = pm_rect(R, sdx, sdy, mux, muy, -e, e, -e, e);
      ok = (h <= hmax) && (1 - p < 0.001);
    }
    const unsigned long long m = __ballot(ok);
    const unsigned gm = (unsigned)((m >> (16 * grp)) & 0xffffull);
    if (found < 0 && gm) found = base + __ffs((int)gm) - 1;
    if (__ballot(found < 0) == 0ull) break;
  }
  return found < 0 ? hmax : found;
}

// ---------------------------------------------------------------- h_flight_prob
// ParasitoidModel.py:282-309 with f_time_prob :243-267 and g_wind_prob :231-240.
// One block per day; the two cumulative sums run sequentially like np.cumsum.
__global__ void k_hprob(const double* __restrict__ wind, ModelParams mp, const int* day_idx,
                        double* hprob /*[nd][T]*/, double* scratch /*unused*/) {
  extern __shared__ double hp_lds[];  // f[n], g[n], c2[n]
  const int d = blockIdx.x;
  const int n = mp.T;
  const double* w = wind + (int64_t)day_idx[d] * n * 3;
  double* f = hp_lds;
  double* g = f + n;
  double* c2 = g + n;
  double* h = hprob + (int64_t)d * n;
  const double lam = mp.hp[0], aw = mp.hp[1], bw = mp.hp[2], a1 = mp.hp[3], b1 = mp.hp[4],
               a2 = mp.hp[5], b2 = mp.hp[6];
  const double stop = 24 - 24. / n;
  const double step = n > 1 ? stop / (n - 1) : 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double t = (i == n - 1 && n > 1) ? stop : i * step;
    const double lik = fmax(1.0 / (1. + exp(-b1 * (t - a1))) - 1.0 / (1. + exp(-b2 * (t - a2))), 0.0);
    f[i] = lik;
    g[i] = 1.0 / (1. + exp(bw * (w[i * 3 + 2] - aw)));
  }
  __syncthreads();
  __shared__ double s_max, s_sum;
  __shared__ double s_red[256];
  if (threadIdx.x == 0) {
    // sequential sum (numpy's pairwise .sum() differs by round-off only); eight LDS reads in
    // flight per step, the additions in index order
    double s = 0.0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = f[i + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < n; ++i) s += f[i];
    s_sum = s;
  }
  __syncthreads();
  {
    // normalisation, its maximum and the integrand of the second cumulative sum are
    // independent per sample: all threads
    const double s = s_sum;
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double fi = f[i] / s;
      f[i] = fi;
      mx = fmax(mx, fi);
      c2[i] = fi - fi * g[i];
    }
    s_red[threadIdx.x] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double mx = 0.0;
    for (int t = 0; t < (int)blockDim.x; ++t) mx = fmax(mx, s_red[t]);
    s_max = mx;
    // the two running sums, sequential like np.cumsum
    double c1 = 0.0, cc = 0.0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
      double vf[8], vd[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { vf[u] = f[i + u]; vd[u] = c2[i + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c1 += vf[u];
        cc += (1 - c1) * vd[u];
        c2[i + u] = cc;
      }
    }
    for (; i < n; ++i) {
      c1 += f[i];
      cc += (1 - c1) * c2[i];
      c2[i] = cc;
    }
  }
  __syncthreads();
  const double mx = s_max;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double fg = f[i] * g[i];
    const double tv = (double)(i + 1);
    const double integral_avg = fg / tv / mx * c2[i];
    h[i] = lam * (fg + integral_avg);
  }
}

// ---------------------------------------------------------------- per period
// advection, residual mean, window centre, support (ParasitoidModel.py:439-499).
// Sixteen lanes per (day, period): a wave takes four periods (pm_support16).
__global__ void k_periods(const double* __restrict__ wind, const int* __restrict__ day_keys,
                          ModelParams mp, const int* day_idx, const double* start_time,
                          const double* hprob, PeriodInfo* pinfo) {
  const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int T = mp.T;
  const int d = blockIdx.y;
  if (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 >= T) return;   // whole wave beyond the day
  const bool valid = grp < T;          // the last wave of a day may hold fewer than four periods
  const int t = valid ? grp : T - 1;
  const int di = day_idx[d];
  const double* dw = wind + (int64_t)di * T * 3;
  PeriodInfo pi;
  const double st = start_time[d];
  const int start_indx = st < 0 ? 0 : (int)floor(st * T);
  pi.skip = t < start_indx;
  pi.hprob = hprob[(int64_t)d * T + t];
  const int np_ = mp.n_periods;
  double mx, my;
  const bool has_next = (di + 1 < mp.ndays_wind) && (day_keys[di + 1] == day_keys[di] + 1);
  if (!mp.test_run && np_ > 1) {
    if (t + np_ - 1 < T) {
      double sx = 0, sy = 0;
      for (int i = t; i < t + np_; ++i) { sx += dw[i * 3]; sy += dw[i * 3 + 1]; }
      mx = sx / np_;
      my = sy / np_;
    } else if (has_next) {
      const double* nw = dw + (int64_t)T * 3;
      double sx, sy;
      if (t != T - 1) {
        sx = 0; sy = 0;
        for (int i = t; i < T; ++i) { sx += dw[i * 3]; sy += dw[i * 3 + 1]; }
      } else {
        sx = dw[(T - 1) * 3];
        sy = dw[(T - 1) * 3 + 1];
      }
      const int wrap = np_ - (T - t);
      if (wrap != 1) {
        double ax = 0, ay = 0;
        for (int i = 0; i < wrap; ++i) { ax += nw[i * 3]; ay += nw[i * 3 + 1]; }
        sx += ax;
        sy += ay;
      } else {
        sx += nw[0];
        sy += nw[1];
      }
      mx = sx / np_;
      my = sy / np_;
    } else {
      if (t != T - 1) {
        double sx = 0, sy = 0;
        for (int i = t; i < T; ++i) { sx += dw[i * 3]; sy += dw[i * 3 + 1]; }
        mx = sx / (T - t);
        my = sy / (T - t);
      } else {
        mx = dw[(T - 1) * 3];
        my = dw[(T - 1) * 3 + 1];
      }
    }
  } else if (!mp.test_run) {
    mx = dw[t * 3];
    my = dw[t * 3 + 1];
  } else {
    mx = dw[0];
    my = dw[1];
  }
  const double fac = 86400.0 * ((double)np_ / (double)T);
  mx = mx * fac; my = my * fac;
  mx = mx * mp.mu_r; my = my * mp.mu_r;
  const double c = mp.cell;
  const double rx = rint(mx / c), ry = rint(my / c);
  pi.mux = mx - rx * c;
  pi.muy = my - ry * c;
  pi.cc = mp.rad_res + (int)rint(mx / c);
  pi.rc = mp.rad_res + (int)rint(-my / c);
  pi.H = pm_support16(mp.rule, mp.sdx, mp.sdy, pi.mux, pi.muy, c, 4 * mp.rad_res + 64, valid && !pi.skip);
  if (valid && (threadIdx.x & 15) == 0) pinfo[(int64_t)d * T + t] = pi;
}

// per day: losses in period order (ParasitoidModel.py:541-558), hprob bounds
// (:528-537), bounding box of the windows, local-stamp support.  One block per day.
__global__ void k_day_prep(ModelParams mp, const double* start_time, const PeriodInfo* pinfo,
                           DayInfo* dinfo, const int* day_idx) {
  const int d = blockIdx.x;
  const int T = mp.T, N = mp.N;
  extern __shared__ PeriodInfo dp_lds[];  // [T]
  PeriodInfo* pi = dp_lds;
  for (int t = threadIdx.x; t < T; t += blockDim.x) pi[t] = pinfo[(int64_t)d * T + t];
  __shared__ int s_box[4];
  __shared__ int s_hl;
  if (threadIdx.x == 0) { s_box[0] = N; s_box[1] = -1; s_box[2] = N; s_box[3] = -1; }
  __syncthreads();
  int r0 = N, r1 = -1, c0 = N, c1 = -1;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    if (pi[t].skip) continue;
    const int a0 = max(0, pi[t].rc - pi[t].H), a1 = min(N - 1, pi[t].rc + pi[t].H);
    const int b0 = max(0, pi[t].cc - pi[t].H), b1 = min(N - 1, pi[t].cc + pi[t].H);
    if (a0 <= a1 && b0 <= b1) {
      r0 = min(r0, a0); r1 = max(r1, a1); c0 = min(c0, b0); c1 = max(c1, b1);
    }
  }
  atomicMin(&s_box[0], r0); atomicMax(&s_box[1], r1);
  atomicMin(&s_box[2], c0); atomicMax(&s_box[3], c1);
  if (threadIdx.x < 64) {
    const int hl = pm_support(mp.lrule, mp.lsdx, mp.lsdy, 0.0, 0.0, mp.cell, 4 * mp.rad_res + 64);
    if (threadIdx.x == 0) s_hl = hl;
  }
  __syncthreads();
  // per-period loss terms in parallel (most are exactly 0), then summed in period order by
  // one thread so the floating-point result is the reference's sequential `loss +=`
  double* lterm = reinterpret_cast<double*>(pi + T);   // [T], after the staged PeriodInfo
  __shared__ int s_status, s_warned;
  if (threadIdx.x == 0) { s_status = 0; s_warned = 0; }
  __syncthreads();
  const double c = mp.cell;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    double term = 0.0;
    if (!pi[t].skip) {
      const double hp = pi[t].hprob;
      if (!(-1e-9 <= hp && hp <= 1.000000001)) atomicMin(&s_status, -7);
      const int H = pi[t].H, rc = pi[t].rc, cc = pi[t].cc;
      const int rmin = rc - H, rmax = rc + H, cmin = cc - H, cmax = cc + H;
      // stamp index ranges kept after clipping (ParasitoidModel.py:508-527)
      int rs = 0, re = 2 * H + 1, cs = 0, ce = 2 * H + 1;
      if (rmax + 1 > N) re = max(0, re - (rmax + 1 - N));
      if (cmax + 1 > N) ce = max(0, ce - (cmax + 1 - N));
      if (rmin < 0) rs = -rmin;
      if (cmin < 0) cs = -cmin;
      const bool clipped = rs > 0 || re < 2 * H + 1 || cs > 0 || ce < 2 * H + 1;
      const bool empty = rs >= re || cs >= ce;
      if (empty) {
        // wasps have left the domain (ParasitoidModel.py:547-558).  Python's negative
        // slice stops make the reference raise (and warn) only for exits through
        // the top / left edge.
        if ((rmax <= -2 && rmax >= -N) || (cmax <= -2 && cmax >= -N)) atomicOr(&s_warned, 1);
        term = hp;
      } else if (clipped) {
        // sum of the kept stamp cells == rectangle probability of the kept block
        const double xl = (cs - H) * c - c / 2, xu = (ce - 1 - H) * c + c / 2;
        const double yu = (H - rs) * c + c / 2, yl = (H - (re - 1)) * c - c / 2;
        const double kept = pm_rect(mp.rule, mp.sdx, mp.sdy, pi[t].mux, pi[t].muy, xl, xu, yl, yu);
        term = (1 - kept) * hp;
      }
    }
    lterm[t] = term;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    DayInfo di;
    di.status = s_status; di.warned = s_warned;
    double loss = 0.0;
    for (int t = 0; t < T; ++t) loss += lterm[t];
    di.loss = loss;
    const double st = start_time[d];
    di.start_indx = st < 0 ? 0 : (int)floor(st * T);
    di.r0 = s_box[0]; di.r1 = s_box[1]; di.c0 = s_box[2]; di.c1 = s_box[3];
    di.Hl = s_hl;
    di.pmfsum = 0; di.total = 0; di.delta = 0; di.pmfmin = 0; di.ksum = 0; di.nnz = 0; di.rad = 0;
    di.day_idx = day_idx[d];
    dinfo[d] = di;
  }
}

// ---------------------------------------------------------------- accumulate
// pmf[window] += hprob[t] * stamp_t for every period t (ParasitoidModel.py:539-540), with the
// reference's accumulation order per cell (ascending t) but the expensive part -- the corner
// values of the bivariate normal -- fully parallel:
//   k_tile_count / k_tile_fill   per 16 x 16 pmf tile: the ORDERED list of periods whose stamp
//                                window touches the tile ("pairs"); two passes around a scan,
//                                so that only real pairs are stored
//   k_pair_masses                one wave per (tile, period) pair: the Phi factors of the needed
//                                column/row edges, the BVU corner values of the part of the tile
//                                under the window, the 256 cell masses times hprob[t] -> a 2 KB
//                                record per pair.  Independent waves: no chain through a tile
//   k_tile_accumulate            per tile: the records of its pairs added in list order with
//                                explicit __dadd_rn (bit-identical to a sequential loop over t)
#define PM_NC ((PM_TS + 1) * (PM_TS + 1))
#define PM_CELLS (PM_TS * PM_TS)

__device__ __forceinline__ bool pm_tile_in_box(const DayInfo& di, int i0, int j0) {
  return !(i0 > di.r1 || i0 + PM_TS - 1 < di.r0 || j0 > di.c1 || j0 + PM_TS - 1 < di.c0);
}
__device__ __forceinline__ bool pm_hits(const PeriodInfo& p, int i0, int j0) {
  return !p.skip && !(p.rc + p.H < i0 || p.rc - p.H > i0 + PM_TS - 1 || p.cc + p.H < j0 ||
                      p.cc - p.H > j0 + PM_TS - 1);
}

// tcnt[(d * nt + ty) * nt + tx] = number of periods touching the tile (256 threads)
__global__ void __launch_bounds__(256)
k_tile_count(ModelParams mp, const PeriodInfo* __restrict__ pinfo, const DayInfo* __restrict__ dinfo,
             long long* __restrict__ tcnt) {
  const int d = blockIdx.z, T = mp.T;
  const int i0 = blockIdx.y * PM_TS, j0 = blockIdx.x * PM_TS;
  const int64_t tile = ((int64_t)d * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  __shared__ int s_cnt[4];
  int n = 0;
  if (pm_tile_in_box(dinfo[d], i0, j0)) {
    const PeriodInfo* pi = pinfo + (int64_t)d * T;
    for (int t = threadIdx.x; t < T; t += 256) n += pm_hits(pi[t], i0, j0) ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) tcnt[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ordered list of the tile's periods at toff[tile] - base: period index and tile id per pair
__global__ void __launch_bounds__(256)
k_tile_fill(ModelParams mp, const PeriodInfo* __restrict__ pinfo, const DayInfo* __restrict__ dinfo,
            const long long* __restrict__ tcnt, const long long* __restrict__ toff, int d0, long long base,
            long long cap, int* __restrict__ pair_t, int* __restrict__ pair_tile) {
  const int d = d0 + blockIdx.z, T = mp.T;
  const int i0 = blockIdx.y * PM_TS, j0 = blockIdx.x * PM_TS;
  const int64_t tile = ((int64_t)d * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (tcnt[tile] == 0) return;
  const long long out0 = toff[tile] - base;
  const PeriodInfo* pi = pinfo + (int64_t)d * T;
  __shared__ int s_wcnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nlist = 0;
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const bool hit = t < T && pm_hits(pi[t], i0, j0);
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int pos = nlist;
    for (int w = 0; w < wave; ++w) pos += s_wcnt[w];
    if (hit) {
      const long long o = out0 + pos + __popcll(m & ((1ull << lane) - 1ull));
      if (o < cap) {   // (the lists were sized from the previous batch: an overflow is detected and redone)
        pair_t[o] = t;
        pair_tile[o] = (int)(tile - (int64_t)d0 * gridDim.y * gridDim.x);
      }
    }
    nlist += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
}

// LDS traffic of ONE wave: its own writes are visible to its own later reads once the LDS queue
// has drained (in-order per wave); the "memory" clobber keeps the compiler from moving accesses
#define PM_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// One wave per SEGMENT of `seg` consecutive entries of the ordered pair list (4 segments per
// 256-thread workgroup; the waves share nothing, so there is no workgroup barrier and a wave whose
// pairs have few corners under their window does not wait for its neighbours).  For every pair:
// the Phi factors of the needed column / row edges, the BVU corner values of the part of the tile
// under the stamp window, the 256 cell masses times hprob[t] -- added, in list order, to a record
// held in registers.  The record goes to hm[first pair of the run] whenever the tile changes inside
// the segment and at its end: runs start at the first pair of a tile and at every multiple of
// `seg`, which is where k_tile_accumulate looks for them.  With seg = 1 every pair has its own
// record and the sums are those of a sequential loop over the periods bit for bit (the reference's
// order, ParasitoidModel.py:539-540); seg = 8 (default) adds eight periods at a time before the
// ordered pass -- 1/8 of the 2 KB records (850 MB written and read back per 18-day evaluation at
// R = 400 otherwise), sums regrouped at the 1e-16 level, still the same on every run.
// HIGH: the correlation takes Genz's |rho| >= 0.925 branch (uniform per batch; the host picks the
// instance).  Keeping that branch out of the common instance is worth 50 registers: 165 -> ~110,
// i.e. four resident waves per SIMD instead of three.
// LG: the rule's number of node pairs as a compile-time constant (3 or 6; 0 = read it at run time)
template <bool HIGH, int LG = 0>
__global__ void __launch_bounds__(256, HIGH ? 2 : (LG >= 6 ? 2 : (LG > 0 ? 3 : 4)))   // run-time-count instance: 128 registers (2 spilled), four waves per SIMD; unrolled: 168, three
k_pair_masses(ModelParams mp, const PeriodInfo* __restrict__ pinfo, int d0, int nt, long long npairs_, int seg,
              const long long* __restrict__ np_dev,   // != nullptr: the real pair count (npairs_ = capacity of the lists)
              const int* __restrict__ pair_t, const int* __restrict__ pair_tile, double* __restrict__ hm) {
  const long long npairs = np_dev ? (*np_dev < npairs_ ? *np_dev : npairs_) : npairs_;
  __shared__ double s_b[4][PM_NC];
  __shared__ double s_px[4][PM_TS + 1], s_py[4][PM_TS + 1];   // Phi(-h_a), Phi(-k_b)
  __shared__ double s_hx[4][PM_TS + 1], s_ky[4][PM_TS + 1];   // h_a, k_b
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long p0 = ((long long)blockIdx.x * 4 + g) * seg;
  if (p0 >= npairs) return;                       // whole wave; no workgroup barrier below
  const long long p1 = p0 + seg < npairs ? p0 + seg : npairs;
  const double c = mp.cell;
  const int N = mp.N;
  double acc[PM_CELLS / 64];
#pragma unroll
  for (int u = 0; u < PM_CELLS / 64; ++u) acc[u] = 0.0;
  int cur = pair_tile[p0];
  long long run0 = p0;
  auto flush = [&]() {
    double* out = hm + run0 * PM_CELLS;
#pragma unroll
    for (int u = 0; u < PM_CELLS / 64; ++u) {
      out[lane + 64 * u] = acc[u];
      acc[u] = 0.0;
    }
  };
  for (long long pr = p0; pr < p1; ++pr) {
    const int tl = pair_tile[pr];
    if (tl != cur) {
      flush();
      cur = tl;
      run0 = pr;
    }
    const int d = d0 + tl / (nt * nt), rem = tl % (nt * nt);
    const int i0 = (rem / nt) * PM_TS, j0 = (rem % nt) * PM_TS;
    const PeriodInfo p = pinfo[(int64_t)d * mp.T + pair_t[pr]];
    // only the part of the tile under the stamp window needs corner values
    const int ja0 = max(j0, p.cc - p.H), ja1 = min(j0 + PM_TS - 1, p.cc + p.H);
    const int ib0 = max(i0, p.rc - p.H), ib1 = min(i0 + PM_TS - 1, p.rc + p.H);
    const int a0 = ja0 - j0, na = ja1 - ja0 + 2;     // corners of columns ja0..ja1
    const int b0 = ib0 - i0, nb = ib1 - ib0 + 2;     // corners of rows ib0..ib1
    if (lane < na) {                 // column edges: x of the lower edge of column j0+a
      const int a = a0 + lane;
      const double x = (j0 + a - p.cc) * c - c / 2;
      const double h = (x - p.mux) / mp.sdx;
      s_hx[g][a] = h;
      s_px[g][a] = pm_phi(-h);
    } else if (lane >= 32 && lane < 32 + nb) {   // row edges: upper y edge of row i0+b
      const int b = b0 + lane - 32;
      const double y = (p.rc - (i0 + b)) * c + c / 2;
      const double k = (y - p.muy) / mp.sdy;
      s_ky[g][b] = k;
      s_py[g][b] = pm_phi(-k);
    }
    PM_WAVE_LDS_SYNC();
    // idx / na by multiplication: na <= 17 and idx < 289, where (idx * ceil(2^16 / na)) >> 16 is exact
    const unsigned na_magic = (65536u + (unsigned)na - 1u) / (unsigned)na;   // wave-uniform
    for (int idx = lane; idx < na * nb; idx += 64) {
      const int qb = (int)(((unsigned)idx * na_magic) >> 16);
      const int b = b0 + qb, a = a0 + (idx - qb * na);
      if (HIGH) s_b[g][b * (PM_TS + 1) + a] = pm_bvu(mp.rule, s_hx[g][a], s_ky[g][b]);
      else if (LG > 0) s_b[g][b * (PM_TS + 1) + a] = pm_bvu_low_phi_t<(LG > 0 ? LG : 1)>(mp.rule, s_hx[g][a], s_ky[g][b], s_px[g][a], s_py[g][b]);
      else s_b[g][b * (PM_TS + 1) + a] = pm_bvu_low_phi(mp.rule, s_hx[g][a], s_ky[g][b], s_px[g][a], s_py[g][b]);
    }
    PM_WAVE_LDS_SYNC();
#pragma unroll
    for (int u = 0; u < PM_CELLS / 64; ++u) {
      const int cell = lane + 64 * u;
      const int li = cell / PM_TS, lj = cell % PM_TS;
      const int i = i0 + li, j = j0 + lj;
      const int ii = j - p.cc, jj = p.rc - i;
      double v = 0.0;    // +0.0 leaves the accumulator bit-identical when the cell is outside
      if (ii >= -p.H && ii <= p.H && jj >= -p.H && jj <= p.H && i < N && j < N) {
        // cell [xl,xu] x [yl,yu]: BVU(xl,yl) - BVU(xu,yl) - BVU(xl,yu) + BVU(xu,yu)
        const double ll = s_b[g][(li + 1) * (PM_TS + 1) + lj];
        const double ul = s_b[g][(li + 1) * (PM_TS + 1) + lj + 1];
        const double lu = s_b[g][li * (PM_TS + 1) + lj];
        const double uu = s_b[g][li * (PM_TS + 1) + lj + 1];
        const double mass = ((ll - ul) - lu) + uu;
        v = __dmul_rn(p.hprob, mass);
      }
      acc[u] = __dadd_rn(acc[u], v);
    }
    PM_WAVE_LDS_SYNC();              // the next pair overwrites the tables
  }
  flush();
}

// number of pairs of a batch = last offset + last count (device-side, so that the host does not
// have to wait for the scan before it can enqueue the pair kernels)
static __global__ void k_pair_total(const long long* __restrict__ toff, const long long* __restrict__ tcnt, long long ntot,
                                    long long* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = toff[ntot - 1] + tcnt[ntot - 1];
}

// per tile: add the run records of its pairs in list (= period) order.  Runs start at the tile's
// first pair and at every multiple of `seg` inside its range (k_pair_masses).
__global__ void __launch_bounds__(PM_CELLS)
k_tile_accumulate(ModelParams mp, const long long* __restrict__ tcnt, const long long* __restrict__ toff,
                  int d0, long long base, int seg, long long cap, const double* __restrict__ hm, double* pmf /*[nd][N][N]*/) {
  const int d = d0 + blockIdx.z, N = mp.N;
  const int64_t tile = ((int64_t)d * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const long long np = tcnt[tile];
  if (np == 0) return;
  const long long o = toff[tile] - base;                      // first pair of the tile in the chunk's list
  if (o + np > cap) return;                                   // beyond the lists' capacity: the batch is redone
  const long long n = (o + np - 1) / seg - o / seg + 1;       // runs
  const long long g0 = o / seg;
  // k-th run record of this tile
  auto at = [&](long long k) -> const double* {
    const long long first = k == 0 ? o : (g0 + k) * seg;
    return hm + first * PM_CELLS + threadIdx.x;
  };
  double acc = 0.0;
  // sixteen loads in flight while the previous sixteen are added; the additions stay in list order
  long long q = 0;
  double v[16];
  const long long nfull = n / 16 * 16;
  if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *at(u);
  }
  for (; q < nfull; q += 16) {
    double w[16];
    const bool more = q + 16 < nfull;
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = more ? *at(q + 16 + u) : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __dadd_rn(acc, v[u]);
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = w[u];
  }
  for (; q < n; ++q) acc = __dadd_rn(acc, *at(q));
  const int i = blockIdx.y * PM_TS + threadIdx.x / PM_TS, j = blockIdx.x * PM_TS + threadIdx.x % PM_TS;
  if (i < N && j < N) pmf[((int64_t)d * N + i) * N + j] = acc;
}

// sum and min of each day's pmf: partials per (day, block) then k_pmf_reduce2
__global__ void k_pmf_reduce1(const double* __restrict__ pmf, int64_t n, double* psum, double* pmin) {
  const int d = blockIdx.y;
  const double* p = pmf + (int64_t)d * n;
  double s = 0.0, m = 1e300;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = p[i];
    s += v;
    m = fmin(m, v);
  }
  __shared__ double ss[256], sm[256];
  ss[threadIdx.x] = s; sm[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ss[threadIdx.x] += ss[threadIdx.x + off];
      sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { psum[d * gridDim.x + blockIdx.x] = ss[0]; pmin[d * gridDim.x + blockIdx.x] = sm[0]; }
}

// per day: finish the reduction, assertion checks (ParasitoidModel.py:566-580), and the
// non-flyer stamp around the origin (:581-599).  phase 0: first block; phase 1: after stamp.
__global__ void k_day_local(ModelParams mp, const double* psum, const double* pmin, int nblk,
                            DayInfo* dinfo, double* pmf, int phase) {
  const int d = blockIdx.x;
  DayInfo& di = dinfo[d];
  __shared__ double s_sum, s_min;
  if (threadIdx.x == 0) {
    double s = 0.0, m = 1e300;
    for (int b = 0; b < nblk; ++b) { s += psum[d * nblk + b]; m = fmin(m, pmin[d * nblk + b]); }
    s_sum = s; s_min = m;
  }
  __syncthreads();
  const double pmfsum = s_sum, pmfmin = s_min;
  if (phase == 1) {
    if (threadIdx.x == 0 && di.total < 0.99999 && !di.status) {
      const double total2 = pmfsum + di.loss;
      if (!(pmfmin >= -1e-8)) di.status = -8;
      else if (!(total2 <= 1.00001)) di.status = -9;
      di.pmfmin = pmfmin;
    }
    return;
  }
  __shared__ double s_total;
  if (threadIdx.x == 0) {
    di.pmfsum = pmfsum;
    di.pmfmin = pmfmin;
    di.total = pmfsum + di.loss;
    if (!di.status) {
      if (!(di.loss >= 0.0)) di.status = -9;
      else if (!(pmfmin >= -1e-8)) di.status = -8;
      else if (!(pmfsum <= 1.00001)) di.status = -9;
    }
    s_total = di.total;
  }
  __syncthreads();
  const double total = s_total;
  if (!(total < 0.99999)) return;
  const int H = di.Hl, side = 2 * H + 1, R = mp.rad_res, N = mp.N;
  const double c = mp.cell;
  for (int q = threadIdx.x; q < side * side; q += blockDim.x) {
    const int rho = q / side, kap = q % side;
    const int ii = kap - H, jj = H - rho;
    const double xl = ii * c - c / 2, yl = jj * c - c / 2;
    const double mass = pm_rect(mp.lrule, mp.lsdx, mp.lsdy, 0.0, 0.0, xl, xl + c, yl, yl + c);
    const int i = R - H + rho, j = R - H + kap;
    if (i >= 0 && i < N && j >= 0 && j < N) {
      double* cellp = pmf + ((int64_t)d * N + i) * N + j;
      *cellp = __dadd_rn(*cellp, __dmul_rn(1 - total, mass));
    }
  }
}

// r_small_vals(prob_model=True) statistics + bounding radius (CalcSol.py:126-135,
// ParasitoidModel.py:605-610): per row count, sum and max |j - R| of kept entries
__global__ void k_pmf_row_stats(const double* __restrict__ pmf, int N, int R, double negval,
                                double* rowsum, long long* rowcnt, int* rowrad) {
  const int d = blockIdx.y, r = blockIdx.x;
  const double* p = pmf + ((int64_t)d * N + r) * N;
  double s = 0.0;
  int c = 0, rad = -1;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const double t = p[j];
    if (t != 0.0 && !(t < negval)) {
      s += t;
      ++c;
      rad = max(rad, abs(j - R));
    }
  }
  __shared__ double ss[256];
  __shared__ int sc[256], sr[256];
  ss[threadIdx.x] = s; sc[threadIdx.x] = c; sr[threadIdx.x] = rad;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ss[threadIdx.x] += ss[threadIdx.x + off];
      sc[threadIdx.x] += sc[threadIdx.x + off];
      sr[threadIdx.x] = max(sr[threadIdx.x], sr[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rowsum[(int64_t)d * N + r] = ss[0];
    rowcnt[(int64_t)d * N + r] = sc[0];
    rowrad[(int64_t)d * N + r] = sc[0] > 0 ? max(sr[0], abs(r - R)) : -1;
  }
}

__global__ void k_pmf_day_stats(const double* rowsum, const long long* rowcnt, const int* rowrad,
                                int N, DayInfo* dinfo) {
  const int d = blockIdx.x;
  __shared__ double ss[256];
  __shared__ long long sc[256];
  __shared__ int sr[256];
  double s = 0.0;
  long long c = 0;
  int rad = -1;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    s += rowsum[(int64_t)d * N + r];
    c += rowcnt[(int64_t)d * N + r];
    rad = max(rad, rowrad[(int64_t)d * N + r]);
  }
  ss[threadIdx.x] = s; sc[threadIdx.x] = c; sr[threadIdx.x] = rad;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ss[threadIdx.x] += ss[threadIdx.x + off];
      sc[threadIdx.x] += sc[threadIdx.x + off];
      sr[threadIdx.x] = max(sr[threadIdx.x], sr[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    DayInfo& di = dinfo[d];
    di.nnz = sc[0];
    di.ksum = ss[0];
    di.delta = sc[0] > 0 ? (1 - ss[0]) / (double)sc[0] : 0.0;
    di.rad = sr[0];
    if (!di.status && sc[0] == 0) di.status = -11;
  }
}

// ordered compaction of one day's pmf into COO with the shrink offset (:611-613)
__global__ void k_pmf_compact(const double* __restrict__ pmf, int N, double negval, double delta,
                              int idx_off, const long long* rowoff, int* orow, int* ocol, double* oval) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= N) return;
  long long base = rowoff[wave];
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int c = c0 + lane;
    double t = 0.0;
    bool keep = false;
    if (c < N) {
      t = pmf[(int64_t)wave * N + c];
      keep = (t != 0.0) && !(t < negval);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int o = __popcll(m & ((1ull << lane) - 1ull));
      orow[base + o] = wave + idx_off;
      ocol[base + o] = c + idx_off;
      oval[base + o] = t + delta;
    }
    base += __popcll(m);
  }
}

// all days of a batch in one launch (blockIdx.y = day): threshold, renormalisation shift and
// shrink offset come from the day's DayInfo, the output position from `off`
__global__ void k_pmf_compact_batch(const double* __restrict__ pmf_, int N, double negval, int rad_res,
                                    const DayInfo* __restrict__ dinfo, const long long* __restrict__ off,
                                    const long long* __restrict__ rowoff_, int* orow, int* ocol, double* oval) {
  const int d = blockIdx.y;
  const DayInfo& di = dinfo[d];
  if (di.status != 0 || di.nnz == 0) return;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= N) return;
  const double* pmf = pmf_ + (int64_t)d * N * N;
  const double delta = di.delta;
  const int idx_off = -rad_res + di.rad;
  long long base = off[d] + rowoff_[(int64_t)d * N + wave];
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int c = c0 + lane;
    double t = 0.0;
    bool keep = false;
    if (c < N) {
      t = pmf[(int64_t)wave * N + c];
      keep = (t != 0.0) && !(t < negval);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int o = __popcll(m & ((1ull << lane) - 1ull));
      orow[base + o] = wave + idx_off;
      ocol[base + o] = c + idx_off;
      oval[base + o] = t + delta;
    }
    base += __popcll(m);
  }
}

// get_mvn_cdf_values (ParasitoidModel.py:311-380) for one (cell, mu, S)
__global__ void k_mvn_cdf_values(BvuRule rule, double sdx, double sdy, double mux, double muy,
                                 double cell, int hmax, int* Hout, double* out, long long cap) {
  __shared__ int s_h;
  if (threadIdx.x < 64) {
    const int h = pm_support(rule, sdx, sdy, mux, muy, cell, hmax);
    if (threadIdx.x == 0) s_h = h;
  }
  __syncthreads();
  const int H = s_h, side = 2 * H + 1;
  if (threadIdx.x == 0) *Hout = H;
  if ((long long)side * side > cap) return;
  for (int q = threadIdx.x; q < side * side; q += blockDim.x) {
    const int rho = q / side, kap = q % side;
    const int ii = kap - H, jj = H - rho;
    const double xl = ii * cell - cell / 2, yl = jj * cell - cell / 2;
    out[q] = pm_rect(rule, sdx, sdy, mux, muy, xl, xl + cell, yl, yl + cell);
  }
}
