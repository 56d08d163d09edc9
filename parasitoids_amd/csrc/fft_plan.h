// Host-side planning for the in-LDS FFT programs: factorisation into radix
// stages, the row-mode split/padding, twiddle tables and digit-reversal tables.
// Pure host C++ (no HIP calls) so tests/host/fft_emul.cpp can use it directly.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "fft_core.h"

struct HostFftPlan {
  FftProg prog;  // device pointers left null here; filled by the uploader
  std::vector<cplx> tw_lo, tw_hi, tw_gen;
  std::vector<cplx> tw_all;  // lo | hi | gen, the layout the kernels keep in LDS
  std::vector<uint32_t> pos, pos_phys;
  int max_prime = 1;
};

inline bool ps_is_smooth7(int n) {
  for (int p : {2, 3, 5, 7})
    while (n % p == 0) n /= p;
  return n == 1;
}

// smallest even 7-smooth integer >= n (fast-mode FFT size)
inline int ps_next_fast_len(int n) {
  int v = n + (n & 1);
  while (!ps_is_smooth7(v)) v += 2;
  return v;
}

// `big`: also use the radix-18 / radix-16 register butterflies (fewer stages; they need
// ~200 VGPRs, so only the row passes, which run 512-thread workgroups, ask for them)
inline std::vector<int> ps_factor_radices(int L, int* max_prime, bool big = false) {
  std::vector<int> r;
  int n = L;
  *max_prime = 1;
  if (big) {
    while (n % 18 == 0 && n > 18) { r.push_back(18); n /= 18; }
    while (n % 16 == 0) { r.push_back(16); n /= 16; }
  }
  while (n % 9 == 0) { r.push_back(9); n /= 9; }
  while (n % 8 == 0) { r.push_back(8); n /= 8; }
  while (n % 7 == 0) { r.push_back(7); n /= 7; }
  while (n % 5 == 0) { r.push_back(5); n /= 5; }
  while (n % 4 == 0) { r.push_back(4); n /= 4; }
  while (n % 3 == 0) { r.push_back(3); n /= 3; }
  while (n % 2 == 0) { r.push_back(2); n /= 2; }
  for (int p = 11; (int64_t)p * p <= n; p += 2)
    while (n % p == 0) { r.push_back(p); n /= p; }
  if (n > 1) r.push_back(n);
  for (int v : r) *max_prime = std::max(*max_prime, v);
  if (r.empty()) r.push_back(1);
  return r;
}

// Build the program for length L.  `row_split`: arrange the radices into a
// leading and a trailing group (row-mode padded layout) when L is large enough.
// `forced`: use exactly these radices in this order, the first `forced_lead` of them as the
// leading group (the specialised 3-stage row kernels need a fixed layout)
inline bool ps_build_plan(int L, bool row_split, HostFftPlan* out, bool big = false,
                          const std::vector<int>* forced = nullptr, int forced_lead = 0) {
  HostFftPlan& hp = *out;
  FftProg& P = hp.prog;
  P = FftProg();
  P.L = L;
  std::vector<int> rad = ps_factor_radices(L, &hp.max_prime, big);
  if (forced) {
    rad = *forced;
    hp.max_prime = 1;
    for (int v : rad) hp.max_prime = std::max(hp.max_prime, v);
  }
  if (L == 1) rad.clear();
  if ((int)rad.size() > PS_MAX_STAGES) return false;
  for (int v : rad)
    if (v > 18 && v > PS_MAX_GENERIC_RADIX) return false;
  // choose the trailing group: subset with product closest to sqrt(L)
  std::vector<int> lead = rad, trail;
  if (row_split && L >= 512 && rad.size() >= 2) {
    const int nr = (int)rad.size();
    double best = 1e300;
    int bestmask = 0;
    for (int mask = 1; mask < (1 << nr) - 1; ++mask) {
      double prod = 1;
      for (int i = 0; i < nr; ++i)
        if (mask >> i & 1) prod *= rad[i];
      double score = fabs(log(prod) - 0.5 * log((double)L));
      if (score < best) { best = score; bestmask = mask; }
    }
    lead.clear();
    for (int i = 0; i < nr; ++i) (bestmask >> i & 1 ? trail : lead).push_back(rad[i]);
    int lbp = 1;
    for (int v : trail) lbp *= v;
    if (lbp < 8 || L / lbp < 8) {  // not worth splitting
      lead = rad;
      trail.clear();
    }
  }
  // large (generic) radices first inside each group, then descending
  auto order = [](std::vector<int>& v) { std::sort(v.begin(), v.end(), std::greater<int>()); };
  if (forced) {
    lead.assign(rad.begin(), rad.begin() + forced_lead);
    trail.assign(rad.begin() + forced_lead, rad.end());
  } else {
    order(lead);
    order(trail);
  }
  P.ns = 0;
  int n = L;
  for (int v : lead) { P.radix[P.ns] = v; P.n[P.ns] = n; P.m[P.ns] = n / v; n /= v; ++P.ns; }
  P.sa = P.ns;
  P.Lb = 1;
  for (int v : trail) P.Lb *= v;
  for (int v : trail) { P.radix[P.ns] = v; P.n[P.ns] = n; P.m[P.ns] = n / v; n /= v; ++P.ns; }
  if (trail.empty()) {
    P.sa = P.ns;
    P.Lb = 1;
    P.Lbp = 1;
    P.La = L;
  } else {
    P.La = L / P.Lb;
    P.Lbp = (P.Lb % 2 == 0) ? P.Lb + 1 : P.Lb;
  }
  for (int s = 0; s < P.ns; ++s) {
    P.step[s] = L / P.n[s];
    P.mg_m[s] = ps_magic((uint32_t)P.m[s]);
    P.mg_mh[s] = ps_magic((uint32_t)std::max(1, P.m[s] / P.Lb));
    P.mg_nbf[s] = ps_magic((uint32_t)(L / P.radix[s]));
  }
  P.mg_Lb = ps_magic((uint32_t)P.Lb);
  P.mg_La = ps_magic((uint32_t)P.La);
  // twiddles: w_L^t = exp(-2 pi i t / L) (forward sign), two-level tables
  P.tw_shift = (L <= 4096) ? 6 : 7;
  const int B = 1 << P.tw_shift;
  P.n_lo = B;
  P.n_hi = (L + B - 1) / B;
  hp.tw_lo.resize(P.n_lo);
  hp.tw_hi.resize(P.n_hi);
  const long double twopi = 6.283185307179586476925286766559L;
  for (int t = 0; t < P.n_lo; ++t) {
    long double a = twopi * (long double)(t % L) / (long double)L;
    hp.tw_lo[t] = make_double2((double)cosl(a), (double)-sinl(a));
  }
  for (int u = 0; u < P.n_hi; ++u) {
    long double a = twopi * (long double)(((int64_t)u * B) % L) / (long double)L;
    hp.tw_hi[u] = make_double2((double)cosl(a), (double)-sinl(a));
  }
  // w_r^t tables of the wave-cooperative prime radices
  P.n_gen = 0;
  hp.tw_gen.clear();
  for (int s = 0; s < P.ns; ++s) {
    P.gen_off[s] = -1;
    const int r = P.radix[s];
    if (r <= 9 || r == 16 || r == 18) continue;
    for (int s2 = 0; s2 < s; ++s2)
      if (P.radix[s2] == r) P.gen_off[s] = P.gen_off[s2];
    if (P.gen_off[s] >= 0) continue;
    P.gen_off[s] = (int)hp.tw_gen.size();
    for (int t = 0; t < r; ++t) {
      long double a = twopi * (long double)t / (long double)r;
      hp.tw_gen.push_back(make_double2((double)cosl(a), (double)-sinl(a)));
    }
  }
  P.n_gen = (int)hp.tw_gen.size();
  hp.tw_all = hp.tw_lo;
  hp.tw_all.insert(hp.tw_all.end(), hp.tw_hi.begin(), hp.tw_hi.end());
  hp.tw_all.insert(hp.tw_all.end(), hp.tw_gen.begin(), hp.tw_gen.end());
  // digit reversal: k = k0 + r0 (k1 + r1 (...)) lives at sum_s k_s m_s
  hp.pos.resize(L);
  hp.pos_phys.resize(L);
  for (int k = 0; k < L; ++k) {
    int rem = k, p = 0;
    for (int s = 0; s < P.ns; ++s) {
      int d = rem % P.radix[s];
      rem /= P.radix[s];
      p += d * P.m[s];
    }
    hp.pos[k] = (uint32_t)p;
    hp.pos_phys[k] = (uint32_t)row_phys(P, p);
  }
  return true;
}

// two-level table for the 4-step twiddles w_P^t, t < P
struct HostTwiddle {
  int shift = 0, n_lo = 0, n_hi = 0;
  std::vector<cplx> lo, hi;
};
inline void ps_build_twiddle(int P, HostTwiddle* tw) {
  tw->shift = (P <= 4096) ? 6 : 7;
  const int B = 1 << tw->shift;
  tw->n_lo = B;
  tw->n_hi = (P + B - 1) / B;
  tw->lo.resize(tw->n_lo);
  tw->hi.resize(tw->n_hi);
  const long double twopi = 6.283185307179586476925286766559L;
  for (int t = 0; t < tw->n_lo; ++t) {
    long double a = twopi * (long double)(t % P) / (long double)P;
    tw->lo[t] = make_double2((double)cosl(a), (double)-sinl(a));
  }
  for (int u = 0; u < tw->n_hi; ++u) {
    long double a = twopi * (long double)(((int64_t)u * B) % P) / (long double)P;
    tw->hi[u] = make_double2((double)cosl(a), (double)-sinl(a));
  }
}

// Split a column transform of length P into L1 * L2 so that each sub-transform
// tile fits LDS.  Returns L1 (L2 = P / L1); L1 == P means a single pass.
inline int ps_choose_col_split(int P, int single_pass_max) {
  if (P <= single_pass_max) return P;
  int best = -1;
  double bestscore = 1e300;
  for (int d = 2; (int64_t)d * d <= P; ++d) {
    if (P % d) continue;
    for (int a : {d, P / d}) {
      int b = P / a;
      int mx = std::max(a, b);
      double score = mx;
      if (score < bestscore) { bestscore = score; best = std::min(a, b); }
    }
  }
  if (best < 0) return P;  // prime: single pass, narrow tile
  return best;
}
