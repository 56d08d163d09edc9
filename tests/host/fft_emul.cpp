// Host emulation of the in-LDS FFT program (parasitoids_amd/csrc/fft_core.h):
// runs every stage thread by thread (wave-cooperative stages lane by lane with
// "all lanes compute, then all lanes store") and compares against a long-double
// DFT.  Usage: fft_emul L [L ...]   -> prints max relative error per case.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <complex>
#include "../../parasitoids_amd/csrc/fft_plan.h"
#include "../../parasitoids_amd/csrc/fft_rs.h"
#include <array>

typedef std::complex<long double> lc;

static void naive(const std::vector<lc>& x, std::vector<lc>& y, int sign) {
  int L = (int)x.size();
  const long double twopi = 6.283185307179586476925286766559L;
  std::vector<lc> w(L);
  for (int t = 0; t < L; ++t) w[t] = lc(cosl(twopi * t / L), sign * sinl(twopi * t / L));
  for (int k = 0; k < L; ++k) {
    lc acc = 0;
    for (int q = 0; q < L; ++q) acc += x[q] * w[(int)(((int64_t)q * k) % L)];
    y[k] = acc;
  }
}

template <int DIR>
static void emul_stage(std::vector<cplx>& data, const HostFftPlan& hp, int s, int mode, int nb,
                       int wsh, int bs, int nthr) {
  const FftProg& P = hp.prog;
  int r = P.radix[s];
  bool spec = (r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8 || r == 9 || r == 16 || r == 18);
  if (spec) {
    for (int tid = 0; tid < nthr; ++tid)
      run_stage<DIR>(data.data(), hp.tw_all.data(), hp.tw_all.data() + P.n_lo, P, s, mode, nb, wsh, bs, tid, nthr);
  } else {
    int nbutter = (P.L / r) * nb;
    int G = r < 64 ? 64 / r : 1;
    int nwaves = nthr / 64;
    const cplx* tlo = hp.tw_all.data();
    const cplx* thi = tlo + P.n_lo;
    // inverse: element-wise input twiddles first (a barrier separates the phases on the GPU)
    for (int tid = 0; tid < nthr; ++tid)
      gen_pretwiddle<DIR>(data.data(), tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr);
    for (int wave = 0; wave < nwaves; ++wave)
      for (int g0 = wave * G; g0 < nbutter; g0 += nwaves * G) {
        std::vector<GenAcc> acc(64);
        for (int lane = 0; lane < 64; ++lane)
          gen_compute<DIR>(acc[lane], data.data(), tlo, thi, P, s, mode, nb, wsh, bs, g0, lane);
        for (int lane = 0; lane < 64; ++lane) gen_store(acc[lane], data.data(), r);
      }
  }
}

// COL mode, psh != 0: the [L][W] tile sits at a row pitch of 1 << psh (k_col_fused_dual)
static double run_case(int L, int mode, int nb, int wsh, bool split, bool big = false, int psh = 0) {
  HostFftPlan hp;
  if (!ps_build_plan(L, split, &hp, big)) { printf("plan failed L=%d\n", L); return -1; }
  const FftProg& P = hp.prog;
  int bs = mode == PS_MODE_COL ? psh : row_pitch(P);
  int W = 1 << wsh;
  const int pitch = psh ? 1 << psh : W;
  int nbatch = mode == PS_MODE_COL ? W : nb;
  size_t ldsn = mode == PS_MODE_COL ? (size_t)L * pitch : (size_t)bs * nb;
  std::vector<cplx> data(ldsn, make_double2(1e300, 1e300));
  std::vector<std::vector<lc>> x(nbatch, std::vector<lc>(L)), y(nbatch, std::vector<lc>(L));
  srand(L * 7 + mode);
  auto addr = [&](int b, int i_logical) {
    return mode == PS_MODE_COL ? (size_t)i_logical * pitch + b : (size_t)b * bs + row_phys(P, i_logical);
  };
  for (int b = 0; b < nbatch; ++b)
    for (int i = 0; i < L; ++i) {
      double re = rand() / (double)RAND_MAX - 0.5, im = rand() / (double)RAND_MAX - 0.5;
      x[b][i] = lc(re, im);
      data[addr(b, i)] = make_double2(re, im);
    }
  int nthr = 256;
  int nbk = mode == PS_MODE_COL ? 1 : nb;
  for (int s = 0; s < P.ns; ++s) emul_stage<PS_FWD>(data, hp, s, mode, mode == PS_MODE_COL ? W : nbk, wsh, bs, nthr);
  double maxerr = 0, maxv = 0;
  for (int b = 0; b < nbatch; ++b) {
    naive(x[b], y[b], -1);
    for (int k = 0; k < L; ++k) {
      cplx v = data[addr(b, hp.pos[k])];
      if (mode == PS_MODE_ROW) {
        cplx v2 = data[(size_t)b * bs + hp.pos_phys[k]];
        if (v2.x != v.x || v2.y != v.y) { printf("pos_phys mismatch\n"); return -1; }
      }
      double e = (double)std::abs(lc(v.x, v.y) - y[b][k]);
      maxerr = e > maxerr ? e : maxerr;
      double a = (double)std::abs(y[b][k]);
      maxv = a > maxv ? a : maxv;
    }
  }
  double fwd = maxerr / maxv;
  // inverse: run the inverse stages on the forward result; must return L * x
  for (int s = P.ns - 1; s >= 0; --s) emul_stage<PS_INV>(data, hp, s, mode, mode == PS_MODE_COL ? W : nbk, wsh, bs, nthr);
  double ierr = 0;
  for (int b = 0; b < nbatch; ++b)
    for (int i = 0; i < L; ++i) {
      cplx v = data[addr(b, i)];
      double e = (double)std::abs(lc(v.x, v.y) / (long double)L - x[b][i]);
      ierr = e > ierr ? e : ierr;
    }
  printf("L=%d mode=%s nb=%d split=%d stages=", L, mode == PS_MODE_COL ? "col" : "row", nbatch, (int)split);
  for (int s = 0; s < P.ns; ++s) printf("%d%s", P.radix[s], s + 1 == P.sa ? "|" : ".");
  printf(" La=%d Lb=%d fwd_rel=%.3e inv_abs=%.3e\n", P.La, P.Lb, fwd, ierr);
  return fwd > ierr ? fwd : ierr;
}


// Register-resident three-stage transform (fft_rs.h): every thread keeps its butterfly in
// x[j]; the exchange goes through a padded word buffer exactly as on the GPU (real parts,
// then imaginary parts).  Also counts out-of-range / never-written exchange reads.
template <int R1, int R2, int R3, int DIR>
static double run_rs() {
  using S = Rs<R1, R2, R3>;
  const int L = S::L;
  HostFftPlan hp;
  if (!ps_build_plan(L, true, &hp)) return -1;
  const FftProg& P = hp.prog;
  const cplx* tlo = hp.tw_all.data();
  const cplx* thi = tlo + P.n_lo;
  std::vector<std::array<cplx, S::RMAX>> x(S::NTHR);
  const double poison = 1e300;
  std::vector<double> ex(S::XWORDS, poison);
  std::vector<lc> in(L), ref(L);
  srand(L * 3 + DIR);
  for (int i = 0; i < L; ++i) in[i] = lc(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
  for (int j = 0; j < S::T1; ++j) {
    for (int q = 0; q < R1; ++q) x[j][q] = make_double2((double)in[j + q * S::T1].real(), (double)in[j + q * S::T1].imag());
    bfly<R1, DIR>(x[j].data());
  }
  auto fresh = [&]() { std::fill(ex.begin(), ex.end(), poison); };
  fresh();
  for (int j = 0; j < S::T1; ++j) rs_put<R1, 0>(ex.data(), S::x1_w(j), 1, x[j].data());
  for (int j = 0; j < S::T2; ++j) rs_get<R2, 0>(ex.data(), S::x_r(j), S::X1_RS, x[j].data());
  fresh();
  for (int j = 0; j < S::T1; ++j) rs_put<R1, 1>(ex.data(), S::x1_w(j), 1, x[j].data());
  for (int j = 0; j < S::T2; ++j) rs_get<R2, 1>(ex.data(), S::x_r(j), S::X1_RS, x[j].data());
  for (int j = 0; j < S::T2; ++j)
    rs_stage<R2, DIR>(x[j].data(), tw_lookup(tlo, thi, P.tw_shift, S::tw2(j)), true);
  fresh();
  for (int j = 0; j < S::T2; ++j) rs_put<R2, 0>(ex.data(), S::x2_w(j), 17, x[j].data());
  for (int j = 0; j < S::T3; ++j) rs_get<R3, 0>(ex.data(), S::x_r(j), S::X2_RS, x[j].data());
  fresh();
  for (int j = 0; j < S::T2; ++j) rs_put<R2, 1>(ex.data(), S::x2_w(j), 17, x[j].data());
  for (int j = 0; j < S::T3; ++j) rs_get<R3, 1>(ex.data(), S::x_r(j), S::X2_RS, x[j].data());
  for (int j = 0; j < S::T3; ++j)
    rs_stage<R3, DIR>(x[j].data(), tw_lookup(tlo, thi, P.tw_shift, S::tw3(j)), true);
  naive(in, ref, DIR == PS_FWD ? -1 : 1);
  double maxerr = 0, maxv = 0;
  for (int j = 0; j < S::T3; ++j)
    for (int q = 0; q < R3; ++q) {
      const lc r = ref[j + q * S::T3];
      const double dr = (double)fabsl((long double)x[j][q].x - r.real());
      const double di = (double)fabsl((long double)x[j][q].y - r.imag());
      maxerr = std::max(maxerr, std::max(dr, di));
      maxv = std::max(maxv, (double)std::abs(r));
    }
  printf("RS L=%d (%d,%d,%d) dir=%d rel err %.3e\n", L, R1, R2, R3, DIR, maxerr / maxv);
  return maxerr / maxv;
}

int main(int argc, char** argv) {
  double worst = 0;
  if (argc > 1 && !strcmp(argv[1], "rs")) {
    for (double e : {run_rs<16, 18, 18, PS_FWD>(), run_rs<16, 18, 18, PS_INV>(),
                     run_rs<16, 9, 8, PS_FWD>(), run_rs<16, 16, 18, PS_INV>(),
                     run_rs<16, 18, 9, PS_INV>(), run_rs<16, 9, 9, PS_INV>(),
                     run_rs<16, 10, 12, PS_FWD>(), run_rs<16, 14, 15, PS_INV>(), run_rs<16, 20, 18, PS_FWD>(),
                     run_rs<16, 15, 20, PS_INV>(), run_rs<16, 12, 7, PS_FWD>(),
                     run_rs<16, 25, 18, PS_FWD>(), run_rs<16, 25, 25, PS_INV>(), run_rs<16, 25, 16, PS_INV>(),
                     // radices 21 = 3 x 7 and 24 = 3 x 8 (round 2)
                     run_rs<16, 21, 7, PS_FWD>(), run_rs<16, 21, 16, PS_INV>(), run_rs<16, 24, 18, PS_FWD>(),
                     run_rs<16, 25, 21, PS_INV>(), run_rs<16, 24, 24, PS_FWD>(), run_rs<16, 21, 21, PS_INV>()}) {
      if (e < 0) return 2;
      worst = e > worst ? e : worst;
    }
    printf("WORST %.3e\n", worst);
    return worst < 1e-13 ? 0 : 1;
  }
  for (int a = 1; a < argc; ++a) {
    int L = atoi(argv[a]);
    double e1 = run_case(L, PS_MODE_COL, 1, 2, false);
    double e0 = run_case(L, PS_MODE_COL, 1, 2, false, false, 3);
    if (e0 > e1) e1 = e0;
    double e2 = run_case(L, PS_MODE_ROW, 1, 0, true);
    double e3 = run_case(L, PS_MODE_ROW, 2, 0, false);
    double e4 = run_case(L, PS_MODE_ROW, 1, 0, true, true);   // radix-18/16 plans
    for (double e : {e1, e2, e3, e4}) {
      if (e < 0) return 2;
      worst = e > worst ? e : worst;
    }
  }
  printf("WORST %.3e\n", worst);
  return worst < 1e-13 ? 0 : 1;
}
