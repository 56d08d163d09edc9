// In-LDS mixed-radix FFT program (fp64 complex), shared by every FFT pass kernel.
//
// One 1-D transform of length L is a list of radix stages executed in place on
// an LDS-resident array: forward = decimation in frequency (natural order in,
// digit-reversed positions out), inverse = the exact stage-by-stage inverse
// (decimation in time, digit-reversed in, natural out, unnormalised).  The
// digit reversal is undone for free at the LDS<->global boundary (`pos` table),
// so HBM always holds natural order.
//
// Radices 2,3,4,5,7,8,9 have register butterflies; any other prime factor r
// (<= 1024) runs as a wave-cooperative O(r) butterfly, which is what lets the
// solver transform on the reference's *exact* pad size P = N + K//2
// (CalcSol.py:20-21), e.g. 573 = 3*191 or 1121 = 19*59.
//
// Everything here is plain C++ that compiles for host and device: the host build
// (tests/host/fft_emul.cpp) emulates the thread grid to check the indexing
// against a long-double DFT without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PS_HD __host__ __device__ __forceinline__
#else
#define PS_HD inline
struct double2 {
  double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
#endif

typedef double2 cplx;

#define PS_MAX_STAGES 14
#define PS_FWD 0
#define PS_INV 1
#define PS_MAX_GENERIC_RADIX 1024
#define PS_GEN_KPL (PS_MAX_GENERIC_RADIX / 64)

struct FftProg {
  int32_t L, ns;
  int32_t radix[PS_MAX_STAGES];
  int32_t n[PS_MAX_STAGES];  // block length entering stage s
  int32_t m[PS_MAX_STAGES];  // n / radix
  int32_t step[PS_MAX_STAGES];  // L / n: twiddle stride of the stage
  // row-mode LDS layout: logical index i lives at (i / Lb) * Lbp + i % Lb.
  // stages [0,sa) have m % Lb == 0 ("leading"), stages [sa,ns) have n <= Lb.
  int32_t sa, La, Lb, Lbp;
  // multiply-shift reciprocals (ps_magic) of the divisors used by bf_decode
  uint32_t mg_m[PS_MAX_STAGES], mg_mh[PS_MAX_STAGES], mg_nbf[PS_MAX_STAGES], mg_Lb, mg_La;
  int32_t tw_shift, n_lo, n_hi;  // w_L^t = tw_hi[t >> shift] * tw_lo[t & mask]
  // per-stage table w_r^t (t < r) of a wave-cooperative prime radix: offset into the
  // block that follows tw_hi (the three tables are contiguous: lo | hi | gen), or -1
  int32_t n_gen, gen_off[PS_MAX_STAGES];
  const cplx* tw_lo;             // device
  const cplx* tw_hi;             // device (= tw_lo + n_lo)
  const uint32_t* pos;           // device: digit-reversed logical position of k
  const uint32_t* pos_phys;      // device: row-mode physical position of k
};

// ---------------------------------------------------------------- complex ops
PS_HD cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
PS_HD cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
PS_HD cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
PS_HD cplx cmulc(cplx a, cplx b) {  // a * conj(b)
  return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
PS_HD cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
PS_HD cplx cscale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }
// multiply by -i (forward) or +i (inverse)
template <int DIR>
PS_HD cplx cmul_mi(cplx a) {
  return DIR == PS_FWD ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
}

// ------------------------------------------------------------ register DFTs
// All butterflies compute y_k = sum_q x_q w^(qk), w = exp(-+2 pi i / R).

template <int DIR>
PS_HD void bfly2(cplx* x) {
  cplx a = x[0], b = x[1];
  x[0] = cadd(a, b);
  x[1] = csub(a, b);
}

template <int DIR>
PS_HD void bfly4(cplx* x) {
  cplx a = cadd(x[0], x[2]), b = csub(x[0], x[2]);
  cplx c = cadd(x[1], x[3]), d = cmul_mi<DIR>(csub(x[1], x[3]));
  x[0] = cadd(a, c);
  x[2] = csub(a, c);
  x[1] = cadd(b, d);
  x[3] = csub(b, d);
}

// odd prime R with the symmetric form: a_q = x_q + x_{R-q}, b_q = x_q - x_{R-q}
template <int R>
struct OddTab;
template <>
struct OddTab<3> {
  static PS_HD double c(int t) {
    const double v[3] = {1.0, -0.5, -0.5};
    return v[t];
  }
  static PS_HD double s(int t) {
    const double v[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
    return v[t];
  }
};
template <>
struct OddTab<5> {
  static PS_HD double c(int t) {
    const double v[5] = {1.0, 0.30901699437494742410, -0.80901699437494742410,
                         -0.80901699437494742410, 0.30901699437494742410};
    return v[t];
  }
  static PS_HD double s(int t) {
    const double v[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917,
                         -0.58778525229247312917, -0.95105651629515357212};
    return v[t];
  }
};
template <>
struct OddTab<7> {
  static PS_HD double c(int t) {
    const double v[7] = {1.0,
                         0.62348980185873353053,
                         -0.22252093395631440429,
                         -0.90096886790241912624,
                         -0.90096886790241912624,
                         -0.22252093395631440429,
                         0.62348980185873353053};
    return v[t];
  }
  static PS_HD double s(int t) {
    const double v[7] = {0.0,
                         0.78183148246802980871,
                         0.97492791218182360702,
                         0.43388373911755812048,
                         -0.43388373911755812048,
                         -0.97492791218182360702,
                         -0.78183148246802980871};
    return v[t];
  }
};

template <int R, int DIR>
PS_HD void bfly_odd(cplx* x) {
  constexpr int Hh = (R - 1) / 2;
  cplx a[Hh], b[Hh];
#pragma unroll
  for (int q = 1; q <= Hh; ++q) {
    a[q - 1] = cadd(x[q], x[R - q]);
    b[q - 1] = csub(x[q], x[R - q]);
  }
  cplx x0 = x[0];
  cplx y0 = x0;
#pragma unroll
  for (int q = 0; q < Hh; ++q) y0 = cadd(y0, a[q]);
  x[0] = y0;
#pragma unroll
  for (int k = 1; k <= Hh; ++k) {
    cplx c = x0, d = make_double2(0.0, 0.0);
#pragma unroll
    for (int q = 1; q <= Hh; ++q) {
      const int t = (q * k) % R;
      const double ct = OddTab<R>::c(t), st = OddTab<R>::s(t);
      c.x += a[q - 1].x * ct;
      c.y += a[q - 1].y * ct;
      d.x += b[q - 1].x * st;
      d.y += b[q - 1].y * st;
    }
    cplx id = cmul_mi<DIR>(d);  // -i d (fwd), +i d (inv)
    x[k] = cadd(c, id);
    x[R - k] = csub(c, id);
  }
}

template <int DIR>
PS_HD void bfly8(cplx* x) {
  // even / odd DFT-4 then radix-2 combine with w8^k
  cplx e[4] = {x[0], x[2], x[4], x[6]};
  cplx o[4] = {x[1], x[3], x[5], x[7]};
  bfly4<DIR>(e);
  bfly4<DIR>(o);
  const double h = 0.70710678118654752440;
  // w8^1 = (h, -+h), w8^2 = -+i, w8^3 = (-h, -+h)
  cplx t1 = DIR == PS_FWD ? make_double2((o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h)
                          : make_double2((o[1].x - o[1].y) * h, (o[1].y + o[1].x) * h);
  cplx t2 = cmul_mi<DIR>(o[2]);
  cplx t3 = DIR == PS_FWD ? make_double2((o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h)
                          : make_double2(-(o[3].x + o[3].y) * h, (o[3].x - o[3].y) * h);
  x[0] = cadd(e[0], o[0]);
  x[4] = csub(e[0], o[0]);
  x[1] = cadd(e[1], t1);
  x[5] = csub(e[1], t1);
  x[2] = cadd(e[2], t2);
  x[6] = csub(e[2], t2);
  x[3] = cadd(e[3], t3);
  x[7] = csub(e[3], t3);
}

template <int DIR>
PS_HD void bfly9(cplx* x) {
  // q = q0 + 3 q1 ; y_{k1 + 3 k0} = sum_q0 w9^(q0 k1) (sum_q1 x w3^(q1 k1)) w3^(q0 k0)
  const double c1 = 0.76604444311897803520, s1 = 0.64278760968653932632;   // 2pi/9
  const double c2 = 0.17364817766693034885, s2 = 0.98480775301220805937;   // 4pi/9
  const double c4 = -0.93969262078590838405, s4 = 0.34202014332566873304;  // 8pi/9
  cplx u[3][3];
#pragma unroll
  for (int q0 = 0; q0 < 3; ++q0) {
    cplx t[3] = {x[q0], x[q0 + 3], x[q0 + 6]};
    bfly_odd<3, DIR>(t);
    u[q0][0] = t[0];
    u[q0][1] = t[1];
    u[q0][2] = t[2];
  }
  const double sg = DIR == PS_FWD ? -1.0 : 1.0;
  u[1][1] = cmul(u[1][1], make_double2(c1, sg * s1));
  u[1][2] = cmul(u[1][2], make_double2(c2, sg * s2));
  u[2][1] = cmul(u[2][1], make_double2(c2, sg * s2));
  u[2][2] = cmul(u[2][2], make_double2(c4, sg * s4));
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    cplx t[3] = {u[0][k1], u[1][k1], u[2][k1]};
    bfly_odd<3, DIR>(t);
    x[k1] = t[0];
    x[k1 + 3] = t[1];
    x[k1 + 6] = t[2];
  }
}

template <int DIR>
PS_HD void bfly16(cplx* x) {
  // q = q0 + 4 q1 ; y_{k1 + 4 k0} = sum_q0 w16^(q0 k1) (sum_q1 x w4^(q1 k1)) w4^(q0 k0)
  const double c = 0.92387953251128673848, sn = 0.38268343236508978178, h = 0.70710678118654752440;
  const double sg = DIR == PS_FWD ? -1.0 : 1.0;
  cplx u[4][4];
#pragma unroll
  for (int q0 = 0; q0 < 4; ++q0) {
    cplx t[4] = {x[q0], x[q0 + 4], x[q0 + 8], x[q0 + 12]};
    bfly4<DIR>(t);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) u[q0][k1] = t[k1];
  }
  u[1][1] = cmul(u[1][1], make_double2(c, sg * sn));    // w^1
  u[1][2] = cmul(u[1][2], make_double2(h, sg * h));     // w^2
  u[1][3] = cmul(u[1][3], make_double2(sn, sg * c));    // w^3
  u[2][1] = cmul(u[2][1], make_double2(h, sg * h));     // w^2
  u[2][2] = cmul_mi<DIR>(u[2][2]);                      // w^4
  u[2][3] = cmul(u[2][3], make_double2(-h, sg * h));    // w^6
  u[3][1] = cmul(u[3][1], make_double2(sn, sg * c));    // w^3
  u[3][2] = cmul(u[3][2], make_double2(-h, sg * h));    // w^6
  u[3][3] = cmul(u[3][3], make_double2(-c, -sg * sn));  // w^9
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    cplx t[4] = {u[0][k1], u[1][k1], u[2][k1], u[3][k1]};
    bfly4<DIR>(t);
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) x[k1 + 4 * k0] = t[k0];
  }
}

template <int DIR>
PS_HD void bfly18(cplx* x) {
  // even / odd DFT-9 then radix-2 combine with w18^k
  cplx e[9], o[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    e[i] = x[2 * i];
    o[i] = x[2 * i + 1];
  }
  bfly9<DIR>(e);
  bfly9<DIR>(o);
  const double cs[9] = {1.0, 0.93969262078590838405, 0.76604444311897803520, 0.5,
                        0.17364817766693034885, -0.17364817766693034885, -0.5,
                        -0.76604444311897803520, -0.93969262078590838405};
  const double ss[9] = {0.0, 0.34202014332566873304, 0.64278760968653932632,
                        0.86602540378443864676, 0.98480775301220805937, 0.98480775301220805937,
                        0.86602540378443864676, 0.64278760968653932632, 0.34202014332566873304};
  const double sg = DIR == PS_FWD ? -1.0 : 1.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const cplx t = k == 0 ? o[0] : cmul(o[k], make_double2(cs[k], sg * ss[k]));
    x[k] = cadd(e[k], t);
    x[k + 9] = csub(e[k], t);
  }
}

// Prime-factor (Good-Thomas) butterfly for R = A * B with gcd(A, B) = 1: no internal
// twiddles, only the index maps  n = (B n1 + A n2) mod R  (input)  and  k = k1 (mod A),
// k = k2 (mod B)  (output, CRT) -- all compile-time constants after unrolling.
constexpr int ps_modinv(int a, int m) {
  for (int v = 1; v < m; ++v)
    if ((a * v) % m == 1) return v;
  return 1;
}
template <int R, int DIR>
PS_HD void bfly(cplx* x);

template <int A, int B, int DIR>
PS_HD void bfly_pfa(cplx* x) {
  constexpr int N = A * B;
  constexpr int ca = B * ps_modinv(B % A, A);   // == 1 mod A, == 0 mod B
  constexpr int cb = A * ps_modinv(A % B, B);   // == 0 mod A, == 1 mod B
  cplx y[A][B];
#pragma unroll
  for (int n1 = 0; n1 < A; ++n1) {
    cplx t[B];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) t[n2] = x[(B * n1 + A * n2) % N];
    bfly<B, DIR>(t);
#pragma unroll
    for (int k2 = 0; k2 < B; ++k2) y[n1][k2] = t[k2];
  }
#pragma unroll
  for (int k2 = 0; k2 < B; ++k2) {
    cplx t[A];
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) t[n1] = y[n1][k2];
    bfly<A, DIR>(t);
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) x[(k1 * ca + k2 * cb) % N] = t[k1];
  }
}

// radix 25 = 5 x 5 (not coprime: Cooley-Tukey with the twiddles w25^(q0 k1))
template <int DIR>
PS_HD void bfly25(cplx* x) {
  // q = q0 + 5 q1 ; y_{k1 + 5 k0} = sum_q0 w25^(q0 k1) (sum_q1 x w5^(q1 k1)) w5^(q0 k0)
  const double c[17] = {1.0, 0.96858316112863108, 0.87630668004386358, 0.72896862742141155,
                        0.53582679497899666, 0.30901699437494742, 0.062790519529313374,
                        -0.18738131458572463, -0.42577929156507272, -0.63742398974868975,
                        -0.80901699437494742, -0.92977648588825146, -0.99211470131447788,
                        -0.99211470131447788, -0.92977648588825146, -0.80901699437494742,
                        -0.63742398974868975};
  const double sn[17] = {0.0, 0.24868988716485479, 0.48175367410171532, 0.68454710592868873,
                         0.84432792550201508, 0.95105651629515357, 0.99802672842827156,
                         0.98228725072868872, 0.90482705246601958, 0.77051324277578925,
                         0.58778525229247313, 0.36812455268467797, 0.12533323356430426,
                         -0.12533323356430426, -0.36812455268467797, -0.58778525229247313,
                         -0.77051324277578925};
  const double sg = DIR == PS_FWD ? -1.0 : 1.0;
  cplx u[5][5];
#pragma unroll
  for (int q0 = 0; q0 < 5; ++q0) {
    cplx t[5] = {x[q0], x[q0 + 5], x[q0 + 10], x[q0 + 15], x[q0 + 20]};
    bfly_odd<5, DIR>(t);
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
      const int e = q0 * k1;   // <= 16
      u[q0][k1] = e == 0 ? t[k1] : cmul(t[k1], make_double2(c[e], sg * sn[e]));
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    cplx t[5] = {u[0][k1], u[1][k1], u[2][k1], u[3][k1], u[4][k1]};
    bfly_odd<5, DIR>(t);
#pragma unroll
    for (int k0 = 0; k0 < 5; ++k0) x[k1 + 5 * k0] = t[k0];
  }
}

template <int R, int DIR>
PS_HD void bfly(cplx* x) {
  if (R == 2) bfly2<DIR>(x);
  else if (R == 3) bfly_odd<3, DIR>(x);
  else if (R == 4) bfly4<DIR>(x);
  else if (R == 5) bfly_odd<5, DIR>(x);
  else if (R == 7) bfly_odd<7, DIR>(x);
  else if (R == 8) bfly8<DIR>(x);
  else if (R == 9) bfly9<DIR>(x);
  else if (R == 10) bfly_pfa<2, 5, DIR>(x);
  else if (R == 12) bfly_pfa<3, 4, DIR>(x);
  else if (R == 14) bfly_pfa<2, 7, DIR>(x);
  else if (R == 15) bfly_pfa<3, 5, DIR>(x);
  else if (R == 16) bfly16<DIR>(x);
  else if (R == 18) bfly_pfa<2, 9, DIR>(x);   // 2 x 9 coprime: no internal twiddles (32 operations fewer than the even/odd split of bfly18)
  else if (R == 20) bfly_pfa<4, 5, DIR>(x);
  else if (R == 21) bfly_pfa<3, 7, DIR>(x);
  else if (R == 24) bfly_pfa<3, 8, DIR>(x);
  else if (R == 25) bfly25<DIR>(x);
}

// ------------------------------------------------------------------ twiddles
// forward twiddle w_L^t (t < L) from the two-level tables (LDS copies).
PS_HD cplx tw_lookup(const cplx* tlo, const cplx* thi, int shift, int t) {
  return cmul(thi[t >> shift], tlo[t & ((1 << shift) - 1)]);
}

// ------------------------------------------------------------- fast division
// q = x / d for 0 <= x < 2^32 / d via one 32x32->hi multiply; d == 1 has magic 0.
PS_HD uint32_t ps_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }
PS_HD int ps_div(int x, uint32_t mg) {
#if defined(__HIP_DEVICE_COMPILE__)
  return mg ? (int)__umulhi((uint32_t)x, mg) : x;
#else
  return mg ? (int)(((uint64_t)(uint32_t)x * mg) >> 32) : x;
#endif
}

// ------------------------------------------------------ butterfly addressing
// mode COL: LDS holds a [L][W] tile (W = 1 << wsh columns, the batch), FFT along
//           the rows; lanes run along the batch.  bs != 0: row pitch 1 << bs instead of W.
// mode ROW: LDS holds nb padded rows; lanes run along whichever digit is
//           contiguous (leading stages) or along the La padded sub-rows
//           (trailing stages), so both are bank-conflict free.
#define PS_MODE_COL 0
#define PS_MODE_ROW 1

struct BfAddr {
  int addr0, qstride, j;
};

PS_HD BfAddr bf_decode(const FftProg& P, int s, int mode, int item, int wsh, int bs) {
  const int n = P.n[s], m = P.m[s];
  BfAddr a;
  if (mode == PS_MODE_COL) {
    const int b = item & ((1 << wsh) - 1);
    const int bf = item >> wsh;
    const int blk = ps_div(bf, P.mg_m[s]);
    const int j = bf - blk * m;
    const int psh = bs ? bs : wsh;  // COL mode: bs != 0 gives the tile's row pitch (log2)
    a.addr0 = ((blk * n + j) << psh) + b;
    a.qstride = m << psh;
    a.j = j;
  } else {
    const int nbf = m * (P.L / n);
    const int b = ps_div(item, P.mg_nbf[s]);
    const int bf = item - b * nbf;
    if (s < P.sa) {
      const int mh = ps_div(m, P.mg_Lb);
      const int t = ps_div(bf, P.mg_Lb);
      const int j_lo = bf - t * P.Lb;
      const int blk = ps_div(t, P.mg_mh[s]);
      const int j_hi = t - blk * mh;
      a.addr0 = b * bs + (blk * ps_div(n, P.mg_Lb) + j_hi) * P.Lbp + j_lo;
      a.qstride = mh * P.Lbp;
      a.j = j_hi * P.Lb + j_lo;
    } else {
      const int inner = ps_div(bf, P.mg_La);
      const int rho = bf - inner * P.La;
      const int sub = ps_div(inner, P.mg_m[s]);
      const int j = inner - sub * m;
      a.addr0 = b * bs + rho * P.Lbp + sub * n + j;
      a.qstride = m;
      a.j = j;
    }
  }
  return a;
}

// w[k] = w1^k, k = 1..R-1, by a multiplication tree of depth <= 5
template <int R>
PS_HD void tw_powers(cplx* w, cplx w1) {
  w[1] = w1;
  if (R > 2) w[2] = cmul(w[1], w[1]);
  if (R > 3) w[3] = cmul(w[2], w[1]);
  if (R > 4) w[4] = cmul(w[2], w[2]);
  if (R > 5) w[5] = cmul(w[4], w[1]);
  if (R > 6) w[6] = cmul(w[3], w[3]);
  if (R > 7) w[7] = cmul(w[4], w[3]);
  if (R > 8) w[8] = cmul(w[4], w[4]);
  if (R > 9) w[9] = cmul(w[8], w[1]);
  if (R > 10) w[10] = cmul(w[5], w[5]);
  if (R > 11) w[11] = cmul(w[8], w[3]);
  if (R > 12) w[12] = cmul(w[6], w[6]);
  if (R > 13) w[13] = cmul(w[8], w[5]);
  if (R > 14) w[14] = cmul(w[7], w[7]);
  if (R > 15) w[15] = cmul(w[8], w[7]);
  if (R > 16) w[16] = cmul(w[8], w[8]);
  if (R > 17) w[17] = cmul(w[16], w[1]);
}

// ------------------------------------------------------- register-radix stage
template <int R, int DIR>
PS_HD void run_stage_r(cplx* data, const cplx* tlo, const cplx* thi, const FftProg& P,
                       int s, int mode, int nb, int wsh, int bs, int tid, int nthr) {
  const int nitems = P.m[s] * P.step[s] * nb;
  const int step = P.step[s];
  const bool has_tw = P.m[s] > 1;
  for (int item = tid; item < nitems; item += nthr) {
    const BfAddr a = bf_decode(P, s, mode, item, wsh, bs);
    cplx x[R];
#pragma unroll
    for (int q = 0; q < R; ++q) x[q] = data[a.addr0 + q * a.qstride];
    cplx w[R];  // w[k] = w_n^(j k)
    if (has_tw && a.j != 0) {
      tw_powers<R>(w, tw_lookup(tlo, thi, P.tw_shift, a.j * step));
      if (DIR == PS_INV) {
#pragma unroll
        for (int q = 1; q < R; ++q) x[q] = cmulc(x[q], w[q]);
      }
    }
    bfly<R, DIR>(x);
    if (DIR == PS_FWD && has_tw && a.j != 0) {
#pragma unroll
      for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) data[a.addr0 + q * a.qstride] = x[q];
  }
}

// ------------------------------------------------ wave-cooperative prime stage
// Odd prime radix r (11 <= r <= 1024).  A wave owns G = max(1, 64 / r) butterflies at a
// time; lane computes outputs k = sub-lane + 64 u from the symmetric form
//   y_k = x_0 + sum_{q=1}^{(r-1)/2} [ (x_q + x_{r-q}) cos(2 pi q k / r)
//                                     -+ i (x_q - x_{r-q}) sin(2 pi q k / r) ],
// i.e. real multiplies against the stage's w_r^t table in LDS.  All lanes finish reading
// the r inputs (uniform q loop) before any lane stores, so the update is in place.  The
// inverse applies its input twiddles element-wise beforehand (gen_pretwiddle + barrier).
// Split in compute/store calls so the host emulation can order "all lanes compute"
// before "all lanes store".
struct GenAcc {
  cplx acc[PS_GEN_KPL];
  int addr0, qstride, k0, active;
};

// inverse only: x_q *= conj(w_n^(j q)) for every element of the stage (once per element)
template <int DIR>
PS_HD void gen_pretwiddle(cplx* data, const cplx* tlo, const cplx* thi, const FftProg& P, int s,
                          int mode, int nb, int wsh, int bs, int tid, int nthr) {
  if (DIR != PS_INV || P.m[s] <= 1) return;
  const int r = P.radix[s];
  const int nbutter = P.m[s] * P.step[s] * nb;
  const int step = P.step[s];
  const int tot = nbutter * (r - 1);
  const uint32_t mg = ps_magic((uint32_t)(r - 1));
  for (int it = tid; it < tot; it += nthr) {
    const int bid = ps_div(it, mg);
    const int q = it - bid * (r - 1) + 1;
    const BfAddr a = bf_decode(P, s, mode, bid, wsh, bs);
    if (a.j == 0) continue;
    const int e = a.addr0 + q * a.qstride;
    data[e] = cmulc(data[e], tw_lookup(tlo, thi, P.tw_shift, a.j * q * step));
  }
}

template <int DIR, int KPL>
PS_HD void gen_compute_k(GenAcc& g, const cplx* data, const cplx* tlo, const cplx* thi,
                         const FftProg& P, int s, int mode, int nb, int wsh, int bs,
                         int group0, int lane) {
  const int r = P.radix[s];
  const int nbutter = P.m[s] * P.step[s] * nb;
  const int G = r < 64 ? 64 / r : 1;
  int sub = 0, k0 = lane;
  if (r < 64) {
    sub = lane / r;
    k0 = lane - sub * r;
  }
  const int bid = group0 + sub;
  g.active = (sub < G) && (bid < nbutter);
  g.k0 = k0;
  if (!g.active) return;
  const BfAddr a = bf_decode(P, s, mode, bid, wsh, bs);
  g.addr0 = a.addr0;
  g.qstride = a.qstride;
  const cplx* wr = thi + P.n_hi + P.gen_off[s];  // w_r^t = (cos, -sin)(2 pi t / r)
  const cplx x0 = data[a.addr0];
  cplx acc[KPL];
  int idx[KPL], kk[KPL];
#pragma unroll
  for (int u = 0; u < KPL; ++u) {
    acc[u] = x0;
    idx[u] = 0;
    kk[u] = k0 + 64 * u;
    if (kk[u] >= r) kk[u] = 0;  // parked lanes recompute output 0; never stored
  }
  const int h = (r - 1) / 2;
  for (int q = 1; q <= h; ++q) {
    const cplx xa = data[a.addr0 + q * a.qstride];
    const cplx xb = data[a.addr0 + (r - q) * a.qstride];
    const cplx sa = cadd(xa, xb), sb = csub(xa, xb);
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      idx[u] += kk[u];
      if (idx[u] >= r) idx[u] -= r;
      const cplx w = wr[idx[u]];
      const double c = w.x, sn = -w.y;  // cos, sin of 2 pi q k / r
      if (DIR == PS_FWD) {
        acc[u].x += sa.x * c + sb.y * sn;
        acc[u].y += sa.y * c - sb.x * sn;
      } else {
        acc[u].x += sa.x * c - sb.y * sn;
        acc[u].y += sa.y * c + sb.x * sn;
      }
    }
  }
  if (DIR == PS_FWD && a.j != 0) {
    const int step = P.step[s];
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const int k = k0 + 64 * u;
      if (k < r && k != 0) acc[u] = cmul(acc[u], tw_lookup(tlo, thi, P.tw_shift, a.j * k * step));
    }
  }
#pragma unroll
  for (int u = 0; u < KPL; ++u) g.acc[u] = acc[u];
}

template <int DIR>
PS_HD void gen_compute(GenAcc& g, const cplx* data, const cplx* tlo, const cplx* thi,
                       const FftProg& P, int s, int mode, int nb, int wsh, int bs,
                       int group0, int lane) {
  const int kpl = (P.radix[s] + 63) / 64;
  if (kpl <= 1) gen_compute_k<DIR, 1>(g, data, tlo, thi, P, s, mode, nb, wsh, bs, group0, lane);
  else if (kpl <= 2) gen_compute_k<DIR, 2>(g, data, tlo, thi, P, s, mode, nb, wsh, bs, group0, lane);
  else if (kpl <= 4) gen_compute_k<DIR, 4>(g, data, tlo, thi, P, s, mode, nb, wsh, bs, group0, lane);
  else if (kpl <= 8) gen_compute_k<DIR, 8>(g, data, tlo, thi, P, s, mode, nb, wsh, bs, group0, lane);
  else gen_compute_k<DIR, 16>(g, data, tlo, thi, P, s, mode, nb, wsh, bs, group0, lane);
}

PS_HD void gen_store(const GenAcc& g, cplx* data, int r) {
  if (!g.active) return;
  const int kpl = (r + 63) / 64;
  for (int u = 0; u < kpl; ++u) {
    const int k = g.k0 + 64 * u;
    if (k < r) data[g.addr0 + k * g.qstride] = g.acc[u];
  }
}

template <int DIR>
PS_HD void run_stage_generic(cplx* data, const cplx* tlo, const cplx* thi, const FftProg& P,
                             int s, int mode, int nb, int wsh, int bs, int tid, int nthr) {
  const int r = P.radix[s];
  const int nbutter = P.m[s] * P.step[s] * nb;
  const int G = r < 64 ? 64 / r : 1;
  const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
  for (int g0 = wave * G; g0 < nbutter; g0 += nwaves * G) {
    GenAcc g;
    gen_compute<DIR>(g, data, tlo, thi, P, s, mode, nb, wsh, bs, g0, lane);
    gen_store(g, data, r);
  }
}

template <int DIR>
PS_HD void run_stage(cplx* data, const cplx* tlo, const cplx* thi, const FftProg& P, int s,
                     int mode, int nb, int wsh, int bs, int tid, int nthr) {
  switch (P.radix[s]) {
    case 2: run_stage_r<2, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 3: run_stage_r<3, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 4: run_stage_r<4, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 5: run_stage_r<5, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 7: run_stage_r<7, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 8: run_stage_r<8, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 9: run_stage_r<9, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 16: run_stage_r<16, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    case 18: run_stage_r<18, DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
    default: run_stage_generic<DIR>(data, tlo, thi, P, s, mode, nb, wsh, bs, tid, nthr); break;
  }
}

// row-mode physical position of logical index i
PS_HD int row_phys(const FftProg& P, int i) {
  if (P.Lb == 1) return i;
  const int hi = ps_div(i, P.mg_Lb);
  return hi * P.Lbp + (i - hi * P.Lb);
}
// LDS elements one row-mode transform occupies
PS_HD int row_pitch(const FftProg& P) { return P.La * P.Lbp; }

#if defined(__HIPCC__)
// ------------------------------------------------------------ wave reductions
// Sum / maximum over the 64 lanes of a wave through DPP (row-internal permutes, then the two
// row broadcasts of wave64): VALU-speed moves instead of the ds_bpermute round trips __shfl_down
// compiles to -- 18 dependent LDS-pipe operations per double reduced, 0.8 us per row pair in the
// inverse row pass (measured by leaving them out).  The result is valid in LANE 63 only.  Fixed
// combination order: the same on every run and in every kernel that uses it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double ps_dpp_f64(double v, double ident) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(ident), lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ps_wave_sum(double v) {       // -> lane 63
  v += ps_dpp_f64<0xB1, 0xf>(v, 0.0);    // quad_perm [1,0,3,2]
  v += ps_dpp_f64<0x4E, 0xf>(v, 0.0);    // quad_perm [2,3,0,1]
  v += ps_dpp_f64<0x141, 0xf>(v, 0.0);   // row_half_mirror
  v += ps_dpp_f64<0x140, 0xf>(v, 0.0);   // row_mirror: every lane of a row holds the row's sum
  v += ps_dpp_f64<0x142, 0xa>(v, 0.0);   // row_bcast15 into rows 1 and 3
  v += ps_dpp_f64<0x143, 0xc>(v, 0.0);   // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double ps_wave_max0(double v) {      // values >= 0; -> lane 63
  v = fmax(v, ps_dpp_f64<0xB1, 0xf>(v, 0.0));
  v = fmax(v, ps_dpp_f64<0x4E, 0xf>(v, 0.0));
  v = fmax(v, ps_dpp_f64<0x141, 0xf>(v, 0.0));
  v = fmax(v, ps_dpp_f64<0x140, 0xf>(v, 0.0));
  v = fmax(v, ps_dpp_f64<0x142, 0xa>(v, 0.0));
  v = fmax(v, ps_dpp_f64<0x143, 0xc>(v, 0.0));
  return v;
}
__device__ __forceinline__ int ps_wave_sum_i32(int v) {         // -> lane 63
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
#endif
