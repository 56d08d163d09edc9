// Micro-benchmark of the register-resident three-stage transform (fft_rs.h) with NO global
// memory in the loop: what one CU can do per column-day of the chained full-column pass, and
// what changes it -- more resident waves, independent workgroups, lockstep roles, the exchange
// (LDS) and butterfly (VALU) halves on their own.  Answers "what bounds the on-CU time" for
// DESIGN 4.1d.  Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Iparasitoids_amd/csrc \
//         -o scripts/microbench/fftcore scripts/microbench/fftcore.hip
// Run: scripts/microbench/fftcore [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "fft_rs.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

struct Tw {
  const cplx* lo;
  const cplx* hi;
  int shift;
};


// Half-buffer exchange: the buffer holds L/2 words (+ padding); each real/imaginary pass of an
// exchange is two half passes (writers: threads below / above T/2; readers: q below / above R/2).
template <class S, int R1, int R2, int R3, int DIR>
__device__ __forceinline__ void tail_half(cplx* x, double* ex, const int j, const cplx w2, const cplx w3) {
  static_assert(R2 % 2 == 0 && R3 % 2 == 0 && S::T1 % 2 == 0 && S::T2 % 32 == 0, "half split");
  constexpr int H1 = 17 * (S::T1 / 2);              // words written by the lower half of the stage-1 threads
  constexpr int H1R = (R2 / 2) * S::X1_RS;          // first word read by q >= R2/2
  constexpr int H2 = ((S::T2 / 2) * R2) + ((S::T2 / 2) * R2 >> 4);
  constexpr int H2R = (R3 / 2) * S::X2_RS;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (j < S::T1 && (j < S::T1 / 2) == (h == 0)) {
        if (part) rs_put<R1, 1>(ex, S::x1_w(j) - h * H1, 1, x);
        else rs_put<R1, 0>(ex, S::x1_w(j) - h * H1, 1, x);
      }
      __syncthreads();
      if (j < S::T2) {
#pragma unroll
        for (int q = 0; q < R2 / 2; ++q) {
          const double v = ex[S::x_r(j) + (q + h * (R2 / 2)) * S::X1_RS - h * H1R];
          if (part) x[q + h * (R2 / 2)].y = v;
          else x[q + h * (R2 / 2)].x = v;
        }
      }
      if (!(part == 1 && h == 1)) __syncthreads();
    }
  }
  if (j < S::T2) rs_stage<R2, DIR>(x, w2, true);
  __syncthreads();
#pragma unroll
  for (int part = 0; part < 2; ++part) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (j < S::T2 && (j < S::T2 / 2) == (h == 0)) {
        if (part) rs_put<R2, 1>(ex, S::x2_w(j) - h * H2, 17, x);
        else rs_put<R2, 0>(ex, S::x2_w(j) - h * H2, 17, x);
      }
      __syncthreads();
      if (j < S::T3) {
#pragma unroll
        for (int q = 0; q < R3 / 2; ++q) {
          const double v = ex[S::x_r(j) + (q + h * (R3 / 2)) * S::X2_RS - h * H2R];
          if (part) x[q + h * (R3 / 2)].y = v;
          else x[q + h * (R3 / 2)].x = v;
        }
      }
      if (!(part == 1 && h == 1)) __syncthreads();
    }
  }
  if (j < S::T3) rs_stage<R3, DIR>(x, w3, true);
}

// what a variant leaves out: 0 = everything, 1 = no butterflies (LDS + barriers only),
// 2 = no exchange (butterflies only; barriers kept), 3 = no exchange and no barriers
template <class S, int R1, int R2, int R3, int DIR, int CUT>
__device__ __forceinline__ void tail(cplx* x, double* ex, const int j, const cplx w2, const cplx w3) {
  if constexpr (CUT == 0) {
    rs_tail<S, R1, R2, R3, DIR>(x, ex, j, w2, w3);
  } else if constexpr (CUT == 4) {
    __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
    __syncthreads(); __syncthreads(); __syncthreads();
  } else if constexpr (CUT == 5) {
    rs_tail_c<S, R1, R2, R3, DIR>(x, reinterpret_cast<cplx*>(ex), j, w2, w3);
  } else if constexpr (CUT == 6) {
    tail_half<S, R1, R2, R3, DIR>(x, ex, j, w2, w3);
  } else if constexpr (CUT == 1) {
    if (j < S::T1) rs_put<R1, 0>(ex, S::x1_w(j), 1, x);
    __syncthreads();
    if (j < S::T2) rs_get<R2, 0>(ex, S::x_r(j), S::X1_RS, x);
    __syncthreads();
    if (j < S::T1) rs_put<R1, 1>(ex, S::x1_w(j), 1, x);
    __syncthreads();
    if (j < S::T2) rs_get<R2, 1>(ex, S::x_r(j), S::X1_RS, x);
    __syncthreads();
    if (j < S::T2) rs_put<R2, 0>(ex, S::x2_w(j), 17, x);
    __syncthreads();
    if (j < S::T3) rs_get<R3, 0>(ex, S::x_r(j), S::X2_RS, x);
    __syncthreads();
    if (j < S::T2) rs_put<R2, 1>(ex, S::x2_w(j), 17, x);
    __syncthreads();
    if (j < S::T3) rs_get<R3, 1>(ex, S::x_r(j), S::X2_RS, x);
  } else {
    if (CUT == 2) { __syncthreads(); __syncthreads(); __syncthreads(); }
    if (j < S::T2) rs_stage<R2, DIR>(x, w2, true);
    if (CUT == 2) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    if (j < S::T3) rs_stage<R3, DIR>(x, w3, true);
  }
}

// One column per workgroup of NTHR threads: per iteration a forward transform, a product with a
// "state" (in LDS when STATE, as the chained pass keeps it; else a register constant), the
// natural -> input order relayout, an inverse transform.  = the chained day step without HBM.
template <int R1, int R2, int R3, bool STATE, int CUT, int MINW, bool HOLD = true>
__global__ void __launch_bounds__((Rs<R1, R2, R3>::NTHR), MINW) mb_seq(Tw tw, const cplx* in, double* out, int iters, int lds_pad) {
  using S = Rs<R1, R2, R3>;
  double* ex = reinterpret_cast<double*>(lds_raw);
  cplx* sst = reinterpret_cast<cplx*>(ex + ((S::XWORDS + 15) & ~15) + 64);
  const int j0 = threadIdx.x;
  const cplx w2c = tw_lookup(tw.lo, tw.hi, tw.shift, S::tw2(j0));
  const cplx w3c = tw_lookup(tw.lo, tw.hi, tw.shift, j0 < S::T3 ? S::tw3(j0) : 0);
  cplx xn[HOLD ? R1 : 1];
#pragma unroll
  for (int q = 0; q < (HOLD ? R1 : 1); ++q) xn[q] = in[(j0 + q * S::T1) % 4096];
  if (STATE && j0 < S::T3) {
#pragma unroll
    for (int q = 0; q < R3; ++q) sst[j0 + q * S::T3] = make_double2(1.0, 1e-9 * q);
  }
  double acc = 0.0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    cplx x[S::RMAX];
    int j = threadIdx.x;
    asm volatile("" : "+v"(j));
    cplx w2 = w2c, w3 = w3c;
    asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
    if (j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) x[q] = HOLD ? xn[q] : make_double2(xn[0].x + (double)(q + it), xn[0].y - (double)q);
      if (CUT != 1) bfly<R1, PS_FWD>(x);
    }
    tail<S, R1, R2, R3, PS_FWD, CUT>(x, ex, j, w2, w3);
    if (j < S::T3) {
      if (STATE) {
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          x[q] = cmul(sst[j + q * S::T3], x[q]);
          sst[j + q * S::T3] = x[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < R3; ++q) x[q] = cmul(make_double2(1.0, 1e-9), x[q]);
      }
    }
    __syncthreads();
    if (STATE) {
      if (j < S::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q] = sst[j + q * S::T1];
      }
    } else {   // real / imaginary relayout through the exchange buffer, as k_colfull without CHAIN
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
      __syncthreads();
    }
    asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
    if (j < S::T1 && CUT != 1) bfly<R1, PS_INV>(x);
    tail<S, R1, R2, R3, PS_INV, CUT>(x, ex, j, w2, w3);
    if (j < S::T3) {
#pragma unroll
      for (int q = 0; q < R3; ++q) acc += x[q].x * 1e-300 + x[q].y * 1e-300;
    }
    __syncthreads();
  }
  if (acc == 12345.678) out[blockIdx.x] = acc;
}

// Two transforms per workgroup in lockstep (2 x NTHR threads): role 0 runs inverse transforms,
// role 1 forward transforms, each on its own exchange buffer; the barriers are shared.  One
// iteration = one transform per role = the work of one chained day step.
template <int R1, int R2, int R3>
__global__ void __launch_bounds__((2 * Rs<R1, R2, R3>::NTHR)) mb_dual(Tw tw, const cplx* in, double* out, int iters) {
  using S = Rs<R1, R2, R3>;
  const int role = threadIdx.x / S::NTHR;
  double* ex = reinterpret_cast<double*>(lds_raw) + role * (((S::XWORDS + 15) & ~15) + 64);
  const int j0 = threadIdx.x - role * S::NTHR;
  const cplx w2c = tw_lookup(tw.lo, tw.hi, tw.shift, S::tw2(j0));
  const cplx w3c = tw_lookup(tw.lo, tw.hi, tw.shift, j0 < S::T3 ? S::tw3(j0) : 0);
  const cplx x0 = in[j0 % 4096];
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    cplx x[S::RMAX];
    int j = j0;
    asm volatile("" : "+v"(j));
    cplx w2 = w2c, w3 = w3c;
    asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(w3.x), "+v"(w3.y));
    if (j < S::T1) {
#pragma unroll
      for (int q = 0; q < R1; ++q) x[q] = make_double2(x0.x + (double)(q + it), x0.y - (double)q);
    }
    if (role == 0) {
      if (j < S::T1) bfly<R1, PS_INV>(x);
      rs_tail<S, R1, R2, R3, PS_INV>(x, ex, j, w2, w3);
    } else {
      if (j < S::T1) bfly<R1, PS_FWD>(x);
      rs_tail<S, R1, R2, R3, PS_FWD>(x, ex, j, w2, w3);
    }
    if (j < S::T3) {
#pragma unroll
      for (int q = 0; q < R3; ++q) acc += x[q].x * 1e-300 + x[q].y * 1e-300;
    }
    __syncthreads();
  }
  if (acc == 12345.678) out[blockIdx.x] = acc;
}

template <class K, class... A>
static double time_kernel(K kern, dim3 grid, dim3 block, size_t lds, int reps, A... args) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, grid, block, lds, 0, args...);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, block, lds, 0, args...);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3 / reps;
}

template <int R2, int R3>
static void run_size(int iters) {
  using S = Rs<16, R2, R3>;
  constexpr int L = S::L;
  const int shift = 7, B = 1 << shift, nhi = (L + B - 1) / B;
  std::vector<cplx> tab(B + nhi);
  const long double twopi = 6.283185307179586476925286766559L;
  for (int t = 0; t < B; ++t) tab[t] = make_double2((double)cosl(twopi * t / L), (double)-sinl(twopi * t / L));
  for (int u = 0; u < nhi; ++u) tab[B + u] = make_double2((double)cosl(twopi * ((long long)u * B % L) / L), (double)-sinl(twopi * ((long long)u * B % L) / L));
  cplx* dtab;
  CK(hipMalloc(&dtab, tab.size() * sizeof(cplx)));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(cplx), hipMemcpyHostToDevice));
  std::vector<cplx> in(4096);
  for (int i = 0; i < 4096; ++i) in[i] = make_double2(std::sin(0.37 * i), std::cos(0.11 * i));
  cplx* din;
  CK(hipMalloc(&din, in.size() * sizeof(cplx)));
  CK(hipMemcpy(din, in.data(), in.size() * sizeof(cplx), hipMemcpyHostToDevice));
  double* dout;
  CK(hipMalloc(&dout, 1 << 20));
  Tw tw{dtab, dtab + B, shift};
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t exb = (size_t)(((S::XWORDS + 15) & ~15) + 64) * 8;
  const size_t stb = (size_t)L * 16;
  printf("L = %d (16 x %d x %d), NTHR %d, CUs %d, exchange %zu B, state %zu B, iters %d\n", L, R2, R3, S::NTHR, ncu, exb, stb, iters);
  auto rep = [&](const char* name, double us, int wg_per_cu, int ffts_per_iter) {
    printf("  %-58s %9.1f us/launch  %7.3f us per day-step per CU slot (%d WG/CU requested)  %6.3f us/FFT/CU\n", name, us, us / iters,
           wg_per_cu, us / iters / ffts_per_iter / wg_per_cu);
  };
  // 1. the chained day step, state in LDS, one workgroup per CU
  rep("seq, state in LDS, 1 WG/CU", time_kernel(mb_seq<16, R2, R3, true, 0, 1>, dim3(ncu), dim3(S::NTHR), exb + stb, 3, tw, din, dout, iters, 0), 1, 2);
  rep("  same, no butterflies (LDS + barriers)", time_kernel(mb_seq<16, R2, R3, true, 1, 1>, dim3(ncu), dim3(S::NTHR), exb + stb, 3, tw, din, dout, iters, 0), 1, 2);
  rep("  same, no exchange (butterflies + barriers)", time_kernel(mb_seq<16, R2, R3, true, 2, 1>, dim3(ncu), dim3(S::NTHR), exb + stb, 3, tw, din, dout, iters, 0), 1, 2);
  rep("  same, barriers only (no LDS ops, no butterflies)", time_kernel(mb_seq<16, R2, R3, true, 4, 1>, dim3(ncu), dim3(S::NTHR), exb + stb, 3, tw, din, dout, iters, 0), 1, 2);
  rep("  same, half-buffer exchange (16 phases per transform)", time_kernel(mb_seq<16, R2, R3, true, 6, 1>, dim3(ncu), dim3(S::NTHR), exb + stb, 3, tw, din, dout, iters, 0), 1, 2);
  if (2 * exb <= 160 * 1024) rep("seq, no LDS state, complex exchange words (3 barriers per transform)", time_kernel(mb_seq<16, R2, R3, false, 5, 1, false>, dim3(ncu), dim3(S::NTHR), 2 * exb, 3, tw, din, dout, iters, 0), 1, 2);
  rep("  same, butterflies only", time_kernel(mb_seq<16, R2, R3, true, 3, 1>, dim3(ncu), dim3(S::NTHR), exb + stb, 3, tw, din, dout, iters, 0), 1, 2);
  // 2. no LDS state (relayout through the exchange buffer): 1, 2, 3 workgroups per CU
  for (int w = 1; w <= 3; ++w) {
    char nm[96];
    snprintf(nm, sizeof nm, "seq, no LDS state, grid = %d x CUs", w);
    rep(nm, time_kernel(mb_seq<16, R2, R3, false, 0, 1, false>, dim3(ncu * w), dim3(S::NTHR), exb, 3, tw, din, dout, iters, 0), w, 2);
  }
  for (int w = 1; w <= 3; ++w) {
    char nm[96];
    snprintf(nm, sizeof nm, "seq, no LDS state, <= 168 VGPRs (3 waves/SIMD), grid = %d x CUs", w);
    rep(nm, time_kernel(mb_seq<16, R2, R3, false, 0, 3, false>, dim3(ncu * w), dim3(S::NTHR), exb, 3, tw, din, dout, iters, 0), w, 2);
  }
  for (int w = 1; w <= 3; ++w) {
    char nm[96];
    snprintf(nm, sizeof nm, "seq, no LDS state, <= 128 VGPRs (4 waves/SIMD), grid = %d x CUs", w);
    rep(nm, time_kernel(mb_seq<16, R2, R3, false, 0, 4, false>, dim3(ncu * w), dim3(S::NTHR), exb, 3, tw, din, dout, iters, 0), w, 2);
  }
  // 3. two transforms per workgroup in lockstep (12 waves)
  rep("dual roles (inv | fwd) in lockstep, 1 WG/CU", time_kernel(mb_dual<16, R2, R3>, dim3(ncu), dim3(2 * S::NTHR), 2 * exb, 3, tw, din, dout, iters), 1, 2);
  CK(hipFree(dtab));
  CK(hipFree(din));
  CK(hipFree(dout));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  run_size<18, 18>(iters);   // 5184
  run_size<16, 16>(iters);   // 4096: four full waves
  run_size<18, 20>(iters);   // 5760
  return 0;
}
