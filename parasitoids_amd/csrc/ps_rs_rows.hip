// Instantiations and launchers of the register-resident row kernels (fft_rs_kernels.h), one
// translation unit so that they compile in parallel with the rest of the library.
#include "rs_cfg.h"
#include "rs_launch.h"
#include <algorithm>
#include <cstdlib>

bool rs_lookup(int L, int* r2, int* r3) {
#define X(A, B) if (L == 16 * A * B) { *r2 = A; *r3 = B; return true; }
  PS_RS_SIZES(X)
#undef X
  return false;
}

int rs_next_size(int n) {
  int best = 0;
#define X(A, B) if (16 * A * B >= n && (best == 0 || 16 * A * B < best)) best = 16 * A * B;
  PS_RS_SIZES(X)
#undef X
  return best;
}

bool rs_info(int r2, int r3, RsInfo* out) {
#define X(A, B)                                                                       \
  if (r2 == A && r3 == B) {                                                           \
    using C = RsCfg<A, B>;                                                            \
    *out = RsInfo{C::NP, C::S::NTHR, C::LDS, C::LDSC1, C::LDSC, C::CHAIN, RsPLds<16, A, B>::fits};            \
    return true;                                                                      \
  }
  PS_RS_SIZES(X)
#undef X
  return false;
}

int rs_rows_set_attrs() {
#define X(A, B)                                                                                              \
  {                                                                                                          \
    using C = RsCfg<A, B>;                                                                                   \
    constexpr int np = C::NP;                                                                                \
    using Z = RsPLds<16, A, B>;                                                                              \
    if constexpr (Z::fits) {                                                                                 \
      auto kp = k_row_inv_rsp<16, A, B>;                                                                     \
      if (hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z::bytes) != hipSuccess) return -1; \
    }                                                                                                        \
    using Z2 = Rs2Lds<16, A, B>;                                                                             \
    if constexpr (Z2::ok) {                                                                                  \
      auto k2 = k_row_inv_rs2<16, A, B>;                                                                     \
      if (Z2::bytes > 48 * 1024 &&                                                                           \
          hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z2::bytes) != hipSuccess) return -1; \
    }                                                                                                        \
    if (C::LDSC1 > 48 * 1024) {                                                                              \
      auto kr = k_row_inv_fold<16, A, B>;                                                                    \
      if (hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDSC1) != hipSuccess) return -1; \
    }                                                                                                        \
    if (C::LDS > 48 * 1024) {                                                                                \
      auto ki = k_row_inv_rs<16, A, B, np>;                                                                  \
      auto kf = k_row_fwd_rs<16, A, B, np>;                                                                  \
      if (hipFuncSetAttribute((const void*)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS) != hipSuccess) return -1; \
      if (hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS) != hipSuccess) return -1; \
    }                                                                                                        \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}

static int device_cus();
static int grid_x(int npairs, int np, int tstride) {
  int gx = (npairs + np - 1) / np;
  if (tstride) {   // column-major side: the workgroups sharing a 128-byte line sit on one XCD
    const int m = 8 * (np >= 4 ? 1 : 4 / np);
    gx = (gx + m - 1) / m * m;
  }
  return gx;
}

int rs_launch_row_fwd(int r2, int r3, const RowFwdArgs& a, int npairs, int batch, hipStream_t st) {
#define X(A, B)                                                                                              \
  if (r2 == A && r3 == B) {                                                                                  \
    using C = RsCfg<A, B>;                                                                                   \
    constexpr int np = C::NP;                                                                                \
    /* small transforms pack 2-4 row pairs into a 12-wave workgroup; a single field with few live rows   \
       (the 1201-row torus of the R = 400 fold child on 1920 points: 151 workgroups for 256 CUs) then     \
       leaves CUs idle -- half the pairs per workgroup there */                                             \
    constexpr int np2 = (np >= 2 && C::S::L <= 2560) ? np / 2 : np;                                          \
    if constexpr (np2 != np) {                                                                               \
      const int live = a.rmap.n1 + (a.P > a.rmap.lo2 ? a.P - a.rmap.lo2 : 0);                                \
      if ((((live + 1) / 2 + np - 1) / np) * batch < device_cus()) {                                         \
        auto k2 = k_row_fwd_rs<16, A, B, np2>;                                                               \
        using Y2 = RsInvLds<16, A, B>;                                                                       \
        constexpr size_t lds2 = Y2::bytes(np2);                                                              \
        hipLaunchKernelGGL(k2, dim3(grid_x(npairs, np2, a.tstride), batch), dim3(C::S::NTHR * np2), lds2, st, a); \
        return 1;                                                                                            \
      }                                                                                                      \
    }                                                                                                        \
    auto kern = k_row_fwd_rs<16, A, B, np>;                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid_x(npairs, np, a.tstride), batch), dim3(C::S::NTHR * np), C::LDS, st, a); \
    return 1;                                                                                                \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}

int rs_launch_row_fold(int r2, int r3, const RowFoldArgs& a, int units, hipStream_t st) {
#define X(A, B)                                                                                              \
  if (r2 == A && r3 == B) {                                                                                  \
    using C = RsCfg<A, B>;                                                                                   \
    auto kern = k_row_inv_fold<16, A, B>;                                                                    \
    hipLaunchKernelGGL(kern, dim3(units), dim3(C::S::NTHR), C::LDSC1, st, a);                                \
    return 1;                                                                                                \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}

// workgroups of the persistent kernel: one per CU
static int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

int rs_launch_row_inv(int r2, int r3, const RowInvArgs& a, int npairs, int batch, hipStream_t st) {
#define X(A, B)                                                                                              \
  if (r2 == A && r3 == B) {                                                                                  \
    using C = RsCfg<A, B>;                                                                                   \
    using Z = RsPLds<16, A, B>;                                                                              \
    using Z2 = Rs2Lds<16, A, B>;                                                                             \
    if constexpr (Z2::ok) {                                                                                  \
      if (a.persistent == 2 && a.tstride == 0) {   /* two roles in anti-phase, one workgroup per CU */      \
        const int units = npairs * batch;                                                                    \
        const int wgs = std::min(device_cus(), (units + 1) / 2);                                             \
        auto k2 = k_row_inv_rs2<16, A, B>;                                                                   \
        hipLaunchKernelGGL(k2, dim3(wgs), dim3(2 * C::S::NTHR), Z2::bytes, st, a, npairs, units);            \
        return 1;                                                                                            \
      }                                                                                                      \
    }                                                                                                        \
    if constexpr (Z::fits) {                                                                                 \
      if (a.persistent && a.tstride == 0) {                                                                  \
        const int units = npairs * batch;                                                                    \
        /* resident workgroups per CU: 8 wave slots at the kernel's ~210 registers, 160 KB of LDS */         \
        constexpr int W = C::S::NTHR / 64;                                                                   \
        constexpr int kw = 8 / W < 1 ? 1 : 8 / W, kl = (int)((size_t)160 * 1024 / Z::bytes);                 \
        constexpr int per_cu = kw < kl ? kw : kl;                                                            \
        const int slots = device_cus() * per_cu;                                                             \
        const int wgs = units < slots ? units : slots;                                                       \
        auto kp = k_row_inv_rsp<16, A, B>;                                                                   \
        hipLaunchKernelGGL(kp, dim3(wgs), dim3(C::S::NTHR), Z::bytes, st, a, npairs, units); \
        return 1;                                                                                            \
      }                                                                                                      \
    }                                                                                                        \
    constexpr int np = C::NP;                                                                                \
    auto kern = k_row_inv_rs<16, A, B, np>;                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid_x(npairs, np, a.tstride), batch), dim3(C::S::NTHR * np), C::LDS, st, a); \
    return 1;                                                                                                \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}
