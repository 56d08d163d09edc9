// Instantiations and launcher of the full-column passes (fft_colfull_kernels.h), a translation
// unit of their own (4 kernels x 65 sizes + the chained pass of the 50 sizes that have one).
#include "fft_colfull_kernels.h"
#include "rs_cfg.h"
#include "rs_launch.h"
#include <cstdlib>

// the chained day pass exists only for the sizes whose state column fits LDS; elsewhere the slot
// holds the single-day kernel (never launched as a chained pass) instead of an instantiation
// nobody calls
template <int A, int B>
static auto chained_kernel() {
  using C = RsCfg<A, B>;
  if constexpr (C::CHAIN) return k_colfull<16, A, B, true, 0>;
  else return k_colfull_day<16, A, B, C::CEX, false>;
}

int rs_colfull_set_attrs() {
#define X(A, B)                                                                                              \
  {                                                                                                          \
    using C = RsCfg<A, B>;                                                                                   \
    if (C::LDSC > 48 * 1024 || C::LDSD > 48 * 1024) {                                                        \
      const void* kc[6] = {(const void*)k_colfull_day<16, A, B, C::CEX, false>, (const void*)k_colfull<16, A, B, false, 1>, \
                           (const void*)k_colfull<16, A, B, false, 2>, (const void*)k_colfull<16, A, B, false, 3>, \
                           (const void*)chained_kernel<A, B>(), (const void*)k_colfull_day<16, A, B, C::CEX, C::ALT>}; \
      for (const void* kk : kc)                                                                              \
        if (hipFuncSetAttribute(kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDSC > C::LDSD ? C::LDSC : C::LDSD)) != hipSuccess) return -1; \
    }                                                                                                        \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}

bool rs_colfull_alt_ok(int r2, int r3) {
#define X(A, B) if (r2 == A && r3 == B) return RsCfg<A, B>::ALT;
  PS_RS_SIZES(X)
#undef X
  return false;
}

int rs_launch_colfull(int r2, int r3, const ColFullArgs& a, int lines8, int batch, hipStream_t st) {
  // see k_colfull: columns are handed out in lines of 8, lines round-robin over the 8 XCDs
  const dim3 grid((unsigned)(lines8 * 64), batch);
#define X(A, B)                                                                                        \
  if (r2 == A && r3 == B) {                                                                            \
    using C = RsCfg<A, B>;                                                                             \
    /* single passes take the state straight from HBM (162 registers, 255 us per day at 5184); only a  \
       group of chained days parks it in LDS (128 + 64 prefetch registers: 192 us per day for eight) */   \
    auto k0 = k_colfull_day<16, A, B, C::CEX, false>;                                                  \
    auto k0a = k_colfull_day<16, A, B, C::CEX, C::ALT>;   /* pending re-transform of a flagged day */   \
    if (a.alt_pred && (!C::ALT || a.mode != 0 || a.nd != 1)) return 0;                                 \
    auto k1 = k_colfull<16, A, B, false, 1>;                                                           \
    auto k2 = k_colfull<16, A, B, false, 2>;                                                           \
    auto k3 = k_colfull<16, A, B, false, 3>;                                                           \
    auto kc = chained_kernel<A, B>();                                                                  \
    const bool chained = a.mode == 0 && a.nd > 1;                                                      \
    if (chained && !C::CHAIN) return 0;                                                                \
    auto kern = chained ? kc : (a.mode == 0 ? (a.alt_pred ? k0a : k0) : a.mode == 1 ? k1 : a.mode == 2 ? k2 : k3);          \
    hipLaunchKernelGGL(kern, grid, dim3(C::S::NTHR), chained ? C::LDSC : (a.mode == 0 ? C::LDSD : C::LDSC1), st, a); \
    return 1;                                                                                          \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}
