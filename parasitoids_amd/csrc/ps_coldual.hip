// Instantiations and launcher of the two-role chained full-column pass (k_colfull_dual,
// fft_colfull_kernels.h): the sizes whose single-role chained pass is alone on its CU (state
// column + exchange buffer beyond half of the LDS) and whose doubled workgroup still fits 12
// waves; a translation unit of its own so that it compiles next to the others.
#include "fft_colfull_kernels.h"
#include "rs_cfg.h"
#include "rs_launch.h"

template <int A, int B>
struct DualCfg {
  using D = RsDual<16, A, B>;
  using C = RsCfg<A, B>;
  // worth it only where one chained workgroup fills the CU's LDS (two of them do not fit)
  static constexpr bool on = D::ok && C::CHAIN && C::LDSC > (size_t)80 * 1024;
};

bool rs_dual_ok(int r2, int r3) {
#define X(A, B) \
  if (r2 == A && r3 == B) return DualCfg<A, B>::on;
  PS_RS_SIZES(X)
#undef X
  return false;
}

int rs_coldual_set_attrs() {
#define X(A, B)                                                                                         \
  if constexpr (DualCfg<A, B>::on) {                                                                    \
    using DD = RsDual<16, A, B>;                                                                        \
    auto kern = k_colfull_dual<16, A, B>;                                                               \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DD::bytes) != hipSuccess) \
      return -1;                                                                                        \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}

int rs_launch_coldual(int r2, int r3, const ColFullArgs& a, int lines8, int batch, hipStream_t st) {
  const dim3 grid((unsigned)(lines8 * 64), batch);
#define X(A, B)                                                                                         \
  if (r2 == A && r3 == B) {                                                                             \
    if constexpr (DualCfg<A, B>::on) {                                                                  \
      using DD = RsDual<16, A, B>;                                                                      \
      auto kern = k_colfull_dual<16, A, B>;                                                             \
      hipLaunchKernelGGL(kern, grid, dim3(2 * DD::S::NTHR), DD::bytes, st, a);                          \
      return 1;                                                                                         \
    }                                                                                                   \
    return 0;                                                                                           \
  }
  PS_RS_SIZES(X)
#undef X
  return 0;
}
