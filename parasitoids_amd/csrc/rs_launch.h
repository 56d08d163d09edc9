// Launchers of the register-resident kernel families (fft_rs_kernels.h, fft_colfull_kernels.h).
// The 65 sizes x (2 row kernels + 5 full-column kernels) are instantiated in translation units
// of their own (ps_rs_rows.hip, ps_colfull.hip), compiled in parallel with ps_solver.hip; the
// solver sees only these functions.
#pragma once
#include <hip/hip_runtime.h>
#include "fft_kernels.h"

struct ColFullArgs;   // fft_colfull_kernels.h

// per size L = 16 * r2 * r3: row pairs per workgroup of the row kernels, their LDS, whether the
// full-column pass can chain days (state column parked in LDS); false when (r2, r3) is not served
struct RsInfo {
  int np, nthr;
  size_t lds_rows, lds_col1, lds_colc;
  bool chain;
  bool rsp;   // k_row_inv_rsp exists for this size (its staged rows fit in LDS)
};
bool rs_lookup(int L, int* r2, int* r3);
int rs_next_size(int n);   // smallest served size >= n, or 0
bool rs_info(int r2, int r3, RsInfo* out);
int rs_rows_set_attrs();       // hipFuncSetAttribute(MaxDynamicSharedMemorySize) for every row kernel
int rs_colfull_set_attrs();    // ... and every full-column kernel
// launches; return 0 when (r2, r3) is not a served size
int rs_launch_row_fwd(int r2, int r3, const RowFwdArgs& a, int npairs, int batch, hipStream_t st);
int rs_launch_row_inv(int r2, int r3, const RowInvArgs& a, int npairs, int batch, hipStream_t st);
// mode 0 (nd >= 1 days), 1, 2, 3 as in fft_colfull_kernels.h; lines8 = 128-byte lines per XCD
int rs_launch_colfull(int r2, int r3, const ColFullArgs& a, int lines8, int batch, hipStream_t st);
// inverse row pass + fold onto the reference torus (PS_MODE_FOLD; k_row_inv_fold), one workgroup per unit
struct RowFoldArgs;
int rs_launch_row_fold(int r2, int r3, const RowFoldArgs& a, int units, hipStream_t st);
// the single-day pass of this size can take the state column from a pending re-transform (ColFullArgs::alt_src)
bool rs_colfull_alt_ok(int r2, int r3);
// two-role chained pass (k_colfull_dual; mode 0, nd >= 2): sizes for which rs_dual_ok
bool rs_dual_ok(int r2, int r3);
int rs_coldual_set_attrs();
int rs_launch_coldual(int r2, int r3, const ColFullArgs& a, int lines8, int batch, hipStream_t st);
