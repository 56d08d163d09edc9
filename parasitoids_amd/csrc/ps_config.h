// Tuning and A/B knobs of the library, frozen per handle.
//
// The reference has ONE switch (globalvars.py:5).  This library has about fifty knobs that exist
// for measurements and for the A/B legs of the bit-identity tests.  They used to be environment
// variables looked up wherever they were needed -- per run, per day, per launch.  Now the
// environment is read in exactly one place, ps_config_from_env(), when a handle is created
// (ps_solver_create / ps_model_create); from then on the handle's own copy is all the library
// looks at, and ps_solver_set_option / ps_model_set_option (include/parasitoid_hip.h) are the way to
// change a knob of a live handle.  Helper solvers of an auto-mode front copy their parent's table.
//
// PS_OPTION_TABLE(X): X(key, field, default, kind, when)
//   kind  FLAG  environment: present and not "0" -> 1        INT / NUM  environment: the value
//   when  C     takes effect at creation only (set_option refuses it on a live handle)
//         R     read by the next run / launch
#pragma once
#include <cstdlib>
#include <cstring>

#define PS_OPTION_TABLE(X)                                                                          \
  /* register-resident kernels, pipelines (creation) */                                             \
  X("PS_NO_RS", no_rs, 0, FLAG, C)                                                                  \
  X("PS_NO_RS_FWD", no_rs_fwd, 0, FLAG, C)                                                          \
  X("PS_NO_BIG_RADIX", no_big_radix, 0, FLAG, C)                                                    \
  X("PS_NO_TPIPE", no_tpipe, 0, FLAG, C)                                                            \
  X("PS_NO_FOLD_TPIPE", no_fold_tpipe, 0, FLAG, C)                                                  \
  X("PS_COL_SINGLE_MAX", col_single_max, 1200, INT, C)                                              \
  X("PS_COL_L1", col_l1, 0, INT, C)                                                                 \
  X("PS_CHUNK_DAYS", chunk_days, 0, INT, C)                                                         \
  X("PS_FAST_SIZE", fast_size, 0, INT, C)                                                           \
  X("PS_RS_AREA", rs_area, 0.0, NUM, C)                                                             \
  /* launch shapes */                                                                               \
  X("PS_COL_WSH", col_wsh, -1, INT, R)                                                              \
  X("PS_COL_THREADS", col_threads, 256, INT, R)                                                     \
  X("PS_FUSED_WSH", fused_wsh, -1, INT, R)                                                          \
  X("PS_FUSED_THREADS", fused_threads, 0, INT, R)                                                   \
  X("PS_FUSED_G", fused_g, 2, INT, R)                                                               \
  X("PS_MULTI_WSH", multi_wsh, -1, INT, R)                                                          \
  X("PS_DUAL_WSH", dual_wsh, -1, INT, R)                                                            \
  X("PS_FUSED_DAYS", fused_days, 8, INT, R)                                                         \
  X("PS_DIRECT_DAYS", direct_days, 4, INT, R)                                                       \
  X("PS_DIRECT_MAX_TERMS", direct_max_terms, 8, INT, R)                                             \
  X("PS_TPIPE_DAYS", tpipe_days, 32, INT, R)                                                        \
  X("PS_DUAL_MIN_DAYS", dual_min_days, 6, INT, R)                                                   \
  /* routes (A/B legs of the bit-identity tests) */                                                 \
  X("PS_TINV", tinv, 0, INT, R)                                                                     \
  X("PS_RSP", rsp, -1, INT, R)                                                                      \
  X("PS_ROW2", row2, -1, INT, R)                                                                    \
  X("PS_TPIPE", tpipe, -1, INT, R)                                                                  \
  X("PS_NO_PAIR_ROWS", no_pair_rows, 0, FLAG, R)                                                    \
  X("PS_TPIPE_SPLIT", tpipe_split, 0, INT, R)                                                       \
  X("PS_NO_CONJ", no_conj, 0, FLAG, R)                                                              \
  X("PS_NO_DUAL", no_dual, 0, FLAG, R)                                                              \
  X("PS_NO_DIRECT", no_direct, 0, FLAG, R)                                                          \
  X("PS_NO_ROW_BATCH", no_row_batch, 0, FLAG, R)                                                    \
  X("PS_NO_PAD_QUIET", no_pad_quiet, 0, FLAG, R)                                                    \
  X("PS_NO_TAIL_SPLIT", no_tail_split, 0, FLAG, R)                                                  \
  X("PS_NO_DEFER_REFFT", no_defer_refft, 0, FLAG, R)                                                \
  X("PS_NO_FILTER_CACHE", no_filter_cache, 0, FLAG, R)                                              \
  X("PS_NO_FOLD_FUSE", no_fold_fuse, 0, FLAG, R)                                                    \
  X("PS_NO_FOLD_ALT", no_fold_alt, 0, FLAG, R)                                                      \
  X("PS_NO_FOLD_ROWS", no_fold_rows, 0, FLAG, R)                                                    \
  /* chain control */                                                                               \
  X("PS_NO_SPECULATION", no_speculation, 0, FLAG, R)                                                \
  X("PS_SPEC_DEPTH", spec_depth, 2, INT, R)                                                         \
  X("PS_NO_FLAG_HISTORY", no_flag_history, 0, FLAG, R)                                              \
  X("PS_NO_WINDOW_HINT", no_window_hint, 0, FLAG, R)                                                \
  X("PS_FIRST_WINDOW", first_window, -1, INT, R)                                                    \
  X("PS_KT_SPLIT", kt_split, -1, INT, R)                                                            \
  X("PS_NO_LAZY_KT", no_lazy_kt, 0, FLAG, R)                                                        \
  X("PS_NO_WIDE", no_wide, 0, FLAG, R)                                                              \
  X("PS_NO_ROUTE_HISTORY", no_route_history, 0, FLAG, R)                                            \
  X("PS_WIDE_MIN_N", wide_min_n, 1500, INT, R)                                                      \
  X("PS_AUTO_WINDOW", auto_window, 4, INT, R)                                                       \
  /* prob_mass (ps_model handles) */                                                                \
  X("PS_PM_SEG", pm_seg, 8, INT, R)                                                                 \
  X("PS_PM_SYNC", pm_sync, 0, FLAG, R)                                                              \
  X("PS_PM_NO_UNROLL", pm_no_unroll, 0, FLAG, R)

struct ps_config {
#define PS_CFG_FIELD_FLAG int
#define PS_CFG_FIELD_INT int
#define PS_CFG_FIELD_NUM double
#define X(key, field, dflt, kind, when) PS_CFG_FIELD_##kind field = dflt;
  PS_OPTION_TABLE(X)
#undef X
};

// The one place the library reads its environment.
inline void ps_config_from_env(ps_config* c) {
  *c = ps_config();
#define PS_CFG_ENV_FLAG(field) c->field = (e[0] == '\0' || std::strcmp(e, "0") != 0) ? 1 : 0
#define PS_CFG_ENV_INT(field) c->field = std::atoi(e)
#define PS_CFG_ENV_NUM(field) c->field = std::atof(e)
#define X(key, field, dflt, kind, when) \
  if (const char* e = std::getenv(key)) { PS_CFG_ENV_##kind(field); }
  PS_OPTION_TABLE(X)
#undef X
}

// 0 ok, -1 unknown key, -2 creation-time key on a live handle (live = false: any key)
inline int ps_config_set(ps_config* c, const char* key, double value, bool live) {
  if (!key) return -1;
#define PS_CFG_SET_FLAG(field) c->field = value != 0.0 ? 1 : 0
#define PS_CFG_SET_INT(field) c->field = (int)value
#define PS_CFG_SET_NUM(field) c->field = value
#define PS_CFG_WHEN_C true
#define PS_CFG_WHEN_R false
#define X(k, field, dflt, kind, when)         \
  if (std::strcmp(key, k) == 0) {             \
    if (live && PS_CFG_WHEN_##when) return -2; \
    PS_CFG_SET_##kind(field);                 \
    return 0;                                 \
  }
  PS_OPTION_TABLE(X)
#undef X
  return -1;
}

inline int ps_config_get(const ps_config* c, const char* key, double* value) {
  if (!key || !value) return -1;
#define X(k, field, dflt, kind, when) \
  if (std::strcmp(key, k) == 0) {     \
    *value = (double)c->field;        \
    return 0;                         \
  }
  PS_OPTION_TABLE(X)
#undef X
  return -1;
}
