// libparasitoid_hip.so -- day-chain FFT convolution solver (C ABI in
// include/parasitoid_hip.h).  Replaces cuda_lib.CudaSolve (cuda_lib.py:16-221) and
// the CPU chain of CalcSol.py:140-325 with fp64 HIP kernels for gfx950.
#include <deque>

#include "chain_kernels.h"
#include "fft_kernels.h"
#include "fft_colfull_kernels.h"   // ColFullArgs (the kernels are instantiated in ps_colfull.hip)
#include "rs_launch.h"
#include "ps_common.h"
#include "ps_config.h"

thread_local std::string ps_tls_error;

extern "C" const char* ps_last_error(void) { return ps_tls_error.c_str(); }
extern "C" int ps_version(void) { return 100; }

extern "C" int ps_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return ps_fail(PS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

extern "C" int ps_device_info(int device, char* name, int n, int* cus, int64_t* hbm_bytes) {
  hipDeviceProp_t p;
  PS_HIP(hipGetDeviceProperties(&p, device));
  if (name && n > 0) snprintf(name, n, "%s (%s)", p.name, p.gcnArchName);
  if (cus) *cus = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return PS_OK;
}

int ps_use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return ps_fail(PS_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
  if (device < 0 || device >= n) return ps_fail(PS_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, n);
  PS_HIP(hipSetDevice(device));
  return PS_OK;
}

static const int kMaxLds = 160 * 1024;

// ---------------------------------------------------------------- caching device allocator
#include <map>
#include <mutex>
#include <unordered_map>
namespace {
struct DevPool {
  std::mutex mu;
  std::map<int, std::multimap<size_t, void*>> free_;   // device -> size -> block
  std::unordered_map<void*, std::pair<int, size_t>> live;   // block -> (device, size)
  size_t cached = 0;
  size_t limit() {
    static const size_t gb = getenv("PS_POOL_GB") ? (size_t)atoi(getenv("PS_POOL_GB")) : 16;
    return gb << 30;
  }
  void flush_locked() {
    for (auto& dv : free_)
      for (auto& kv : dv.second) (void)hipFree(kv.second);
    free_.clear();
    cached = 0;
  }
};
DevPool& pool() {
  static DevPool* p = new DevPool();   // never destroyed: blocks may outlive static teardown order
  return *p;
}
size_t round_size(size_t b) {
  const size_t g = b >= (1u << 20) ? (size_t)1 << 20 : 256;
  return (b + g - 1) / g * g;
}
}  // namespace

// PS_POISON=1 (test aid): every block handed out is filled with 0xFF bytes -- NaNs as doubles,
// -1 as integers -- so that a pass reading rows "known to be zero" that nobody wrote (the
// passes skip dead rows on both sides) shows up deterministically instead of only when the
// caching allocator happens to recycle a dirty block.
static hipError_t poison(void* p, size_t n) {
  static const bool on = getenv("PS_POISON") != nullptr;
  if (!on) return hipSuccess;
  hipError_t e = hipMemset(p, 0xFF, n);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e;
}

hipError_t ps_dev_malloc(void** out, size_t bytes) {
  DevPool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t want = round_size(bytes ? bytes : 1);
  std::lock_guard<std::mutex> lk(P.mu);
  auto& fr = P.free_[dev];
  auto it = fr.lower_bound(want);
  if (it != fr.end() && it->first <= want + want / 4 + (1u << 20)) {
    *out = it->second;
    P.live[*out] = {dev, it->first};
    P.cached -= it->first;
    const size_t got = it->first;
    fr.erase(it);
    return poison(*out, got);
  }
  hipError_t e = hipMalloc(out, want);
  if (e != hipSuccess) {   // give the cached blocks back and try once more
    (void)hipGetLastError();
    P.flush_locked();
    e = hipMalloc(out, want);
  }
  if (e == hipSuccess) {
    P.live[*out] = {dev, want};
    e = poison(*out, want);
  }
  return e;
}

void ps_dev_free(void* p) {
  if (!p) return;
  DevPool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.live.find(p);
  if (it == P.live.end()) {
    (void)hipFree(p);
    return;
  }
  const int dev = it->second.first;
  const size_t sz = it->second.second;
  P.live.erase(it);
  if (P.cached + sz > P.limit()) {
    (void)hipFree(p);
    return;
  }
  P.free_[dev].emplace(sz, p);
  P.cached += sz;
}

void ps_dev_quiesce() { (void)hipDeviceSynchronize(); }
static const int kPredGrid = 512;  // grid of the flag-conditional (usually empty) launches
// PS_MODE_AUTO: a day is "clean" when nothing above this lies outside the N x N domain of the
// fast torus.  Then the reference torus and the fast torus hold the same field up to that much
// (a convolution with a pmf cannot raise the maximum of the dust it moves), so the fast chain IS
// the exact-torus chain to <= 4 * days * kCleanEps.  FFT round-off in the pad is ~1e-18.
static const double kCleanEps = 1e-15;
#define PS_PROF_NCLS 16
enum { PS_PROF_ROW_FWD = 0, PS_PROF_COL_FWD_A = 1, PS_PROF_COL_FWD_B = 2, PS_PROF_COL_INV_A = 3,
       PS_PROF_COL_INV_B = 4, PS_PROF_ROW_INV = 5, PS_PROF_REFFT = 6, PS_PROF_COL_INV_A2 = 7,
       PS_PROF_COL_INV_A4 = 8, PS_PROF_COL_INV_A8 = 9, PS_PROF_ROW_INV2 = 10, PS_PROF_ROW_INV4 = 11,
       PS_PROF_ROW_INV8 = 12,
       PS_PROF_COL_INV_AN = 13,   // any other number of chained days in one launch (days counted in prof_days)
       PS_PROF_ROW_INVN = 14,
       PS_PROF_COL_TAIL = 15 };   // the three launches for the columns taken out of a chained pass (conv_inv_multi)
#define PS_MAX_GROUP_DAYS 32       // most days one chained full-column pass / batched row pass takes

struct ColPass {
  DevPlan* plan;
  int n_outer, in_base_mul, in_stride, out_base_mul, out_stride, tw_mode;
  bool second = false;  // second sub-pass of a split column transform
};

struct ps_solver {
  int device = 0;
  hipStream_t stream = nullptr;
  int N = 0, M = 0, Pref = 0, Pf = 0, H = 0, ld = 0, mode = 0;
  DevPlan row_plan, col_plan1, col_plan2;
  bool split = false;
  // full-column pipeline (fft_colfull_kernels.h): spectra are column-major and a column
  // transform is ONE pass.  On for register-resident sizes in fast mode (and the fast-torus
  // front of auto mode); the tiled two-sub-pass column kernels serve every other case.
  bool tpipe = false;
  bool tpipe_ok = false;    // the solver CAN run the full-column pipeline (register-resident size, fast mode)
  // A/B knob PS_TINV=1: the full-column pass writes its output column-major too (contiguous)
  // and the inverse row pass does the transposition on its READ side.  Measured at 5184: day
  // pass 264 -> 237 us, chained 218 -> 186 us per day, but the row pass 145 -> 252 us (32-byte
  // pieces at a column's stride per lane): a net loss, so the column pass keeps its 16-byte
  // row-major stores.  It also shows what those stores cost the day pass: 27 us of 264.
  bool tinv = false;
  ps_config cfg;   // knobs of this handle: the environment as ps_solver_create found it, then ps_solver_set_option
  // The state's spectrum is built lazily from its spatial record (PS_REC_STATE), in the layout
  // of the pipeline the first consumer picks: ps_chain_run takes the tiled pipeline for compact
  // day kernels (multi-day fused passes with direct-sum kernels) and the full-column one
  // otherwise; the per-call API stays on the solver's default.
  bool spec_valid = false;
  bool row_big = false;  // row plan uses the radix-18/16 butterflies (512-thread workgroups)
  int num_cu = 256;
  // flag speculation in ps_chain_run: on until this solver has seen a boundary flag
  bool speculate = true;
  int fused_days = 8;   // days per fused column pass (1, 2, 4, 8)
  int spec_window = 1;
  hipEvent_t spec_ev[2] = {nullptr, nullptr};
  // second stream: the kernel transforms of the later days of a chunk run there, behind the
  // first windows of day passes (see chain_run); kt_from = first day whose transform is pending
  hipStream_t stream2 = nullptr;
  hipEvent_t kt_ev = nullptr, kt_ev0 = nullptr;
  int kt_from = -1;
  // auto mode, front solver: kernels from this day on are transformed only if the run gets there
  // (the previous run left the front at its first unclean day); -1 = nothing pending
  int kt_lazy_from = -1, kt_lazy_c0 = 0, kt_lazy_cn = 0, kt_lazy_direct = 0;
  unsigned long long* hflags = nullptr;   // pinned host copy of the pad maxima
  int hflags_n = 0;
  // ... and of the previous chain run's, copied behind it and read when the next run starts: a solver that
  // has seen a flag (speculate off) still chains the stretches of days that raised none last time
  unsigned long long* hhist = nullptr;
  int hhist_n = 0, hist_first = -1, hist_count = 0;
  hipEvent_t hist_ev = nullptr;
  int rs_r2 = 0, rs_r3 = 0;   // register-resident row kernels (fft_rs.h) for Pf = 16 * rs_r2 * rs_r3, or 0
  int L1 = 0, L2 = 0;
  DevBuf<cplx> tp_lo, tp_hi;
  int tp_shift = 0;
  std::vector<ColPass> fwd_passes, inv_passes;
  // spectra
  DevBuf<cplx> Ahat, Chat, T1, T2, Bhat;
  int chunk_days = 1;
  DevBuf<int> srange;          // [lo, hi] rows of the state record that hold anything (host-supplied states)
  bool srange_valid = false;
  // kernels (device COO + dense staging)
  DevBuf<int> krow, kcol;
  DevBuf<double> kval;
  std::vector<int64_t> koff;
  std::vector<int> kshape;
  int nk = 0, Kmax = 0;
  bool kernels_on_device = false;
  DevBuf<double> kdense;
  DevBuf<long long> dkoff;   // device copies of koff / kshape for the batched scatter
  DevBuf<int> dkshape;
  int bhat_first = -1, bhat_count = 0;  // which days' transforms Bhat currently holds
  // records
  std::vector<double*> recs[4];
  // stats
  DevBuf<double> rowsum;       // [nstat][N]
  DevBuf<long long> rowcnt;    // [nstat][N]
  DevBuf<long long> rowoff;    // [N]
  DevBuf<unsigned long long> padmax;
  DevBuf<DayStats> dstats;
  int nstat = 0;
  int last_renorm = 0;
  RowLive kt_live{0, {0, 0, 0, 0}, nullptr};  // live rows of the kernel transforms (non-split fused pass)
  bool kt_direct = false;   // Bhat holds row-pass outputs: first column sub-pass by direct sum (kt_direct_fill)
  DevBuf<int> krange;                         // [nk][2] live source rows of each day kernel
  std::vector<int> hkrange;
  // staging for fetch / uploads
  DevBuf<int> orow, ocol;
  DevBuf<double> oval;
  DevBuf<const double*> wptr;
  DevBuf<double> wval;
  bool have_state = false;
  // PS_MODE_AUTO (auto_exact): this solver is the fast-torus front; `child` is the fold-mode solver
  // that takes over from the first day with anything above kCleanEps outside the domain.  The
  // child shares this solver's stream, chain records and statistics slots' meaning (day index).
  bool auto_exact = false;
  double pad_floor = 0.5e-8;   // inverse row pass publishes pad maxima above this
  ps_solver* child = nullptr;  // fold-mode helper (the reference torus itself)
  ps_solver* wide = nullptr;   // fast-torus helper sized N + 2M (flagged and clean days past the clean prefix)
  // fast-torus helper on the front's own size (N + M): days that history says are STRONGLY flagged -- more
  // than 4e-8 outside the domain, which makes the reference's flag certain whatever overlaps (auto_handover)
  ps_solver* narrow = nullptr;
  bool narrow_kernels = false;
  bool borrowed = false;       // helper: stream and chain records belong to the parent
  bool child_kernels = false, wide_kernels = false;   // the helper holds the current day kernels
  std::vector<signed char> owner;   // per chain day of the last run: 0 this solver, 1 wide, 2 child
  // the route of the last COMPLETED hand-over (days [route_first, route_end)): the next run over the same
  // days sizes its helper windows by it instead of feeling its way four days at a time (auto_handover)
  std::vector<signed char> route_hist;
  int route_first = -1, route_end = -1;
  int auto_first_regime = 2;        // helper the last hand-over started with (1 wide, 2 child)
  int auto_first = -1;         // first day of the last chain_run that ran in the child (-1: none)
  int auto_hint = -1;          // the same, relative to `first`, remembered for the next run
  int noflag_hint = 0;         // leading days of the last chain_run that raised no flag (auto: stayed clean), when it ran to its end
  long long auto_runs = 0;
  // back_solve: partial spectra of the N x N release-day filters, keyed by content.  The
  // reference calls back_solve with the same r_spread[:-1] on every simulated day
  // (CalcSol.py:308-323); each filter is uploaded, scattered and transformed once per solver.
  struct FiltKey { uint64_t h1, h2; int64_t n; };
  std::vector<FiltKey> filt_keys;
  DevBuf<cplx> Fhat;
  long long filt_hits = 0, filt_misses = 0;
  // PS_MODE_FOLD: spatial state on the reference torus + linear-convolution scratch
  DevBuf<double> torus, lin, fold_rowsum;
  DevBuf<long long> fold_rowcnt;
  DevBuf<unsigned long long> fold_padmax;
  // two-role chained pass -> batched row pass: per (day, column) energy of the pad-only rows of the
  // intermediate, and the per-day verdict "no pad-only row pair can raise the flag" (k_pad_quiet)
  DevBuf<double> pad_energy;
  DevBuf<int> pad_quiet;
  DevBuf<cplx> tail_hat;       // [day][column][Pf]: kernel spectra / chained products of the columns taken out of a chained pass
  // Deferred column half of a flagged day's re-transform (full-column pipeline, one day per pass): the
  // row pass of the truncated field has written Trow if *refft_pending fired, and then the state's
  // spectrum is FFT_columns(Trow), not Ahat -- the next day pass transforms the column itself
  // (k_colfull_day, alt_src), anything else goes through resolve_refft first
  DevBuf<cplx> Trow;
  // ps_chain_block_prefix / _finish (one simulation split over GPUs by days): the running products
  // K_first ... K_{first+i} of this solver's block of days, 2-D spectra in the pipeline's layout
  DevBuf<cplx> blk;
  int blk_first = -1, blk_count = 0;
  DevBuf<unsigned long long> one_flag;   // a pad maximum that always counts as "fired" (PS_MODE_FOLD day passes)
  const unsigned long long* refft_pending = nullptr;
  int ncu = 0;                 // compute units of the device
  // optional per-kernel-class HIP event timing (bench.py roofline leg)
  bool prof_on = false;
  struct ProfRec { int cls; hipEvent_t a, b; int days; };
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[PS_PROF_NCLS] = {0};
  long long prof_cnt[PS_PROF_NCLS] = {0};   // sampled (timed) launches
  long long prof_seen[PS_PROF_NCLS] = {0};  // all launches
  long long prof_days[PS_PROF_NCLS] = {0};  // grid-days of the timed launches (multi-day classes)
  int prof_every = 1;                       // time every n-th launch of a class
};

static hipEvent_t prof_event(ps_solver* s) {
  if (!s->prof_pool.empty()) {
    hipEvent_t e = s->prof_pool.back();
    s->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  // timing only: without the system-scope fence a default event carries -- the cache write-back and
  // invalidation around a launch that wrote gigabytes showed as 45 us of idle time on either side of it
  (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  return e;
}
struct ProfScope {
  ps_solver* s;
  ps_solver::ProfRec r;
  bool on;
  ProfScope(ps_solver* s_, int cls, int days = 1) : s(s_), on(false) {
    r.days = days;
    if (!s->prof_on) return;
    // the few multi-day launches are all timed; the others every prof_every-th per class
    on = cls >= PS_PROF_COL_INV_A2 || (s->prof_seen[cls]++ % s->prof_every) == 0;
    if (!on) return;
    r.cls = cls;
    r.a = prof_event(s);
    r.b = prof_event(s);
    (void)hipEventRecord(r.a, s->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, s->stream);
    s->prof_pending.push_back(r);
  }
};

// ------------------------------------------------------------------ helpers
static int row_threads(int L, bool big = false) { return L <= 1024 ? 256 : ((L <= 2560 || big) ? 512 : 1024); }
static int row_pairs(const FftProg& P) {
  int rp = 2048 / std::max(1, P.L);
  rp = std::max(1, std::min(8, rp));
  return rp;
}
static int col_wsh(const ps_solver* s, int L) {
  int w = 4096 / std::max(1, L);
  int sh = 2;
  while ((1 << (sh + 1)) <= w && sh < 5) ++sh;
  if (s->cfg.col_wsh >= 0) sh = s->cfg.col_wsh;   // tuning knob (tile width 2^sh columns)
  return sh;
}
static int col_threads(const ps_solver* s) { return s->cfg.col_threads; }   // tuning knob

// register-resident kernel families: rs_launch.h (instantiated in ps_rs_rows.hip / ps_colfull.hip)
static int set_lds_attr() {
  static bool done = false;
  if (done) return PS_OK;
  const void* ks[] = {(const void*)k_row_fwd<false, false>, (const void*)k_row_fwd<true, false>,
                      (const void*)k_row_fwd<false, true>,
                      (const void*)k_row_inv<false, false>, (const void*)k_row_inv<true, false>,
                      (const void*)k_row_inv<false, true>,
                      (const void*)k_col<PS_FWD, false>, (const void*)k_col<PS_FWD, true>,
                      (const void*)k_col<PS_INV, false>, (const void*)k_col<PS_INV, true>,
                      (const void*)k_col_fused<false>, (const void*)k_col_fused<true>,
                      (const void*)k_col_fused_dual<false>, (const void*)k_col_fused_dual<true>,
#define PS_MULTI_K(ND) \
  (const void*)k_col_fused_multi<false, ND, 1>, (const void*)k_col_fused_multi<true, ND, 1>, \
  (const void*)k_col_fused_multi<false, ND, 2>, (const void*)k_col_fused_multi<true, ND, 2>
                      PS_MULTI_K(1), PS_MULTI_K(2), PS_MULTI_K(4), PS_MULTI_K(8)};
#undef PS_MULTI_K
  for (const void* k : ks) PS_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
  if (rs_rows_set_attrs() != 0 || rs_colfull_set_attrs() != 0 || rs_coldual_set_attrs() != 0) return ps_fail(PS_ERR_HIP, "hipFuncSetAttribute failed for a register-resident kernel");
  done = true;
  return PS_OK;
}

static int launch_row_fwd(ps_solver* s, const double* src, int64_t src_bstride, int src_ld,
                          SrcMap rmap, SrcMap cmap, cplx* dst, int batch,
                          const unsigned long long* pred, int skip_zero = 0, const int* rowrange = nullptr,
                          const unsigned long long* trunc_pred = nullptr, int trunc_n = 0) {
  RowFwdArgs a;
  if (trunc_pred && s->rs_r2 == 0) return ps_fail(PS_ERR_STATE, "row pass: only the register-resident kernels truncate on the fly");
  a.trunc_pred = trunc_pred; a.trunc_n = trunc_n;
  a.src = src; a.src_bstride = src_bstride; a.src_ld = src_ld;
  a.rmap = rmap; a.cmap = cmap;
  a.dst = dst; a.dst_bstride = (int64_t)s->Pf * s->ld;
  a.H = s->H; a.ld = s->ld; a.P = s->Pf;
  a.tstride = s->tpipe ? s->Pf : 0;   // full-column pipeline: column-major output
  a.prog = s->row_plan.prog;
  a.rp = row_pairs(a.prog);
  a.pred = pred;
  a.skip_zero = skip_zero;
  a.rowrange = rowrange;
  const int npairs = (s->Pf + 1) / 2;
  a.nblocks = (npairs + a.rp - 1) / a.rp;
  dim3 grid(pred ? std::min(a.nblocks, kPredGrid) : a.nblocks, batch);
  const int thr = row_threads(a.prog.L, s->row_big);
  const size_t lds = ((size_t)a.rp * row_pitch(a.prog) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen) * sizeof(cplx);
  if (lds > (size_t)kMaxLds) return ps_fail(PS_ERR_UNSUPPORTED, "row pass needs %zu B LDS", lds);
  ProfScope prof(s, pred ? PS_PROF_REFFT : PS_PROF_ROW_FWD);
  if (s->rs_r2 != 0 && !s->cfg.no_rs_fwd) {
    if (!rs_launch_row_fwd(s->rs_r2, s->rs_r3, a, npairs, batch, s->stream))
      return ps_fail(PS_ERR_STATE, "no register-resident row kernel for 16 x %d x %d", s->rs_r2, s->rs_r3);
  } else if (s->row_plan.generic)
    hipLaunchKernelGGL((k_row_fwd<true, false>), grid, dim3(thr), lds, s->stream, a);
  else if (s->row_big)
    hipLaunchKernelGGL((k_row_fwd<false, true>), grid, dim3(thr), lds, s->stream, a);
  else
    hipLaunchKernelGGL((k_row_fwd<false, false>), grid, dim3(thr), lds, s->stream, a);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

template <int DIR>
static int launch_col(ps_solver* s, const ColPass& cp, const cplx* src, const cplx* src2,
                      cplx* prod, cplx* dst, int batch, int64_t src2_bstride,
                      const unsigned long long* pred, RowLive live = RowLive{0, {0, 0, 0, 0}, nullptr}) {
  ColArgs a;
  a.src = src; a.src2 = src2; a.prod_dst = prod; a.dst = dst;
  const int64_t bs = (int64_t)s->Pf * s->ld;
  a.src_bstride = bs; a.src2_bstride = src2_bstride; a.prod_bstride = bs; a.dst_bstride = bs;
  a.ld = s->ld; a.ncols = s->H;
  a.prog = cp.plan->prog;
  a.wsh = col_wsh(s, a.prog.L);
  a.n_outer = cp.n_outer;
  a.in_base_mul = cp.in_base_mul; a.in_stride = cp.in_stride;
  a.out_base_mul = cp.out_base_mul; a.out_stride = cp.out_stride;
  a.tw_mode = cp.tw_mode;
  a.tp_lo = s->tp_lo.p; a.tp_hi = s->tp_hi.p; a.tp_shift = s->tp_shift;
  a.pred = pred;
  a.live = live;
  const int W = 1 << a.wsh;
  const int ntiles = (s->H + W - 1) / W;
  dim3 grid((unsigned)(ntiles * cp.n_outer), batch);
  size_t lds = (((size_t)a.prog.L << a.wsh) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen + a.prog.L) * sizeof(cplx) + (size_t)a.prog.L * sizeof(int);
  while (lds > (size_t)kMaxLds && a.wsh > 0) {
    --a.wsh;
    lds = (((size_t)a.prog.L << a.wsh) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen + a.prog.L) * sizeof(cplx) + (size_t)a.prog.L * sizeof(int);
  }
  if (lds > (size_t)kMaxLds) return ps_fail(PS_ERR_UNSUPPORTED, "column pass needs %zu B LDS", lds);
  {
    const int W2 = 1 << a.wsh;
    a.nblocks = ((s->H + W2 - 1) / W2) * cp.n_outer;
    grid.x = (unsigned)(pred ? std::min(a.nblocks, kPredGrid) : a.nblocks);
  }
  ProfScope prof(s, pred ? PS_PROF_REFFT
                          : (DIR == PS_FWD ? PS_PROF_COL_FWD_A : PS_PROF_COL_INV_A) + (cp.second ? 1 : 0));
  if (cp.plan->generic)
    hipLaunchKernelGGL((k_col<DIR, true>), grid, dim3(col_threads(s)), lds, s->stream, a);
  else
    hipLaunchKernelGGL((k_col<DIR, false>), grid, dim3(col_threads(s)), lds, s->stream, a);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

// the persistent row kernel serves this solver's size (and is not switched off)
static bool row_inv_persistent(const ps_solver* s) {
  const int knob = s->cfg.rsp;   // A/B knob: 0 never, 1 wherever it exists, -1 the rule below
  RsInfo info;
  if (s->rs_r2 == 0 || !rs_info(s->rs_r2, s->rs_r3, &info) || !info.rsp || knob == 0) return false;
  // The tiled pipeline hands each day's spectrum over through the Infinity Cache (215 MB written by
  // the column sub-pass just before): the one-shot kernel reads it at 108 us per day where this one
  // needs 126.  The full-column pipeline writes whole groups of days first, the rows come from HBM,
  // and the prefetch wins (127 against 144 us).
  // (below ~1500 points a row pair is two waves: nothing to gain, -2 % on the R = 400 Bayes chain)
  return knob == 1 || (s->tpipe && s->Pf >= 1536);
}

// Layout of the intermediate between the inverse column pass and the inverse row pass of the full-column
// pipeline (ColFullArgs::dst_t): 1 column-major (PS_TINV, an A/B leg), 2 row pairs interleaved -- the default:
// the column pass's lanes 2i, 2i + 1 store 32 contiguous bytes instead of 16 in two lines, and the row pass
// transforms rows 2p, 2p + 1 as one complex row anyway (fft_colfull_kernels.h: colfull_dst; the 30-day
// two-role launch at 5184: 4.36 -> 4.00 ms) -- 0 row-major (PS_NO_PAIR_ROWS, fold mode: k_row_inv_fold pairs
// other rows).  The same values in other places: every result is bit-identical.
static int inter_layout(const ps_solver* s) {
  if (!s->tpipe) return 0;
  if (s->tinv) return 1;
  return (s->mode != PS_MODE_FOLD && !s->cfg.no_pair_rows && s->Pf % 2 == 0) ? 2 : 0;
}

// `recs_multi` != nullptr: batch entry b writes recs_multi[b] (separately allocated day records;
// needs row_inv_persistent(s) and batch <= 8)
static int launch_row_inv(ps_solver* s, const cplx* src, double* rec, int stat_slot, int batch,
                          double negval, double stat_scale, bool full_field = false,
                          double* const* recs_multi = nullptr, const int* pad_quiet = nullptr) {
  RowInvArgs a;
  a.pad_quiet = pad_quiet;
  a.persistent = (row_inv_persistent(s) && !(s->tpipe && s->tinv && !full_field)) ? 1 : 0;
  // the two-role anti-phase kernel (k_row_inv_rs2) where the persistent one would run and the size has it
  // (PS_ROW2: A/B knob; the launcher falls back to k_row_inv_rsp for the other sizes)
  // Default (-1): from 4096 points on -- below, several one-role workgroups share a CU and do better
  // (2688: 0.78 against 0.86 ms per chain); at 5184 the two roles take the 30-day launch from 2.67 to 2.54 ms.
  if (a.persistent && (s->cfg.row2 > 0 || (s->cfg.row2 < 0 && s->Pf >= 4096))) a.persistent = 2;
  a.nrec = 0;
  for (int i = 0; i < PS_MAX_GROUP_DAYS; ++i) a.rec_multi[i] = nullptr;
  if (recs_multi) {
    if (!a.persistent || batch > PS_MAX_GROUP_DAYS) return ps_fail(PS_ERR_STATE, "row pass: a record table needs the persistent kernel and at most %d entries", PS_MAX_GROUP_DAYS);
    a.nrec = batch;
    for (int i = 0; i < batch; ++i) a.rec_multi[i] = recs_multi[i];
  }
  a.src = src; a.src_bstride = (int64_t)s->Pf * s->ld;
  a.H = s->H; a.ld = s->ld; a.P = s->Pf; a.N = s->N;
  a.tstride = (s->tpipe && s->tinv && !full_field) ? s->Pf : 0;   // full-column pipeline: column-major intermediate
  a.pair_src = (!full_field && inter_layout(s) == 2) ? 1 : 0;
  a.prog = s->row_plan.prog;
  a.rp = row_pairs(a.prog);
  a.scale = 1.0 / ((double)s->Pf * (double)s->Pf);
  a.rec = rec; a.rec_bstride = (int64_t)s->N * s->N;
  a.negval = negval; a.stat_scale = stat_scale;
  a.rowsum = s->rowsum.p + (int64_t)stat_slot * s->N;
  a.rowcnt = s->rowcnt.p + (int64_t)stat_slot * s->N;
  a.padmax = s->padmax.p + stat_slot;
  a.pad_floor = s->pad_floor;
  a.stat_bstride = s->N;
  if (full_field) {   // PS_MODE_FOLD: the whole Pf x Pf real field, statistics into scratch
    a.N = s->Pf;
    a.rec_bstride = (int64_t)s->Pf * s->Pf;
    a.rowsum = s->fold_rowsum.p; a.rowcnt = s->fold_rowcnt.p; a.padmax = s->fold_padmax.p;
    a.stat_bstride = s->Pf;
  }
  const int npairs = (s->Pf + 1) / 2;
  dim3 grid((npairs + a.rp - 1) / a.rp, batch);
  const int thr = row_threads(a.prog.L, s->row_big);
  const size_t lds = ((size_t)a.rp * row_pitch(a.prog) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen) * sizeof(cplx) +
                     4 * (thr / 64) * sizeof(double);
  if (lds > (size_t)kMaxLds) return ps_fail(PS_ERR_UNSUPPORTED, "row pass needs %zu B LDS", lds);
  ProfScope prof(s, !recs_multi ? PS_PROF_ROW_INV : batch == 2 ? PS_PROF_ROW_INV2 : batch == 4 ? PS_PROF_ROW_INV4
                                : batch == 8 ? PS_PROF_ROW_INV8 : batch == 1 ? PS_PROF_ROW_INV : PS_PROF_ROW_INVN,
                 recs_multi ? batch : 1);
  if (s->rs_r2 != 0) {
    if (!rs_launch_row_inv(s->rs_r2, s->rs_r3, a, npairs, batch, s->stream))
      return ps_fail(PS_ERR_STATE, "no register-resident row kernel for 16 x %d x %d", s->rs_r2, s->rs_r3);
  } else if (s->row_plan.generic)
    hipLaunchKernelGGL((k_row_inv<true, false>), grid, dim3(thr), lds, s->stream, a);
  else if (s->row_big)
    hipLaunchKernelGGL((k_row_inv<false, true>), grid, dim3(thr), lds, s->stream, a);
  else
    hipLaunchKernelGGL((k_row_inv<false, false>), grid, dim3(thr), lds, s->stream, a);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

// forward 2-D transform of `batch` real sources into `out`
// one full-column pass (k_colfull): mode 0 day step (kernel -> x state -> inverse -> dst), 1 forward
// (src -> state), 2 inverse (state -> dst), 3 product only (state *= FFT(src))
// does a mode-0 launch of nd days go to the two-role chained pass?
static bool colfull_dual(const ps_solver* s, int nd) {
  const int dual_min = s->cfg.dual_min_days;   // A/B knob; 0 = never
  return dual_min > 0 && nd >= dual_min && rs_dual_ok(s->rs_r2, s->rs_r3);
}

static SrcMap map_plain(int n, int P);
// state column of a single-day pass from a row-pass output instead of the stored spectrum (ColFullArgs::alt_src)
struct ColAlt { const cplx* src; const unsigned long long* pred; RowLive live; };

// col0 / ncols: the launch covers columns [col0, ncols) (default: all H); state_bstride / dst_bstride
// override the per-batch-entry strides (default: one spectrum)
static int launch_colfull(ps_solver* s, int mode, const cplx* src, cplx* state, int store_prod, cplx* dst, int batch,
                          RowLive live, const unsigned long long* pred, int nd = 1, double* pad_energy = nullptr,
                          int col0 = 0, int ncols = -1, int64_t state_bstride = -1, bool timed = true,
                          const ColAlt* alt = nullptr) {
  ColFullArgs a;
  a.pad_energy = nullptr;
  a.alt_src = nullptr; a.alt_pred = nullptr; a.alt_live = RowLive{0, {0, 0, 0, 0}, nullptr};
  if (alt) {   // PS_MODE_FOLD: the state column always comes from the row pass of the torus
    if (!(mode == 0 && nd == 1 && batch == 1 && !pred) || s->refft_pending)
      return ps_fail(PS_ERR_STATE, "full-column pass: an alternative state source needs a plain single-day pass");
    a.alt_src = alt->src; a.alt_pred = alt->pred; a.alt_live = alt->live;
  }
  if (s->refft_pending) {
    if (!(mode == 0 && nd == 1 && batch == 1 && !pred && state == s->Ahat.p))
      return ps_fail(PS_ERR_STATE, "full-column pass: a flagged day's re-transform is still pending");
    a.alt_src = s->Trow.p; a.alt_pred = s->refft_pending;
    a.alt_live = RowLive{1, map_plain(s->N, s->Pf), nullptr};
  }
  a.pad_row0 = 2 * ((s->N + 1) / 2);   // first row of the first pad-only row pair
  a.ncols_total = s->H;
  a.col0 = col0;
  const int64_t spec = (int64_t)s->Pf * s->ld;   // T-layout arrays ([H][Pf]) fit the row-major allocation ([Pf][ld])
  a.src = src; a.src_bstride = spec;
  a.state = state; a.state_bstride = state_bstride >= 0 ? state_bstride : spec;
  a.dst = dst; a.dst_bstride = spec;
  a.ld = s->ld; a.ncols = ncols >= 0 ? ncols : s->H; a.mode = mode; a.store_prod = store_prod;
  a.nd = nd; a.src_dstride = spec; a.dst_dstride = spec;
  colfull_set_layout(a, s->tinv ? 1 : inter_layout(s), s->Pf);   // (after a.ld)
  a.live = live;
  a.pred = pred;
  a.prog = s->row_plan.prog;
  const int groups = (a.ncols - col0 + 7) / 8;   // 128-byte lines of the row-major output
  const int lines8 = (groups + 7) / 8;           // per XCD
  if (mode == 2) a.pad_energy = pad_energy;
  const bool prof_was = s->prof_on;
  if (!timed) s->prof_on = false;   // part of a scope of the caller's
  ProfScope prof(s, pred ? PS_PROF_REFFT
                          : (mode == 1 ? PS_PROF_COL_FWD_A
                                       : (nd == 2 ? PS_PROF_COL_INV_A2 : nd == 4 ? PS_PROF_COL_INV_A4
                                          : nd == 8 ? PS_PROF_COL_INV_A8 : nd == 1 ? PS_PROF_COL_INV_A : PS_PROF_COL_INV_AN)),
                 nd);
  s->prof_on = prof_was;
  // long chained groups: the two-role pass (inverse of day d next to the forward transform of day
  // d + 1 in one 12-wave workgroup; nd + 1 slots for nd days, so it pays from ~6 days on)
  if (mode == 0 && !pred && store_prod && colfull_dual(s, nd)) {
    a.pad_energy = pad_energy;
    if (!rs_launch_coldual(s->rs_r2, s->rs_r3, a, lines8, batch, s->stream))
      return ps_fail(PS_ERR_STATE, "two-role full-column pass: size 16 x %d x %d is not served", s->rs_r2, s->rs_r3);
    PS_HIP(hipGetLastError());
    return PS_OK;
  }
  if (!rs_launch_colfull(s->rs_r2, s->rs_r3, a, lines8, batch, s->stream))
    return ps_fail(PS_ERR_UNSUPPORTED, "full-column pass: size 16 x %d x %d cannot run this (chained days need the state column in LDS)",
                   s->rs_r2, s->rs_r3);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

// can the full-column pass of this size chain days (state column parked in LDS)?
static bool colfull_chains(const ps_solver* s) {
  RsInfo info;
  return rs_info(s->rs_r2, s->rs_r3, &info) && info.chain;
}

static int fwd2d(ps_solver* s, const double* src, int64_t src_bstride, int src_ld, SrcMap rmap,
                 SrcMap cmap, cplx* out, int batch, const unsigned long long* pred, const int* rowrange = nullptr) {
  // zero source rows (outside the row map, or outside [rowrange[0], rowrange[1]] when the caller
  // knows where the source ends) are neither written by the row pass nor read by the first column pass
  PS_TRY(launch_row_fwd(s, src, src_bstride, src_ld, rmap, cmap, s->T1.p, batch, pred, 1, rowrange));
  if (s->tpipe) return launch_colfull(s, 1, s->T1.p, out, 0, nullptr, batch, RowLive{1, rmap, rowrange}, pred);
  if (s->fwd_passes.size() == 1) {
    PS_TRY(launch_col<PS_FWD>(s, s->fwd_passes[0], s->T1.p, nullptr, nullptr, out, batch, 0, pred, RowLive{1, rmap, rowrange}));
  } else {
    PS_TRY(launch_col<PS_FWD>(s, s->fwd_passes[0], s->T1.p, nullptr, nullptr, s->T2.p, batch, 0, pred, RowLive{1, rmap, rowrange}));
    PS_TRY(launch_col<PS_FWD>(s, s->fwd_passes[1], s->T2.p, nullptr, nullptr, out, batch, 0, pred));
  }
  return PS_OK;
}

// inverse 2-D transform of (A [* B]) into record `rec`; optionally stores the product
static int inv2d(ps_solver* s, const cplx* A, const cplx* B, cplx* prod, double* rec,
                 int stat_slot, double negval, double stat_scale) {
  if (s->tpipe) {   // only get_cursol comes here: inverse of the state itself
    if (B || prod) return ps_fail(PS_ERR_STATE, "inv2d: product form is not used by the full-column pipeline");
    PS_TRY(launch_colfull(s, 2, nullptr, const_cast<cplx*>(A), 0, s->T1.p, 1, RowLive{0, {0, 0, 0, 0}, nullptr}, nullptr));
    return launch_row_inv(s, s->T1.p, rec, stat_slot, 1, negval, stat_scale);
  }
  if (s->inv_passes.size() == 1) {
    PS_TRY(launch_col<PS_INV>(s, s->inv_passes[0], A, B, prod, s->T1.p, 1, 0, nullptr));
    PS_TRY(launch_row_inv(s, s->T1.p, rec, stat_slot, 1, negval, stat_scale));
  } else {
    PS_TRY(launch_col<PS_INV>(s, s->inv_passes[0], A, B, prod, s->T1.p, 1, 0, nullptr));
    PS_TRY(launch_col<PS_INV>(s, s->inv_passes[1], s->T1.p, nullptr, nullptr, s->T2.p, 1, 0, nullptr));
    PS_TRY(launch_row_inv(s, s->T2.p, rec, stat_slot, 1, negval, stat_scale));
  }
  return PS_OK;
}

// forward transform of day kernels up to (not including) the last column sub-pass: that
// one is fused with the spectral product and the first inverse sub-pass (k_col_fused)
static int fwd2d_partial(ps_solver* s, const double* src, int64_t src_bstride, int src_ld,
                         SrcMap rmap, SrcMap cmap, cplx* out, int batch, const int* rowrange = nullptr,
                         bool direct = false) {
  const RowLive live{1, rmap, rowrange};
  s->kt_live = RowLive{0, {0, 0, 0, 0}, nullptr};
  s->kt_direct = s->split && direct && !s->tpipe;
  if (s->tpipe) {   // the day step's full-column pass reads the (column-major) row-pass output itself
    s->kt_live = live;
    return launch_row_fwd(s, src, src_bstride, src_ld, rmap, cmap, out, batch, nullptr, 1, rowrange);
  }
  if (!s->split || direct) {
    s->kt_live = live;  // the fused pass reads the row-pass output directly (one day at a time)
    return launch_row_fwd(s, src, src_bstride, src_ld, rmap, cmap, out, batch, nullptr, 1, rowrange);
  }
  PS_TRY(launch_row_fwd(s, src, src_bstride, src_ld, rmap, cmap, s->T1.p, batch, nullptr, 1, rowrange));
  return launch_col<PS_FWD>(s, s->fwd_passes[0], s->T1.p, nullptr, nullptr, out, batch, 0, nullptr, live);
}

static void fused_direct_args(ps_solver* s, ColFusedArgs& a) {
  a.direct = s->kt_direct ? 1 : 0;
  a.no_conj = s->cfg.no_conj;
  a.tp_lo = s->tp_lo.p; a.tp_hi = s->tp_hi.p; a.tp_shift = s->tp_shift;
  a.mgL2 = ps_magic((uint32_t)a.L2);
}
// LDS bytes of the direct-sum twiddles (stw[L2], wj[L1]) + alignment slack
static size_t fused_direct_lds(const ColFusedArgs& a) { return a.direct ? ((size_t)a.L1 + a.L2 + 1) * sizeof(cplx) + 64 : 0; }   // + slack

static int launch_col_fused(ps_solver* s, const cplx* kt, cplx* state, int store_prod, cplx* dst,
                            const int* rowrange) {
  ColFusedArgs a;
  DevPlan* plan = s->split ? &s->col_plan2 : &s->col_plan1;
  a.src = kt; a.state = state; a.dst = dst;
  a.src_bstride = 0;
  a.ld = s->ld; a.ncols = s->H;
  a.L1 = s->split ? s->L1 : 1;
  a.L2 = s->split ? s->L2 : s->Pf;
  a.store_prod = store_prod;
  a.live = s->kt_live;
  a.live.range = rowrange;
  a.prog = plan->prog;
  a.dst_dstride = 0;
  a.live2 = RowLive{0, {0, 0, 0, 0}, nullptr};
  fused_direct_args(s, a);
  // 16-column tiles: the fused pass keeps four tile transfers in flight per workgroup, and
  // twice as many (smaller) workgroups per CU beat the 512-byte segments of W = 32 (+1.7 %)
  a.wsh = std::min(col_wsh(s, a.prog.L), 4);
  if (s->cfg.fused_wsh >= 0) a.wsh = s->cfg.fused_wsh;   // tuning knob
  auto need = [&](int wsh) {
    return (((size_t)a.prog.L << wsh) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen) * sizeof(cplx) +
           (size_t)(a.prog.L + 4) * sizeof(int) + fused_direct_lds(a);
  };
  while (need(a.wsh) > (size_t)kMaxLds && a.wsh > 0) --a.wsh;
  if (need(a.wsh) > (size_t)kMaxLds) return ps_fail(PS_ERR_UNSUPPORTED, "column pass needs %zu B LDS", need(a.wsh));
  const int W = 1 << a.wsh;
  const int ntiles = (s->H + W - 1) / W;
  dim3 grid((unsigned)((a.direct ? (ntiles + 7) / 8 * 8 : ntiles) * a.L1), 1);   // see fused_tile_map
  ProfScope prof(s, PS_PROF_COL_INV_A);
  const int fthr = s->cfg.fused_threads > 0 ? s->cfg.fused_threads : col_threads(s);   // tuning knob
  if (plan->generic)
    hipLaunchKernelGGL(k_col_fused<true>, grid, dim3(fthr), need(a.wsh), s->stream, a);
  else
    hipLaunchKernelGGL(k_col_fused<false>, grid, dim3(fthr), need(a.wsh), s->stream, a);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

// nd (2, 4 or 8) consecutive days in one fused pass (k_col_fused_multi): kernels at kt + i*spec,
// outputs at dst + i*spec.  *done = 0 when the wider tile does not fit in LDS.
static int launch_col_fused_multi(ps_solver* s, const cplx* kt, int nd, cplx* state, cplx* dst,
                                  const int* rowrange, int* done) {
  ColFusedArgs a;
  DevPlan* plan = s->split ? &s->col_plan2 : &s->col_plan1;
  const int64_t spec = (int64_t)s->Pf * s->ld;
  a.src = kt; a.state = state; a.dst = dst;
  a.src_bstride = spec;
  a.dst_dstride = spec;
  a.ld = s->ld; a.ncols = s->H;
  a.L1 = s->split ? s->L1 : 1;
  a.L2 = s->split ? s->L2 : s->Pf;
  a.store_prod = 1;
  a.live = s->kt_live;
  a.live.range = rowrange;
  a.live2 = RowLive{0, {0, 0, 0, 0}, nullptr};
  a.prog = plan->prog;
  fused_direct_args(s, a);
  const int ndsh = nd == 1 ? 0 : nd == 2 ? 1 : (nd == 4 ? 2 : 3);
  // direct mode: two outer indices per workgroup halve the re-reads of the live rows
  const int gopt = s->cfg.fused_g;   // tuning knob
  const int G = (a.direct && gopt == 2 && a.L1 % 2 == 0) ? 2 : 1;
  const int gsh = G == 2 ? 1 : 0;
  // 8-column tiles for four or more days: 37 KB of LDS per workgroup keeps four of them on a CU
  a.wsh = std::min(col_wsh(s, a.prog.L), nd >= 4 ? 3 : 4);
  if (s->cfg.multi_wsh >= 0) a.wsh = s->cfg.multi_wsh;   // tuning knob
  auto need = [&](int wsh) {
    return (((size_t)a.prog.L << (wsh + ndsh + gsh)) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen) * sizeof(cplx) +
           (size_t)(a.prog.L + 4) * sizeof(int) + G * fused_direct_lds(a) +
           (a.direct ? (size_t)a.L2 * nd * sizeof(unsigned) : 0);   // + live-term table (kt_direct_entry)
  };
  // at least two workgroups per CU, or the single-day pass does better
  const size_t lds_cap = (size_t)kMaxLds / 2;
  while (need(a.wsh) > lds_cap && a.wsh > 2) --a.wsh;
  *done = 0;
  if (need(a.wsh) > lds_cap) return PS_OK;
  const int W = 1 << a.wsh;
  const int ntiles = (s->H + W - 1) / W;
  dim3 grid((unsigned)((a.direct ? (ntiles + 7) / 8 * 8 : ntiles) * (a.L1 / G)), 1);   // see fused_tile_map
  ProfScope prof(s, nd == 1 ? PS_PROF_COL_INV_A : nd == 2 ? PS_PROF_COL_INV_A2 : nd == 4 ? PS_PROF_COL_INV_A4 : PS_PROF_COL_INV_A8);
  // the paired tile (74 KB at two or four days) leaves two workgroups per CU: 512 threads each keep 16 waves there
  const int fenv = s->cfg.fused_threads;   // tuning knob
  const int fthr = fenv > 0 ? fenv : (G == 2 && nd > 1 ? 512 : col_threads(s));
#define PS_MULTI_LAUNCH(ND, GG)                                                                                          \
  if (plan->generic) hipLaunchKernelGGL((k_col_fused_multi<true, ND, GG>), grid, dim3(fthr), need(a.wsh), s->stream, a); \
  else hipLaunchKernelGGL((k_col_fused_multi<false, ND, GG>), grid, dim3(fthr), need(a.wsh), s->stream, a)
  if (G == 2) {
    if (nd == 1) { PS_MULTI_LAUNCH(1, 2); }
    else if (nd == 2) { PS_MULTI_LAUNCH(2, 2); }
    else if (nd == 4) { PS_MULTI_LAUNCH(4, 2); }
    else { PS_MULTI_LAUNCH(8, 2); }
  } else {
    if (nd == 1) { PS_MULTI_LAUNCH(1, 1); }
    else if (nd == 2) { PS_MULTI_LAUNCH(2, 1); }
    else if (nd == 4) { PS_MULTI_LAUNCH(4, 1); }
    else { PS_MULTI_LAUNCH(8, 1); }
  }
#undef PS_MULTI_LAUNCH
  PS_HIP(hipGetLastError());
  *done = 1;
  return PS_OK;
}

// PS_MODE_FOLD: kernel and state both one column sub-pass short of their spectra
// (k_col_fused_dual).  *done = 0 when the double-width tile does not fit in LDS: the caller
// takes the separate-pass route.
static int launch_col_fused_dual(ps_solver* s, const cplx* kt, const cplx* state_part, RowLive state_live,
                                 cplx* dst, const int* rowrange, int* done) {
  ColFusedArgs a;
  DevPlan* plan = s->split ? &s->col_plan2 : &s->col_plan1;
  a.src = kt; a.state = const_cast<cplx*>(state_part); a.dst = dst;
  a.src_bstride = 0;
  a.ld = s->ld; a.ncols = s->H;
  a.L1 = s->split ? s->L1 : 1;
  a.L2 = s->split ? s->L2 : s->Pf;
  a.store_prod = 0;
  a.live = s->kt_live;
  a.live.range = rowrange;
  a.live2 = state_live;
  a.prog = plan->prog;
  a.dst_dstride = 0;
  fused_direct_args(s, a);   // never direct in PS_MODE_FOLD (transform_kernels)
  a.wsh = std::min(col_wsh(s, a.prog.L), 4);
  if (s->cfg.dual_wsh >= 0) a.wsh = s->cfg.dual_wsh;   // tuning knob
  auto need = [&](int wsh) {
    return (((size_t)a.prog.L << (wsh + 1)) + a.prog.n_lo + a.prog.n_hi + a.prog.n_gen) * sizeof(cplx);
  };
  while (need(a.wsh) > (size_t)kMaxLds && a.wsh > 2) --a.wsh;
  *done = 0;
  if (need(a.wsh) > (size_t)kMaxLds || s->cfg.no_dual) return PS_OK;
  const int W = 1 << a.wsh;
  const int ntiles = (s->H + W - 1) / W;
  dim3 grid((unsigned)((a.direct ? (ntiles + 7) / 8 * 8 : ntiles) * a.L1), 1);   // see fused_tile_map
  ProfScope prof(s, PS_PROF_COL_INV_A);
  const int fthr = s->cfg.fused_threads > 0 ? s->cfg.fused_threads : col_threads(s);   // tuning knob
  if (plan->generic)
    hipLaunchKernelGGL(k_col_fused_dual<true>, grid, dim3(fthr), need(a.wsh), s->stream, a);
  else
    hipLaunchKernelGGL(k_col_fused_dual<false>, grid, dim3(fthr), need(a.wsh), s->stream, a);
  PS_HIP(hipGetLastError());
  *done = 1;
  return PS_OK;
}

// one day step: state_hat <- state_hat * K_hat (stored when store_prod), rec <- ifft2(...)
static int conv_inv(ps_solver* s, const cplx* kt, cplx* state, int store_prod, double* rec,
                    int stat_slot, double negval, double stat_scale, const int* rowrange = nullptr) {
  if (s->tpipe) {
    RowLive live = s->kt_live;
    live.range = rowrange;
    const int split2 = s->cfg.tpipe_split;   // A/B knob
    if (split2 && store_prod) {
      // two passes at <= 128 registers (two workgroups per CU each) instead of one at 162:
      // product into the state, then the inverse of the state.  Measured: 2 x 159 us against
      // 255-265 us for the single pass at 5184 -- the row-major 16-byte stores, not the
      // occupancy, are what the day pass waits for.  Off.
      PS_TRY(launch_colfull(s, 3, kt, state, 1, nullptr, 1, live, nullptr));
      PS_TRY(launch_colfull(s, 2, nullptr, state, 0, s->T1.p, 1, RowLive{0, {0, 0, 0, 0}, nullptr}, nullptr));
    } else {
      PS_TRY(launch_colfull(s, 0, kt, state, store_prod, s->T1.p, 1, live, nullptr));
    }
    return launch_row_inv(s, s->T1.p, rec, stat_slot, 1, negval, stat_scale);
  }
  int done = 0;
  // direct-sum kernels: the one-day instance of the multi kernel pairs outer indices (+5 %)
  if (s->kt_direct && store_prod) PS_TRY(launch_col_fused_multi(s, kt, 1, state, s->T1.p, rowrange, &done));
  if (!done) PS_TRY(launch_col_fused(s, kt, state, store_prod, s->T1.p, rowrange));
  if (!s->split) return launch_row_inv(s, s->T1.p, rec, stat_slot, 1, negval, stat_scale);
  PS_TRY(launch_col<PS_INV>(s, s->inv_passes[1], s->T1.p, nullptr, nullptr, s->T2.p, 1, 0, nullptr));
  return launch_row_inv(s, s->T2.p, rec, stat_slot, 1, negval, stat_scale);
}

// nd un-flagged day steps with one fused pass; recs/stat slots of days d0 .. d0+nd-1
static int conv_inv_multi(ps_solver* s, const cplx* kt, int nd, cplx* state, double* const* recs, int d0,
                          double negval, double stat_scale, const int* rowrange, int* done) {
  const size_t spec = (size_t)s->Pf * s->ld;
  PS_TRY(s->T1.ensure(spec * nd));
  if (s->tpipe) {   // nd days chained on chip by the full-column pass, then the row passes
    *done = 0;
    if (!colfull_chains(s)) return PS_OK;
    RowLive live = s->kt_live;
    live.range = rowrange;
    const bool batched_rows = row_inv_persistent(s) && !s->tinv && nd <= PS_MAX_GROUP_DAYS && !s->cfg.no_row_batch;
    // The two-role pass also sums |x|^2 over the pad-only rows of every day's intermediate; when even
    // that total cannot lift one row pair to the flag threshold, the row pass leaves the pad-only
    // pairs (21 % of the rows at N = 4097 on 5184) unread -- the per-pair Parseval test it would
    // otherwise make after fetching them, decided for the whole day.  PS_NO_PAD_QUIET=1: A/B knob.
    const bool quiet = batched_rows && colfull_dual(s, nd) && s->N + 1 < s->Pf && !s->cfg.no_pad_quiet;
    if (quiet) {
      PS_TRY(s->pad_energy.ensure((size_t)nd * s->H));
      PS_TRY(s->pad_quiet.ensure(PS_MAX_GROUP_DAYS));
    }
    // A chained pass is one workgroup per column for all nd days: 2593 columns on 256 CUs are ten
    // full rounds and an eleventh for 33 columns (9 % of the launch).  Those last columns are taken out
    // and spread over the chip day by day instead: their kernel columns' forward transforms (mode 1,
    // every (column, day) its own workgroup), the chain of products element by element (k_prefix_cols),
    // the inverse transforms (mode 2) -- three short launches, the same arithmetic in the same order.
    // PS_NO_TAIL_SPLIT=1: A/B knob.
    int hmain = s->H;
    if (!s->ncu) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, s->device) == hipSuccess) s->ncu = v;
    }
    if (colfull_dual(s, nd) && s->ncu > 0 && s->H > s->ncu && !s->tinv && !s->cfg.no_tail_split) {
      const int rem = s->H % s->ncu;
      if (rem > 0 && rem * 3 <= s->ncu) hmain = s->H - rem;
    }
    PS_TRY(launch_colfull(s, 0, kt, state, 1, s->T1.p, 1, live, nullptr, nd, quiet ? s->pad_energy.p : nullptr, 0, hmain));
    if (hmain < s->H) {
      const int rem = s->H - hmain;
      PS_TRY(s->tail_hat.ensure((size_t)nd * rem * s->Pf));
      // pointers shifted so that column c lands at [c - hmain] of the scratch block
      cplx* th = s->tail_hat.p - (int64_t)hmain * s->Pf;
      {
        ProfScope prof(s, PS_PROF_COL_TAIL, nd);
        PS_TRY(launch_colfull(s, 1, kt, th, 0, nullptr, nd, live, nullptr, 1, nullptr, hmain, s->H, (int64_t)rem * s->Pf, false));
        hipLaunchKernelGGL(k_prefix_cols, dim3((s->Pf + 255) / 256, rem), dim3(256), 0, s->stream,
                           state + (int64_t)hmain * s->Pf, s->tail_hat.p, s->Pf, rem, nd);
        PS_HIP(hipGetLastError());
        PS_TRY(launch_colfull(s, 2, nullptr, th, 0, s->T1.p, nd, RowLive{0, {0, 0, 0, 0}, nullptr}, nullptr, 1,
                              quiet ? s->pad_energy.p : nullptr, hmain, s->H, (int64_t)rem * s->Pf, false));
      }
    }
    if (quiet) {
      hipLaunchKernelGGL(k_pad_quiet, dim3(nd), dim3(256), 0, s->stream, s->pad_energy.p, s->H, (double)s->Pf,
                         1.0 / ((double)s->Pf * (double)s->Pf), s->pad_floor, s->pad_quiet.p);
      PS_HIP(hipGetLastError());
    }
    if (batched_rows) {
      // one launch for the rows of all nd days: nd x 2593 units over 256 persistent workgroups leave
      // 1/80 of a round idle at the end instead of 1/11 per day
      PS_TRY(launch_row_inv(s, s->T1.p, nullptr, d0, nd, negval, stat_scale, false, recs, quiet ? s->pad_quiet.p : nullptr));
    } else {
      for (int i = 0; i < nd; ++i)
        PS_TRY(launch_row_inv(s, s->T1.p + i * spec, recs[i], d0 + i, 1, negval, stat_scale));
    }
    *done = 1;
    return PS_OK;
  }
  PS_TRY(launch_col_fused_multi(s, kt, nd, state, s->T1.p, rowrange, done));
  if (!*done) return PS_OK;
  for (int i = 0; i < nd; ++i) {
    cplx* t = s->T1.p + i * spec;
    if (s->split) {
      PS_TRY(launch_col<PS_INV>(s, s->inv_passes[1], t, nullptr, nullptr, s->T2.p, 1, 0, nullptr));
      t = s->T2.p;
    }
    PS_TRY(launch_row_inv(s, t, recs[i], d0 + i, 1, negval, stat_scale));
  }
  return PS_OK;
}

static SrcMap map_plain(int n, int P) { return SrcMap{n, 0, P, 0}; }
// odd kernel of half width m centred at the torus origin (CalcSol.py:61-64)
static SrcMap map_wrap(int m, int P) { return SrcMap{m + 1, m, P - m, 0}; }

static int ensure_record(ps_solver* s, int kind, int idx) {
  auto& v = s->recs[kind];
  if ((int)v.size() <= idx) v.resize(idx + 1, nullptr);
  if (!v[idx]) {
    hipError_t e = ps_dev_malloc((void**)&v[idx], (size_t)s->N * s->N * sizeof(double));
    if (e != hipSuccess) return ps_fail(PS_ERR_OOM, "record allocation failed: %s", hipGetErrorString(e));
  }
  return PS_OK;
}

// n day slots plus one scratch slot (index nstat) for on-demand record statistics.
// Growing discards earlier statistics; they are rewritten by every run.
static int ensure_stats(ps_solver* s, int n) {
  if (n <= s->nstat) return PS_OK;
  PS_TRY(s->rowsum.ensure((size_t)(n + 1) * s->N));
  PS_TRY(s->rowcnt.ensure((size_t)(n + 1) * s->N));
  PS_TRY(s->padmax.ensure(n + 1));
  PS_TRY(s->dstats.ensure(n + 1));
  s->nstat = n;
  return PS_OK;
}

// reduce the per-row statistics of `count` consecutive day slots (one block each)
static int finalize_days(ps_solver* s, int slot, int count, int renorm) {
  if (count <= 0) return PS_OK;
  hipLaunchKernelGGL(k_day_finalize, dim3(count), dim3(256), 0, s->stream,
                     s->rowsum.p + (int64_t)slot * s->N, s->rowcnt.p + (int64_t)slot * s->N,
                     s->padmax.p + slot, s->N, renorm, s->dstats.p + slot);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

// truncate to the domain and re-transform when the day's flag is set (CalcSol.py:200-201)
static int refft_if_flag(ps_solver* s, const double* rec, cplx* hat, int slot) {
  return fwd2d(s, rec, 0, s->N, map_plain(s->N, s->Pf), map_plain(s->N, s->Pf), hat, 1,
               s->padmax.p + slot);
}

// ... the row half only, into Trow; the next day pass (or resolve_refft) does the columns
static int refft_rows_if_flag(ps_solver* s, const double* rec, int slot) {
  PS_TRY(s->Trow.ensure((size_t)s->Pf * s->ld));
  const SrcMap m = map_plain(s->N, s->Pf);
  PS_TRY(launch_row_fwd(s, rec, 0, s->N, m, m, s->Trow.p, 1, s->padmax.p + slot, 1, nullptr));
  s->refft_pending = s->padmax.p + slot;
  return PS_OK;
}
// the pending column half as a pass of its own: the state's spectrum is in Ahat afterwards
static int resolve_refft(ps_solver* s) {
  if (!s->refft_pending) return PS_OK;
  const unsigned long long* pred = s->refft_pending;
  s->refft_pending = nullptr;
  return launch_colfull(s, 1, s->Trow.p, s->Ahat.p, 0, nullptr, 1, RowLive{1, map_plain(s->N, s->Pf), nullptr}, pred);
}

// fast-mode FFT size for a reference pad P: the smallest even 7-smooth size, or a size served
// by the register-resident row kernels (row passes 35-50 % faster) when that costs at most
// 8 % more work
static int fast_size(const ps_config& cfg, int Pref) {
  if (cfg.fast_size >= Pref) return cfg.fast_size;   // experiments: force the fast torus (must be >= the reference pad)
  int Pf = ps_next_fast_len(Pref);
  if (!cfg.no_rs) {
    const int L = rs_next_size(Pref);
    // ... or whenever the 7-smooth size is beyond the LDS-resident row limit (~9700)
    // (PS_RS_AREA: A/B knob for the accepted area ratio)
    // Below 2048 points the tiled kernels of an awkward 7-smooth size (1372 = 28 x 49 for the R = 512
    // Bayes chain: a split column transform with tiny sub-transforms) lose far more than 20 % of area
    // costs: measured 1.05 M -> 1.37 M samples/hour (fast mode) with 1.2.
    const double area = cfg.rs_area > 0.0 ? cfg.rs_area : (Pf < 2048 ? 1.2 : 1.08);
    if (L > 0 && ((double)L * L <= area * (double)Pf * Pf || Pf > 9700)) Pf = L;
  }
  return Pf;
}

extern "C" int ps_fast_size(int dom_len, int max_shape) {
  if (dom_len < 1 || max_shape < 1) return 0;
  // a query without a handle: the knobs a solver created now would start from
  ps_config cfg;
  ps_config_from_env(&cfg);
  return fast_size(cfg, dom_len + max_shape / 2);
}

// ------------------------------------------------------------------- create
static int solver_create(ps_solver** out, int device, int dom_len, int max_shape, int mode, const ps_config& cfg);

extern "C" int ps_solver_create(ps_solver** out, int device, int dom_len, int max_shape, int mode) {
  ps_config cfg;
  ps_config_from_env(&cfg);   // the only look at the environment a solver ever takes
  return solver_create(out, device, dom_len, max_shape, mode, cfg);
}

static int solver_create(ps_solver** out, int device, int dom_len, int max_shape, int mode, const ps_config& cfg) {
  if (!out) return ps_fail(PS_ERR_BAD_ARG, "null output handle");
  *out = nullptr;
  if (dom_len < 1 || max_shape < 1) return ps_fail(PS_ERR_BAD_SHAPE, "dom_len=%d max_shape=%d", dom_len, max_shape);
  if (mode != PS_MODE_EXACT && mode != PS_MODE_FAST && mode != PS_MODE_FOLD && mode != PS_MODE_AUTO)
    return ps_fail(PS_ERR_BAD_ARG, "mode %d", mode);
  const bool auto_exact = mode == PS_MODE_AUTO;
  if (auto_exact) mode = PS_MODE_FAST;   // the front of an auto solver is a fast-torus solver
  PS_TRY(ps_use_device(device));
  PS_TRY(set_lds_attr());
  ps_solver* s = new ps_solver();
  s->cfg = cfg;
  s->tinv = cfg.tinv != 0;
  s->fused_days = cfg.fused_days;
  s->device = device;
  s->N = dom_len;
  s->M = max_shape / 2;
  s->Pref = dom_len + s->M;  // CalcSol.py:20-21, cuda_lib.py:26-28
  s->mode = mode;
  s->auto_exact = auto_exact;
  if (auto_exact) s->pad_floor = 0.5 * kCleanEps;
  s->Pf = mode == PS_MODE_FAST ? fast_size(cfg, s->Pref) : s->Pref;
  // fold mode: room for the whole linear convolution, P + K - 1 = N + 3 (K//2)
  if (mode == PS_MODE_FOLD) s->Pf = fast_size(cfg, s->Pref + 2 * s->M);
  s->H = s->Pf / 2 + 1;
  s->ld = (s->H + 7) & ~7;
  auto fail = [&](int rc) {
    ps_solver_destroy(s);
    return rc;
  };
  // large smooth rows: radix-18/16 stages (3 instead of 4 stages at 5184) when that removes a stage
  {
    HostFftPlan small, big;
    const bool ok_s = ps_build_plan(s->Pf, true, &small), ok_b = ps_build_plan(s->Pf, true, &big, true);
    if (ok_s && ok_b && small.max_prime <= 9 && s->Pf > 1024 && big.prog.ns < small.prog.ns &&
        !cfg.no_big_radix) {
      s->row_plan.host = big;
      s->row_big = true;
    }
  }
  if (!s->row_big && !ps_build_plan(s->Pf, true, &s->row_plan.host))
    return fail(ps_fail(PS_ERR_UNSUPPORTED, "cannot plan a length-%d FFT (prime factor > %d); use PS_MODE_FAST",
                        s->Pf, PS_MAX_GENERIC_RADIX));
  {
    int r2 = 0, r3 = 0;
    const bool rs = !cfg.no_rs && rs_lookup(s->Pf, &r2, &r3);   // rows stay in registers
    if (!rs && (size_t)(row_pitch(s->row_plan.host.prog) + 512) * sizeof(cplx) > (size_t)kMaxLds)
      return fail(ps_fail(PS_ERR_UNSUPPORTED, "pad size %d exceeds the LDS-resident row limit", s->Pf));
  }
  int rc = s->row_plan.upload();
  if (rc) return fail(rc);
  if (cfg.no_rs || !rs_lookup(s->Pf, &s->rs_r2, &s->rs_r3)) s->rs_r2 = s->rs_r3 = 0;
  s->tpipe_ok = (mode == PS_MODE_FAST || mode == PS_MODE_FOLD) && s->rs_r2 != 0 && !cfg.no_rs &&
                !cfg.no_rs_fwd && !cfg.no_tpipe;
  if (mode == PS_MODE_FOLD && cfg.no_fold_tpipe) s->tpipe_ok = false;   // A/B knob
  s->tpipe = s->tpipe_ok;   // fold mode: always the full-column pipeline when its size allows (no compact-kernel route there)
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) s->num_cu = ncu;
  }
  {
    s->L1 = ps_choose_col_split(s->Pf, cfg.col_single_max);   // tuning knob
  }
  if (cfg.col_l1 > 1 && cfg.col_l1 < s->Pf && s->Pf % cfg.col_l1 == 0) s->L1 = cfg.col_l1;   // tuning knob: first sub-pass length of the column split
  s->split = s->L1 != s->Pf;
  if (s->split) {
    s->L2 = s->Pf / s->L1;
    if (!ps_build_plan(s->L1, false, &s->col_plan1.host) || !ps_build_plan(s->L2, false, &s->col_plan2.host))
      return fail(ps_fail(PS_ERR_UNSUPPORTED, "cannot plan column FFT %d = %d x %d", s->Pf, s->L1, s->L2));
    if ((rc = s->col_plan1.upload())) return fail(rc);
    if ((rc = s->col_plan2.upload())) return fail(rc);
    s->fwd_passes = {ColPass{&s->col_plan1, s->L2, 1, s->L2, 1, s->L2, 1},
                     ColPass{&s->col_plan2, s->L1, s->L2, 1, 1, s->L1, 0, true}};
    s->inv_passes = {ColPass{&s->col_plan2, s->L1, 1, s->L1, s->L2, 1, 0},
                     ColPass{&s->col_plan1, s->L2, 1, s->L2, 1, s->L2, 2, true}};
  } else {
    s->L2 = 1;
    if (!ps_build_plan(s->Pf, false, &s->col_plan1.host))
      return fail(ps_fail(PS_ERR_UNSUPPORTED, "cannot plan column FFT %d", s->Pf));
    if ((rc = s->col_plan1.upload())) return fail(rc);
    s->fwd_passes = {ColPass{&s->col_plan1, 1, 0, 1, 0, 1, 0}};
    s->inv_passes = {ColPass{&s->col_plan1, 1, 0, 1, 0, 1, 0}};
  }
  {
    HostTwiddle tw;
    ps_build_twiddle(s->Pf, &tw);
    s->tp_shift = tw.shift;
    if ((rc = s->tp_lo.ensure(tw.lo.size()))) return fail(rc);
    if ((rc = s->tp_hi.ensure(tw.hi.size()))) return fail(rc);
    hipError_t e1 = hipMemcpy(s->tp_lo.p, tw.lo.data(), tw.lo.size() * sizeof(cplx), hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(s->tp_hi.p, tw.hi.data(), tw.hi.size() * sizeof(cplx), hipMemcpyHostToDevice);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(ps_fail(PS_ERR_HIP, "twiddle upload failed"));
  }
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return fail(ps_fail(PS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
  const size_t spec = (size_t)s->Pf * s->ld;
  // bytes of batched kernel spectra per chunk: 1/16 of the device memory (18 GB of the MI355X's 288),
  // so the 30 kernels of a 4097^2 stack (6.4 GB) are one chunk and the speculation windows keep
  // growing (2, 4, 8, 8, 8 days) instead of starting over at every chunk boundary
  size_t budget = (size_t)3 << 30;
  {
    size_t mfree = 0, mtotal = 0;
    if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && mtotal / 16 > budget) budget = mtotal / 16;
  }
  s->chunk_days = (int)std::max<size_t>(1, std::min<size_t>(64, budget / (spec * sizeof(cplx))));
  if (cfg.chunk_days > 0) s->chunk_days = cfg.chunk_days;   // tuning knob
  if ((rc = s->Ahat.ensure(spec))) return fail(rc);
  if ((rc = s->rowoff.ensure(s->N))) return fail(rc);
  if ((rc = ensure_stats(s, 4))) return fail(rc);
  *out = s;
  return PS_OK;
}

extern "C" int ps_solver_destroy(ps_solver* s) {
  if (!s) return PS_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->child) {
    ps_solver_destroy(s->child);
    s->child = nullptr;
  }
  if (s->wide) {
    ps_solver_destroy(s->wide);
    s->wide = nullptr;
  }
  if (s->narrow) {
    ps_solver_destroy(s->narrow);
    s->narrow = nullptr;
  }
  if (s->borrowed) {   // stream and chain records are the parent's
    s->stream = nullptr;
    for (auto& p : s->recs[PS_REC_CHAIN]) p = nullptr;
  }
  if (s->stream2) {
    (void)hipStreamSynchronize(s->stream2);
    (void)hipStreamDestroy(s->stream2);
  }
  if (s->kt_ev) (void)hipEventDestroy(s->kt_ev);
  if (s->kt_ev0) (void)hipEventDestroy(s->kt_ev0);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  ps_dev_quiesce();
  s->row_plan.release(); s->col_plan1.release(); s->col_plan2.release();
  s->tp_lo.release(); s->tp_hi.release();
  s->Ahat.release(); s->Chat.release(); s->T1.release(); s->T2.release(); s->Bhat.release(); s->Fhat.release();
  s->Trow.release(); s->blk.release(); s->one_flag.release(); s->tail_hat.release(); s->pad_energy.release(); s->pad_quiet.release();
  s->srange.release();
  s->krow.release(); s->kcol.release(); s->kval.release(); s->kdense.release(); s->krange.release();
  for (auto& v : s->recs)
    for (double* p : v)
      if (p) ps_dev_free(p);
  s->rowsum.release(); s->rowcnt.release(); s->rowoff.release(); s->padmax.release();
  s->dstats.release();
  s->orow.release(); s->ocol.release(); s->oval.release(); s->wptr.release(); s->wval.release();
  for (auto& r : s->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : s->prof_pool) (void)hipEventDestroy(e);
  for (auto e : s->spec_ev) if (e) (void)hipEventDestroy(e);
  if (s->hflags) (void)hipHostFree(s->hflags);
  if (s->hhist) (void)hipHostFree(s->hhist);
  if (s->hist_ev) (void)hipEventDestroy(s->hist_ev);
  s->dkoff.release(); s->dkshape.release();
  s->torus.release(); s->lin.release(); s->fold_rowsum.release(); s->fold_rowcnt.release(); s->fold_padmax.release();
  delete s;
  return PS_OK;
}

extern "C" int ps_solver_retarget(ps_solver* s, int max_shape) {
  if (!s || max_shape < 1) return ps_fail(PS_ERR_BAD_ARG, "retarget: bad arguments");
  if (s->auto_exact) {
    // the front only needs its FFT size to hold the new reference torus; the fold child is
    // re-targeted when it fits and rebuilt on demand when it does not
    const int m = max_shape / 2;
    if (s->N + m > s->Pf)
      return ps_fail(PS_ERR_BAD_SHAPE, "retarget: max_shape %d needs an FFT size >= %d, the solver has %d", max_shape,
                     s->N + m, s->Pf);
    PS_HIP(hipSetDevice(s->device));
    PS_HIP(hipStreamSynchronize(s->stream));
    if (s->child && ps_solver_retarget(s->child, max_shape) != PS_OK) {
      ps_solver_destroy(s->child);
      s->child = nullptr;
    }
    if (s->wide) {   // a fast-mode solver cannot move its torus: rebuilt on demand
      ps_solver_destroy(s->wide);
      s->wide = nullptr;
    }
    if (s->narrow) {
      ps_solver_destroy(s->narrow);
      s->narrow = nullptr;
    }
    s->narrow_kernels = false;
    s->child_kernels = false;
    s->wide_kernels = false;
    s->route_first = s->route_end = -1;   // another torus, another route
    s->M = m;
    s->Pref = s->N + m;
    s->have_state = false;
    s->bhat_first = -1;
    s->bhat_count = 0;
    return PS_OK;
  }
  if (s->mode != PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "retarget: only PS_MODE_FOLD solvers can change their torus");
  const int m = max_shape / 2;
  if (s->N + 3 * m > s->Pf)
    return ps_fail(PS_ERR_BAD_SHAPE, "retarget: max_shape %d needs an FFT size >= %d, the solver has %d", max_shape,
                   s->N + 3 * m, s->Pf);
  PS_HIP(hipSetDevice(s->device));
  PS_HIP(hipStreamSynchronize(s->stream));
  s->M = m;
  s->Pref = s->N + m;
  s->have_state = false;
  s->bhat_first = -1;
  s->bhat_count = 0;
  return PS_OK;
}

extern "C" int ps_solver_info(ps_solver* s, int* dom_len, int* P, int* Pfft, int* H) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  if (dom_len) *dom_len = s->N;
  if (P) *P = s->Pref;
  if (Pfft) *Pfft = s->Pf;
  if (H) *H = s->H;
  return PS_OK;
}

extern "C" int ps_solver_sync(ps_solver* s) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  PS_HIP(hipSetDevice(s->device));
  PS_HIP(hipStreamSynchronize(s->stream));
  return PS_OK;
}

static int ensure_temps(ps_solver* s, int batch) {
  const size_t spec = (size_t)s->Pf * s->ld;
  PS_TRY(s->T1.ensure(spec * batch));
  if (s->split) PS_TRY(s->T2.ensure(spec * batch));
  return PS_OK;
}

static int check_coo(const int32_t* row, const int32_t* col, int64_t nnz, int lim, const char* what) {
  for (int64_t i = 0; i < nnz; ++i)
    if (row[i] < 0 || row[i] >= lim || col[i] < 0 || col[i] >= lim)
      return ps_fail(PS_ERR_BAD_SHAPE, "%s: entry %lld (%d,%d) outside %d x %d", what, (long long)i, row[i],
                     col[i], lim, lim);
  return PS_OK;
}

static int scatter_from_device(ps_solver* s, const int* row, const int* col, const double* val,
                               int64_t nnz, double* dst, int ld, int off) {
  if (nnz <= 0) return PS_OK;
  const int thr = 256;
  const int blocks = (int)std::min<int64_t>((nnz + thr - 1) / thr, 4096);
  hipLaunchKernelGGL(k_scatter_coo, dim3(blocks), dim3(thr), 0, s->stream, row, col, val, nnz, dst, ld, off, off);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

static int upload_coo(ps_solver* s, const int32_t* row, const int32_t* col, const double* val, int64_t nnz) {
  PS_TRY(s->orow.ensure(std::max<int64_t>(nnz, 1)));
  PS_TRY(s->ocol.ensure(std::max<int64_t>(nnz, 1)));
  PS_TRY(s->oval.ensure(std::max<int64_t>(nnz, 1)));
  if (nnz > 0) {
    PS_HIP(hipMemcpyAsync(s->orow.p, row, nnz * 4, hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->ocol.p, col, nnz * 4, hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->oval.p, val, nnz * 8, hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipStreamSynchronize(s->stream));  // host buffers are caller-owned: copy completes before return
  }
  return PS_OK;
}

int ps_solver_set_state_device_coo(ps_solver* s, const int* row, const int* col, const double* val,
                                   int64_t nnz, int off) {
  PS_TRY(ensure_record(s, PS_REC_STATE, 0));
  PS_TRY(ensure_temps(s, 1));
  s->srange_valid = false;   // set_state_coo knows the rows and says so afterwards
  double* rec = s->recs[PS_REC_STATE][0];
  PS_HIP(hipMemsetAsync(rec, 0, (size_t)s->N * s->N * sizeof(double), s->stream));
  PS_TRY(scatter_from_device(s, row, col, val, nnz, rec, s->N, off));
  if (s->mode == PS_MODE_FOLD) {   // the state lives in space, on the reference torus
    PS_TRY(s->torus.ensure((size_t)s->Pref * s->Pref));
    PS_HIP(hipMemsetAsync(s->torus.p, 0, (size_t)s->Pref * s->Pref * sizeof(double), s->stream));
    PS_HIP(hipMemcpy2DAsync(s->torus.p, (size_t)s->Pref * sizeof(double), rec, (size_t)s->N * sizeof(double),
                            (size_t)s->N * sizeof(double), (size_t)s->N, hipMemcpyDeviceToDevice, s->stream));
    s->have_state = true;
    return PS_OK;
  }
  s->spec_valid = false;   // transformed by the first consumer (ensure_spectrum)
  s->have_state = true;
  return PS_OK;
}

// Ahat <- transform of the spatial state record, in the layout of the current pipeline
static int ensure_spectrum(ps_solver* s) {
  if (s->spec_valid || s->mode == PS_MODE_FOLD) return PS_OK;
  if (s->recs[PS_REC_STATE].empty() || !s->recs[PS_REC_STATE][0]) return ps_fail(PS_ERR_STATE, "no state record");
  PS_TRY(ensure_temps(s, 1));
  PS_TRY(fwd2d(s, s->recs[PS_REC_STATE][0], 0, s->N, map_plain(s->N, s->Pf), map_plain(s->N, s->Pf), s->Ahat.p, 1, nullptr,
               s->srange_valid ? s->srange.p : nullptr));
  s->spec_valid = true;
  return PS_OK;
}

// switch between the tiled and the full-column pipeline; only possible while the state's
// spectrum has not been built (or can be rebuilt from its record)
static void set_pipeline(ps_solver* s, bool tpipe) {
  if (!s->tpipe_ok) tpipe = false;
  if (s->tpipe == tpipe) return;
  s->tpipe = tpipe;
  s->bhat_first = -1;      // kernel transforms and cached filter spectra are layout-specific
  s->bhat_count = 0;
  s->filt_keys.clear();
}

extern "C" int ps_solver_set_state_coo(ps_solver* s, const int32_t* row, const int32_t* col,
                                       const double* val, int64_t nnz) {
  if (!s || nnz < 0 || (nnz > 0 && (!row || !col || !val))) return ps_fail(PS_ERR_BAD_ARG, "set_state_coo: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(check_coo(row, col, nnz, s->N, "state"));
  PS_TRY(upload_coo(s, row, col, val, nnz));
  PS_TRY(ps_solver_set_state_device_coo(s, s->orow.p, s->ocol.p, s->oval.p, nnz, 0));
  // rows that hold anything: a release state is a handful of rows of the N, and the forward row
  // pass (145 us at 4097^2 for all of them) only has to transform those
  int range[2] = {1, 0};
  for (int64_t i = 0; i < nnz; ++i) {
    if (range[0] > range[1]) range[0] = range[1] = row[i];
    range[0] = std::min(range[0], (int)row[i]);
    range[1] = std::max(range[1], (int)row[i]);
  }
  PS_TRY(s->srange.ensure(2));
  hipLaunchKernelGGL(k_set_int2, dim3(1), dim3(64), 0, s->stream, s->srange.p, range[0], range[1]);
  PS_HIP(hipGetLastError());
  s->srange_valid = true;
  return PS_OK;
}

// compact day kernels (few live rows per residue class of the column split) can take the
// direct-sum first column sub-pass of the tiled pipeline (kt_direct_fill)
static bool direct_possible(const ps_solver* s) {
  return s->split && s->mode != PS_MODE_FOLD && !s->cfg.no_direct &&
         s->L1 <= 255 && s->L2 <= 255;   // kt_direct_entry packs term indices in bytes
}
static bool day_is_compact(const ps_solver* s, int d) {
  const int max_terms = s->cfg.direct_max_terms;   // tuning knob
  const int M = s->Kmax / 2;
  const int lo = s->hkrange[2 * d], hi = s->hkrange[2 * d + 1];
  if (lo > hi) return true;
  // live rows split at the kernel centre M into the two wrapped intervals
  const int below = std::max(0, std::min(hi, M - 1) - lo + 1), above = std::max(0, hi - std::max(lo, M) + 1);
  const int terms = (below + s->L2 - 1) / s->L2 + (above + s->L2 - 1) / s->L2;
  return terms <= max_terms;
}

// transform `count` kernels starting at day `first` into Bhat[0..count)
// The transforms go to slots [slot0, slot0 + count) of buffers sized for `total` slots (a chunk
// transformed in two parts: the second part on the second stream).
// `direct`: -1 decide from these days; 0 / 1 the caller decided for the whole chunk (a chunk
// transformed in two calls must hold ONE format: kt_direct / kt_live are solver-wide)
static int transform_kernels(ps_solver* s, int first, int count, int slot0, int total, int direct_chunk) {
  const int K = s->Kmax, M = K / 2;
  const size_t spec = (size_t)s->Pf * s->ld;
  if (total < 0) total = count;
  PS_TRY(s->kdense.ensure((size_t)total * K * K));
  PS_TRY(s->Bhat.ensure(spec * total));
  PS_TRY(ensure_temps(s, total));
  double* kd = s->kdense.p + (size_t)slot0 * K * K;
  // zero only the band of staging rows some kernel of the chunk writes (the row pass reads
  // nothing else, see krange) with one small kernel, then one batched scatter launch
  int blo = K, bhi = -1;
  int64_t maxn = 0;
  for (int d = first; d < first + count; ++d) {
    if (s->hkrange[2 * d] <= s->hkrange[2 * d + 1]) {
      blo = std::min(blo, s->hkrange[2 * d]);
      bhi = std::max(bhi, s->hkrange[2 * d + 1]);
    }
    maxn = std::max<int64_t>(maxn, s->koff[d + 1] - s->koff[d]);
  }
  if (bhi >= blo) {
    blo = std::max(blo, 0);
    bhi = std::min(bhi, K - 1);
    const int64_t n = (int64_t)(bhi - blo + 1) * K;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_zero_band, dim3(blocks, count), dim3(256), 0, s->stream, kd, K, blo, bhi);
    PS_HIP(hipGetLastError());
  }
  if (maxn > 0) {
    const int thr = 256;
    const int blocks = (int)std::min<int64_t>((maxn + thr - 1) / thr, 1024);
    hipLaunchKernelGGL(k_scatter_coo_batch, dim3(blocks, count), dim3(thr), 0, s->stream, s->krow.p, s->kcol.p,
                       s->kval.p, s->dkoff.p, s->dkshape.p, first, kd, K);
    PS_HIP(hipGetLastError());
  }
  // compact kernels (few live rows per residue class of the column split): skip the first
  // column sub-pass, the fused pass sums the live rows directly (kt_direct_fill)
  // (measured: +14 % on the headline stack, whose kernels need 6-8 terms per output; break-even
  // near 8 terms at FFT size 5760; the fold-mode fused pass, with the state in the other half of
  // its tile, does not gain)
  bool direct = direct_possible(s);
  if (direct_chunk >= 0) direct = direct && direct_chunk != 0;
  else for (int d = first; d < first + count && direct; ++d) direct = day_is_compact(s, d);
  PS_TRY(fwd2d_partial(s, kd, (int64_t)K * K, K, map_wrap(M, s->Pf), map_wrap(M, s->Pf), s->Bhat.p + (size_t)slot0 * spec, count,
                       s->krange.p + 2 * first, direct));
  s->bhat_first = first - slot0;
  s->bhat_count = slot0 + count;
  return PS_OK;
}

// main stream: wait for the second stream's kernel transforms before day `d` is used
static int transform_kernels(ps_solver* s, int first, int count, int slot0 = 0, int total = -1, int direct_chunk = -1);
static int kernels_ready(ps_solver* s, int d) {
  if (s->kt_lazy_from >= 0 && d >= s->kt_lazy_from) {
    const int from = s->kt_lazy_from, done = from - s->kt_lazy_c0;
    s->kt_lazy_from = -1;
    PS_TRY(transform_kernels(s, from, s->kt_lazy_cn - done, done, s->kt_lazy_cn, s->kt_lazy_direct));
  }
  if (s->kt_from >= 0 && d >= s->kt_from) {
    PS_HIP(hipStreamWaitEvent(s->stream, s->kt_ev, 0));
    s->kt_from = -1;
  }
  return PS_OK;
}

// Before the kernel triplets, their offset / shape tables or the live-row ranges are re-allocated
// or overwritten: everything queued that may still read them has to be done -- a previous run's
// scatter of the later days sits on the low-priority second stream, and callers such as
// PopModel.evaluate(want_stats=False) return from ps_chain_run without synchronising.  (The
// main stream is ordered behind the second one at the end of every ps_chain_run.)
static int drain_kernel_readers(ps_solver* s) {
  if (s->stream2) PS_HIP(hipStreamSynchronize(s->stream2));
  PS_HIP(hipStreamSynchronize(s->stream));
  return PS_OK;
}

// `rows`: host copy of the COO row indices, or nullptr (kernels already on the device: the
// whole K_d box is taken as live -- prob_mass kernels are shrunk to their support)
static int set_kernels_common(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                              const int32_t* rows) {
  s->koff.assign(off, off + nk + 1);
  s->kshape.assign(kshape, kshape + nk);
  s->nk = nk;
  s->Kmax = 1;
  for (int d = 0; d < nk; ++d) {
    if (kshape[d] < 1 || kshape[d] % 2 == 0)
      return ps_fail(PS_ERR_BAD_SHAPE, "kernel %d has even/invalid shape %d (CalcSol.py:58)", d, kshape[d]);
    if (kshape[d] / 2 > s->M)
      return ps_fail(PS_ERR_BAD_SHAPE, "kernel %d shape %d exceeds max_shape %d", d, kshape[d], 2 * s->M + 1);
    s->Kmax = std::max(s->Kmax, kshape[d]);
  }
  if (2 * (s->Kmax / 2) + 1 > s->Pf) return ps_fail(PS_ERR_BAD_SHAPE, "kernel larger than the pad");
  // live rows of each kernel inside the Kmax x Kmax staging block
  s->hkrange.assign(2 * (size_t)std::max(nk, 1), 0);
  const int M = s->Kmax / 2;
  for (int d = 0; d < nk; ++d) {
    const int o = M - kshape[d] / 2;
    int lo = 0, hi = kshape[d] - 1;
    if (rows && off[d + 1] > off[d]) {
      lo = kshape[d];
      hi = -1;
      for (int64_t i = off[d]; i < off[d + 1]; ++i) {
        lo = std::min(lo, (int)rows[i]);
        hi = std::max(hi, (int)rows[i]);
      }
    } else if (rows) {
      lo = 1;
      hi = 0;  // empty kernel: nothing live
    }
    s->hkrange[2 * d] = lo + o;
    s->hkrange[2 * d + 1] = hi + o;
  }
  PS_TRY(s->krange.ensure(s->hkrange.size()));
  PS_TRY(s->dkoff.ensure((size_t)nk + 1));
  PS_TRY(s->dkshape.ensure((size_t)std::max(nk, 1)));
  {
    std::vector<long long> o64(off, off + nk + 1);
    PS_HIP(hipMemcpy(s->dkoff.p, o64.data(), o64.size() * sizeof(long long), hipMemcpyHostToDevice));
    if (nk > 0) PS_HIP(hipMemcpy(s->dkshape.p, kshape, (size_t)nk * sizeof(int), hipMemcpyHostToDevice));
  }
  PS_HIP(hipMemcpyAsync(s->krange.p, s->hkrange.data(), s->hkrange.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  s->bhat_first = -1;
  s->bhat_count = 0;
  s->kernels_on_device = true;
  s->child_kernels = false;
  s->wide_kernels = false;
  s->narrow_kernels = false;
  return PS_OK;
}

extern "C" int ps_chain_set_kernels(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                                    const int32_t* row, const int32_t* col, const double* val) {
  if (!s || nk < 0 || !off || (nk > 0 && !kshape)) return ps_fail(PS_ERR_BAD_ARG, "set_kernels: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  const int64_t tot = off[nk];
  for (int d = 0; d < nk; ++d) {
    if (off[d + 1] < off[d]) return ps_fail(PS_ERR_BAD_ARG, "offsets not monotone");
    PS_TRY(check_coo(row + off[d], col + off[d], off[d + 1] - off[d], kshape[d], "kernel"));
  }
  PS_TRY(drain_kernel_readers(s));
  PS_TRY(s->krow.ensure(std::max<int64_t>(tot, 1)));
  PS_TRY(s->kcol.ensure(std::max<int64_t>(tot, 1)));
  PS_TRY(s->kval.ensure(std::max<int64_t>(tot, 1)));
  if (tot > 0) {
    PS_HIP(hipMemcpyAsync(s->krow.p, row, tot * 4, hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->kcol.p, col, tot * 4, hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->kval.p, val, tot * 8, hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipStreamSynchronize(s->stream));
  }
  return set_kernels_common(s, nk, off, kshape, row);
}

// used by the model module (kernels already on the device)
int ps_chain_adopt_device_kernels(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                                  const int* row, const int* col, const double* val) {
  const int64_t tot = off[nk];
  PS_TRY(drain_kernel_readers(s));
  PS_TRY(s->krow.ensure(std::max<int64_t>(tot, 1)));
  PS_TRY(s->kcol.ensure(std::max<int64_t>(tot, 1)));
  PS_TRY(s->kval.ensure(std::max<int64_t>(tot, 1)));
  if (tot > 0) {
    PS_HIP(hipMemcpyAsync(s->krow.p, row, tot * 4, hipMemcpyDeviceToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->kcol.p, col, tot * 4, hipMemcpyDeviceToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->kval.p, val, tot * 8, hipMemcpyDeviceToDevice, s->stream));
  }
  return set_kernels_common(s, nk, off, kshape, nullptr);
}

// ---- PS_MODE_AUTO beyond the clean prefix ----------------------------------------------------
// From the first unclean day on, the days of the run alternate between two helpers that share
// this solver's stream, kernels and records:
//   * the WIDE solver: a fast-torus solver sized for N + 2M (not N + M).  A day that starts from a
//     domain-supported state (after a flag's truncation, or a clean day) is a true LINEAR
//     convolution there: its domain part is exactly the reference's, and with m' = the largest
//     value outside the domain (overhangs do not overlap on this torus) the reference's flag --
//     the maximum over its pad, where up to four overhang pieces are summed -- is certainly
//     raised when m' > 1e-8 (pieces are non-negative) and certainly not when m' < 0.25e-8.  So
//     the wide solver keeps the days with m' > 1e-8 (flagged: truncate + re-FFT on the device, as
//     in fast mode) and the clean ones (m' < 1e-15);
//   * the fold CHILD (PS_MODE_FOLD: the reference torus itself, pad dust included) takes every
//     other day -- dust between 1e-15 and 1e-8 that the reference carries around its torus --
//     starting from the field before that day, until one of its days raises the flag: the
//     truncated field is domain-supported again and the wide solver resumes.
// Days run in windows of 4; the window's pad maxima are read back and the first day that
// belongs to the other helper restarts there (what was enqueued behind it is overwritten in
// stream order).  `owner[d]` remembers which solver holds day d's statistics.
static int auto_attach(ps_solver* s, ps_solver** slot, int mode, int max_shape, bool* kernels_ok, int end) {
  if (!*slot) {
    ps_solver* c = nullptr;
    PS_TRY(solver_create(&c, s->device, s->N, max_shape, mode, s->cfg));   // the parent's knobs, not the environment's
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
    c->stream = s->stream;
    c->borrowed = true;
    c->speculate = false;                 // predicated re-FFT: the window logic here does the host checks
    c->pad_floor = 0.5 * kCleanEps;
    c->prof_on = s->prof_on;              // a profiled run times its helpers' launches too (ps_prof_read_owner)
    c->prof_every = s->prof_every;
    *slot = c;
    *kernels_ok = false;
  }
  ps_solver* c = *slot;
  if (!*kernels_ok) {
    PS_TRY(ps_chain_adopt_device_kernels(c, s->nk, s->koff.data(), s->kshape.data(), s->krow.p, s->kcol.p, s->kval.p));
    // the front may know tighter live-row ranges (host COO rows) than the whole K_d box; they are
    // relative to the Kmax x Kmax staging block, which is the same in every helper
    c->hkrange = s->hkrange;
    PS_HIP(hipMemcpyAsync(c->krange.p, c->hkrange.data(), c->hkrange.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    PS_HIP(hipStreamSynchronize(c->stream));
    *kernels_ok = true;
  }
  // statistics slots for the whole run up front: growing them later would drop the windows
  // already computed (ensure_stats)
  PS_TRY(ensure_stats(c, std::max(4, end)));
  return PS_OK;
}

// the field the chain had before day d: record d-1, or the first-day state
static const double* auto_prev_field(ps_solver* s, int first, int d) {
  if (d > 0 && d - 1 < (int)s->recs[PS_REC_CHAIN].size() && s->recs[PS_REC_CHAIN][d - 1] && (d > first || first > 0))
    return s->recs[PS_REC_CHAIN][d - 1];
  if (d == first && first == 0 && !s->recs[PS_REC_STATE].empty()) return s->recs[PS_REC_STATE][0];
  return nullptr;
}

static int auto_alias_records(ps_solver* s, ps_solver* c, int lo, int hi) {
  auto& cr = c->recs[PS_REC_CHAIN];
  if ((int)cr.size() < hi) cr.resize(hi, nullptr);
  for (int d = lo; d < hi; ++d) {
    PS_TRY(ensure_record(s, PS_REC_CHAIN, d));
    cr[d] = s->recs[PS_REC_CHAIN][d];
  }
  return PS_OK;
}

static int auto_read_padmax(ps_solver* s, ps_solver* c, int d, int w, double* out) {
  if (s->hflags_n < d + w) {
    if (s->hflags) (void)hipHostFree(s->hflags);
    s->hflags = nullptr;
    s->hflags_n = 0;
    PS_HIP(hipHostMalloc((void**)&s->hflags, (size_t)(d + w + 64) * sizeof(unsigned long long), hipHostMallocDefault));
    s->hflags_n = d + w + 64;
  }
  PS_HIP(hipMemcpyAsync(s->hflags + d, c->padmax.p + d, (size_t)w * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  for (int i = 0; i < w; ++i) __builtin_memcpy(&out[i], &s->hflags[d + i], sizeof(double));
  return PS_OK;
}

// `m_front`: what the front saw outside the domain on day f (its torus superposes the overhangs,
// so the wide solver's m' is not larger): <= 1e-8 means day f cannot be a flagged day there --
// start with the child.
static int auto_handover(ps_solver* s, int first, int f, int end, double negval, double stat_scale, int renorm,
                         double m_front) {
  // The wide helper pays off where a day step is long against a host round trip per window and
  // a solver per kernel-shape class is not rebuilt all the time: big grids (Carnarvon R = 2048
  // rad_dist 10 km: 495 -> 570 grid-days/s; the R = 400 sampler, whose kernel extent moves with
  // every proposal, ran 2x SLOWER with it).  PS_WIDE_MIN_N: smallest domain that uses it.
  const int wide_min_n = s->cfg.wide_min_n;
  const bool no_wide = s->cfg.no_wide || s->N < wide_min_n;   // fold child for everything
  const int kWin = std::min(64, std::max(1, s->cfg.auto_window));
  // Route history: sampler chains and repeated runs take much the same route every time.  With the
  // previous run's route at hand a helper gets its whole stretch of days in ONE ps_chain_run (no host
  // round trip every four days, no days enqueued behind a hand-over and thrown away), the wide helper
  // is not even tried on a day that was dusty last time, and the child keeps going over a flagged day
  // when the day after it was dusty again (it is exact on any day; only speed says to leave it).  Every
  // window is still checked against what the days really did: a wrong history costs time, never
  // correctness.  PS_NO_ROUTE_HISTORY=1: A/B knob.
  std::vector<signed char> hist;
  if (!s->cfg.no_route_history && s->route_first == first && s->route_end == end) hist = s->route_hist;
  if ((int)s->owner.size() < end) s->owner.resize(end, 0);
  for (int d = first; d < f; ++d) s->owner[d] = 0;
  enum { WIDE = 1, CHILD = 2, NARROW = 3 };
  // what the next run should try per day: the owner, except that a wide-torus day whose flag was strong
  // (or that was clean with room to spare) is marked for the narrow helper
  std::vector<signed char> hint((size_t)end, 0);
  auto hist_owner = [&](int q) { return q >= 0 && q < (int)hist.size() && q < end ? (int)hist[q] : -1; };
  auto stretch = [&](int q, int who) {   // days from q on that belonged to `who` last time (0: no history)
    int e = q;
    while (e < end && e - q < 64 && hist_owner(e) == who) ++e;
    return e - q;
  };
  int regime = (no_wide || !(m_front > 1e-8)) ? CHILD : WIDE;
  s->auto_first_regime = regime;
  bool wide_live = false, child_live = false, narrow_live = false;   // the helper's state continues the chain at day d
  int no_narrow_day = -1;     // the narrow torus could not decide this day: the wide one does
  int d = f;
  while (d < end) {
    const double* prev = auto_prev_field(s, first, d);
    if (regime == NARROW && hist_owner(d) != NARROW) {
      regime = WIDE;
      narrow_live = false;
    }
    if (regime == WIDE && hist_owner(d) == NARROW && d != no_narrow_day) {
      regime = NARROW;
      wide_live = false;
    }
    if (regime == WIDE && hist_owner(d) == CHILD) {   // dusty last time: do not try the wide torus on it
      regime = CHILD;
      wide_live = false;
      continue;
    }
    if (regime == NARROW) {
      // A fast torus of the FRONT's size (N + M).  Overhangs overlap there as they do on the reference's
      // torus, in other places -- but a pad value is a sum of at most four non-negative overhang pieces
      // on either torus, so with m_f the largest value outside the domain here and m' the largest single
      // piece: m' <= m_f <= 4 m'.  m_f > 4e-8 makes m' > 1e-8: the reference raises its flag for certain,
      // truncates, and the domain part (no wrap into the domain on any torus >= N + M_d) is exactly its
      // field; m_f < 1e-15 is a clean day.  Anything in between goes back to the wide torus.  Only ever
      // entered on history: a day the wide torus saw above 4e-8 last time (0.55 instead of 0.81 ms at
      // N = 4097).
      PS_TRY(auto_attach(s, &s->narrow, PS_MODE_FAST, 2 * s->M + 1, &s->narrow_kernels, end));
      ps_solver* c = s->narrow;
      if (!narrow_live) {
        if (!prev) return ps_fail(PS_ERR_STATE, "auto mode: no field to continue day %d from", d);
        PS_TRY(ensure_record(c, PS_REC_STATE, 0));
        PS_HIP(hipMemcpyAsync(c->recs[PS_REC_STATE][0], prev, (size_t)s->N * s->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        c->spec_valid = false;
        c->srange_valid = false;
        c->have_state = true;
        narrow_live = true;
        wide_live = child_live = false;
      }
      const int w = std::min(std::max(1, stretch(d, NARROW)), end - d);
      PS_TRY(auto_alias_records(s, c, d, d + w));
      PS_TRY(ps_chain_run(c, d, w, negval, stat_scale, renorm));
      double m[64];
      PS_TRY(auto_read_padmax(s, c, d, std::min(w, 64), m));
      int x = -1;
      for (int i = 0; i < w && x < 0; ++i)
        if (m[i] >= kCleanEps && !(m[i] > 4e-8)) x = d + i;
      const int keep = x < 0 ? d + w : x;
      for (int q = d; q < keep; ++q) s->owner[q] = hint[q] = NARROW;
      d = keep;
      if (x >= 0) {
        regime = WIDE;
        narrow_live = false;
        no_narrow_day = x;
      }
      continue;
    }
    if (regime == WIDE) {
      PS_TRY(auto_attach(s, &s->wide, PS_MODE_FAST, 4 * s->M + 1, &s->wide_kernels, end));
      ps_solver* c = s->wide;
      if (!wide_live) {
        if (!prev) return ps_fail(PS_ERR_STATE, "auto mode: no field to continue day %d from", d);
        PS_TRY(ensure_record(c, PS_REC_STATE, 0));
        PS_HIP(hipMemcpyAsync(c->recs[PS_REC_STATE][0], prev, (size_t)s->N * s->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        c->spec_valid = false;
        c->srange_valid = false;   // a whole field: every row may hold something
        c->have_state = true;
        wide_live = true;
        child_live = narrow_live = false;
      }
      const int hw = stretch(d, WIDE);
      const int w = std::min(hw > 0 ? hw : kWin, end - d);
      PS_TRY(auto_alias_records(s, c, d, d + w));
      PS_TRY(ps_chain_run(c, d, w, negval, stat_scale, renorm));
      double m[64];
      PS_TRY(auto_read_padmax(s, c, d, std::min(w, 64), m));
      int x = -1;
      for (int i = 0; i < w && x < 0; ++i)
        if (m[i] >= kCleanEps && !(m[i] > 1e-8)) x = d + i;     // dust the reference would carry: not ours
      const int keep = x < 0 ? d + w : x;
      for (int q = d; q < keep; ++q) {
        s->owner[q] = WIDE;
        const double mq = m[q - d];
        hint[q] = (mq > 4e-8 || mq < 0.25 * kCleanEps) ? NARROW : WIDE;
      }
      d = keep;
      if (x >= 0) {
        regime = CHILD;
        wide_live = false;
      }
    } else {
      PS_TRY(auto_attach(s, &s->child, PS_MODE_FOLD, 2 * s->M + 1, &s->child_kernels, end));
      ps_solver* c = s->child;
      if (!child_live) {
        if (!prev) return ps_fail(PS_ERR_STATE, "auto mode: no field to continue day %d from", d);
        PS_TRY(c->torus.ensure((size_t)c->Pref * c->Pref));
        PS_HIP(hipMemsetAsync(c->torus.p, 0, (size_t)c->Pref * c->Pref * sizeof(double), c->stream));
        PS_HIP(hipMemcpy2DAsync(c->torus.p, (size_t)c->Pref * sizeof(double), prev, (size_t)s->N * sizeof(double),
                                (size_t)s->N * sizeof(double), (size_t)s->N, hipMemcpyDeviceToDevice, c->stream));
        c->have_state = true;
        child_live = true;
        wide_live = narrow_live = false;
      }
      const int hc = stretch(d, CHILD);
      const int w = no_wide ? end - d : std::min(hc > 0 ? hc : kWin, end - d);
      PS_TRY(auto_alias_records(s, c, d, d + w));
      PS_TRY(ps_chain_run(c, d, w, negval, stat_scale, renorm));
      int g = -1;
      if (!no_wide) {
        double m[64];
        PS_TRY(auto_read_padmax(s, c, d, std::min(w, 64), m));
        for (int i = 0; i < w && g < 0; ++i)
          if (m[i] > 1e-8 && hist_owner(d + i + 1) != CHILD) g = d + i;   // flagged: truncated, domain-supported again
                                                                          // (and not dusty again the day after, last time)
      }
      const int keep = g < 0 ? d + w : g + 1;
      for (int q = d; q < keep; ++q) s->owner[q] = hint[q] = CHILD;
      d = keep;
      if (g >= 0) {
        regime = WIDE;
        child_live = false;
      }
    }
  }
  s->auto_first = f;
  s->auto_hint = f - first;
  s->route_hist.assign(hint.begin(), hint.end());
  s->route_first = first;
  s->route_end = end;
  s->have_state = false;   // the front's spectrum is void now: per-call API needs a new state
  return PS_OK;
}

static int chain_run_body(ps_solver* s, int first, int count, double negval, double stat_scale, int renorm);

extern "C" int ps_chain_run(ps_solver* s, int first, int count, double negval, double stat_scale,
                            int renorm) {
  const int rc = chain_run_body(s, first, count, negval, stat_scale, renorm);
  // whatever path the run took (error, hand-over to a helper before the later days were reached):
  // nothing of this solver's second stream is left unordered behind the main stream
  if (s && s->kt_from >= 0) {
    (void)hipStreamWaitEvent(s->stream, s->kt_ev, 0);
    s->kt_from = -1;
  }
  if (s) s->kt_lazy_from = -1;   // kernels this run never reached stay untransformed
  // the last day's re-transform, if it was deferred: every other entry point finds the spectrum in Ahat
  if (s && s->refft_pending) {
    if (rc == PS_OK) return resolve_refft(s);
    s->refft_pending = nullptr;
  }
  return rc;
}

static int chain_run_body(ps_solver* s, int first, int count, double negval, double stat_scale, int renorm) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  if (!s->have_state)
    return ps_fail(PS_ERR_STATE, s->auto_exact && s->auto_first >= 0
                                     ? "auto mode: set the state before every chain run (the previous run handed days over to the "
                                       "exact-torus helpers, the front solver's spectrum is void)"
                                     : "chain_run before set_state");
  if (!s->kernels_on_device || first < 0 || count < 0 || first + count > s->nk)
    return ps_fail(PS_ERR_STATE, "chain_run: days [%d,%d) not uploaded (nk=%d)", first, first + count, s->nk);
  PS_HIP(hipSetDevice(s->device));
  // sized for every uploaded day at once: a run continued in a second call must not lose the statistics
  // of the first (growing the buffers does not keep their contents)
  PS_TRY(ensure_stats(s, std::max(4, s->nk)));
  for (int d = first; d < first + count; ++d) PS_TRY(ensure_record(s, PS_REC_CHAIN, d));
  PS_HIP(hipMemsetAsync(s->padmax.p + first, 0, (size_t)count * sizeof(unsigned long long), s->stream));
  s->last_renorm = renorm;
  if (s->mode == PS_MODE_FOLD) {
    // Per day: transform the torus field (zero-padded to the FFT size), multiply with the day
    // kernel's spectrum inside the fused column pass, invert to the full linear-convolution
    // field, fold it back modulo P (k_fold: next state, record, statistics, pad maximum) and
    // drop the pad region if the day raised the flag.  Same results as the direct transform on
    // the reference torus up to round-off, on fast FFT sizes.
    const size_t spec = (size_t)s->Pf * s->ld;
    PS_TRY(s->Ahat.ensure(spec));
    PS_TRY(s->fold_rowsum.ensure((size_t)s->Pf));
    PS_TRY(s->fold_rowcnt.ensure((size_t)s->Pf));
    PS_TRY(s->fold_padmax.ensure(1));
    PS_TRY(ensure_temps(s, 1));
    PS_TRY(s->T2.ensure(spec));
    // Full-column pipeline on the register-resident kernels: the day is row pass of the torus -> day pass
    // (state column + kernel column -> product -> inverse) -> row pass that folds on chip.  The forward
    // column pass of the state is no launch of its own on sizes whose day pass can transform the state
    // column itself (ALT; the others keep it), the truncation of a flagged day's torus never is (the next
    // row pass reads the flag and leaves the pad region out; the torus itself is truncated once, when the
    // run ends), and k_fold is gone with the M x M field.  PS_NO_FOLD_FUSE=1, PS_NO_FOLD_ALT=1: A/B knobs.
    const bool fuse = s->tpipe && s->rs_r2 != 0 && !s->cfg.no_fold_fuse;
    const bool fuse_alt = fuse && rs_colfull_alt_ok(s->rs_r2, s->rs_r3) && !s->cfg.no_fold_alt;
    // the folding row pass needs a torus row to fold at most once; PS_NO_FOLD_ROWS=1: A/B knob.  Only the
    // two-kernel form needs the M x M field in memory (564 MB at 8400 points)
    const bool fold_rows = fuse && s->Pref >= 2 * s->M && !s->cfg.no_fold_rows;
    if (!fold_rows) PS_TRY(s->lin.ensure((size_t)s->Pf * s->Pf));
    if (fuse_alt && !s->one_flag.p) {
      PS_TRY(s->one_flag.ensure(1));
      const double one = 1.0;
      PS_HIP(hipMemcpyAsync(s->one_flag.p, &one, sizeof(double), hipMemcpyHostToDevice, s->stream));
      PS_HIP(hipStreamSynchronize(s->stream));   // `one` is a stack variable
    }
    for (int c0 = first; c0 < first + count; c0 += s->chunk_days) {
      const int cn = std::min(s->chunk_days, first + count - c0);
      PS_TRY(transform_kernels(s, c0, cn));
      for (int d = c0; d < c0 + cn; ++d) {
        const cplx* B = s->Bhat.p + (size_t)(d - s->bhat_first) * s->Pf * s->ld;
        const SrcMap tmap = map_plain(s->Pref, s->Pf);
        if (fuse) {
          PS_TRY(launch_row_fwd(s, s->torus.p, 0, s->Pref, tmap, tmap, s->T1.p, 1, nullptr, 1, nullptr,
                                d > first ? s->padmax.p + d - 1 : nullptr, s->N));
          RowLive klive = s->kt_live;
          klive.range = s->krange.p + 2 * d;
          if (fuse_alt) {
            const ColAlt alt{s->T1.p, s->one_flag.p, RowLive{1, tmap, nullptr}};
            PS_TRY(launch_colfull(s, 0, B, s->Ahat.p, 0, s->T2.p, 1, klive, nullptr, 1, nullptr, 0, -1, -1, true, &alt));
          } else {
            PS_TRY(launch_colfull(s, 1, s->T1.p, s->Ahat.p, 0, nullptr, 1, RowLive{1, tmap, nullptr}, nullptr));
            PS_TRY(launch_colfull(s, 0, B, s->Ahat.p, 0, s->T2.p, 1, klive, nullptr));
          }
          if (fold_rows) {   // inverse row pass and fold in one kernel
            RowFoldArgs fa;
            fa.src = s->T2.p; fa.ld = s->ld; fa.M = s->Pf; fa.P = s->Pref; fa.N = s->N; fa.m = s->M;
            fa.scale = 1.0 / ((double)s->Pf * (double)s->Pf); fa.negval = negval; fa.stat_scale = stat_scale;
            fa.torus = s->torus.p; fa.rec = s->recs[PS_REC_CHAIN][d];
            fa.rowsum = s->rowsum.p + (int64_t)d * s->N; fa.rowcnt = s->rowcnt.p + (int64_t)d * s->N;
            fa.padmax = s->padmax.p + d;
            fa.prog = s->row_plan.prog;
            const int units = 2 * s->M + (s->Pref - 2 * s->M + 1) / 2;
            ProfScope prof(s, PS_PROF_ROW_INV);
            if (!rs_launch_row_fold(s->rs_r2, s->rs_r3, fa, units, s->stream))
              return ps_fail(PS_ERR_STATE, "no fold row kernel for 16 x %d x %d", s->rs_r2, s->rs_r3);
            PS_HIP(hipGetLastError());
          } else {
            PS_TRY(launch_row_inv(s, s->T2.p, s->lin.p, d, 1, negval, stat_scale, true));
            hipLaunchKernelGGL(k_fold, dim3(s->Pref), dim3(256), 0, s->stream, s->lin.p, s->Pf, s->Pref, s->N, s->M,
                               s->torus.p, s->recs[PS_REC_CHAIN][d], negval, stat_scale,
                               s->rowsum.p + (int64_t)d * s->N, s->rowcnt.p + (int64_t)d * s->N, s->padmax.p + d);
            PS_HIP(hipGetLastError());
          }
          if (d == first + count - 1) {   // the run ends here: what the next run (or a hand-over) finds is truncated for real
            hipLaunchKernelGGL(k_truncate_if_flag, dim3(s->Pref), dim3(256), 0, s->stream, s->torus.p, s->Pref, s->N,
                               s->padmax.p + d);
            PS_HIP(hipGetLastError());
          }
          continue;
        }
        if (s->tpipe) {
          // full-column pipeline: state row pass (column-major out) -> one forward column pass ->
          // day pass (kernel column x state column -> inverse; the product is not kept: the state
          // lives in space) -> row pass to the full linear-convolution field -> fold
          PS_TRY(launch_row_fwd(s, s->torus.p, 0, s->Pref, tmap, tmap, s->T1.p, 1, nullptr, 1));
          PS_TRY(launch_colfull(s, 1, s->T1.p, s->Ahat.p, 0, nullptr, 1, RowLive{1, tmap, nullptr}, nullptr));
          RowLive klive = s->kt_live;
          klive.range = s->krange.p + 2 * d;
          PS_TRY(launch_colfull(s, 0, B, s->Ahat.p, 0, s->T2.p, 1, klive, nullptr));
          PS_TRY(launch_row_inv(s, s->T2.p, s->lin.p, d, 1, negval, stat_scale, true));
          hipLaunchKernelGGL(k_fold, dim3(s->Pref), dim3(256), 0, s->stream, s->lin.p, s->Pf, s->Pref, s->N, s->M,
                             s->torus.p, s->recs[PS_REC_CHAIN][d], negval, stat_scale,
                             s->rowsum.p + (int64_t)d * s->N, s->rowcnt.p + (int64_t)d * s->N, s->padmax.p + d);
          PS_HIP(hipGetLastError());
          hipLaunchKernelGGL(k_truncate_if_flag, dim3(s->Pref), dim3(256), 0, s->stream, s->torus.p, s->Pref, s->N,
                             s->padmax.p + d);
          PS_HIP(hipGetLastError());
          continue;
        }
        // state: row pass (+ first column sub-pass); its last forward sub-pass runs inside the
        // fused kernel next to the kernel's
        PS_TRY(launch_row_fwd(s, s->torus.p, 0, s->Pref, tmap, tmap, s->T1.p, 1, nullptr, 1));
        const cplx* part = s->T1.p;
        cplx* fused_out = s->T2.p;
        RowLive part_live{1, tmap, nullptr};
        if (s->split) {
          PS_TRY(launch_col<PS_FWD>(s, s->fwd_passes[0], s->T1.p, nullptr, nullptr, s->T2.p, 1, 0, nullptr, part_live));
          part = s->T2.p;
          fused_out = s->T1.p;
          part_live = RowLive{0, {0, 0, 0, 0}, nullptr};
        }
        int dual = 0;
        PS_TRY(launch_col_fused_dual(s, B, part, part_live, fused_out, s->krange.p + 2 * d, &dual));
        if (!dual) {  // finish the state's spectrum, then the ordinary fused pass
          const ColPass& last = s->fwd_passes.back();
          PS_TRY(launch_col<PS_FWD>(s, last, part, nullptr, nullptr, s->Ahat.p, 1, 0, nullptr, part_live));
          PS_TRY(launch_col_fused(s, B, s->Ahat.p, 0, fused_out, s->krange.p + 2 * d));
        }
        if (s->split) {
          PS_TRY(launch_col<PS_INV>(s, s->inv_passes[1], s->T1.p, nullptr, nullptr, s->T2.p, 1, 0, nullptr));
          PS_TRY(launch_row_inv(s, s->T2.p, s->lin.p, d, 1, negval, stat_scale, true));
        } else {
          PS_TRY(launch_row_inv(s, s->T2.p, s->lin.p, d, 1, negval, stat_scale, true));
        }
        hipLaunchKernelGGL(k_fold, dim3(s->Pref), dim3(256), 0, s->stream, s->lin.p, s->Pf, s->Pref, s->N, s->M,
                           s->torus.p, s->recs[PS_REC_CHAIN][d], negval, stat_scale,
                           s->rowsum.p + (int64_t)d * s->N, s->rowcnt.p + (int64_t)d * s->N, s->padmax.p + d);
        PS_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_truncate_if_flag, dim3(s->Pref), dim3(256), 0, s->stream, s->torus.p, s->Pref, s->N,
                           s->padmax.p + d);
        PS_HIP(hipGetLastError());
      }
    }
    return PS_OK;
  }
  s->spec_window = 2;   // 2 + 4 + 8 = one chunk of 14 days, all of them in multi-day fused passes
  if (s->cfg.no_speculation) s->speculate = false;
  // A solver whose previous run went through without a flag (sampler chains and ensemble members re-run
  // the same solver on new kernels) does not feel its way 2, 4, 8, ...: it opens with windows of up to
  // PS_MAX_GROUP_DAYS = 32 days (a 30-day run: one window), each ONE chained full-column
  // pass and ONE row launch.  A flag inside such a window is handled like any other (days behind
  // it are redone); the hint is then gone.  PS_NO_WINDOW_HINT=1: A/B knob.
  const int hint = s->noflag_hint;
  s->noflag_hint = 0;
  bool hinted = false;
  int flagged_at = -1;
  // PS_MODE_AUTO: always speculate, with "clean" (nothing above kCleanEps outside the domain)
  // in the role of "no flag"; the first unclean day hands the rest of the chain to the fold path.
  // A solver whose previous run was unclean from its very first day skips the front (and looks
  // again every 16th run: the parameters may have moved).
  const double flag_thr = s->auto_exact ? kCleanEps : 1e-8;
  int hint_abs = -1;
  if (s->auto_exact) {
    s->speculate = true;
    s->auto_first = -1;
    ++s->auto_runs;
    if (count > 0 && s->auto_hint == 0 && s->auto_runs % 16 != 0)
      return auto_handover(s, first, first, first + count, negval, stat_scale, renorm,
                           s->auto_first_regime == 1 ? 1.0 : 0.0);   // start where the last run's hand-over started
    if (s->auto_hint > 0) hint_abs = first + s->auto_hint;
  }
  // flags of the previous run over the same days (see `guided` below)
  std::vector<char> hist;
  if (!s->auto_exact && !s->speculate && s->hist_count == count && s->hist_first == first && count >= 2 &&
      !s->cfg.no_flag_history && !s->cfg.no_speculation) {
    PS_HIP(hipEventSynchronize(s->hist_ev));
    hist.resize(count);
    for (int i = 0; i < count; ++i) {
      double m;
      __builtin_memcpy(&m, &s->hhist[i], sizeof(double));
      hist[i] = !(m <= 1e-8);   // NaN counts as flagged: no speculation on it
    }
  }
  s->hist_count = 0;
  if (s->speculate || !hist.empty()) {
    for (int i = 0; i < 2; ++i)
      if (!s->spec_ev[i]) PS_HIP(hipEventCreateWithFlags(&s->spec_ev[i], hipEventDisableTiming));
    if (s->hflags_n < first + count) {
      if (s->hflags) (void)hipHostFree(s->hflags);
      s->hflags = nullptr;
      s->hflags_n = 0;
      PS_HIP(hipHostMalloc((void**)&s->hflags, (size_t)(first + count) * sizeof(unsigned long long), hipHostMallocDefault));
      s->hflags_n = first + count;
    }
  }
  // Pipeline of this run, chosen while the state's spectrum is still to be built: compact day
  // kernels on a split column transform do best in the tiled pipeline (multi-day fused passes
  // with direct-sum kernels: 2650 against 2420 grid-days/s on the synthetic N = 4097 stack);
  // everything else -- broad prob_mass kernels, flagged days -- in the full-column pipeline
  // (Carnarvon R = 2048: 1040 -> 1440 grid-days/s).
  if (s->tpipe_ok && !s->spec_valid) {
    bool compact = direct_possible(s);
    for (int d = first; d < first + count && compact; ++d) compact = day_is_compact(s, d);
    // Compact kernels on a split column size have the direct-sum tiled route (4-day groups, no kernel
    // spectra in HBM); it still wins where the full-column pass cannot chain days (state column + exchange
    // buffer beyond LDS, L > 6400).  Everywhere else the full-column pipeline is ahead since the
    // persistent row kernel and the pipelined kernel fetch: 10.5 against 11.7 ms per 30-day stack at 5184,
    // 8.2 against 9.4 at 4608 (single-pass column sizes too: +6 % on the flag-heavy R = 400 Bayes chain).
    bool want = !(s->split && compact) || colfull_chains(s);
    if (s->cfg.tpipe >= 0) want = s->cfg.tpipe != 0;   // A/B knob
    set_pipeline(s, want);
  }
  PS_TRY(ensure_spectrum(s));
  if (s->speculate && s->tpipe && colfull_chains(s) && hint >= count && count >= 4 && !s->cfg.no_window_hint) {
    // windows of up to PS_MAX_GROUP_DAYS days, the whole run if it fits.  Measured on the 30-day stack
    // (PS_FIRST_WINDOW): 6 / 8 / 10 / 15 days first and the rest behind -- with the later windows'
    // kernel transforms on the low-priority stream -- 7.98 ms each, one 30-day window 7.84: the idle
    // slot of a second chained pass and the launches saved outweigh the 0.2 ms of kernel transforms
    // that are no longer hidden.
    int w0 = s->cfg.first_window > 0 ? s->cfg.first_window : count;   // tuning knob
    w0 = std::max(2, std::min(std::min(w0, count), PS_MAX_GROUP_DAYS));
    s->spec_window = w0;
    hinted = true;
  }
  // A solver that has seen a flag no longer speculates blindly, but sampler chains and repeated runs raise
  // their flags on much the same days every time: with the previous run's flags at hand, the stretches of
  // >= 2 days that raised none go through the chained pass again (verified like any speculation window:
  // a flag inside one sends the rest of the run down the safe path), every other day runs with its
  // predicated re-transform, which is right whatever its flag says.  PS_NO_FLAG_HISTORY=1: A/B knob.
  bool guided = false;
  if (!hist.empty() && s->tpipe && colfull_chains(s))
    for (int i = 0; i + 1 < count && !guided; ++i) guided = !hist[i] && !hist[i + 1];
  for (int c0 = first; c0 < first + count; c0 += s->chunk_days) {
    const int cn = std::min(s->chunk_days, first + count - c0);
    // Full-column pipeline, long chunk: only the kernels of the first three windows (2 + 4 + 8 days)
    // are transformed ahead of the day passes; the rest go to a second, low-priority stream and
    // run in the CUs the day passes leave idle -- a chained pass has 2593 columns for 256 CUs, its
    // eleventh round occupies 33 of them -- instead of 0.2 ms up front.
    const int split_days = s->cfg.kt_split >= 0 ? s->cfg.kt_split : (hinted ? s->spec_window : 14);   // = the first window(s); A/B knob: 0 = off
    const bool lazy_tail = s->auto_exact && hint_abs >= c0 && hint_abs + 2 < c0 + cn &&
                           !s->cfg.no_lazy_kt;   // see the branch below (A/B knob)
    if (!lazy_tail && s->tpipe && s->speculate && split_days > 0 && cn >= split_days + (hinted ? 4 : 8)) {
      if (!s->stream2) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = numerically greatest = lowest priority
        PS_HIP(hipStreamCreateWithPriority(&s->stream2, hipStreamNonBlocking, lo));
        PS_HIP(hipEventCreateWithFlags(&s->kt_ev, hipEventDisableTiming));
        PS_HIP(hipEventCreateWithFlags(&s->kt_ev0, hipEventDisableTiming));
      }
      PS_TRY(transform_kernels(s, c0, split_days, 0, cn));
      // the second stream starts behind everything the main stream has queued so far (an earlier
      // run may still be reading these slots)
      PS_HIP(hipEventRecord(s->kt_ev0, s->stream));
      PS_HIP(hipStreamWaitEvent(s->stream2, s->kt_ev0, 0));
      hipStream_t main_stream = s->stream;
      s->stream = s->stream2;
      const int rc2 = transform_kernels(s, c0 + split_days, cn - split_days, split_days, cn);
      s->stream = main_stream;
      if (rc2 != PS_OK) return rc2;
      PS_HIP(hipEventRecord(s->kt_ev, s->stream2));
      s->kt_from = c0 + split_days;
    } else if (lazy_tail) {
      // the previous run handed over at day hint_abs: this one will most likely not need the front's
      // spectra of the kernels behind it (29 kernels at 5600 points are 2.6 ms of a 47 ms Carnarvon
      // chain, 17 at 1120 points a twentieth of a Bayes evaluation); kernels_ready transforms the
      // rest if the run stays clean for longer
      // (one format for the whole chunk, decided over all its days: the two halves share kt_direct / kt_live)
      const int now = hint_abs + 2 - c0;
      bool direct_all = direct_possible(s);
      for (int q = c0; q < c0 + cn && direct_all; ++q) direct_all = day_is_compact(s, q);
      s->kt_lazy_direct = direct_all ? 1 : 0;
      PS_TRY(transform_kernels(s, c0, now, 0, cn, s->kt_lazy_direct));
      s->kt_lazy_from = c0 + now;
      s->kt_lazy_c0 = c0;
      s->kt_lazy_cn = cn;
    } else {
      PS_TRY(transform_kernels(s, c0, cn));
    }
    auto day = [&](int d, bool with_refft) -> int {
      PS_TRY(kernels_ready(s, d));
      const cplx* B = s->Bhat.p + (size_t)(d - s->bhat_first) * s->Pf * s->ld;
      double* rec = s->recs[PS_REC_CHAIN][d];
      PS_TRY(conv_inv(s, B, s->Ahat.p, 1, rec, d, negval, stat_scale, s->krange.p + 2 * d));
      if (with_refft) {
        // full-column pipeline: only the row half now, the column half inside the next day's pass
        // (PS_NO_DEFER_REFFT=1: A/B knob)
        const bool defer = !s->cfg.no_defer_refft && !s->cfg.tpipe_split;
        if (s->tpipe && defer && rs_colfull_alt_ok(s->rs_r2, s->rs_r3)) PS_TRY(refft_rows_if_flag(s, rec, d));
        else PS_TRY(refft_if_flag(s, rec, s->Ahat.p, d));
      }
      return PS_OK;
    };
    // Speculation on the boundary flag (CalcSol.py:200-201): windows of days are enqueued
    // WITHOUT the three flag-conditional re-FFT launches (~6 us each even when they do
    // nothing) and their pad maxima are copied to pinned host memory behind them.  The host
    // checks a window only after the NEXT one has been enqueued, so the GPU never waits for
    // the check.  No flag: the days stand as they are (the conditional launches would have
    // returned at once).  First flag at day f: its truncated field is re-transformed for
    // real, every day after f -- computed from a spectrum the reference would have replaced
    // -- is redone, and this solver stops speculating.  Windows grow 2, 4, 8, ... so an
    // early flag wastes little.
    struct Win { int d0, w, ev; };
    std::deque<Win> q;
    (void)flagged_at;
    int d = c0, nev = 0;
    while (true) {
      if (!s->speculate && !guided) {
        for (; d < c0 + cn; ++d) PS_TRY(day(d, true));
        break;
      }
      const size_t depth = (size_t)std::max(1, s->cfg.spec_depth);
      // auto mode: stop enqueueing right after the day that was the first unclean one last time
      // until its check is in -- the days behind it are likely to be redone by the fold path
      while (q.size() < depth && d < c0 + cn && !(hint_abs >= 0 && d == hint_abs + 1 && !q.empty())) {
        int w = std::min(s->spec_window, c0 + cn - d);
        if (hint_abs >= d) w = std::min(w, hint_abs - d + 1);
        if (guided) {
          int g = 0;   // days from d on that raised no flag last time
          while (d + g < c0 + cn && g < PS_MAX_GROUP_DAYS && !hist[d + g - first]) ++g;
          if (g < 2) {   // nothing to verify behind a day that carries its own re-transform
            PS_TRY(day(d, true));
            ++d;
            continue;
          }
          w = g;
          PS_TRY(resolve_refft(s));   // the chained pass reads the state's spectrum
        }
        // inside a window no flag is expected: days go through the fused pass in groups
        for (int i = 0; i < w;) {
          int g = 0;
          // direct mode pairs outer indices (twice the tile): four days per pass there
          const int direct_days = s->cfg.direct_days;   // tuning knob
          // (the full-column pipeline chains the days with the state column in registers)
          const int tpipe_days = std::min(PS_MAX_GROUP_DAYS, std::max(1, s->cfg.tpipe_days));   // tuning knob
          const int maxd = s->tpipe ? std::min(s->fused_days >= 8 ? PS_MAX_GROUP_DAYS : s->fused_days, tpipe_days)
                                    : (s->kt_direct ? std::min(s->fused_days, direct_days) : s->fused_days);
          if (maxd > 1 && w - i >= 2) {
            // the chained full-column pass takes any number of days; the tiled fused pass 2, 4 or 8
            int nd = s->tpipe ? std::min(maxd, w - i)
                              : (maxd >= 8 && w - i >= 8) ? 8 : (maxd >= 4 && w - i >= 4) ? 4 : 2;
            // a chained group keeps nd intermediates (6.9 GB for 32 days at 5184, per solver -- helper and
            // shape-class solvers each own theirs): when that does not fit, shorter groups, not an abort
            // (per-solver footprint: PS_TPIPE_DAYS x one half spectrum)
            while (nd > 2 && s->T1.ensure((size_t)s->Pf * s->ld * nd) == PS_ERR_OOM) {
              (void)hipGetLastError();
              nd = std::max(2, nd / 2);
              s->cfg.tpipe_days = std::min(s->cfg.tpipe_days, nd);   // and stay there: the memory will not come back mid-run
            }
            PS_TRY(kernels_ready(s, d + i + nd - 1));
            const cplx* B = s->Bhat.p + (size_t)(d + i - s->bhat_first) * s->Pf * s->ld;
            PS_TRY(conv_inv_multi(s, B, nd, s->Ahat.p, &s->recs[PS_REC_CHAIN][d + i], d + i, negval, stat_scale,
                                  s->krange.p + 2 * (d + i), &g));
            if (g) g = nd;
            else s->fused_days = 1;   // tile does not fit: single days from here on
          }
          if (!g) {
            PS_TRY(day(d + i, false));
            g = 1;
          }
          i += g;
        }
        PS_HIP(hipMemcpyAsync(s->hflags + d, s->padmax.p + d, (size_t)w * sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, s->stream));
        PS_HIP(hipEventRecord(s->spec_ev[nev & 1], s->stream));
        q.push_back(Win{d, w, nev & 1});
        ++nev;
        d += w;
        s->spec_window = hinted ? PS_MAX_GROUP_DAYS : std::min(64, 2 * s->spec_window);
      }
      if (q.empty()) break;
      const Win x = q.front();
      q.pop_front();
      PS_HIP(hipEventSynchronize(s->spec_ev[x.ev]));
      int f = -1;
      for (int i = 0; i < x.w && f < 0; ++i) {
        double m;
        __builtin_memcpy(&m, &s->hflags[x.d0 + i], sizeof(double));
        if (m > flag_thr) f = x.d0 + i;
      }
      if (f < 0) continue;
      q.clear();   // whatever was enqueued after day f is void; stream order keeps it harmless
      flagged_at = f;
      if (s->auto_exact) {
        double mf;
        __builtin_memcpy(&mf, &s->hflags[f], sizeof(double));
        return auto_handover(s, first, f, first + count, negval, stat_scale, renorm, mf);
      }
      s->refft_pending = nullptr;   // belongs to a day behind f
      guided = false;
      PS_TRY(fwd2d(s, s->recs[PS_REC_CHAIN][f], 0, s->N, map_plain(s->N, s->Pf), map_plain(s->N, s->Pf),
                   s->Ahat.p, 1, nullptr));
      if (d - f - 1 > 0)
        PS_HIP(hipMemsetAsync(s->padmax.p + f + 1, 0, (size_t)(d - f - 1) * sizeof(unsigned long long), s->stream));
      s->speculate = false;
      d = f + 1;
    }
  }
  if (s->speculate && flagged_at < 0) s->noflag_hint = count;
  if (!s->auto_exact && !s->speculate && count >= 2 && !s->cfg.no_flag_history) {
    if (s->hhist_n < count) {
      if (s->hist_ev) PS_HIP(hipEventSynchronize(s->hist_ev));   // an earlier run's copy may still be on its way
      if (s->hhist) (void)hipHostFree(s->hhist);
      s->hhist = nullptr;
      s->hhist_n = 0;
      PS_HIP(hipHostMalloc((void**)&s->hhist, (size_t)(count + 32) * sizeof(unsigned long long), hipHostMallocDefault));
      s->hhist_n = count + 32;
    }
    if (!s->hist_ev) PS_HIP(hipEventCreateWithFlags(&s->hist_ev, hipEventDisableTiming));
    PS_HIP(hipMemcpyAsync(s->hhist, s->padmax.p + first, (size_t)count * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    PS_HIP(hipEventRecord(s->hist_ev, s->stream));
    s->hist_first = first;
    s->hist_count = count;
  }
  return PS_OK;
}

extern "C" int ps_chain_stats(ps_solver* s, int first, int count, ps_day_stats* out) {
  if (!s || !out || first < 0 || count < 0 || first + count > s->nstat) return ps_fail(PS_ERR_BAD_ARG, "chain_stats: bad range");
  PS_HIP(hipSetDevice(s->device));
  if (s->auto_exact && s->auto_first >= 0 && first + count > s->auto_first) {
    // runs of days by owner: this solver (clean prefix), the wide helper, the fold child
    int d = first;
    while (d < first + count) {
      const int o = d < (int)s->owner.size() ? s->owner[d] : 0;
      int e = d + 1;
      while (e < first + count && (e < (int)s->owner.size() ? s->owner[e] : 0) == o) ++e;
      ps_solver* c = o == 1 ? s->wide : (o == 2 ? s->child : (o == 3 ? s->narrow : s));
      if (!c) return ps_fail(PS_ERR_STATE, "auto mode: day %d has no owner", d);
      if (c == s) {
        PS_TRY(finalize_days(s, d, e - d, s->last_renorm));
        PS_HIP(hipStreamSynchronize(s->stream));
        PS_HIP(hipMemcpy(out + (d - first), s->dstats.p + d, (size_t)(e - d) * sizeof(DayStats), hipMemcpyDeviceToHost));
      } else {
        PS_TRY(ps_chain_stats(c, d, e - d, out + (d - first)));
      }
      d = e;
    }
    return PS_OK;
  }
  PS_TRY(finalize_days(s, first, count, s->last_renorm));
  PS_HIP(hipStreamSynchronize(s->stream));
  static_assert(sizeof(ps_day_stats) == sizeof(DayStats), "stats layout");
  PS_HIP(hipMemcpy(out, s->dstats.p + first, (size_t)count * sizeof(DayStats), hipMemcpyDeviceToHost));
  return PS_OK;
}

__global__ void k_cmul_inplace(cplx* a, const cplx* b, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    a[i] = cmul(a[i], b[i]);
}

extern "C" int ps_solver_fftconv2_coo(ps_solver* s, const int32_t* row, const int32_t* col,
                                      const double* val, int64_t nnz, int kshape) {
  if (!s || nnz < 0) return ps_fail(PS_ERR_BAD_ARG, "fftconv2: bad arguments");
  if (s->mode == PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "fftconv2: PS_MODE_FOLD offers the chain API only");
  if (!s->have_state) return ps_fail(PS_ERR_STATE, "fftconv2 before set_state");
  if (kshape < 1 || kshape % 2 == 0) return ps_fail(PS_ERR_BAD_SHAPE, "kernel shape %d must be odd (CalcSol.py:58)", kshape);
  if (kshape > s->Pf) return ps_fail(PS_ERR_BAD_SHAPE, "kernel shape %d larger than the pad %d", kshape, s->Pf);
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(check_coo(row, col, nnz, kshape, "kernel"));
  PS_TRY(ensure_spectrum(s));
  PS_TRY(upload_coo(s, row, col, val, nnz));
  const int K = kshape, M = K / 2;
  const size_t spec = (size_t)s->Pf * s->ld;
  PS_TRY(s->kdense.ensure((size_t)K * K));
  PS_TRY(s->Bhat.ensure(spec));
  PS_TRY(ensure_temps(s, 1));
  s->bhat_first = -1;
  PS_HIP(hipMemsetAsync(s->kdense.p, 0, (size_t)K * K * sizeof(double), s->stream));
  PS_TRY(scatter_from_device(s, s->orow.p, s->ocol.p, s->oval.p, nnz, s->kdense.p, K, 0));
  PS_TRY(fwd2d(s, s->kdense.p, 0, K, map_wrap(M, s->Pf), map_wrap(M, s->Pf), s->Bhat.p, 1, nullptr));
  hipLaunchKernelGGL(k_cmul_inplace, dim3(2048), dim3(256), 0, s->stream, s->Ahat.p, s->Bhat.p, (int64_t)spec);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

extern "C" int ps_solver_get_cursol(ps_solver* s, double negval, double stat_scale, int renorm,
                                    ps_day_stats* stats) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  if (s->mode == PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "get_cursol: PS_MODE_FOLD offers the chain API only");
  if (!s->have_state) return ps_fail(PS_ERR_STATE, "get_cursol before set_state");
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(ensure_record(s, PS_REC_CHAIN, 0));
  PS_TRY(ensure_temps(s, 1));
  PS_TRY(ensure_spectrum(s));
  PS_HIP(hipMemsetAsync(s->padmax.p, 0, sizeof(unsigned long long), s->stream));
  double* rec = s->recs[PS_REC_CHAIN][0];
  PS_TRY(inv2d(s, s->Ahat.p, nullptr, nullptr, rec, 0, negval, stat_scale));
  s->last_renorm = renorm;
  PS_TRY(refft_if_flag(s, rec, s->Ahat.p, 0));
  if (stats) PS_TRY(ps_chain_stats(s, 0, 1, stats));
  return PS_OK;
}

extern "C" int ps_solver_back_solve(ps_solver* s, int nfilt, const int64_t* off, const int32_t* row,
                                    const int32_t* col, const double* val, double negval,
                                    double stat_scale, ps_day_stats* stats) {
  if (!s || nfilt < 0 || !off) return ps_fail(PS_ERR_BAD_ARG, "back_solve: bad arguments");
  if (s->mode == PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "back_solve: PS_MODE_FOLD offers the chain API only");
  if (!s->have_state) return ps_fail(PS_ERR_STATE, "back_solve before set_state");
  if (s->N % 2 == 0) return ps_fail(PS_ERR_BAD_SHAPE, "back_solve needs an odd domain (filters are N x N)");
  if (2 * (s->N / 2) + 1 > s->Pf) return ps_fail(PS_ERR_BAD_SHAPE, "filter larger than the pad");
  PS_HIP(hipSetDevice(s->device));
  const size_t spec = (size_t)s->Pf * s->ld;
  const int K = s->N, M = K / 2;
  PS_TRY(s->Chat.ensure(spec));
  PS_TRY(s->Bhat.ensure(spec));
  PS_TRY(s->kdense.ensure((size_t)K * K));
  PS_TRY(ensure_temps(s, 1));
  PS_TRY(ensure_stats(s, std::max(4, nfilt)));
  PS_TRY(ensure_spectrum(s));
  s->bhat_first = -1;
  // the back-solve statistics live in their own slots after the call; reuse slots [0,nfilt)
  PS_HIP(hipMemcpyAsync(s->Chat.p, s->Ahat.p, spec * sizeof(cplx), hipMemcpyDeviceToDevice, s->stream));
  PS_HIP(hipMemsetAsync(s->padmax.p, 0, (size_t)std::max(1, nfilt) * sizeof(unsigned long long), s->stream));
  const bool no_cache = s->cfg.no_filter_cache != 0;   // A/B knob
  // at most 32 cached filters, fewer when a spectrum is large (4 GB in total)
  const size_t kMaxFilt = std::max<size_t>(4, std::min<size_t>(32, ((size_t)4 << 30) / (spec * sizeof(cplx))));
  for (int i = nfilt - 1; i >= 0; --i) {
    const int64_t o = off[i], n = off[i + 1] - o;
    PS_TRY(check_coo(row + o, col + o, n, K, "filter"));
    PS_TRY(ensure_record(s, PS_REC_BACK, i));
    // content key of the filter: two independent 64-bit hashes over (row, col, val) + length
    ps_solver::FiltKey key{1469598103934665603ull, 0x9E3779B97F4A7C15ull, n};
    {
      auto mix = [&](const void* p, size_t bytes) {
        const unsigned char* b = (const unsigned char*)p;
        size_t q = 0;
        for (; q + 8 <= bytes; q += 8) {
          uint64_t w;
          __builtin_memcpy(&w, b + q, 8);
          key.h1 = (key.h1 ^ w) * 1099511628211ull;
          key.h2 = (key.h2 + w) * 0xD6E8FEB86659FD93ull;
          key.h2 ^= key.h2 >> 32;
        }
        for (; q < bytes; ++q) {
          key.h1 = (key.h1 ^ b[q]) * 1099511628211ull;
          key.h2 = (key.h2 + b[q]) * 0xD6E8FEB86659FD93ull;
        }
      };
      mix(row + o, (size_t)n * 4);
      mix(col + o, (size_t)n * 4);
      mix(val + o, (size_t)n * 8);
    }
    int slot = -1;
    if (!no_cache)
      for (size_t q = 0; q < s->filt_keys.size(); ++q)
        if (s->filt_keys[q].h1 == key.h1 && s->filt_keys[q].h2 == key.h2 && s->filt_keys[q].n == n) slot = (int)q;
    const cplx* B = nullptr;
    if (slot >= 0) {
      ++s->filt_hits;
      B = s->Fhat.p + (size_t)slot * spec;
      // what fwd2d_partial leaves behind for the fused pass that consumes B
      s->kt_direct = false;
      s->kt_live = (s->split && !s->tpipe) ? RowLive{0, {0, 0, 0, 0}, nullptr} : RowLive{1, map_wrap(M, s->Pf), nullptr};
    } else {
      ++s->filt_misses;
      cplx* dst = s->Bhat.p;
      if (!no_cache) {
        if (s->filt_keys.size() >= kMaxFilt) s->filt_keys.clear();
        // growing Fhat keeps the cached spectra (DevBuf::ensure would drop them): size it once
        if (!s->Fhat.p) PS_TRY(s->Fhat.ensure(spec * kMaxFilt));
        dst = s->Fhat.p + s->filt_keys.size() * spec;
        s->filt_keys.push_back(key);
      }
      PS_TRY(upload_coo(s, row + o, col + o, val + o, n));
      PS_HIP(hipMemsetAsync(s->kdense.p, 0, (size_t)K * K * sizeof(double), s->stream));
      PS_TRY(scatter_from_device(s, s->orow.p, s->ocol.p, s->oval.p, n, s->kdense.p, K, 0));
      PS_TRY(fwd2d_partial(s, s->kdense.p, 0, K, map_wrap(M, s->Pf), map_wrap(M, s->Pf), dst, 1));
      B = dst;
    }
    double* rec = s->recs[PS_REC_BACK][i];
    PS_TRY(conv_inv(s, B, s->Chat.p, 1, rec, i, negval, stat_scale));
    PS_TRY(refft_if_flag(s, rec, s->Chat.p, i));  // cuda_lib.py:208-214 semantics
  }
  s->last_renorm = 0;
  if (stats && nfilt > 0) PS_TRY(ps_chain_stats(s, 0, nfilt, stats));
  return PS_OK;
}

// ---- get_populations with a multi-day release (r_dur > 1) on the chain API --------------------
// CalcSol.py:296-323 / cuda_lib.py:145-221.  The uploaded kernel list is [day kernels ..., filters ...]:
// its last `nfilt` entries are the release days' spreads r_spread[0 .. nfilt-1] in chronological order,
// cut to their support box about the centre (odd shapes like any kernel; an N x N filter wrapped about
// its centre lands on the torus exactly where this kernel does, CalcSol.py:86-91).
//   count > 0: for every day d of [first, first + count): the cohort of the last release day moves one
//     day on (state *= K_d, inverse, truncate + re-transform on its flag, CalcSol.py:311-316), then the
//     back-solve -- a copy of the state's spectrum times filter nuse-1, inverse, flag / re-transform
//     (cuda_lib.py:208-214 semantics), times filter nuse-2, ... -- and the day's population
//     sum_i w[i] back_i + w[nuse] cohort  ->  chain record d  (CalcSol.py:322)
//   count == 0: the back-solve alone, from the state as it stands (a release day, CalcSol.py:298-306):
//     sum_i w[i] back_i + w[nuse] state field  ->  record (PS_REC_WSUM, 0)
// Everything is enqueued; no host round trip per day.  *certified (optional): PS_MODE_AUTO -- 1 when
// no field of the run (cohort or back-solve) had anything above 1e-15 outside the domain, i.e. the fast
// torus held exactly what the reference's torus holds; 0 tells the caller to redo the run on the exact
// torus.  Other modes: 1.
extern "C" int ps_chain_run_release(ps_solver* s, int first, int count, double negval, int nfilt, int nuse,
                                    const double* weights, int* certified) {
  if (!s || !weights || nfilt < 0 || nuse < 0 || nuse > nfilt || count < 0 || first < 0)
    return ps_fail(PS_ERR_BAD_ARG, "run_release: bad arguments");
  if (s->mode == PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "run_release: not offered in PS_MODE_FOLD");
  if (!s->have_state) return ps_fail(PS_ERR_STATE, "run_release before set_state");
  const int ndays = s->nk - nfilt;          // day kernels [0, ndays), filters [ndays, nk)
  if (!s->kernels_on_device || ndays < 0 || first + count > ndays)
    return ps_fail(PS_ERR_STATE, "run_release: days [%d,%d) + %d filters not uploaded (nk=%d)", first, first + count, nfilt, s->nk);
  PS_HIP(hipSetDevice(s->device));
  if (certified) *certified = 1;
  const size_t spec = (size_t)s->Pf * s->ld;
  const int per = nuse + 1;                  // fields per unit of work: the cohort's + nuse back-solves
  const int units = std::max(1, count);
  // statistics slots: [0, nk) the population days, then one slot per (unit, field) so that every pad
  // maximum of the run survives until the end (the auto-mode certificate reads them all)
  const int slot0 = std::max(4, s->nk);
  PS_TRY(ensure_stats(s, slot0 + units * per));
  PS_TRY(ensure_temps(s, 1));
  PS_TRY(s->Chat.ensure(spec));
  PS_TRY(ensure_record(s, PS_REC_STATE, 1));                       // the cohort's field of the day
  for (int i = 0; i < nuse; ++i) PS_TRY(ensure_record(s, PS_REC_BACK, i));
  for (int d = first; d < first + count; ++d) PS_TRY(ensure_record(s, PS_REC_CHAIN, d));
  if (count == 0) PS_TRY(ensure_record(s, PS_REC_WSUM, 0));
  PS_HIP(hipMemsetAsync(s->padmax.p + slot0, 0, (size_t)units * per * sizeof(unsigned long long), s->stream));
  if (count > 0) PS_HIP(hipMemsetAsync(s->padmax.p + first, 0, (size_t)count * sizeof(unsigned long long), s->stream));
  s->last_renorm = 0;
  PS_TRY(resolve_refft(s));
  if (s->tpipe_ok && !s->spec_valid) set_pipeline(s, true);
  PS_TRY(ensure_spectrum(s));
  // transforms: the run's day kernels and the filters side by side in one buffer, ONE format
  // (direct_chunk = 0: no direct-sum route, the two halves of the buffer are consumed alike)
  // (long runs go chunk by chunk like ps_chain_run; the filters are transformed again with every chunk)
  const int cdays = std::max(1, s->chunk_days - nuse);
  // the weighted sum's pointer / weight tables are the same every day
  {
    std::vector<const double*> ptrs(per);
    for (int i = 0; i < nuse; ++i) ptrs[i] = s->recs[PS_REC_BACK][i];
    ptrs[nuse] = count > 0 ? s->recs[PS_REC_STATE][1] : s->recs[PS_REC_STATE][0];
    PS_TRY(s->wptr.ensure(per));
    PS_TRY(s->wval.ensure(per));
    PS_HIP(hipMemcpyAsync(s->wptr.p, ptrs.data(), per * sizeof(double*), hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipMemcpyAsync(s->wval.p, weights, per * sizeof(double), hipMemcpyHostToDevice, s->stream));
    PS_HIP(hipStreamSynchronize(s->stream));   // host temporaries
  }
  const int64_t tot = (int64_t)s->N * s->N;
  int c0 = 0, cn = 0;                        // days [c0, c0 + cn) of the run have their spectra in the buffer
  for (int u = 0; u < units; ++u) {
    const int d = first + u, sl = slot0 + u * per;
    if (u >= c0 + cn) {
      c0 = u;
      cn = count > 0 ? std::min(cdays, count - c0) : 0;
      if (cn > 0) PS_TRY(transform_kernels(s, first + c0, cn, 0, cn + nuse, 0));
      if (nuse > 0) PS_TRY(transform_kernels(s, ndays, nuse, cn, cn + nuse, 0));
      s->bhat_first = -1;                    // the buffer is not a plain run of days: nothing to reuse
      s->bhat_count = 0;
      if (count == 0) cn = 1;                // the single back-solve unit
    }
    if (count > 0) {
      const cplx* B = s->Bhat.p + (size_t)(u - c0) * spec;
      double* rec = s->recs[PS_REC_STATE][1];
      PS_TRY(conv_inv(s, B, s->Ahat.p, 1, rec, sl + nuse, negval, 1.0, s->krange.p + 2 * d));
      PS_TRY(refft_if_flag(s, rec, s->Ahat.p, sl + nuse));
    }
    if (nuse > 0) PS_HIP(hipMemcpyAsync(s->Chat.p, s->Ahat.p, spec * sizeof(cplx), hipMemcpyDeviceToDevice, s->stream));
    for (int i = nuse - 1; i >= 0; --i) {
      const cplx* F = s->Bhat.p + (size_t)((count > 0 ? cn : 0) + i) * spec;
      double* rec = s->recs[PS_REC_BACK][i];
      PS_TRY(conv_inv(s, F, s->Chat.p, 1, rec, sl + i, negval, 1.0, s->krange.p + 2 * (ndays + i)));
      PS_TRY(refft_if_flag(s, rec, s->Chat.p, sl + i));
    }
    double* out = count > 0 ? s->recs[PS_REC_CHAIN][d] : s->recs[PS_REC_WSUM][0];
    hipLaunchKernelGGL(k_weighted_sum, dim3(2048), dim3(256), 0, s->stream, (const double* const*)s->wptr.p,
                       s->wval.p, per, tot, out);
    PS_HIP(hipGetLastError());
    if (count > 0) {   // the day's statistics (r_small_vals on the population, no renormalisation)
      hipLaunchKernelGGL(k_row_stats, dim3(s->N), dim3(256), 0, s->stream, out, s->N, 1.0, negval,
                         s->rowsum.p + (int64_t)d * s->N, s->rowcnt.p + (int64_t)d * s->N);
      PS_HIP(hipGetLastError());
    }
  }
  s->noflag_hint = 0;
  s->hist_count = 0;
  if (s->auto_exact) {
    // the certificate: every pad maximum of the run below kCleanEps (the inverse row pass of an auto
    // front publishes maxima above 0.5e-15)
    std::vector<unsigned long long> h((size_t)units * per);
    PS_HIP(hipMemcpyAsync(h.data(), s->padmax.p + slot0, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    PS_HIP(hipStreamSynchronize(s->stream));
    int ok = 1;
    for (unsigned long long v : h) {
      double m;
      __builtin_memcpy(&m, &v, sizeof(double));
      if (!(m < kCleanEps)) ok = 0;
    }
    if (certified) *certified = ok;
    s->auto_first = -1;
  }
  return PS_OK;
}

// ------------------------------------------------------------------ one simulation over G GPUs
// While no day raises the boundary flag the chain is a product of spectra, A_d = A_0 K_1 ... K_d, and products
// re-associate: the rank that owns days [first, first + count) forms L_i = K_first ... K_{first+i} on its own
// (ps_chain_block_prefix), the ranks exchange their block totals L_{count-1} (ONE all-gather of a spectrum per
// rank, parallel.chain_prefix_split), and every rank finishes its own days as A_0 (T_0 ... T_{g-1}) L_i
// (ps_chain_block_finish).  Same transforms as the chain (kernel row pass, full column transform, inverse
// column pass, inverse row pass with the usual epilogue); the PRODUCTS are taken in another order, so a day's
// field differs from the sequential chain's in the last bits (<= a few 1e-16 of its maximum per day of the run;
// tests/test_prefix_split_gpu.py), and a flag anywhere voids the split: *flagged tells, the caller reruns with
// ps_chain_run.  Full-column pipeline only (fast mode on a register-resident size).
extern "C" int ps_chain_block_prefix(ps_solver* s, int first, int count, const void** total_dev, int64_t* total_bytes) {
  if (!s || first < 0 || count < 1 || !total_dev) return ps_fail(PS_ERR_BAD_ARG, "block_prefix: bad arguments");
  if (s->mode != PS_MODE_FAST || !s->tpipe_ok)
    return ps_fail(PS_ERR_UNSUPPORTED, "block_prefix: fast mode on a register-resident FFT size only (this solver: mode %d, size %d)", s->mode, s->Pf);
  if (!s->have_state) return ps_fail(PS_ERR_STATE, "block_prefix before set_state");
  if (!s->kernels_on_device || first + count > s->nk)
    return ps_fail(PS_ERR_STATE, "block_prefix: days [%d,%d) not uploaded (nk=%d)", first, first + count, s->nk);
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(resolve_refft(s));
  if (!s->spec_valid) set_pipeline(s, true);
  if (!s->tpipe) return ps_fail(PS_ERR_UNSUPPORTED, "block_prefix: the state's spectrum is held for the tiled pipeline; set the state again");
  PS_TRY(ensure_spectrum(s));
  const size_t spec = (size_t)s->Pf * s->ld;
  const int64_t n = (int64_t)s->H * s->Pf;        // the column-major half spectrum [H][Pf]
  PS_TRY(transform_kernels(s, first, count, 0, count, 0));
  s->bhat_first = -1;                              // row-pass outputs only: nothing a later chain run could reuse as is
  s->bhat_count = 0;
  PS_TRY(s->blk.ensure(spec * count));
  RowLive live = s->kt_live;
  live.range = s->krange.p + 2 * first;
  PS_TRY(launch_colfull(s, 1, s->Bhat.p, s->blk.p, 0, nullptr, count, live, nullptr));
  for (int i = 1; i < count; ++i) {
    hipLaunchKernelGGL(k_cmul_inplace, dim3(2048), dim3(256), 0, s->stream, s->blk.p + (size_t)i * spec,
                       s->blk.p + (size_t)(i - 1) * spec, n);
    PS_HIP(hipGetLastError());
  }
  s->blk_first = first;
  s->blk_count = count;
  *total_dev = s->blk.p + (size_t)(count - 1) * spec;
  if (total_bytes) *total_bytes = (int64_t)(spec * sizeof(cplx));
  PS_HIP(hipStreamSynchronize(s->stream));        // the caller hands the block total to a collective on another stream
  return PS_OK;
}

extern "C" int ps_chain_block_finish(ps_solver* s, int first, int count, int nprev, const void* const* prev_totals,
                                     double negval, double stat_scale, int renorm, int* flagged) {
  if (!s || nprev < 0 || (nprev > 0 && !prev_totals)) return ps_fail(PS_ERR_BAD_ARG, "block_finish: bad arguments");
  if (s->blk_first != first || s->blk_count != count || count < 1)
    return ps_fail(PS_ERR_STATE, "block_finish: days [%d,%d) were not prepared by block_prefix", first, first + count);
  PS_HIP(hipSetDevice(s->device));
  const size_t spec = (size_t)s->Pf * s->ld;
  const int64_t n = (int64_t)s->H * s->Pf;
  PS_TRY(ensure_stats(s, std::max(4, s->nk)));
  for (int d = first; d < first + count; ++d) PS_TRY(ensure_record(s, PS_REC_CHAIN, d));
  PS_HIP(hipMemsetAsync(s->padmax.p + first, 0, (size_t)count * sizeof(unsigned long long), s->stream));
  // what came before this block: A_0 T_0 ... T_{g-1}, the earlier blocks in their order
  PS_TRY(s->Chat.ensure(spec));
  PS_HIP(hipMemcpyAsync(s->Chat.p, s->Ahat.p, spec * sizeof(cplx), hipMemcpyDeviceToDevice, s->stream));
  for (int j = 0; j < nprev; ++j) {
    if (!prev_totals[j]) return ps_fail(PS_ERR_BAD_ARG, "block_finish: block total %d is null", j);
    hipLaunchKernelGGL(k_cmul_inplace, dim3(2048), dim3(256), 0, s->stream, s->Chat.p, (const cplx*)prev_totals[j], n);
    PS_HIP(hipGetLastError());
  }
  for (int i = 0; i < count; ++i) {
    hipLaunchKernelGGL(k_cmul_inplace, dim3(2048), dim3(256), 0, s->stream, s->blk.p + (size_t)i * spec, s->Chat.p, n);
    PS_HIP(hipGetLastError());
  }
  s->blk_first = -1;                               // the products are spent
  s->blk_count = 0;
  PS_TRY(s->T1.ensure(spec * count));
  PS_TRY(launch_colfull(s, 2, nullptr, s->blk.p, 0, s->T1.p, count, RowLive{0, {0, 0, 0, 0}, nullptr}, nullptr));
  if (row_inv_persistent(s) && !s->tinv && count <= PS_MAX_GROUP_DAYS && !s->cfg.no_row_batch) {
    std::vector<double*> recs(count);
    for (int i = 0; i < count; ++i) recs[i] = s->recs[PS_REC_CHAIN][first + i];
    PS_TRY(launch_row_inv(s, s->T1.p, nullptr, first, count, negval, stat_scale, false, recs.data(), nullptr));
  } else {
    for (int i = 0; i < count; ++i)
      PS_TRY(launch_row_inv(s, s->T1.p + (size_t)i * spec, s->recs[PS_REC_CHAIN][first + i], first + i, 1, negval, stat_scale));
  }
  s->last_renorm = renorm;
  PS_TRY(finalize_days(s, first, count, renorm));
  // the state after the block's last day, as a chain run would leave it
  PS_HIP(hipMemcpyAsync(s->Ahat.p, s->blk.p + (size_t)(count - 1) * spec, spec * sizeof(cplx), hipMemcpyDeviceToDevice, s->stream));
  s->noflag_hint = 0;
  s->hist_count = 0;
  std::vector<unsigned long long> h((size_t)count);
  PS_HIP(hipMemcpyAsync(h.data(), s->padmax.p + first, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  int any = 0;
  for (unsigned long long v : h) {
    double m;
    __builtin_memcpy(&m, &v, sizeof(double));
    if (!(m <= 1e-8)) any = 1;                     // NaN counts as flagged
  }
  if (flagged) *flagged = any;
  return PS_OK;
}

// device-to-device copy between this library's buffers and a caller's (a tensor of the collective)
extern "C" int ps_device_copy(void* dst, const void* src, int64_t bytes) {
  if (!dst || !src || bytes < 0) return ps_fail(PS_ERR_BAD_ARG, "device_copy: bad arguments");
  PS_HIP(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice));
  return PS_OK;
}

// ------------------------------------------------------------------ records
static int get_record(ps_solver* s, int kind, int idx, double** out) {
  if (kind < 0 || kind > 3 || idx < 0 || idx >= (int)s->recs[kind].size() || !s->recs[kind][idx])
    return ps_fail(PS_ERR_STATE, "record (%d,%d) does not exist", kind, idx);
  *out = s->recs[kind][idx];
  return PS_OK;
}

extern "C" int ps_record_stats(ps_solver* s, int kind, int idx, double negval, double stat_scale,
                               int renorm, ps_day_stats* out) {
  if (!s || !out) return ps_fail(PS_ERR_BAD_ARG, "record_stats: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  double* rec;
  PS_TRY(get_record(s, kind, idx, &rec));
  PS_TRY(ensure_stats(s, 4));
  const int slot = s->nstat;  // scratch slot
  PS_HIP(hipMemsetAsync(s->padmax.p + slot, 0, sizeof(unsigned long long), s->stream));
  hipLaunchKernelGGL(k_row_stats, dim3(s->N), dim3(256), 0, s->stream, rec, s->N, stat_scale, negval,
                     s->rowsum.p + (int64_t)slot * s->N, s->rowcnt.p + (int64_t)slot * s->N);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_day_finalize, dim3(1), dim3(256), 0, s->stream, s->rowsum.p + (int64_t)slot * s->N,
                     s->rowcnt.p + (int64_t)slot * s->N, s->padmax.p + slot, s->N, renorm,
                     s->dstats.p + slot);
  PS_HIP(hipGetLastError());
  PS_HIP(hipStreamSynchronize(s->stream));
  PS_HIP(hipMemcpy(out, s->dstats.p + slot, sizeof(DayStats), hipMemcpyDeviceToHost));
  return PS_OK;
}

// shared body of the COO and CSR fetches: statistics, row offsets, ordered compaction
static int fetch_sparse(ps_solver* s, int kind, int idx, double negval, double stat_scale, double delta,
                        double post_scale, int32_t* row, int32_t* indptr, int32_t* col, double* val,
                        int64_t cap, int64_t* nnz_out) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  PS_HIP(hipSetDevice(s->device));
  double* rec;
  PS_TRY(get_record(s, kind, idx, &rec));
  PS_TRY(ensure_stats(s, 4));
  const int slot = s->nstat;  // scratch slot
  double* rs = s->rowsum.p + (int64_t)slot * s->N;
  long long* rc = s->rowcnt.p + (int64_t)slot * s->N;
  hipLaunchKernelGGL(k_row_stats, dim3(s->N), dim3(256), 0, s->stream, rec, s->N, stat_scale, negval, rs, rc);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, s->stream, rc, s->N, s->rowoff.p);
  PS_HIP(hipGetLastError());
  PS_HIP(hipStreamSynchronize(s->stream));
  long long last_off = 0, last_cnt = 0;
  PS_HIP(hipMemcpy(&last_off, s->rowoff.p + (s->N - 1), sizeof(long long), hipMemcpyDeviceToHost));
  PS_HIP(hipMemcpy(&last_cnt, rc + (s->N - 1), sizeof(long long), hipMemcpyDeviceToHost));
  const int64_t nnz = last_off + last_cnt;
  if (nnz_out) *nnz_out = nnz;
  if ((!row && !indptr) || !col || !val) return PS_OK;  // count only
  if (cap < nnz) return ps_fail(PS_ERR_BAD_ARG, "fetch: capacity %lld < nnz %lld", (long long)cap, (long long)nnz);
  if (indptr) {
    if (nnz > 0x7fffffffLL) return ps_fail(PS_ERR_UNSUPPORTED, "fetch_csr: %lld entries do not fit int32 offsets", (long long)nnz);
    std::vector<long long> off((size_t)s->N);
    PS_HIP(hipMemcpy(off.data(), s->rowoff.p, (size_t)s->N * sizeof(long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < s->N; ++i) indptr[i] = (int32_t)off[(size_t)i];
    indptr[s->N] = (int32_t)nnz;
  }
  if (nnz == 0) return PS_OK;
  PS_TRY(s->orow.ensure(nnz));
  PS_TRY(s->ocol.ensure(nnz));
  PS_TRY(s->oval.ensure(nnz));
  const int thr = 256;
  const int blocks = (s->N * 64 + thr - 1) / thr;
  hipLaunchKernelGGL(k_compact_rows, dim3(blocks), dim3(thr), 0, s->stream, rec, s->N, stat_scale, negval,
                     delta, post_scale, s->rowoff.p, s->orow.p, s->ocol.p, s->oval.p);
  PS_HIP(hipGetLastError());
  if (row) PS_HIP(hipMemcpyAsync(row, s->orow.p, nnz * 4, hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipMemcpyAsync(col, s->ocol.p, nnz * 4, hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipMemcpyAsync(val, s->oval.p, nnz * 8, hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  return PS_OK;
}

extern "C" int ps_record_fetch_coo(ps_solver* s, int kind, int idx, double negval, double stat_scale,
                                   double delta, double post_scale, int32_t* row, int32_t* col,
                                   double* val, int64_t cap, int64_t* nnz_out) {
  return fetch_sparse(s, kind, idx, negval, stat_scale, delta, post_scale, row, nullptr, col, val, cap, nnz_out);
}

extern "C" int ps_record_fetch_csr(ps_solver* s, int kind, int idx, double negval, double stat_scale,
                                   double delta, double post_scale, int32_t* indptr, int32_t* indices,
                                   double* data, int64_t cap, int64_t* nnz_out) {
  return fetch_sparse(s, kind, idx, negval, stat_scale, delta, post_scale, nullptr, indptr, indices, data, cap, nnz_out);
}

extern "C" int ps_record_fetch_dense(ps_solver* s, int kind, int idx, double* out) {
  if (!s || !out) return ps_fail(PS_ERR_BAD_ARG, "fetch_dense: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  double* rec;
  PS_TRY(get_record(s, kind, idx, &rec));
  PS_HIP(hipMemcpyAsync(out, rec, (size_t)s->N * s->N * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  return PS_OK;
}

extern "C" int ps_record_gather(ps_solver* s, int kind, int idx, int64_t n, const int32_t* rows,
                                const int32_t* cols, double scale, double negval, double* out) {
  if (!s || n < 0 || (n > 0 && (!rows || !cols || !out))) return ps_fail(PS_ERR_BAD_ARG, "record_gather: bad arguments");
  if (n == 0) return PS_OK;
  PS_HIP(hipSetDevice(s->device));
  double* rec;
  PS_TRY(get_record(s, kind, idx, &rec));
  PS_TRY(check_coo(rows, cols, n, s->N, "gather point"));
  PS_TRY(s->orow.ensure(n));
  PS_TRY(s->ocol.ensure(n));
  PS_TRY(s->oval.ensure(n));
  PS_HIP(hipMemcpyAsync(s->orow.p, rows, n * 4, hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipMemcpyAsync(s->ocol.p, cols, n * 4, hipMemcpyHostToDevice, s->stream));
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(k_gather_points, dim3(blocks), dim3(256), 0, s->stream, rec, s->N, s->orow.p, s->ocol.p, n,
                     scale, negval, s->oval.p);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(out, s->oval.p, n * 8, hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  return PS_OK;
}

extern "C" int ps_record_gather_multi(ps_solver* s, int nrec, const int32_t* kind, const int32_t* idx, int64_t n,
                                      const int32_t* rows, const int32_t* cols, double scale, double negval,
                                      double* out) {
  if (!s || nrec < 0 || n < 0 || (nrec > 0 && (!kind || !idx)) || (n > 0 && nrec > 0 && (!rows || !cols || !out)))
    return ps_fail(PS_ERR_BAD_ARG, "record_gather_multi: bad arguments");
  if (n == 0 || nrec == 0) return PS_OK;
  PS_HIP(hipSetDevice(s->device));
  std::vector<const double*> ptrs(nrec);
  for (int r = 0; r < nrec; ++r) {
    double* rec;
    PS_TRY(get_record(s, kind[r], idx[r], &rec));
    ptrs[r] = rec;
  }
  PS_TRY(check_coo(rows, cols, n, s->N, "gather point"));
  PS_TRY(s->orow.ensure(n));
  PS_TRY(s->ocol.ensure(n));
  PS_TRY(s->oval.ensure((size_t)n * nrec));
  PS_TRY(s->wptr.ensure(nrec));
  PS_HIP(hipMemcpyAsync(s->orow.p, rows, n * 4, hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipMemcpyAsync(s->ocol.p, cols, n * 4, hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipMemcpyAsync(s->wptr.p, ptrs.data(), nrec * sizeof(double*), hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));   // ptrs is a host temporary
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(k_gather_points_multi, dim3(blocks, nrec), dim3(256), 0, s->stream,
                     (const double* const*)s->wptr.p, s->N, s->orow.p, s->ocol.p, n, scale, negval, s->oval.p);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemcpyAsync(out, s->oval.p, (size_t)n * nrec * 8, hipMemcpyDeviceToHost, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  return PS_OK;
}

extern "C" int ps_weighted_sum(ps_solver* s, int n, const int32_t* kind, const int32_t* idx, const double* w) {
  if (!s || n < 1 || !kind || !idx || !w) return ps_fail(PS_ERR_BAD_ARG, "weighted_sum: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  std::vector<const double*> ptrs(n);
  for (int d = 0; d < n; ++d) {
    double* r;
    PS_TRY(get_record(s, kind[d], idx[d], &r));
    ptrs[d] = r;
  }
  PS_TRY(ensure_record(s, PS_REC_WSUM, 0));
  PS_TRY(s->wptr.ensure(n));
  PS_TRY(s->wval.ensure(n));
  PS_HIP(hipMemcpyAsync(s->wptr.p, ptrs.data(), n * sizeof(double*), hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipMemcpyAsync(s->wval.p, w, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
  PS_HIP(hipStreamSynchronize(s->stream));
  const int64_t tot = (int64_t)s->N * s->N;
  hipLaunchKernelGGL(k_weighted_sum, dim3(2048), dim3(256), 0, s->stream, (const double* const*)s->wptr.p,
                     s->wval.p, n, tot, s->recs[PS_REC_WSUM][0]);
  PS_HIP(hipGetLastError());
  return PS_OK;
}

// ----------------------------------------------------------------- spectrum
extern "C" int ps_solver_get_spectrum(ps_solver* s, double* out) {
  if (!s || !out) return ps_fail(PS_ERR_BAD_ARG, "get_spectrum: bad arguments");
  if (s->mode == PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "get_spectrum: PS_MODE_FOLD offers the chain API only");
  if (!s->have_state) return ps_fail(PS_ERR_STATE, "get_spectrum before set_state");
  PS_HIP(hipSetDevice(s->device));
  if (s->tpipe) {   // the P x P spectrum interface is row-major: tiled pipeline
    if (s->spec_valid) return ps_fail(PS_ERR_UNSUPPORTED, "get_spectrum: the state is held column-major (full-column pipeline)");
    set_pipeline(s, false);
  }
  PS_TRY(ensure_spectrum(s));
  const size_t full = (size_t)s->Pf * s->Pf;
  DevBuf<cplx> tmp;
  PS_TRY(tmp.ensure(full));
  hipLaunchKernelGGL(k_expand_spectrum, dim3(2048), dim3(256), 0, s->stream, s->Ahat.p, s->Pf, s->H, s->ld, tmp.p);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, tmp.p, full * sizeof(cplx), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  tmp.release();
  if (e != hipSuccess) return ps_fail(PS_ERR_HIP, "get_spectrum: %s", hipGetErrorString(e));
  return PS_OK;
}

extern "C" int ps_solver_set_spectrum(ps_solver* s, const double* in) {
  if (!s || !in) return ps_fail(PS_ERR_BAD_ARG, "set_spectrum: bad arguments");
  if (s->mode == PS_MODE_FOLD) return ps_fail(PS_ERR_UNSUPPORTED, "set_spectrum: PS_MODE_FOLD offers the chain API only");
  PS_HIP(hipSetDevice(s->device));
  set_pipeline(s, false);   // row-major half spectrum
  const size_t full = (size_t)s->Pf * s->Pf;
  DevBuf<cplx> tmp;
  PS_TRY(tmp.ensure(full));
  hipError_t e = hipMemcpyAsync(tmp.p, in, full * sizeof(cplx), hipMemcpyHostToDevice, s->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_take_half_spectrum, dim3(2048), dim3(256), 0, s->stream, tmp.p, s->Pf, s->H, s->ld, s->Ahat.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  tmp.release();
  if (e != hipSuccess) return ps_fail(PS_ERR_HIP, "set_spectrum: %s", hipGetErrorString(e));
  s->have_state = true;
  s->spec_valid = true;
  return PS_OK;
}

// ------------------------------------------------------------------- options
extern "C" int ps_solver_set_option(ps_solver* s, const char* key, double value) {
  if (!s || !key) return ps_fail(PS_ERR_BAD_ARG, "set_option: bad arguments");
  const int rc = ps_config_set(&s->cfg, key, value, true);
  if (rc == -1) return ps_fail(PS_ERR_BAD_ARG, "set_option: unknown option %s", key);
  if (rc == -2) return ps_fail(PS_ERR_STATE, "set_option: %s takes effect when a solver is created", key);
  s->tinv = s->cfg.tinv != 0;
  s->fused_days = s->cfg.fused_days;
  // the helpers of an auto-mode front follow their parent
  if (s->child) PS_TRY(ps_solver_set_option(s->child, key, value));
  if (s->wide) PS_TRY(ps_solver_set_option(s->wide, key, value));
  if (s->narrow) PS_TRY(ps_solver_set_option(s->narrow, key, value));
  return PS_OK;
}

extern "C" int ps_solver_get_option(ps_solver* s, const char* key, double* value) {
  if (!s || !key || !value) return ps_fail(PS_ERR_BAD_ARG, "get_option: bad arguments");
  if (ps_config_get(&s->cfg, key, value) != 0) return ps_fail(PS_ERR_BAD_ARG, "get_option: unknown option %s", key);
  return PS_OK;
}

// ----------------------------------------------------------------- profiling
static int prof_drain(ps_solver* s) {
  PS_HIP(hipStreamSynchronize(s->stream));
  for (auto& r : s->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      s->prof_ms[r.cls] += ms;
      s->prof_cnt[r.cls] += 1;
      s->prof_days[r.cls] += r.days;
    }
    s->prof_pool.push_back(r.a);
    s->prof_pool.push_back(r.b);
  }
  s->prof_pending.clear();
  return PS_OK;
}

extern "C" int ps_prof_enable(ps_solver* s, int on) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(prof_drain(s));
  for (int i = 0; i < PS_PROF_NCLS; ++i) { s->prof_ms[i] = 0; s->prof_cnt[i] = 0; s->prof_seen[i] = 0; s->prof_days[i] = 0; }
  s->prof_on = on != 0;
  s->prof_every = on > 1 ? on : 1;   // on = n > 1: time every n-th launch of each class
  for (ps_solver* c : {s->wide, s->child, s->narrow})   // the helpers of an auto-mode front follow it
    if (c) PS_TRY(ps_prof_enable(c, on));
  return PS_OK;
}

extern "C" int ps_prof_read(ps_solver* s, int ncls, double* total_ms, int64_t* count) {
  if (!s || !total_ms || !count) return ps_fail(PS_ERR_BAD_ARG, "prof_read: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(prof_drain(s));
  for (int i = 0; i < ncls; ++i) {
    total_ms[i] = i < PS_PROF_NCLS ? s->prof_ms[i] : 0.0;
    count[i] = i < PS_PROF_NCLS ? s->prof_cnt[i] : 0;
  }
  return PS_OK;
}

extern "C" int ps_prof_read_days(ps_solver* s, int ncls, int64_t* days) {
  if (!s || !days) return ps_fail(PS_ERR_BAD_ARG, "prof_read_days: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(prof_drain(s));
  for (int i = 0; i < ncls; ++i) days[i] = i < PS_PROF_NCLS ? s->prof_days[i] : 0;
  return PS_OK;
}

extern "C" int ps_prof_read_launches(ps_solver* s, int ncls, int64_t* launches) {
  if (!s || !launches) return ps_fail(PS_ERR_BAD_ARG, "prof_read_launches: bad arguments");
  PS_HIP(hipSetDevice(s->device));
  PS_TRY(prof_drain(s));
  // the multi-day classes are all timed; of the others every prof_every-th launch is, prof_seen counts all
  for (int i = 0; i < ncls; ++i)
    launches[i] = i >= PS_PROF_NCLS ? 0 : (i >= PS_PROF_COL_INV_A2 ? s->prof_cnt[i] : s->prof_seen[i]);
  return PS_OK;
}

extern "C" int ps_prof_read_owner(ps_solver* s, int owner, int ncls, double* total_ms, int64_t* count,
                                  int64_t* launches, int64_t* days) {
  if (!s || owner < 0 || owner > 3 || !total_ms || !count || !launches || !days)
    return ps_fail(PS_ERR_BAD_ARG, "prof_read_owner: bad arguments");
  ps_solver* c = owner == 1 ? s->wide : (owner == 2 ? s->child : (owner == 3 ? s->narrow : s));
  if (!c) {   // a helper this run never needed
    for (int i = 0; i < ncls; ++i) { total_ms[i] = 0.0; count[i] = 0; launches[i] = 0; days[i] = 0; }
    return PS_OK;
  }
  PS_TRY(ps_prof_read(c, ncls, total_ms, count));
  PS_TRY(ps_prof_read_launches(c, ncls, launches));
  return ps_prof_read_days(c, ncls, days);
}

extern "C" int ps_solver_kernels_direct(ps_solver* s) { return s && s->kt_direct ? 1 : 0; }

extern "C" int ps_solver_pipeline(ps_solver* s) { return s && s->tpipe ? 1 : 0; }

extern "C" int ps_solver_auto_route(ps_solver* s, int first, int count, int32_t* owner) {
  if (!s || !owner || first < 0 || count < 0) return ps_fail(PS_ERR_BAD_ARG, "auto_route: bad arguments");
  for (int i = 0; i < count; ++i) {
    const int d = first + i;
    owner[i] = (s->auto_exact && s->auto_first >= 0 && d >= s->auto_first && d < (int)s->owner.size()) ? s->owner[d] : 0;
  }
  return PS_OK;
}

extern "C" int ps_solver_auto_info(ps_solver* s, int* first_fold_day, int* fold_fft) {
  if (!s) return ps_fail(PS_ERR_BAD_ARG, "null solver");
  if (first_fold_day) *first_fold_day = s->auto_exact ? s->auto_first : -1;
  if (fold_fft) *fold_fft = s->child ? s->child->Pf : 0;
  return PS_OK;
}

extern "C" int ps_solver_owner_fft(ps_solver* s, int owner) {
  if (!s) return 0;
  ps_solver* c = owner == 1 ? s->wide : (owner == 2 ? s->child : (owner == 3 ? s->narrow : (owner == 0 ? s : nullptr)));
  return c ? c->Pf : 0;
}

int ps_solver_dom_len_internal(ps_solver* s) { return s->N; }
int ps_solver_device_internal(ps_solver* s) { return s->device; }


