// libparasitoid_hip.so -- per-day probability-mass kernel construction on the device
// (C ABI in include/parasitoid_hip.h).  Replaces ParasitoidModel.prob_mass
// (ParasitoidModel.py:384-613) and what it calls: h_flight_prob (:282-309),
// get_mvn_cdf_values (:311-380) with the bivariate-normal rectangle probabilities of
// scipy.stats.mvn.mvnun (Genz BVU), r_small_vals (CalcSol.py:112-136).
#include <math.h>
#include <string.h>

#include "chain_kernels.h"
#include "model_kernels.h"
#include "ps_common.h"
#include "ps_config.h"

struct ps_model {
  int device = 0;
  ps_config cfg;   // PS_PM_SEG / PS_PM_SYNC as ps_model_create found them, then ps_model_set_option
  hipStream_t stream = nullptr;
  // wind
  DevBuf<double> wind;
  DevBuf<int> day_keys;
  int ndw = 0, T = 0, test_run = 0;
  // last batch
  int nd = 0, N = 0, R = 0;
  DevBuf<int> day_idx;
  DevBuf<double> start_time, hprob, scratch, pmf, psum, pmin, rowsum;
  DevBuf<long long> rowcnt, rowoff, doff, tcnt, toff;
  DevBuf<int> pair_t, pair_tile;
  DevBuf<double> hm;
  DevBuf<long long> np_dev;          // pair count of the batch in flight
  long long* np_host = nullptr;      // pinned copy
  long long last_np = 0;             // pair count of the previous batch (sizes the next one's lists without a host sync)
  long long guessed_cap = -1;        // capacity the batch in flight was enqueued with (-1: exact, host-read offsets)
  long long guess_key = -1, max_np = 0;   // batch shape (days, grid) the counts belong to; largest count seen for it
  DevBuf<int> rowrad;
  DevBuf<PeriodInfo> pinfo;
  DevBuf<DayInfo> dinfo;
  std::vector<DayInfo> hinfo;
  // results: concatenated COO of the batch
  DevBuf<int> orow, ocol;
  DevBuf<double> oval;
  std::vector<int64_t> off;
  std::vector<int> kshape;
  // scratch for mvn_cdf_values
  DevBuf<double> stamp;
  DevBuf<int> stampH;
};

// Gauss-Legendre half rules of Genz's BVU (published with MVNDST/TVPACK)
static const double kW1[3] = {0.1713244923791705, 0.3607615730481384, 0.4679139345726904};
static const double kX1[3] = {-0.9324695142031522, -0.6612093864662647, -0.2386191860831970};
static const double kW2[6] = {0.4717533638651177e-01, 0.1069393259953183, 0.1600783285433464,
                              0.2031674267230659, 0.2334925365383547, 0.2491470458134029};
static const double kX2[6] = {-0.9815606342467191, -0.9041172563704750, -0.7699026741943050,
                              -0.5873179542866171, -0.3678314989981802, -0.1252334085114692};
static const double kW3[10] = {0.1761400713915212e-01, 0.4060142980038694e-01, 0.6267204833410906e-01,
                               0.8327674157670475e-01, 0.1019301198172404, 0.1181945319615184,
                               0.1316886384491766, 0.1420961093183821, 0.1491729864726037,
                               0.1527533871307259};
static const double kX3[10] = {-0.9931285991850949, -0.9639719272779138, -0.9122344282513259,
                               -0.8391169718222188, -0.7463319064601508, -0.6360536807265150,
                               -0.5108670019508271, -0.3737060887154196, -0.2277858511416451,
                               -0.7652652113349733e-01};

// Dmat (ParasitoidModel.py:269-280) + mvnun's standardisation: marginal std devs and
// correlation, then the BVU rule for that correlation.
static int make_rule(double sig_x, double sig_y, double rho, double* sdx, double* sdy, BvuRule* R) {
  if (!(sig_x > 0)) return ps_fail(PS_ERR_BAD_ARG, "sig_x must be positive");
  if (!(sig_y > 0)) return ps_fail(PS_ERR_BAD_ARG, "sig_y must be positive");
  if (!(-1 <= rho && rho <= 1)) return ps_fail(PS_ERR_BAD_ARG, "correlation must be between -1 and 1");
  const double s00 = sig_x * sig_x, s11 = sig_y * sig_y, s01 = rho * sig_x * sig_y;
  *sdx = sqrt(s00);
  *sdy = sqrt(s11);
  const double r = s01 / *sdx / *sdy;
  memset(R, 0, sizeof *R);
  R->r = r;
  const double *w, *x;
  if (fabs(r) < 0.3) { R->lg = 3; w = kW1; x = kX1; }
  else if (fabs(r) < 0.75) { R->lg = 6; w = kW2; x = kX2; }
  else { R->lg = 10; w = kW3; x = kX3; }
  R->high = fabs(r) >= 0.925;
  R->asr = R->high ? 0.0 : asin(r);
  for (int i = 0; i < R->lg; ++i) {
    R->w[i] = w[i];
    R->x[i] = x[i];
    R->sn1[i] = sin(R->asr * (x[i] + 1) / 2);
    R->sn2[i] = sin(R->asr * (-x[i] + 1) / 2);
    R->iv1[i] = 1.0 / (1 - R->sn1[i] * R->sn1[i]);
    R->iv2[i] = 1.0 / (1 - R->sn2[i] * R->sn2[i]);
  }
  return PS_OK;
}

extern "C" int ps_model_create(ps_model** out, int device) {
  if (!out) return ps_fail(PS_ERR_BAD_ARG, "null output handle");
  *out = nullptr;
  PS_TRY(ps_use_device(device));
  PS_HIP(hipFuncSetAttribute((const void*)k_day_prep, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  PS_HIP(hipFuncSetAttribute((const void*)k_hprob, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  ps_model* m = new ps_model();
  ps_config_from_env(&m->cfg);   // the only look at the environment a model handle takes
  m->device = device;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete m;
    return ps_fail(PS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  *out = m;
  return PS_OK;
}

extern "C" int ps_model_set_option(ps_model* m, const char* key, double value) {
  if (!m || !key) return ps_fail(PS_ERR_BAD_ARG, "model set_option: bad arguments");
  if (ps_config_set(&m->cfg, key, value, false) != 0) return ps_fail(PS_ERR_BAD_ARG, "model set_option: unknown option %s", key);
  return PS_OK;
}

extern "C" int ps_model_destroy(ps_model* m) {
  if (!m) return PS_OK;
  (void)hipSetDevice(m->device);
  if (m->stream) {
    (void)hipStreamSynchronize(m->stream);
    (void)hipStreamDestroy(m->stream);
  }
  ps_dev_quiesce();
  m->wind.release(); m->day_keys.release(); m->day_idx.release(); m->start_time.release();
  m->hprob.release(); m->scratch.release(); m->pmf.release(); m->psum.release(); m->pmin.release();
  m->rowsum.release(); m->rowcnt.release(); m->rowoff.release(); m->doff.release(); m->rowrad.release();
  m->tcnt.release(); m->toff.release(); m->pair_t.release(); m->pair_tile.release(); m->hm.release();
  m->np_dev.release();
  if (m->np_host) { (void)hipHostFree(m->np_host); m->np_host = nullptr; }
  m->pinfo.release(); m->dinfo.release(); m->orow.release(); m->ocol.release(); m->oval.release();
  m->stamp.release(); m->stampH.release();
  delete m;
  return PS_OK;
}

extern "C" int ps_model_set_wind(ps_model* m, const double* wind, const int32_t* day_keys,
                                 int ndays_wind, int T, int test_run) {
  if (!m || !wind || !day_keys || ndays_wind < 1 || T < 1) return ps_fail(PS_ERR_BAD_ARG, "set_wind: bad arguments");
  if (test_run && T != 1) return ps_fail(PS_ERR_BAD_ARG, "test_run wind must have one period");
  if ((size_t)T * (sizeof(PeriodInfo) + sizeof(double)) > 120 * 1024) return ps_fail(PS_ERR_UNSUPPORTED, "more than %d periods per day", (int)(120 * 1024 / (sizeof(PeriodInfo) + sizeof(double))));
  PS_HIP(hipSetDevice(m->device));
  const size_t n = (size_t)ndays_wind * T * 3;
  PS_TRY(m->wind.ensure(n));
  PS_TRY(m->day_keys.ensure(ndays_wind));
  PS_HIP(hipMemcpyAsync(m->wind.p, wind, n * sizeof(double), hipMemcpyHostToDevice, m->stream));
  PS_HIP(hipMemcpyAsync(m->day_keys.p, day_keys, ndays_wind * sizeof(int), hipMemcpyHostToDevice, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  m->ndw = ndays_wind;
  m->T = T;
  m->test_run = test_run;
  m->nd = 0;
  return PS_OK;
}

// the pair stage: the instance for the rule's branch and number of Gauss-Legendre node pairs
// (|rho| < 0.3: 3 node pairs as a compile-time constant, see pm_bvu_low_phi_t; 6, 10 and the
// |rho| >= 0.925 branch read the count at run time).  PMC (profiles/r04_prob_mass_pair_kernel_pmc.txt): the
// kernel issues 4.06e8 wave-instructions per 18-day batch at R = 400, nearly all of them fp64 at 4 cycles
// each on a SIMD's 16 fp64 lanes -- 1.59e6 cycles per SIMD of the 1.7e6 the launch lasts: it is at the
// vector ISSUE limit (SQ_ACTIVE_INST_VALU / SQ_BUSY 0.93), not latency-bound; the unrolled instance halves
// the scalar traffic and the waiting (SQ_WAIT_ANY 3.1e8 -> 1.8e8) and gains 1 %, because waiting was not
// what the time went to.  Average active lanes 71 % (partly filled corner iterations, the erfc phase on
// half a wave).
static void launch_pair_masses(ps_model* m, const ModelParams& mp, unsigned blocks, hipStream_t st, int d0, int nt,
                               long long np, int seg, const long long* np_dev) {
#define PS_PM_LAUNCH(...) \
  hipLaunchKernelGGL((k_pair_masses<__VA_ARGS__>), dim3(blocks), dim3(256), 0, st, mp, m->pinfo.p, d0, nt, np, seg, np_dev, \
                     m->pair_t.p, m->pair_tile.p, m->hm.p)
  if (mp.rule.high) PS_PM_LAUNCH(true, 0);
  else if (mp.rule.lg == 3 && !m->cfg.pm_no_unroll) PS_PM_LAUNCH(false, 3);
  else PS_PM_LAUNCH(false, 0);   // (six node pairs unrolled: measured slower, 1.69 against 1.65 ms per 18-day batch)
#undef PS_PM_LAUNCH
}

static int prob_mass_impl(ps_model* m, int nd, const int32_t* day_idx, const double* start_time,
                          const double* hparams, const double* Dparams, const double* Dlparams,
                          double mu_r, int n_periods, double rad_dist, int rad_res,
                          int32_t* kshape, int64_t* nnz, int32_t* warned, int32_t* status, bool allow_guess);

extern "C" int ps_model_prob_mass(ps_model* m, int nd, const int32_t* day_idx, const double* start_time,
                                  const double* hparams, const double* Dparams, const double* Dlparams,
                                  double mu_r, int n_periods, double rad_dist, int rad_res,
                                  int32_t* kshape, int64_t* nnz, int32_t* warned, int32_t* status) {
  return prob_mass_impl(m, nd, day_idx, start_time, hparams, Dparams, Dlparams, mu_r, n_periods, rad_dist, rad_res,
                        kshape, nnz, warned, status, true);
}

static int prob_mass_impl(ps_model* m, int nd, const int32_t* day_idx, const double* start_time,
                          const double* hparams, const double* Dparams, const double* Dlparams,
                          double mu_r, int n_periods, double rad_dist, int rad_res,
                          int32_t* kshape, int64_t* nnz, int32_t* warned, int32_t* status, bool allow_guess) {
  if (!m || nd < 1 || !day_idx || !start_time || !hparams || !Dparams || !Dlparams)
    return ps_fail(PS_ERR_BAD_ARG, "prob_mass: bad arguments");
  if (m->ndw == 0) return ps_fail(PS_ERR_STATE, "prob_mass before set_wind");
  if (rad_res < 1 || !(rad_dist > 0) || n_periods < 1) return ps_fail(PS_ERR_BAD_ARG, "prob_mass: bad domain/flight parameters");
  for (int i = 0; i < nd; ++i)
    if (day_idx[i] < 0 || day_idx[i] >= m->ndw) return ps_fail(PS_ERR_BAD_ARG, "day index %d out of range", day_idx[i]);
  PS_HIP(hipSetDevice(m->device));
  ModelParams mp;
  memset(&mp, 0, sizeof mp);
  for (int i = 0; i < 7; ++i) mp.hp[i] = hparams[i];
  PS_TRY(make_rule(Dparams[0], Dparams[1], Dparams[2], &mp.sdx, &mp.sdy, &mp.rule));
  PS_TRY(make_rule(Dlparams[0], Dlparams[1], Dlparams[2], &mp.lsdx, &mp.lsdy, &mp.lrule));
  mp.mu_r = mu_r;
  mp.rad_dist = rad_dist;
  mp.cell = rad_dist / rad_res;  // ParasitoidModel.py:408
  mp.n_periods = n_periods;
  mp.rad_res = rad_res;
  mp.N = 2 * rad_res + 1;
  mp.T = m->T;
  mp.test_run = m->test_run;
  mp.ndays_wind = m->ndw;
  const int N = mp.N, T = m->T;
  const int64_t n2 = (int64_t)N * N;
  const int nblk = 64;
  m->nd = 0;
  PS_TRY(m->day_idx.ensure(nd));
  PS_TRY(m->start_time.ensure(nd));
  PS_TRY(m->hprob.ensure((size_t)nd * T));
  PS_TRY(m->scratch.ensure((size_t)nd * 3 * T));
  PS_TRY(m->pinfo.ensure((size_t)nd * T));
  PS_TRY(m->dinfo.ensure(nd));
  PS_TRY(m->pmf.ensure((size_t)nd * n2));
  PS_TRY(m->psum.ensure((size_t)nd * nblk));
  PS_TRY(m->pmin.ensure((size_t)nd * nblk));
  PS_TRY(m->rowsum.ensure((size_t)nd * N));
  PS_TRY(m->rowcnt.ensure((size_t)nd * N));
  PS_TRY(m->rowrad.ensure((size_t)nd * N));
  PS_TRY(m->rowoff.ensure(N));
  hipStream_t st = m->stream;
  PS_HIP(hipMemcpyAsync(m->day_idx.p, day_idx, nd * sizeof(int), hipMemcpyHostToDevice, st));
  PS_HIP(hipMemcpyAsync(m->start_time.p, start_time, nd * sizeof(double), hipMemcpyHostToDevice, st));
  // (no synchronisation: copies from pageable host memory are staged before hipMemcpyAsync returns)
  hipLaunchKernelGGL(k_hprob, dim3(nd), dim3(256), (size_t)3 * T * sizeof(double), st, m->wind.p, mp, m->day_idx.p,
                     m->hprob.p, m->scratch.p);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_periods, dim3((T * 16 + 255) / 256, nd), dim3(256), 0, st, m->wind.p, m->day_keys.p, mp,
                     m->day_idx.p, m->start_time.p, m->hprob.p, m->pinfo.p);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_day_prep, dim3(nd), dim3(256), (size_t)T * (sizeof(PeriodInfo) + sizeof(double)), st, mp, m->start_time.p,
                     m->pinfo.p, m->dinfo.p, m->day_idx.p);
  PS_HIP(hipGetLastError());
  PS_HIP(hipMemsetAsync(m->pmf.p, 0, (size_t)nd * n2 * sizeof(double), st));
  // stamp accumulation (model_kernels.h): ordered (tile, period) pair lists, one wave per pair
  // for the masses, ordered per-tile accumulation; days are processed in chunks that keep the
  // pair records (2 KB each) within ~2 GB
  const int nt = (N + PM_TS - 1) / PM_TS;
  {
    const int64_t ntl = (int64_t)nt * nt, ntot = ntl * nd;
    PS_TRY(m->tcnt.ensure((size_t)ntot));
    PS_TRY(m->toff.ensure((size_t)ntot));
    hipLaunchKernelGGL(k_tile_count, dim3(nt, nt, nd), dim3(256), 0, st, mp, m->pinfo.p, m->dinfo.p, m->tcnt.p);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, st, m->tcnt.p, (int)ntot, m->toff.p);
    PS_HIP(hipGetLastError());
    const int seg = std::max(1, m->cfg.pm_seg);   // periods per record
    const long long max_pairs = (long long)1 << 20;   // 2 GB of records per chunk
    // The pair lists' length is known on the device only.  A batch like the previous one (sampler
    // chains, ensemble members: same days, nearby parameters) sizes its lists from that one's count
    // plus a quarter and enqueues everything without waiting for the scan; the real count comes back
    // with the batch's statistics, and a batch that did not fit is redone the slow way (host reads the
    // day offsets, chunks of at most 2 GB of records).  PS_PM_SYNC=1: always the slow way.
    // (capacity: the previous batch's count plus a quarter, and never less than the largest count a
    // batch of this shape has had -- an ensemble's members differ a lot, the first large one is redone
    // once and sets the size for the rest)
    const long long key = (long long)nd * 1000003LL + N;
    if (key != m->guess_key) { m->guess_key = key; m->max_np = 0; m->last_np = 0; }
    const long long cap = std::max(m->last_np + m->last_np / 4, m->max_np) + 4096;
    bool guessed = false;
    if (allow_guess && m->last_np > 0 && cap <= max_pairs && !m->cfg.pm_sync) {
      guessed = true;
      PS_TRY(m->np_dev.ensure(1));
      if (!m->np_host) PS_HIP(hipHostMalloc((void**)&m->np_host, sizeof(long long), hipHostMallocDefault));
      PS_TRY(m->pair_t.ensure((size_t)cap));
      PS_TRY(m->pair_tile.ensure((size_t)cap));
      PS_TRY(m->hm.ensure((size_t)cap * PM_CELLS));
      hipLaunchKernelGGL(k_pair_total, dim3(1), dim3(64), 0, st, m->toff.p, m->tcnt.p, (long long)ntot, m->np_dev.p);
      PS_HIP(hipGetLastError());
      PS_HIP(hipMemcpyAsync(m->np_host, m->np_dev.p, sizeof(long long), hipMemcpyDeviceToHost, st));
      hipLaunchKernelGGL(k_tile_fill, dim3(nt, nt, nd), dim3(256), 0, st, mp, m->pinfo.p, m->dinfo.p, m->tcnt.p,
                         m->toff.p, 0, 0LL, cap, m->pair_t.p, m->pair_tile.p);
      PS_HIP(hipGetLastError());
      const long long nseg = (cap + seg - 1) / seg;
      launch_pair_masses(m, mp, (unsigned)((nseg + 3) / 4), st, 0, nt, cap, seg, m->np_dev.p);
      PS_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_tile_accumulate, dim3(nt, nt, nd), dim3(PM_CELLS), 0, st, mp, m->tcnt.p, m->toff.p, 0,
                         0LL, seg, cap, m->hm.p, m->pmf.p);
      PS_HIP(hipGetLastError());
    }
    if (!guessed) {
    std::vector<long long> dayoff((size_t)nd + 1, 0);
    long long last_cnt = 0;
    // the first offset of every day in one strided copy
    PS_HIP(hipMemcpy2DAsync(dayoff.data(), sizeof(long long), m->toff.p, (size_t)ntl * sizeof(long long),
                            sizeof(long long), (size_t)nd, hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(&dayoff[(size_t)nd], m->toff.p + (ntot - 1), sizeof(long long), hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(&last_cnt, m->tcnt.p + (ntot - 1), sizeof(long long), hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    dayoff[(size_t)nd] += last_cnt;
    m->last_np = dayoff[(size_t)nd];
    m->max_np = std::max(m->max_np, m->last_np);
    int d0 = 0;
    while (d0 < nd) {
      int d1 = d0 + 1;
      while (d1 < nd && dayoff[(size_t)d1 + 1] - dayoff[(size_t)d0] <= max_pairs) ++d1;
      const long long base = dayoff[(size_t)d0], np = dayoff[(size_t)d1] - base;
      if (np > 0) {
        PS_TRY(m->pair_t.ensure((size_t)np));
        PS_TRY(m->pair_tile.ensure((size_t)np));
        PS_TRY(m->hm.ensure((size_t)np * PM_CELLS));
        hipLaunchKernelGGL(k_tile_fill, dim3(nt, nt, d1 - d0), dim3(256), 0, st, mp, m->pinfo.p, m->dinfo.p, m->tcnt.p,
                           m->toff.p, d0, base, np, m->pair_t.p, m->pair_tile.p);
        PS_HIP(hipGetLastError());
        // periods per record (PS_PM_SEG; 1 = one record per (tile, period) pair: sequential-loop sums)
        const long long nseg = (np + seg - 1) / seg;
        launch_pair_masses(m, mp, (unsigned)((nseg + 3) / 4), st, d0, nt, np, seg, nullptr);
        PS_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_tile_accumulate, dim3(nt, nt, d1 - d0), dim3(PM_CELLS), 0, st, mp, m->tcnt.p, m->toff.p, d0,
                           base, seg, np, m->hm.p, m->pmf.p);
        PS_HIP(hipGetLastError());
      }
      d0 = d1;
    }
    }   // !guessed
    m->guessed_cap = guessed ? cap : -1;
  }
  hipLaunchKernelGGL(k_pmf_reduce1, dim3(nblk, nd), dim3(256), 0, st, m->pmf.p, n2, m->psum.p, m->pmin.p);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_day_local, dim3(nd), dim3(256), 0, st, mp, m->psum.p, m->pmin.p, nblk, m->dinfo.p, m->pmf.p, 0);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_pmf_reduce1, dim3(nblk, nd), dim3(256), 0, st, m->pmf.p, n2, m->psum.p, m->pmin.p);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_day_local, dim3(nd), dim3(256), 0, st, mp, m->psum.p, m->pmin.p, nblk, m->dinfo.p, m->pmf.p, 1);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_pmf_row_stats, dim3(N, nd), dim3(256), 0, st, m->pmf.p, N, rad_res, 1e-8, m->rowsum.p,
                     m->rowcnt.p, m->rowrad.p);
  PS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_pmf_day_stats, dim3(nd), dim3(256), 0, st, m->rowsum.p, m->rowcnt.p, m->rowrad.p, N, m->dinfo.p);
  PS_HIP(hipGetLastError());
  m->hinfo.resize(nd);
  PS_HIP(hipMemcpyAsync(m->hinfo.data(), m->dinfo.p, nd * sizeof(DayInfo), hipMemcpyDeviceToHost, st));
  PS_HIP(hipStreamSynchronize(st));
  if (m->guessed_cap >= 0) {
    const long long np_real = *m->np_host;
    m->last_np = np_real;
    m->max_np = std::max(m->max_np, np_real);
    if (np_real > m->guessed_cap)      // the lists were too short: everything behind them is void
      return prob_mass_impl(m, nd, day_idx, start_time, hparams, Dparams, Dlparams, mu_r, n_periods, rad_dist, rad_res,
                            kshape, nnz, warned, status, false);
  }
  m->off.assign(nd + 1, 0);
  m->kshape.assign(nd, 0);
  for (int d = 0; d < nd; ++d) {
    const DayInfo& di = m->hinfo[d];
    const bool ok = di.status == 0;
    m->off[d + 1] = m->off[d] + (ok ? di.nnz : 0);
    m->kshape[d] = ok ? 2 * di.rad + 1 : 0;
    if (kshape) kshape[d] = m->kshape[d];
    if (nnz) nnz[d] = ok ? di.nnz : 0;
    if (warned) warned[d] = di.warned;
    if (status) status[d] = di.status;
  }
  const int64_t tot = m->off[nd];
  PS_TRY(m->orow.ensure(std::max<int64_t>(tot, 1)));
  PS_TRY(m->ocol.ensure(std::max<int64_t>(tot, 1)));
  PS_TRY(m->oval.ensure(std::max<int64_t>(tot, 1)));
  if (tot > 0) {   // one scan and one compaction launch for the whole batch of days
    PS_TRY(m->rowoff.ensure((size_t)nd * N));
    PS_TRY(m->doff.ensure((size_t)nd + 1));
    {
      std::vector<long long> o64(m->off.begin(), m->off.end());
      PS_HIP(hipMemcpyAsync(m->doff.p, o64.data(), o64.size() * sizeof(long long), hipMemcpyHostToDevice, st));
      PS_HIP(hipStreamSynchronize(st));   // o64 is a temporary
    }
    hipLaunchKernelGGL(k_scan_rows, dim3(nd), dim3(1024), 0, st, m->rowcnt.p, N, m->rowoff.p);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_pmf_compact_batch, dim3((N * 64 + 255) / 256, nd), dim3(256), 0, st, m->pmf.p, N, 1e-8,
                       rad_res, m->dinfo.p, m->doff.p, m->rowoff.p, m->orow.p, m->ocol.p, m->oval.p);
    PS_HIP(hipGetLastError());
  }
  PS_HIP(hipStreamSynchronize(st));
  m->nd = nd;
  m->N = N;
  m->R = rad_res;
  return PS_OK;
}

extern "C" int ps_model_fetch_coo(ps_model* m, int i, int32_t* row, int32_t* col, double* val, int64_t cap) {
  if (!m || !row || !col || !val) return ps_fail(PS_ERR_BAD_ARG, "fetch_coo: bad arguments");
  if (i < 0 || i >= m->nd) return ps_fail(PS_ERR_STATE, "fetch_coo: day %d not in the last batch", i);
  const int64_t n = m->off[i + 1] - m->off[i];
  if (cap < n) return ps_fail(PS_ERR_BAD_ARG, "fetch_coo: capacity %lld < nnz %lld", (long long)cap, (long long)n);
  if (n == 0) return PS_OK;
  PS_HIP(hipSetDevice(m->device));
  PS_HIP(hipMemcpyAsync(row, m->orow.p + m->off[i], n * 4, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipMemcpyAsync(col, m->ocol.p + m->off[i], n * 4, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipMemcpyAsync(val, m->oval.p + m->off[i], n * 8, hipMemcpyDeviceToHost, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));
  return PS_OK;
}

extern "C" int ps_model_fetch_debug(ps_model* m, int i, double* hprob, int32_t* Hs, double* loss, double* pmfsum) {
  if (!m) return ps_fail(PS_ERR_BAD_ARG, "null model");
  if (i < 0 || i >= m->nd) return ps_fail(PS_ERR_STATE, "fetch_debug: day %d not in the last batch", i);
  PS_HIP(hipSetDevice(m->device));
  const int T = m->T;
  if (hprob) PS_HIP(hipMemcpy(hprob, m->hprob.p + (int64_t)i * T, T * sizeof(double), hipMemcpyDeviceToHost));
  if (Hs) {
    std::vector<PeriodInfo> pi(T);
    PS_HIP(hipMemcpy(pi.data(), m->pinfo.p + (int64_t)i * T, T * sizeof(PeriodInfo), hipMemcpyDeviceToHost));
    for (int t = 0; t < T; ++t) Hs[t] = pi[t].skip ? -1 : pi[t].H;
  }
  if (loss) *loss = m->hinfo[i].loss;
  if (pmfsum) *pmfsum = m->hinfo[i].pmfsum;
  return PS_OK;
}

extern "C" int ps_model_hflight(ps_model* m, int day_i, const double* hparams, double* out) {
  if (!m || !hparams || !out) return ps_fail(PS_ERR_BAD_ARG, "hflight: bad arguments");
  if (m->ndw == 0) return ps_fail(PS_ERR_STATE, "hflight before set_wind");
  if (day_i < 0 || day_i >= m->ndw) return ps_fail(PS_ERR_BAD_ARG, "day index %d out of range", day_i);
  PS_HIP(hipSetDevice(m->device));
  ModelParams mp;
  memset(&mp, 0, sizeof mp);
  for (int i = 0; i < 7; ++i) mp.hp[i] = hparams[i];
  mp.T = m->T;
  const int T = m->T;
  DevBuf<double> hp, sc;
  DevBuf<int> di;
  int rc = hp.ensure(T);
  if (!rc) rc = sc.ensure((size_t)3 * T);
  if (!rc) rc = di.ensure(1);
  hipError_t e = hipSuccess;
  if (!rc) {
    e = hipMemcpy(di.p, &day_i, sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_hprob, dim3(1), dim3(256), (size_t)3 * T * sizeof(double), m->stream, m->wind.p, mp, di.p,
                         hp.p, sc.p);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e == hipSuccess) e = hipMemcpy(out, hp.p, T * sizeof(double), hipMemcpyDeviceToHost);
  }
  hp.release(); sc.release(); di.release();
  if (rc) return rc;
  if (e != hipSuccess) return ps_fail(PS_ERR_HIP, "hflight: %s", hipGetErrorString(e));
  return PS_OK;
}

extern "C" int ps_model_mvn_cdf_values(ps_model* m, double cell, double mu_x, double mu_y, double sig_x,
                                       double sig_y, double rho, int32_t* H, double* out, int64_t cap) {
  if (!m || !H || !(cell > 0)) return ps_fail(PS_ERR_BAD_ARG, "mvn_cdf_values: bad arguments");
  PS_HIP(hipSetDevice(m->device));
  BvuRule rule;
  double sdx, sdy;
  PS_TRY(make_rule(sig_x, sig_y, rho, &sdx, &sdy, &rule));
  const int64_t dcap = std::max<int64_t>(cap, 1);
  PS_TRY(m->stamp.ensure(dcap));
  PS_TRY(m->stampH.ensure(1));
  hipLaunchKernelGGL(k_mvn_cdf_values, dim3(1), dim3(256), 0, m->stream, rule, sdx, sdy, mu_x, mu_y, cell, 1 << 14,
                     m->stampH.p, m->stamp.p, (long long)(out ? cap : 0));
  PS_HIP(hipGetLastError());
  PS_HIP(hipStreamSynchronize(m->stream));
  int h = 0;
  PS_HIP(hipMemcpy(&h, m->stampH.p, sizeof(int), hipMemcpyDeviceToHost));
  *H = h;
  const int64_t need = (int64_t)(2 * h + 1) * (2 * h + 1);
  if (out && cap >= need) PS_HIP(hipMemcpy(out, m->stamp.p, need * sizeof(double), hipMemcpyDeviceToHost));
  return PS_OK;
}

extern "C" int ps_chain_set_kernels_from_model(ps_solver* s, ps_model* m, int first, int count) {
  if (!s || !m) return ps_fail(PS_ERR_BAD_ARG, "null handle");
  if (first < 0 || count < 0 || first + count > m->nd) return ps_fail(PS_ERR_STATE, "days [%d,%d) not in the last batch", first, first + count);
  if (ps_solver_device_internal(s) != m->device) return ps_fail(PS_ERR_BAD_ARG, "solver and model live on different devices");
  for (int d = first; d < first + count; ++d)
    if (m->hinfo[d].status != 0) return ps_fail(m->hinfo[d].status, "day %d of the batch failed its checks", d);
  std::vector<int64_t> off(count + 1);
  for (int d = 0; d <= count; ++d) off[d] = m->off[first + d] - m->off[first];
  return ps_chain_adopt_device_kernels(s, count, off.data(), m->kshape.data() + first, m->orow.p + m->off[first],
                                       m->ocol.p + m->off[first], m->oval.p + m->off[first]);
}

// ---- device-resident exchange of day kernels between ranks (days sharded over the GPUs, the
// COO triplets all-gathered with RCCL).  The library neither owns nor sees the communicator: the caller
// (parallel.prob_mass_sharded_device) hands in device buffers it allocated for the collective -- raw
// device pointers, e.g. torch tensors' data_ptr() -- and the kernels never pass through host memory.
extern "C" int ps_model_export_device(ps_model* m, int first, int count, void* row_dev, void* col_dev, void* val_dev,
                                      int64_t cap) {
  if (!m || !row_dev || !col_dev || !val_dev) return ps_fail(PS_ERR_BAD_ARG, "export_device: bad arguments");
  if (first < 0 || count < 0 || first + count > m->nd) return ps_fail(PS_ERR_STATE, "export_device: days [%d,%d) not in the last batch", first, first + count);
  for (int d = first; d < first + count; ++d)
    if (m->hinfo[d].status != 0) return ps_fail(m->hinfo[d].status, "day %d of the batch failed its checks", d);
  const int64_t o = m->off[first], n = m->off[first + count] - o;
  if (cap < n) return ps_fail(PS_ERR_BAD_ARG, "export_device: capacity %lld < %lld entries", (long long)cap, (long long)n);
  if (n == 0) return PS_OK;
  PS_HIP(hipSetDevice(m->device));
  PS_HIP(hipMemcpyAsync(row_dev, m->orow.p + o, n * 4, hipMemcpyDeviceToDevice, m->stream));
  PS_HIP(hipMemcpyAsync(col_dev, m->ocol.p + o, n * 4, hipMemcpyDeviceToDevice, m->stream));
  PS_HIP(hipMemcpyAsync(val_dev, m->oval.p + o, n * 8, hipMemcpyDeviceToDevice, m->stream));
  PS_HIP(hipStreamSynchronize(m->stream));   // the caller's collective runs on another stream
  return PS_OK;
}

// ps_chain_set_kernels with the triplets already in device memory (caller-owned; copied before return)
extern "C" int ps_chain_set_kernels_device(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                                           const void* row_dev, const void* col_dev, const void* val_dev) {
  if (!s || nk < 0 || !off || (nk > 0 && (!kshape || !row_dev || !col_dev || !val_dev)))
    return ps_fail(PS_ERR_BAD_ARG, "set_kernels_device: bad arguments");
  for (int d = 0; d < nk; ++d)
    if (off[d + 1] < off[d]) return ps_fail(PS_ERR_BAD_ARG, "offsets not monotone");
  PS_HIP(hipSetDevice(ps_solver_device_internal(s)));
  PS_TRY(ps_chain_adopt_device_kernels(s, nk, off, kshape, (const int*)row_dev, (const int*)col_dev, (const double*)val_dev));
  return ps_solver_sync(s);                  // the caller may reuse its buffers
}

// ps_solver_set_state_coo from device triplets of an odd kshape x kshape kernel, re-centred into the
// domain (Run.py:454-458: offset = rad_res - kshape // 2)
extern "C" int ps_solver_set_state_device(ps_solver* s, const void* row_dev, const void* col_dev, const void* val_dev,
                                          int64_t nnz, int kshape) {
  if (!s || nnz < 0 || (nnz > 0 && (!row_dev || !col_dev || !val_dev)) || kshape < 1 || kshape % 2 == 0)
    return ps_fail(PS_ERR_BAD_ARG, "set_state_device: bad arguments");
  const int N = ps_solver_dom_len_internal(s);
  if (kshape > N) return ps_fail(PS_ERR_BAD_SHAPE, "set_state_device: kernel %d larger than the domain %d", kshape, N);
  PS_HIP(hipSetDevice(ps_solver_device_internal(s)));
  PS_TRY(ps_solver_set_state_device_coo(s, (const int*)row_dev, (const int*)col_dev, (const double*)val_dev, nnz,
                                        N / 2 - kshape / 2));
  return ps_solver_sync(s);
}

extern "C" int ps_solver_set_state_from_model(ps_solver* s, ps_model* m, int i) {
  if (!s || !m) return ps_fail(PS_ERR_BAD_ARG, "null handle");
  if (i < 0 || i >= m->nd) return ps_fail(PS_ERR_STATE, "day %d not in the last batch", i);
  if (ps_solver_device_internal(s) != m->device) return ps_fail(PS_ERR_BAD_ARG, "solver and model live on different devices");
  if (m->hinfo[i].status != 0) return ps_fail(m->hinfo[i].status, "day %d of the batch failed its checks", i);
  const int N = ps_solver_dom_len_internal(s);
  if (N != m->N) return ps_fail(PS_ERR_BAD_SHAPE, "solver domain %d != model domain %d", N, m->N);
  // Run.py:454-458: offset = rad_res - K//2
  const int off = m->R - m->kshape[i] / 2;
  return ps_solver_set_state_device_coo(s, m->orow.p + m->off[i], m->ocol.p + m->off[i], m->oval.p + m->off[i],
                                        m->off[i + 1] - m->off[i], off);
}
