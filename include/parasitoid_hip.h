/* parasitoid_hip.h -- C ABI of libparasitoid_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the drift-diffusion forward solver of
 * mountaindust/Parasitoids.  Each entry point names the reference interface it
 * replaces (file:line relative to the reference repository).  Plain pointers and
 * sizes only; all inputs are caller-owned host buffers that are copied before
 * the call returns and never modified; outputs go to caller-allocated buffers
 * sized from a preceding count call (two-phase COO output).
 *
 * Every function returns PS_OK (0) or a negative error class; ps_last_error()
 * gives the message of the last failure on the calling thread.  One handle =
 * one device + one HIP stream; a handle is not thread-safe, different handles
 * are independent.  No HIP call is made at library load (fork-safe: the
 * reference forks a multiprocessing pool before it touches the GPU,
 * Run.py:422-425, Bayes_Run.py:706).
 */
#ifndef PARASITOID_HIP_H
#define PARASITOID_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_OK 0
#define PS_ERR_NO_DEVICE (-1)   /* no gfx950 device / HIP runtime unusable          */
#define PS_ERR_OOM (-2)         /* device allocation failed (cuda_lib.py:31,:72,:108,:159 asserts) */
#define PS_ERR_BAD_SHAPE (-3)   /* even kernel shape (CalcSol.py:58), kernel larger than the pad, indices out of range */
#define PS_ERR_BAD_ARG (-4)
#define PS_ERR_UNSUPPORTED (-5) /* FFT length not plannable (prime factor > 1024 or length > LDS) */
#define PS_ERR_HIP (-6)         /* HIP runtime error                                 */
#define PS_ERR_HPROB_BOUNDS (-7)/* ParasitoidModel.py:528-537  hprob out of bounds   */
#define PS_ERR_PMF_NEGATIVE (-8)/* ParasitoidModel.py:570,:589 pmf.min() < -1e-8     */
#define PS_ERR_FLIGHT_PROB (-9) /* ParasitoidModel.py:569,:571,:590 flight prob > 1 / negative loss */
#define PS_ERR_STATE (-10)      /* call order (e.g. fetch before run)                */
#define PS_ERR_EMPTY (-11)      /* prob_mass produced no entry >= 1e-8               */

#define PS_MODE_EXACT 0 /* FFT on the reference's torus P = N + K//2 (CalcSol.py:20-21) */
#define PS_MODE_FAST 1  /* FFT on the next even 7-smooth size >= P                       */
#define PS_MODE_FOLD 2  /* the reference's torus P, computed as a linear convolution on a fast
                         * FFT size >= P + K - 1 and folded back modulo P: exact-torus semantics
                         * at fast-size speed when P has large prime factors.  Chain API only
                         * (set_state, set_kernels, chain_run, records); the per-call
                         * fftconv2 / get_cursol / back_solve / spectrum calls return
                         * PS_ERR_UNSUPPORTED. */
#define PS_MODE_AUTO 3  /* exact reference-torus results by the cheapest route, chain API only: the
                         * day chain runs on the PS_MODE_FAST torus for as long as nothing above
                         * 1e-15 lies outside the N x N domain (then the two tori cannot differ by
                         * more than that per day: the pad holds no dust to wrap around).  Past
                         * that prefix, days that start from a domain-supported field and end
                         * flagged (largest value outside the domain > 1e-8) or clean run on a
                         * fast torus sized N + 2M, where they are true linear convolutions; the
                         * days in between -- sub-threshold dust the reference carries around its
                         * torus -- run as PS_MODE_FOLD.  ps_solver_auto_route tells which. */

typedef struct ps_solver ps_solver;
typedef struct ps_model ps_model;

typedef struct ps_day_stats {
  int64_t nnz;   /* entries with value*scale >= negval (r_small_vals, CalcSol.py:126-132) */
  double sum;    /* their sum                                                            */
  double delta;  /* (1-sum)/nnz added to every entry when renorm (CalcSol.py:134-135)    */
  double padmax; /* max over the pad region, clamped at 0 (CalcSol.py:36-37)             */
  int32_t flag;  /* padmax > 1e-8 -> state was truncated and re-transformed (:38,:200-201) */
  int32_t pad_;
} ps_day_stats;

/* ---- library ---------------------------------------------------------------- */
int ps_version(void);
int ps_device_count(void);             /* < 0 on error */
const char* ps_last_error(void);
/* device properties for reports: name[<=n], CU count, total HBM bytes */
int ps_device_info(int device, char* name, int n, int* cus, int64_t* hbm_bytes);

/* ---- solver: replaces class cuda_lib.CudaSolve (cuda_lib.py:16-221) and the
 *      CPU primitives CalcSol.fft2/fftconv2/ifft2/back_solve (CalcSol.py:11-109) */

/* CudaSolve.__init__ shape logic (cuda_lib.py:26-28): pad = dom_len + max_shape//2. */
int ps_solver_create(ps_solver** out, int device, int dom_len, int max_shape, int mode);
/* FFT size a PS_MODE_FAST solver would use for this domain and kernel shape (no device needed):
 * lets a caller size max_shape so that one solver serves a range of kernel shapes. */
int ps_fast_size(int dom_len, int max_shape);
/* PS_MODE_FOLD only: change the kernel shape limit (and with it the reference torus
 * P = N + max_shape/2) of an existing solver, keeping its FFT size, plans and buffers -- valid
 * while N + 3 (max_shape/2) fits the solver's FFT size (ps_solver_info).  The state has to be
 * set again afterwards. */
int ps_solver_retarget(ps_solver* s, int max_shape);
int ps_solver_destroy(ps_solver* s);
/* P = reference torus, Pfft = transform size in use, H = Pfft/2+1 */
int ps_solver_info(ps_solver* s, int* dom_len, int* P, int* Pfft, int* H);
int ps_solver_sync(ps_solver* s);
/* Tuning / A-B knobs.  The reference has one switch (globalvars.py:5: `cuda`); this library's
 * knobs (csrc/ps_config.h) exist for measurements and for the A/B legs of
 * the bit-identity tests.  Their names are those of the environment variables that seed them: the
 * environment is read once, when a handle is created, and never again -- afterwards a knob of a
 * live handle is changed with these calls (an auto-mode front passes it on to its helpers).
 * Creation-time knobs (register-resident kernels on/off, column split, chunk size ...) are refused
 * with PS_ERR_STATE, unknown names with PS_ERR_BAD_ARG. */
int ps_solver_set_option(ps_solver* s, const char* key, double value);
int ps_solver_get_option(ps_solver* s, const char* key, double* value);
/* measurement aid: 1 when the day kernels of the last transformed chunk were compact enough
 * for the direct-sum first column sub-pass inside the fused kernel (no separate launch) */
int ps_solver_kernels_direct(ps_solver* s);
/* measurement aid: 1 when the solver runs the full-column pipeline (column-major spectra, one
 * pass per column transform: register-resident FFT sizes in PS_MODE_FAST), 0 for the tiled
 * two-sub-pass column kernels */
int ps_solver_pipeline(ps_solver* s);
/* PS_MODE_AUTO: *first_fold_day = first chain day of the last ps_chain_run that ran on the folded
 * reference torus (-1: every day was clean and ran on the fast torus); *fold_fft = FFT size of the
 * fold path (0 if it was never needed).  Other modes: -1 / 0. */
int ps_solver_auto_info(ps_solver* s, int* first_fold_day, int* fold_fft);
/* PS_MODE_AUTO: which solver produced each day [first, first+count) of the last ps_chain_run:
 * 0 the fast-torus front (clean prefix), 1 the wide fast-torus helper (N + 2M: flagged and clean
 * days past the prefix), 2 the fold child (days whose sub-threshold dust the reference carries),
 * 3 the narrow fast-torus helper (N + M, like the front: days a previous run saw flagged above 4e-8,
 * which makes the reference's flag certain on any torus). */
int ps_solver_auto_route(ps_solver* s, int first, int count, int32_t* owner);

/* CudaSolve.__init__ (cuda_lib.py:34-54) / CalcSol.fft2 (CalcSol.py:11-24):
 * state_hat = FFT2(zero-padded N x N sparse field). */
int ps_solver_set_state_coo(ps_solver* s, const int32_t* row, const int32_t* col,
                            const double* val, int64_t nnz);

/* CudaSolve.fftconv2 (cuda_lib.py:58-94) / CalcSol.fftconv2 (CalcSol.py:45-66):
 * wrap the odd kshape x kshape kernel to the origin, FFT, state_hat *= B_hat. */
int ps_solver_fftconv2_coo(ps_solver* s, const int32_t* row, const int32_t* col,
                           const double* val, int64_t nnz, int kshape);

/* CudaSolve.get_cursol (cuda_lib.py:98-140) / CalcSol.ifft2 (+ re-FFT, CalcSol.py:28-41,
 * :200-201): inverse transform, boundary flag, truncate + re-transform when flagged.
 * The N x N field stays on the device as chain record 0.  Statistics are taken on
 * value*stat_scale >= negval; renorm != 0 computes the prob-model delta. */
int ps_solver_get_cursol(ps_solver* s, double negval, double stat_scale, int renorm,
                         ps_day_stats* stats);

/* CudaSolve.back_solve (cuda_lib.py:145-221) / CalcSol.back_solve (CalcSol.py:72-109):
 * nfilt N x N filters in chronological order, concatenated COO with offsets
 * off[nfilt+1].  Results stay on the device as back-solve records 0..nfilt-1 in
 * emergence order; stats[nfilt] optional. */
int ps_solver_back_solve(ps_solver* s, int nfilt, const int64_t* off, const int32_t* row,
                         const int32_t* col, const double* val, double negval,
                         double stat_scale, ps_day_stats* stats);

/* ---- whole day chain: replaces the loops of CalcSol.get_solutions
 *      (CalcSol.py:191-201) and CalcSol.get_populations (:308-323, r_dur == 1) ---- */

/* Upload nk day kernels (pmf_list entries, ParasitoidModel.prob_mass output): COO
 * concatenated with offsets off[nk+1]; kshape[d] odd.  All kernel transforms are
 * batched up front (they do not depend on the state). */
int ps_chain_set_kernels(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                         const int32_t* row, const int32_t* col, const double* val);
/* Run days [first, first+count) from the current state: per day
 * fftconv2 -> ifft2 -> statistics/flag -> (flagged) truncate + re-FFT, all enqueued on
 * the handle's stream.  The flag decisions are taken on the device; the call itself waits for
 * the device only where it verifies a window of days that it ran without the conditional
 * launches (it returns with the last window verified).  Day d's field is chain record d; the
 * statistics of every day run since the kernels were uploaded stay readable (ps_chain_stats).
 * A run may be continued (first > 0 right after a run that ended at first) in PS_MODE_EXACT / FAST /
 * FOLD.  PS_MODE_AUTO runs need a fresh state (ps_solver_set_state_*) before EVERY call: once a run
 * has handed days over to its helpers the front solver's spectrum is void, and whether that happened
 * depends on the data (PS_ERR_STATE "auto mode: set the state before every chain run" otherwise). */
int ps_chain_run(ps_solver* s, int first, int count, double negval, double stat_scale,
                 int renorm);
int ps_chain_stats(ps_solver* s, int first, int count, ps_day_stats* out); /* synchronises */

/* get_populations with a multi-day release, r_dur > 1 (CalcSol.py:296-323; CudaSolve.back_solve,
 * cuda_lib.py:145-221, with its re-FFT semantics :208-214) on the chain API.  The LAST `nfilt`
 * entries of the uploaded kernel list are the release days' spreads r_spread[0 .. nfilt-1] in
 * chronological order, each cut to its support box about the centre (an odd kernel; it lands on the
 * torus where the reference wraps the N x N filter, CalcSol.py:86-91); the entries before them are
 * the day kernels.
 *   count > 0: for each day d of [first, first+count): the last cohort moves one day on (state *=
 *     K_d, inverse, truncate + re-transform on its flag), the back-solve runs through filters
 *     nuse-1 .. 0 from a copy of the state's spectrum, and the population
 *     sum_{i<nuse} weights[i] back_i + weights[nuse] cohort  becomes chain record d (its statistics:
 *     ps_chain_stats, scale 1, no renormalisation).
 *   count == 0: the back-solve alone from the state as it stands (a release day, CalcSol.py:298-306):
 *     sum_{i<nuse} weights[i] back_i + weights[nuse] state  ->  record (3, 0).
 * Enqueued without a host round trip per day.  *certified (may be NULL): PS_MODE_AUTO -- 1 when no
 * field of the run had anything above 1e-15 outside the domain, so the fast torus held what the
 * reference's torus holds (<= 4e-15 per day); 0: redo the run in PS_MODE_EXACT.  Other modes: 1.
 * PS_MODE_FOLD: PS_ERR_UNSUPPORTED. */
int ps_chain_run_release(ps_solver* s, int first, int count, double negval, int nfilt, int nuse,
                         const double* weights, int* certified);

/* One simulation split over G GPUs by days -- SURVEY.md 8e, the flag-free special case (the reference has no
 * counterpart: CalcSol.py:140-201 is one sequential loop; while no day raises its boundary flag that loop is the
 * product A_d = A_0 K_1 ... K_d of spectra, and products re-associate).  Every rank holds the same state and
 * the kernels of (at least) its own days.
 *   ps_chain_block_prefix: the running products L_i = K_first ... K_{first+i} of days [first, first+count)
 *     (2-D spectra in the solver's own layout, kept in the solver); *total_dev = device pointer of the block
 *     total L_{count-1}, *total_bytes its size -- valid until the next block call on s.  The ranks exchange
 *     these (one all-gather; parasitoids_amd/parallel.py:chain_prefix_split).
 *   ps_chain_block_finish: chain records and statistics of the same days from A_0 T_0 ... T_{nprev-1} L_i
 *     (prev_totals: device pointers of the nprev EARLIER blocks' totals, in block order, on this device; same
 *     layout, i.e. solvers of the same dom_len / max_shape); negval / stat_scale / renorm as ps_chain_run;
 *     *flagged = 1 when one of these days raised the flag -- the split does not apply then, rerun with
 *     ps_chain_run.  The state afterwards is the spectrum after the block's last day.
 * Same transforms and epilogue as ps_chain_run, the spectral products in another order: a field differs from
 * the sequential chain's by rounding only (tests/test_prefix_split_gpu.py: tested at <= 1e-14 of the
 * day's maximum over 12 days in up to 5 blocks, measured 1e-19 absolute on the 30-day headline stack in 8).  PS_MODE_FAST on a register-resident FFT size (PS_ERR_UNSUPPORTED otherwise). */
int ps_chain_block_prefix(ps_solver* s, int first, int count, const void** total_dev, int64_t* total_bytes);
int ps_chain_block_finish(ps_solver* s, int first, int count, int nprev, const void* const* prev_totals,
                          double negval, double stat_scale, int renorm, int* flagged);
/* device-to-device copy between a buffer of this library and a caller's (e.g. a tensor of the collective) */
int ps_device_copy(void* dst, const void* src, int64_t bytes);

/* ---- records (device-resident N x N fields) ----
 * kind 0: chain/get_cursol records, 1: back_solve records, 2: state (first day),
 * 3: scratch result of ps_weighted_sum. */
#define PS_REC_CHAIN 0
#define PS_REC_BACK 1
#define PS_REC_STATE 2
#define PS_REC_WSUM 3
/* (re)compute statistics of any record with the given threshold/scale */
int ps_record_stats(ps_solver* s, int kind, int idx, double negval, double stat_scale,
                    int renorm, ps_day_stats* out);
/* r_small_vals + coo_matrix(dense) order (CalcSol.py:112-136): entries with
 * v*stat_scale >= negval, row-major; value = (v*stat_scale + delta) * post_scale.
 * cap must be >= the nnz reported by the matching stats call. */
int ps_record_fetch_coo(ps_solver* s, int kind, int idx, double negval, double stat_scale,
                        double delta, double post_scale, int32_t* row, int32_t* col,
                        double* val, int64_t cap, int64_t* nnz_out);
/* The same entries as CSR triplets (indptr[N+1], indices, data): what `.tocsr()` of the COO
 * result holds and what the reference's result files store per day (Run.py:490-516,
 * read back by Plot_Result.py:511-524) -- straight from the device compaction, no host sort. */
int ps_record_fetch_csr(ps_solver* s, int kind, int idx, double negval, double stat_scale,
                        double delta, double post_scale, int32_t* indptr /* N+1 */,
                        int32_t* indices, double* data, int64_t cap, int64_t* nnz_out);
int ps_record_fetch_dense(ps_solver* s, int kind, int idx, double* out /* N*N */);
/* Point gather from a record: out[i] = v*scale at (rows[i], cols[i]), 0 where v*scale < negval
 * -- what indexing the thresholded daily CSR solutions returns in Bayes_funcs.popdensity_grid /
 * popdensity_to_emergence (Bayes_funcs.py:20-179), without shipping the field to the host. */
int ps_record_gather(ps_solver* s, int kind, int idx, int64_t n, const int32_t* rows,
                     const int32_t* cols, double scale, double negval, double* out);
/* The same cells from nrec records in one launch and one transfer: out[r * n + i].
 * popdensity_to_emergence reads the same field cells on every day between release and
 * collection (Bayes_funcs.py:60-118): one call per collection instead of one per day. */
int ps_record_gather_multi(ps_solver* s, int nrec, const int32_t* kind, const int32_t* idx, int64_t n,
                           const int32_t* rows, const int32_t* cols, double scale, double negval,
                           double* out);
/* sum_d w[d] * record(kind[d], idx[d]) -> record (PS_REC_WSUM,0)  (CalcSol.py:322) */
int ps_weighted_sum(ps_solver* s, int n, const int32_t* kind, const int32_t* idx,
                    const double* w);

/* ---- measurement: HIP-event timing per kernel class on the handle's stream ----
 * classes: 0 row_fwd, 1 col_fwd (first/single pass), 2 col_fwd (second sub-pass),
 * 3 col_inv (first pass, fused spectral product), 4 col_inv (second), 5 row_inv+epilogue,
 * 6 the three predicated passes of the flag-conditional re-FFT (no-ops when the flag is clear),
 * 7/8/9 class 3 for 2/4/8 consecutive days in one launch,
 * 10/11/12 class 5 for the 2/4/8 days of a chained group in one launch (full-column pipeline),
 * 13/14 classes 3/5 for any other number of days per launch (up to 32; a solver whose previous run
 *       raised no flag opens with such windows) -- ps_prof_read_days gives the grid-days they covered,
 * 15 the three short launches for the columns taken out of a chained pass (its thin last round) */
#define PS_PROF_NCLS 16
int ps_prof_enable(ps_solver* s, int on);   /* 0 off, 1 every launch, n > 1 every n-th launch per class */
int ps_prof_read(ps_solver* s, int ncls, double* total_ms, int64_t* count); /* synchronises */
int ps_prof_read_days(ps_solver* s, int ncls, int64_t* days); /* grid-days of the timed launches per class */
int ps_prof_read_launches(ps_solver* s, int ncls, int64_t* launches); /* ALL launches per class since ps_prof_enable (timed or not) */
/* the same four arrays for one owner of an auto-mode run: 0 the front itself, 1 wide, 2 fold child, 3 narrow
 * (the numbering of ps_solver_auto_route).  ps_prof_enable on the front switches its helpers too, helpers
 * attached later inherit it; a helper the run never needed reads as zeros. */
int ps_prof_read_owner(ps_solver* s, int owner, int ncls, double* total_ms, int64_t* count, int64_t* launches,
                       int64_t* days);
int ps_solver_owner_fft(ps_solver* s, int owner); /* torus size of that owner, 0 while it does not exist */

/* full P x P complex spectrum in/out (function-level CalcSol.fft2/fftconv2/ifft2 mirrors;
 * only valid in PS_MODE_EXACT) */
int ps_solver_get_spectrum(ps_solver* s, double* out /* P*P*2 */);
int ps_solver_set_spectrum(ps_solver* s, const double* in /* P*P*2 */);

/* ---- per-day kernel construction: replaces ParasitoidModel.prob_mass
 *      (ParasitoidModel.py:384-613) incl. h_flight_prob (:282-309) and
 *      get_mvn_cdf_values (:311-380) ---- */
int ps_model_create(ps_model** out, int device);
int ps_model_destroy(ps_model* m);
/* "PS_PM_SEG" (periods summed per prob_mass record, 8) and "PS_PM_SYNC" (1: never size the pair
 * lists from the previous batch); see ps_solver_set_option */
int ps_model_set_option(ps_model* m, const char* key, double value);
/* wind: float64 [ndays_wind][T][3] (windx, windy, windr), rows in the order of the
 * sorted day keys; day_keys[ndays_wind] the integer keys (prob_mass looks up day+1).
 * T == 1 rows with test_run != 0 reproduce the single-period mode (:422-428). */
int ps_model_set_wind(ps_model* m, const double* wind, const int32_t* day_keys, int ndays_wind,
                      int T, int test_run);
/* Build the kernels of nd days in one batch.  day_idx[i] indexes the wind rows;
 * start_time[i] < 0 means None.  hparams = (lam,aw,bw,a1,b1,a2,b2), Dparams/Dlparams =
 * (sig_x,sig_y,rho).  Outputs per day: kshape (odd side of the shrunk kernel), nnz,
 * warned (1 if a period left the domain, ParasitoidModel.py:549-557), status
 * (PS_OK or the assertion class that fired). */
int ps_model_prob_mass(ps_model* m, int nd, const int32_t* day_idx, const double* start_time,
                       const double* hparams, const double* Dparams, const double* Dlparams,
                       double mu_r, int n_periods, double rad_dist, int rad_res,
                       int32_t* kshape, int64_t* nnz, int32_t* warned, int32_t* status);
int ps_model_fetch_coo(ps_model* m, int i, int32_t* row, int32_t* col, double* val, int64_t cap);
/* diagnostics for tests: hprob[T] of day i of the last batch, stamp half-widths H[T] */
int ps_model_fetch_debug(ps_model* m, int i, double* hprob, int32_t* Hs, double* loss,
                         double* pmfsum);
/* h_flight_prob (ParasitoidModel.py:282-309) of wind row day_i: out[T] */
int ps_model_hflight(ps_model* m, int day_i, const double* hparams, double* out);
/* get_mvn_cdf_values (ParasitoidModel.py:311-380) for one (cell, mu, S): returns the
 * half width; fills out[(2H+1)^2] when cap allows. */
int ps_model_mvn_cdf_values(ps_model* m, double cell, double mu_x, double mu_y, double sig_x,
                            double sig_y, double rho, int32_t* H, double* out, int64_t cap);
/* hand the kernels of the last batch straight to a solver on the same device
 * (no host round trip): equivalent to ps_chain_set_kernels with days [first, first+count). */
int ps_chain_set_kernels_from_model(ps_solver* s, ps_model* m, int first, int count);
int ps_solver_set_state_from_model(ps_solver* s, ps_model* m, int i);

/* ---- device-resident exchange between ranks (Run.py:412-425 maps prob_mass over a process pool; here
 * the days are sharded over the GPUs of a node and the COO kernels all-gathered with RCCL over xGMI).
 * The library does not know the communicator: the caller owns the device buffers of the collective and
 * passes raw DEVICE pointers (the only entry points that take device memory from outside).
 *   ps_model_export_device      concatenated triplets (int32 row, int32 col, float64 val) of days
 *                               [first, first+count) of the last batch -> caller's buffers (cap entries)
 *   ps_chain_set_kernels_device ps_chain_set_kernels from device triplets (copied before return)
 *   ps_solver_set_state_device  ps_solver_set_state_coo from the device triplets of an odd
 *                               kshape x kshape kernel, re-centred into the domain (Run.py:454-458) */
int ps_model_export_device(ps_model* m, int first, int count, void* row_dev, void* col_dev, void* val_dev,
                           int64_t cap);
int ps_chain_set_kernels_device(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                                const void* row_dev, const void* col_dev, const void* val_dev);
int ps_solver_set_state_device(ps_solver* s, const void* row_dev, const void* col_dev, const void* val_dev,
                               int64_t nnz, int kshape);

#ifdef __cplusplus
}
#endif
#endif
