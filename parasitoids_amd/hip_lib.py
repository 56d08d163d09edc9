"""HIP methods for convolutions -- the MI355X counterpart of the reference's
`cuda_lib.py`: class `HipSolve` keeps `cuda_lib.CudaSolve`'s constructor and
its three methods (cuda_lib.py:18, :58, :98, :145), computing in fp64 on the
device through libparasitoid_hip.so.

Importing this module raises ImportError when the shared library or a GPU is
missing, exactly the condition under which the reference's
`CalcSol.get_solutions` falls back (CalcSol.py:161-172).
"""
import ctypes as C
import warnings

import numpy as np
from scipy import sparse

from . import _lib as L

L.load()
L.require_device()


def _coo_arrays(A):
    A = sparse.coo_matrix(A)
    return L.i32(A.row), L.i32(A.col), L.f64(A.data), A.shape


def _smooth7(n):
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


def _create_solver(lib, handle, dev, dom_len, max_shape, mode, chain_only=False):
    """ps_solver_create with mode
      'exact'  transform on the reference's pad P = N + K//2 itself (any P whose prime factors
               are <= 1024);
      'fold'   the same torus, computed as a linear convolution on a fast FFT size >= P + K - 1
               and folded back modulo P -- chain API only;
      'fast'   a convenient FFT size >= P (pad-region dust lives on a different torus);
      'auto'   exact reference-torus results by the cheapest route: exact when P is 7-smooth
               (cheap); else, when the caller only needs the chain API (`chain_only`), the
               library's PS_MODE_AUTO -- the chain runs on the fast torus while nothing above
               1e-15 lies outside the domain (then the tori cannot differ) and continues on the
               folded reference torus from the first day that does; else exact; fast if none of
               them can be planned.
    Returns the mode actually used ('auto' = PS_MODE_AUTO)."""
    if mode not in ('exact', 'fold', 'fast', 'auto'):
        raise ValueError("mode must be 'exact', 'fold', 'fast' or 'auto'")
    order = {'exact': ['exact'], 'fold': ['fold'], 'fast': ['fast']}.get(mode)
    if order is None:
        P = dom_len + max_shape // 2
        if _smooth7(P) or not chain_only:
            order = ['exact', 'fast']
        else:
            order = ['auto', 'fold', 'exact', 'fast']
    code = {'exact': L.MODE_EXACT, 'fold': L.MODE_FOLD, 'fast': L.MODE_FAST, 'auto': L.MODE_AUTO}
    rc = L.PS_OK
    for m in order:
        rc = lib.ps_solver_create(C.byref(handle), dev, dom_len, max_shape, code[m])
        if rc == L.PS_OK:
            if m == 'fast' and mode == 'auto':
                # not silent: results are then no longer on the reference's N + K//2 torus
                # (sub-1e-8 pad dust wraps differently: <= 1e-8 per un-flagged day)
                info = [C.c_int32() for _ in range(4)]
                lib.ps_solver_info(handle, *[C.byref(v) for v in info])
                warnings.warn("HipSolve(mode='auto'): the reference pad %d cannot be planned exactly "
                              "(prime factor > 1024 or size > 9720); falling back to the fast torus of "
                              "size %d -- states may differ from the reference's by <= 1e-8 per un-flagged "
                              "day near the boundary.  Pass mode='exact' to fail instead."
                              % (dom_len + max_shape // 2, info[2].value), RuntimeWarning, stacklevel=3)
            return m
        if rc != L.PS_ERR_UNSUPPORTED:
            break
    L.check(rc)


class HipSolve():
    """Device-resident Fourier-space solution, cf. `cuda_lib.CudaSolve`."""

    def __init__(self, A, max_shape, mode='auto', device=None, chain_only=False):
        '''Initialize the solver with the fft of the solution after the first day.

        Args:
            A: First day's spread, sparse matrix (square, N x N)
            max_shape: Shape of the largest filter (cuda_lib.py:18-28)
            mode: 'exact' | 'fold' | 'fast' | 'auto', see `_create_solver`.  The default 'auto'
                  is exact reference-torus arithmetic whenever the pad can be planned and the
                  fast size otherwise, so a `cuda_lib` shim never
                  fails at construction for an awkward pad (prime factor > 1024, size > 9720)
            chain_only: the caller uses set_kernels / run_chain / records only (lets 'auto'
                  pick the fold mode, which has no per-call fftconv2 / get_cursol / back_solve)'''
        self._h = L._VP()
        self._lib = L.load()
        row, col, val, shape = _coo_arrays(A)
        if shape[0] != shape[1]:
            raise ValueError('domain must be square, got {}'.format(shape))
        ms = np.array(max_shape).astype(int).ravel()
        if ms.size == 1:
            ms = np.array([ms[0], ms[0]])
        if ms[0] != ms[1]:
            raise ValueError('max_shape must be square, got {}'.format(tuple(ms)))
        self.dom_len = int(shape[0])
        mmid = ms // 2
        self.pad_shape = (int(shape[0] + mmid[0]), int(shape[1] + mmid[1]))
        dev = L.default_device() if device is None else device
        self.mode = _create_solver(self._lib, self._h, dev, self.dom_len, int(ms[0]), mode, chain_only)
        info = [C.c_int32() for _ in range(4)]
        L.check(self._lib.ps_solver_info(self._h, *[C.byref(v) for v in info]))
        self.fft_len = info[2].value
        L.check(self._lib.ps_solver_set_state_coo(self._h, L.p_i32(row), L.p_i32(col),
                                                  L.p_f64(val), len(val)))
        self._nk = 0

    @classmethod
    def from_model(cls, model, i, max_shape, mode='auto', device=None, chain_only=False):
        '''Solver whose first-day state is day i of `model`'s last device batch, re-centred
        into the domain (Run.py:454-458) without leaving the GPU.'''
        self = cls.__new__(cls)
        self._h = L._VP()
        self._lib = L.load()
        ms = int(np.array(max_shape).ravel()[0])
        N = 2 * int(model.last['args'][6]) + 1
        self.dom_len = N
        self.pad_shape = (N + ms // 2, N + ms // 2)
        dev = L.default_device() if device is None else device
        self.mode = _create_solver(self._lib, self._h, dev, N, ms, mode, chain_only)
        info = [C.c_int32() for _ in range(4)]
        L.check(self._lib.ps_solver_info(self._h, *[C.byref(v) for v in info]))
        self.fft_len = info[2].value
        self._nk = 0
        self.set_state_from_model(model, i)
        return self

    @classmethod
    def from_device_kernels(cls, g, i, max_shape, mode='auto', device=None, chain_only=True):
        '''Solver whose first-day state is kernel `i` of a gathered device kernel set
        (`parallel.prob_mass_sharded_device`), re-centred into the domain (Run.py:454-458); the other
        kernels go in with `set_kernels_device`.  Nothing passes through host memory.'''
        self = cls.__new__(cls)
        self._h = L._VP()
        self._lib = L.load()
        ms = int(np.array(max_shape).ravel()[0])
        N = int(g['dom_len'])
        self.dom_len = N
        self.pad_shape = (N + ms // 2, N + ms // 2)
        dev = L.default_device() if device is None else device
        self.mode = _create_solver(self._lib, self._h, dev, N, ms, mode, chain_only)
        info = [C.c_int32() for _ in range(4)]
        L.check(self._lib.ps_solver_info(self._h, *[C.byref(v) for v in info]))
        self.fft_len = info[2].value
        self._nk = 0
        self.set_state_device(g, i)
        return self

    def set_state_device(self, g, i):
        import torch
        torch.cuda.synchronize()
        o, n = int(g['off'][i]), int(g['off'][i + 1] - g['off'][i])
        L.check(self._lib.ps_solver_set_state_device(
            self._h, g['row'].data_ptr() + 4 * o, g['col'].data_ptr() + 4 * o, g['val'].data_ptr() + 8 * o,
            n, int(g['kshape'][i])))

    def set_kernels_device(self, g, first, count):
        '''Day kernels [first, first+count) of a gathered device kernel set as this solver's chain
        kernels (ps_chain_set_kernels_device: device to device).'''
        import torch
        torch.cuda.synchronize()
        o = int(g['off'][first])
        off = np.ascontiguousarray(g['off'][first:first + count + 1] - o, dtype=np.int64)
        ks = np.ascontiguousarray(g['kshape'][first:first + count], dtype=np.int32)
        L.check(self._lib.ps_chain_set_kernels_device(
            self._h, count, L.p_i64(off), L.p_i32(ks), g['row'].data_ptr() + 4 * o,
            g['col'].data_ptr() + 4 * o, g['val'].data_ptr() + 8 * o))
        self._nk = count

    def retarget(self, max_shape):
        '''fold / auto mode: move the solver to another kernel shape limit / reference torus,
        keeping its FFT size, plans and buffers (the state has to be set again)'''
        ms = int(np.array(max_shape).ravel()[0])
        L.check(self._lib.ps_solver_retarget(self._h, ms))
        self.pad_shape = (self.dom_len + ms // 2, self.dom_len + ms // 2)
        self._nk = 0

    def set_state_from_model(self, model, i):
        L.check(self._lib.ps_solver_set_state_from_model(self._h, model._h, int(i)))

    def set_kernels_from_model(self, model, first, count):
        '''Adopt days [first, first+count) of the model's last batch as the chain's day
        kernels, device to device.'''
        L.check(self._lib.ps_chain_set_kernels_from_model(self._h, model._h, int(first), int(count)))
        self._nk = int(count)

    def set_state(self, A):
        '''Replace the Fourier-space solution by fft2(A) (what the constructor does,
        cuda_lib.py:34-54), keeping plans, buffers and uploaded kernels.'''
        row, col, val, shape = _coo_arrays(A)
        if shape != (self.dom_len, self.dom_len):
            raise ValueError('state shape {} != solver domain'.format(shape))
        L.check(self._lib.ps_solver_set_state_coo(self._h, L.p_i32(row), L.p_i32(col),
                                                  L.p_f64(val), len(val)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.ps_solver_destroy(self._h)
            self._h = L._VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -------------------------------------------------------- CudaSolve API
    def fftconv2(self, B, mem_print=False):
        '''Update current fourier solution with filter B (cuda_lib.py:58-94).

        Args:
            B: 2D sparse matrix with odd, square shape'''
        row, col, val, shape = _coo_arrays(B)
        if shape[0] != shape[1]:
            raise ValueError('filter must be square, got {}'.format(shape))
        L.check(self._lib.ps_solver_fftconv2_coo(self._h, L.p_i32(row), L.p_i32(col),
                                                 L.p_f64(val), len(val), int(shape[0])))

    def get_cursol(self, dom_shape, negval=1e-8, fetch=True):
        '''Return the current solution (requires ifft) with small values removed
        (cuda_lib.py:98-140); re-transforms the truncated solution when mass has
        reached the pad region.

        Returns:
            coo matrix, current solution with shape (dom_len,dom_len)'''
        self._check_dom(dom_shape)
        st = L.DayStats()
        L.check(self._lib.ps_solver_get_cursol(self._h, negval, 1.0, 0, C.byref(st)))
        self.last_flag = bool(st.flag)
        if not fetch:       # leave the field on the device (record PS_REC_CHAIN,0)
            return None
        return self._fetch(L.REC_CHAIN, 0, negval, 1.0, 0.0, 1.0, st.nnz)

    def back_solve(self, prev_spread, dom_shape, negval=1e-8, fetch=True):
        '''For each filter in prev_spread, convolute progressively in reverse order
        (cuda_lib.py:145-221).  Returns coo matrices in order of emergence.'''
        self._check_dom(dom_shape)
        nf = len(prev_spread)
        if nf == 0:
            return []
        rows, cols, vals, off = [], [], [], [0]
        for B in prev_spread:
            r, c, v, shape = _coo_arrays(B)
            if shape != (self.dom_len, self.dom_len):
                raise ValueError('back_solve filters must be dom_len x dom_len')
            rows.append(r); cols.append(c); vals.append(v)
            off.append(off[-1] + len(v))
        row = L.i32(np.concatenate(rows)); col = L.i32(np.concatenate(cols))
        val = L.f64(np.concatenate(vals)); off = np.ascontiguousarray(off, dtype=np.int64)
        stats = (L.DayStats * nf)()
        L.check(self._lib.ps_solver_back_solve(self._h, nf, L.p_i64(off), L.p_i32(row),
                                               L.p_i32(col), L.p_f64(val), negval, 1.0, stats))
        self.last_back_flags = [bool(s.flag) for s in stats]
        if not fetch:       # leave the fields on the device (records PS_REC_BACK,i)
            return None
        return [self._fetch(L.REC_BACK, i, negval, 1.0, 0.0, 1.0, stats[i].nnz)
                for i in range(nf)]

    # ------------------------------------------------------ whole-chain API
    def set_kernels(self, pmf_list):
        '''Upload a list of odd-shaped sparse day kernels (prob_mass outputs).'''
        rows, cols, vals, off, ks = [], [], [], [0], []
        for B in pmf_list:
            r, c, v, shape = _coo_arrays(B)
            if shape[0] != shape[1]:
                raise ValueError('kernel must be square, got {}'.format(shape))
            rows.append(r); cols.append(c); vals.append(v)
            off.append(off[-1] + len(v)); ks.append(shape[0])
        cat = lambda xs, dt: np.ascontiguousarray(
            np.concatenate(xs) if xs else np.zeros(0), dtype=dt)
        row, col, val = cat(rows, np.int32), cat(cols, np.int32), cat(vals, np.float64)
        off = np.ascontiguousarray(off, dtype=np.int64)
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        L.check(self._lib.ps_chain_set_kernels(self._h, len(ks), L.p_i64(off), L.p_i32(ks),
                                               L.p_i32(row), L.p_i32(col), L.p_f64(val)))
        self._nk = len(ks)

    def set_release(self, pmf_days, r_spread):
        '''Kernels of a multi-day release on the chain API (get_populations with r_dur > 1,
        CalcSol.py:296-323): the day kernels followed by the release days' spreads as back-solve
        filters -- each N x N spread cut to its support box about the centre, an odd kernel that lands
        on the torus exactly where the reference wraps the whole filter (CalcSol.py:86-91).  Returns
        False (nothing uploaded) when a filter reaches further from the centre than max_shape // 2:
        the caller then keeps the per-call back_solve.'''
        M = self.pad_shape[0] - self.dom_len
        mid = self.dom_len // 2
        filt = []
        for F in r_spread:
            F = sparse.coo_matrix(F)
            if F.shape != (self.dom_len, self.dom_len):
                raise ValueError('release filters must be dom_len x dom_len')
            h = int(max(np.abs(F.row - mid).max(), np.abs(F.col - mid).max())) if F.nnz else 0
            if h > M:
                return False
            filt.append(sparse.coo_matrix((F.data, (F.row - mid + h, F.col - mid + h)),
                                          shape=(2 * h + 1, 2 * h + 1)))
        self.set_kernels(list(pmf_days) + filt)
        self._nfilt = len(filt)
        return True

    def run_release(self, first, count, weights, nuse=None, negval=1e-8):
        '''Days [first, first+count) of a multi-day release (ps_chain_run_release): per day the last
        cohort's step, the back-solve through the filters and the population
        sum_i weights[i] back_i + weights[nuse] cohort -> chain record d; count = 0: the back-solve of
        the current state alone -> record (REC_WSUM, 0).  Returns False when an 'auto' solver could
        not certify the run as exact on its fast torus (redo it in 'exact' mode).'''
        nuse = self._nfilt if nuse is None else nuse
        w = L.f64(weights)
        if len(w) != nuse + 1:
            raise ValueError('need nuse + 1 weights')
        ok = C.c_int(1)
        L.check(self._lib.ps_chain_run_release(self._h, first, count, negval, self._nfilt, nuse,
                                               L.p_f64(w), C.byref(ok)))
        return bool(ok.value)

    def block_prefix(self, first, count):
        '''One simulation over several GPUs (ps_chain_block_prefix): the running products of the day
        kernels [first, first+count) stay in the solver; -> (device pointer, bytes) of the block total, valid
        until the next block call (parallel.chain_prefix_split exchanges these).'''
        ptr, nbytes = C.c_void_p(), C.c_int64(0)
        L.check(self._lib.ps_chain_block_prefix(self._h, int(first), int(count), C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def block_finish(self, first, count, prev_totals=(), negval=1e-8, scale=1.0, renorm=True):
        '''... and the chain records / statistics of those days from state x (earlier blocks' totals, device
        pointers in block order) x own products (ps_chain_block_finish).  -> True when one of the days raised
        the boundary flag: the split does not apply, rerun with run_chain.'''
        n = len(prev_totals)
        arr = (C.c_void_p * max(n, 1))(*[int(p) for p in prev_totals])
        flag = C.c_int(0)
        L.check(self._lib.ps_chain_block_finish(self._h, int(first), int(count), n, arr, negval, scale,
                                                int(bool(renorm)), C.byref(flag)))
        return bool(flag.value)

    def device_copy(self, dst_ptr, src_ptr, nbytes):
        '''device-to-device copy between raw pointers (a block total into a tensor of the collective)'''
        L.check(self._lib.ps_device_copy(C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), int(nbytes)))

    def run_chain(self, first=0, count=None, negval=1e-8, scale=1.0, renorm=True):
        '''Enqueue days [first, first+count) of the uploaded kernels (no host sync).'''
        count = self._nk - first if count is None else count
        L.check(self._lib.ps_chain_run(self._h, first, count, negval, scale, int(bool(renorm))))

    def chain_stats(self, first, count):
        stats = (L.DayStats * count)()
        L.check(self._lib.ps_chain_stats(self._h, first, count, stats))
        return list(stats)

    def chain_solution(self, day, stats, negval=1e-8, scale=1.0, fmt='coo'):
        '''r_small_vals(A, prob_model=renorm) of chain day `day` as a coo (or csr) matrix.'''
        return self._fetch(L.REC_CHAIN, day, negval, scale, stats.delta, 1.0, stats.nnz, fmt)

    def record_stats(self, kind, idx, negval=1e-8, scale=1.0, renorm=False):
        st = L.DayStats()
        L.check(self._lib.ps_record_stats(self._h, kind, idx, negval, scale, int(renorm),
                                          C.byref(st)))
        return st

    def weighted_sum(self, kinds, idxs, weights):
        k = L.i32(kinds); i = L.i32(idxs); w = L.f64(weights)
        L.check(self._lib.ps_weighted_sum(self._h, len(w), L.p_i32(k), L.p_i32(i), L.p_f64(w)))

    def gather(self, kind, idx, rows, cols, scale=1.0, negval=1e-8):
        '''Values of the thresholded record at the given cells (device gather).'''
        r, c = L.i32(rows), L.i32(cols)
        out = np.empty(len(r))
        L.check(self._lib.ps_record_gather(self._h, kind, idx, len(r), L.p_i32(r), L.p_i32(c),
                                           float(scale), float(negval), L.p_f64(out)))
        return out

    def gather_multi(self, kinds, idxs, rows, cols, scale=1.0, negval=1e-8):
        '''The same cells from several records in one launch and one transfer -> [nrec, n].'''
        k, i = L.i32(kinds), L.i32(idxs)
        r, c = L.i32(rows), L.i32(cols)
        out = np.empty((len(k), len(r)))
        L.check(self._lib.ps_record_gather_multi(self._h, len(k), L.p_i32(k), L.p_i32(i), len(r),
                                                 L.p_i32(r), L.p_i32(c), float(scale), float(negval),
                                                 L.p_f64(out)))
        return out

    def dense(self, kind, idx):
        out = np.empty((self.dom_len, self.dom_len))
        L.check(self._lib.ps_record_fetch_dense(self._h, kind, idx, L.p_f64(out)))
        return out

    def sync(self):
        L.check(self._lib.ps_solver_sync(self._h))

    def set_option(self, key, value):
        '''Change a tuning / A-B knob of this solver (names as the environment
        variables that seed a new solver, e.g. 'PS_RSP').  The library reads the environment only
        when a solver is created.'''
        L.check(self._lib.ps_solver_set_option(self._h, key.encode(), float(value)))

    def get_option(self, key):
        v = C.c_double()
        L.check(self._lib.ps_solver_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    @property
    def full_column(self):
        '''True when the solver runs the full-column pipeline -- measurement aid.'''
        return bool(self._lib.ps_solver_pipeline(self._h))

    def auto_info(self):
        '''PS_MODE_AUTO: (first chain day of the last run_chain that ran on the folded reference
        torus, or -1 when every day was clean and ran on the fast torus; FFT size of the fold
        path, 0 if it was never needed)'''
        a, b = C.c_int32(-1), C.c_int32(0)
        L.check(self._lib.ps_solver_auto_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def auto_route(self, first, count):
        '''PS_MODE_AUTO: per chain day of the last run_chain, 0 = fast-torus front (clean prefix),
        1 = wide fast-torus helper (flagged / clean days), 2 = fold child (dusty days)'''
        out = np.zeros(count, dtype=np.int32)
        L.check(self._lib.ps_solver_auto_route(self._h, int(first), int(count), L.p_i32(out)))
        return out

    @property
    def kernels_direct(self):
        '''True when the last chunk of day kernels took the direct-sum first column sub-pass
        (compact kernels on a split column transform) -- measurement aid.'''
        return bool(self._lib.ps_solver_kernels_direct(self._h))

    PROF_CLASSES = ('row_fwd', 'col_fwd_a', 'col_fwd_b', 'col_inv_a', 'col_inv_b', 'row_inv',
                    'refft_pred', 'col_inv_a_x2', 'col_inv_a_x4', 'col_inv_a_x8',
                    'row_inv_x2', 'row_inv_x4', 'row_inv_x8', 'col_inv_a_xn', 'row_inv_xn', 'col_tail')

    def prof_enable(self, on=True, every=1):
        '''HIP-event timing per kernel class on the solver's stream; `every` = n times only
        every n-th launch of each class (the events themselves cost ~1 us per launch).'''
        L.check(self._lib.ps_prof_enable(self._h, (max(1, int(every)) if on else 0)))

    def prof_read(self):
        '''-> {class: (total_ms, launches)} (synchronises)'''
        n = len(self.PROF_CLASSES)
        ms = np.zeros(n)
        cnt = np.zeros(n, dtype=np.int64)
        L.check(self._lib.ps_prof_read(self._h, n, L.p_f64(ms), L.p_i64(cnt)))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.PROF_CLASSES)}

    def prof_days(self):
        '''-> {class: grid-days covered by its timed launches} (the multi-day classes; `_xn` =
        any number of chained days per launch)'''
        n = len(self.PROF_CLASSES)
        days = np.zeros(n, dtype=np.int64)
        L.check(self._lib.ps_prof_read_days(self._h, n, L.p_i64(days)))
        return {k: int(days[i]) for i, k in enumerate(self.PROF_CLASSES)}

    def prof_launches(self):
        '''-> {class: all launches since prof_enable}, timed or not'''
        n = len(self.PROF_CLASSES)
        cnt = np.zeros(n, dtype=np.int64)
        L.check(self._lib.ps_prof_read_launches(self._h, n, L.p_i64(cnt)))
        return {k: int(cnt[i]) for i, k in enumerate(self.PROF_CLASSES)}

    PROF_OWNERS = ('front', 'wide', 'fold', 'narrow')

    def prof_owner(self, owner):
        '''-> {class: dict(ms, timed, launches, days)} for one owner of an auto-mode run ('front', 'wide',
        'fold', 'narrow' or 0..3, the numbering of auto_route); classes never launched are left out'''
        o = self.PROF_OWNERS.index(owner) if isinstance(owner, str) else int(owner)
        n = len(self.PROF_CLASSES)
        ms = np.zeros(n)
        cnt, lau, days = (np.zeros(n, dtype=np.int64) for _ in range(3))
        L.check(self._lib.ps_prof_read_owner(self._h, o, n, L.p_f64(ms), L.p_i64(cnt), L.p_i64(lau), L.p_i64(days)))
        return {k: dict(ms=float(ms[i]), timed=int(cnt[i]), launches=int(lau[i]), days=int(days[i]))
                for i, k in enumerate(self.PROF_CLASSES) if lau[i] or cnt[i]}

    def helper_fft_len(self, owner):
        '''torus size of one owner of an auto-mode run (0 while that helper does not exist)'''
        o = self.PROF_OWNERS.index(owner) if isinstance(owner, str) else int(owner)
        return int(self._lib.ps_solver_owner_fft(self._h, o))

    def get_spectrum(self):
        P = self.fft_len
        out = np.empty((P, P), dtype=np.complex128)
        L.check(self._lib.ps_solver_get_spectrum(self._h, out.ctypes.data_as(L._F64P)))
        return out

    def set_spectrum(self, A_hat):
        A_hat = np.ascontiguousarray(A_hat, dtype=np.complex128)
        if A_hat.shape != (self.fft_len, self.fft_len):
            raise ValueError('spectrum shape {} != pad shape'.format(A_hat.shape))
        L.check(self._lib.ps_solver_set_spectrum(self._h, A_hat.ctypes.data_as(L._F64P)))

    # -------------------------------------------------------------- helpers
    def _check_dom(self, dom_shape):
        if int(dom_shape[0]) != self.dom_len or int(dom_shape[1]) != self.dom_len:
            raise ValueError('dom_shape {} != solver domain {}'.format(
                tuple(dom_shape), self.dom_len))

    def _fetch(self, kind, idx, negval, scale, delta, post, nnz_hint, fmt='coo'):
        """Thresholded record as a scipy sparse matrix.  fmt='coo': row-major COO (what
        `coo_matrix(dense)` gives, CalcSol.py:41); fmt='csr': the CSR triplets straight from the
        device compaction (what `.tocsr()` / the reference's result files hold)."""
        n = int(nnz_hint)
        cap = max(n, 1)
        N = self.dom_len
        while True:
            first = np.empty(cap if fmt == 'coo' else N + 1, dtype=np.int32)
            col = np.empty(cap, dtype=np.int32)
            val = np.empty(cap, dtype=np.float64)
            nnz = C.c_int64()
            fn = self._lib.ps_record_fetch_coo if fmt == 'coo' else self._lib.ps_record_fetch_csr
            rc = fn(self._h, kind, idx, negval, scale, delta, post, L.p_i32(first), L.p_i32(col),
                    L.p_f64(val), cap, C.byref(nnz))
            if rc == L.PS_ERR_BAD_ARG and nnz.value > cap:
                cap = nnz.value
                continue
            L.check(rc)
            n = nnz.value
            if fmt == 'coo':
                return sparse.coo_matrix((val[:n], (first[:n], col[:n])), shape=(N, N))
            return sparse.csr_matrix((val[:n], col[:n], first), shape=(N, N))
