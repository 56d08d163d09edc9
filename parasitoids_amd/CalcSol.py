"""Solution steps of a multi-day simulation on the MI355X -- counterpart of the
reference's `CalcSol.py` with the same function names and signatures.  All transforms,
products, thresholding and accumulation run on the device (hip_lib / libparasitoid_hip.so).
"""
import numpy as np
from scipy import sparse

from . import globalvars
from . import _lib as L


def _hip():
    # lazy, like the reference's `import cuda_lib` inside the solver functions
    # (CalcSol.py:162, :243): the GPU is touched only after any process pool is done
    from . import hip_lib
    return hip_lib


def _need_device_backend():
    if not globalvars.cuda:
        raise NotImplementedError(
            'parasitoids_amd has no CPU path: globalvars.cuda is False. Use the reference '
            'CalcSol for --no_cuda runs.')


def _square(shape, what):
    shape = tuple(int(v) for v in np.array(shape).ravel())
    if len(shape) == 1:
        shape = (shape[0], shape[0])
    if shape[0] != shape[1]:
        raise ValueError('{} must be square, got {}'.format(what, shape))
    return shape[0]


# ----------------------------------------------------------- function-level API

def fft2(A, filt_shape):
    '''fft of the sparse signal A zero-padded by filt_shape//2 (CalcSol.py:11-24).
    Returns the full complex P x P array (computed on the device).'''
    solver = _hip().HipSolve(sparse.coo_matrix(A), filt_shape, mode='exact')
    try:
        return solver.get_spectrum()
    finally:
        solver.close()


def _solver_for_spectrum(A_hat, dom_len):
    P = _square(A_hat.shape, 'A_hat')
    if not 0 < dom_len <= P:
        raise ValueError('domain {} does not fit the pad {}'.format(dom_len, P))
    ms = 2 * (P - dom_len) + 1
    solver = _hip().HipSolve(sparse.coo_matrix((dom_len, dom_len)), [ms, ms], mode='exact')
    solver.set_spectrum(A_hat)
    return solver


def ifft2(A_hat, Ashape):
    '''ifft of A_hat truncated to Ashape as a coo matrix, plus the flag telling that
    mass has reached the pad region (CalcSol.py:28-41).'''
    n = _square(Ashape, 'Ashape')
    solver = _solver_for_spectrum(A_hat, n)
    try:
        solver.get_cursol([n, n])
        return sparse.coo_matrix(solver.dense(L.REC_CHAIN, 0)), solver.last_flag
    finally:
        solver.close()


def fftconv2(A_hat, B):
    '''A_hat *= fft2(B wrapped to the origin), in place (CalcSol.py:45-66).'''
    B = sparse.coo_matrix(B)
    K = _square(B.shape, 'B')
    P = _square(A_hat.shape, 'A_hat')
    solver = _solver_for_spectrum(A_hat, max(1, P - K // 2))
    try:
        solver.fftconv2(B)
        A_hat[...] = solver.get_spectrum()
    finally:
        solver.close()


def back_solve(prev_spread, cursol_hat, dom_shape):
    '''Convolve progressively with the filters of prev_spread in reverse order; returns
    coo matrices in order of emergence (CalcSol.py:72-109; re-FFT as cuda_lib.py:208-214).'''
    n = _square(dom_shape, 'dom_shape')
    solver = _solver_for_spectrum(np.asarray(cursol_hat), n)
    try:
        solver.back_solve(prev_spread, [n, n])
        return [sparse.coo_matrix(solver.dense(L.REC_BACK, i)) for i in range(len(prev_spread))]
    finally:
        solver.close()


def r_small_vals(A, prob_model=False, negval=1e-8):
    '''Remove entries below negval; for the probability model add the removed mass back
    evenly so the result stays a pmf (CalcSol.py:112-136).'''
    A = sparse.coo_matrix(A)
    if A.shape[0] != A.shape[1]:
        raise ValueError('r_small_vals expects a square domain')
    solver = _hip().HipSolve(A, [1, 1], mode='fast')
    try:
        st = solver.record_stats(L.REC_STATE, 0, negval, 1.0, prob_model)
        return solver._fetch(L.REC_STATE, 0, negval, 1.0, st.delta, 1.0, st.nnz)
    finally:
        solver.close()


# ------------------------------------------------------------------ day chains

def get_solutions(modelsol, pmf_list, days, ndays, dom_len, max_shape):
    '''Find model solutions from a list of daily probability densities, given the
    distribution after the first day (CalcSol.py:140-201).

    Args:
        modelsol: list of model solutions with the first day's already entered
        pmf_list: list of probability densities. len(pmf_list) == len(days)
        days: list of day dictionary keys, mostly for feedback
        ndays: number of days to run simulation
        dom_len: number of cells across one side of the domain
        max_shape: largest filter shape, based on largest in pmf_list

    Modifies:
        modelsol'''
    _need_device_backend()
    hip_lib = _hip()
    nk = len(days[1:ndays])
    solver = hip_lib.HipSolve(modelsol[0], max_shape, mode=globalvars.fft_mode, chain_only=True)
    try:
        if solver.dom_len != dom_len:
            raise ValueError('dom_len {} != first solution {}'.format(dom_len, solver.dom_len))
        if nk == 0:
            return
        solver.set_kernels(pmf_list[1:1 + nk])
        solver.run_chain(0, nk, negval=1e-8, scale=1.0, renorm=True)
        stats = solver.chain_stats(0, nk)
        for n in range(nk):
            modelsol.append(solver.chain_solution(n, stats[n]))
    finally:
        solver.close()


# diagnostic: which route the last get_populations(r_dur > 1) took -- 'chain' (ps_chain_run_release),
# 'chain-uncertified' (an 'auto' run that had to be redone) or 'per-call'
last_release_route = None


def _release_on_chain(r_spread, pmf_list, days, ndays, dom_len, max_shape, r_dur, r_number, dist):
    '''get_populations for r_dur > 1 through the chain API (ps_chain_run_release): the day kernels and
    the release days' spreads go to the device once, every release day is one back-solve call and all
    days after the release ONE enqueued run -- no per-day host COO upload, no per-day synchronisation.
    The default mode ('auto') runs on the fast torus and certifies afterwards that nothing above 1e-15
    ever lay outside the domain (then it IS the exact-torus result); an uncertified run, a filter
    wider than max_shape or a mode without this entry point returns None and the caller takes the
    per-call route (CalcSol.py:296-323 literally).'''
    hip_lib = _hip()
    mode = globalvars.fft_mode
    if mode == 'fold':
        return None
    mid = dom_len // 2
    nk = len(days[r_dur:ndays])
    solver = hip_lib.HipSolve(r_spread[0], max_shape, mode=mode, chain_only=True)
    try:
        if solver.mode == 'fold' or not solver.set_release(pmf_list[r_dur:r_dur + nk], r_spread[:r_dur - 1]):
            return None
        popmodel = []
        st = solver.record_stats(L.REC_STATE, 0, 1e-8, 1.0, False)
        first = solver._fetch(L.REC_STATE, 0, 1e-8, 1.0, 0.0, r_number * dist(1), st.nnz, 'csr')
        first[mid, mid] += r_number * (1 - dist(1))
        popmodel.append(first)
        # successive release days (CalcSol.py:296-306)
        for day in range(1, r_dur):
            solver.set_state(r_spread[day])
            w = [dist(d + 1) * r_number for d in range(day + 1)]
            if not solver.run_release(0, 0, w, nuse=day):
                return None
            stw = solver.record_stats(L.REC_WSUM, 0, 1e-8, 1.0, False)
            pop = solver._fetch(L.REC_WSUM, 0, 1e-8, 1.0, 0.0, 1.0, stw.nnz, 'csr')
            pop[mid, mid] += (1 - sum(dist(d + 1) for d in range(day + 1))) * r_number
            popmodel.append(pop)
        # days after the release (CalcSol.py:308-323): one enqueued run
        if nk:
            w = [dist(d + 1) * r_number for d in range(r_dur)]
            if not solver.run_release(0, nk, w):
                return None
            stats = solver.chain_stats(0, nk)
            for n in range(nk):
                popmodel.append(solver._fetch(L.REC_CHAIN, n, 1e-8, 1.0, 0.0, 1.0, stats[n].nnz, 'csr'))
        return popmodel
    finally:
        solver.close()


def get_populations(r_spread, pmf_list, days, ndays, dom_len, max_shape,
                    r_dur, r_number, dist):
    '''Find expected wasp densities from a list of daily probability densities, given
    the spread of each release day (CalcSol.py:205-325).

    Returns:
        popmodel: expected wasp population numbers on each day (list of csr matrices)'''
    _need_device_backend()
    hip_lib = _hip()
    mid = dom_len // 2
    global last_release_route
    if r_dur > 1:
        pop = _release_on_chain(r_spread, pmf_list, days, ndays, dom_len, max_shape, r_dur, r_number, dist)
        if pop is not None:
            last_release_route = 'chain'
            return pop
        last_release_route = 'per-call'
    popmodel = []
    solver = hip_lib.HipSolve(r_spread[0], max_shape, mode=globalvars.fft_mode, chain_only=(r_dur == 1))
    try:
        # first day: r_small_vals(r_spread[0]) * r_number * dist(1), rest still at the origin
        st = solver.record_stats(L.REC_STATE, 0, 1e-8, 1.0, False)
        first = solver._fetch(L.REC_STATE, 0, 1e-8, 1.0, 0.0, r_number * dist(1), st.nnz, 'csr')
        first[mid, mid] += r_number * (1 - dist(1))
        popmodel.append(first)

        if r_dur == 1:
            nk = len(days[r_dur:ndays])
            if nk:
                scale = dist(1) * r_number
                solver.set_kernels(pmf_list[r_dur:r_dur + nk])
                solver.run_chain(0, nk, negval=1e-8, scale=scale, renorm=False)
                stats = solver.chain_stats(0, nk)
                for n in range(nk):
                    popmodel.append(solver._fetch(L.REC_CHAIN, n, 1e-8, scale, 0.0, 1.0,
                                                  stats[n].nnz, 'csr'))
            return popmodel

        def weighted(kinds, idxs, ndist):
            w = [dist(d + 1) * r_number for d in range(ndist)]
            solver.weighted_sum(kinds, idxs, w)
            stw = solver.record_stats(L.REC_WSUM, 0, 1e-8, 1.0, False)
            return solver._fetch(L.REC_WSUM, 0, 1e-8, 1.0, 0.0, 1.0, stw.nnz, 'csr')

        # successive release days (CalcSol.py:296-306)
        for day in range(1, r_dur):
            solver.set_state(r_spread[day])
            solver.back_solve(r_spread[:day], [dom_len, dom_len], fetch=False)
            pop = weighted([L.REC_BACK] * day + [L.REC_STATE], list(range(day)) + [0], day + 1)
            pop[mid, mid] += (1 - sum(dist(d + 1) for d in range(day + 1))) * r_number
            popmodel.append(pop)
        # days after the release (CalcSol.py:308-323)
        for n, _day in enumerate(days[r_dur:ndays]):
            solver.fftconv2(pmf_list[n + r_dur])
            solver.get_cursol([dom_len, dom_len], fetch=False)
            solver.back_solve(r_spread[:-1], [dom_len, dom_len], fetch=False)
            popmodel.append(weighted([L.REC_BACK] * (r_dur - 1) + [L.REC_CHAIN],
                                     list(range(r_dur - 1)) + [0], r_dur))
        return popmodel
    finally:
        solver.close()
