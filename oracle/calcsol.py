"""Oracle for the day-chain FFT convolution solver (reference: CalcSol.py).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

`scipy.fft` (pocketfft) stands where the reference calls `scipy.fftpack`
(the same pocketfft c2c kernels).  `back_solve` follows the *intended*
semantics of the re-FFT step (cuda_lib.py:208-214): the reference's CPU line
CalcSol.py:104-105 passes a pad shape where a filter shape is expected and
crashes on the next multiply; parity for `back_solve`
is pinned where that flag never fires.
"""
import numpy as np
from scipy import sparse
from scipy import fft as sfft

from .model import r_small_vals  # noqa: F401  (CalcSol.py:112-136)


def pad_shape(dom_shape, filt_shape):
    """CalcSol.py:20-21: P = N + filt//2."""
    return tuple(int(v) for v in (np.array(dom_shape) + np.array(filt_shape) // 2))


def fft2(A, filt_shape):
    """CalcSol.py:11-24."""
    ps = pad_shape(A.shape, filt_shape)
    A_hat = np.zeros(ps)
    A_hat[:A.shape[0], :A.shape[1]] = A.toarray()
    return sfft.fft2(A_hat)


def fft2_to_pad(A, ps):
    """Zero-pad sparse/dense A (top-left) to exactly `ps` and FFT
    (the intended re-FFT of cuda_lib.py:208-214)."""
    A_hat = np.zeros(ps)
    Ad = A.toarray() if sparse.issparse(A) else np.asarray(A)
    A_hat[:Ad.shape[0], :Ad.shape[1]] = Ad
    return sfft.fft2(A_hat)


def ifft2(A_hat, Ashape):
    """CalcSol.py:28-41. Returns (coo, flag)."""
    A = sfft.ifft2(A_hat).real
    n0, n1 = Ashape
    flag = bool(max(A[n0:, n1:].max(), A[:n0, n1:].max(),
                    A[n0:, :n1].max()) > 1e-8)
    return sparse.coo_matrix(A[:n0, :n1]), flag


def wrap_kernel(B, ps):
    """CalcSol.py:58-64: odd-shaped kernel, centre moved to [0,0] with
    wrap-around, inside a zero array of shape `ps`."""
    B = sparse.csr_matrix(B)
    m0, m1 = np.array(B.shape) // 2
    out = np.zeros(ps)
    out[:m0 + 1, :m1 + 1] = B[m0:, m1:].toarray()
    out[:m0 + 1, -m1:] = B[m0:, :m1].toarray()
    out[-m0:, -m1:] = B[:m0, :m1].toarray()
    out[-m0:, :m1 + 1] = B[:m0, m1:].toarray()
    return out


def fftconv2(A_hat, B):
    """CalcSol.py:45-66. In-place A_hat *= fft2(wrapped B)."""
    A_hat *= sfft.fft2(wrap_kernel(B, A_hat.shape))


def back_solve(prev_spread, cursol_hat, dom_shape):
    """CalcSol.py:72-109 with the re-FFT as in cuda_lib.py:208-214."""
    bcksol = []
    hat = np.array(cursol_hat)
    for B in prev_spread[::-1]:
        hat = sfft.fft2(wrap_kernel(B, hat.shape)) * hat
        sol, flag = ifft2(hat, dom_shape)
        if flag:
            hat = fft2_to_pad(sol, hat.shape)
        bcksol.append(sol)
    return bcksol[::-1]


def get_solutions(modelsol, pmf_list, days, ndays, dom_len, max_shape,
                  trace=None):
    """CalcSol.py:140-201 (CPU branch :187-201).  Mutates modelsol.
    `trace`, if a dict, receives the flag sequence and the raw (unthresholded)
    domain fields for state-level parity checks."""
    cursol_hat = fft2(modelsol[0], max_shape)
    for n, _day in enumerate(days[1:ndays]):
        fftconv2(cursol_hat, pmf_list[n + 1].tocsr())
        A, flag = ifft2(cursol_hat, [dom_len, dom_len])
        modelsol.append(r_small_vals(A, prob_model=True))
        if trace is not None:
            trace.setdefault('flags', []).append(flag)
            trace.setdefault('raw', []).append(A.toarray())
        if flag:
            cursol_hat = fft2(A, max_shape)


def get_populations(r_spread, pmf_list, days, ndays, dom_len, max_shape,
                    r_dur, r_number, dist, trace=None):
    """CalcSol.py:205-325 (CPU branch :290-323)."""
    cur = [0 for _ in range(r_dur)]
    pop = []
    pop.append(r_small_vals(r_spread[0]).tocsr() * r_number * dist(1))
    pop[0][dom_len // 2, dom_len // 2] += r_number * (1 - dist(1))
    cur[0] = r_spread[0].tocoo()
    dom = [dom_len, dom_len]

    def wsum(nd):
        acc = cur[0] * dist(1)
        for d in range(1, nd):
            acc = acc + cur[d] * dist(d + 1)
        return acc

    if r_dur == 1:
        cursol_hat = fft2(r_spread[0], max_shape)
    for day in range(1, r_dur):
        cursol_hat = fft2(r_spread[day], max_shape)
        cur[day] = r_spread[day].tocoo()
        cur[:day] = back_solve(r_spread[:day], cursol_hat, dom)
        pop.append(r_small_vals(wsum(day + 1) * r_number).tocsr())
        pop[-1][dom_len // 2, dom_len // 2] += (1 - sum(
            dist(d + 1) for d in range(day + 1))) * r_number
    for n, _day in enumerate(days[r_dur:ndays]):
        fftconv2(cursol_hat, pmf_list[n + r_dur].tocsr())
        cur[-1], flag = ifft2(cursol_hat, dom)
        if trace is not None:
            trace.setdefault('flags', []).append(flag)
        if flag:
            cursol_hat = fft2(cur[-1], max_shape)
        cur[:-1] = back_solve(r_spread[:-1], cursol_hat, dom)
        pop.append(r_small_vals(wsum(r_dur) * r_number).tocsr())
    return pop
