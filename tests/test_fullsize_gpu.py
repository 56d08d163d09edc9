"""Full-size (BASELINE.json config 3: N = 4097, K = 2049, reference torus P = 5121) checks
through size-independent properties, where the CPU oracle would take minutes:
 * the benchmarked fast mode (P' = 5184) against the exact reference torus (5121 = 9*569,
   wave-cooperative radix-569) on the same stack,
 * mass conservation and additivity of mean / variance under convolution,
 * linearity of the whole chain in the initial state (moderate size),
 * r_small_vals idempotence on device results."""
import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip_lib():
    from parasitoids_amd import hip_lib
    return hip_lib


def _moments(raw):
    idx = np.arange(raw.shape[0], dtype=np.float64)
    s = raw.sum()
    r, c = raw.sum(1), raw.sum(0)
    mr, mc = (r * idx).sum() / s, (c * idx).sum() / s
    return s, mr, mc, (r * (idx - mr) ** 2).sum() / s, (c * (idx - mc) ** 2).sum() / s


def test_fullsize_fast_equals_exact_torus_and_moments(hip_lib):
    from parasitoids_amd import synthetic
    R, K, nd = 2048, 2049, 3
    state, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=20240613)
    fields = {}
    for mode in ('fast', 'exact'):
        s = hip_lib.HipSolve(state, [K, K], mode=mode)
        assert s.pad_shape == (5121, 5121)
        assert s.fft_len == (5184 if mode == 'fast' else 5121)
        s.set_kernels(kernels)
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        assert not any(x.flag for x in st)
        fields[mode] = [s.dense(0, d) for d in range(nd)]
        for x in st:
            assert abs(x.sum + x.delta * x.nnz - 1.0) < 1e-12
        s.close()
    for d in range(nd):
        assert np.abs(fields['fast'][d] - fields['exact'][d]).max() < 1e-13
    # mean and variance add under convolution; mass is conserved
    km = [synthetic.moments(k) for k in kernels]
    s, mr, mc, vr, vc = _moments(fields['fast'][-1])
    assert abs(s - 1.0) < 1e-11
    assert abs(mr - (R + sum(m[1] - K // 2 for m in km))) < 1e-7
    assert abs(mc - (R + sum(m[2] - K // 2 for m in km))) < 1e-7
    assert abs(vr / sum(m[3] for m in km) - 1) < 1e-8
    assert abs(vc / sum(m[4] for m in km) - 1) < 1e-8


def test_chain_is_linear_in_the_state(hip_lib):
    from parasitoids_amd import synthetic
    R, K, nd = 256, 129, 4
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=5, sigma=(3.0, 8.0), shift=10.0)
    N = 2 * R + 1
    rng = np.random.default_rng(0)

    def blob(seed):
        r = np.random.default_rng(seed)
        ij = r.integers(R - 40, R + 40, size=(200, 2))
        return sparse.coo_matrix((r.random(200), (ij[:, 0], ij[:, 1])), shape=(N, N))

    s1, s2 = blob(1), blob(2)
    a, b = 0.3, 1.7
    outs = []
    for st in (s1, s2, (a * s1 + b * s2)):
        s = hip_lib.HipSolve(st, [K, K], mode='exact')
        s.set_kernels(kernels)
        s.run_chain(renorm=False)
        s.chain_stats(0, nd)
        outs.append(s.dense(0, nd - 1))
        s.close()
    scale = np.abs(outs[2]).max()
    assert np.abs(a * outs[0] + b * outs[1] - outs[2]).max() < 1e-13 * max(scale, 1.0)


def test_r_small_vals_idempotent(hip_lib):
    from parasitoids_amd import synthetic, CalcSol
    state, kernels, _ = synthetic.make_stack(R=100, K=65, ndays=2, seed=3, sigma=(2.0, 5.0), shift=5.0)
    s = hip_lib.HipSolve(state, [65, 65])
    s.set_kernels(kernels)
    s.run_chain(renorm=True)
    st = s.chain_stats(0, 2)
    sol = s.chain_solution(1, st[1])
    again = CalcSol.r_small_vals(sol, prob_model=True)
    assert again.nnz == sol.nnz
    assert np.abs(again.tocsr() - sol.tocsr()).max() < 1e-16
    s.close()


def test_empty_and_degenerate_inputs(hip_lib):
    N = 33
    empty = sparse.coo_matrix((N, N))
    s = hip_lib.HipSolve(empty, [9, 9])
    s.fftconv2(sparse.coo_matrix(([1.0], ([4], [4])), shape=(9, 9)))
    out = s.get_cursol([N, N])
    assert out.nnz == 0 and not s.last_flag
    # identity kernel (1 x 1) leaves a state unchanged
    st = sparse.coo_matrix(([0.25, 0.75], ([3, 30], [5, 31])), shape=(N, N))
    s.set_state(st)
    s.set_kernels([sparse.coo_matrix(([1.0], ([0], [0])), shape=(1, 1))])
    s.run_chain(renorm=True)
    got = s.chain_solution(0, s.chain_stats(0, 1)[0])
    assert np.abs(got.toarray() - st.toarray()).max() < 1e-15
    # mass pushed over the edge raises the boundary flag and is truncated
    shift = sparse.coo_matrix(([1.0], ([4], [8])), shape=(9, 9))      # 4 cells to the right
    s.set_state(st)
    s.fftconv2(shift)
    out = s.get_cursol([N, N])
    assert s.last_flag
    assert abs(out.sum() - 0.25) < 1e-14 and out.toarray()[3, 9] == pytest.approx(0.25, abs=1e-15)
    s.close()


@pytest.mark.parametrize('R,K,mode', [(600, 199, 'exact'),     # P = 1300 = 2^2 5^2 13: split columns, radix 13
                                      (1150, 547, 'exact'),    # P = 2574 = 2 3^2 11 13: both sub-passes generic
                                      (512, 257, 'fast'),      # P = 1153 (prime): fast size 1176, single pass
                                      (700, 301, 'auto')])     # P = 1551 = 3 11 47: split, generic
def test_chain_with_flags_at_awkward_sizes(hip_lib, R, K, mode):
    '''Mid-size pads that exercise the split column transform with wave-cooperative radices,
    with kernels that push mass over the boundary so the device-side flag and the predicated
    re-FFT run; checked against the CPU oracle (same torus, or 5e-8 in fast mode).'''
    from parasitoids_amd import synthetic
    from oracle import calcsol as OC
    nd = 4
    state, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=R + K, sigma=(6.0, 18.0),
                                             shift=K // 2 - 60)
    N = 2 * R + 1
    # start near the edge so that the first days already spill
    state = sparse.coo_matrix(([0.6, 0.4], ([40, N - 30], [N - 50, 25])), shape=(N, N))
    ms = np.array([K, K])
    ref = [state]
    trace = {}
    OC.get_solutions(ref, [None] + kernels, list(range(nd + 1)), nd + 1, N, ms, trace=trace)
    s = hip_lib.HipSolve(state, ms, mode=mode)
    s.set_kernels(kernels)
    s.run_chain(renorm=True)
    st = s.chain_stats(0, nd)
    exact = s.mode == 'exact'
    assert any(trace['flags']), 'fixture should raise the boundary flag'
    tol = 1e-12 if exact else 5e-8
    for d in range(nd):
        np.testing.assert_allclose(s.dense(0, d), trace['raw'][d], rtol=0, atol=tol)
        if exact:
            assert bool(st[d].flag) == bool(trace['flags'][d])
            assert abs(s.chain_solution(d, st[d]).tocsr() - ref[d + 1].tocsr()).max() < 1e-12
    s.close()


@pytest.mark.parametrize('R,K,nd', [(512, 513, 6), (1024, 1025, 4), (2048, 2049, 3),
                                    (800, 601, 4),      # 1920 = 16*12*10 (prime-factor radices)
                                    (1400, 1101, 3),    # 3360 = 16*15*14
                                    (2048, 3201, 2),    # 5760 = 16*20*18
                                    (2600, 2299, 2),    # 6400 = 16*20*20
                                    (2900, 2737, 2)])   # 7200 = 16*25*18 (radix 25)
def test_register_resident_row_kernels_match_lds_kernels(hip_lib, monkeypatch, R, K, nd):
    '''fft_rs.h kernels (sizes 1296, 2592, 5184: forward and inverse rows, incl. the re-FFT of
    flagged days) against the LDS-resident program on the same fast size: same flags, fields
    equal to round-off.'''
    from parasitoids_amd import synthetic
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=R, sigma=(5.0, 15.0), shift=K // 4)
    # two blobs, one near the edge so that days flag and the re-FFT path runs
    state = sparse.coo_matrix(([0.7, 0.3], ([R, N - 20], [R, 30])), shape=(N, N))
    out = []
    for rs in (False, True):
        if rs:
            monkeypatch.delenv('PS_NO_RS', raising=False)
        else:
            monkeypatch.setenv('PS_NO_RS', '1')
        s = hip_lib.HipSolve(state, [K, K], mode='fast')
        assert s.fft_len in (1296, 2592, 5184, 1920, 3360, 5760, 6400, 7200)
        out.append(s.fft_len)
        s.set_kernels(kernels)
        s.run_chain(renorm=True)
        st = s.chain_stats(0, nd)
        out.append(([s.dense(0, d) for d in range(nd)], [bool(x.flag) for x in st], [x.nnz for x in st]))
        s.close()
    assert out[0] == out[2]          # same FFT size with and without the rs kernels
    out = [out[1], out[3]]
    assert out[0][1] == out[1][1] and any(out[0][1])
    for a, b in zip(out[0][0], out[1][0]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-14)
    assert out[0][2] == out[1][2]


def _rs_sizes():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, 'parasitoids_amd', 'csrc', 'fft_rs_sizes.h')).read()
    return sorted(16 * int(a) * int(b) for a, b in re.findall(r'X\((\d+), (\d+)\)', txt))


def test_every_register_resident_size(hip_lib, monkeypatch):
    '''One short flagged chain per size of fft_rs_sizes.h (all 43 forward + inverse kernel
    instantiations) against the LDS-resident program on the same FFT size.'''
    from parasitoids_amd import synthetic, _lib as L
    lib = L.load()
    sizes = _rs_sizes()
    assert len(sizes) >= 40 and 5184 in sizes and 1296 in sizes
    done = 0
    for Lfft in sizes:
        # a domain / kernel shape whose fast size is exactly Lfft with and without the rs kernels
        cand = None
        for K in (2 * (Lfft // 6) + 1, 2 * (Lfft // 8) + 1, 2 * (Lfft // 5) + 1):
            N = Lfft - K // 2
            N -= (N + 1) % 2                       # odd domain
            for dn in (0, 2, 4, 6):
                n = N - dn
                monkeypatch.delenv('PS_NO_RS', raising=False)
                a = lib.ps_fast_size(n, K)
                monkeypatch.setenv('PS_NO_RS', '1')
                b = lib.ps_fast_size(n, K)
                if a == Lfft and b == Lfft:
                    cand = (n, K)
                    break
            if cand:
                break
        if cand is None:
            continue                               # the 7-smooth neighbour is smaller: rs size unused there
        N, K = cand
        R = N // 2
        _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=2, seed=Lfft, sigma=(4.0, 9.0), shift=K // 5)
        state = sparse.coo_matrix(([0.6, 0.4], ([R, N - 12], [R, 20])), shape=(N, N))
        res = []
        for rs in (False, True):
            if rs:
                monkeypatch.delenv('PS_NO_RS', raising=False)
            else:
                monkeypatch.setenv('PS_NO_RS', '1')
            try:
                s = hip_lib.HipSolve(state, [K, K], mode='fast')
            except L.HipError as e:
                assert not rs and e.code == L.PS_ERR_UNSUPPORTED and Lfft > 9700   # beyond the LDS-resident rows
                res.append(None)
                continue
            assert s.fft_len == Lfft
            s.set_kernels(kernels)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, 2)
            res.append(([s.dense(0, d) for d in range(2)], [bool(x.flag) for x in st]))
            s.close()
        if res[0] is None:
            # no LDS-resident program at this size: invariants of the transform instead
            for f in res[1][0]:
                assert f.min() > -1e-13 and 0.3 < f.sum() <= 1.0 + 1e-12
            assert any(res[1][1])
        else:
            assert res[0][1] == res[1][1], Lfft
            for a, b in zip(res[0][0], res[1][0]):
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-14, err_msg=str(Lfft))
        done += 1
    monkeypatch.delenv('PS_NO_RS', raising=False)
    assert done >= 35


def test_every_register_resident_size_in_fold_mode(hip_lib, monkeypatch):
    """PS_MODE_FOLD, one short chain with a flag per size of fft_rs_sizes.h: the three-launch day (state
    column transformed inside the day pass where the size has an ALT instance, truncation inside the next
    row pass, k_row_inv_fold) against the six-launch day of round 2 (PS_NO_FOLD_FUSE=1: separate forward
    column pass, k_row_inv_rs(p) + k_fold, k_truncate_if_flag).  Rows-then-columns against element-wise
    sums: round-off; flags and counts equal."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    done = flagged = 0
    for Lfft in _rs_sizes():
        m = Lfft // 9
        K = 2 * m + 1
        hit = None
        for dn in range(0, 40, 2):
            N = Lfft - 3 * m - dn
            N -= (N + 1) % 2                       # odd domain
            R = N // 2
            state = sparse.coo_matrix(([0.6, 0.4], ([R, N - 9], [R, 14])), shape=(N, N))
            monkeypatch.delenv('PS_NO_FOLD_FUSE', raising=False)
            s = hip_lib.HipSolve(state, [K, K], mode='fold', chain_only=True)
            ok = s.fft_len == Lfft and s.full_column
            if ok:
                hit = (N, R, state, s)
                break
            s.close()
            if s.fft_len < Lfft:
                break                              # the next smaller size serves everything below
        if hit is None:
            continue
        N, R, state, s = hit
        _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=3, seed=Lfft, sigma=(4.0, 9.0), shift=K // 5)
        res = []
        for fused in (True, False):
            if not fused:
                monkeypatch.setenv('PS_NO_FOLD_FUSE', '1')
                s = hip_lib.HipSolve(state, [K, K], mode='fold', chain_only=True)
            s.set_kernels(kernels)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, 3)
            res.append(([s.dense(0, d) for d in range(3)], [(bool(x.flag), x.nnz) for x in st], [x.sum for x in st]))
            s.close()
        flagged += any(f for f, _ in res[1][1])
        assert res[0][1] == res[1][1], Lfft
        for a, b in zip(res[0][0], res[1][0]):
            assert np.abs(a - b).max() < 1e-15, Lfft
        for a, b in zip(res[0][2], res[1][2]):
            assert abs(a - b) < 1e-13, Lfft
        done += 1
    monkeypatch.delenv('PS_NO_FOLD_FUSE', raising=False)
    assert done >= 35 and flagged >= done // 2, (done, flagged)   # the truncation path ran for most sizes


def test_direct_sum_subpass_matches_fft_subpass(hip_lib, monkeypatch):
    """Compact day kernels on a split column transform take the direct-sum first column
    sub-pass inside the fused kernel (kt_direct_fill); PS_NO_DIRECT=1 forces the FFT sub-pass.
    Same chains, both ways: supports on one side of the kernel centre only, a single row,
    spans at and beyond the term limit (fallback), an empty day kernel, mixed shapes; fast
    mode (multi-day fused passes; single-day pass and re-FFT once flags fire); fold mode runs
    the same chains for reference (it never takes the direct route)."""
    monkeypatch.setenv('PS_TPIPE', '0')   # the tiled pipeline is what this test is about
    R, K = 640, 801
    N, M = 2 * R + 1, K // 2
    rng = np.random.default_rng(77)
    st = sparse.coo_matrix((rng.random(300) + 0.1,
                            (rng.integers(R - 60, R + 60, 300), rng.integers(R - 60, R + 60, 300))), shape=(N, N))
    st = (st / st.sum()).tocoo()

    def kern(rows_lo, rows_hi, shape=K, n=400, seed=0):
        r = np.random.default_rng(seed)
        m = shape // 2
        if rows_hi < rows_lo:
            return sparse.coo_matrix((shape, shape))
        k = sparse.coo_matrix((r.random(n) + 0.05,
                               (r.integers(rows_lo, rows_hi + 1, n), r.integers(m - 40, m + 41, n))),
                              shape=(shape, shape))
        k.sum_duplicates()
        return (k / k.sum()).tocoo()

    spans = [(M - 20, M + 20), (M + 3, M + 30), (M - 35, M - 2), (M, M), (0, K - 1), (1, 0),
             (M - 70, M + 69), (M - 5, M + 5), (M - 12, M + 9), (M - 1, M + 40), (M - 40, M + 1)]
    kernels = [kern(lo, hi, seed=i) for i, (lo, hi) in enumerate(spans)]
    kernels.append(kern(40, 60, shape=101, seed=99))   # smaller shape inside the staging box
    kernels.append(kern(M - 8, M + 8, seed=100))
    kernels.append(kern(M - 3, M + 25, seed=101))
    nd = len(kernels)
    compact = [k for i, k in enumerate(kernels) if spans[min(i, len(spans) - 1)] != (0, K - 1)]
    assert len(compact) % 2 == 1    # windows 2, 4, 6 -> one single-day fused pass at the end
    # a second state next to the domain edge: boundary flags fire, speculation ends, the
    # single-day fused pass and the flag-conditional re-FFT run with direct-sum kernels
    edge = sparse.coo_matrix((rng.random(50) + 0.1, (rng.integers(2, 30, 50), rng.integers(R - 20, R + 20, 50))),
                             shape=(N, N))
    edge = (edge / edge.sum()).tocoo()
    flagged = False
    for mode, ks, st in (('fast', kernels, st), ('fast', compact, st), ('fold', compact, st), ('fold', kernels, st),
                         ('fast', compact, edge), ('fold', compact, edge)):
        out = {}
        for tag in ('direct', 'fft'):
            if tag == 'fft':
                monkeypatch.setenv('PS_NO_DIRECT', '1')
            else:
                monkeypatch.delenv('PS_NO_DIRECT', raising=False)
            s = hip_lib.HipSolve(st, [K, K], mode=mode, chain_only=(mode == 'fold'))
            assert s.fft_len > 1200      # split column transform
            s.set_kernels(ks)
            s.run_chain(renorm=False)
            stats = s.chain_stats(0, len(ks))
            out[tag] = ([s.dense(0, d) for d in range(len(ks))], [x.flag for x in stats], s.kernels_direct)
            s.close()
        assert out['fft'][2] is False
        # the full-height kernel falls back; fold mode always takes the FFT sub-pass
        assert out['direct'][2] is (ks is compact and mode == 'fast')
        assert out['direct'][1] == out['fft'][1]
        flagged = flagged or (st is edge and any(out['direct'][1]))
        for a, b in zip(out['direct'][0], out['fft'][0]):
            assert np.abs(a - b).max() <= 1e-15 * max(1.0, np.abs(b).max()) + 2e-17
    assert flagged


@pytest.mark.parametrize('R,K,mode', [(640, 801, 'fast'), (600, 221, 'exact'), (400, 201, 'exact'), (700, 301, 'fast')])
def test_multi_day_fused_pass_is_bit_identical(hip_lib, monkeypatch, R, K, mode):
    """Un-flagged days go through the fused column pass in groups of up to 8
    (k_col_fused_multi): the same products in the same order as single-day passes, so every
    record must be BIT-identical to PS_FUSED_DAYS=1 -- split and single-pass column
    transforms, smooth and generic (prime factor 19, 23, 17, 53) plans, 11 days (groups of
    2, 4, 4 + 1).  The direct-sum route is switched off here: it changes round-off."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_NO_DIRECT', '1')
    monkeypatch.setenv('PS_TPIPE', '0')   # k_col_fused_multi belongs to the tiled pipeline
    nd = 11
    state, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=3, sigma=(2.0, 6.0), shift=4.0)
    out = {}
    for days in ('1', '8'):
        monkeypatch.setenv('PS_FUSED_DAYS', days)
        s = hip_lib.HipSolve(state, [K, K], mode=mode)
        s.set_kernels(kernels)
        s.run_chain(renorm=False)
        st = s.chain_stats(0, nd)
        assert not any(x.flag for x in st)
        out[days] = [s.dense(0, d) for d in range(nd)]
        s.close()
    for a, b in zip(out['1'], out['8']):
        assert np.array_equal(a, b)


def test_full_column_pipeline_matches_tiled_pipeline(hip_lib, monkeypatch):
    """The full-column pipeline (k_colfull, column-major spectra; what ps_chain_run picks for
    broad day kernels on register-resident FFT sizes in fast mode) against the tiled pipeline on
    the same fast torus: chains with boundary flags (re-FFT through the full-column forward
    pass), the per-call API (fftconv2 / get_cursol / back_solve incl. the cached filter spectra),
    at two register-resident sizes.  Same arithmetic up to round-off: 1e-14."""
    from parasitoids_amd import _lib as L
    rng = np.random.default_rng(5)
    for R, K in ((640, 801), (1000, 1201)):
        N = 2 * R + 1

        def kern(seed, n=3000):
            r = np.random.default_rng(seed)
            k = sparse.coo_matrix((r.random(n) + 0.05, (r.integers(0, K, n), r.integers(K // 2 - 60, K // 2 + 61, n))),
                                  shape=(K, K))
            k.sum_duplicates()
            return (k / k.sum()).tocoo()

        kernels = [kern(i) for i in range(6)]
        # mass next to the upper domain edge: the broad kernels push it over -> flags
        st = sparse.coo_matrix((rng.random(80) + 0.1, (rng.integers(2, 40, 80), rng.integers(R - 30, R + 30, 80))),
                               shape=(N, N))
        st = (st / st.sum()).tocoo()
        out = {}
        for tag in ('full_column', 'tiled'):
            monkeypatch.setenv('PS_TPIPE', '1' if tag == 'full_column' else '0')
            s = hip_lib.HipSolve(st, [K, K], mode='fast', chain_only=True)
            s.set_kernels(kernels)
            s.run_chain(renorm=True)
            stats = s.chain_stats(0, len(kernels))
            assert s.full_column == (tag == 'full_column')
            fields = [s.dense(0, d) for d in range(len(kernels))]
            flags = [bool(x.flag) for x in stats]
            # per-call API on a fresh state: two day steps, then the back-solve of two filters, twice
            # (the second call takes the filters' spectra from the cache)
            s.set_state(st)
            per = []
            for d in range(2):
                s.fftconv2(kernels[d])
                per.append(s.get_cursol([N, N]).toarray())
            filt = [sparse.coo_matrix(np.pad(kernels[d].toarray(), (N - K) // 2)) for d in (2, 3)]
            for rep in range(2):
                per += [m.toarray() for m in s.back_solve(filt, [N, N])]
            out[tag] = (fields, flags, per, s.fft_len)
            s.close()
        assert out['full_column'][3] == out['tiled'][3] and out['tiled'][3] > 1200
        assert out['full_column'][1] == out['tiled'][1] and any(out['tiled'][1])
        for a, b in zip(out['full_column'][0] + out['full_column'][2], out['tiled'][0] + out['tiled'][2]):
            assert np.abs(a - b).max() <= 1e-14 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize('R,K,fft', [(1200, 1601, 3360), (1300, 1801, 3584)])
def test_two_role_chained_pass_is_bit_identical(hip_lib, monkeypatch, R, K, fft):
    """Long chained groups go through k_colfull_dual (two roles per workgroup: the inverse of day d
    next to the forward transform of day d + 1, half-pass exchanges, inverse = conj-forward-conj,
    kernel columns staged HBM -> LDS), and the batched row pass then leaves a day's pad-only row pairs
    unread when the column pass found their total energy too small to matter.  Records, flags and
    statistics must be BIT-identical to the single-role chained pass (PS_DUAL_MIN_DAYS=0) and to the
    row pass that reads every pair (PS_NO_PAD_QUIET=1) -- on a clean chain (every day quiet), on one
    whose mass reaches the pad without raising the flag, and on one that flags.  FFT size 3360 =
    16 x 15 x 14, the smallest size the two-role pass serves; and 3584, whose 1793 columns are seven
    rounds of 256 CUs plus ONE: that column leaves the chained pass and goes through the forward /
    prefix-product / inverse launches (conv_inv_multi; PS_NO_TAIL_SPLIT=1 keeps it in)."""
    from parasitoids_amd import synthetic
    monkeypatch.setenv('PS_TPIPE', '1')
    nd = 14
    N = 2 * R + 1
    _, kernels, _ = synthetic.make_stack(R=R, K=K, ndays=nd, seed=11, sigma=(15.0, 40.0), shift=40)
    for start, want_flag in ((R, False), (N - 251, None), (N - 71, True)):
        state = sparse.coo_matrix(([1.0], ([start], [start])), shape=(N, N))
        runs = {}
        for tag, env in (('dual', {}), ('single', {'PS_DUAL_MIN_DAYS': '0'}), ('read_all', {'PS_NO_PAD_QUIET': '1'}),
                         ('no_tail', {'PS_NO_TAIL_SPLIT': '1'}),
                         ('row2', {'PS_ROW2': '1'}),           # the two-role row kernel (default only from 4096 points on)
                         # the intermediate row-major instead of row pairs interleaved (two-role pass, tail columns)
                         ('row_major', {'PS_NO_PAIR_ROWS': '1'}), ('row_major_row2', {'PS_NO_PAIR_ROWS': '1', 'PS_ROW2': '1'})):
            for k in ('PS_DUAL_MIN_DAYS', 'PS_NO_PAD_QUIET', 'PS_NO_TAIL_SPLIT', 'PS_ROW2', 'PS_NO_PAIR_ROWS'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            s = hip_lib.HipSolve(sparse.coo_matrix(([1.0], ([R], [R])), shape=(N, N)), [K, K], mode='fast', chain_only=True)
            assert s.fft_len == fft and s.full_column
            s.set_kernels(kernels)
            s.run_chain(renorm=True)           # clean first run: the next one opens with one 14-day window
            assert not any(x.flag for x in s.chain_stats(0, nd))
            s.set_state(state)
            s.prof_enable(True, every=1)
            s.run_chain(renorm=True)
            st = s.chain_stats(0, nd)
            if not any(x.flag for x in st):
                assert s.prof_days()['col_inv_a_xn'] == nd       # one chained launch for the whole run
            runs[tag] = ([s.dense(0, d) for d in range(nd)], [(x.flag, x.nnz, x.sum, x.delta, x.padmax) for x in st])
            s.close()
        flags = [f for f, *_ in runs['single'][1]]
        if want_flag is not None:
            assert any(flags) == want_flag, flags
        for tag in runs:
            assert runs[tag][1] == runs['single'][1], tag
            for a, b in zip(runs[tag][0], runs['single'][0]):
                assert np.array_equal(a, b), tag
