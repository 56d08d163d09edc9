"""Process-wide backend switches read by CalcSol (cf. reference globalvars.py:5).

`cuda` keeps the reference's name so `Run.main` / `Bayes_Run` style callers can set it
unchanged; here it selects the HIP backend (there is no CPU path in this package).
`fft_mode`: 'exact' transforms on the reference's pad size P = N + K//2; 'fold' computes the
same torus as a linear convolution on a fast FFT size folded back modulo P (chain API only);
'fast' uses a convenient FFT size >= P; 'auto' (default) = exact-torus results by the
cheapest of the first two (see DESIGN.md section 5 for the tolerance each mode carries).
"""
cuda = True
fft_mode = 'auto'
