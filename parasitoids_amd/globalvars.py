"""Process-wide backend switches read by CalcSol (cf. reference globalvars.py:5).

`cuda` keeps the reference's name so `Run.main` / `Bayes_Run` style callers can set it
unchanged; here it selects the HIP backend (there is no CPU path in this package).
`fft_mode`: 'exact' transforms on the reference's pad size P = N + K//2, 'fast' on the
next even 7-smooth size, 'auto' (default) = exact whenever P can be planned (all prime
factors <= 1024), else fast (see DESIGN.md for the tolerance each mode carries).
"""
cuda = True
fft_mode = 'auto'
