"""CPU oracle: numpy/scipy restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`parasitoids_amd/`) imports this; only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg do, and only as the checker.

Pinned against the reference itself: `tests/golden/make_golden.py` imports the
reference CPU path (`/root/reference/{ParasitoidModel,CalcSol}.py`) in the build
container and stores its outputs as fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function here against them.
"""
