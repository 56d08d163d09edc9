run() { BENCH_NO_PROF=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])"; }
export PS_NO_SPECULATION=1
PS_NO_DIRECT=1 run nospec_fft
run nospec_direct_g1
PS_SINGLE_G2=1 run nospec_direct_g2
PS_SINGLE_G2=1 PS_FUSED_THREADS=256 run nospec_direct_g2_t256
