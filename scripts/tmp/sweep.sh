run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k.startswith('col_inv_a')})"; }
run base
run base2
