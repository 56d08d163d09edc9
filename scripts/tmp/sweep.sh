run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k.startswith('col_inv_a')})"; }
PS_FUSED_G=2 PS_FUSED_THREADS=512 run g2d4t512
PS_FUSED_G=2 PS_FUSED_THREADS=768 run g2d4t768
PS_FUSED_G=2 PS_FUSED_THREADS=1024 run g2d4t1024
PS_FUSED_G=2 PS_DIRECT_DAYS=2 PS_FUSED_THREADS=512 run g2d2t512
PS_FUSED_G=2 PS_DIRECT_DAYS=2 PS_MULTI_WSH=4 PS_FUSED_THREADS=512 run g2d2w4t512
PS_FUSED_G=1 PS_FUSED_THREADS=512 run g1d4t512
PS_FUSED_G=1 PS_DIRECT_DAYS=8 PS_FUSED_THREADS=512 run g1d8t512
PS_FUSED_G=2 PS_DIRECT_DAYS=8 PS_FUSED_THREADS=1024 run g2d8t1024
PS_NO_DIRECT=1 PS_FUSED_THREADS=512 run nodirect512
