run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])"; }
PS_FUSED_DAYS=4 run d4
PS_FUSED_DAYS=4 PS_CHUNK_DAYS=16 run d4c16
PS_FUSED_DAYS=4 PS_CHUNK_DAYS=16 PS_FUSED_THREADS=512 run d4c16t512
PS_FUSED_DAYS=8 PS_CHUNK_DAYS=16 run d8c16
PS_FUSED_DAYS=8 PS_CHUNK_DAYS=16 PS_FUSED_THREADS=512 run d8c16t512
PS_FUSED_DAYS=8 PS_CHUNK_DAYS=16 PS_MULTI_WSH=2 run d8c16w2
PS_FUSED_DAYS=8 PS_CHUNK_DAYS=30 run d8c30
PS_FUSED_DAYS=8 run d8
PS_FUSED_DAYS=4 run d4again
PS_FUSED_DAYS=1 run d1
