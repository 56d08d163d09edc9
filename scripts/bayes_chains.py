"""k chains side by side on one GPU: evaluations/hour for the k given on the command line.
    python scripts/bayes_chains.py 400 auto 1 2 4 6 8"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_extras as B   # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
mode = sys.argv[2] if len(sys.argv) > 2 else 'auto'
ks = [int(v) for v in sys.argv[3:]] or [1, 2, 4]
for k in ks:
    rec = B.bayes_multi_case(R, mode, k, 400, 40)
    print(json.dumps({x: rec[x] for x in ('chains_per_gpu', 'value', 'evaluations_per_hour', 'ms_per_sample_aggregate',
                                          'evaluation_fraction', 'per_chain_samples_per_hour')}), flush=True)
