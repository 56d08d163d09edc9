"""`python bench.py --gpus N` must start N ranks itself (the driver calls it exactly like
that) -- exercised here on CPU through the launcher path with gloo and no device work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *argv, env=None):
    e = dict(os.environ, BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + list(argv), env=e,
                          capture_output=True, text=True, timeout=300)


def test_bench_gpus2_launches_two_ranks():
    p = _run('bench.py', '--gpus', '2', '--rehearse')
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                      # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['rehearsal'] is True
    # BASELINE configs 4 and 5 over the ranks (stand-in work here): members round-robin and
    # gathered in member order, one chain per rank with seeds 1000 + rank, per-rank parity figures
    mg = d['multi_gpu']
    assert mg['n_gpus'] == 2
    ens = mg['ensemble']
    assert ens['members_gathered'] == 8 and len(ens['per_rank_members_per_s']) == 2 and ens['value'] > 0
    by = mg['bayes']
    assert by['seeds'] == [1000, 1001] and len(by['per_rank_samples_per_hour']) == 2
    assert abs(by['value'] - sum(by['per_rank_samples_per_hour'])) < 1.0
    assert by['per_rank_samples_per_hour'][1] - by['per_rank_samples_per_hour'][0] == 1.0   # each rank ran its own seed
    assert d['parity']['per_rank_max_abs'] == [1e-19, 2e-19]


def test_bench_refuses_a_world_that_is_not_gpus():
    # as one rank of a 3-rank torchrun job but told --gpus 2: must not silently report n_gpus 1
    p = _run('bench.py', '--gpus', '2', '--rehearse', env={'RANK': '0', 'WORLD_SIZE': '3', 'LOCAL_RANK': '0'})
    assert p.returncode != 0 and 'WORLD_SIZE 3 != --gpus 2' in p.stderr
    p = _run('bench.py', '--gpus', '1', '--rehearse', env={'RANK': '0', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0'})
    assert p.returncode != 0


def test_bench_bayes_gpus2_launches_two_ranks():
    p = _run('bench_bayes.py', '--gpus', '2', '--rehearse')
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    assert json.loads(lines[0])['n_gpus'] == 2


def test_a_failing_rank_does_not_hang_the_multi_gpu_record():
    """every rank takes part in every collective of multi_gpu_record; a stage that fails on
    one rank arrives on rank 0 as an error record instead of leaving the others in a barrier."""
    for stage, other in (('ensemble', 'bayes'), ('bayes', 'ensemble')):
        p = _run('bench.py', '--gpus', '2', '--rehearse', env={'BENCH_FAIL_STAGE': stage})
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
        mg = d['multi_gpu']
        assert 'error' in mg[stage] and any('rank 1' in m for m in mg[stage]['error']), mg[stage]
        assert 'error' not in mg[other]
        assert d['world_size'] == 2 and [r['rank'] for r in d['ranks']] == [0, 1]
