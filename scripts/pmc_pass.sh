#!/bin/bash
# One rocprofv3 counter pass over a short bench run, summarised per kernel.
# usage: scripts/pmc_pass.sh TAG COUNTER [COUNTER ...]   (run on the GPU box from the repo root)
# Counters are collected in their own run (--kernel-trace only), as the pool requires.
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
BENCH_NO_PROF=1 rocprofv3 --kernel-trace --pmc "$@" -d "$out" -o pmc --output-format csv -- \
  python3 "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/bench.log" 2>&1
python3 "$root/scripts/pmc_summary.py" "$out" | tee "$out/summary.txt"
