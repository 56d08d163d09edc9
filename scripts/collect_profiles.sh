#!/bin/bash
# rocprofv3 evidence for one round, run on the GPU box from the repo root:
#   kernel-trace statistics of the default bench.py command and of the real-wind chains
#   (bench_extras.py real_wind), and HBM traffic per kernel from two PMC passes each
#   (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; counters are collected in their
#   own runs with --kernel-trace only, as the pool requires).
# usage: scripts/collect_profiles.sh TAG      -> gpurun_out/prof_TAG/
set -e
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# provenance: which source tree and which library build these numbers belong to (the GPU box has no
# .git: scripts/gpu_profiles.sh leaves the commit in .git_head before it ships the tree)
{ echo "git_head $(cat "$root/.git_head" 2>/dev/null || git -C "$root" rev-parse HEAD 2>/dev/null || echo unknown)";
  echo "lib_sha256 $(sha256sum "$root/parasitoids_amd/libparasitoid_hip.so" | cut -d' ' -f1)";
  echo "date $(date -u +%Y-%m-%dT%H:%M:%SZ)"; } > "$out/provenance.txt"
run() {  # name, rocprof args..., -- program args
  local name=$1; shift
  rocprofv3 "$@" > "$out/$name.log" 2>&1 || { echo "rocprofv3 failed for $name"; tail -5 "$out/$name.log"; }
}
# 1. kernel-trace statistics of the bench command (HIP-event profiling on, as in the driver's run)
run bench_stats --kernel-trace --stats -d "$out/bench_stats" -o b --output-format csv -- \
  python3 "$root/bench.py" --steps 5 --warmup 2 --no-extras --no-cpu-baseline
# 2. HBM traffic of the same command
BENCH_NO_PROF=1 run bench_fetch --kernel-trace --pmc FETCH_SIZE -d "$out/bench_fetch" -o p --output-format csv -- \
  python3 "$root/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline
BENCH_NO_PROF=1 run bench_write --kernel-trace --pmc WRITE_SIZE -d "$out/bench_write" -o p --output-format csv -- \
  python3 "$root/bench.py" --steps 4 --warmup 2 --no-extras --no-cpu-baseline
python3 "$root/scripts/hbm_traffic.py" "$out/bench_fetch" "$out/bench_write" "$out/bench_hbm_traffic_pmc.json" "$out/provenance.txt" > "$out/bench_hbm_traffic.txt"
# 3. the real-wind chains (Carnarvon, R = 2048: full-column pipeline, flags, fold path)
run rw_stats --kernel-trace --stats -d "$out/rw_stats" -o b --output-format csv -- \
  python3 "$root/bench_extras.py" real_wind
run rw_fetch --kernel-trace --pmc FETCH_SIZE -d "$out/rw_fetch" -o p --output-format csv -- \
  python3 "$root/bench_extras.py" real_wind
run rw_write --kernel-trace --pmc WRITE_SIZE -d "$out/rw_write" -o p --output-format csv -- \
  python3 "$root/bench_extras.py" real_wind
python3 "$root/scripts/hbm_traffic.py" "$out/rw_fetch" "$out/rw_write" "$out/rw_hbm_traffic_pmc.json" "$out/provenance.txt" > "$out/rw_hbm_traffic.txt"
# 4. the prob_mass half of a Bayes evaluation: kernel statistics and HBM bytes of the 18-day batch
#    (bench_extras.py prob_mass builds it 1 + 10 times)
run pm_stats --kernel-trace --stats -d "$out/pm_stats" -o b --output-format csv -- \
  python3 "$root/bench_extras.py" prob_mass
run pm_fetch --kernel-trace --pmc FETCH_SIZE -d "$out/pm_fetch" -o p --output-format csv -- \
  python3 "$root/bench_extras.py" prob_mass
run pm_write --kernel-trace --pmc WRITE_SIZE -d "$out/pm_write" -o p --output-format csv -- \
  python3 "$root/bench_extras.py" prob_mass
(cd "$root/scripts" && python3 prob_mass_traffic.py "$out/pm_fetch" "$out/pm_write" 11 "$out/prob_mass_hbm_traffic.json") > "$out/prob_mass_hbm_traffic.txt"
# 4b. a whole Bayes evaluation (18 x prob_mass + 17-day exact-torus chain, R = 400, auto mode): kernel statistics
run bayes_stats --kernel-trace --stats -d "$out/bayes_stats" -o b --output-format csv -- \
  python3 "$root/bench_bayes.py" --evals 60 --warmup 5 --no-cpu-baseline
# 5. one stack launch by launch (start, duration, idle gap before every kernel) from the bench trace
python3 "$root/scripts/trace_timeline.py" "$out/bench_stats" -150 150 > "$out/stack_timeline.txt" 2>&1 || true
find "$out" -name "*kernel_stats.csv" | head
# keep the merge small: raw traces stay on the box
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*counter_collection.csv" -delete
du -sh "$out"
