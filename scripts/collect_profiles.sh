#!/bin/bash
# Collect the judged artefacts of one round on the GPU box: bench line, rocprofv3 kernel
# stats of the same command, HBM traffic counters (two passes), Bayes bench, parity report,
# HBM calibration.  usage: scripts/collect_profiles.sh TAG   (from the repo root; writes
# gpurun_out/TAG_* -- copy into profiles/ afterwards)
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
python3 "$root/bench.py" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
python3 "$root/bench_bayes.py" > "$out/${tag}_bench_bayes.json" 2>> "$out/${tag}_bench.err"
python3 "$root/bench_bayes.py" --mode fast --no-cpu-baseline >> "$out/${tag}_bench_bayes.json" 2>> "$out/${tag}_bench.err"
python3 "$root/scripts/run_mcmc.py" --samples 200 > "$out/${tag}_mcmc.json" 2>> "$out/${tag}_bench.err"
python3 "$root/scripts/run_mcmc.py" --samples 200 --mode fast >> "$out/${tag}_mcmc.json" 2>> "$out/${tag}_bench.err"
for m in auto exact; do python3 "$root/bench.py" --mode $m --steps 3 --warmup 1 --no-cpu-baseline >> "$out/${tag}_bench_modes.json" 2>> "$out/${tag}_bench.err"; done
for cfg in "1024 1025" "512 513" "1024 1641" "2048 3201"; do set -- $cfg; python3 "$root/bench.py" --rad-res $1 --kshape $2 --no-cpu-baseline >> "$out/${tag}_bench_other_sizes.json" 2>> "$out/${tag}_bench.err"; done
python3 "$root/scripts/hbm_calib.py" > "$out/${tag}_hbm_calibration.txt" 2>&1
python3 "$root/tests/parity_report.py" > "$out/${tag}_parity_report.txt" 2>&1 || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/${tag}_kt" -o kt --output-format csv -- \
  python3 "$root/bench.py" --no-cpu-baseline > "$out/${tag}_bench_under_rocprof.json" 2>> "$out/${tag}_bench.err"
cp "$out/${tag}_kt"/*kernel_stats.csv "$out/${tag}_bench_kernel_stats.csv"
BENCH_NO_PROF=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/${tag}_pmc_fetch" -o pmc --output-format csv -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> "$out/${tag}_bench.err"
BENCH_NO_PROF=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/${tag}_pmc_write" -o pmc --output-format csv -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> "$out/${tag}_bench.err"
python3 "$root/scripts/hbm_traffic.py" "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write" "$out/${tag}_hbm_traffic_pmc.json"
rm -rf "$out/${tag}_kt" "$out/${tag}_pmc_fetch" "$out/${tag}_pmc_write"
echo done
