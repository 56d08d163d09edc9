#!/bin/bash
# SQ counters of the prob_mass pair kernel, run-time-count instance (exp() calls) next to the unrolled one
# (pm_exp_many): what k_pair_masses waits on.  Run on the GPU box from the repo root.
# usage: scripts/pmc_prob_mass.sh TAG   -> gpurun_out/pmc_pm_TAG/summary.txt
set -e
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_pm_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM \
  -d "$out/a" -o pmc --output-format csv -- python3 "$root/scripts/time_prob_mass.py" 0.253 > "$out/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
  -d "$out/b" -o pmc --output-format csv -- python3 "$root/scripts/time_prob_mass.py" 0.253 > "$out/b.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/c" -o st --output-format csv -- python3 "$root/scripts/time_prob_mass.py" 0.253 > "$out/c.log" 2>&1
{ echo "== pass a"; python3 "$root/scripts/pmc_summary.py" "$out/a" | grep -i "pair_masses\|tile_acc\|k_periods"; echo "== pass b"; python3 "$root/scripts/pmc_summary.py" "$out/b" | grep -i "pair_masses\|tile_acc\|k_periods";
  echo "== kernel stats"; grep -h "pair_masses" $(find "$out/c" -name "*kernel_stats.csv") ; } | tee "$out/summary.txt"
find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete
