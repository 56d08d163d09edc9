#!/bin/bash
# LDS side of the two hot kernels of the headline stack (rocprofv3 counter passes over a short bench run):
# bank-conflict cycles against all LDS-array cycles, LDS instructions, issue stalls, wave cycles.
# usage (GPU box, repo root): scripts/pmc_lds.sh TAG
set -e
tag=${1:-lds}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
  local name=$1; shift
  BENCH_NO_PROF=1 rocprofv3 --kernel-trace --pmc "$@" -d "$out/$name" -o pmc --output-format csv -- \
    python3 "$root/bench.py" --steps 3 --warmup 2 --no-extras --no-cpu-baseline > "$out/$name.log" 2>&1 || tail -3 "$out/$name.log"
  python3 "$root/scripts/pmc_summary.py" "$out/$name" | grep -E "k_colfull_dual|k_row_inv_rs2|k_row_fwd_rs" > "$out/$name.txt" || true
  cat "$out/$name.txt"
}
pass lds1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
pass lds2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass lds3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
find "$out" -name "*counter_collection.csv" -delete
find "$out" -name "*kernel_trace.csv" -delete
