#!/bin/bash
# Run from the build container: leaves the commit in .git_head (the GPU box gets the tree without
# .git), ships the tree and collects the round's rocprofv3 evidence (scripts/collect_profiles.sh).
# usage: scripts/gpu_profiles.sh TAG      -> gpurun_out/prof_TAG/
set -e
cd "$(dirname "$0")/.."
git rev-parse HEAD > .git_head
exec /usr/local/graft/bin/gpurun --timeout 1100 -- "bash scripts/collect_profiles.sh ${1:-r03}"
