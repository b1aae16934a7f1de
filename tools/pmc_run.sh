#!/bin/bash
# tools/pmc_run.sh <tag> "<counter list>" [env assignments...]  -- one bounded rocprofv3 PMC pass of bench.py
# (counters in their own run with --kernel-trace only; every pass wrapped in `timeout`)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
set_=$1; shift
mkdir -p gpurun_out
env "$@" timeout 70 rocprofv3 --pmc $set_ --kernel-trace -d gpurun_out/pmc_$tag -o p -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 > gpurun_out/pmc_$tag.out 2> gpurun_out/pmc_$tag.err
echo "pmc_$tag rc=$?"
grep -E "Memory access fault|Segmentation|error" gpurun_out/pmc_$tag.err | head -3
