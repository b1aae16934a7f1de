#!/bin/bash
# rocprofv3 kernel-trace of the default bench command (no self-play, no cpu baseline): per-kernel durations in steady state
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --selfplay-seconds 0 ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof/kt.out 2> $GRAFT_REPO_ROOT/gpurun_out/prof/kt.err); echo "kernel-trace rc=$?"
find gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -14'
tail -1 gpurun_out/prof/kt.out | cut -c1-400
