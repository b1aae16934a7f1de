#!/bin/bash
# the driver-style default line and the kernel-trace stats of the same tree (the short form of profile.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r05}
O=gpurun_out/$TAG
mkdir -p $O/prof
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump"
rm -rf $O/prof/kt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof/kt -o p --output-format csv -- $B --steps 100 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof/kt.out 2> $GRAFT_REPO_ROOT/$O/prof/kt.err); echo "kernel-trace rc=$?"
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump --steps 100 --warmup 10   ($(date -u +%FT%TZ))"
  echo "# bench line of the same command:"; tail -1 $O/prof/kt.out | sed 's/^/# /' | cut -c1-1500
  find $O/prof/kt -name "*kernel_stats.csv" | head -1 | xargs cat; } > $O/rocprofv3_kernel_stats.txt
head -7 $O/rocprofv3_kernel_stats.txt | cut -c1-200
rm -rf $O/prof/kt/*/p_kernel_trace.csv $O/prof/kt/*/p_agent_info.csv
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
tail -1 $O/bench_default.json | cut -c1-400
