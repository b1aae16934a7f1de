#!/bin/bash
# configs[4] (40b x 384, mixed boards) under rocprofv3 --kernel-trace --stats: per-kernel launches and durations of the chained pass and
# the one-chain pass of bench.py's config5 segment together (a launch that shares the chip with another chain's is longer than alone)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5kt; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for ch in 0 1; do
  if [ $ch = 0 ]; then unset SAYURI_CHAINS; else export SAYURI_CHAINS=1; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt$ch -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --selfplay-seconds 0 --no-pump --steps 4 --warmup 1 --config5 > $GRAFT_REPO_ROOT/$O/kt$ch.out 2> $GRAFT_REPO_ROOT/$O/kt$ch.err)
  echo "## SAYURI_CHAINS=${SAYURI_CHAINS:-auto} rc=$?  (config5 of the same command: $(tail -1 $O/kt$ch.out | python -c "import json,sys;d=json.loads(sys.stdin.read())['config5'];print({k:d[k] for k in ('chains','evals_per_sec','ms_per_step','evals_per_sec_one_chain','tower_conv_avg_launch_us')})"))"
  find $O/kt$ch -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-220 | head -16
done > $O/summary.txt 2>&1
cat $O/summary.txt
find $O -name "*.csv" -delete
