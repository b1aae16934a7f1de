#!/bin/bash
# Round 5: the pump (two tickets in flight through submit / wait) runs 60-100 us per batch behind the microbench (one forward after
# another on one stream).  Kernel and copy timestamps of one bench run, and what lies between consecutive tower launches in either
# mode -- with the packed records read in place (default) and copied first (SAYURI_IO_ZC_IN=0, rounds 2-4).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for zc in 1 0 1 0; do
  O=gpurun_out/pump_gaps/zc_in_$zc
  rm -rf $O; mkdir -p $O
  export SAYURI_IO_ZC_IN=$zc
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/$O/trace -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-config5 --selfplay-seconds 0 > $GRAFT_REPO_ROOT/$O/bench.out 2> $GRAFT_REPO_ROOT/$O/bench.err); echo "SAYURI_IO_ZC_IN=$zc trace rc=$?"
  tail -1 $O/bench.out | python -c "import json,sys;d=json.loads(sys.stdin.read());print('under the tracer:', d['value'], d['ms_per_step'], d['config'].get('pump'), d['config'].get('pump_packed'))" | cut -c1-400
  python tools/pump_gaps.py $O/trace --detail 2 > $O/gaps.txt 2>&1; cat $O/gaps.txt | cut -c1-200
  find $O/trace -name "*.csv" -delete
  # and without the tracer
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 > $O/bench_plain.out 2> $O/bench_plain.err
  tail -1 $O/bench_plain.out | python -c "import json,sys;d=json.loads(sys.stdin.read());print('plain:', d['value'], d['ms_per_step'], 'pump', d['config']['pump']['nn_evals_per_sec'], 'pump_packed', d['config']['pump_packed']['nn_evals_per_sec'], d['config']['pump_packed']['ms_per_batch'])"
done
unset SAYURI_IO_ZC_IN
