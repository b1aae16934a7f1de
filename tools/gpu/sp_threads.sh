#!/bin/bash
# round 4: how many scheduler threads should carry the 512 fibers?  (default 4 x usable cores = 64: a game whose result is back waits for
# its thread to get to it; with more threads fewer games are on the host at any moment and the two batches in flight hold more of the 512)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/spth
python bench.py --steps 30 --warmup 5 --selfplay-seconds 0 --no-cpu-baseline --no-config5 --no-pump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench', d['value'], d['ms_per_step'])"
run() {
  local name=$1; shift
  ( env SAYURI_PIPE_TRACE=1 timeout 300 python tools/selfplay_bench.py --seconds 50 --games 512 --stagger 360 "$@" ) 2> gpurun_out/spth/$name.err | tail -1 > gpurun_out/spth/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/spth/%s.json'%n))
    sh=d.get('second_half',{})
    print("== %-10s evals/s %8.1f (2nd half %8.1f) mean_batch %.1f partial %d/%d cores %.1f sys %.1f ctx/s %d"%(n,d['nn_evals_per_sec'],sh.get('nn_evals_per_sec',0),d['mean_batch'],d['partial_batches'],d['batches'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['ctx_switches_per_sec']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  grep -h "runs again:" gpurun_out/spth/$name.err | cut -c1-200
}
if [ -n "$SWEEP" ]; then for t in $SWEEP; do run t$t --game-threads $t; done; exit 0; fi
run t64 --game-threads 64
run t128 --game-threads 128
run t256 --game-threads 256
run t64b --game-threads 64
run t192 --game-threads 192
