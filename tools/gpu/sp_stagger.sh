#!/bin/bash
# round 4: why does the 150 s window of bench.py (games started 0-360 moves into their games) show a mean batch of 249 and the 27-minute
# run from the empty board 254?  Same pipe, same replacement of finished games, with and without the staggered start.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/spst
python bench.py --steps 30 --warmup 5 --selfplay-seconds 0 --no-cpu-baseline --no-config5 --no-pump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench', d['value'], d['ms_per_step'])"
run() {
  local name=$1; shift
  ( env SAYURI_PIPE_TRACE=1 timeout 400 python tools/selfplay_bench.py --seconds ${SECONDS_:-100} --games 512 --num-games 100000 "$@" ) 2> gpurun_out/spst/$name.err | tail -1 > gpurun_out/spst/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.load(open('gpurun_out/spst/%s.json'%n))
sh=d.get('second_half',{})
print("== %-10s evals/s %8.1f (2nd half %8.1f) mean_batch %.1f partial %d/%d cores %.1f sys %.1f games_done %d pump %s"%(n,d['nn_evals_per_sec'],sh.get('nn_evals_per_sec',0),d['mean_batch'],d['partial_batches'],d['batches'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['games_done'],d['pump_us_per_batch']))
PY
  grep -h "closed:" gpurun_out/spst/$name.err | cut -c1-200
}
run empty
run stag360 --stagger 360
run empty2
run stag360b --stagger 360
