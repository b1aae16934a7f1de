#!/bin/bash
# configs[2] (512 games, batch 256) under one switch at a time on ONE box: which host change of round 3 costs the self-play rate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/spab
SECS=${SECS:-60}
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc $(nproc)"
run() {  # name, dir, env...
  local name=$1 dir=$2; shift 2
  ( cd "$dir" && env "$@" SAYURI_PIPE_TRACE=1 SAYURI_HIP_FWDSTAT=1 timeout 300 python tools/selfplay_bench.py --seconds $SECS --games 512 --stagger 360 ${EXTRA} ) 2> gpurun_out/spab/$name.err | tail -1 > gpurun_out/spab/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/spab/%s.json'%n))
    sh=d.get('second_half',{})
    print("== %-18s evals/s %8.1f (2nd half %8.1f) mean_batch %.1f partial %d/%d cores %.1f sys %.1f ctx/s %d pump %s"%(n,d['nn_evals_per_sec'],sh.get('nn_evals_per_sec',0),d['mean_batch'],d['partial_batches'],d['batches'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['ctx_switches_per_sec'],d['pump_us_per_batch']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  grep -h "fwdstat\|closed:" gpurun_out/spab/$name.err
}
# the microbench of this box first (resident inputs)
python bench.py --steps 20 --warmup 5 --selfplay-seconds 0 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench', d['value'], d['ms_per_step'], d['config'].get('pump_packed',{}).get('nn_evals_per_sec'))" 
for round in 1 2; do
run head_$round . A=1
run r02_$round ab/r02 A=1
run r02x_$round ab/r02x A=1
done
run norotnotify . SAYURI_AB_ROTATE_NOTIFY=0
run arenakeep . SAYURI_AB_ARENA_KEEP_MB=64
run nopin . SAYURI_NO_PIN=1
run tail2 . SAYURI_PIPE_TAIL=2.0
run tower0 . SAYURI_TOWER=0
