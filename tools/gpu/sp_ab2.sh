#!/bin/bash
# round 4: the self-play queue with nothing small copied (SAYURI_IO_V2, engine.hip submit()) against the old copies, one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/spab2
SECS=${SECS:-60}
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_dropin.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -3
run() {
  local name=$1; shift
  ( env "$@" SAYURI_PIPE_TRACE=1 SAYURI_HIP_FWDSTAT=1 timeout 300 python tools/selfplay_bench.py --seconds $SECS --games ${GAMES:-512} --stagger 360 ) 2> gpurun_out/spab2/$name.err | tail -1 > gpurun_out/spab2/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/spab2/%s.json'%n))
    sh=d.get('second_half',{})
    print("== %-14s evals/s %8.1f (2nd half %8.1f) mean_batch %.1f partial %d/%d cores %.1f sys %.1f pump %s"%(n,d['nn_evals_per_sec'],sh.get('nn_evals_per_sec',0),d['mean_batch'],d['partial_batches'],d['batches'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['pump_us_per_batch']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  grep -h "fwdstat\|closed:" gpurun_out/spab2/$name.err
}
python bench.py --steps 20 --warmup 5 --selfplay-seconds 0 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench', d['value'], d['ms_per_step'], 'pump', d['config'].get('pump',{}).get('nn_evals_per_sec'), 'pump_packed', d['config'].get('pump_packed',{}).get('nn_evals_per_sec'))"
SAYURI_IO_V2=0 python bench.py --steps 20 --warmup 5 --selfplay-seconds 0 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench io_v2=0', d['value'], d['ms_per_step'], 'pump', d['config'].get('pump',{}).get('nn_evals_per_sec'), 'pump_packed', d['config'].get('pump_packed',{}).get('nn_evals_per_sec'))"
run v2_1 A=1
run old_1 SAYURI_IO_V2=0
run v2_2 A=1
run old_2 SAYURI_IO_V2=0
run v2_tail2 SAYURI_PIPE_TAIL=2.0
run v2_tower0 SAYURI_TOWER=0
