#!/bin/bash
# configs[2] for a short while with the pipe's arrival / batch-size trace (SAYURI_PIPE_TRACE): who is late for a batch, and why
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sp
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc $(nproc)"; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
taskset -p $$ | head -1
for v in pin nopin threads32 nopin_threads128; do
unset SAYURI_NO_PIN; extra=""
case $v in nopin) export SAYURI_NO_PIN=1;; threads32) extra="--game-threads 32";; nopin_threads128) export SAYURI_NO_PIN=1; extra="--game-threads 128";; esac
SAYURI_PIPE_TRACE=1 timeout 300 python tools/selfplay_bench.py --seconds ${SECS:-100} --games 512 --stagger 360 $extra 2> gpurun_out/sp/trace.err | tail -1 > gpurun_out/sp/trace.json
echo "== $v"
python -c "
import json
d=json.load(open('gpurun_out/sp/trace.json'))
print({k:d[k] for k in ('nn_evals_per_sec','mean_batch','partial_batches','batches','host_cpu_cores_busy','pump_us_per_batch')})"
grep "pipe trace" gpurun_out/sp/trace.err | grep -v "next request\|sizes"
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
done
