#!/bin/bash
# round 4: where do the host cores go during self-play on the GPU box?  (PC sampling by tools/prof/pcsample.c, 1 ms of CPU per sample)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/hostprof
gcc -O2 -fPIC -shared tools/prof/pcsample.c -o /tmp/pcsample.so -ldl -lrt -lpthread || exit 1
( env LD_PRELOAD=/tmp/pcsample.so PCSAMPLE_OUT=$GRAFT_REPO_ROOT/gpurun_out/hostprof/prof.txt timeout 300 python tools/selfplay_bench.py --seconds ${SECONDS_:-60} --games 512 --stagger 360 "$@" ) 2> gpurun_out/hostprof/sp.err | tail -1 > gpurun_out/hostprof/sp.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/hostprof/sp.json'))
print("evals/s %.1f mean_batch %.1f cores %.1f sys %.1f ctx/s %d"%(d['nn_evals_per_sec'],d['mean_batch'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['ctx_switches_per_sec']))
PY
for f in gpurun_out/hostprof/prof.txt*; do echo "== $f"; head -60 $f | cut -c1-200; done
nproc; cat /proc/cpuinfo | grep "model name" | head -1; cat /sys/fs/cgroup/cpu.max 2>/dev/null
