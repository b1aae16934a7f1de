#!/bin/bash
# round 4: every event record / wait between the three streams is a marker for the runtime's signal thread (a full host core in
# self-play, tools/gpu/sp_rt_env.sh).  SAYURI_IO_INORDER=1 puts a ticket's upload, forward and download on one stream.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/inorder
gcc -O2 -fPIC -shared tools/prof/pcsample.c -o /tmp/pcsample.so -ldl -lrt -lpthread || exit 1
for v in 0 1; do
  SAYURI_IO_INORDER=$v python bench.py --steps 30 --warmup 5 --selfplay-seconds 0 --no-cpu-baseline --no-config5 --no-pump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench inorder=$v', d['value'], d['ms_per_step'])"
done
run() {
  local name=$1; shift
  rm -f gpurun_out/inorder/$name.prof.*
  ( env "$@" LD_PRELOAD=/tmp/pcsample.so PCSAMPLE_OUT=$GRAFT_REPO_ROOT/gpurun_out/inorder/$name.prof timeout 200 python tools/selfplay_bench.py --seconds ${SECONDS_:-40} --games 512 --stagger 360 ) 2> gpurun_out/inorder/$name.err | tail -1 > gpurun_out/inorder/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/inorder/%s.json'%n))
    print("== %-14s evals/s %8.1f mean_batch %.1f cores %.2f sys %.2f ctx/s %d"%(n,d['nn_evals_per_sec'],d['mean_batch'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['ctx_switches_per_sec']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  for f in gpurun_out/inorder/$name.prof.*; do head -1 $f; sed -n '/# threads/,$p' $f | grep -v sayuri-games | sort -k3 -n -r | head -3 | cut -c1-200; done
}
run base SAYURI_IO_INORDER=0
run inorder SAYURI_IO_INORDER=1
run base2 SAYURI_IO_INORDER=0
run inorder2 SAYURI_IO_INORDER=1
timeout 900 env SAYURI_IO_INORDER=1 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -3
