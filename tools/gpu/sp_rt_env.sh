#!/bin/bash
# round 4: one runtime thread burns a whole core in ioctl during self-play (tools/gpu/sp_hostprof.sh).  Which thread is it, and which
# runtime setting quiets it?  25 s of 512-game self-play per setting, with the per-thread PC sampler.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/rtenv
gcc -O2 -fPIC -shared tools/prof/pcsample.c -o /tmp/pcsample.so -ldl -lrt -lpthread || exit 1
run() {
  local name=$1; shift
  rm -f gpurun_out/rtenv/$name.prof.*
  ( env "$@" LD_PRELOAD=/tmp/pcsample.so PCSAMPLE_OUT=$GRAFT_REPO_ROOT/gpurun_out/rtenv/$name.prof timeout 200 python tools/selfplay_bench.py --seconds ${SECONDS_:-25} --games 512 --stagger 360 ) 2> gpurun_out/rtenv/$name.err | tail -1 > gpurun_out/rtenv/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/rtenv/%s.json'%n))
    print("== %-14s evals/s %8.1f mean_batch %.1f cores %.2f sys %.2f ctx/s %d"%(n,d['nn_evals_per_sec'],d['mean_batch'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['ctx_switches_per_sec']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  for f in gpurun_out/rtenv/$name.prof.*; do head -1 $f; sed -n '/# threads/,$p' $f | grep -v sayuri-games | sort -k3 -n -r | head -4 | cut -c1-200; done
}
run base
run nodirect AMD_DIRECT_DISPATCH=0
run activewait0 ROC_ACTIVE_WAIT_TIMEOUT=0
run nointerrupt HSA_ENABLE_INTERRUPT=0
run base2
