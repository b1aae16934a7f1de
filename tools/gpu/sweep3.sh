#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for g in 2048 4096; do
  MALLOC_ARENA_MAX=1 timeout 400 python tools/selfplay_bench.py --seconds 60 --games $g --num-games 1000000 > gpurun_out/x_g$g.json 2> gpurun_out/x_g$g.err
  echo "arena_max=1 games=$g rc=$?: $(python -c "import json;d=json.load(open('gpurun_out/x_g$g.json'));print({k:d[k] for k in d if k in ('nn_evals_per_sec','second_half','mean_batch','host_cpu_cores_busy','host_sys_cores')})" 2>&1 | tail -1)"
done
which strace perf 2>&1 | head -2
