#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null; free -g | head -2; cat /sys/kernel/mm/transparent_hugepage/enabled; nproc; ulimit -a | grep -E "processes|memory|stack"
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for g in 1024 2048; do
  timeout 400 python tools/selfplay_bench.py --seconds 30 --games $g --num-games 1000000 --game-threads 64 > gpurun_out/m_g$g.json 2> gpurun_out/m_g$g.err
  echo "games=$g: $(python -c "import json;d=json.load(open('gpurun_out/m_g$g.json'));print({k:d[k] for k in d if k in ('nn_evals_per_sec','max_rss_gb','host_cpu_cores_busy','host_sys_cores','ctx_switches_per_sec')})" 2>&1 | tail -1)"
  grep -E "ctxt|procs" /proc/stat | head -3
done
