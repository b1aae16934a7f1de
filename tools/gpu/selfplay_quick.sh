#!/bin/bash
# GPU search-parity tests, then the configs[2] self-play workload for 50 s at the given game counts (pump statistics)
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -2
for g in ${GAMES:-512 512}; do
  timeout 200 python tools/selfplay_bench.py --seconds 50 --games $g 2>/dev/null | tail -1 > gpurun_out/sp_$g.json
  python -c "
import json
d=json.load(open('gpurun_out/sp_$g.json'))
print(d['concurrent_games'], d['nn_evals_per_sec'], d.get('second_half',{}).get('nn_evals_per_sec'), d['mean_batch'], d['pump_us_per_batch']['gpu_queue_empty_us'], d['host_cpu_cores_busy'], d['max_rss_gb'])"
done
