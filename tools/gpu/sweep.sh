#!/bin/bash
# GPU test suite, then the concurrent-games sweep of the self-play loop (one OS thread per game)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/sweep_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/sweep_tests.log
for g in 1024 2048 4096; do
  timeout 400 python tools/selfplay_bench.py --seconds 60 --games $g --num-games 1000000 > gpurun_out/r02_selfplay_g$g.json 2> gpurun_out/r02_selfplay_g$g.err
  echo "games=$g rc=$?: $(python -c "import json;d=json.load(open('gpurun_out/r02_selfplay_g$g.json'));print({k:d[k] for k in d if k in ('nn_evals_per_sec','playouts_per_sec','moves_per_sec','mean_batch','host_cores_busy','games')})" 2>&1 | tail -1)"
done
