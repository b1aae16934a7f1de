#!/bin/bash
# weight hand-over between the layers of the persistent run: parity first, then A/B on one box (SAYURI_TOWER_CHAIN=0 | 1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_smallops.py -x -q 2>&1 | tail -8
for v in 0 1 0 1 0 1; do
  SAYURI_TOWER_CHAIN=$v timeout 300 python bench.py --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/chain_$v.json 2> gpurun_out/chain_$v.err
  python -c "import json;d=json.load(open('gpurun_out/chain_$v.json'));r=d['roofline'];print('chain=$v evals/s', d['value'], 'ms/step', d['ms_per_step'], 'whole-net frac', d['config']['whole_net_mfma_frac'], '| us', r['avg_launch_us'], 'frac', r['frac'])" || tail -5 gpurun_out/chain_$v.err
done
