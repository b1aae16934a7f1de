#!/bin/bash
# persistent tower launch: net-level parity, then A/B against one launch per layer on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -15
for v in 0 1 0 1; do
  SAYURI_TOWER=$v timeout 300 python bench.py --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/tab_$v.json 2> gpurun_out/tab_$v.err
  python -c "import json;d=json.load(open('gpurun_out/tab_$v.json'));r=d['roofline'];print('tower=$v evals/s', d['value'], 'ms/step', d['ms_per_step'], 'whole-net frac', d['config']['whole_net_mfma_frac'], '| dominant', r['kernel'][:24], 'us', r['avg_launch_us'], 'frac', r['frac'])" || tail -5 gpurun_out/tab_$v.err
done
