#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_layers.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k "not fp16_error" 2>&1 | tail -3
for cfg in "SAYURI_NO_ARITH=1" "SAYURI_X=1" "SAYURI_NO_ARITH=1" "SAYURI_X=1" "SAYURI_NO_ARITH=1" "SAYURI_X=1"; do
env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --selfplay-seconds 0 --no-pump > gpurun_out/ar.json 2> gpurun_out/ar.err
python -c "import json;d=json.load(open('gpurun_out/ar.json'));print('$cfg', 'evals/s', d['value'], 'ms', d['ms_per_step'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
