#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_layers.py -x -q -k "board" 2>&1 | tail -2
for m in 1 0; do
SAYURI_SE_FUSED=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/q2_$m.json 2> gpurun_out/q2_$m.err
python -c "import json;d=json.load(open('gpurun_out/q2_$m.json'));print('SE_FUSED=$m evals/s', d['value'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'whole', d['config']['whole_net_mfma_frac'])"
grep -A13 "kernel class" gpurun_out/q2_$m.err
done
