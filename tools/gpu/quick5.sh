#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_layers.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k "not fp16_error" 2>&1 | tail -3
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_net.py -x -q -k "batch256_properties or se_unit_fused or config5 or mixed" 2>&1 | tail -1; done
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --no-pump --profile > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err
python -c "import json;d=json.load(open('gpurun_out/q_bench.json'));print('evals/s', d['value'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'whole', d['config']['whole_net_mfma_frac'])"
done
grep -A8 "kernel class" gpurun_out/q_bench.err
