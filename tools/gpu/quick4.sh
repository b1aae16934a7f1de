#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_layers.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k "not fp16_error" 2>&1 | tail -3
SAYURI_BOARD_DBG=5 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 --no-pump --profile > gpurun_out/q_dbg.json 2> gpurun_out/q_dbg.err
grep "board timeline wg[12] wave[04]" gpurun_out/q_dbg.err
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --no-pump --profile > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err
python -c "import json;d=json.load(open('gpurun_out/q_bench.json'));print('evals/s', d['value'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'whole', d['config']['whole_net_mfma_frac'])"
done
grep -A8 "kernel class" gpurun_out/q_bench.err
if [ -f /tmp/prev_hip.so ]; then echo; fi
