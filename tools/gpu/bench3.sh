#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/b3_$i.json 2> gpurun_out/b3_$i.err
python -c "import json;d=json.load(open('gpurun_out/b3_$i.json'));print('evals/s', d['value'], 'ms', d['ms_per_step'], 'tower us', d['roofline']['avg_launch_us'], 'n', d['roofline']['launches_timed'], 'frac', d['roofline']['frac'], 'whole', d['config']['whole_net_mfma_frac'], 'pump', d['config'].get('pump',{}).get('nn_evals_per_sec'))"
done
grep -A8 "kernel class" gpurun_out/b3_2.err
