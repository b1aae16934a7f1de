#!/bin/bash
# round 4: the epilogue of the persistent launch as generated assembly (tower_seam.py epi_hook) -- parity first, then old/new on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip_new.so sayuri_amd/lib/libsayuri_hip.so
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -x -q --timeout 300 -k "bit_identical or tower or persistent or launch" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_smallops.py -m gpu -x -q --timeout 150 -k "se_unit or conv_with_se" 2>&1 | tail -4
SAYURI_TOWER_GEN_EPI=0 timeout 600 python -m pytest tests/test_gpu_smallops.py tests/test_gpu_net.py -m gpu -x -q --timeout 300 -k "se_unit or conv_with_se or bit_identical or golden" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py -m gpu -x -q --timeout 300 2>&1 | tail -5
for v in old new old new old new; do
cp sayuri_amd/lib/libsayuri_hip_$v.so sayuri_amd/lib/libsayuri_hip.so
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$v', 'evals/s', d['value'], 'ms/step', d['ms_per_step'], 'whole-net', d['config']['whole_net_mfma_frac'], 'dominant us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
cp sayuri_amd/lib/libsayuri_hip_new.so sayuri_amd/lib/libsayuri_hip.so
