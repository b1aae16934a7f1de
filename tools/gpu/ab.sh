#!/bin/bash
# A/B of two builds of libsayuri_hip.so on one box (boxes differ by several per cent): old new old new ...
# (build the two variants to sayuri_amd/lib/libsayuri_hip_{old,new}.so first; TESTS="..." runs pytest on the new one first)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip_new.so sayuri_amd/lib/libsayuri_hip.so
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q 2>&1 | tail -8; fi
for v in old new old new old new; do
cp sayuri_amd/lib/libsayuri_hip_$v.so sayuri_amd/lib/libsayuri_hip.so
timeout 300 python bench.py --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump $( [ -n "$NOPROF" ] || echo --profile ) > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$v', 'evals/s', d['value'], 'ms/step', d['ms_per_step'], 'whole-net', d['config']['whole_net_mfma_frac'], 'dominant us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
grep "conv3x3_tower_se\|pack_input\|conv3x3_input\|heads_fused" gpurun_out/ab.err | tr '\n' ';'; echo
done
cp sayuri_amd/lib/libsayuri_hip_new.so sayuri_amd/lib/libsayuri_hip.so
