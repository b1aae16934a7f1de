#!/bin/bash
# configs[4] (40b x 384, mixed boards): two channel tiles of 192 (default) against three of 128 (SAYURI_BOARD_KOT=128, experiments build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip.so /tmp/product.so
cp sayuri_amd/lib/libsayuri_hip_exp.so sayuri_amd/lib/libsayuri_hip.so
for v in 0 128 0 128; do
  if [ $v = 0 ]; then unset SAYURI_BOARD_KOT; else export SAYURI_BOARD_KOT=$v; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --no-pump --config5 > gpurun_out/c5_$v.json 2> gpurun_out/c5_$v.err
  python -c "import json;d=json.load(open('gpurun_out/c5_$v.json'))['config5'];print('kot=$v', {k:d[k] for k in ('evals_per_sec','ms_per_step','whole_net_mfma_frac','tower_conv_avg_launch_us','tower_conv_mfma_frac')})" || tail -3 gpurun_out/c5_$v.err
done
unset SAYURI_BOARD_KOT
cp /tmp/product.so sayuri_amd/lib/libsayuri_hip.so
