#!/bin/bash
# configs[4] (40b x 384, mixed boards): channel tile (three of 128 / two of 192, SAYURI_BOARD_KOT: experiments build) x number of
# chains.  With one chain the tile that needs fewer ROUNDS of workgroups wins (128: 450 workgroups = 2 rounds of 128 channels, 192: 300 =
# 2 rounds of 192); chains fill the rounds, and what counts then is the CU time per board -- does the 192-channel tile's better ratio of
# MFMAs per fragment read (2.4 against 1.7) show?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip.so /tmp/product.so
cp sayuri_amd/lib/libsayuri_hip_exp.so sayuri_amd/lib/libsayuri_hip.so
for rep in 1 2; do
for kot in 128 192; do
  for ch in 0 2 3 4; do
    export SAYURI_BOARD_KOT=$kot
    if [ $ch = 0 ]; then unset SAYURI_CHAINS; else export SAYURI_CHAINS=$ch; fi
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --no-pump --config5 > gpurun_out/c5kc_${kot}_$ch.json 2> gpurun_out/c5kc_${kot}_$ch.err
    python -c "import json;d=json.load(open('gpurun_out/c5kc_${kot}_$ch.json'))['config5'];print('kot=$kot chains=$ch', {k:d.get(k) for k in ('chains','evals_per_sec','ms_per_step','whole_net_mfma_frac','evals_per_sec_one_chain','tower_conv_avg_launch_us','tower_conv_mfma_frac')})" || tail -3 gpurun_out/c5kc_${kot}_$ch.err
  done
done
done
unset SAYURI_BOARD_KOT SAYURI_CHAINS
cp /tmp/product.so sayuri_amd/lib/libsayuri_hip.so
