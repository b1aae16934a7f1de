#!/bin/bash
# round 4: does configs[4]'s 128-wide channel tile wait for the LDS?  The measuring builds of tools/gpu/lds_streams.sh on the 40b x 384 forward.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5drop
for v in full drop1 drop2 full; do
  lib=""; [ $v != full ] && lib="$GRAFT_REPO_ROOT/sayuri_amd/lib/libsayuri_hip_$v.so"
  SAYURI_FAKE_HIP_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --no-pump --config5 > gpurun_out/c5drop/$v.json 2> gpurun_out/c5drop/$v.err
  python -c "import json;d=json.load(open('gpurun_out/c5drop/$v.json'));c=d['config5'];print('$v', d['value'], {k:c[k] for k in ('evals_per_sec','ms_per_step','whole_net_mfma_frac','tower_conv_avg_launch_us','tower_conv_mfma_frac')})" || tail -3 gpurun_out/c5drop/$v.err
done
