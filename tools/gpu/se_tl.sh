#!/bin/bash
# in-kernel timeline of an SE-carrying convolution and of a plain one (experiments build, libsayuri_hip_exp.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip.so /tmp/product.so
cp sayuri_amd/lib/libsayuri_hip_exp.so sayuri_amd/lib/libsayuri_hip.so
for dbg in -3 5; do
SAYURI_BOARD_DBG=$dbg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump --profile > gpurun_out/se_tl.json 2> gpurun_out/se_tl.err
grep "timeline wg[01]" gpurun_out/se_tl.err | head -16
done
cp /tmp/product.so sayuri_amd/lib/libsayuri_hip.so
