#!/bin/bash
# what does the activation cost the board convolution?  Experiments build (libsayuri_hip_exp.so, -DSAYURI_EXPERIMENTS), one launch
# per layer (--profile times them), Mish (as the network asks) against ReLU and identity forced on every board convolution
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip.so /tmp/product.so
cp sayuri_amd/lib/libsayuri_hip_exp.so sayuri_amd/lib/libsayuri_hip.so
for v in none 1 0 none 1 0; do
  if [ $v = none ]; then unset SAYURI_ACT_OVERRIDE; else export SAYURI_ACT_OVERRIDE=$v; fi
  SAYURI_TOWER=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump --profile > gpurun_out/act_$v.json 2> gpurun_out/act_$v.err
  echo "act=$v"; grep "conv3x3_tower" gpurun_out/act_$v.err | tr '\n' ';'; echo
done
unset SAYURI_ACT_OVERRIDE
cp /tmp/product.so sayuri_amd/lib/libsayuri_hip.so
