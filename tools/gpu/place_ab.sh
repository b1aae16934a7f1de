#!/bin/bash
# placement of the tower's convolution body (tower_seam.py --align / --pad; SAYURI_TOWER_PAD=N builds as libsayuri_hip_padN.so):
# the product build (pad 32) against the other multiples of 8, three rounds on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for round in 1 2 3; do
for v in product pad0 pad8 pad16 pad24 pad40 pad48 pad56; do
  lib=""; [ $v != product ] && lib="$GRAFT_REPO_ROOT/sayuri_amd/lib/libsayuri_hip_$v.so"
  SAYURI_FAKE_HIP_LIB=$lib timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/pl.json 2> gpurun_out/pl.err
  python -c "import json;d=json.load(open('gpurun_out/pl.json'));print('$v', 'evals/s', d['value'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
done
