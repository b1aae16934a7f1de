#!/bin/bash
# placement of the plain tower body (tower_seam.py --align / --pad): product build against four pads, two rounds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp sayuri_amd/lib/libsayuri_hip.so /tmp/product.so
for round in 1 2 3; do
for v in product pad24 pad40 se32 se96; do
  if [ $v = product ]; then cp /tmp/product.so sayuri_amd/lib/libsayuri_hip.so; else cp sayuri_amd/lib/libsayuri_hip_$v.so sayuri_amd/lib/libsayuri_hip.so; fi
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/pl.json 2> gpurun_out/pl.err
  python -c "import json;d=json.load(open('gpurun_out/pl.json'));print('$v', 'evals/s', d['value'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
done
cp /tmp/product.so sayuri_amd/lib/libsayuri_hip.so
