#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for act in 5 1 0; do
for c in 1 40; do
SAYURI_ACT_OVERRIDE=$act SAYURI_BOARD_DBG=$c timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/e.json 2> gpurun_out/e.err
echo "act=$act tower conv #$c: $(python -c "import json;d=json.load(open('gpurun_out/e.json'));print(d['roofline']['avg_launch_us'])") us"; grep "board timeline wg[1] wave[04]" gpurun_out/e.err
done
done
