#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for p in 0 1 2 3 0; do
  SAYURI_BOARD_PRIO=$p SAYURI_BOARD_DBG=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/prio$p.json 2> gpurun_out/prio$p.err
  echo "PRIO=$p: $(python -c "import json;d=json.load(open('gpurun_out/prio$p.json'));print(d['value'], d['roofline']['avg_launch_us'])")"
  grep "board timeline wg1 wave[04]" gpurun_out/prio$p.err
done
# epilogue at half the chip: batch 128
SAYURI_BOARD_DBG=1 timeout 300 python bench.py --batch 128 --steps 10 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/b128.json 2> gpurun_out/b128.err
echo "batch128: $(python -c "import json;d=json.load(open('gpurun_out/b128.json'));print(d['value'], d['roofline']['avg_launch_us'])")"
grep "board timeline wg1 wave[04]" gpurun_out/b128.err
