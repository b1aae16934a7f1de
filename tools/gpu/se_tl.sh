#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
SAYURI_BOARD_DBG=-3 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/se_tl.json 2> gpurun_out/se_tl.err
grep "timeline wg[12]" gpurun_out/se_tl.err
grep -A10 "kernel class" gpurun_out/se_tl.err
