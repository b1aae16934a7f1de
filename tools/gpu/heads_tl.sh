#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
SAYURI_HEADS_DBG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/htl.json 2> gpurun_out/htl.err
grep "heads timeline" gpurun_out/htl.err
grep -A9 "kernel class" gpurun_out/htl.err
