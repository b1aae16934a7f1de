#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_smallops.py -x -q 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_net.py -x -q -k "40b384 or config5 or fp16_error" 2>&1 | tail -8
cat gpurun_out/fp16_error_20b256.json
