#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k "not fp16_error" 2>&1 | tail -8
bash tools/gpu/heads_tl.sh
