#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k "packed" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -5
