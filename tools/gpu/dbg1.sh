#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for hf in 1 0; do for sf in 1 0; do
echo "HEADS_FUSED=$hf SE_FUSED=$sf"
SAYURI_HEADS_FUSED=$hf SAYURI_SE_FUSED=$sf timeout 300 python -m pytest tests/test_gpu_net.py -x -q -k batch256_properties 2>&1 | tail -3
done; done
