#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== mfma_rate sustained variants"
for m in random rot k32; do timeout 30 tools/ubench/mfma_rate.so 4 $m | tail -2; done
echo "== A/B pack"
bash tools/gpu/ab.sh
