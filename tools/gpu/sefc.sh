#!/bin/bash
# round 4: se_fc_kernel with 1024 threads: parity of the separate SE kernels and of configs[4], then the 40b x 384 forward
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_smallops.py tests/test_gpu_net.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --no-pump --config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config5']; print({k:c[k] for k in ('evals_per_sec','ms_per_step','whole_net_mfma_frac','tower_conv_avg_launch_us','tower_conv_mfma_frac')})"
done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sefc_kt -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --selfplay-seconds 0 --no-pump --steps 2 --warmup 1 --config5 > /dev/null 2>&1)
find gpurun_out/sefc_kt -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | grep -i "se_\|Name"
rm -rf gpurun_out/sefc_kt
