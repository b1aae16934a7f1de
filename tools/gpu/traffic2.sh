#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
i=0
for set in "TCC_READ_sum TCC_WRITE_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  (cd /tmp && SAYURI_CONV=glds timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/traffic/g$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/traffic/g$i.out 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/g$i.err)
  echo "== glds set $i [$set] rc=$?"
  python tools/pmc_summary.py gpurun_out/traffic/g$i "conv_glds_kernel<8, 3" 2>&1 | tail -4
done
