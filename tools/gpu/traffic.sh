#!/bin/bash
# HBM traffic counters of the tower convolution: try each candidate set in its own bounded pass
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
(cd /tmp && timeout 60 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/traffic/counters_list.txt 2>&1)
grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z0-9_]*sum\|FETCH_SIZE\|WRITE_SIZE\|TCC_BUBBLE[A-Z_]*" gpurun_out/traffic/counters_list.txt | sort -u | tr '\n' ' ' | head -c 3000; echo
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_READ_sum TCC_WRITE_sum" "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/traffic/p$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/traffic/p$i.out 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/p$i.err)
  echo "== set $i [$set] rc=$?"
  grep -E "Memory access fault|Segmentation|rror" gpurun_out/traffic/p$i.err | head -2
  python tools/pmc_summary.py gpurun_out/traffic/p$i "conv_board_kernel<4" 2>&1 | tail -6
  (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/traffic/c$i -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench/hbm_calib.so > $GRAFT_REPO_ROOT/gpurun_out/traffic/c$i.out 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/c$i.err)
  python tools/pmc_summary.py gpurun_out/traffic/c$i "calib_read" 2>&1 | tail -4
  python tools/pmc_summary.py gpurun_out/traffic/c$i "calib_write" 2>&1 | tail -4
done
