#!/bin/bash
# Round-3 evidence pass on one box:
#   1. the MFMA issue-rate micro-benchmark (random / zero operands) with package power and clocks sampled beside it
#      -> gpurun_out/r03_mfma_rate_ubench.txt   (the "power-limited ceiling" the design argues from)
#   2. the in-kernel timeline of the SE-carrying convolution
#   3. HBM-side traffic of the tower convolution: read + write counter candidates, each set in its own pass, with the
#      calibration streams (incl. the board kernel's 64-byte store pattern) under the same set
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
U=gpurun_out/r03_mfma_rate_ubench.txt
{
  echo "# tools/ubench/mfma_rate.so on $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' | sed 's/.*: *//') -- $(date -u +%FT%TZ)"
  echo "# idle:"; rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | sed 's/^/#   /'
  echo "## burst mode (4000 iterations per line, 256 workgroups x 512 threads, 48 independent accumulators per wave)"
  timeout 120 tools/ubench/mfma_rate.so
  for mode in random zero; do
    echo "## sustained 12 s, $mode operands, rocm-smi sampled every 2 s beside it"
    if [ $mode = zero ]; then (timeout 60 tools/ubench/mfma_rate.so 12 zero > gpurun_out/mfma_sustain_$mode.txt 2>&1 &); else (timeout 60 tools/ubench/mfma_rate.so 12 > gpurun_out/mfma_sustain_$mode.txt 2>&1 &); fi
    sleep 3
    for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | sed 's/^.*GPU\[0\][^:]*: *//' | tr '\n' ';'; echo; sleep 2; done
    sleep 4
    cat gpurun_out/mfma_sustain_$mode.txt
  done
} > $U 2>&1
cat $U

echo "=== SE timeline"
SAYURI_BOARD_DBG=-3 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --profile > gpurun_out/se_tl.json 2> gpurun_out/se_tl.err
grep "timeline wg[12]" gpurun_out/se_tl.err | head -20
grep -A12 "kernel class" gpurun_out/se_tl.err | head -30
python -c "import json;d=json.load(open('gpurun_out/se_tl.json'));print('evals/s', d['value'], d['roofline'])"

echo "=== traffic"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITE_sum TCC_WRITE_SECTORS_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_sum" "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_WRITE_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/traffic/p$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > $GRAFT_REPO_ROOT/gpurun_out/traffic/p$i.out 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/p$i.err)
  echo "## set $i [$set] rc=$?"
  grep -E "Memory access fault|Segmentation|rror" gpurun_out/traffic/p$i.err | head -2
  python tools/pmc_summary.py gpurun_out/traffic/p$i "conv_board_kernel<4" 2>&1 | tail -n +2
  (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/traffic/c$i -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench/hbm_calib.so > $GRAFT_REPO_ROOT/gpurun_out/traffic/c$i.out 2> $GRAFT_REPO_ROOT/gpurun_out/traffic/c$i.err)
  for k in calib_read calib_write_kernel calib_write64; do python tools/pmc_summary.py gpurun_out/traffic/c$i $k 2>&1 | tail -n +2 | sed "s/^/   $k  /"; done
done > gpurun_out/r03_traffic_raw.txt 2>&1
cat gpurun_out/r03_traffic_raw.txt
