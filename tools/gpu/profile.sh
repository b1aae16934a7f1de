#!/bin/bash
# Evidence pass of a round on one box (TAG=r03 ...): GPU suite, kernel-trace stats, PMC passes (one counter set per run,
# --kernel-trace only), HBM-side traffic with calibration streams, the default bench line, configs[4], fp32.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-r05}
O=gpurun_out/$TAG
mkdir -p $O/prof
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump"
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.log
fi
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof/kt -o p --output-format csv -- $B --steps 100 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof/kt.out 2> $GRAFT_REPO_ROOT/$O/prof/kt.err); echo "kernel-trace rc=$?"
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump --steps 100 --warmup 10   ($(date -u +%FT%TZ))"
  echo "# bench line of the same command:"; tail -1 $O/prof/kt.out | sed 's/^/# /' | cut -c1-1500
  find $O/prof/kt -name "*kernel_stats.csv" | head -1 | xargs cat; } > $O/rocprofv3_kernel_stats.txt
head -8 $O/rocprofv3_kernel_stats.txt | cut -c1-220
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_WRITE_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  # counters are collected on the persistent tower launch itself (round 4: its descriptor declares no private segment any more,
  # which is what made rocprofv3's counter collection fault in rounds 2-3); the first set also on the per-layer launches
  # (SAYURI_TOWER=0: the compiled kernels) for the per-layer rows
  (cd /tmp && timeout 100 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof/p$i -o p --output-format csv -- $B --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof/p$i.out 2> $GRAFT_REPO_ROOT/$O/prof/p$i.err)
  echo "## set $i [$set] rc=$?"
  grep -E "Memory access fault|Segmentation|rror" $O/prof/p$i.err | head -2
  python tools/pmc_summary.py $O/prof/p$i "conv_tower_kernel<4" 2>&1 | tail -n +2
  if [ $i -le 2 ]; then
    (cd /tmp && SAYURI_TOWER=0 timeout 100 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof/l$i -o p --output-format csv -- $B --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof/l$i.out 2> $GRAFT_REPO_ROOT/$O/prof/l$i.err)
    python tools/pmc_summary.py $O/prof/l$i "conv_board_kernel<4" 2>&1 | tail -n +2 | sed "s/^/   per-layer plain  /"
    python tools/pmc_summary.py $O/prof/l$i "conv_board_se_kernel<4" 2>&1 | tail -n +2 | sed "s/^/   per-layer SE     /"
  fi
  if [ $i -ge 3 ]; then
    (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof/c$i -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/ubench/hbm_calib.so > $GRAFT_REPO_ROOT/$O/prof/c$i.out 2> $GRAFT_REPO_ROOT/$O/prof/c$i.err)
    for k in calib_read calib_write_kernel calib_write64; do python tools/pmc_summary.py $O/prof/c$i $k 2>&1 | tail -n +2 | sed "s/^/   $k (1 GiB)  /"; done
  fi
done > $O/pmc_raw.txt 2>&1
cat $O/pmc_raw.txt
timeout 900 python bench.py --no-cpu-baseline --selfplay-seconds 0 --no-pump --config5 > $O/config5.json 2> $O/config5.err; echo "config5 rc=$?"
python -c "import json;d=json.load(open('$O/config5.json'));print(d['value'], d['roofline']['frac'], d['config5'])"
timeout 300 python bench.py --fp32 --steps 5 --warmup 1 --no-cpu-baseline --no-config5 --selfplay-seconds 0 > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "fp32 rc=$?"
python -c "import json;d=json.load(open('$O/bench_fp32.json'));print(d['value'], d['roofline'])"
rm -rf $O/prof/*/p_kernel_trace.csv $O/prof/*/p_agent_info.csv
# the default line, as the driver runs it (cpu_baseline and the self-play window included)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
tail -1 $O/bench_default.json | cut -c1-600
