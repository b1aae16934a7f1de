#!/bin/bash
# round 4: counters of the 128-channel board kernel on configs[4] (40b x 384, mixed boards)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5pmc
mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --selfplay-seconds 0 --no-pump --steps 2 --warmup 1 --config5"
for v in full nodeep; do
  lib=""; [ $v != full ] && lib="$GRAFT_REPO_ROOT/sayuri_amd/lib/libsayuri_hip_$v.so"
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && SAYURI_FAKE_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/$O/$v$i -o p --output-format csv -- $B > $GRAFT_REPO_ROOT/$O/$v$i.out 2> $GRAFT_REPO_ROOT/$O/$v$i.err)
    echo "## $v set $i rc=$?"
    python tools/pmc_summary.py $O/$v$i "conv_board_kernel<2" 2>&1 | tail -n +2
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p --output-format csv -- $B > $GRAFT_REPO_ROOT/$O/kt.out 2> $GRAFT_REPO_ROOT/$O/kt.err)
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -30
rm -rf $O/*/p_kernel_trace.csv $O/*/p_agent_info.csv
