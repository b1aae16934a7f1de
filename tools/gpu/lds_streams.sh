#!/bin/bash
# round 4: the K loop's LDS bank conflicts split by fragment stream.  Three libraries: the shipped one and two
# measuring builds that leave out the A-fragment / the B-fragment ds_read_b128 stream of conv_board's K loop
# (-DSAYURI_DROP_STREAM=1 / 2, conv_board.h); counters on the per-layer launches (SAYURI_TOWER=0: the same K loop).
#   cd sayuri_amd/lib/obj && for n in 1 2; do hipcc -DSAYURI_DROP_STREAM=$n --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -c ../../csrc/hip/engine.hip \
#       -o /tmp/e$n.o && hipcc --offload-arch=gfx950 -shared /tmp/e$n.o tower_blob.o -o ../libsayuri_hip_drop$n.so; done
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/lds_streams
mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump --steps 2 --warmup 1"
for v in full drop1 drop2; do
  lib=""; [ $v != full ] && lib="$GRAFT_REPO_ROOT/sayuri_amd/lib/libsayuri_hip_$v.so"
  i=0
  for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && SAYURI_FAKE_HIP_LIB=$lib SAYURI_TOWER=0 timeout 120 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/$O/$v$i -o p --output-format csv -- $B > $GRAFT_REPO_ROOT/$O/$v$i.out 2> $GRAFT_REPO_ROOT/$O/$v$i.err)
    echo "## $v set $i rc=$?"
    python tools/pmc_summary.py $O/$v$i "conv_board_kernel<4" 2>&1 | tail -n +2
  done
  rm -rf $O/$v*/p_kernel_trace.csv $O/$v*/p_agent_info.csv
done
