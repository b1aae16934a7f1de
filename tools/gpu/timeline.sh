#!/bin/bash
# in-kernel timeline of the board convolution + PMC passes of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
SAYURI_BOARD_DBG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/tl.json 2> gpurun_out/tl.err
grep "board timeline" gpurun_out/tl.err | head -40
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 120 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.out 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.err)
  echo "pmc $tag rc=$?"
  python tools/pmc_summary.py gpurun_out/pmc_$tag conv_board 2>&1 | tail -12
done
