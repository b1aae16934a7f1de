#!/bin/bash
# round-2 evidence: GPU suite, kernel-trace stats, PMC passes, config 5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_gpu_tests.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/kt.out 2> $GRAFT_REPO_ROOT/gpurun_out/prof/kt.err); echo "kernel-trace rc=$?"
ls gpurun_out/prof/kt | head; head -12 gpurun_out/prof/kt/p_kernel_stats.csv 2>/dev/null | cut -c1-220
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag.out 2> $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$tag.err)
  echo "pmc $tag rc=$?"
  python tools/pmc_summary.py gpurun_out/prof/pmc_$tag "conv_board_kernel<4" 2>&1 | tail -9
done
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --selfplay-seconds 0 --config5 > gpurun_out/r02_config5.json 2> gpurun_out/r02_config5.err; echo "config5 rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r02_config5.json'));print(d['config5'])"
timeout 300 python bench.py --fp32 --steps 5 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 > gpurun_out/r02_bench_fp32.json 2> gpurun_out/r02_bench_fp32.err; echo "fp32 rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r02_bench_fp32.json'));print(d['value'], d['roofline'])"
