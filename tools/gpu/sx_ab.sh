#!/bin/bash
# Round 6: the SE unit of the 384-channel layers inside the convolution (conv_board_sx.h) against the three separate kernels
# (SAYURI_SE_SPLIT=0), configs[4]'s batch, interleaved on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
  for mode in 1 0; do
    SAYURI_SE_SPLIT=$mode timeout 300 python bench.py --no-cpu-baseline --selfplay-seconds 0 --no-pump --steps 30 --warmup 5 > gpurun_out/sx_ab_${mode}_$rep.json 2> gpurun_out/sx_ab_${mode}_$rep.err
    python - <<PY
import json
d=json.load(open("gpurun_out/sx_ab_${mode}_$rep.json"))
c=d["config5"]
print("SE_SPLIT=$mode rep $rep: microbench", d["value"], "| config5 evals/s", c["evals_per_sec"], "chains", c["chains"], "one chain", c["evals_per_sec_one_chain"], "whole-net frac", c["whole_net_mfma_frac"], "tower conv us", c["tower_conv_avg_launch_us"], "frac", c["tower_conv_mfma_frac"])
PY
  done
done
