#!/bin/bash
# Round 6: what would hiding the epilogues be worth?  The persistent launch with the generated epilogue of its 35 plain layers
# SKIPPED from the 12th launch on (SAYURI_TOWER_NOEPI_AFTER: nothing stored, the activations stay what the last complete forward
# left -- realistic operands, not zeros) against the product, 600 steps each (2 s: under the power cap), interleaved on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in product noepi; do
    if [ $v = noepi ]; then export SAYURI_TOWER_NOEPI_AFTER=11; else unset SAYURI_TOWER_NOEPI_AFTER; fi
    timeout 300 python bench.py --steps 600 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/noepi.json 2> gpurun_out/noepi.err
    python -c "import json;d=json.load(open('gpurun_out/noepi.json'));print('$v', 'evals/s', d['value'], 'ms/step', d['ms_per_step'], 'tower launch us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
  done
done
