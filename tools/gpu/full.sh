#!/bin/bash
# whole GPU suite + smoke + the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/full_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1200 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"
cat gpurun_out/full_bench.json; tail -5 gpurun_out/full_bench.err
