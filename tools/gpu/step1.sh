#!/bin/bash
# first GPU pass of the board kernel: layer tests, net tests, A/B bench against the glds kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_layers.py -x -q -k "board" > gpurun_out/s1_layers.log 2>&1; echo "layers rc=$?"; tail -15 gpurun_out/s1_layers.log
timeout 600 python bench.py --steps 20 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/s1_bench_board.json 2> gpurun_out/s1_bench_board.err; echo "bench board rc=$?"; cat gpurun_out/s1_bench_board.json; tail -20 gpurun_out/s1_bench_board.err
SAYURI_CONV=glds timeout 600 python bench.py --steps 20 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/s1_bench_glds.json 2> gpurun_out/s1_bench_glds.err; echo "bench glds rc=$?"; cat gpurun_out/s1_bench_glds.json; tail -20 gpurun_out/s1_bench_glds.err
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_layers.py -x -q > gpurun_out/s1_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/s1_tests.log
