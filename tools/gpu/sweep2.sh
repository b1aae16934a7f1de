#!/bin/bash
# concurrent-games sweep with the games as fibers (auto above 1024 games), plus a forced-fiber point at 512 and 1024
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
nproc
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err
python -c "import json;d=json.load(open('gpurun_out/q_bench.json'));print('evals/s', d['value'], 'tower us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'whole', d['config']['whole_net_mfma_frac'])"
for g in 1024 2048 4096 8192; do
  timeout 400 python tools/selfplay_bench.py --seconds 90 --games $g --num-games 1000000 --game-threads $([ $g -le 1024 ] && echo 64 || echo 0) > gpurun_out/r02_selfplay_fib_g$g.json 2> gpurun_out/r02_selfplay_fib_g$g.err
  echo "games=$g rc=$?: $(python -c "import json;d=json.load(open('gpurun_out/r02_selfplay_fib_g$g.json'));print({k:d[k] for k in d if k in ('nn_evals_per_sec','second_half','mean_batch','host_cpu_cores_busy','host_sys_cores','max_rss_gb')})" 2>&1 | tail -1)"
done
for g in 512; do
  timeout 400 python tools/selfplay_bench.py --seconds 90 --games $g --num-games 1000000 --game-threads 64 > gpurun_out/r02_selfplay_fib_g$g.json 2> gpurun_out/r02_selfplay_fib_g$g.err
  echo "games=$g on 64 fiber threads rc=$?: $(python -c "import json;d=json.load(open('gpurun_out/r02_selfplay_fib_g$g.json'));print({k:d[k] for k in d if k in ('nn_evals_per_sec','second_half','mean_batch','host_cpu_cores_busy','host_sys_cores','max_rss_gb')})" 2>&1 | tail -1)"
done
