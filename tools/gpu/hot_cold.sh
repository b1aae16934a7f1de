#!/bin/bash
# Is the long self-play run's ~70 k evals/s the box or the chip's steady state?  The microbench on a fresh box (cold), eight minutes of
# configs[2] self-play with per-minute rates, the microbench again at once (hot), and clocks / power / temperature beside it.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/hot_cold; mkdir -p $O
mb() { python bench.py --steps 100 --warmup 10 --selfplay-seconds 0 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 microbench:', d['value'], 'evals/s,', d['ms_per_step'], 'ms; tower launch', d['roofline']['avg_launch_us'], 'us; submit/wait packed', d['config'].get('pump_packed',{}).get('nn_evals_per_sec'))"; }
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|Temperature" | tr '\n' ';' | cut -c1-400; echo; }
echo "idle: $(smi)"
mb cold | tee $O/cold.txt
( for i in $(seq 1 16); do sleep 30; echo "t=$((i*30))s $(smi)"; done ) > $O/smi.txt &
SMI=$!
timeout 900 python tools/selfplay_bench.py --seconds ${SECS:-480} --games 512 --num-games 100000 --stagger 360 2> $O/sp.err | tail -1 > $O/sp.json
python -c "
import json
d=json.load(open('$O/sp.json'))
print({k:d.get(k) for k in ('nn_evals_per_sec','mean_batch','host_cpu_cores_busy','second_half','pump_us_per_batch')})"
mb hot | tee $O/hot.txt
sleep 60
mb after_60s_idle | tee -a $O/hot.txt
kill $SMI 2>/dev/null
cat $O/smi.txt
