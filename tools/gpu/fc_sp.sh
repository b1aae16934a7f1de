#!/bin/bash
# round 4: the squeeze FC with four rows in flight + ReLU / identity in the generated epilogue (parity, old/new), then the self-play
# queue's trace on the new build (where do the 512 games lose against the resident-input rate?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fcsp
cp sayuri_amd/lib/libsayuri_hip_new.so sayuri_amd/lib/libsayuri_hip.so
timeout 900 python -m pytest tests/test_gpu_smallops.py tests/test_gpu_net.py tests/test_gpu_layers.py -m gpu -x -q --timeout 300 2>&1 | tail -4
for v in old new old new old new; do
cp sayuri_amd/lib/libsayuri_hip_$v.so sayuri_amd/lib/libsayuri_hip.so
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$v', 'evals/s', d['value'], 'ms/step', d['ms_per_step'], 'whole-net', d['config']['whole_net_mfma_frac'], 'dominant us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
cp sayuri_amd/lib/libsayuri_hip_new.so sayuri_amd/lib/libsayuri_hip.so
run() {
  local name=$1; shift
  ( env "$@" SAYURI_PIPE_TRACE=1 SAYURI_HIP_FWDSTAT=1 timeout 300 python tools/selfplay_bench.py --seconds 60 --games ${GAMES:-512} --stagger 360 ) 2> gpurun_out/fcsp/$name.err | tail -1 > gpurun_out/fcsp/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/fcsp/%s.json'%n))
    sh=d.get('second_half',{})
    print("== %-14s evals/s %8.1f (2nd half %8.1f) mean_batch %.1f partial %d/%d cores %.1f sys %.1f pump %s"%(n,d['nn_evals_per_sec'],sh.get('nn_evals_per_sec',0),d['mean_batch'],d['partial_batches'],d['batches'],d['host_cpu_cores_busy'],d['host_sys_cores'],d['pump_us_per_batch']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  grep -h "fwdstat\|closed:\|arrivals\|runs again:" gpurun_out/fcsp/$name.err | cut -c1-400
}
run default A=1


GAMES=768 run g768 A=1
run tail60 SAYURI_PIPE_TAIL=0.6
