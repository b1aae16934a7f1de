#!/bin/bash
# configs[2] (512 games, batch 256, writer on) with the packed records read in place by the first kernel (default) against copied first
# (SAYURI_IO_ZC_IN=0: rounds 2-4), interleaved on one box; SAYURI_HIP_FWDSTAT gives the device-side time from a forward's end to its
# results' arrival (do the downloads still take the copy engine?), SAYURI_PIPE_TRACE the pump's waiting per batch.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sp_zc_in
SECS=${SECS:-60}
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc $(nproc)"
python bench.py --steps 50 --warmup 10 --selfplay-seconds 0 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench', d['value'], d['ms_per_step'], 'pump_packed', d['config'].get('pump_packed',{}).get('nn_evals_per_sec'))"
for round in 1 2; do
for zc in 1 0; do
  name=zc_in_${zc}_$round
  SAYURI_IO_ZC_IN=$zc SAYURI_PIPE_TRACE=1 SAYURI_HIP_FWDSTAT=1 timeout 300 python tools/selfplay_bench.py --seconds $SECS --games 512 --stagger 360 2> gpurun_out/sp_zc_in/$name.err | tail -1 > gpurun_out/sp_zc_in/$name.json
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open('gpurun_out/sp_zc_in/%s.json'%n))
    sh=d.get('second_half',{})
    print("== %-12s evals/s %8.1f (2nd half %8.1f) mean_batch %.1f partial %d/%d cores %.1f pump %s"%(n,d['nn_evals_per_sec'],sh.get('nn_evals_per_sec',0),d['mean_batch'],d['partial_batches'],d['batches'],d['host_cpu_cores_busy'],d['pump_us_per_batch']))
except Exception as e:
    print("== %s FAILED %s"%(n,e))
PY
  grep -h "fwdstat" gpurun_out/sp_zc_in/$name.err | cut -c1-400
done
done
