#!/bin/bash
# the reference-definition games/hour run: configs[2] for 27 minutes from the empty board (finished games / wall over more than
# one generation), every finished game written out as gzip'ed chunks + SGF (scratch directory, the reference's pool of 512
# games in the writer); SHORT=1 adds 90 s at 768, 2048 and 8192 games
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sp
if [ -z "$SKIP_LONG" ]; then
# the box's own microbench first: boxes of the pool differ by +-2.5 %, and the long run is read against it
python bench.py --steps 50 --warmup 10 --selfplay-seconds 0 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('microbench of this box:', d['value'], 'evals/s,', d['ms_per_step'], 'ms per batch; submit/wait packed', d['config'].get('pump_packed',{}).get('nn_evals_per_sec'))" | tee gpurun_out/sp/${TAG:-r05}_microbench.txt
SAYURI_MEMSTAT=1 timeout 2100 python tools/selfplay_bench.py --seconds 1620 --games 512 --num-games 100000 2> gpurun_out/sp/long.err | tail -1 > gpurun_out/sp/${TAG:-r05}_selfplay_27min_512games.json
python -c "
import json
d=json.load(open('gpurun_out/sp/${TAG:-r05}_selfplay_27min_512games.json'))
print({k:d[k] for k in ('nn_evals_per_sec','games_done','games_per_hour','moves_per_sec','mean_batch','host_cpu_cores_busy','max_rss_gb','second_half','chunks_saved','chunks_saved_window','bytes_written','writer_cpu_seconds','writer_cpu_seconds_window','writer_flush_seconds')})"
fi
if [ -n "$SHORT" ]; then
for g in 768 2048 8192; do
  SAYURI_MEMSTAT=1 timeout 300 python tools/selfplay_bench.py --seconds 90 --games $g 2>gpurun_out/sp/g$g.err | tail -1 > gpurun_out/sp/${TAG:-r05}_selfplay_g$g.json
  python -c "
import json
d=json.load(open('gpurun_out/sp/${TAG:-r05}_selfplay_g$g.json'))
print($g, {k:d[k] for k in ('nn_evals_per_sec','mean_batch','host_cpu_cores_busy','max_rss_gb','second_half')})"
done
fi
grep -h memstat gpurun_out/sp/*.err
