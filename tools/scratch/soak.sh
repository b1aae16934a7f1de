mkdir -p /tmp/selfplay_out
timeout 1800 python tools/selfplay_bench.py --seconds 1620 --games 512 --num-games 100000 --out /tmp/selfplay_out > gpurun_out/soak.log 2>&1
tail -1 gpurun_out/soak.log > gpurun_out/soak.json
du -sh /tmp/selfplay_out | tail -1
ls /tmp/selfplay_out | head
find /tmp/selfplay_out -name "*.gz" | wc -l
tail -1 gpurun_out/soak.log | cut -c1-1200
