#!/bin/bash
# statistical profile (tools/prof/pcsample.c) of the host side of the configs[2] self-play workload on the GPU box
cd "$GRAFT_REPO_ROOT" || exit 1
gcc -O2 -fPIC -shared tools/prof/pcsample.c -o /tmp/pcsample.so -ldl || exit 1
rm -f /tmp/prof.txt.*
LD_PRELOAD=/tmp/pcsample.so PCSAMPLE_OUT=/tmp/prof.txt timeout 200 python tools/selfplay_bench.py --seconds 40 --games 512 2>/dev/null | tail -1 | cut -c1-300
for f in /tmp/prof.txt.*; do head -45 $f | c++filt | cut -c1-150; done > gpurun_out/selfplay_hostprofile.txt
cat gpurun_out/selfplay_hostprofile.txt
