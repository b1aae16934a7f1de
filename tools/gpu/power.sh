#!/bin/bash
# package power and clocks while the forward runs back to back (evidence for "the matrix cores run at the clock the power budget allows")
cd "$GRAFT_REPO_ROOT" || exit 1
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^=\|^$" | head -20
echo "--- under load"
(timeout 120 python bench.py --steps 20000 --warmup 10 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > /dev/null 2>&1 &)
sleep 30
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ';'; echo; sleep 2; done
wait
