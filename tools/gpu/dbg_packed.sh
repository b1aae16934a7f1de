#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=default timeout 400 python tools/gpu/dbg_packed.py tiny_all 300 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -x -q 2>&1 | tail -3
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmcdbg; mkdir -p $O
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/plain -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --selfplay-seconds 0 --no-pump > $O/plain.out 2> $O/plain.err); echo "pmc on the tower launch rc=$? $(tail -1 $O/plain.out | cut -c1-120)"
grep -iE "fault|error|abort|signal" $O/plain.err | head -3
ls $O/plain/*/ 2>/dev/null | head
