#!/bin/bash
# board-convolution timeline at several batch sizes: does the epilogue shrink when fewer workgroups share the chip?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for b in 256 128 64 16; do
echo "== batch $b"
SAYURI_BOARD_DBG=5 timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --selfplay-seconds 0 --profile > gpurun_out/tlb_$b.json 2> gpurun_out/tlb_$b.err
grep "board timeline wg0 wave[04]\|board timeline wg3 wave[04]" gpurun_out/tlb_$b.err
grep "conv3x3_tower " gpurun_out/tlb_$b.err
done
