#!/bin/bash
# The path's one collective beside the persistent launch, on the one GPU of the box: bench.py's self-play window with
# torch.distributed initialised on nccl (= RCCL) for a world of one, with and without the periodic exchange, interleaved.
# Writes gpurun_out/rccl/{ex,noex}_K.json and a summary (copied to profiles/r05_rccl_tick.txt).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/rccl
SECS=${SECS:-60}
for k in 1 2; do
  for mode in ex noex; do
    extra=""; [ "$mode" = noex ] && extra="--no-exchange"
    MASTER_ADDR=127.0.0.1 MASTER_PORT=2957$k timeout 600 python bench.py --steps 10 --warmup 3 --force-dist --dist-backend nccl \
      --selfplay-seconds $SECS --no-cpu-baseline --no-config5 --no-pump $extra > gpurun_out/rccl/${mode}_$k.json 2> gpurun_out/rccl/${mode}_$k.err
  done
done
python - <<'PY' | tee gpurun_out/rccl/summary.txt
import json, glob
rows = {}
for f in sorted(glob.glob("gpurun_out/rccl/*_?.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(f, "unreadable", e); continue
    sp = d["selfplay"]
    rows[f] = sp
    print(f.split("/")[-1], "self-play evals/s", sp["nn_evals_per_sec"], "microbench", d["value"], "mean batch", sp["mean_batch"],
          "rounds", sp["exchange_rounds"], "exchange", sp["exchange"])
ex = [v["nn_evals_per_sec"] for k, v in rows.items() if "/ex_" in k]
no = [v["nn_evals_per_sec"] for k, v in rows.items() if "/noex_" in k]
if ex and no:
    a, b = sum(ex) / len(ex), sum(no) / len(no)
    print("with the exchange %.1f evals/s, without %.1f: %+.2f %%" % (a, b, 100 * (a - b) / b))
PY
