"""Games-parallel sharding helpers: the self-play path shards by independent games, one
process per GPU; the only exchange is a periodic gather of a fixed-size stats record
(the reference has no distributed layer at all and gathers through files,
src/selfplay/pipe.cc:116-175).  Backend is whatever torch.distributed was initialised with:
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist

STAT_KEYS = ("games_done", "nn_queries", "nn_batches", "cache_hits", "moves", "playouts", "records", "elapsed")


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) share of `total` independent units (games / positions)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_stats(local: Dict[str, float]) -> Dict:
    """All-gather one small record per rank; returns sums, the max elapsed time and the per-rank
    records.  O(100 B) per rank: latency-bound on any fabric, issued every few seconds at most."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    vec = torch.tensor([float(local.get(k, 0.0)) for k in STAT_KEYS], dtype=torch.float64)
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        vec = vec.to(dev)
        out = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(out, vec)
        rows = [o.cpu().tolist() for o in out]
    else:
        rows = [vec.tolist()]
    per_rank = [dict(zip(STAT_KEYS, r)) for r in rows]
    total = {k: sum(r[k] for r in per_rank) for k in STAT_KEYS if k != "elapsed"}
    total["elapsed_max"] = max(r["elapsed"] for r in per_rank)
    total["per_rank"] = per_rank
    return total
