"""Games-parallel sharding helpers: the self-play path shards by independent games, one
process per GPU; the only exchange is a periodic gather of a fixed-size stats record
(the reference has no distributed layer at all and gathers through files,
src/selfplay/pipe.cc:116-175).  Backend is whatever torch.distributed was initialised with:
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist

STAT_KEYS = ("games_done", "nn_queries", "nn_batches", "cache_hits", "moves", "playouts", "records", "elapsed", "halt", "done")


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


class PeriodicGather:
    """The periodic exchange of the games-parallel path: every couple of seconds each rank contributes
    its counters, its halt wish (newer weights seen, reference Engine::ShouldHalt) and whether its own loop has ended.

    Every rank must take part in every round, so a rank whose self-play loop has returned keeps calling `tick(...,
    done=True)` (see `drain`) until all ranks report done -- the rounds are collective calls, ranks that are ahead wait
    for the slowest one there (a few round trips of ~100 bytes per rank)."""

    def __init__(self):
        self.rounds = 0
        self.history = []   # (elapsed_max, total games_done, total nn_queries) per round, rank-agnostic
        self.any_halt = False
        self.all_done = False
        self.last = None

    def tick(self, local: Dict[str, float], halt: bool = False, done: bool = False) -> bool:
        rec = dict(local)
        rec["halt"] = 1.0 if halt else 0.0
        rec["done"] = 1.0 if done else 0.0
        tot = gather_stats(rec)
        world = len(tot["per_rank"])
        self.rounds += 1
        self.any_halt = self.any_halt or tot["halt"] > 0
        self.all_done = tot["done"] >= world
        self.history.append((tot["elapsed_max"], tot["games_done"], tot["nn_queries"]))
        self.last = tot
        return self.any_halt

    def drain(self, final: Dict[str, float]) -> Dict:
        """After the local loop has ended: keep answering rounds until every rank has ended too; returns the last totals."""
        while not self.all_done:
            self.tick(final, halt=self.any_halt, done=True)
        return self.last
