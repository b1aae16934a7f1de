"""Games-parallel sharding helpers: the self-play path shards by independent games, one
process per GPU; the only exchange is a periodic gather of a fixed-size stats record
(the reference has no distributed layer at all and gathers through files,
src/selfplay/pipe.cc:116-175).  Backend is whatever torch.distributed was initialised with:
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

STAT_KEYS = ("games_done", "nn_queries", "nn_batches", "cache_hits", "moves", "playouts", "records", "elapsed", "halt", "done")


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) share of `total` independent units (games / positions)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _reduce_rows(rows: List[List[float]]) -> Dict:
    per_rank = [dict(zip(STAT_KEYS, r)) for r in rows]
    total = {k: sum(r[k] for r in per_rank) for k in STAT_KEYS if k != "elapsed"}
    total["elapsed_max"] = max(r["elapsed"] for r in per_rank)
    total["per_rank"] = per_rank
    return total


def _device() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_stats(local: Dict[str, float]) -> Dict:
    """All-gather one small record per rank (blocking); returns sums, the max elapsed time and the per-rank
    records.  O(100 B) per rank: latency-bound on any fabric, issued every few seconds at most.  With an initialised
    process group the collective runs whatever the world size (a world of one still goes through RCCL on the nccl
    backend: that is how the path is exercised on a one-GPU box)."""
    vec = torch.tensor([float(local.get(k, 0.0)) for k in STAT_KEYS], dtype=torch.float64)
    if dist.is_initialized():
        world = dist.get_world_size()
        vec = vec.to(_device())
        out = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(out, vec)
        rows = [o.cpu().tolist() for o in out]
    else:
        rows = [vec.tolist()]
    return _reduce_rows(rows)


class PeriodicGather:
    """The periodic exchange of the games-parallel path: every couple of seconds each rank contributes
    its counters, its halt wish (newer weights seen, reference Engine::ShouldHalt) and whether its own loop has ended.

    Every rank takes part in every round, in the same order; a rank whose self-play loop has returned keeps calling
    `tick(..., done=True)` (see `drain`) until all ranks report done.

    A round never blocks the caller for longer than `timeout` seconds.  On the GPU the record travels through RCCL
    kernels and two small copies on torch's stream, and those are KERNELS to the runtime: beside the persistent tower
    launch of the evaluation engine, which holds every CU for a whole forward, they wait for a CU
    like any other small kernel.  So the collective and the read-back are issued asynchronously (`async_op`, pinned
    read-back buffer + event) and polled; a round that has not landed by the deadline stays in flight, the caller goes
    on with the totals of the last completed round, and the next `tick` first collects it (no new round is issued while
    one is in flight, so the ranks' sequences of collectives stay identical).  `latencies_ms` holds issue -> landed per
    completed round, `late_rounds` counts the rounds that missed their deadline."""

    def __init__(self, timeout: float = 1.0, poll: float = 2e-4):
        self.rounds = 0
        self.history = []   # (elapsed_max, total games_done, total nn_queries) per round, rank-agnostic
        self.any_halt = False
        self.all_done = False
        self.last: Optional[Dict] = None
        self.timeout = float(timeout)
        self.poll = float(poll)
        self.latencies_ms: List[float] = []
        self.late_rounds = 0
        self.skipped_ticks = 0
        self.backend = dist.get_backend() if dist.is_initialized() else "none"
        self._inflight = None  # (work, event | None, t_issue, late)
        self._world = dist.get_world_size() if dist.is_initialized() else 1
        if dist.is_initialized():
            dev = _device()
            pin = dev.type == "cuda"
            self._src_host = torch.zeros(len(STAT_KEYS), dtype=torch.float64, pin_memory=pin)
            self._dst_host = torch.zeros(self._world * len(STAT_KEYS), dtype=torch.float64, pin_memory=pin)
            self._src = torch.zeros(len(STAT_KEYS), dtype=torch.float64, device=dev)
            self._dst = torch.zeros(self._world * len(STAT_KEYS), dtype=torch.float64, device=dev)
            # a side stream: the exchange never waits for (or holds up) whatever else this process has on torch's stream
            self._stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    # ---- one round: issue, poll, land
    def _issue(self, rec: Dict[str, float]) -> None:
        for i, k in enumerate(STAT_KEYS):
            self._src_host[i] = float(rec.get(k, 0.0))
        t0 = time.perf_counter()
        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                self._src.copy_(self._src_host, non_blocking=True)
                work = dist.all_gather_into_tensor(self._dst, self._src, async_op=True)
                work.wait()  # stream-level dependency only (nccl): the host does not block here
                self._dst_host.copy_(self._dst, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            self._inflight = (work, ev, t0, False)
        else:
            self._src.copy_(self._src_host)
            work = dist.all_gather_into_tensor(self._dst, self._src, async_op=True)
            self._inflight = (work, None, t0, False)

    def _landed(self) -> bool:
        work, ev, _, _ = self._inflight
        if ev is not None:
            return ev.query()
        return work.is_completed()

    def _collect(self) -> None:
        work, ev, t0, late = self._inflight
        if ev is None:
            work.wait()
            self._dst_host.copy_(self._dst)
        else:
            ev.synchronize()  # the read-back has landed (a no-op for every caller that polled _landed() first)
        self._inflight = None
        self.latencies_ms.append((time.perf_counter() - t0) * 1e3)
        rows = self._dst_host.view(self._world, len(STAT_KEYS)).tolist()
        self._account(_reduce_rows(rows))

    def _account(self, tot: Dict) -> None:
        world = len(tot["per_rank"])
        self.rounds += 1
        self.any_halt = self.any_halt or tot["halt"] > 0
        self.all_done = tot["done"] >= world
        self.history.append((tot["elapsed_max"], tot["games_done"], tot["nn_queries"]))
        self.last = tot

    def _wait(self, deadline: float) -> bool:
        while not self._landed():
            if time.perf_counter() >= deadline:
                return False
            time.sleep(self.poll)
        return True

    def tick(self, local: Dict[str, float], halt: bool = False, done: bool = False) -> bool:
        rec = dict(local)
        rec["halt"] = 1.0 if halt else 0.0
        rec["done"] = 1.0 if done else 0.0
        if not dist.is_initialized():
            self._account(_reduce_rows([[float(rec.get(k, 0.0)) for k in STAT_KEYS]]))
            self.latencies_ms.append(0.0)
            return self.any_halt
        if self.all_done and self._inflight is None:
            return self.any_halt  # every rank has reported done and seen that round: a new collective would have no peer
        deadline = time.perf_counter() + self.timeout
        if self._inflight is not None:  # a round that missed its deadline: collect it first, issue nothing beside it
            if not self._wait(deadline):
                self.skipped_ticks += 1
                return self.any_halt
            self._collect()
            if self.all_done:  # every rank is draining and has seen this round: nobody issues another one
                return self.any_halt
        self._issue(rec)
        if self._wait(deadline):
            self._collect()
        else:
            work, ev, t0, _ = self._inflight
            self._inflight = (work, ev, t0, True)
            self.late_rounds += 1
        return self.any_halt

    def drain(self, final: Dict[str, float]) -> Dict:
        """After the local loop has ended: keep answering rounds until every rank has ended too; returns the last totals."""
        while not self.all_done:
            self.tick(final, halt=self.any_halt, done=True)
        if self._inflight is not None:  # nothing may stay in flight behind the caller's back
            if not self._wait(time.perf_counter() + 60.0):
                raise RuntimeError("PeriodicGather.drain: the last exchange round has not landed after 60 s (a peer rank is gone?)")
            self._collect()
        return self.last

    def latency_summary(self) -> Dict:
        """p50 / p99 / max of issue -> landed over the completed rounds (ms), for the bench line and profiles/."""
        xs = sorted(self.latencies_ms)
        if not xs:
            return {"rounds": 0}
        q = lambda p: xs[min(len(xs) - 1, int(p * len(xs)))]
        return {"rounds": len(xs), "backend": self.backend, "world": self._world, "p50_ms": round(q(0.5), 3), "p99_ms": round(q(0.99), 3),
                "max_ms": round(xs[-1], 3), "mean_ms": round(sum(xs) / len(xs), 3), "late_rounds": self.late_rounds,
                "skipped_ticks": self.skipped_ticks, "timeout_s": self.timeout}
