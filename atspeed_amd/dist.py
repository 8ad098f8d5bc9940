"""User sharding across the GPUs of one node.

Users are independent units of the beam-SD path (the reference handles one user per call,
`code/inference.py:162-176`, and offers manual slicing with `--L/--R`, `:123-124`).  Each rank
owns a contiguous slice of the user list with a full weight replica; the ONLY exchange is one
all-gather of per-rank counters at the end (RCCL over xGMI with backend "nccl", gloo on CPU).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

COUNTER_FIELDS = ("n_users", "n_run", "accept_steps", "elapsed_ns")


def shard_range(n_users: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of `rank`'s users: contiguous, sizes differ by at most one, covers everything."""
    base, rem = divmod(n_users, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class Counters:
    n_users: int = 0
    n_run: int = 0
    accept_steps: int = 0
    elapsed_ns: int = 0

    def add_result(self, out: Dict) -> None:
        """Accumulate one BSSD() result (beamSD.py:527-541 keys)."""
        self.n_users += 1
        self.n_run += int(out["n_run"])
        self.accept_steps += int(out["total_accept_steps"])

    def tensor(self, device) -> torch.Tensor:
        return torch.tensor([self.n_users, self.n_run, self.accept_steps, self.elapsed_ns], dtype=torch.int64, device=device)


def all_gather_counters(c: Counters, device="cpu") -> List[Counters]:
    """The path's single collective: int64[4] per rank."""
    t = c.tensor(device)
    if not (dist.is_available() and dist.is_initialized()):
        return [c]
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [Counters(*[int(x) for x in o.cpu().tolist()]) for o in outs]


def aggregate(per_rank: Sequence[Counters], items_per_user: int) -> Dict[str, float]:
    """Whole-job numbers: items/s uses the slowest rank's time (max over ranks)."""
    users = sum(c.n_users for c in per_rank)
    runs = sum(c.n_run for c in per_rank)
    acc = sum(c.accept_steps for c in per_rank)
    t = max(c.elapsed_ns for c in per_rank) * 1e-9
    return {"users": users, "items_per_s": users * items_per_user / t if t > 0 else 0.0,
            "mean_accept_len": acc / runs if runs else 0.0, "elapsed_s": t,
            "ave_accept_tokens": acc * items_per_user / runs if runs else 0.0}
