"""Multi-GPU layer of the hot path: streams are independent, so they are
sharded across ranks (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm) with NO collective in the hot loop.  The only
exchange of the path is the gather of final token ids at utterance end
(the reference's equivalent is its per-segment process
pool, speechcatcher/speechcatcher.py:474-497)."""
from typing import List, Sequence, Tuple

import torch


def shard_streams(n_streams_total: int, rank: int, world: int) -> range:
    """Contiguous block partition of global stream ids for this rank (block
    sizes differ by at most one)."""
    base, rem = divmod(n_streams_total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def pack_hypotheses(hyps: Sequence[Sequence[int]], scores: Sequence[float], max_len: int,
                    device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Token ids of the best hypothesis of every local stream -> fixed-shape
    tensors (ids [n, max_len] int32 padded with -1 + length in column 0 of a
    second tensor, score float64) that a collective can move."""
    n = len(hyps)
    ids = torch.full((n, max_len + 1), -1, dtype=torch.int32)
    for i, h in enumerate(hyps):
        h = list(h)[:max_len]
        ids[i, 0] = len(h)
        if h:
            ids[i, 1:1 + len(h)] = torch.tensor(h, dtype=torch.int32)
    sc = torch.tensor(list(scores), dtype=torch.float64)
    return ids.to(device), sc.to(device)


def gather_final_hypotheses(ids: torch.Tensor, scores: torch.Tensor, n_local_max: int, group=None):
    """all_gather of the (padded) per-rank results.  Every rank passes tensors
    of the same shape [n_local_max, L+1] / [n_local_max] (pad rows with
    length 0).  Returns per global rank the list of (token ids, score)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if ids.shape[0] < n_local_max:
        pad = torch.full((n_local_max - ids.shape[0], ids.shape[1]), -1, dtype=ids.dtype, device=ids.device)
        pad[:, 0] = 0
        ids = torch.cat([ids, pad], 0)
        scores = torch.cat([scores, torch.zeros(n_local_max - scores.shape[0], dtype=scores.dtype, device=scores.device)])
    out_ids = [torch.empty_like(ids) for _ in range(world)]
    out_sc = [torch.empty_like(scores) for _ in range(world)]
    dist.all_gather(out_ids, ids, group=group)
    dist.all_gather(out_sc, scores, group=group)
    res: List[List[Tuple[List[int], float]]] = []
    for r in range(world):
        rows = []
        a, s = out_ids[r].cpu(), out_sc[r].cpu()
        for i in range(a.shape[0]):
            n = int(a[i, 0])
            rows.append((a[i, 1:1 + n].tolist(), float(s[i])))
        res.append(rows)
    return res


def max_over_ranks(value: float, device, group=None) -> float:
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
