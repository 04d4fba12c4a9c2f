"""Multi-GPU layer of the hot path: streams are independent, so they are
sharded across ranks (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm) with NO collective in the hot loop.  The only
exchange of the path is the gather of the final text (token ids + their
encoder-frame positions + length + score per stream) at utterance end
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


def pack_hypotheses(ids: Sequence[Sequence[int]], xpos: Sequence[Sequence[int]], scores: Sequence[float], max_len: int,
                    device) -> torch.Tensor:
    """Best hypothesis of every local stream -> ONE fixed-shape int32 tensor [n, 3 + 2*max_len] that a collective can
    move - the payload of SURVEY 8(e): row = [len, score (float64 bits, 2 words), ids[max_len], xpos[max_len]]
    (token ids and their encoder-frame positions, padded with -1; the positions are the token timestamps,
    frame / 24 s: speechcatcher.py:48,522)."""
    n = len(ids)
    out = torch.full((n, 3 + 2 * max_len), -1, dtype=torch.int32)
    sc = torch.tensor([float(x) for x in scores], dtype=torch.float64).view(torch.int32).reshape(n, 2) if n else \
        torch.zeros((0, 2), dtype=torch.int32)
    out[:, 1:3] = sc
    for i in range(n):
        h = [int(t) for t in ids[i]][:max_len]
        x = [int(t) for t in xpos[i]][:max_len]
        assert len(h) == len(x), "every token carries a position"
        out[i, 0] = len(h)
        if h:
            out[i, 3:3 + len(h)] = torch.tensor(h, dtype=torch.int32)
            out[i, 3 + max_len:3 + max_len + len(h)] = torch.tensor(x, dtype=torch.int32)
    return out.to(device)


def unpack_hypotheses(payload: torch.Tensor) -> List[Tuple[List[int], List[int], float]]:
    """rows of pack_hypotheses -> (token ids, positions, score) per stream (rows with length 0: padding)"""
    a = payload.cpu()
    max_len = (a.shape[1] - 3) // 2
    sc = a[:, 1:3].contiguous().view(torch.float64).reshape(-1)
    rows = []
    for i in range(a.shape[0]):
        n = int(a[i, 0])
        rows.append((a[i, 3:3 + n].tolist(), a[i, 3 + max_len:3 + max_len + n].tolist(), float(sc[i])))
    return rows


def gather_final_hypotheses(payload: torch.Tensor, n_local_max: int, group=None):
    """The path's ONE collective: all_gather of the per-rank payloads (RCCL over xGMI with backend "nccl").  Every
    rank passes [n_local, 3 + 2L]; rows are padded to n_local_max (length 0).  Returns per global rank the list of
    (token ids, positions, score)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if payload.shape[0] < n_local_max:
        pad = torch.full((n_local_max - payload.shape[0], payload.shape[1]), -1, dtype=payload.dtype, device=payload.device)
        pad[:, 0:3] = 0
        payload = torch.cat([payload, pad], 0)
    out = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(out, payload.contiguous(), group=group)
    return [unpack_hypotheses(t) for t in out]


def max_over_ranks(value: float, device, group=None) -> float:
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
