"""The N>1 path on CPU: two gloo ranks shard 5 streams (3+2), each rank runs
its own streams through the engine (spec backend) and the final-text gather
reassembles all results on every rank; sharding changes nothing in the
per-stream output."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from speechcatcher_amd import synth
    from speechcatcher_amd.distributed import (gather_final_hypotheses, max_over_ranks, pack_hypotheses,
                                               shard_streams)
    from test_engine_spec import make_batch
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mine = shard_streams(n_total, rank, world)
    sb = make_batch("TINY", 1234, "meanstd", 5, False, n_streams=len(mine), max_frames=96, max_tokens=200,
                    pcm_capacity=1 << 16)
    n = 30000
    for pos in range(0, n, 10240):
        end = min(pos + 10240, n)
        sb.push([(i, synth.synth_audio(100 + g, n)[pos:end], end >= n) for i, g in enumerate(mine)])
    hyps = [sb.hypotheses(i)[0] for i in range(len(mine))]
    payload = pack_hypotheses([h["yseq"] for h in hyps], [h["xpos"] for h in hyps], [h["score"] for h in hyps], 256, "cpu")
    n_max = -(-n_total // world)
    allres = gather_final_hypotheses(payload, n_max)
    t = max_over_ranks(float(rank + 1), "cpu")
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = (list(mine), allres, t)


def test_two_rank_sharding_and_gather():
    world, n_total = 2, 5
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_total, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    mine0, res0, t0 = ret[0]
    mine1, res1, t1 = ret[1]
    assert mine0 == [0, 1, 2] and mine1 == [3, 4]
    assert t0 == t1 == 2.0
    assert res0 == res1                       # every rank sees the same gathered result
    flat = [row for r in range(world) for row in res0[r] if row[0]]
    assert len(flat) == n_total
    # unsharded run of the same 5 streams gives the same hypotheses
    from speechcatcher_amd import synth
    from test_engine_spec import make_batch
    sb = make_batch("TINY", 1234, "meanstd", 5, False, n_streams=n_total, max_frames=96, max_tokens=200,
                    pcm_capacity=1 << 16)
    n = 30000
    for pos in range(0, n, 10240):
        end = min(pos + 10240, n)
        sb.push([(g, synth.synth_audio(100 + g, n)[pos:end], end >= n) for g in range(n_total)])
    for g in range(n_total):
        h = sb.hypotheses(g)[0]
        assert flat[g][0] == h["yseq"] and flat[g][1] == h["xpos"]   # ids and their token timestamps (frame positions)
        assert abs(flat[g][2] - h["score"]) < 1e-4   # batch size changes the CPU GEMM blocking
