#!/usr/bin/env python3
"""Probe: can the encoder pass of the NEXT chunk step hide behind the latency-bound
decode steps of the current one?  Replays the captured encoder-layers graph and a
captured decode-step graph (a) alone, (b) concurrently from two host threads on two
HIP streams, and reports the time per replay.  Usage (GPU box): python tools/overlap_probe.py"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

S, STEPS = 128, 8
sb, be = bench.build_batch(S, 10, False, STEPS + 4, "cuda:0")
bench.preload_audio(sb, STEPS + 4)
bench.run_steps(sb, STEPS)
torch.cuda.synchronize()
enc_graph = list(be._enc_graphs.values())[-1]
dec_graphs = sb._sc_decode_graphs
dec_graph = dec_graphs[max(dec_graphs)]          # full-occupancy bucket
sE, sD = torch.cuda.Stream(), torch.cuda.Stream()
be.bind_stream(sE)
be.bind_stream(sD)
lib = be.lib


def loop(graph, stream, n, out, key):
    t0 = time.perf_counter()
    for _ in range(n):
        lib.sc_graph_launch(graph, stream.cuda_stream)
    stream.synchronize()
    out[key] = (time.perf_counter() - t0) / n * 1e3


res = {}
for _ in range(2):
    loop(enc_graph, sE, 5, res, "enc_alone_ms")
    loop(dec_graph, sD, 40, res, "dec_alone_ms")
n_enc, n_dec = 10, int(10 * res["enc_alone_ms"] / res["dec_alone_ms"])
tA = threading.Thread(target=loop, args=(enc_graph, sE, n_enc, res, "enc_concurrent_ms"))
tB = threading.Thread(target=loop, args=(dec_graph, sD, n_dec, res, "dec_concurrent_ms"))
t0 = time.perf_counter()
tA.start(); tB.start(); tA.join(); tB.join()
wall = (time.perf_counter() - t0) * 1e3
serial = n_enc * res["enc_alone_ms"] + n_dec * res["dec_alone_ms"]
print({k: round(v, 3) for k, v in res.items()})
print(f"{n_enc} encoder passes + {n_dec} decode steps: serial {serial:.1f} ms, concurrent wall {wall:.1f} ms "
      f"({serial / wall:.2f}x)")
