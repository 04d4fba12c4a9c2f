#!/usr/bin/env python3
"""Host-only cost of StreamBatch.push (no kernels): a null backend that only
drives the control flow.  CPU, no GPU needed."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechcatcher_amd import synth
from speechcatcher_amd.config import MICRO, SearchConfig
from speechcatcher_amd.engine import StreamBatch
from speechcatcher_amd.weights import PackedWeights


class NullBackend:
    def __init__(self, steps_per_block=8):
        self.n = 0
        self.spb = steps_per_block

    def __getattr__(self, name):
        def f(*a, **k):
            return None
        return f

    def decode_step(self, sb):
        self.n += 1
        sb.flags.fill_(1 if self.n % self.spb == 0 else 0)


S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = MICRO
sd = synth.make_state_dict(cfg, 1)
w = PackedWeights(sd, cfg, "cpu")
sb = StreamBatch(w, NullBackend(), S, SearchConfig(beam_size=10), max_frames=700, max_tokens=400,
                 pcm_capacity=10240 * 40, max_chunk_samples=10240)
items = [(s, 10240, False) for s in range(S)]
for _ in range(5):
    sb.push(items, pcm_resident=True)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
n = 10
for _ in range(n):
    sb.push(items, pcm_resident=True)
pr.disable()
dt = (time.perf_counter() - t0) / n
print(f"host-only push: {dt*1e3:.2f} ms per step at S={S}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
