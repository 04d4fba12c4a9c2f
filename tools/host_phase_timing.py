#!/usr/bin/env python3
"""Host-side phase timers of the engine (SC_TIMING=1) over the bench workload: where a chunk step's wall
time goes as seen from the host thread (frontend / encoder enqueue, decode-graph launches, waiting for the
stop flags).  Usage (GPU box): python tools/host_phase_timing.py [streams] [steps]"""
import os
os.environ.setdefault("SC_TEST_HOOKS", "1")   # the library reads its SC_* switches only with this set
import sys
os.environ["SC_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = 6
sb, be = bench.build_batch(S, 10, False, warm + steps, "cuda:0")
if S > 1:
    sb.set_defer_threshold(3 * S // 8)
bench.preload_audio(sb, warm + steps)
bench.run_steps(sb, warm)
sb.flush()
torch.cuda.synchronize()
sb.timing.clear()
n0 = sum(st.n_steps_total for st in sb.st)
t0 = time.perf_counter()
bench.run_steps(sb, steps)
sb.flush()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
its = sb.stats.get("dec_steps", 0)
print(f"streams {S}: {wall / steps * 1e3:.2f} ms per chunk step, "
      f"{(sum(st.n_steps_total for st in sb.st) - n0) / S / steps:.1f} decode steps per stream-hop")
for k, v in sorted(sb.timing.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {v / steps * 1e3:8.3f} ms per chunk step")
