#!/usr/bin/env python3
"""Probe: G independent sub-batches of S/G streams, one host thread + HIP streams each (ctypes releases the GIL inside
sc_push), against one batch of S streams - strict lock-step in every sub-batch.  python tools/split_batch_probe.py [S]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS, WARM = 20, 6
w = bench.make_weights("cuda:0")
eng = None
for G in (1, 2, 4):
    sbs = []
    for g in range(G):
        sb = bench.build_native(w, S // G, 10, False, STEPS + WARM + 2, engine=eng)
        eng = sb.engine
        n = bench.CHUNK * (STEPS + WARM + 2)
        for s in range(sb.S):
            sb.write_pcm(s, 0, bench.synth.synth_audio(g * (S // G) + s, n))
        sbs.append(sb)
    torch.cuda.synchronize()

    def run(sb, n):
        bench.run_steps(sb, n)

    for sb in sbs:
        run(sb, WARM)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(sb, STEPS)) for sb in sbs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S} as {G} x {S // G}: {S * STEPS * bench.CHUNK / 16000.0 / dt:.1f} audio-s/s, {dt / STEPS * 1e3:.2f} ms per chunk step of all streams")
    del sbs
