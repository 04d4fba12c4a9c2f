#!/usr/bin/env python3
"""Experiment: the 128 streams of the default bench as TWO independent batches of 64 on the same GPU, each served by its own
host thread (sc_submit / sc_poll release the GIL) - the decoder layer kernels of a 64-stream batch occupy half of the CUs
(four heads per 1024-thread workgroup: one workgroup per CU), so the prologue / projection phases of one batch could run
beside the K|V walks of the other.  Prints audio-s/s of 1 x 128, 2 x 64 and 4 x 32.
Usage (GPU box): python tools/two_batches.py [steps]"""
import os
import sys
import threading
import time
os.environ.setdefault("SC_TEST_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S, preroll, warm = 128, 21, 5
total = preroll + warm + steps + bench.SERVED_SPARE
w = bench.make_weights("cuda:0")
audio = bench.make_audio(S, total)
for parts in (1, 2, 4):
    n = S // parts
    sbs, a3s, nxts = [], [], []
    for k in range(parts):
        a = audio[k * n:(k + 1) * n]
        sb = bench.build_native(w, n, 10, False, total)
        if os.environ.get("SC_HPW_MIN_PARTS"):
            pass
        bench.roll(sb, a, preroll)
        sbs.append(sb)
        a3s.append(a.reshape(n, -1, bench.CHUNK))
        nxts.append(np.full(n, preroll, np.int64))
    group = max(1, n // 8)
    res = [None] * parts

    def run(k, nsteps):
        res[k] = bench.serve(sbs[k], a3s[k], nxts[k], nsteps, group)

    for nsteps in (warm, steps):
        th = [threading.Thread(target=run, args=(k, nsteps)) for k in range(parts)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
    # every part stops its clock after its own n x steps replies; the whole job: all parts' audio over the slowest part
    slowest = max(r["elapsed"] for r in res)
    print(f"{parts} x {n} streams: {S * steps * bench.CHUNK / 16000.0 / slowest:8.1f} audio-s/s (per part "
          f"{[round(n * steps * 0.64 / r['elapsed'], 1) for r in res]}, wall incl. drain {wall:.3f} s)", flush=True)
    for sb in sbs:
        sb.close()
