"""Continuous batching (sc_submit / sc_poll) at 128 streams: throughput against the number of replies sc_poll waits for
(= the size of the next admission group), with the decode iterations by compaction bucket.
    gpurun -- 'python tools/served_sweep.py 4 8 16 32 e1 e64 [bbd] [d2]'      (eN: encoder batch of N streams; bbd: block-
boundary detection on; dN: N chunks per stream at the engine)"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402

BBD = "bbd" in sys.argv[1:]
DEPTH = max([1] + [int(a[1:]) for a in sys.argv[1:] if a.startswith("d") and a[1:].isdigit()])
groups = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 16, 32]
encb = [int(a[1:]) for a in sys.argv[1:] if a.startswith("e") and a[1:].isdigit()] or [64]
S, pre, warm, steps = 128, 21, 5, 20
w = bench.make_weights("cuda:0")
total = pre + warm + steps + bench.SERVED_SPARE + 2
audio = bench.make_audio(S, total)
def clear(sb):
    sec, it = (C.c_double * 17)(), (C.c_long * 17)()
    sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
    a, b = C.c_double(), C.c_double()
    sb.lib.sc_streams_host_times(sb.handle, C.byref(a), C.byref(b))


def hist(sb):
    sec, it = (C.c_double * 17)(), (C.c_long * 17)()
    sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
    a, b = C.c_double(), C.c_double()
    sb.lib.sc_streams_host_times(sb.handle, C.byref(a), C.byref(b))
    tot = sum(it)
    return (" ".join(f"{k}:{it[k] / steps:.1f}x{sec[k] / max(it[k], 1) * 1e3:.2f}" for k in range(17) if it[k]) +
            f" | loop {sum(sec) / steps * 1e3:.2f} ms/step (issue {a.value / steps * 1e3:.2f}, wait {b.value / steps * 1e3:.2f}), {tot / steps:.1f} it/step")


a3 = audio.reshape(S, -1, bench.CHUNK)


def run(mode, g=8, eb=64):
    sb = bench.build_native(w, S, 10, BBD, total)
    sb.set_encoder_batch(eb)
    if DEPTH > 1:
        sb.set_queue_depth(DEPTH)
    bench.roll(sb, audio, pre)
    if mode == "strict":
        bench.run_host(sb, bench.step_blocks(audio, pre, pre + warm), np.arange(S, dtype=np.int32))
        e, dsh, _ = bench.strict_window(sb, audio, pre + warm, steps, before_timing=clear)
        print(f"strict: {S * steps * 0.64 / e:8.1f} audio-s/s  {e / steps * 1e3:6.2f} ms/step  {dsh:.2f} decode steps/hop\n"
              f"    bucket:iterations per step x ms  {hist(sb)}", flush=True)
    else:
        nxt = np.full(S, pre, np.int64)
        bench.serve(sb, a3, nxt, warm, g, depth=DEPTH)
        n0, t0c = C.c_long(), C.c_double()
        sb.lib.sc_streams_capture_stats(sb.handle, C.byref(n0), C.byref(t0c))
        r = bench.serve(sb, a3, nxt, steps, g, before_timing=clear, depth=DEPTH)
        e = r["elapsed"]
        n1, t1c = C.c_long(), C.c_double()
        sb.lib.sc_streams_capture_stats(sb.handle, C.byref(n1), C.byref(t1c))
        print(f"    encoder graphs captured in the timed window (+drain): {n1.value - n0.value} in {(t1c.value - t0c.value) * 1e3:.1f} ms of host time "
              f"({n0.value} before)")
        print(f"group {g:3d} encoder batch {eb:3d}: {S * steps * 0.64 / e:8.1f} audio-s/s  {e / steps * 1e3:6.2f} ms/step-eq  "
              f"{r['iterations_per_step']:5.2f} iterations/step  {r['polls_per_step']:.1f} polls/step  spread {r['chunks_per_stream_min_max']}\n"
              f"    bucket:iterations per step x ms  {hist(sb)}", flush=True)
    sb.close()


run("strict")
for g, eb in [(g, eb) for eb in encb for g in groups]:
    run("continuous", g, eb)
