"""Decode iterations of the strict lock-step chunk step by compaction bucket (C++ engine, 128 streams, the bench's
window): iterations per step, ms per step, us per iteration; host issue / wait time per iteration.
    gpurun -- 'python tools/host_times_native.py'   (tools/served_sweep.py prints the same for continuous batching)"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402

S, pre, steps = 128, 26, 20
w = bench.make_weights("cuda:0")
audio = bench.make_audio(S, pre + steps)
sb = bench.build_native(w, S, 10, False, pre + steps)
bench.roll(sb, audio, pre)
a, b = C.c_double(), C.c_double()
sec, it = (C.c_double * 17)(), (C.c_long * 17)()
sb.lib.sc_streams_host_times(sb.handle, C.byref(a), C.byref(b))
sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
st0 = sb.stats["dec_steps"]
t0 = time.perf_counter()
bench.run_host(sb, bench.step_blocks(audio, pre, pre + steps), np.arange(S, dtype=np.int32))
dt = time.perf_counter() - t0
sb.lib.sc_streams_host_times(sb.handle, C.byref(a), C.byref(b))
sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
n = sb.stats["dec_steps"] - st0
print("  bucket (active streams <= k*S/16): iterations per step, ms per step, us per iteration")
for k in range(17):
    if it[k]:
        print(f"    {k:2d}: {it[k] / steps:5.2f}  {sec[k] / steps * 1e3:6.2f}  {sec[k] / it[k] * 1e6:7.1f}")
print(f"S={S}: {dt / steps * 1e3:.2f} ms/step, {n / steps:.1f} iterations/step, per iteration: issue {a.value / n * 1e6:.1f} us, "
      f"wait {b.value / n * 1e6:.1f} us, total loop {(a.value + b.value) / steps * 1e3:.2f} ms/step")
