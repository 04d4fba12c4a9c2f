import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import bench
for S in (128,):
    w = bench.make_weights("cuda:0")
    sb = bench.build_native(w, S, 10, False, 30)
    bench.preload_audio(sb, 30)
    bench.run_steps(sb, 6)
    a, b = C.c_double(), C.c_double()
    sb.lib.sc_streams_host_times(sb.handle, C.byref(a), C.byref(b))
    sec, it = (C.c_double * 17)(), (C.c_long * 17)()
    sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
    st0 = sb.stats["dec_steps"]
    t0 = time.perf_counter()
    bench.run_steps(sb, 20)
    dt = time.perf_counter() - t0
    sb.lib.sc_streams_host_times(sb.handle, C.byref(a), C.byref(b))
    n = sb.stats["dec_steps"] - st0
    sec, it = (C.c_double * 17)(), (C.c_long * 17)()
    sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
    print("  bucket (active streams <= k*S/16): iterations per step, ms per step, us per iteration")
    for k in range(17):
        if it[k]:
            print(f"    {k:2d}: {it[k]/20:5.2f}  {sec[k]/20*1e3:6.2f}  {sec[k]/it[k]*1e6:7.1f}")
    print(f"S={S}: {dt/20*1e3:.2f} ms/step, {n/20:.1f} iterations/step, per iteration: issue {a.value/n*1e6:.1f} us, wait {b.value/n*1e6:.1f} us, total loop {(a.value+b.value)/20*1e3:.2f} ms/step")
    del sb
