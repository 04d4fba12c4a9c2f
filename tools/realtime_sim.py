"""Open-loop real-time load: S concurrent streams, each delivering one 640 ms chunk of 16 kHz audio every 640 ms of WALL
CLOCK time (random phase per stream), served by ONE MI355X through continuous batching (sc_submit / sc_poll) - the
operational meaning of BASELINE.json's "concurrent real-time streams".  A chunk is submitted when it has arrived and
the stream's previous reply has been delivered (one call outstanding per stream, like the reference's session loop,
speechcatcher_server.py:359-397); its latency is reply time - arrival time.  XL dims, beam 10, host PCM in, best
hypothesis read back per reply.
    gpurun -- 'python tools/realtime_sim.py [streams=1024] [seconds=30] [bbd=0] [kv_dtype=float32] [ffn_dtype=float32]'"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
bbd = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
kv = sys.argv[4] if len(sys.argv) > 4 else "float32"
bench.FFN_DTYPE = sys.argv[5] if len(sys.argv) > 5 else "float32"     # "split16": DESIGN section 4a
HOP = bench.CHUNK / 16000.0
n_chunks = int(seconds / HOP)
w = bench.make_weights("cuda:0")
audio = bench.make_audio(S, n_chunks, stream_offset=3000, shared=True).reshape(S, n_chunks, bench.CHUNK)
sb = bench.build_native(w, S, 10, bbd, n_chunks + 2, kv_dtype=kv)
sb.set_encoder_batch(max(8, S // 16))
# warm the graphs / staging with one throw-away chunk on a second batch would cost memory: warm THIS batch's graphs by a
# dry poll instead (captures happen on first use; the first second of the run is excluded from the statistics)
rng = np.random.default_rng(0)
phase = rng.random(S) * HOP
nxt = np.zeros(S, np.int64)                 # next chunk index of a stream
busy = np.zeros(S, bool)                    # a chunk is outstanding
arrival = np.zeros(S)                       # arrival time of the outstanding chunk
lat, when = [], []
poll_s = 0.0
t0 = time.perf_counter()
while True:
    now = time.perf_counter() - t0
    ready = np.nonzero(~busy & (nxt < n_chunks) & (phase + nxt * HOP <= now))[0].astype(np.int32)
    if len(ready):
        sb.submit_block(ready, audio[ready, nxt[ready]])
        arrival[ready] = phase[ready] + nxt[ready] * HOP
        busy[ready] = True
        nxt[ready] += 1
    if busy.any():
        tp = time.perf_counter()
        done, st = sb.poll_ids(1)
        poll_s += time.perf_counter() - tp
        assert (st >= 0).all()
        sb.hypotheses_arrays(done, nbest=1)
        tnow = time.perf_counter() - t0
        lat.extend((tnow - arrival[done]).tolist())
        when.extend([tnow] * len(done))
        busy[done] = False
    elif (nxt >= n_chunks).all():
        break
    else:
        nxt_arr = (phase + nxt * HOP)[nxt < n_chunks].min()
        time.sleep(max(0.0, min(0.002, nxt_arr - (time.perf_counter() - t0))))
wall = time.perf_counter() - t0
lat, when = np.array(lat), np.array(when)
steady = lat[when > 2.0] * 1e3
T = [st.T_enc for st in sb.st]
out = {"streams": S, "seconds_of_audio_per_stream": round(n_chunks * HOP, 2), "wall_s": round(wall, 2), "bbd": int(bbd), "kv_dtype": kv, "ffn_dtype": bench.FFN_DTYPE,
       "replies": int(len(lat)), "engine_busy_fraction": round(poll_s / wall, 3),
       "latency_ms_after_the_first_2s": {"p50": round(float(np.percentile(steady, 50)), 2), "p90": round(float(np.percentile(steady, 90)), 2),
                                         "p99": round(float(np.percentile(steady, 99)), 2), "max": round(float(steady.max()), 2)},
       "encoder_frames_T_at_the_end": [min(T), max(T)],
       "note": "arrival-to-reply latency of every 640 ms chunk; a stream is real-time while its latency stays below the chunk period"}
print(json.dumps(out))
