"""Served throughput THROUGH THE SERVER LOOP (speechcatcher_amd.server_session.ServerLoop over StreamScheduler over the
C++ engine): S connected clients, each sends its next 640 ms int16 chunk as soon as it has the reply to the previous
one (the reference's per-client handler loop, speechcatcher_server.py:359-397).  strict: one batched sc_push per step,
every client waits for the slowest stream of the batch; continuous: sc_submit / sc_poll - a client is answered when
ITS chunk is decoded.  Endpointing is switched off (finalize_update_iters huge) so that both legs decode the same
audio; XL dims, beam 10, the bench's window (21 pre-roll chunks).
    gpurun -- 'python tools/serve_bench.py [streams] [chunks per client]'"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from speechcatcher_amd.scheduler import StreamScheduler  # noqa: E402
from speechcatcher_amd.server_session import ServerLoop  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pre = 26
total = pre + n_timed + bench.SERVED_SPARE
w = bench.make_weights("cuda:0")
audio = bench.make_audio(S, total)
pcm16 = np.clip(np.round(audio * 32767.0), -32768, 32767).astype(np.int16).reshape(S, total, bench.CHUNK)
res = {}
for mode in ("strict", "continuous"):
    sb = bench.build_native(w, S, 10, False, total)
    loop = ServerLoop(StreamScheduler(sb, None, result_format="espnet"), finalize_update_iters=10 ** 9, max_partial_iters=10 ** 9,
                      continuous=(mode == "continuous"), min_replies=max(1, S // 16))
    sids = [loop.connect() for _ in range(S)]
    nxt = {sid: 0 for sid in sids}
    row = {sid: i for i, sid in enumerate(sids)}

    def send(sid):
        loop.submit(sid, pcm16[row[sid], nxt[sid]])
        nxt[sid] += 1

    for sid in sids:
        send(sid)
    n_rep, t0, target = 0, None, None
    while True:
        rep = loop.step()
        n_rep += sum(len(v) for v in rep.values())
        if t0 is None and n_rep >= S * pre:       # pre-roll done (untimed)
            t0, target = time.perf_counter(), n_rep + S * n_timed
        if target is not None and n_rep >= target:
            dt = time.perf_counter() - t0
            break
        for sid in rep:
            if nxt[sid] < total:
                send(sid)
    res[mode] = S * n_timed * 0.64 / dt
    print(f"{mode:10s}: {res[mode]:8.1f} audio-s/s through ServerLoop ({S} sessions, {dt / n_timed * 1e3:.2f} ms per {S} replies)", flush=True)
    while loop.pending():
        loop.step()
    sb.close()
print(f"continuous / strict = {res['continuous'] / res['strict']:.3f}")
