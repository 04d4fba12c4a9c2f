"""Soak test of queued chunks (sc_streams_set_queue_depth): S streams of random utterances (random chunk lengths, final
chunks, resets, new utterances on the same slot; a small PCM ring so that compaction happens with chunks in the queue),
submitted ahead up to the depth in random subsets and polled in random portions, against the one-call-at-a-time
protocol (sc_push) on a second batch.  Every reply must carry the same hypotheses with bit-identical scores.  Tiny dims, beam 5.
    gpurun -- 'python tools/soak_queue.py [depth=3] [streams=32] [utterances=6] [seed=0]'"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np  # noqa: E402

from speechcatcher_amd import synth  # noqa: E402
from test_engine_spec import make_batch  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n_utt = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bbd = bool(seed % 2)
kw = dict(n_streams=S, max_frames=500, max_tokens=420, pcm_capacity=1 << 16, max_chunk_samples=24000, strict_reference=bool(seed % 3 == 0))
ref = make_batch("TINY", 1234, "meanstd", 5, bbd, backend="native", **kw)
que = make_batch("TINY", 1234, "meanstd", 5, bbd, backend="native", **kw)
que.set_queue_depth(depth)
rng = np.random.default_rng(seed)
plan, expect = [], []
for s in range(S):
    chunks = []
    for u in range(n_utt):
        n_chunks = int(rng.integers(2, 12))
        lens = [int(rng.choice([300, 700, 1600, 4000, 8192, 10240, 16000, 24000])) for _ in range(n_chunks)]
        lens[-1] = max(lens[-1], 8192)          # (a final chunk of a few frames fails in the reference as well)
        audio = synth.synth_audio(100000 * seed + 100 * s + u, sum(lens))
        pos = 0
        for k, n in enumerate(lens):
            chunks.append((audio[pos:pos + n], k == n_chunks - 1))
            pos += n
    plan.append(chunks)
    rec = []
    for pcm, fin in chunks:
        out = ref.push([(s, pcm, fin)])
        rec.append((bool(out[s]), [(h["yseq"], h["xpos"], h["score"]) for h in ref.hypotheses(s)]))
        if fin:
            ref.reset(s)
    expect.append(rec)
sub, rep, wait_reset = [0] * S, [0] * S, [False] * S
n_ahead = n_replies = longest = 0
while any(rep[s] < len(plan[s]) for s in range(S)):
    if rng.random() < 0.1:
        que.set_encoder_batch(int(rng.choice([1, 4, S // 2, S])))
    for _ in range(depth):
        items = []
        for s in range(S):
            if sub[s] - rep[s] < depth and sub[s] < len(plan[s]) and not wait_reset[s] and rng.random() < 0.7:
                pcm, fin = plan[s][sub[s]]
                items.append((s, pcm, fin))
                n_ahead += sub[s] > rep[s]
                sub[s] += 1
                wait_reset[s] = fin
        if items:
            que.submit(items)
    if not que.outstanding:
        continue
    got = que.poll(int(rng.integers(1, max(2, S // 2))))
    hy = que.hypotheses_batch(list(got))
    for s, has in got.items():
        exp_has, exp_hyps = expect[s][rep[s]]
        assert not isinstance(has, Exception), (s, has)
        assert bool(has) == exp_has, (s, rep[s])
        if [(h["yseq"], h["xpos"]) for h in hy[s]] != [(e[0], e[1]) for e in exp_hyps]:
            print("MISMATCH stream", s, "reply", rep[s], "has", has, "exp_has", exp_has, "chunk len", len(plan[s][rep[s]][0]), "fin", plan[s][rep[s]][1],
                  "\n got", [(h["yseq"][:8], len(h["yseq"])) for h in hy[s]][:3], "\n exp", [(e[0][:8], len(e[0])) for e in exp_hyps][:3],
                  "\n sub/rep", sub[s], rep[s], "info", que.st[s].T_enc, que.st[s].L)
            raise SystemExit(1)
        for x, y in zip(hy[s], exp_hyps):   # (round 5: one summation order - the scores are the same bits)
            assert x["score"] == y[2], (s, rep[s], x["score"], y[2])
        longest = max([longest] + [len(h["yseq"]) for h in hy[s]])
        n_replies += 1
        fin = plan[s][rep[s]][1]
        rep[s] += 1
        if fin:
            que.reset(s)
            wait_reset[s] = False
print(f"ok: depth {depth}, {S} streams, {n_replies} replies equal the one-at-a-time protocol ({n_ahead} chunks were submitted behind an "
      f"outstanding one; longest hypothesis {longest} tokens; bbd {int(bbd)})")
