"""The C++ engine (the thing bench.py times) at BASELINE.json's sizes: configs[2] (XL dims, 128 streams, beam 10),
the per-GPU share of configs[4] (256 streams, fp16 feed-forward + fp16 K|V caches, graphs on, token positions),
a stand-in for configs[3] (`_l` dims are not available offline: config.L_LIKE, 128 streams) and the 2-D feature
input of the reference API (speech2text_streaming.py:438-449) - through the stream-level C ABI."""
import ctypes as C

import numpy as np
import pytest

from speechcatcher_amd import synth
from test_engine_spec import make_batch

pytestmark = pytest.mark.gpu

CHUNK = 10240


def _same(a, b, tol, what):
    assert len(a) == len(b) > 0, what
    for x, y in zip(a, b):
        assert x["yseq"] == y["yseq"] and x["xpos"] == y["xpos"], what
        assert abs(x["score"] - y["score"]) <= tol * max(1.0, abs(y["score"]) * 1e-2), (what, x["score"], y["score"])


def _feed(sb, audio, n_steps, rows=None):
    rows = list(range(sb.S)) if rows is None else rows
    ids = np.arange(len(rows), dtype=np.int32)
    for k in range(n_steps):
        st = sb.push_block(ids, np.ascontiguousarray(audio[rows, k * CHUNK:(k + 1) * CHUNK]))
        assert (st >= 0).all()


def test_xl_128_streams_native_equals_python_engine_solo_runs_and_continuous_batching():
    """BASELINE configs[2] on the product engine: 128 streams x 8 chunk steps (5 decode blocks each).  Every stream's
    hypotheses (ids and positions exact, scores <= 1e-3) equal those of the Python engine over the same kernels, of
    solo runs, of its duplicate elsewhere in the batch, and of the same streams served by continuous batching; the
    encoder ran beside the decode loop and (nearly) every compaction bucket was used."""
    from speechcatcher_amd.hip_backend import HipBackend
    S, n, beam = 128, 8, 10
    audio = np.stack([synth.synth_audio(500 + (s if s < 120 else s - 120), CHUNK * n) for s in range(S)])   # 120..127 = 0..7
    kw = dict(n_streams=S, max_frames=200, max_tokens=160, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
    nat = make_batch("XL", 1234, "meanstd", beam, False, backend="native", **kw)
    sec, it = (C.c_double * 17)(), (C.c_long * 17)()
    _feed(nat, audio, n)
    nat.lib.sc_streams_bucket_times(nat.handle, sec, it)
    assert sum(1 for k in range(17) if it[k]) >= 12, list(it)          # the ragged step loop went through the buckets
    assert nat.stats["dec_blocks"] == S * (n - 3), nat.stats
    got = nat.hypotheses_batch(list(range(S)))
    assert all(len(got[s]) == beam and len(got[s][0]["yseq"]) > 10 for s in range(S))
    for s in range(8):
        _same(got[120 + s], got[s], 1e-6, f"duplicate of stream {s}")
    # continuous batching: same streams, every reply followed by that stream's next chunk
    cont = make_batch("XL", 1234, "meanstd", beam, False, backend="native", **kw)
    nxt = np.zeros(S, np.int64)
    a3 = audio.reshape(S, n, CHUNK)
    cont.submit_block(np.arange(S, dtype=np.int32), np.ascontiguousarray(a3[:, 0]))
    nxt += 1
    while cont.outstanding:
        done, st = cont.poll_ids(16)
        assert (st >= 0).all()
        again = done[nxt[done] < n]
        if len(again):
            cont.submit_block(again, a3[again, nxt[again]])
            nxt[again] += 1
    assert cont.stats["dec_blocks"] == nat.stats["dec_blocks"] and cont.stats["dec_steps"] < nat.stats["dec_steps"]
    gc_ = cont.hypotheses_batch(list(range(S)))
    for s in range(S):
        _same(gc_[s], got[s], 1e-3, f"continuous batching, stream {s}")
    cont.close()
    # solo runs of three streams
    for s in (0, 50, 127):
        solo = make_batch("XL", 1234, "meanstd", beam, False, backend="native", **dict(kw, n_streams=1))
        _feed(solo, audio, n, rows=[s])
        _same(solo.hypotheses(0), got[s], 1e-3, f"solo run of stream {s}")
        solo.close()
    nat.close()
    # three streams against the oracle run solo on the same audio (hypothesis sets equal, best exact, scores 1e-3)
    from helpers import oracle_calls_parallel
    seeds = {s: 500 + (s if s < 120 else s - 120) for s in (5, 70, 123)}
    ora = oracle_calls_parallel("XL", list(seeds.values()), CHUNK * n, CHUNK, beam, False)
    for s, sd in seeds.items():
        ref = ora[sd][-1]
        assert sorted(tuple(h["yseq"]) for h in got[s]) == sorted(tuple(y) for y in ref["yseq"]), s
        by = {tuple(y): (x, sc) for y, x, sc in zip(ref["yseq"], ref["xpos"], ref["score"])}
        for h in got[s]:
            assert h["xpos"] == by[tuple(h["yseq"])][0] and abs(h["score"] - by[tuple(h["yseq"])][1]) <= 1e-3, s
        assert got[s][0]["yseq"] == ref["yseq"][0], s
    # the Python engine over the same kernels
    pye = make_batch("XL", 1234, "meanstd", beam, False, backend=HipBackend("cuda:0"), device="cuda:0", **kw)
    for k in range(n):
        pye.push([(s, audio[s, k * CHUNK:(k + 1) * CHUNK], False) for s in range(S)])
    for s in range(S):
        _same(pye.hypotheses(s), got[s], 1e-3, f"python engine, stream {s}")


def _serve_continuous(sb, a3, poll, track=()):
    """bench.py's serving loop: sc_submit one chunk per stream, sc_poll(poll), every answered stream gets its next chunk.
    Returns {tracked stream: [(hypotheses, process_idx, T) after each of its replies]}."""
    S, n = a3.shape[0], a3.shape[1]
    nxt = np.zeros(S, np.int64)
    seen = {s: [] for s in track}
    sb.submit_block(np.arange(S, dtype=np.int32), np.ascontiguousarray(a3[:, 0]))
    nxt += 1
    while sb.outstanding:
        done, st = sb.poll_ids(min(poll, sb.outstanding))
        assert (st >= 0).all()
        for s in done:
            if int(s) in seen:
                seen[int(s)].append((sb.hypotheses(int(s)), sb.st[int(s)].process_idx, sb.st[int(s)].T_enc))
        again = done[nxt[done] < n]
        if len(again):
            sb.submit_block(again, a3[again, nxt[again]])
            nxt[again] += 1
    return seen


def test_xl_128_streams_continuous_batching_in_the_headline_regime_vs_oracle():
    """The regime and the mode bench.py times: XL dims, 128 streams, beam 10, no block-boundary
    detection, CONTINUOUS batching (sc_submit / sc_poll(16)) on the C++ engine, 60 chunks per stream so that the streams
    reach T >= 700 encoder frames and >= 330 tokens with full compaction buckets (large-bucket kernels, multi-chunk K/V
    walks).  The decisive streams of 24 candidates are compared CALL BY CALL with the oracle run solo on the same audio: token ids / positions /
    process_idx exact after every reply, cumulative scores within 1e-3 (north star), drift per decode step <= 1e-4; all
    128 streams well formed.
    Which streams are compared is decided by the ORACLE alone, before anything is compared: 24 candidates are run through it
    (eight until round 5), and a stream is compared over its whole run if the oracle never cut its beam by
    less than 2.5e-5 (`margins` of ref_port.py: the score gap between the last survivor and the first loser of a step; 2.5e-5 =
    twice the largest per-step drift between engine and oracle ever measured, 1.2e-5) - at least TWELVE must be such streams (the
    oracle's smallest cuts of the 24 range from 3.1e-6 to 2.1e-4; 13 are >= 2.5e-5), and none of them may leave the oracle's path (no
    escape: the engine is bit-reproducible since round 5, test_serving_is_bit_reproducible, so this run is the same run
    every time).  A candidate whose oracle run contains a cut below 2.5e-5 is compared up to that call only: two correct fp32
    implementations that sum in different orders - the reference on the CPU and this engine - may decide such a cut
    differently.
    The same run in the split-precision form and with fp16 K|V storage against the f32 engine."""
    import json
    import os
    from helpers import oracle_calls_parallel
    from test_engine_spec import check_hyps
    S, n, beam, poll = 128, 60, 10, 16
    # 24 candidates (round 6; eight until round 5): the oracle's smallest beam cut of each, computed once on the CPU box -
    # 13 of them never cut by less than 2.5e-5 (7, 17, 29, 50, 58, 64, 77, 83, 90, 96, 101, 111, 118)
    tracked = (3, 29, 64, 90, 111, 125, 7, 50, 1, 12, 17, 23, 36, 41, 47, 58, 71, 77, 83, 96, 101, 107, 118, 122)
    audio = np.stack([synth.synth_audio(4000 + s, CHUNK * n) for s in range(S)])
    a3 = audio.reshape(S, n, CHUNK)
    kw = dict(n_streams=S, max_frames=16 * n + 80, max_tokens=640, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
    sb = make_batch("XL", 1234, "meanstd", beam, False, backend="native", **kw)
    seen = _serve_continuous(sb, a3, poll, tracked)
    sec, it = (C.c_double * 17)(), (C.c_long * 17)()
    sb.lib.sc_streams_bucket_times(sb.handle, sec, it)
    assert sum(it[12:]) > 0.5 * sum(it), list(it)                    # most decode iterations ran with >= 3/4 of the streams
    base = sb.hypotheses_arrays(list(range(S)))
    T = [sb.st[s].T_enc for s in range(S)]
    assert min(T) >= 700 and base["lens"][:, 0].min() >= 330, (min(T), int(base["lens"][:, 0].min()))
    assert (base["n_hyps"] == beam).all()
    for s in range(S):                                                # well formed: sos first, positions non-decreasing
        L = base["lens"][s, 0]
        assert base["ids"][s, 0, 0] == 1023 and (np.diff(base["xpos"][s, 0, :L]) >= 0).all() and base["xpos"][s, 0, L - 1] < T[s]
    sb.close()
    # (one intra-op thread per oracle process: 24 processes x 8 threads oversubscribed the box - 21 minutes; ~110 s this way)
    ora = oracle_calls_parallel("XL", [4000 + s for s in tracked], CHUNK * n, CHUNK, beam, False, threads=1)
    report = {}
    decisive = []
    for s in tracked:
        calls = ora[4000 + s]
        assert len(seen[s]) == len(calls) == n
        # the oracle's own verdict on its run: the first call (if any) where it cut the beam by less than 2.5e-5
        coin_flip = next((k for k, ref in enumerate(calls) if ref["min_margin"] < 2.5e-5), None)
        if coin_flip is None:
            decisive.append(s)
        worst, worst_step, prev_diff, prev_pidx, compared = 0.0, 0.0, 0.0, 0, 0
        for k, ((hyps, pidx, t_enc), ref) in enumerate(zip(seen[s], calls)):
            assert t_enc == ref["T"], (s, k)
            if coin_flip is not None and k >= coin_flip:
                break
            if not ref["yseq"] or pidx == prev_pidx:
                continue
            check_hyps(hyps, pidx, ref, 1e-3)
            by = {tuple(y): sc for y, sc in zip(ref["yseq"], ref["score"])}
            diff = max(abs(h["score"] - by[tuple(h["yseq"])]) for h in hyps)
            worst = max(worst, diff)
            worst_step = max(worst_step, abs(diff - prev_diff) / max(pidx - prev_pidx, 1))
            prev_diff, prev_pidx, compared = diff, pidx, compared + 1
        assert worst_step <= 1e-4 and (coin_flip is not None or compared >= n * 3 // 4), (s, compared, worst_step)   # (calls without a decode step are skipped)
        report[s] = {"calls_compared": compared, "max_abs_total_score_diff": worst, "max_drift_per_decode_step": worst_step,
                     "T_end": seen[s][-1][2], "tokens_end": len(seen[s][-1][0][0]["yseq"]),
                     "oracle_min_beam_cut_margin": min(ref["min_margin"] for ref in calls),
                     "compared_up_to_call": n if coin_flip is None else coin_flip}
    assert len(decisive) >= 12, (decisive, {s: report[s]["oracle_min_beam_cut_margin"] for s in tracked})
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "r06_xl_continuous_128_parity.json"), "w") as f:
        json.dump({"streams": S, "chunks": n, "poll": poll, "T_min_max": [min(T), max(T)],
                   "tokens_min_max": [int(base["lens"][:, 0].min()), int(base["lens"][:, 0].max())],
                   "bucket_iterations": list(it), "tracked": report}, f)

    def hyp(o, s, j):
        n_ = o["lens"][s, j]
        return tuple(o["ids"][s, j, :n_].tolist()), tuple(o["xpos"][s, j, :n_].tolist())

    # the opt-in forms against the f32 engine, same audio, same serving loop: the best hypothesis of (nearly) every stream
    moved = {}
    for name, extra in (("split16", dict(ffn_dtype="split16", proj_dtype="split16")), ("kv_fp16", dict(kv_dtype="float16"))):
        alt = make_batch("XL", 1234, "meanstd", beam, False, backend="native", **dict(kw, **extra))
        _serve_continuous(alt, a3, poll)
        o = alt.hypotheses_arrays(list(range(S)))
        alt.close()
        diff = [s for s in range(S) if hyp(o, s, 0) != hyp(base, s, 0)]
        same = [s for s in range(S) if s not in diff]
        moved[name] = {"best_hypothesis_moved": diff, "max_score_diff_of_the_others": float(np.abs(o["score"][same, 0] - base["score"][same, 0]).max())}
        print(f"{name}: best hypothesis moved on {len(diff)} of {S} streams {diff}; scores of the others within "
              f"{moved[name]['max_score_diff_of_the_others']:.2e}")
        # 60 chunks x ~9 steps of a random-weight model without boundary detection: a near-tie decided the other way
        # moves a stream onto another path for good; bar: <= 3 % of the streams, scores of the rest 1e-3
        assert len(diff) <= max(1, S * 3 // 100), (name, diff)
        assert moved[name]["max_score_diff_of_the_others"] <= (1e-3 if name == "split16" else 2e-2), (name, moved[name])
    with open(os.path.join("gpurun_out", "r06_xl_continuous_128_forms.json"), "w") as f:
        json.dump(moved, f)


def test_serving_is_bit_reproducible():
    """the same audio gives the same BITS whoever else is on the GPU.  XL dims, 128 streams, 44 chunks
    each (T = 700 encoder frames, ~330 tokens: every kernel form of the decoder layers, both CTC scan forms, encoder groups
    of every size) served three ways - continuous batching with sc_poll groups of 8 and of 32 (other buckets, other encoder
    groups, other kernel forms per stream and step) and strict lock-step - and three of the streams decoded SOLO in a batch of
    one.  Token ids, positions and all three float64 totals of every hypothesis must be IDENTICAL: every sum of the path
    is evaluated in one order (csrc/common.h: canonical summation).  Reference: the tie rule of the search is exact
    comparison of float64 totals (hypothesis.py:132-142, beam_search.py:721-758) - a last-bit difference can move a token."""
    S, n, beam = 128, 44, 10
    audio = np.stack([synth.synth_audio(4000 + s, CHUNK * n) for s in range(S)])
    a3 = audio.reshape(S, n, CHUNK)
    kw = dict(max_frames=16 * n + 80, max_tokens=640, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)

    def served(poll):
        sb = make_batch("XL", 1234, "meanstd", beam, False, backend="native", n_streams=S, **kw)
        if poll:
            _serve_continuous(sb, a3, poll)
        else:
            _feed(sb, audio, n)
        o = sb.hypotheses_arrays(list(range(S)))
        T = [sb.st[s].T_enc for s in range(S)]
        sb.close()
        return o, T

    def same_bits(a, b, what, rows_a=None, rows_b=None):
        ra = slice(None) if rows_a is None else rows_a
        rb = slice(None) if rows_b is None else rows_b
        for key in ("n_hyps", "lens", "ids", "xpos"):
            assert np.array_equal(a[key][ra], b[key][rb]), (what, key)
        for key in ("score", "score_dec", "score_ctc"):      # float64 totals: bit for bit
            x, y = a[key][ra], b[key][rb]
            bad = np.nonzero(x.view(np.int64) != y.view(np.int64))
            assert len(bad[0]) == 0, (what, key, len(bad[0]), float(np.abs(x - y).max()))

    p8, T = served(8)
    assert min(T) >= 650 and p8["lens"][:, 0].min() >= 250, (min(T), int(p8["lens"][:, 0].min()))
    p32, _ = served(32)
    same_bits(p8, p32, "sc_poll groups of 8 against groups of 32")
    lock, _ = served(0)
    same_bits(p8, lock, "continuous batching against strict lock-step")
    for s in (3, 64, 125):
        solo = make_batch("XL", 1234, "meanstd", beam, False, backend="native", n_streams=1, **kw)
        _feed(solo, audio, n, rows=[s])
        o = solo.hypotheses_arrays([0])
        solo.close()
        Lm = min(o["ids"].shape[2], p8["ids"].shape[2])
        for key in ("ids", "xpos"):
            o[key] = o[key][:, :, :Lm]
        ref = {k: (v[:, :, :Lm] if k in ("ids", "xpos") else v) for k, v in p8.items()}
        same_bits(ref, o, f"stream {s} in the batch of 128 against its solo run", rows_a=[s], rows_b=[0])


@pytest.mark.parametrize("S,n,kv", [(72, 14, "float32"), (128, 30, "float32"), (72, 14, "float16")])
def test_stream_resident_layers_give_the_bits_of_the_head_parallel_forms(monkeypatch, S, n, kv):
    """Round 6: the stream-resident form of the decoder layers (csrc/decoder_stream.hip: one workgroup per stream, both
    attentions of a layer in one launch, two launches per layer) against the head-parallel launches it replaces at large
    buckets (four heads per workgroup, three launches per layer): the same audio in strict lock-step must give IDENTICAL
    token ids, positions and float64 totals - every sum follows the canonical order of csrc/common.h.  n = 30 chunks reach
    T = 480 frames / ~220 tokens (several tiles per attention slot, ragged buckets at the end of every chunk step).
    Reference semantics: decoder_layer.py:80-132, multi_head_attention.py:63-133."""
    beam = 10
    audio = np.stack([synth.synth_audio(7000 + s, CHUNK * n) for s in range(S)])
    kw = dict(max_frames=16 * n + 80, max_tokens=480, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK, kv_dtype=kv)   # (fp16 K|V storage: the KVH variants)

    def run(form, split="0"):
        monkeypatch.setenv("SC_DEC_STREAM", form)
        monkeypatch.setenv("SC_DEC_FFN_SPLIT", split)
        sb = make_batch("XL", 1234, "meanstd", beam, False, backend="native", n_streams=S, **kw)
        _feed(sb, audio, n)
        o = sb.hypotheses_arrays(list(range(S)))
        sb.close()
        return o

    a, b = run("1"), run("0")
    assert a["lens"][:, 0].min() >= (60 if n < 20 else 150), int(a["lens"][:, 0].min())
    # ... and the four-head form with the head partials summed in a launch of their own (sc_dec_layer_reduce_ln, an A/B hook)
    for other, what in ((b, "four heads per workgroup"), (run("0", "1"), "four heads per workgroup, partials summed once per row")):
        for key in ("n_hyps", "lens", "ids", "xpos"):
            assert np.array_equal(a[key], other[key]), (what, key)
        for key in ("score", "score_dec", "score_ctc"):
            x, y = a[key], other[key]
            bad = np.nonzero(x.view(np.int64) != y.view(np.int64))
            assert len(bad[0]) == 0, (what, key, len(bad[0]), float(np.abs(x - y).max()))


def test_fp16_mode_error_per_decode_step_is_bounded():
    """the fp16 mode (BASELINE configs[4]; the reference has no fp16 run to compare with, quirk A9,
    speechcatcher.py:205-212) gets a NUMBER.  tools/fp16_step_error.py: the fp32 engine drives the search, a shadow batch in the
    fp16 mode is reset to the fp32 state before EVERY decode step (hypotheses, scores, CTC state, K|V rows converted to its
    storage type) and takes the same step - the difference of what the step produced is one step's worth of fp16 arithmetic on
    the same prefix, no drift, no path divergence.  Measured (8 streams x 12 chunks, profiles/r06_fp16_step_error.txt):
        fp16 K|V only            max |d log-prob| 2.4e-5   max |d fused score| 1.2e-5
        + fp16 feed-forward                        7.2e-5                       4.1e-5
        + fp16 decoder projections / partials      1.0e-4                       6.1e-5
        the whole mode                             1.2e-4                       6.7e-5
    (fp32 engine against the oracle: 1.2e-5 per step).  Asserted here on 4 streams x 8 chunks with a factor of 4 in hand."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fp16_step_error
    w = fp16_step_error.run_mode("float16", S=4, n=8)
    assert w["steps"] >= 40
    assert w["dlogp"] <= 5e-4 and w["dscore"] <= 3e-4, w
    assert w["cand_tok_mismatch"] <= w["cand_total"] // 500, w      # (near-ties among the W x W candidates: 0.05 % measured)
    k = fp16_step_error.run_mode("kv16", S=4, n=8)
    assert k["dlogp"] <= 1e-4 and k["dscore"] <= 6e-5, k


def test_xl_256_streams_fp16_mode_keeps_the_fp32_token_ids():
    """(also: the split-precision form `split16` on the same 256 streams - identical beams, see the end.)
    Per-GPU share of BASELINE configs[4]: 256 streams, fp16 feed-forward + encoder attention-projection weights / MFMA
    inputs and fp16 K|V caches
    (hipGraph replay on, token positions read back) against the fp32 engine on the same audio: the best hypothesis
    keeps its token ids AND positions for at least 97 % of the streams, the whole beam for 95 % (fp16 rounding of the
    feed-forward reorders hypotheses whose fp32 scores are closer than its error; no fp16 run of the reference's native
    decoder exists to compare with: speechcatcher.py:205-210 disables it).  The counts of the run, and for every moved
    stream where the fp16 engine's best hypothesis sits in the fp32 beam and how many leading tokens the two best
    hypotheses share, are written to gpurun_out/r05_fp16_mode_256.json (round 5: 6 / 8 of 256; one move is a reordering
    inside the final beam 9e-6 apart, five are beam cuts at an earlier step - the two best hypotheses share 35-95 % of their leading tokens)."""
    S, n, beam = 256, 7, 10
    audio = np.stack([synth.synth_audio(900 + s, CHUNK * n) for s in range(S)])
    kw = dict(n_streams=S, max_frames=200, max_tokens=160, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
    out = {}
    for mode in ("float32", "float16", "split16"):
        sb = make_batch("XL", 1234, "meanstd", beam, False, backend="native", ffn_dtype=mode, proj_dtype=mode,
                        dec_dtype="float16" if mode == "float16" else "float32",      # (the whole fp16 mode, decoder included)
                        kv_dtype="float32" if mode == "split16" else mode, **kw)
        _feed(sb, audio, n)
        out[mode] = sb.hypotheses_arrays(list(range(S)))
        sb.close()
    a, b = out["float32"], out["float16"]
    assert (a["n_hyps"] == beam).all() and (b["n_hyps"] == beam).all() and a["lens"][:, 0].min() > 10

    def hyp(o, s, j):
        n_ = o["lens"][s, j]
        return tuple(o["ids"][s, j, :n_].tolist()), tuple(o["xpos"][s, j, :n_].tolist())

    # measured (tools/fp16_mode_stats.py): fp16 K|V caches alone change no hypothesis of the 256 streams (best scores
    # within 7e-5); the fp16 feed-forward moves the best hypothesis of 2 of 256 streams (one exact tie, one stream onto
    # another path), the fp16 encoder attention projections of 3, all three together of 4 (6 beams)
    best_differs = [s for s in range(S) if hyp(a, s, 0) != hyp(b, s, 0)]
    beam_differs = [s for s in range(S) if {hyp(a, s, j) for j in range(beam)} != {hyp(b, s, j) for j in range(beam)}]
    print(f"fp16 mode: best hypothesis moved on {len(best_differs)} of {S} streams, beam set on {len(beam_differs)}")
    # where the fp16 engine's best hypothesis sits in the fp32 beam, and how far below the fp32 best
    margins = {}
    for s in best_differs:
        at = [j for j in range(beam) if hyp(a, s, j) == hyp(b, s, 0)]
        ia, ib = hyp(a, s, 0)[0], hyp(b, s, 0)[0]
        common = next((k for k in range(min(len(ia), len(ib))) if ia[k] != ib[k]), min(len(ia), len(ib)))
        margins[s] = {"fp32_rank": at[0] if at else -1, "fp32_margin": float(a["score"][s, 0] - a["score"][s, at[0]]) if at else None,
                      "common_prefix": common, "len_fp32": len(ia), "len_fp16": len(ib)}
    print("  the hypothesis the fp16 engine ranks first, in the fp32 run:", margins)
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r05_fp16_mode_256.json", "w") as fh:
        json.dump({"streams": S, "chunks": n, "best_moved": best_differs, "beam_moved": beam_differs,
                   "fp16_best_in_the_fp32_run": {str(k): v for k, v in margins.items()}}, fh)
    # rounds 2-3 measured 4 / 6 streams, round 4 (other fp32 summation order of the partial products, other near-ties)
    # 6 / 9, round 5 (canonical order) 6 / 8: a statistic of random-weight near-ties, 2.3 % / 3.1 % - bar 3 % / 5 %
    # (2 % / 4 % = 5 / 10 streams would sit ON the measured count: the bar has to clear the statistic it bounds)
    assert len(best_differs) <= S * 3 // 100 and len(beam_differs) <= S // 20, (best_differs, beam_differs)
    same = [s for s in range(S) if s not in best_differs]
    assert np.abs(a["score"][same, 0] - b["score"][same, 0]).max() < 0.5
    # the split-precision form (fp32 operands as fp16 hi + lo pairs) is held to the fp32 engine itself:
    # EVERY hypothesis of EVERY beam - ids, positions, order - and the scores to fp32 rounding level
    c = out["split16"]
    assert (c["n_hyps"] == beam).all()
    for s in range(S):
        for j in range(beam):
            assert hyp(a, s, j) == hyp(c, s, j), (s, j)
    assert np.abs(a["score"] - c["score"]).max() < 2e-4


def test_l_like_dims_128_streams_equal_solo_oracle_runs():
    """Stand-in for BASELINE configs[3] (`_l` dims assumed: config.L_LIKE = 256 / 4 heads of 64 / 18 + 8 blocks) at its
    per-GPU batch: 128 streams, beam 10; three of them against the oracle run alone on the same audio."""
    from helpers import oracle_model, run_oracle_stream
    S, n, beam = 128, 7, 10
    audio = np.stack([synth.synth_audio(700 + s, CHUNK * n) for s in range(S)])
    sb = make_batch("L_LIKE", 1234, "meanstd", beam, False, n_streams=S, backend="native", max_frames=200, max_tokens=160,
                    pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
    for k in range(n):
        fin = np.full(S, 1 if k == n - 1 else 0, np.uint8)
        st = sb.push_block(np.arange(S, dtype=np.int32), np.ascontiguousarray(audio[:, k * CHUNK:(k + 1) * CHUNK]), fin)
        assert (st >= 0).all()
    got = sb.hypotheses_batch(list(range(S)))
    model = oracle_model("L_LIKE", 1234, "meanstd")
    for s in (3, 64, 125):
        ora, _, _, _ = run_oracle_stream(model, audio[s], CHUNK, beam, False)
        ref = ora.running_hyps
        assert sorted(tuple(h["yseq"]) for h in got[s]) == sorted(tuple(h.yseq) for h in ref), s
        assert got[s][0]["yseq"] == list(ref[0].yseq) and got[s][0]["xpos"] == list(ref[0].xpos), s
        assert abs(got[s][0]["score"] - ref[0].score) < 1e-3, (s, got[s][0]["score"], ref[0].score)   # (2e-3 until round 5)


@pytest.mark.parametrize("cfg_name", ["TINY", "XL"])
def test_feature_input_on_the_native_engine_matches_oracle(cfg_name):
    """2-D (T, 80) feature matrices instead of PCM (speech2text_streaming.py:438-449: normalised, then used) through
    sc_push_features, two streams with different splits in one batch, against the oracle fed the same features."""
    import torch
    from helpers import oracle_model
    from oracle.ref_port import RefPortStreaming
    model = oracle_model(cfg_name, 1234, "meanstd")
    g = torch.Generator().manual_seed(5)
    feats = [(torch.randn(150, 80, generator=g) * 2.0 - 8.0).numpy().astype(np.float32) for _ in range(2)]
    splits = [((0, 70), (70, 150)), ((0, 90), (90, 150))]
    sb = make_batch(cfg_name, 1234, "meanstd", 5, False, n_streams=3, backend="native", max_frames=128, max_tokens=200,
                    pcm_capacity=1 << 14, max_chunk_samples=32768)
    oras = [RefPortStreaming(model, beam_size=5) for _ in range(2)]
    for part in range(2):
        items = []
        for i in range(2):
            a, b = splits[i][part]
            fin = b == 150
            oras[i](torch.from_numpy(feats[i][a:b]), is_final=fin, finalize_all=fin)
            norm = ((feats[i][a:b] - model.mean) / model.std).astype(np.float32)
            items.append((2 * i, norm, fin))                      # streams 0 and 2; stream 1 stays idle
        out = sb.push_features(items)
        assert all(out[s] is True for s, _, _ in items)
    for i in range(2):
        got, ref = sb.hypotheses(2 * i), oras[i].running_hyps
        assert [h["yseq"] for h in got] == [list(h.yseq) for h in ref], i
        assert [h["xpos"] for h in got] == [list(h.xpos) for h in ref], i
        np.testing.assert_allclose([h["score"] for h in got], [h.score for h in ref], atol=1e-3, rtol=0)
    assert sb.hypotheses(1) == [] or len(sb.hypotheses(1)[0]["yseq"]) == 1
    # a feature matrix that does not fit fails that stream only
    from speechcatcher_amd.engine import EngineError
    big = np.zeros((400, 80), np.float32)
    out = sb.push_features([(1, big, False)], isolate_faults=True)
    assert isinstance(out[1], EngineError)
