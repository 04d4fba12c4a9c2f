"""Server sessions on CPU with the spec backend: endpointing,
Vosk JSON replies and control messages of several clients batched on one engine
equal what each client would get from a private session of the oracle."""
import numpy as np
import pytest

from speechcatcher_amd import synth
from speechcatcher_amd.scheduler import ServerBusy, StreamScheduler
from speechcatcher_amd.server_session import Endpointer, ServerLoop, scale_server_pcm, vosk_result
from test_engine_spec import make_batch


def _ref_decisions(lens_stream, fui, max_iters):
    """speechcatcher_server.py:252-265 replayed on a given sequence of partial lengths."""
    hist, out = [], []
    for ln in lens_stream:
        n = len(hist)
        if n < fui:
            fin = False
        elif n > max_iters:
            fin, hist = True, []
        elif all(x == hist[-1] for x in hist[-fui:]):
            fin, hist = True, []
        else:
            fin = False
        out.append(fin)
        if not fin:
            hist.append(ln)
    return out


@pytest.mark.parametrize("fui,max_iters", [(1, 5), (3, 7), (6, 42)])
def test_endpointer_matches_reference_state_machine(fui, max_iters):
    rng = np.random.RandomState(fui)
    lens = list(np.repeat(rng.randint(0, 4, size=40), rng.randint(1, 6, size=40)))[:120]
    ep, got = Endpointer(fui, max_iters), []
    for ln in lens:
        fin = ep.decide()
        got.append(fin)
        if not fin:
            ep.observe(int(ln))
    assert got == _ref_decisions(lens, fui, max_iters)
    assert any(got) and not all(got)


def test_pcm_scaling_is_the_servers():
    x = np.array([-32768, -12345, -1, 0, 1, 777, 32767], dtype=np.int16)
    ref = (x.astype(np.float16) / 32767.0)
    assert ref.dtype == np.float16
    np.testing.assert_array_equal(scale_server_pcm(x), ref.astype(np.float32))


def _pcm16(stream_id, n):
    return np.clip(np.round(synth.synth_audio(stream_id, n) * 32767.0), -32768, 32767).astype(np.int16)


def run_sessions_vs_oracle(vosk, backend=None, device="cpu", continuous=False):
    from helpers import oracle_model
    from oracle.ref_port import RefPortStreaming, RefServerSession
    fui, mpi, beam = 2, 5, 3
    sb = make_batch("TINY", 1234, "meanstd", beam, True, n_streams=2, backend=backend, device=device, max_frames=400,
                    max_tokens=300, pcm_capacity=1 << 18)
    loop = ServerLoop(StreamScheduler(sb, None, result_format="espnet"), vosk_output_format=vosk,
                      finalize_update_iters=fui, max_partial_iters=mpi, continuous=continuous)
    model = oracle_model("TINY", 1234, "meanstd")
    chunk = 10240
    plans = {0: [_pcm16(5, chunk) for _ in range(9)], 1: [_pcm16(6, chunk) for _ in range(7)]}
    if vosk:   # control messages: config, then audio, eof at the end of client 1, reset in the middle of client 0
        plans[0] = ['{"config" : {"sample_rate" : 16000}}'] + plans[0][:4] + ['{"reset" : 1}'] + plans[0][4:]
        plans[1] = plans[1] + ['{"eof" : 1}']
        plans[1][2] = plans[1][2].tobytes()           # raw s16le bytes, as a Vosk client sends them
    sids = {c: loop.connect() for c in plans}
    with pytest.raises(ServerBusy):
        loop.connect()
    for c, msgs in plans.items():
        for m in msgs:
            loop.submit(sids[c], m)
    got = {c: [] for c in plans}
    inv = {v: k for k, v in sids.items()}
    while loop.pending():
        for sid, reps in loop.step().items():
            got[inv[sid]].extend(reps)
    for c, msgs in plans.items():
        ref = RefServerSession(RefPortStreaming(model, beam_size=beam, use_bbd=True, reference_reset_quirk=True), finalize_update_iters=fui,
                               max_partial_iters=mpi, vosk_output_format=vosk)
        want = [ref.reply(m) for m in msgs]
        assert len(got[c]) == len(want)
        n_final = 0
        for g, w in zip(got[c], want):
            if isinstance(w, dict) and "tokens" in w:      # final Vosk result: same tokens and text, real timestamps
                n_final += 1
                assert g["text"] == w["text"] and [x["word"] for x in g["result"]] == [t.replace("▁", " ") for t in w["tokens"]]
                starts = [x["start"] for x in g["result"]]
                assert starts == sorted(starts) and all(x["conf"] == 1.0 for x in g["result"])
            else:
                assert g == w
                n_final += isinstance(w, str) and w.endswith("\n")
        assert n_final >= 1, "the plan must exercise at least one finalised utterance"
    loop.disconnect(sids[0])
    assert loop.connect() is not None   # the slot is free again


@pytest.mark.parametrize("vosk", [False, True])
def test_sessions_batched_equal_private_oracle_sessions(vosk):
    run_sessions_vs_oracle(vosk)


def test_vosk_result_format():
    r = vosk_result(["▁hal", "lo", "▁welt"], [3, 7, 30])
    assert r["text"] == "hallo welt"
    assert [w["word"] for w in r["result"]] == [" hal", "lo", " welt"]
    assert r["result"][2]["start"] == pytest.approx(30 / 24.0)


def test_step_pacer_full_batch_or_deadline():
    """Pacing of batched steps: immediately when every connected client has a message waiting, otherwise when
    the oldest waiting message reaches the latency bound; the replies are those of the plain loop."""
    from speechcatcher_amd.server_session import StepPacer
    sb = make_batch("TINY", 1234, "meanstd", 3, True, n_streams=3, max_frames=400, max_tokens=300,
                    pcm_capacity=1 << 18)
    loop = ServerLoop(StreamScheduler(sb, None, result_format="espnet"))
    now = [100.0]
    pacer = StepPacer(loop, max_wait_s=0.05, clock=lambda: now[0])
    a, b = loop.connect(), loop.connect()
    assert not pacer.due() and pacer.poll() is None           # nothing waiting
    pacer.submit(a, _pcm16(5, 10240))
    assert not pacer.due()                                    # b has sent nothing and a has not waited long
    now[0] += 0.049
    assert pacer.poll() is None
    now[0] += 0.002                                           # a's chunk reaches the latency bound
    rep = pacer.poll()
    assert rep is not None and set(rep) == {a}
    assert not pacer.due()
    pacer.submit(a, _pcm16(5, 10240))
    pacer.submit(a, _pcm16(5, 10240))                         # a runs ahead: two chunks queued
    now[0] += 0.01
    pacer.submit(b, _pcm16(6, 10240))
    assert pacer.due()                                        # full batch: every client has a message
    rep = pacer.poll()
    assert set(rep) == {a, b}
    assert not pacer.due()                                    # a's second chunk waits; b has nothing
    now[0] += 0.045                                           # ... but it has been waiting since before the step
    assert pacer.due()
    rep = pacer.poll()
    assert set(rep) == {a} and not loop.pending()
    loop.disconnect(b)
    pacer.submit(a, _pcm16(5, 10240))
    assert pacer.due()                                        # the only client left: a full batch of one
    assert set(pacer.poll()) == {a}


def test_failing_client_does_not_wedge_the_others():
    """Client B sends 50 samples, then eof: the final chunk has < 7 feature frames and the reference
    raises inside Conv2d (A3) - for that client only.  Client A's chunks of the same batched steps are
    decoded as if B had not been there, and A is not left 'in flight'."""
    sb = make_batch("TINY", 1234, "meanstd", 3, True, n_streams=2, max_frames=400, max_tokens=300,
                    pcm_capacity=1 << 18)
    loop = ServerLoop(StreamScheduler(sb, None, result_format="espnet"), vosk_output_format=True)
    a, b = loop.connect(), loop.connect()
    chunks = [_pcm16(5, 10240) for _ in range(5)]
    for c in chunks:
        loop.submit(a, c)
    loop.submit(b, _pcm16(6, 50))
    loop.submit(b, '{"eof" : 1}')
    got = {a: [], b: []}
    while loop.pending():
        for sid, reps in loop.step().items():
            got[sid].extend(reps)
    assert any(isinstance(r, RuntimeError) for r in got[b])
    assert loop.sessions[a].in_flight is None and loop.sessions[b].in_flight is None
    # A alone on a fresh loop gives the same replies
    sb2 = make_batch("TINY", 1234, "meanstd", 3, True, n_streams=1, max_frames=400, max_tokens=300,
                     pcm_capacity=1 << 18)
    solo = ServerLoop(StreamScheduler(sb2, None, result_format="espnet"), vosk_output_format=True)
    a2 = solo.connect()
    for c in chunks:
        solo.submit(a2, c)
    want = []
    while solo.pending():
        for _sid, reps in solo.step().items():
            want.extend(reps)
    assert got[a] == want and len(want) == 5
    # B's slot is usable again
    loop.submit(b, _pcm16(6, 10240))
    assert not isinstance(loop.step()[b][0], Exception)


def run_strict_server(backend=None, device="cpu"):
    """ServerLoop(strict_reference=True): like the reference server, nothing resets the model after a
    finalised utterance (speechcatcher_server.py:270 - the native decoder keeps decoding into the
    finished stream).  Replies equal a private oracle session that never resets either."""
    from helpers import oracle_model
    from oracle.ref_port import RefPortStreaming, RefServerSession
    fui, mpi, beam = 2, 5, 3
    sb = make_batch("TINY", 1234, "meanstd", beam, True, n_streams=1, backend=backend, device=device, max_frames=400,
                    max_tokens=300, pcm_capacity=1 << 18)
    loop = ServerLoop(StreamScheduler(sb, None, result_format="espnet"), finalize_update_iters=fui,
                      max_partial_iters=mpi, strict_reference=True)
    sid = loop.connect()
    msgs = [_pcm16(5, 10240) for _ in range(9)]
    for m in msgs:
        loop.submit(sid, m)
    got = []
    while loop.pending():
        for _sid, reps in loop.step().items():
            got.extend(reps)
    ref = RefServerSession(RefPortStreaming(oracle_model("TINY", 1234, "meanstd"), beam_size=beam, use_bbd=True),
                           finalize_update_iters=fui, max_partial_iters=mpi, reset_after_final=False)
    want = [ref.reply(m) for m in msgs]
    assert got == want
    assert any(isinstance(w, str) and w.endswith(chr(10)) for w in want)


def test_strict_reference_server_never_resets():
    run_strict_server()
