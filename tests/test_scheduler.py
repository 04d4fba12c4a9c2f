"""Multi-session scheduler on CPU with the spec backend:
sessions with different chunk sizes and arrival patterns, batched together,
get exactly the per-call results the reference returned for each stream alone."""
import pytest

from conftest import load_case
from speechcatcher_amd import synth
from speechcatcher_amd.scheduler import ServerBusy, StreamScheduler
from test_engine_spec import make_batch


def run_sessions_different_chunking(backend=None, device="cpu"):
    cases = {"a": ("tiny_c8192_b10_bbd0", 8192), "b": ("tiny_c10240_b10_bbd0", 10240),
             "c": ("tiny_c25600_b10_bbd0", 25600)}
    sb = make_batch("TINY", 1234, "meanstd", 10, False, n_streams=3, backend=backend, device=device, max_frames=256,
                    max_tokens=160, pcm_capacity=1 << 18)
    sch = StreamScheduler(sb)
    js = {k: load_case(v[0])[0] for k, v in cases.items()}
    audio = synth.synth_audio(0, js["a"]["meta"]["n_samples"])
    sid = {k: sch.open() for k in cases}
    with pytest.raises(ServerBusy):
        sch.open()
    got = {k: [] for k in cases}
    # session c starts two rounds late; all chunks of a session are fed up front
    for k, (_, chunk) in cases.items():
        for pos in range(0, len(audio), chunk):
            end = min(pos + chunk, len(audio))
            sch.feed(sid[k], audio[pos:end], is_final=end >= len(audio), finalize_all=end >= len(audio))
    inv = {v: k for k, v in sid.items()}
    while sch.pending():
        for s, res in sch.step().items():
            got[inv[s]].append(res)
    for k in cases:
        ref_calls = js[k]["calls"]
        assert len(got[k]) == len(ref_calls)
        for g, r in zip(got[k], ref_calls):
            assert [x[2] for x in g] == [x[2] for x in r["results"]], k
    # slots are recycled and start clean
    sch.close(sid["a"])
    s2 = sch.open()
    assert sch.batch.st[sch._slot_of[s2]].T_enc == 0 and sch.n_active == 3


def test_sessions_batched_with_different_chunking_match_reference():
    run_sessions_different_chunking()


def test_recognize_segments_in_parallel_equals_one_by_one():
    """Segments of one recording decoded as parallel streams (2 slots for 3
    segments: one slot is recycled) give the same tokens as decoding each
    segment alone; timestamps = segment start + frame position / 24.
    strict_reference=False: a recycled slot starts clean."""
    import numpy as np
    from speechcatcher_amd.scheduler import recognize_segments
    speech = synth.synth_audio(21, 70000)
    segs = [(0, 30000), (30000, 52000), (52000, 70000)]
    sb = make_batch("TINY", 1234, "meanstd", 5, False, n_streams=2, max_frames=128, max_tokens=300,
                    pcm_capacity=1 << 16, strict_reference=False)
    res = recognize_segments(sb, speech, segs, chunk_length=8192)
    assert len(res) == 3
    for (a, b), r in zip(segs, res):
        solo = make_batch("TINY", 1234, "meanstd", 5, False, n_streams=1, max_frames=128, max_tokens=300,
                          pcm_capacity=1 << 16)
        seg = speech[a:b]
        for pos in range(0, len(seg), 8192):
            end = min(pos + 8192, len(seg))
            solo.push([(0, seg[pos:end], end >= len(seg))])
        h = solo.hypotheses(0)[0]
        ids = [t for t in h["yseq"][1:] if t not in (0, 1, 1023)]
        assert r["token_ids"] == ids
        assert len(r["token_timestamps"]) == len(ids)
        assert all(t >= a / 16000.0 for t in r["token_timestamps"])
        assert np.all(np.diff(r["token_timestamps"]) >= 0)


def run_segments_serial_strict(backend=None, device="cpu"):
    """The reference CLI with one worker (speechcatcher.py:474-479: segments decoded serially on ONE
    Speech2TextStreaming with reset() after every final chunk, :618-619) = one stream slot under
    strict_reference: every segment after the first is scored over the previous segments' stale CTC
    table.  Checked against the oracle doing exactly that (reference_reset_quirk)."""
    from helpers import oracle_model
    from oracle.ref_port import RefPortStreaming
    from speechcatcher_amd.scheduler import recognize_segments
    speech = synth.synth_audio(21, 70000)
    segs = [(0, 30000), (30000, 52000), (52000, 70000)]
    sb = make_batch("TINY", 1234, "meanstd", 5, False, n_streams=1, backend=backend, device=device, max_frames=128,
                    max_tokens=300, pcm_capacity=1 << 16)
    res = recognize_segments(sb, speech, segs, chunk_length=8192)
    ora = RefPortStreaming(oracle_model("TINY", 1234, "meanstd"), beam_size=5, reference_reset_quirk=True)
    for (a, b), r in zip(segs, res):
        seg = speech[a:b]
        for pos in range(0, len(seg), 8192):
            end = min(pos + 8192, len(seg))
            out = ora(seg[pos:end], is_final=end >= len(seg), finalize_all=end >= len(seg))
        ora.reset()
        assert r["token_ids"] == [t for t in out[0][0]], (a, b)
    return res


def test_recognize_segments_serial_strict_equals_reference_cli_semantics():
    run_segments_serial_strict()


class _FakeContinuousBatch:
    """submit / poll stand-in for NativeStreamBatch: every submitted chunk is 'decoded' at once, poll reports the
    oldest outstanding chunk of every slot (one per slot and call, like sc_poll)."""

    def __init__(self, n):
        from collections import deque
        self.S = n
        self.out = {s: deque() for s in range(n)}
        self.n_poll = 0

    def reset(self, slot):
        pass

    def set_queue_depth(self, d):
        self.depth = d

    def submit(self, items):
        for slot, pcm, fin in items:
            self.out[slot].append(float(pcm[0]))

    def poll(self, min_done, isolate_faults=True):
        self.n_poll += 1
        has = {}
        for s, q in self.out.items():
            if q:                     # every complete chunk is reported (min_done is a lower bound)
                q.popleft()
                has[s] = False        # "the reference's early return []": no hypotheses to read back
        return has


def test_close_parks_other_sessions_replies_and_drain_terminates():
    """a reply of another session that becomes ready inside close() must be handed out by the next
    pump() / drain() - drain() used to spin on it - and a second reply of the same session must not overwrite it."""
    import numpy as np
    one = np.ones(4, dtype=np.float32)

    def scenario():
        fb = _FakeContinuousBatch(2)
        sch = StreamScheduler(fb, queue_depth=2)
        a, b = sch.open(), sch.open()
        sch.feed(b, one * 1)
        sch.feed(b, one * 2)
        sch.feed(a, one * 9)
        sch.pump(0, _collect=False, _feed=True)      # everything to the engine, no reply collected
        assert set(sch._in_flight) == {a, b} and len(sch._in_flight[b]) == 2
        sch.close(a)                                 # polls until a's chunk is reported: b's first reply comes with it
        assert len(sch._stash[b]) == 1 and len(sch._in_flight[b]) == 1 and sch.pending() == 1
        return fb, sch, b

    fb, sch, b = scenario()
    n0 = fb.n_poll
    assert sch.pump() == {b: []} and fb.n_poll == n0       # the parked reply, without another poll
    assert sch.pending() == 1
    assert sch.pump() == {b: []} and fb.n_poll == n0 + 1   # the second reply of b: not overwritten, not lost
    assert sch.pending() == 0 and not sch._stash and not sch._in_flight

    fb, sch, b = scenario()
    assert sch.drain() == {b: []} and sch.pending() == 0   # (the old drain() never returned here)
