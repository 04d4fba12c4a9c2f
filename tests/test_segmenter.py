"""Segmentation: the cut search against golden vectors from
the real reference class, the chunk-aligned segment plan, and the file-level
loop on CPU with the spec backend."""
import json

import numpy as np

from conftest import GOLDEN
from speechcatcher_amd import synth
from speechcatcher_amd.segmenter import (CutSearch, constrain_segments, interpolate_repeating_positions,
                                         log_fbank_energy, merge_paragraphs, plan_segments, recognize_recording,
                                         segment_speech)


def test_cut_search_matches_reference_golden():
    cases = json.loads((GOLDEN / "segmenter.json").read_text())["cases"]
    assert len(cases) >= 5
    for c in cases:
        e = synth.synth_energy_curve(c["seed"], c["n"])
        s = CutSearch(ideal_segment_len=int(c["average_segment_length"] * 100), **c["params"])
        assert [list(x) for x in s.search(e, c["n"])] == c["segments"], c["seed"]


def test_constrain_and_plan():
    assert constrain_segments([(0, 50000), (50000, 52000)], 180) == [(0, 18000), (18000, 36000), (36000, 50000),
                                                                     (50000, 52000)]
    # 100 s of audio, cuts at 30 s and 95 s: the second leaves < 10 s and is dropped (speechcatcher.py:430)
    n, rate, chunk = 1_600_000, 16000, 8192
    r = plan_segments(n, rate, [(0, 3000), (3000, 9500), (9500, 10000)], chunk)
    assert r[0][0] == 0 and r[-1][1] == n and len(r) == 2
    assert all(a % chunk == 0 for a, _ in r) and r[0][1] == r[1][0]
    i_fin = int(np.ceil((30.0 * rate - chunk) / chunk))
    assert r[0][1] == (i_fin + 1) * chunk
    assert plan_segments(50000, rate, [], chunk) == [(0, 50000)]


def test_log_fbank_shape_and_silence_detection():
    rate = 16000
    rng = np.random.RandomState(0)
    x = (rng.randn(rate * 4) * 3000).astype(np.int16)
    x[rate:2 * rate] = (rng.randn(rate) * 30).astype(np.int16)      # one quiet second
    fb = log_fbank_energy(x, rate)
    assert fb.shape == (1 + int(np.ceil((len(x) - 400) / 160)), 26) and np.isfinite(fb).all()
    e = fb.sum(-1)
    assert e[110:190].mean() < e[10:90].mean() - 20      # the pause is far below the speech level
    assert log_fbank_energy(np.zeros(100, np.int16), rate).shape == (1, 26)


def run_recording(n_slots, seconds=70, backend=None, device="cpu", check_oracle=False):
    """File loop: int16 recording -> segments -> streams -> text + per-segment token timestamps.
    check_oracle (n_slots == 1): the reference CLI with one worker decodes the segments serially on ONE
    model with reset() after every final chunk (speechcatcher.py:474-479,618-619) - token ids per segment
    must equal the oracle doing exactly that (stale CTC table after reset included)."""
    from test_engine_spec import make_batch
    from speechcatcher_amd.segmenter import recognize_recording_segments
    rate = 16000
    x = (synth.synth_audio(40, seconds * rate) * 20000)
    for t0 in (18, 41):                                   # two pauses
        if (t0 + 2) * rate < len(x):
            x[t0 * rate:(t0 + 2) * rate] *= 0.01
    x = x.astype(np.int16)
    segs = segment_speech(x, rate, average_segment_length=20.0)
    assert len(segs) >= 2 and segs[0][0] == 0
    sb = make_batch("TINY", 1234, "meanstd", 3, True, n_streams=n_slots, backend=backend, device=device,
                    max_frames=2000, max_tokens=1200, pcm_capacity=1 << 21)
    if check_oracle:
        from helpers import oracle_model
        from oracle.ref_port import RefPortStreaming
        ranges, res = recognize_recording_segments(sb, x, rate, chunk_length=8192, average_segment_length=20.0)
        assert len(ranges) >= 2
        ora = RefPortStreaming(oracle_model("TINY", 1234, "meanstd"), beam_size=3, use_bbd=True,
                               reference_reset_quirk=True)
        got = []
        for (a, b) in ranges:
            seg = x[a:b].astype(np.float32) / 32768.0           # CLI input scaling (speechcatcher.py:421)
            for pos in range(0, len(seg), 8192):
                end = min(pos + 8192, len(seg))
                out = ora(seg[pos:end], is_final=end >= len(seg), finalize_all=end >= len(seg))
            ora.reset()
            got.append([int(t) for t in out[0][0]] if out else [])
        assert [r["token_ids"] for r in res] == got
        return None, res
    text, info = recognize_recording(sb, x, rate, chunk_length=8192, average_segment_length=20.0)
    assert len(info) >= 1 and info[0]["start"] == 0.0 and abs(info[-1]["end"] - float(seconds)) < 1e-6
    assert all(a["end"] == b["start"] for a, b in zip(info[:-1], info[1:]))
    assert isinstance(text, str) and text.endswith("\n")
    for seg in info:
        assert len(seg["tokens"]) == len(seg["token_timestamps"])
        assert all(seg["start"] <= t <= seg["end"] + 1.0 for t in seg["token_timestamps"])
    return text, info


def test_recognize_recording_segments_run_as_parallel_streams():
    run_recording(2, seconds=70)


def test_recognize_recording_serial_equals_reference_cli_semantics():
    run_recording(1, seconds=64, check_oracle=True)


def test_paragraph_merge_and_position_interpolation():
    segs = [{"start": 0.0, "end": 10.0, "text": "hallo welt", "tokens": ["a", "b"], "token_timestamps": [1.0, 2.0]},
            {"start": 10.0, "end": 20.0, "text": "und weiter.", "tokens": ["c"], "token_timestamps": [11.0]},
            {"start": 20.0, "end": 30.0, "text": "neuer satz", "tokens": ["d"], "token_timestamps": [21.0]}]
    text, info = merge_paragraphs(segs)
    assert text == "hallo welt und weiter.\n\nNeuer satz\n"
    assert len(info) == 2 and info[0]["end"] == 20.0 and info[0]["tokens"] == ["a", "b", "c"]
    assert info[0]["token_timestamps"] == [1.0, 2.0, 11.0] and info[1]["text"] == "Neuer satz"
    assert merge_paragraphs([]) == ("\n", [])
    # runs of equal frame positions are spread between the previous value and the run's value; like the
    # reference (speechcatcher.py:343-348, checked against its function on random inputs when this was
    # written) the interpolated values of a run come out in DEcreasing order before the run's last element
    assert interpolate_repeating_positions([4, 4, 8, 8, 8, 9]) == [2.0, 4.0, 4.0 + 8 / 3, 4.0 + 4 / 3, 8.0, 9.0]
    assert interpolate_repeating_positions([3]) == [3.0] and interpolate_repeating_positions([]) == []
