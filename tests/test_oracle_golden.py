"""Pin the oracle (oracle/ref_port.py) against fixtures produced by the REAL
reference (tools/gen_golden.py).  CPU only.

Tolerances: the oracle uses the same torch ops in the same order as the
reference, so on the generating machine the match is exact; a different host
CPU may pick other MKL kernels, hence small float tolerances.  Token ids must
match exactly wherever beam-score margins allow it.
"""
import json

import numpy as np
import pytest

from conftest import GOLDEN, load_case
from helpers import oracle_model, run_oracle_stream
from speechcatcher_amd import synth

ATOL_FEATS = 2e-4
ATOL_ENC = 2e-4
ATOL_SCORE = 1e-3   # north star: hypothesis log-probs within 1e-3 (absolute, on sums of 1e2..1e3)


def _check_blocks(trace, blocks, score_tol=ATOL_SCORE):
    assert len(trace) == len(blocks)
    for k, (a, b) in enumerate(zip(trace, blocks)):
        assert a["T"] == b["T"], k
        assert a["is_final"] == b["is_final"], k
        assert a["process_idx"] == b["process_idx"], k
        assert a["yseq"] == b["yseq"], f"block {k}"
        assert a["xpos"] == b["xpos"], f"block {k}"
        np.testing.assert_allclose(a["score"], b["score"], rtol=0, atol=score_tol)
        np.testing.assert_allclose(a["score_dec"], b["score_dec"], rtol=0, atol=score_tol)
        np.testing.assert_allclose(a["score_ctc"], b["score_ctc"], rtol=0, atol=score_tol)


TINY_CASES = [f"tiny_c{c}_b{b}_bbd{d}" for c in (8192, 10240) for b in (1, 10) for d in (0, 1)] + ["tiny_c25600_b10_bbd0"]


@pytest.mark.parametrize("name", TINY_CASES)
def test_tiny_trajectories(name):
    js, npz = load_case(name)
    meta = js["meta"]
    model = oracle_model("TINY", meta["seed"], meta["stats"])
    audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
    s, feats, encs, calls = run_oracle_stream(model, audio, meta["chunk"], meta["beam"], meta["bbd"])
    _check_blocks(s.trace, js["blocks"])
    # final API tuples: token ids of every returned hypothesis
    ref_final = js["calls"][-1]["results"]
    got_final = calls[-1]["results"]
    assert [r[2] for r in ref_final] == [g[0] for g in got_final]
    for c_ref, c_got in zip(js["calls"], calls):
        assert c_ref["n_blocks"] == c_got["n_blocks"]
        assert len(c_ref["results"]) == len(c_got["results"])
    if npz is not None:
        assert [f.shape[0] for f in feats] == npz["feats_lens"].tolist()
        np.testing.assert_allclose(np.concatenate(feats, 0), npz["feats"], atol=ATOL_FEATS, rtol=0)
        assert [e.shape[0] for e in encs] == npz["enc_lens"].tolist()
        np.testing.assert_allclose(np.concatenate(encs, 0), npz["enc"], atol=ATOL_ENC, rtol=0)


def test_tiny_float64_stats_and_no_finalize_all():
    js, npz = load_case("tiny_stats64_b5")
    meta = js["meta"]
    model = oracle_model("TINY", meta["seed"], meta["stats"])
    audio = synth.synth_audio(0, meta["n_samples"])
    s, feats, encs, calls = run_oracle_stream(model, audio, meta["chunk"], meta["beam"], meta["bbd"], finalize_all=False)
    _check_blocks(s.trace, js["blocks"])
    np.testing.assert_allclose(np.concatenate(feats, 0), npz["feats"], atol=ATOL_FEATS, rtol=0)
    for c_ref, c_got in zip(js["calls"], calls):
        assert [r[2] for r in c_ref["results"]] == [g[0] for g in c_got["results"]]


def test_frontend_counts_and_values():
    from oracle.ref_port import RefPortStreaming, logmel
    import torch
    js = json.loads((GOLDEN / "frontend.json").read_text())
    npz = np.load(GOLDEN / "frontend.npz")
    model = oracle_model("TINY", 1234, "meanstd")
    np.testing.assert_array_equal(model.mel_fb.numpy(), npz["mel_fb"])
    audio = synth.synth_audio(7, 64000)
    lm = logmel(torch.from_numpy(audio[:10480]).unsqueeze(0), model.window, model.mel_fb, 512, 160, 400)
    np.testing.assert_allclose(lm[0].numpy(), npz["logmel_10480"], atol=1e-4, rtol=0)
    for chunk, ref in js.items():
        chunk = int(chunk)
        s = RefPortStreaming(model, beam_size=1)
        st, pos, counts, bufs, allf = None, 0, [], [], []
        while pos < len(audio):
            end = min(pos + chunk, len(audio))
            feats, st = s.apply_frontend(torch.from_numpy(audio[pos:end]), st, end >= len(audio))
            counts.append(-1 if feats is None else int(feats.size(1)))
            bufs.append(-1 if (st is None or st.get("waveform_buffer") is None) else int(st["waveform_buffer"].numel()))
            if feats is not None:
                allf.append(feats[0].numpy())
            pos = end
        assert counts == ref["counts"], chunk
        assert bufs == ref["buffers"], chunk
        if f"feats_{chunk}" in npz:
            np.testing.assert_allclose(np.concatenate(allf, 0), npz[f"feats_{chunk}"], atol=ATOL_FEATS, rtol=0)


def test_short_utterances_and_degenerate_cases():
    from oracle.ref_port import RefPortStreaming
    js = json.loads((GOLDEN / "tiny_short.json").read_text())
    model = oracle_model("TINY", 1234, "meanstd")
    for n in (3000, 9000, 20000):
        a = synth.synth_audio(3, n)
        s = RefPortStreaming(model, beam_size=5)
        s.trace = []
        res = s(a, is_final=True, finalize_all=True)
        ref = js[str(n)]
        _check_blocks(s.trace, ref["blocks"])
        assert [r[2] for r in ref["results"]] == [g[0] for g in res]
        enc = np.load(GOLDEN / f"tiny_short_{n}.npz")["enc"]
        np.testing.assert_allclose(s.last_enc_out[0].numpy(), enc.reshape(-1, enc.shape[-1]), atol=ATOL_ENC, rtol=0)
    # too-short final: the reference raises RuntimeError inside Conv2d (A3)
    assert js["700_exc"] == "RuntimeError"
    s = RefPortStreaming(model, beam_size=5)
    with pytest.raises(RuntimeError):
        s(synth.synth_audio(3, 700), is_final=True, finalize_all=True)
    # literal 640-sample chunks: 2 frames per call, encoder never runs (A2/A3)
    a = synth.synth_audio(4, 6400)
    s = RefPortStreaming(model, beam_size=5)
    s.trace = []
    outs = [s(a[i:i + 640], is_final=False) for i in range(0, 6400 - 640, 640)]
    assert js["640"]["enc_calls"] == 0 and len(s.trace) == 0
    assert [len(o) for o in outs] == [len(o) for o in js["640"]["results"]]
    assert js["640_final_exc"] == "RuntimeError"
    with pytest.raises(RuntimeError):
        s(a[6400 - 640:], is_final=True, finalize_all=True)


def test_reset_quirk_matches_reference():
    """After reset() the reference keeps its stale CTC table (scorer.impl is
    never cleared); the oracle reproduces it under reference_reset_quirk."""
    from oracle.ref_port import RefPortStreaming
    js = json.loads((GOLDEN / "tiny_reset.json").read_text())
    model = oracle_model("TINY", 1234, "meanstd")
    s = RefPortStreaming(model, beam_size=5, reference_reset_quirk=True)
    s.trace = []
    for sid, n in ((5, 40000), (6, 50000)):
        a = synth.synth_audio(sid, n)
        s.reset()
        pos = 0
        while pos < n:
            end = min(pos + 10240, n)
            res = s(a[pos:end], is_final=end >= n, finalize_all=end >= n)
            pos = end
    _check_blocks(s.trace, js["blocks"], score_tol=1e-3)
    assert [r[2] for r in js["final"]] == [g[0] for g in res]


CTC_WEIGHT_CASES = [f"tiny_c10240_b10_bbd{d}_cw{w}" for d in (0, 1) for w in ("00", "05")]


@pytest.mark.parametrize("name", CTC_WEIGHT_CASES + [pytest.param(f"xl_c10240_b10_bbd0_cw{w}", marks=pytest.mark.slow) for w in ("00", "05")])
def test_ctc_weight_is_part_of_the_surface(name):
    """Speech2TextStreaming(ctc_weight=...) (speech2text_streaming.py:143-150): 0.5, and 0.0 = NO CTC scorer at all
    (beam_search.py:925: decoder-only search, score_ctc stays 0).  Fixtures: tools/gen_golden.py --ctc-weights."""
    js, _ = load_case(name)
    meta = js["meta"]
    model = oracle_model(meta["model"], meta["seed"], meta["stats"])
    audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
    s, feats, encs, calls = run_oracle_stream(model, audio, meta["chunk"], meta["beam"], bool(meta["bbd"]),
                                              ctc_weight=meta["ctc_weight"])
    _check_blocks(s.trace, js["blocks"])
    assert [r[2] for r in js["calls"][-1]["results"]] == [g[0] for g in calls[-1]["results"]]
    if meta["ctc_weight"] <= 0:
        assert all(v == 0.0 for b in js["blocks"] for v in b["score_ctc"])


XL_CASES = ["xl_c10240_b10_bbd0", "xl_c10240_b10_bbd1", "xl_c25600_b10_bbd0", "xl_c8192_b10_bbd1",
            "xl_c8192_b5_bbd1", "xl_c10240_b1_bbd0"]


@pytest.mark.slow
@pytest.mark.parametrize("name", XL_CASES)
def test_xl_trajectories(name):
    js, npz = load_case(name)
    meta = js["meta"]
    model = oracle_model("XL", meta["seed"], meta["stats"])
    audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
    s, feats, encs, calls = run_oracle_stream(model, audio, meta["chunk"], meta["beam"], bool(meta["bbd"]))
    _check_blocks(s.trace, js["blocks"], score_tol=1e-3)
    if npz is not None:
        np.testing.assert_allclose(np.concatenate(encs, 0), npz["enc"], atol=1e-3, rtol=0)


@pytest.mark.parametrize("bbd", [0, 1])
def test_calls_after_final_without_reset_match_reference(bbd):
    """The reference server never resets its model (speechcatcher_server.py:270): the stream goes on
    after is_final=True.  Fixture recorded from the real reference (tools/gen_golden.py --after-final)."""
    from oracle.ref_port import RefPortStreaming
    js = json.loads((GOLDEN / f"tiny_after_final_bbd{bbd}.json").read_text())
    s = RefPortStreaming(oracle_model("TINY", 1234, "meanstd"), beam_size=3, use_bbd=bool(bbd))
    s.trace = []
    a = synth.synth_audio(5, 10240 * 10)
    for i, call in enumerate(js["calls"]):
        res = s(a[i * 10240:(i + 1) * 10240], is_final=call["is_final"])
        assert [r[2] for r in call["results"]] == [g[0] for g in res], i
        assert (s.encoder_buffer.shape[1] if s.encoder_buffer is not None else 0) == call["enc_buffer_len"]
    _check_blocks(s.trace, js["blocks"], score_tol=1e-3)
