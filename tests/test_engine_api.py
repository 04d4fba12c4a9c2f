"""API-level behaviour of the engine + result formatting (CPU, spec backend):
what Speech2TextStreaming.__call__ returns per call, reset semantics, buffer
compaction, capacity errors, degenerate chunk sizes, the 2-D feature input."""
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from speechcatcher_amd import synth
from speechcatcher_amd.engine import EngineError
from speechcatcher_amd.speech2text_streaming import hyps_to_results
from test_engine_spec import check_against_blocks, make_batch


def _results(sb, s, is_final, finalize_all, fmt="native"):
    return hyps_to_results(sb.hypotheses(s), is_final, finalize_all, None, fmt)


@pytest.mark.parametrize("name", ["tiny_c10240_b10_bbd0", "tiny_c8192_b10_bbd1", "tiny_stats64_b5"])
def test_per_call_results_match_reference_api(name):
    """(text, tokens, ids) tuples of every call: non-final calls return only
    EOS-terminated hypotheses with EMPTY text (A5), final calls the token ids."""
    js, _ = load_case(name)
    meta = js["meta"]
    fa = meta.get("finalize_all", True)
    sb = make_batch("TINY", meta["seed"], meta["stats"], meta["beam"], meta["bbd"], max_frames=256,
                    max_tokens=160, pcm_capacity=1 << 18)
    audio = synth.synth_audio(0, meta["n_samples"])
    pos = 0
    for call in js["calls"]:
        end = min(pos + meta["chunk"], len(audio))
        fin = end >= len(audio)
        out = sb.push([(0, audio[pos:end], fin)])
        pos = end
        res = _results(sb, 0, fin, fa and fin) if out[0] else []
        assert len(res) == len(call["results"])
        for got, ref in zip(res, call["results"]):
            assert got[2] == ref[2]                     # token ids
            assert got[1] == [str(t) for t in ref[2]]   # no token list -> ids as strings
        assert (sb.st[0].pcm_end - sb.st[0].pcm_start if sb.st[0].fe_started else -1) == call["waveform_buffer"]


def test_espnet_result_format_has_positions_and_hyp():
    js, _ = load_case("tiny_c10240_b10_bbd0")
    sb = make_batch("TINY", 1234, "meanstd", 10, False, max_frames=256, max_tokens=160, pcm_capacity=1 << 18)
    audio = synth.synth_audio(0, js["meta"]["n_samples"])
    for pos in range(0, len(audio), 10240):
        end = min(pos + 10240, len(audio))
        sb.push([(0, audio[pos:end], end >= len(audio))])
    res = _results(sb, 0, True, True, fmt="espnet")
    text, toks, ids, tpos, hyp = res[0]
    assert len(ids) == len(tpos) and len(res[0]) == 5
    assert res[0][-3] == ids and res[0][-2] == tpos     # what the reference CLI indexes (speechcatcher.py:623-632)
    ref = js["blocks"][-1]
    keep = [(t, p) for t, p in zip(ref["yseq"][0][1:], ref["xpos"][0][1:]) if t not in (0, 1, 1023)]
    assert ids == [t for t, _ in keep] and tpos == [p for _, p in keep]


def test_reset_gives_fresh_state_and_small_pcm_buffer_compacts():
    js, _ = load_case("tiny_c10240_b10_bbd0")
    n = js["meta"]["n_samples"]
    # pcm capacity of two chunks: every push has to compact the device buffer
    sb = make_batch("TINY", 1234, "meanstd", 10, False, max_frames=256, max_tokens=160, pcm_capacity=2 * 10240 + 512,
                    strict_reference=False)   # clean reset (the reference's own reset keeps a stale CTC table)
    audio = synth.synth_audio(0, n)
    for rep in range(2):
        sb.reset(0)
        assert sb.st[0].T_enc == 0 and sb.st[0].processed_block == 0 and sb.st[0].L == 1
        pos, nblk = 0, 0
        for call in js["calls"]:
            end = min(pos + 10240, n)
            sb.push([(0, audio[pos:end], end >= n)])
            pos = end
            nblk += call["n_blocks"]
            if call["n_blocks"]:
                check_against_blocks(sb, 0, js["blocks"][nblk - 1])


def test_capacity_limits_raise():
    sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=40, max_tokens=160, pcm_capacity=1 << 18)
    audio = synth.synth_audio(0, 80000)
    with pytest.raises(EngineError):
        for pos in range(0, 80000, 10240):
            sb.push([(0, audio[pos:pos + 10240], False)])
    sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=256, max_tokens=4, pcm_capacity=1 << 18)
    with pytest.raises(EngineError):
        for pos in range(0, 80000, 10240):
            sb.push([(0, audio[pos:pos + 10240], False)])
    with pytest.raises(EngineError):
        make_batch("TINY", 1234, "meanstd", 41, False)    # beam wider than the pre-beam


def test_degenerate_640_sample_chunks():
    """Literal 640-sample calls: 2 frames per call, the encoder never runs and
    the final call dies like the reference's Conv2d."""
    js = json.loads((GOLDEN / "tiny_short.json").read_text())
    sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=64, max_tokens=64, pcm_capacity=1 << 16)
    a = synth.synth_audio(4, 6400)
    outs = []
    for i in range(0, 6400 - 640, 640):
        o = sb.push([(0, a[i:i + 640], False)])
        outs.append(_results(sb, 0, False, False) if o[0] else [])
    assert sb.stats["enc_calls"] == 0 and sb.st[0].T_enc == 0
    assert [len(o) for o in outs] == [len(o) for o in js["640"]["results"]]
    with pytest.raises(RuntimeError):
        sb.push([(0, a[6400 - 640:], True)])


def test_precomputed_feature_input_matches_oracle():
    """2-D (T, 80) feature input (speech2text_streaming.py:438-446)."""
    from helpers import oracle_model
    from oracle.ref_port import RefPortStreaming
    model = oracle_model("TINY", 1234, "meanstd")
    g = torch.Generator().manual_seed(5)
    feats = (torch.randn(150, 80, generator=g) * 2.0 - 8.0).numpy().astype(np.float32)
    ora = RefPortStreaming(model, beam_size=5)
    sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=128, max_tokens=200, pcm_capacity=1 << 14,
                    max_chunk_samples=32768)
    for a, b in ((0, 70), (70, 150)):
        fin = b == 150
        ora(torch.from_numpy(feats[a:b]), is_final=fin, finalize_all=fin)
        norm = ((feats[a:b] - model.mean) / model.std).astype(np.float32)
        sb.push_features([(0, torch.from_numpy(norm), fin)])
    got = sb.hypotheses(0)
    ref = ora.running_hyps
    assert [h["yseq"] for h in got] == [list(h.yseq) for h in ref]
    np.testing.assert_allclose([h["score"] for h in got], [h.score for h in ref], atol=1e-3, rtol=0)


@pytest.mark.parametrize("stream_layers", [False, True])
def test_other_model_dimensions_spec_vs_oracle(stream_layers, monkeypatch):
    """d=128 / 4 heads / 3+2 layers: engine (spec backend) against the reference port.  stream_layers: the decode step as
    the two ops per layer of the stream-resident form (oracle/kernel_spec.py dec_layer_stream / dec_layer_ffn_xn, round 6:
    the per-op references of csrc/decoder_stream.hip in the lock-step GPU test) instead of the three head-parallel ones."""
    from oracle.kernel_spec import SpecBackend
    monkeypatch.setattr(SpecBackend, "stream_layers", stream_layers)
    import helpers
    import test_engine_spec
    from oracle.ref_port import RefPortStreaming
    from speechcatcher_amd.config import ModelConfig
    helpers.CFGS["MID"] = test_engine_spec.CFGS["MID"] = ModelConfig(d_model=128, enc_heads=4, enc_layers=3,
                                                                    dec_heads=4, dec_layers=2)
    model = helpers.oracle_model("MID", 1234, "meanstd")
    ora = RefPortStreaming(model, beam_size=5, use_bbd=True)
    sb = make_batch("MID", 1234, "meanstd", 5, True, max_frames=200, max_tokens=300, pcm_capacity=1 << 17)
    audio = synth.synth_audio(8, 80000)
    for pos in range(0, 80000, 8192):
        end = min(pos + 8192, 80000)
        ora(audio[pos:end], is_final=end >= 80000, finalize_all=end >= 80000)
        sb.push([(0, audio[pos:end], end >= 80000)])
    got, ref = sb.hypotheses(0), ora.running_hyps
    assert [h["yseq"] for h in got] == [list(h.yseq) for h in ref]
    np.testing.assert_allclose([h["score"] for h in got], [h.score for h in ref], atol=1e-3, rtol=0)


def test_model_blob_round_trip(tmp_path):
    """One-file model blob: same tensors, architecture and
    MVN statistics (float64 kept for the sum/count form) as the directory."""
    import numpy as np
    import torch
    from speechcatcher_amd import synth
    from speechcatcher_amd.config import TINY
    from speechcatcher_amd.speech2text_streaming import (config_from_dir, find_checkpoint, load_model_blob,
                                                         load_state_dict, load_stats, save_model_blob)
    for kind in ("meanstd", "sums"):
        mdir = synth.write_model_dir(tmp_path / kind, TINY, seed=7, stats_kind=kind)
        blob = save_model_blob(mdir, tmp_path / f"{kind}.scasr")
        sd, cfg, mean, std, toks = load_model_blob(blob)
        ref_sd = load_state_dict(find_checkpoint(mdir))
        assert cfg == config_from_dir(mdir, ref_sd)
        assert set(sd) == set(ref_sd) and all(torch.equal(sd[k], ref_sd[k]) for k in sd)
        rm, rs = load_stats(mdir)
        assert mean.dtype == np.asarray(rm).dtype
        np.testing.assert_array_equal(mean, rm)
        np.testing.assert_array_equal(std, rs)
        assert toks is None
    with open(tmp_path / "bad.scasr", "wb") as f:
        torch.save({"format": 99}, f)
    import pytest
    with pytest.raises(ValueError):
        load_model_blob(tmp_path / "bad.scasr")


