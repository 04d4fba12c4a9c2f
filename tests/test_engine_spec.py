"""Host logic of the product engine (speechcatcher_amd.engine) driven by the
torch per-kernel spec backend (oracle/kernel_spec.py), checked against the
fixtures produced by the real reference.  CPU only: validates the buffering
arithmetic, block schedule, beam bookkeeping (ancestor tables, ping-pong
hypothesis buffers, rewind) and the kernel decomposition itself."""
import json

import numpy as np
import pytest

from conftest import GOLDEN, load_case
from speechcatcher_amd import synth
from speechcatcher_amd.config import TINY, XL, SearchConfig
from speechcatcher_amd.engine import EngineError, StreamBatch
from speechcatcher_amd.weights import PackedWeights

from speechcatcher_amd.config import L_LIKE, M_DEFAULTS  # noqa: E402
CFGS = {"TINY": TINY, "XL": XL, "L_LIKE": L_LIKE, "M_DEFAULTS": M_DEFAULTS}


def make_batch(cfg_name, seed, stats, beam, bbd, n_streams=1, backend=None, device="cpu", ffn_dtype="float32",
               proj_dtype="float32", dec_dtype="float32", ctc_weight=0.3, **kw):
    """backend: None = the torch spec backend (CPU), a HipBackend = the Python engine over the HIP kernels,
    "native" = the C++ engine behind the stream-level C ABI (speechcatcher_amd.native)."""
    from oracle.kernel_spec import SpecBackend
    cfg = CFGS[cfg_name]
    sd = synth.make_state_dict(cfg, seed)
    mean, std = synth.stats_to_mean_std(synth.make_stats(cfg, kind=stats))
    sc = SearchConfig(beam_size=beam, use_bbd=bbd, ctc_weight=ctc_weight)
    if isinstance(backend, str) and backend == "native":
        from speechcatcher_amd.native import NativeStreamBatch
        return NativeStreamBatch(PackedWeights(sd, cfg, "cuda:0", mean, std, ffn_dtype=ffn_dtype, proj_dtype=proj_dtype,
                                               dec_dtype=dec_dtype),
                                 n_streams, sc, **kw)
    w = PackedWeights(sd, cfg, device, mean, std, ffn_dtype=ffn_dtype, proj_dtype=proj_dtype, dec_dtype=dec_dtype)
    return StreamBatch(w, backend or SpecBackend(), n_streams, sc, **kw)


def check_hyps(hyps, process_idx, block, score_tol=1e-3):
    """Hypotheses must equal the reference's: same token ids / xpos / scores (cumulative log-probs within
    ``score_tol`` ABSOLUTE - north star: 1e-3 - on sums of magnitude 1e2..1e3).
    Exact score ties in the reference (two hypotheses with identical float64
    totals do occur) make the ORDER of the tied entries implementation
    defined, so a permutation is accepted only among hypotheses whose
    reference totals are within score_tol of each other."""
    assert len(hyps) == len(block["yseq"])
    ref_index = {tuple(y): i for i, y in enumerate(block["yseq"])}
    assert sorted(tuple(h["yseq"]) for h in hyps) == sorted(ref_index), "hypothesis sets differ"
    for i, h in enumerate(hyps):
        j = ref_index[tuple(h["yseq"])]
        if j != i:
            assert abs(block["score"][j] - block["score"][i]) <= score_tol, \
                f"rank {i}: got the reference's rank-{j} hypothesis and their scores are not tied"
        assert h["xpos"] == block["xpos"][j]
        np.testing.assert_allclose(h["score"], block["score"][j], rtol=0, atol=score_tol)
        np.testing.assert_allclose(h["score_dec"], block["score_dec"][j], rtol=0, atol=score_tol)
        np.testing.assert_allclose(h["score_ctc"], block["score_ctc"][j], rtol=0, atol=score_tol)
    assert process_idx == block["process_idx"]


def check_against_blocks(sb, s, block, score_tol=1e-3):
    check_hyps(sb.hypotheses(s), sb.st[s].process_idx, block, score_tol)


def run_case(name, n_streams=1, stream=0, score_tol=1e-3, **kw):
    js, npz = load_case(name)
    meta = js["meta"]
    kw.setdefault("max_tokens", 200 if meta["model"] == "XL" else 160)
    if "ctc_weight" in meta:
        kw.setdefault("ctc_weight", meta["ctc_weight"])
    sb = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"],
                    n_streams=n_streams, max_frames=256, pcm_capacity=1 << 18, **kw)
    audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
    chunk = meta["chunk"]
    pos, nblk = 0, 0
    for call in js["calls"]:
        end = min(pos + chunk, len(audio))
        sb.push([(stream, audio[pos:end], end >= len(audio))])
        pos = end
        nblk += call["n_blocks"]
        assert sb.st[stream].T_enc == call["enc_buffer_len"]
        assert sb.st[stream].processed_block == call["processed_block"]
        if call["n_blocks"]:
            check_against_blocks(sb, stream, js["blocks"][nblk - 1], score_tol)
    return sb, js, npz


TINY_CASES = [f"tiny_c{c}_b{b}_bbd{d}" for c in (8192, 10240) for b in (1, 10) for d in (0, 1)] + ["tiny_c25600_b10_bbd0"]


@pytest.mark.parametrize("name", TINY_CASES)
def test_engine_matches_reference_trajectories(name):
    sb, js, npz = run_case(name)
    if npz is not None:
        T = sb.st[0].T_enc
        enc = sb.enc[:T].numpy()
        np.testing.assert_allclose(enc, npz["enc"][:T], atol=5e-4, rtol=0)


CTC_WEIGHT_CASES = [f"tiny_c10240_b10_bbd{d}_cw{w}" for d in (0, 1) for w in ("00", "05")]


@pytest.mark.parametrize("name", CTC_WEIGHT_CASES)
def test_engine_ctc_weight_fixtures(name):
    """Speech2TextStreaming(ctc_weight=...): 0.5, and 0.0 = no CTC scorer at all (beam_search.py:925).  The decoder-only
    tiny fixtures run into max_length = 500 (beam_search.py:701): the step loop's bound is part of the case."""
    run_case(name, max_tokens=520)


def test_engine_other_stream_slot_and_float64_stats():
    # run the same utterance in slot 2 of a 3-stream batch
    run_case("tiny_stats64_b5", n_streams=3, stream=2)


def test_engine_short_utterances():
    js = json.loads((GOLDEN / "tiny_short.json").read_text())
    for n in (3000, 9000, 20000):
        sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=128, max_tokens=600, pcm_capacity=1 << 16)
        sb.push([(0, synth.synth_audio(3, n), True)])
        ref = js[str(n)]
        check_against_blocks(sb, 0, ref["blocks"][-1])
        enc = np.load(GOLDEN / f"tiny_short_{n}.npz")["enc"]
        enc = enc.reshape(-1, enc.shape[-1])
        np.testing.assert_allclose(sb.enc[:sb.st[0].T_enc].numpy(), enc, atol=5e-4, rtol=0)
    sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=128, max_tokens=64, pcm_capacity=1 << 16)
    with pytest.raises(RuntimeError):
        sb.push([(0, synth.synth_audio(3, 700), True)])


def test_engine_two_streams_ragged():
    """Two different utterances interleaved with different chunk sizes in one
    batch give the same trajectories as when run alone."""
    js_a, _ = load_case("tiny_c8192_b10_bbd0")
    js_b, _ = load_case("tiny_c10240_b10_bbd0")
    sb = make_batch("TINY", 1234, "meanstd", 10, False, n_streams=2, max_frames=256, max_tokens=160,
                    pcm_capacity=1 << 18)
    audio = synth.synth_audio(0, js_a["meta"]["n_samples"])
    n = len(audio)
    pa = pb = 0
    ba = bb = 0
    ia = ib = 0
    while pa < n or pb < n:
        items = []
        if pa < n:
            ea = min(pa + 8192, n)
            items.append((0, audio[pa:ea], ea >= n))
        if pb < n:
            eb = min(pb + 10240, n)
            items.append((1, audio[pb:eb], eb >= n))
        sb.push(items)
        if pa < n:
            pa = ea
            ba += js_a["calls"][ia]["n_blocks"]
            if js_a["calls"][ia]["n_blocks"]:
                check_against_blocks(sb, 0, js_a["blocks"][ba - 1])
            ia += 1
        if pb < n:
            pb = eb
            bb += js_b["calls"][ib]["n_blocks"]
            if js_b["calls"][ib]["n_blocks"]:
                check_against_blocks(sb, 1, js_b["blocks"][bb - 1])
            ib += 1


def test_engine_uniform_batch_fast_path():
    """4 streams in the same buffering state take the broadcast planning path
    (plan computed once, offsets per stream); every stream must still follow
    the reference trajectory, including a multi-block-per-call chunk size."""
    for name, chunk in (("tiny_c10240_b10_bbd0", 10240), ("tiny_c25600_b10_bbd0", 25600)):
        js, _ = load_case(name)
        meta = js["meta"]
        sb = make_batch("TINY", 1234, "meanstd", 10, False, n_streams=4, max_frames=256, max_tokens=160,
                        pcm_capacity=1 << 18)
        audio = synth.synth_audio(0, meta["n_samples"])
        pos, nblk = 0, 0
        for call in js["calls"]:
            end = min(pos + chunk, len(audio))
            sb.push([(s, audio[pos:end], end >= len(audio)) for s in range(4)])
            pos = end
            nblk += call["n_blocks"]
            if call["n_blocks"]:
                for s in range(4):
                    check_against_blocks(sb, s, js["blocks"][nblk - 1])


def run_reset_quirk(backend=None, device="cpu", score_tol=1e-3):
    """Two utterances on ONE stream with reset() in between, exactly like
    tools/gen_golden.py recorded the real reference (tests/golden/tiny_reset.json):
    under strict_reference the second utterance is scored over the first one's
    stale CTC table (scorers.py:342-350 never clears impl)."""
    js = json.loads((GOLDEN / "tiny_reset.json").read_text())
    sb = make_batch("TINY", 1234, "meanstd", 5, False, backend=backend, device=device, max_frames=256,
                    max_tokens=200, pcm_capacity=1 << 17)
    nblk = 0
    for sid, n in ((5, 40000), (6, 50000)):
        a = synth.synth_audio(sid, n)
        sb.reset(0)
        pos = 0
        while pos < n:
            end = min(pos + 10240, n)
            b0 = sb.stats["dec_blocks"]
            sb.push([(0, a[pos:end], end >= n)])
            pos = end
            nblk += sb.stats["dec_blocks"] - b0
            if sb.stats["dec_blocks"] > b0:
                check_against_blocks(sb, 0, js["blocks"][nblk - 1], score_tol)
    assert nblk == len(js["blocks"])
    return sb, js


def test_engine_reset_keeps_stale_ctc_table_like_the_reference():
    sb, js = run_reset_quirk()
    assert sb.st[0].T_ctc >= sb.st[0].T_kv


def test_engine_clean_reset_when_not_strict():
    """strict_reference=False: reset() gives a clean stream - the second utterance equals
    the same utterance on a fresh batch."""
    a = synth.synth_audio(6, 50000)

    def run(sb):
        for pos in range(0, len(a), 10240):
            end = min(pos + 10240, len(a))
            sb.push([(0, a[pos:end], end >= len(a))])
        return sb.hypotheses(0)

    sb = make_batch("TINY", 1234, "meanstd", 5, False, max_frames=256, max_tokens=200, pcm_capacity=1 << 17,
                    strict_reference=False)
    b = synth.synth_audio(5, 40000)
    for pos in range(0, len(b), 10240):
        sb.push([(0, b[pos:pos + 10240], pos + 10240 >= len(b))])
    sb.reset(0)
    got = run(sb)
    ref = run(make_batch("TINY", 1234, "meanstd", 5, False, max_frames=256, max_tokens=200, pcm_capacity=1 << 17))
    assert [h["yseq"] for h in got] == [h["yseq"] for h in ref]
    assert all(abs(x["score"] - y["score"]) < 1e-9 for x, y in zip(got, ref))


def run_after_final(bbd, backend=None, device="cpu", score_tol=1e-3):
    """Calls continuing after is_final=True with no reset (what the reference server does,
    speechcatcher_server.py:270) against the fixture recorded from the real reference."""
    js = json.loads((GOLDEN / f"tiny_after_final_bbd{bbd}.json").read_text())
    sb = make_batch("TINY", 1234, "meanstd", 3, bool(bbd), backend=backend, device=device, max_frames=256,
                    max_tokens=200, pcm_capacity=1 << 17)
    a = synth.synth_audio(5, 10240 * 10)
    nblk = 0
    for i, call in enumerate(js["calls"]):
        sb.push([(0, a[i * 10240:(i + 1) * 10240], call["is_final"])])
        nblk += call["n_blocks"]
        assert sb.st[0].T_enc == call["enc_buffer_len"], i
        assert sb.st[0].processed_block == call["processed_block"], i
        if call["n_blocks"]:
            check_against_blocks(sb, 0, js["blocks"][nblk - 1], score_tol)
    return sb


@pytest.mark.parametrize("bbd", [0, 1])
def test_engine_calls_after_final_without_reset(bbd):
    run_after_final(bbd)


def run_kv_pool_exhaustion(backend=None, device="cpu"):
    """The self-attention K|V pool (include/scasr.h: sc_search.skv): rows are handed out per step and reclaimed when no
    live hypothesis descends from them.  A pool of one row per (position, hypothesis) can never run out; the default
    (1.5 rows per position) serves the fixture utterance with the same results; a pool that is too small fails the
    stream with a capacity error - alone, when faults are isolated."""
    js, _ = load_case("tiny_c10240_b10_bbd0")
    meta = js["meta"]
    audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
    kw = dict(max_frames=256, max_tokens=160, pcm_capacity=1 << 18)
    if backend is not None:
        kw.update(backend=backend)
        if not isinstance(backend, str):
            kw.update(device=device)

    def run(sb, streams=(0,)):
        out = None
        for pos in range(0, len(audio), meta["chunk"]):
            end = min(pos + meta["chunk"], len(audio))
            out = sb.push([(s, audio[pos:end], end >= len(audio)) for s in streams], isolate_faults=len(streams) > 1)
        return out

    full = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], kv_pool_rows=160 * 10, **kw)
    run(full)
    dflt = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], **kw)
    run(dflt)
    assert [h["yseq"] for h in dflt.hypotheses(0)] == [h["yseq"] for h in full.hypotheses(0)]
    check_against_blocks(dflt, 0, js["blocks"][-1])
    small = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], kv_pool_rows=24, **kw)
    with pytest.raises(EngineError, match="pool"):
        run(small)
    # two streams, the pool is per stream: both fail in isolation (each is reset), the call itself succeeds
    two = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], n_streams=2, kv_pool_rows=24, **kw)
    failed = set()
    for pos in range(0, len(audio), meta["chunk"]):
        end = min(pos + meta["chunk"], len(audio))
        out = two.push([(s, audio[pos:end], end >= len(audio)) for s in (0, 1) if s not in failed], isolate_faults=True)
        failed |= {s for s, r in out.items() if isinstance(r, EngineError)}
    assert failed == {0, 1}
    assert two.st[0].T_enc == 0 and two.st[1].T_enc == 0       # both have been reset
    # ONE stream runs out: the fault is that stream's alone - the other stream of the batch, on a short
    # utterance, decodes exactly what it decodes in a batch with the default pool - and the failed stream, reset by the
    # fault, is usable again: the same short utterance on it gives the same hypotheses
    short = synth.synth_audio(3, 9000)
    ref = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], **kw)
    ref.push([(0, short, True)])
    want = ref.hypotheses(0)
    assert 1 < len(want[0]["yseq"]) < 16           # well inside a 40-row pool
    mix = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], n_streams=2, kv_pool_rows=40,
                     strict_reference=False, **kw)   # (strict: the next utterance would meet the stale CTC table, quirk A1)
    failed_at = None
    for k, pos in enumerate(range(0, len(audio), meta["chunk"])):
        end = min(pos + meta["chunk"], len(audio))
        items = [(0, audio[pos:end], end >= len(audio))] + ([(1, short, True)] if k == 0 else [])
        out = mix.push(items, isolate_faults=True)
        assert not isinstance(out.get(1), EngineError)
        if isinstance(out[0], EngineError):
            assert "pool" in str(out[0])
            failed_at = k
            break
    assert failed_at is not None and failed_at > 0
    got = mix.hypotheses(1)
    assert [h["yseq"] for h in got] == [h["yseq"] for h in want]
    np.testing.assert_allclose([h["score"] for h in got], [h["score"] for h in want], rtol=0, atol=1e-4)
    assert mix.st[0].T_enc == 0                                 # reset by the fault
    mix.push([(0, short, True)])
    again = mix.hypotheses(0)
    assert [h["yseq"] for h in again] == [h["yseq"] for h in want]
    np.testing.assert_allclose([h["score"] for h in again], [h["score"] for h in want], rtol=0, atol=1e-4)


def test_kv_pool_exhaustion_is_a_capacity_fault():
    run_kv_pool_exhaustion()


def test_split16_is_refused_for_weights_whose_activations_can_leave_the_fp16_range():
    """weights.py split16_operand_bounds: the split-precision kernels split activations into fp16 hi | lo; the bound of
    every such activation follows from the weights (LayerNorm outputs, |W| @ bound + |bias|), and a model that could
    overflow fp16 there is refused at load time instead of producing inf on the device."""
    from speechcatcher_amd.weights import split16_operand_bounds
    sd = synth.make_state_dict(XL, 1234)
    w = PackedWeights(sd, XL, "cpu", ffn_dtype="split16", proj_dtype="split16")
    b = split16_operand_bounds(w.enc[0], "ln2", XL.d_model)
    assert set(b) == {"ffn_in", "ffn_hidden", "proj_in", "attn_context"} and 0 < max(b.values()) < 65504.0
    # the bound holds on actual data: LayerNorm of extreme rows, through the first Linear
    import torch
    torch.manual_seed(0)
    x = torch.randn(64, XL.d_model) * torch.logspace(-3, 3, 64)[:, None]
    x[0, 1:] = 0.0                                                    # one-hot row: the largest z-score
    xn = torch.nn.functional.layer_norm(x, (XL.d_model,), w.enc[0]["ln2_g"], w.enc[0]["ln2_b"], 1e-12)
    h = xn @ w.enc[0]["w1"].T + w.enc[0]["b1"]
    assert float(xn.abs().max()) <= b["ffn_in"] * (1 + 1e-5) and float(h.abs().max()) <= b["ffn_hidden"] * (1 + 1e-5)
    sd2 = dict(sd)
    sd2["encoder.encoders.3.norm2.weight"] = sd["encoder.encoders.3.norm2.weight"] * 3.0e4
    with pytest.raises(ValueError, match="enc layer 3"):
        PackedWeights(sd2, XL, "cpu", ffn_dtype="split16")
    PackedWeights(sd2, XL, "cpu")                                   # the fp32 path has no such limit
