#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference (imported read-only
from /root/reference) on seeded synthetic weights and audio.

Run in the survey container only (the reference never travels):

    python tools/gen_golden.py            # writes tests/golden/*

Inputs are fully determined by speechcatcher_amd.synth (weights seed 1234,
audio seed 1000+stream_id), so the fixtures only hold expected OUTPUTS plus
the few parameters that name the case.  Fixture kinds:

  G1/G2  frontend feats per call for several chunk sizes (+float64 MVN stats)
  G3/G4  encoder output per call (tiny: full tensors)
  G5-G7  per-step decoder log-probs, pre-beam ids, CTC partial scores, fusion
  G8     per-block beam trajectories (yseq / score / per-scorer scores / xpos)
  G9     API behaviour: non-final returns, final tuples, degenerate 640 case,
         behaviour after reset() (stale CTC table quirk)
"""
import json
import os
import sys
import tempfile
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools" / "ref_shim"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from speechcatcher_amd import synth  # noqa: E402
from speechcatcher_amd.config import TINY, XL  # noqa: E402

import logging  # noqa: E402
logging.disable(logging.WARNING)

from speechcatcher.speech2text_streaming import Speech2TextStreaming  # noqa: E402

OUT = ROOT / "tests" / "golden"


def hyps_to_json(hyps):
    return {
        "yseq": [h.yseq.tolist() for h in hyps],
        "score": [float(h.score) for h in hyps],
        "score_dec": [float(h.scores.get("decoder", 0.0)) for h in hyps],
        "score_ctc": [float(h.scores.get("ctc", 0.0)) for h in hyps],
        "xpos": [h.xpos.tolist() for h in hyps],
    }


class Recorder:
    """Hooks into a reference Speech2TextStreaming instance."""

    def __init__(self, s2t, record_steps=0):
        self.s2t = s2t
        self.blocks = []
        self.enc_outs = []
        self.feats = []
        self.steps = []
        self.record_steps = record_steps
        bs = s2t.beam_search
        orig_decode = bs._decode_one_block
        orig_enc = bs.encoder.forward_infer
        orig_score = bs.beam_search.batch_score_hypotheses

        def decode(encoder_out, prev_state, is_final=False):
            p0 = bs.process_idx
            st = orig_decode(encoder_out, prev_state, is_final)
            rec = hyps_to_json(st.hypotheses)
            rec.update({"T": int(encoder_out.size(1)), "is_final": bool(is_final),
                        "process_idx": int(bs.process_idx), "process_idx_in": int(p0)})
            self.blocks.append(rec)
            return st

        def enc(xs_pad, ilens, prev_states, is_final):
            self.feats.append(xs_pad.detach().clone()[0].numpy())
            out = orig_enc(xs_pad, ilens, prev_states, is_final)
            y = out[0].detach().clone()
            if y.dim() == 3:
                y = y[0]
            self.enc_outs.append(y.numpy())
            return out

        def score(hyps, encoder_out, pre_beam_size=40):
            comb, states, indiv = orig_score(hyps, encoder_out, pre_beam_size)
            if len(self.steps) < self.record_steps:
                full = bs.weights["decoder"] * indiv["decoder"]
                _, ids = torch.topk(full, k=min(pre_beam_size, full.size(-1)), dim=-1)
                self.steps.append({
                    "T": int(encoder_out.size(1)),
                    "yseq": np.stack([h.yseq.numpy() for h in hyps]),
                    "logp": indiv["decoder"].detach().clone().numpy(),
                    "ctc": indiv["ctc"].detach().clone().numpy(),
                    "combined": comb.detach().clone().numpy(),
                    "pre_ids": ids.numpy(),
                })
            return comb, states, indiv

        bs._decode_one_block = decode
        bs.encoder.forward_infer = enc
        bs.beam_search.batch_score_hypotheses = score


def results_to_json(res):
    return [[text, list(toks), [int(t) for t in ids]] for text, toks, ids in res]


def run_stream(model_dir, audio, chunk, beam, bbd, record_steps=0, finalize_all=True, ctc_weight=0.3):
    s2t = Speech2TextStreaming(model_dir, beam_size=beam, ctc_weight=ctc_weight, device="cpu", use_bbd=bbd)
    rec = Recorder(s2t, record_steps)
    n = len(audio)
    calls = []
    feats_calls = []
    pos = 0
    while pos < n:
        end = min(pos + chunk, n)
        is_final = end >= n
        nb0 = len(rec.blocks)
        nf0 = len(rec.feats)
        res = s2t(audio[pos:end], is_final=is_final, finalize_all=finalize_all and is_final)
        calls.append({
            "n_samples": end - pos, "is_final": is_final, "n_blocks": len(rec.blocks) - nb0,
            "enc_called": len(rec.feats) - nf0,
            "results": results_to_json(res),
            "waveform_buffer": (int(s2t.frontend_states["waveform_buffer"].numel())
                                if s2t.frontend_states and s2t.frontend_states.get("waveform_buffer") is not None else -1),
            "processed_block": int(s2t.beam_search.processed_block),
            "enc_buffer_len": int(s2t.beam_search.encoder_buffer.shape[1]) if s2t.beam_search.encoder_buffer is not None else 0,
        })
        pos = end
    return s2t, rec, calls


def save_case(name, meta, rec, calls, with_tensors=True, with_steps=False):
    js = {"meta": meta, "calls": calls, "blocks": rec.blocks}
    (OUT / f"{name}.json").write_text(json.dumps(js))
    if with_tensors:
        arrs = {}
        arrs["feats_lens"] = np.array([f.shape[0] for f in rec.feats], np.int64)
        arrs["feats"] = np.concatenate(rec.feats, 0).astype(np.float32) if rec.feats else np.zeros((0, 80), np.float32)
        arrs["enc_lens"] = np.array([e.shape[0] for e in rec.enc_outs], np.int64)
        d = rec.enc_outs[0].shape[-1] if rec.enc_outs else 1
        arrs["enc"] = (np.concatenate([e.reshape(-1, d) for e in rec.enc_outs], 0).astype(np.float32)
                       if rec.enc_outs else np.zeros((0, d), np.float32))
        if with_steps:
            for i, st in enumerate(rec.steps):
                for k in ("yseq", "logp", "ctc", "combined", "pre_ids"):
                    arrs[f"step{i}_{k}"] = st[k]
                arrs[f"step{i}_T"] = np.array(st["T"])
        np.savez_compressed(OUT / f"{name}.npz", **arrs)


def frontend_cases(model_dir):
    """G1/G2: per-call feature frame counts and (for 2 sizes) values."""
    audio = synth.synth_audio(7, 64000)
    out = {}
    arrs = {}
    for chunk in (400, 640, 1000, 8192, 10240, 25600):
        s2t = Speech2TextStreaming(model_dir, beam_size=1, device="cpu")
        st = None
        pos = 0
        counts, bufs = [], []
        allf = []
        while pos < len(audio):
            end = min(pos + chunk, len(audio))
            final = end >= len(audio)
            feats, _, st = s2t.apply_frontend(torch.from_numpy(audio[pos:end]), st, is_final=final)
            counts.append(-1 if feats is None else int(feats.size(1)))
            bufs.append(-1 if (st is None or st.get("waveform_buffer") is None) else int(st["waveform_buffer"].numel()))
            if feats is not None:
                allf.append(feats[0].numpy())
            pos = end
        out[str(chunk)] = {"counts": counts, "buffers": bufs}
        if chunk in (1000, 10240):
            arrs[f"feats_{chunk}"] = np.concatenate(allf, 0).astype(np.float32)
    # raw log-mel (no MVN, no trimming) of one 10 480-sample segment: pins the STFT
    s2t = Speech2TextStreaming(model_dir, beam_size=1, device="cpu")
    lm, _ = s2t.model.frontend(torch.from_numpy(audio[:10480]).unsqueeze(0))
    arrs["logmel_10480"] = lm[0].numpy().astype(np.float32)
    arrs["mel_fb"] = s2t.model.frontend.mel_fb.numpy()
    (OUT / "frontend.json").write_text(json.dumps(out))
    np.savez_compressed(OUT / "frontend.npz", **arrs)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = Path(tempfile.mkdtemp(prefix="golden_"))

    # ---------------- tiny model: full tensors ----------------
    tiny_dir = synth.write_model_dir(tmp / "tiny", TINY, seed=1234, stats_kind="meanstd")
    tiny64_dir = synth.write_model_dir(tmp / "tiny64", TINY, seed=1234, stats_kind="sums")
    frontend_cases(tiny_dir)

    audio = synth.synth_audio(0, 16000 * 6 + 3217)  # ragged tail
    for chunk in (8192, 10240, 25600):
        for beam in (1, 10):
            for bbd in (False, True):
                if chunk == 25600 and (beam == 1 or bbd):
                    continue
                name = f"tiny_c{chunk}_b{beam}_bbd{int(bbd)}"
                steps = 16 if (chunk == 10240 and beam == 10 and not bbd) else 0
                s2t, rec, calls = run_stream(tiny_dir, audio, chunk, beam, bbd, record_steps=steps)
                meta = {"model": "TINY", "seed": 1234, "stats": "meanstd", "audio_stream": 0,
                        "n_samples": len(audio), "chunk": chunk, "beam": beam, "bbd": bbd}
                save_case(name, meta, rec, calls, with_tensors=(beam == 10 and not bbd), with_steps=steps > 0)
                print(name, "blocks", len(rec.blocks), "final", calls[-1]["results"][:1])

    # float64 stats path (G2) + finalize_all False (A5)
    s2t, rec, calls = run_stream(tiny64_dir, audio, 10240, 5, False, finalize_all=False)
    save_case("tiny_stats64_b5", {"model": "TINY", "seed": 1234, "stats": "sums", "audio_stream": 0,
                                  "n_samples": len(audio), "chunk": 10240, "beam": 5, "bbd": False,
                                  "finalize_all": False}, rec, calls, with_tensors=True)

    # short utterances: short-segment path (A13), single final call, sub-window chunks
    short = {}
    for n in (3000, 9000, 20000):
        a = synth.synth_audio(3, n)
        s2t = Speech2TextStreaming(tiny_dir, beam_size=5, device="cpu")
        rec = Recorder(s2t)
        res = s2t(a, is_final=True, finalize_all=True)
        short[str(n)] = {"results": results_to_json(res), "blocks": rec.blocks,
                         "enc_len": [int(e.shape[0]) for e in rec.enc_outs]}
        np.savez_compressed(OUT / f"tiny_short_{n}.npz", enc=rec.enc_outs[0] if rec.enc_outs else np.zeros((0, 64)))
    # too-short final -> exception type (A3)
    a = synth.synth_audio(3, 700)
    s2t = Speech2TextStreaming(tiny_dir, beam_size=5, device="cpu")
    try:
        s2t(a, is_final=True, finalize_all=True)
        short["700_exc"] = None
    except Exception as e:  # noqa: BLE001
        short["700_exc"] = type(e).__name__
    # degenerate 640-sample chunks (A2/A3): nothing is ever encoded
    a = synth.synth_audio(4, 6400)
    s2t = Speech2TextStreaming(tiny_dir, beam_size=5, device="cpu")
    rec = Recorder(s2t)
    outs = []
    for i in range(0, 6400 - 640, 640):
        outs.append(results_to_json(s2t(a[i:i + 640], is_final=False)))
    short["640"] = {"results": outs, "enc_calls": len(rec.feats)}
    try:
        s2t(a[6400 - 640:], is_final=True, finalize_all=True)
        short["640_final_exc"] = None
    except Exception as e:  # noqa: BLE001
        short["640_final_exc"] = type(e).__name__
    (OUT / "tiny_short.json").write_text(json.dumps(short))

    # reset(): second utterance on the same object (stale CTC table quirk)
    s2t = Speech2TextStreaming(tiny_dir, beam_size=5, device="cpu")
    rec = Recorder(s2t)
    a1 = synth.synth_audio(5, 40000)
    a2 = synth.synth_audio(6, 50000)
    for a in (a1, a2):
        s2t.reset()
        pos = 0
        while pos < len(a):
            end = min(pos + 10240, len(a))
            res = s2t(a[pos:end], is_final=end >= len(a), finalize_all=end >= len(a))
            pos = end
    (OUT / "tiny_reset.json").write_text(json.dumps({"blocks": rec.blocks, "final": results_to_json(res)}))

    # ---------------- XL dims: ids / scores / encoder output ----------------
    xl_dir = synth.write_model_dir(tmp / "xl", XL, seed=1234, stats_kind="meanstd")
    audio = synth.synth_audio(0, 16000 * 8)
    for bbd in (True, False):
        s2t, rec, calls = run_stream(xl_dir, audio, 10240, 10, bbd, record_steps=4 if not bbd else 0)
        name = f"xl_c10240_b10_bbd{int(bbd)}"
        meta = {"model": "XL", "seed": 1234, "stats": "meanstd", "audio_stream": 0,
                "n_samples": len(audio), "chunk": 10240, "beam": 10, "bbd": bbd}
        save_case(name, meta, rec, calls, with_tensors=not bbd, with_steps=not bbd)
        print(name, "blocks", len(rec.blocks))
    print("done ->", OUT)


def after_final():
    """The reference SERVER never resets its model: calls go on after is_final=True
    (speechcatcher_server.py:270).  ``python tools/gen_golden.py --after-final`` records a stream whose
    4th and 8th of 10 calls are final, no reset() in between: per-call results and block trajectories."""
    OUT.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = Path(tempfile.mkdtemp(prefix="golden_"))
    tiny_dir = synth.write_model_dir(tmp / "tiny", TINY, seed=1234, stats_kind="meanstd")
    for bbd in (True, False):
        s2t = Speech2TextStreaming(tiny_dir, beam_size=3, ctc_weight=0.3, device="cpu", use_bbd=bbd)
        rec = Recorder(s2t)
        a = synth.synth_audio(5, 10240 * 10)
        calls = []
        for i in range(10):
            fin = i in (3, 7)
            nb0 = len(rec.blocks)
            res = s2t(a[i * 10240:(i + 1) * 10240], is_final=fin)
            calls.append({"is_final": fin, "n_blocks": len(rec.blocks) - nb0, "results": results_to_json(res),
                          "enc_buffer_len": int(s2t.beam_search.encoder_buffer.shape[1])
                          if s2t.beam_search.encoder_buffer is not None else 0,
                          "processed_block": int(s2t.beam_search.processed_block)})
        (OUT / f"tiny_after_final_bbd{int(bbd)}.json").write_text(json.dumps({"blocks": rec.blocks, "calls": calls}))
        print("after_final bbd", bbd, "blocks", len(rec.blocks), [len(c["results"]) for c in calls])


def xl_extra():
    """XL dims at the other two chunk sizes of SURVEY 8(d) (trajectories only: ids / positions / scores):
    25 600-sample calls (2-3 encoder blocks and decode blocks per call) without BBD, and the CLI's 8 192 with BBD.
    plus the reference CLI's default search (beam 5, BBD on, 8 192) and greedy search.
    ``python tools/gen_golden.py --xl-extra [case names]`` writes just these."""
    OUT.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = Path(tempfile.mkdtemp(prefix="golden_"))
    xl_dir = synth.write_model_dir(tmp / "xl", XL, seed=1234, stats_kind="meanstd")
    only = [a for a in sys.argv[1:] if a.startswith("xl_")]
    for chunk, beam, bbd, stream in ((25600, 10, False, 3), (8192, 10, True, 4),
                                     (8192, 5, True, 5),      # the reference CLI's defaults: -b 5, BBD on, 8192
                                     (10240, 1, False, 6)):   # greedy
        name = f"xl_c{chunk}_b{beam}_bbd{int(bbd)}"
        if only and name not in only:
            continue
        audio = synth.synth_audio(stream, 16000 * 8)
        s2t, rec, calls = run_stream(xl_dir, audio, chunk, beam, bbd)
        meta = {"model": "XL", "seed": 1234, "stats": "meanstd", "audio_stream": stream,
                "n_samples": len(audio), "chunk": chunk, "beam": beam, "bbd": bbd}
        save_case(name, meta, rec, calls, with_tensors=False)
        print(name, "blocks", len(rec.blocks))


def ctc_weights():
    """``Speech2TextStreaming(ctc_weight=...)`` is part of the surface (speech2text_streaming.py:143-150); with
    ``ctc_weight <= 0`` the reference builds NO CTC scorer at all (beam_search.py:925: decoder-only search, no "ctc"
    entry in Hypothesis.scores).  ``python tools/gen_golden.py --ctc-weights`` records tiny and XL trajectories at
    ctc_weight 0.0 and 0.5 (the other fixtures all use 0.3)."""
    OUT.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = Path(tempfile.mkdtemp(prefix="golden_"))
    tiny_dir = synth.write_model_dir(tmp / "tiny", TINY, seed=1234, stats_kind="meanstd")
    xl_dir = synth.write_model_dir(tmp / "xl", XL, seed=1234, stats_kind="meanstd")
    for tag, mdir, stream, n in (("tiny", tiny_dir, 0, 16000 * 6 + 3217), ("xl", xl_dir, 7, 16000 * 8)):
        audio = synth.synth_audio(stream, n)
        for cw, bbd in ((0.0, False), (0.5, False), (0.5, True), (0.0, True)):
            if tag == "xl" and bbd:
                continue
            name = f"{tag}_c10240_b10_bbd{int(bbd)}_cw{int(round(cw * 10)):02d}"
            s2t, rec, calls = run_stream(mdir, audio, 10240, 10, bbd, ctc_weight=cw)
            meta = {"model": tag.upper(), "seed": 1234, "stats": "meanstd", "audio_stream": stream,
                    "n_samples": len(audio), "chunk": 10240, "beam": 10, "bbd": bbd, "ctc_weight": cw}
            save_case(name, meta, rec, calls, with_tensors=False)
            print(name, "blocks", len(rec.blocks), "final", calls[-1]["results"][:1])


if __name__ == "__main__":
    if "--ctc-weights" in sys.argv[1:]:
        ctc_weights()
    elif "--xl-extra" in sys.argv[1:]:
        xl_extra()
    elif "--after-final" in sys.argv[1:]:
        after_final()
    else:
        main()
