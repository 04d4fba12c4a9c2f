"""Shared helpers for the parity tests (oracle side)."""
import functools

import numpy as np
import torch

from speechcatcher_amd import synth
from speechcatcher_amd.config import L_LIKE, MICRO, TINY, XL
from speechcatcher_amd.mel import melscale_fbanks_slaney

CFGS = {"TINY": TINY, "XL": XL, "MICRO": MICRO, "L_LIKE": L_LIKE}


@functools.lru_cache(maxsize=4)
def oracle_model(cfg_name="TINY", seed=1234, stats="meanstd"):
    from oracle.ref_port import RefPortModel
    cfg = CFGS[cfg_name]
    sd = synth.make_state_dict(cfg, seed)
    mean, std = synth.stats_to_mean_std(synth.make_stats(cfg, kind=stats))
    mel = melscale_fbanks_slaney(cfg.n_fft // 2 + 1, 0.0, cfg.sample_rate / 2.0, cfg.n_mels, cfg.sample_rate)
    return RefPortModel(sd, cfg, mel, mean, std)


def run_oracle_stream(model, audio, chunk, beam, bbd, finalize_all=True, ctc_weight=0.3, **kw):
    from oracle.ref_port import RefPortStreaming
    s = RefPortStreaming(model, beam_size=beam, ctc_weight=ctc_weight, use_bbd=bbd, **kw)
    s.trace = []
    feats, encs, calls = [], [], []
    pos, n = 0, len(audio)
    while pos < n:
        end = min(pos + chunk, n)
        fin = end >= n
        nb0 = len(s.trace)
        s.last_feats = None
        s.last_enc_out = None
        res = s(audio[pos:end], is_final=fin, finalize_all=finalize_all and fin)
        if s.last_feats is not None and s.last_feats.size(1) >= 3:
            feats.append(s.last_feats[0].numpy())
            encs.append(s.last_enc_out[0].numpy())
        calls.append({"results": res, "n_blocks": len(s.trace) - nb0})
        pos = end
    return s, feats, encs, calls


def _oracle_calls_worker(args):
    """one stream through the oracle, call by call (a fresh interpreter of a spawn pool): the live hypotheses after
    every call as plain lists"""
    cfg_name, seed, stats, stream_seed, n_samples, chunk, beam, bbd, threads = args
    torch.set_num_threads(threads)
    from oracle.ref_port import RefPortStreaming
    ora = RefPortStreaming(oracle_model(cfg_name, seed, stats), beam_size=beam, use_bbd=bbd)
    audio = synth.synth_audio(stream_seed, n_samples)
    calls = []
    for pos in range(0, n_samples, chunk):
        ora.margins = []
        ora(audio[pos:pos + chunk], is_final=False)
        ref = ora.running_hyps or []
        calls.append({"yseq": [list(h.yseq) for h in ref], "xpos": [list(h.xpos) for h in ref],
                      "score": [float(h.score) for h in ref],
                      "score_dec": [float(h.scores.get("decoder", 0.0)) for h in ref],
                      "score_ctc": [float(h.scores.get("ctc", 0.0)) for h in ref],
                      "process_idx": int(ora.process_idx),
                      "min_margin": float(min(ora.margins)) if ora.margins else float("inf"),
                      "T": 0 if ora.encoder_buffer is None else int(ora.encoder_buffer.shape[1])})
    return calls


def oracle_calls_parallel(cfg_name, stream_seeds, n_samples, chunk, beam, bbd, threads=8):
    """The oracle run SOLO on several streams at once (one spawned process per stream, `threads` intra-op threads each):
    {stream seed: [per-call snapshot]}.  Non-final calls only (the bench regime)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")      # never fork a process that holds a GPU context
    with ctx.Pool(len(stream_seeds)) as pool:
        res = pool.map(_oracle_calls_worker, [(cfg_name, 1234, "meanstd", sd, n_samples, chunk, beam, bbd, threads)
                                              for sd in stream_seeds])
    return dict(zip(stream_seeds, res))
