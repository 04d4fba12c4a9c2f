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


def run_oracle_stream(model, audio, chunk, beam, bbd, finalize_all=True, **kw):
    from oracle.ref_port import RefPortStreaming
    s = RefPortStreaming(model, beam_size=beam, ctc_weight=0.3, use_bbd=bbd, **kw)
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
