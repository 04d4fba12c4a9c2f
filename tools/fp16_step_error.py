#!/usr/bin/env python3
"""Per decode step error of the fp16 mode (BASELINE configs[4]; opt-in, never the parity mode) against the fp32 engine ON THE SAME
PREFIX - VERDICT r5 item 8: "a number, not a head-count".

Two batches of the Python engine over the HIP kernels: A (fp32: the reference's arithmetic) drives the search; B is a SHADOW in
the mode under test.  Before every decode step B's whole search state is overwritten with A's (hypotheses, scores, ancestor
tables, CTC state, control rows, the self-attention K|V rows and - once per encoder call - the encoder-side tables: the K|V rows
and tables are CONVERTED to B's storage type by the copy), then both run sc_decode_step and what the step produced is compared:
    max |d log-prob|   over the vocabulary entries of the live hypothesis rows that can matter (fp32 log-prob > -20)
    max |d fused score| over the W x W candidates of every stream whose candidate token agrees (they do, almost always)
i.e. one step's worth of fp16 arithmetic, teacher-forced: no drift, no path divergence.  Modes (component by component):
    kv16     fp16 K|V storage only                          ffn16   + fp16 feed-forward weights / MFMA inputs (decoder side here)
    dec3     fp16 K|V + fp16 decoder projections + fp16 partial products (sc_search.act_half = 3)
    float16  all of it = the mode bench.py's fp16 leg runs
The encoder side of the mode (fp16 feed-forward and attention projections of the 30 encoder layers) is measured separately
as the error of the CTC log-posteriors and of the encoder output of one call (`encoder_error`).
Usage (GPU box): python tools/fp16_step_error.py [streams=8] [chunks=12] > profiles/r06_fp16_step_error.txt"""
import os
import sys
os.environ["SC_TEST_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speechcatcher_amd import synth  # noqa: E402
from speechcatcher_amd.hip_backend import HipBackend  # noqa: E402

CHUNK = 10240
MODES = {"kv16": dict(kv_dtype="float16"),
         "ffn16": dict(ffn_dtype="float16"),
         "dec3": dict(kv_dtype="float16", dec_dtype="float16"),
         "float16": dict(kv_dtype="float16", dec_dtype="float16", ffn_dtype="float16", proj_dtype="float16")}


class ShadowBackend(HipBackend):
    """backend of batch A: every decode step also runs on the shadow batch B from A's state"""

    def __init__(self, device):
        super().__init__(device)
        self.use_graphs = False
        self.b = None
        self.be_b = None
        self.worst = {"dlogp": 0.0, "dscore": 0.0, "steps": 0, "cand_tok_mismatch": 0, "cand_total": 0}

    def attach(self, sb_a, sb_b, be_b):
        self.a, self.b, self.be_b = sb_a, sb_b, be_b
        self.names = [k for k, v in vars(sb_a).items() if isinstance(v, torch.Tensor) and isinstance(getattr(sb_b, k, None), torch.Tensor)
                      and getattr(sb_b, k).shape == v.shape]

    def decode_step(self, sb):
        a, b = self.a, self.b
        for k in self.names:                       # (dtype conversion - fp32 -> fp16 K|V rows - by copy_)
            getattr(b, k).copy_(getattr(a, k))
        b.n_rows_step = a.n_rows_step
        super().decode_step(sb)
        self.be_b.decode_step(b)
        torch.cuda.synchronize()
        ctrl = a.ctrl.cpu().numpy()
        W = a.W
        for s in range(a.S):
            act, cur, fin, T, L, nh, has, _ = [int(v) for v in ctrl[s]]
            if not act or nh <= 0:
                continue
            rows = slice(s * W, s * W + nh)
            la, lb = a.logp[rows], b.logp[rows]
            m = la > -20.0
            self.worst["dlogp"] = max(self.worst["dlogp"], float((la - lb).abs()[m].max()))
            ta, tb = a.cand_tok[rows], b.cand_tok[rows]
            same = ta == tb
            self.worst["cand_tok_mismatch"] += int((~same).sum())
            self.worst["cand_total"] += int(same.numel())
            if same.any():
                self.worst["dscore"] = max(self.worst["dscore"], float((a.cand_score[rows] - b.cand_score[rows]).abs()[same].max()))
        self.worst["steps"] += 1


def run_mode(mode, S=8, n=12, beam=10):
    from test_engine_spec import make_batch
    kw = dict(n_streams=S, max_frames=16 * n + 80, max_tokens=16 * n + 40, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
    sh = ShadowBackend("cuda:0")
    a = make_batch("XL", 1234, "meanstd", beam, False, backend=sh, device="cuda:0", **kw)
    be_b = HipBackend("cuda:0")
    be_b.use_graphs = False
    b = make_batch("XL", 1234, "meanstd", beam, False, backend=be_b, device="cuda:0", **dict(kw, **MODES[mode]))
    sh.attach(a, b, be_b)
    audio = np.stack([synth.synth_audio(4000 + s, CHUNK * n) for s in range(S)])
    for k in range(n):
        a.push([(s, audio[s, k * CHUNK:(k + 1) * CHUNK], False) for s in range(S)])
    return sh.worst


def encoder_error(S=4, n=6):
    """the encoder side of the fp16 mode on its own: the same chunks through an fp32 and an fp16-mode batch (native engine),
    error of the encoder output rows and of the CTC log-posteriors the search reads"""
    from test_engine_spec import make_batch
    kw = dict(n_streams=S, max_frames=16 * n + 80, max_tokens=200, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
    audio = np.stack([synth.synth_audio(4000 + s, CHUNK * n) for s in range(S)])
    outs = {}
    for name, extra in (("f32", {}), ("f16", dict(ffn_dtype="float16", proj_dtype="float16"))):
        be = HipBackend("cuda:0")
        be.use_graphs = False
        sb = make_batch("XL", 1234, "meanstd", 10, False, backend=be, device="cuda:0", **dict(kw, **extra))
        for k in range(n):
            sb.push([(s, audio[s, k * CHUNK:(k + 1) * CHUNK], False) for s in range(S)])
        T = min(st.T_enc for st in sb.st)
        V = sb.cfg.vocab_size
        outs[name] = (sb.enc.view(S, -1, sb.cfg.d_model)[:, :T].float().cpu(), sb.ctcx.view(S, -1, V)[:, :24].float().cpu(), T)
    (ea, ca, T), (eb, cb, _) = outs["f32"], outs["f16"]
    m = ca > -20.0
    return {"frames": int(T), "max_abs_encoder_output_diff": float((ea - eb).abs().max()), "encoder_output_scale": float(ea.abs().max()),
            "max_abs_ctc_logposterior_diff_first_block": float((ca - cb).abs()[m].max())}


if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    print(f"# per decode step error of the fp16 mode against the fp32 engine on the same prefix (tools/fp16_step_error.py): XL dims, {S} streams, "
          f"{n} chunks of {CHUNK} samples, beam 10")
    for mode in MODES:
        w = run_mode(mode, S, n)
        print(f"{mode:8s} steps {w['steps']:4d}  max |d log-prob| {w['dlogp']:.3e}  max |d fused candidate score| {w['dscore']:.3e}  "
              f"candidate tokens that differ {w['cand_tok_mismatch']} of {w['cand_total']}")
    print("encoder side:", encoder_error())
