import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import os
os.environ["SC_TEST_HOOKS"] = "1"
import numpy as np
from speechcatcher_amd import synth
from test_engine_spec import make_batch
CHUNK = 10240
S, n, beam = 256, 7, 10
audio = np.stack([synth.synth_audio(900 + s, CHUNK * n) for s in range(S)])
kw = dict(n_streams=S, max_frames=200, max_tokens=160, pcm_capacity=CHUNK * (n + 2), max_chunk_samples=CHUNK)
out = {}
# decN: fp16 K|V caches + the fp16 decoder mode with sc_search.act_half = N (1 layer-kernel projections, 2 partial products
# between the decoder's kernels, 4 output layer; round 4); float16 = everything
MODES = sys.argv[1:] or ["kv16", "ffn16", "proj16", "dec1", "dec2", "dec4", "dec3", "dec7", "float16", "split16"]
for mode in ["float32"] + MODES:
    os.environ["SC_ACT_HALF"] = mode[3:] if mode.startswith("dec") else "7"
    sb = make_batch("XL", 1234, "meanstd", beam, False, backend="native",
                    ffn_dtype="float16" if mode in ("ffn16", "float16") else "split16" if mode == "split16" else "float32",
                    proj_dtype="float16" if mode in ("proj16", "float16") else "split16" if mode == "split16" else "float32",
                    dec_dtype="float16" if mode.startswith("dec") or mode == "float16" else "float32",
                    kv_dtype="float16" if mode in ("kv16", "float16") or mode.startswith("dec") else "float32", **kw)
    ids = np.arange(S, dtype=np.int32)
    for k in range(n):
        sb.push_block(ids, np.ascontiguousarray(audio[:, k * CHUNK:(k + 1) * CHUNK]))
    out[mode] = sb.hypotheses_arrays(list(range(S)))
    sb.close()
a = out["float32"]
def hyp(o, s, j):
    return tuple(o["ids"][s, j, :o["lens"][s, j]].tolist())
for mode in MODES:
    b = out[mode]
    nbest = sum(hyp(a, s, 0) != hyp(b, s, 0) for s in range(S))
    nset = sum(set(hyp(a, s, j) for j in range(beam)) != set(hyp(b, s, j) for j in range(beam)) for s in range(S))
    gaps = []
    for s in range(S):
        if hyp(a, s, 0) != hyp(b, s, 0):
            fa = {hyp(a, s, j): a["score"][s, j] for j in range(beam)}
            hb = hyp(b, s, 0)
            gaps.append((s, round(float(a["score"][s, 0] - fa[hb]), 4) if hb in fa else None, round(float(a["score"][s, 0] - a["score"][s, 1]), 4)))
    print(mode, "best differs:", nbest, "beam set differs:", nset, "max |dscore| best:", float(np.abs(a["score"][:, 0] - b["score"][:, 0]).max()), gaps[:12])
