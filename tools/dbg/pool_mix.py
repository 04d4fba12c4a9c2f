"""debug: one short final utterance beside a long one at several pool sizes, native vs python engine"""
import os, sys
os.environ["SC_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_case
from test_engine_spec import make_batch
from speechcatcher_amd import synth
from speechcatcher_amd.engine import EngineError
js, _ = load_case("tiny_c10240_b10_bbd0")
meta = js["meta"]
audio = synth.synth_audio(meta["audio_stream"], meta["n_samples"])
short = synth.synth_audio(3, 9000)
kw = dict(max_frames=256, max_tokens=160, pcm_capacity=1 << 18)
for eng in ("native", "python"):
    for rows in (40, 64, 0):
        if eng == "native":
            mix = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], n_streams=2, kv_pool_rows=rows, strict_reference=False, backend="native", **kw)
        else:
            from speechcatcher_amd.hip_backend import HipBackend
            mix = make_batch(meta["model"], meta["seed"], meta["stats"], meta["beam"], meta["bbd"], n_streams=2, kv_pool_rows=rows, strict_reference=False, backend=HipBackend("cuda:0"), device="cuda:0", **kw)
        out = mix.push([(0, audio[:10240], False), (1, short, True)], isolate_faults=True)
        print(eng, rows, {k: (str(v) if isinstance(v, EngineError) else v) for k, v in out.items()},
              "L1", mix.st[1].L, "steps", mix.st[1].n_steps_total, flush=True)
        if not isinstance(out.get(1), EngineError):
            print("   hyps", [len(h["yseq"]) for h in mix.hypotheses(1)])
