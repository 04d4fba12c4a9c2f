#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_native.py -q -x -k "sub_window" 2>&1 | tail -3
python - <<'PY'
import os, sys, json
os.environ["SC_TEST_HOOKS"]="1"
sys.path.insert(0, ".")
import bench, numpy as np, torch
from speechcatcher_amd.config import L_LIKE
for fused in ("1", "0"):
    os.environ["SC_DEC_FUSED"] = fused
    w = bench.make_weights("cuda:0", cfg=L_LIKE)
    S, preroll, warm, steps = 128, 21, 5, 20
    total = preroll + warm + steps + bench.SERVED_SPARE
    audio = bench.make_audio(S, total)
    sb, r = bench.measure(w, audio, S, 10, False, preroll, warm, steps, 16, "continuous", total)
    sb.close()
    print("l_like fused" if fused == "1" else "l_like six-launch", round(r["value"], 1), "audio-s/s", round(r["elapsed"] / steps * 1e3, 2), "ms/step", flush=True)
    del sb, w
PY
