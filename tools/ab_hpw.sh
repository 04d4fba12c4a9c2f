#!/bin/bash
# A/B of the decoder-layer forms on the default bench window (run on the GPU box): one head per workgroup + six-launch
# large buckets (round 3), four / two heads per workgroup at every size, and the default dispatch
B="python bench.py --no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --roofline-steps 0"
export SC_TEST_HOOKS=1
for h in 1 4 2; do SC_DEC_HPW=$h $B > gpurun_out/r04_ab_hpw$h.json 2> gpurun_out/r04_ab_hpw$h.err; done
$B > gpurun_out/r04_ab_default.json 2> gpurun_out/r04_ab_default.err
python - <<'PY'
import json
for n in ("hpw1", "hpw4", "hpw2", "default"):
    try:
        d = json.load(open("gpurun_out/r04_ab_%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d["continuous"])
    except Exception as e:
        print(n, "ERR", e)
PY
