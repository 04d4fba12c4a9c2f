#!/bin/bash
# A/B of the decoder-layer forms on the default bench window (run on the GPU box): the row threshold from which the
# head-parallel layers run four heads per workgroup (SC_HPW_MIN; 100000 = never: round 3's dispatch)
B="python bench.py --no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --roofline-steps 0"
export SC_TEST_HOOKS=1
for m in 100000 0 160 320 640; do SC_HPW_MIN=$m $B > gpurun_out/r04_ab_hpwmin$m.json 2> gpurun_out/r04_ab_hpwmin$m.err; done
python - <<'PY'
import json
for n in (100000, 0, 160, 320, 640):
    try:
        d = json.load(open("gpurun_out/r04_ab_hpwmin%d.json" % n))
        print("SC_HPW_MIN", n, d["value"], d["ms_per_step"], d["continuous"]["iterations_per_step"])
    except Exception as e:
        print(n, "ERR", e)
PY
