#!/bin/bash
# strict lock-step (many small and medium compaction buckets) against the row threshold of the four-heads-per-workgroup
# layer kernels (SC_HPW_MIN), and the single-stream leg
B="python bench.py --mode strict --no-cpu-baseline --no-other-mode --no-resident --no-long-context --roofline-steps 0"
export SC_TEST_HOOKS=1
for m in 96 320 480 640 960 100000; do SC_HPW_MIN=$m $B > gpurun_out/r04_ab_strict_hpwmin$m.json 2> gpurun_out/r04_ab_strict_hpwmin$m.err; done
python - <<'PY'
import json
for n in (96, 320, 480, 640, 960, 100000):
    try:
        d = json.load(open("gpurun_out/r04_ab_strict_hpwmin%d.json" % n))
        print("strict SC_HPW_MIN", n, d["value"], d["ms_per_step"], "single", d["single_stream"])
    except Exception as e:
        print(n, "ERR", e)
PY
