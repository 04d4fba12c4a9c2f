#!/bin/bash
# Generic A/B of library test hooks on the default bench window (run on the GPU box):
#   bash tools/ab_env.sh <out file> "<label>|<ENV=.. ENV=..>|<extra bench args>" ...
# every configuration runs the headline leg only; one line per configuration: label, audio-s/s, ms per step, iterations per step
OUT=$1; shift
B="python bench.py --no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --roofline-steps 0"
export SC_TEST_HOOKS=1
: > $OUT
for cfg in "$@"; do
  IFS='|' read -r label envs extra <<< "$cfg"
  env $envs $B $extra > /tmp/ab_one.json 2> /tmp/ab_one.err
  python - "$label" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab_one.json"))
    print(sys.argv[1], d["value"], d["ms_per_step"], d["continuous"]["iterations_per_step"])
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/ab_one.err").read()[-300:])
PY
done
cat $OUT
