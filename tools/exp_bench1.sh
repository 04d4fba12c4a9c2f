#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py > gpurun_out/r06_bench_default_try1.json 2> gpurun_out/r06_bench_default_try1.err ) 2> gpurun_out/r06_bench_default_try1.time
tail -3 gpurun_out/r06_bench_default_try1.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_bench_default_try1.json"))
print("value", d["value"], "exact", d["value_exact_steps"], "ms", d["ms_per_step"])
print("roof", {k: d["roofline"][k] for k in ("kernel","frac","avg_launch_us","bound")})
print("wide", d["wider_batches"])
print("strict", d["strict_lock_step"]["value"], "single", d["single_stream"], "fp16", d["fp16_mode"]["value"])
for e in d["roofline"]["per_kernel"]: print(e["kernel"][:40], e["launches"], e["avg_launch_us"], e["bound"], e["frac_of_bound"])
PY
python -m pytest tests/test_gpu_ops.py -x -q -k "split_scan or ctc_prefix_scan_long" 2>&1 | tail -3
