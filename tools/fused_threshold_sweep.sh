#!/bin/bash
# strict lock-step ms per chunk step with the head-parallel decoder layers used for compaction buckets up to N rows
# (test hook SC_FUSED_MAX); run on the GPU box: bash tools/fused_threshold_sweep.sh > gpurun_out/fused_sweep.txt
export SC_TEST_HOOKS=1
for n in 320 640 800 960 1120 1280; do
  SC_FUSED_MAX=$n python bench.py --mode strict --no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --roofline-steps 0 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused up to $n rows:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms/step')"
done
