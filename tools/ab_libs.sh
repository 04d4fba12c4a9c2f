#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh "<bench args>" build_ab/libscasr_a.so build_ab/libscasr_b.so ...
# (each variant is copied over speechcatcher_amd/libscasr.so in the box's scratch copy of the repo, then the bench runs)
ARGS=$1; shift
for lib in "$@"; do
  cp $lib speechcatcher_amd/libscasr.so
  for rep in 1 2; do
    python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d.get('strict_lock_step') or d.get('continuous') or {}
print('$lib', 'value', d['value'], 'other', o.get('value'), 'single', (d.get('single_stream') or {}).get('ms_per_hop'))"
  done
done
