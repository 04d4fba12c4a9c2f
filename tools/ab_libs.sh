#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh "<bench args>" build_ab/libscasr_a.so build_ab/libscasr_b.so ...
# (each variant is copied over speechcatcher_amd/libscasr.so in the box's scratch copy of the repo, then the bench runs;
#  REPS=n repeats, interleaved: a b c a b c ... so that a drift of the box hits every variant alike)
ARGS=$1; shift
cp speechcatcher_amd/libscasr.so /tmp/ab_keep.so
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    cp $lib speechcatcher_amd/libscasr.so
    python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d.get('strict_lock_step') or d.get('continuous') or {}
print('$lib', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'other', o.get('value'), 'single', (d.get('single_stream') or {}).get('ms_per_hop'))"
  done
done
cp /tmp/ab_keep.so speechcatcher_amd/libscasr.so
