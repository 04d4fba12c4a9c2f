#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh "<bench args>" build_ab/libscasr_a.so build_ab/libscasr_b.so ...
# (every variant is selected through SC_LIB_VARIANT - read by speechcatcher_amd/_abi.py under SC_TEST_HOOKS=1 only; the product
#  library in the tree is never overwritten;  REPS=n repeats, interleaved: a b c a b c ... so that a drift of the
#  box hits every variant alike).  "product" as a library name = the in-tree build.
ARGS=$1; shift
export SC_TEST_HOOKS=1
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset SC_LIB_VARIANT; else export SC_LIB_VARIANT=$lib; fi
    python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d.get('strict_lock_step') or d.get('continuous') or {}
print('$lib', 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'other', o.get('value'), 'single', (d.get('single_stream') or {}).get('ms_per_hop'))"
  done
done
