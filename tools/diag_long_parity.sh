#!/bin/bash
# diagnostics for a divergence in the 128-stream long run: which kernel forms reproduce the oracle
export SC_TEST_HOOKS=1
echo "== XL single stream, 34 chunks, one-head kernels WITHOUT prefetch (SC_ATTN_DEEP=0): L up to 406 = two 256-position lists"
SC_ATTN_DEEP=0 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "xl_long_run" 2>&1 | tail -2
for m in 100000 0; do
  echo "== headline-regime test with SC_HPW_MIN=$m"
  SC_HPW_MIN=$m python -m pytest tests/test_gpu_baseline_size.py -m gpu -x -q -k headline 2>&1 | grep -E "passed|failed|AssertionError|process_idx = |block = " | cut -c1-200 | head -6
done
