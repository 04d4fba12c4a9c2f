#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export SC_TEST_HOOKS=1
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0"
for cfg in "split51:SC_DEC_FFN_FORCE=1000,5,1" "pro:SC_DEC_FFN_SPLIT=0"; do
  name=${cfg%%:*}; export ${cfg##*:}
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/r06_${name}_prof.log 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1)
  python tools/rocpd_timeline.py $DB hpw4 60 > gpurun_out/r06_${name}_timeline_full.txt 2>&1
  unset SC_DEC_FFN_FORCE SC_DEC_FFN_SPLIT
done
grep -A12 "per kernel over the step" gpurun_out/r06_split51_timeline_full.txt
grep -A10 "per kernel over the step" gpurun_out/r06_pro_timeline_full.txt
