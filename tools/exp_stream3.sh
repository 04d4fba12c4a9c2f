#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export SC_TEST_HOOKS=1
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0"
for cfg in "stream128:SC_ENC_CUS=128" "hpw4:SC_DEC_STREAM=0"; do
  name=${cfg%%:*}; export ${cfg##*:}
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/r06_${name}_prof.log 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1)
  python tools/rocpd_gaps.py $DB 120 40 15 > gpurun_out/r06_${name}_gaps.txt 2>&1
  python tools/rocpd_merged.py $DB 100 14 > gpurun_out/r06_${name}_merged14.txt 2>&1
  unset SC_ENC_CUS SC_DEC_STREAM
done
