#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export SC_TEST_HOOKS=1
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0"
for cu in 0 128; do
  export SC_ENC_CUS=$cu
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/r06_stream_cu${cu}_prof.log 2>&1
  DB=$(find /tmp/pk -name "*.db" | head -1)
  python tools/rocpd_phases.py $DB 150 detail > gpurun_out/r06_stream_cu${cu}_phases.txt 2>&1
  python tools/rocpd_stats.py $DB gpurun_out/r06_stream_cu${cu}_kernel_stats.csv > /dev/null
  python tools/rocpd_timeline.py $DB full 120 > gpurun_out/r06_stream_cu${cu}_timeline_full.txt 2>&1
  python tools/rocpd_merged.py $DB 100 6 > gpurun_out/r06_stream_cu${cu}_merged.txt 2>&1
done
