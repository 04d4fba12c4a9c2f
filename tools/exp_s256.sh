#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export SC_TEST_HOOKS=1
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0 --streams 256"
rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/r06_s256_prof.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r06_s256_kernel_stats.csv > /dev/null
python tools/rocpd_timeline.py $DB full 50 > gpurun_out/r06_s256_timeline_full.txt 2>&1
python tools/rocpd_phases.py $DB 200 detail > gpurun_out/r06_s256_phases.txt 2>&1
head -12 gpurun_out/r06_s256_kernel_stats.csv | cut -c1-140
grep -A14 "per kernel over the step" gpurun_out/r06_s256_timeline_full.txt
head -5 gpurun_out/r06_s256_phases.txt
