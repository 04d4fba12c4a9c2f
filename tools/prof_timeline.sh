#!/bin/bash
# kernel trace of a short default bench run -> per-kernel stats + the timeline of one full-bucket decode iteration
# usage (GPU box): bash tools/prof_timeline.sh <tag> [extra bench args]
TAG=${1:-r04}
shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--no-cpu-baseline --no-single-stream --no-other-mode --no-resident --no-long-context --steps 8 --roofline-steps 0 $@"
rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pk -- python3 bench.py $ARGS > gpurun_out/${TAG}_prof_kernel.log 2>&1
DB=$(find /tmp/pk -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/${TAG}_kernel_stats.csv > /dev/null
for w in full hpw4 -300; do python tools/rocpd_timeline.py $DB $w 120 > gpurun_out/${TAG}_timeline_$w.txt 2>&1; done
python tools/rocpd_busy.py $DB > gpurun_out/${TAG}_busy.txt 2>&1
# ... and of the timed window alone (8 steps of ~25 ms at the end of the run, behind the lock-step pre-roll and the warm-up)
python tools/rocpd_busy.py $DB 180ms > gpurun_out/${TAG}_busy_timed_window.txt 2>&1
head -30 gpurun_out/${TAG}_kernel_stats.csv
