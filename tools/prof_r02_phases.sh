cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for stop in 0 1 2 3; do
rm -rf /tmp/p$stop; SC_DL_STOP=$stop rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p$stop -- python3 bench.py --defer 0 --steps 4 --warmup 3 --no-cpu-baseline --no-single-stream --roofline-steps 0 > gpurun_out/r02/prof_stop$stop.log 2>&1
DB=$(find /tmp/p$stop -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r02/stop${stop}_kernel_stats.csv > /dev/null
if [ $stop = 0 ]; then for w in -3 -60 -100; do python tools/rocpd_timeline.py $DB $w 60 > gpurun_out/r02/fused_timeline_$w.txt 2>&1; done; fi
done
