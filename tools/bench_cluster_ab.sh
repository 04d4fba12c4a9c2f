cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_native.py -m gpu -q -x -k "not fp16 and not two_ranks and not cli" 2>&1 | tail -4
B="timeout 300 python bench.py --no-cpu-baseline --no-deferred --roofline-steps 0"
$B > gpurun_out/r02/cl_on.json 2> gpurun_out/r02/cl.err
SC_TEST_HOOKS=1 SC_DEC_CLUSTER=0 $B > gpurun_out/r02/cl_off.json 2>> gpurun_out/r02/cl.err
