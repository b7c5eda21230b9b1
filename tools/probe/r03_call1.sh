# round 3, GPU call 1: the GPU suite, then the wait-policy / stream-priority A/B and a traced 20-step run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/r03_c1_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r03_c1_tests.log
if [ $rc -gt 1 ]; then echo "tests rc $rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/probe/ab_env.py - UVO_WORKER_WAIT=spin UVO_PNP_PRIORITY=0 UVO_WORKER_WAIT=block-all > gpurun_out/r03_c1_ab.log 2>&1 || exit 1
cat gpurun_out/r03_c1_ab.log
UVO_TRACE=gpurun_out/r03_c1_trace20.csv timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_c1_bench20.json 2> gpurun_out/r03_c1_bench20.err || exit 1
tail -c 1500 gpurun_out/r03_c1_bench20.json
exit $rc
