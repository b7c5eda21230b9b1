cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "integral or surf" > gpurun_out/r03_q_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r03_q_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03q_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03q_sync.log 2>&1 || exit 1
for i in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --timed-only || exit 1; done
timeout -k 10 200 python bench.py --steps 600 --timed-only || exit 1
