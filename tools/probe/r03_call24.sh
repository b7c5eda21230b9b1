cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "not c5" > gpurun_out/r03_c24_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r03_c24_tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert" gpurun_out/r03_c24_tests.log | head; exit $rc; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03f_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03f_sync.log 2>&1 || exit 1
timeout -k 10 200 python tools/probe/ab_env.py - 2>/dev/null || exit 1
