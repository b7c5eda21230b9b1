cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "surf or detect or descriptor or stereo" > gpurun_out/r03_c8_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r03_c8_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -q -x -k "c3 or c2" > gpurun_out/r03_c8_tests2.log 2>&1
rc=$?; tail -4 gpurun_out/r03_c8_tests2.log
if [ $rc -ne 0 ]; then exit $rc; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03b_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03b_sync.log 2>&1 || exit 1
UVO_DESC_PART=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03b_big -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03b_big.log 2>&1 || exit 1
echo ok
