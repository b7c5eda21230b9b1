cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "surf or detect or descriptor" > gpurun_out/r03_desc_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r03_desc_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03c_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03c_sync.log 2>&1 || exit 1
UVO_DESC_PART=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03c_big -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03c_big.log 2>&1 || exit 1
UVO_DESC_PART=2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03c_small -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03c_small.log 2>&1 || exit 1
echo ok
