cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "match or sift or hamming" > gpurun_out/r03_m_tests.log 2>&1
rc=$?; tail -2 gpurun_out/r03_m_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03m_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03m_sync.log 2>&1 || exit 1
