cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_codec.py tests/test_shim.py -q -x > gpurun_out/r03_codec_tests.log 2>&1
rc=$?; tail -12 gpurun_out/r03_codec_tests.log
exit $rc
