cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -q -x -k "submit_blocks" > gpurun_out/r03_c25.log 2>&1
rc=$?; tail -5 gpurun_out/r03_c25.log; exit $rc
