cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_siftvo -- python3 tools/bench_configs.py SIFTVO > gpurun_out/prof_siftvo.log 2>&1
tail -1 gpurun_out/prof_siftvo.log
