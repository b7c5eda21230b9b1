cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/prof_sift.py 20 --cpu > gpurun_out/sift_time.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_sift -- python3 tools/prof_sift.py 5 > gpurun_out/prof_r03_sift.log 2>&1 || exit 1
cat gpurun_out/sift_time.log
