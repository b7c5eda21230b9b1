# baseline kernel profile of round 3 (synchronous and pipelined) + descriptor parts alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03a_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03a_sync.log 2>&1 || exit 1
UVO_DESC_PART=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03a_big -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03a_big.log 2>&1 || exit 1
UVO_DESC_PART=2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03a_small -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r03a_small.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03a_pipe -- python3 bench.py --steps 200 --timed-only > gpurun_out/prof_r03a_pipe.log 2>&1 || exit 1
python tools/probe/kp_hist.py > gpurun_out/r03_kp_hist.log 2>&1
echo done
