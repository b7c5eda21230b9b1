#!/bin/bash
# Hessian kernel A/B: detector parity tests, then kernel times of a synchronous C3 run (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04_hess}
python -m pytest tests/test_gpu_parity.py -x -q -k "surf or hessian or detect or integral or stereo_step or pipelined" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_${T}_sync.log 2>&1
python tools/probe/kstats.py prof_${T}_sync 12
