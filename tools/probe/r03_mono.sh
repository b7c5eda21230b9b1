cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_mono_contract -- python3 tools/prof_mono.py contract 32 > gpurun_out/prof_r03_mono_contract.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_mono_1px -- python3 tools/prof_mono.py 1px 32 > gpurun_out/prof_r03_mono_1px.log 2>&1 || exit 1
tail -1 gpurun_out/prof_r03_mono_contract.log
