cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r03_c13_ab.log
for v in UVO_PNP_SPEC=2 UVO_PNP_SPEC=1,UVO_A_OVERLAP=3 UVO_PNP_SPEC=1,UVO_WORKER_WAIT=spin; do timeout -k 10 200 python tools/probe/ab_env.py $v >> gpurun_out/r03_c13_ab.log 2>&1 || exit 1; done
grep -v amdgpu.ids gpurun_out/r03_c13_ab.log
