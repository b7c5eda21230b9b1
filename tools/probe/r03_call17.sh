cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r03_c17_ab.log
for v in UVO_HESS_LDS=60000 UVO_HESS_LDS=60000,UVO_A_OVERLAP=3 UVO_HESS_LDS=81000 UVO_HESS_LDS=81000,UVO_A_OVERLAP=3; do timeout -k 10 200 python tools/probe/ab_env.py $v >> gpurun_out/r03_c17_ab.log 2>&1 || exit 1; done
grep -v amdgpu.ids gpurun_out/r03_c17_ab.log
