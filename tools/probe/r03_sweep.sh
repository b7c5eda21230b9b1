cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r03_sweep.log
for v in - UVO_A_OVERLAP=3 UVO_MAX_B=4 UVO_A_OVERLAP=3,UVO_MAX_B=4 UVO_MAX_B=2; do
  timeout -k 10 200 python tools/probe/ab_env.py $v >> gpurun_out/r03_sweep.log 2>&1 || exit 1
done
for d in 5 7 8; do
  AB_DEPTH=$d timeout -k 10 200 python tools/probe/ab_env.py AB_DEPTH=$d >> gpurun_out/r03_sweep.log 2>&1 || exit 1
done
AB_DEPTH=8 timeout -k 10 200 python tools/probe/ab_env.py AB_DEPTH=8,UVO_A_OVERLAP=3,UVO_MAX_B=4 >> gpurun_out/r03_sweep.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r03_sweep.log
