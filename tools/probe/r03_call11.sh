cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x > gpurun_out/r03_c11_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r03_c11_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
: > gpurun_out/r03_c11_ab.log
for v in - UVO_PNP_SPEC=0; do timeout -k 10 200 python tools/probe/ab_env.py $v >> gpurun_out/r03_c11_ab.log 2>&1 || exit 1; done
grep -v amdgpu.ids gpurun_out/r03_c11_ab.log
for i in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --timed-only || exit 1; done
UVO_DBG_BSTAGE=1 timeout -k 10 200 python bench.py --steps 600 --timed-only 2>&1 | grep -v amdgpu || exit 1
