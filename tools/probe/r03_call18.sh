cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r03_c18_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r03_c18_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --timed-only || exit 1; done
timeout -k 10 200 python bench.py --steps 600 --timed-only || exit 1
