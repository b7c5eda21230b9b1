cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
UVO_DBG_BSTAGE=1 UVO_TRACE=gpurun_out/r03_c12_trace_spec.csv timeout -k 10 200 python bench.py --steps 60 --warmup 20 --timed-only > gpurun_out/r03_c12_spec.json 2> gpurun_out/r03_c12_spec.err || exit 1
cat gpurun_out/r03_c12_spec.json; grep uvo gpurun_out/r03_c12_spec.err
