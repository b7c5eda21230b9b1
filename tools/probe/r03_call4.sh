cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
UVO_DBG_BSTAGE=1 UVO_TRACE=gpurun_out/r03_c4_trace_w5.csv timeout -k 10 200 python bench.py --steps 20 --warmup 5 --timed-only > gpurun_out/r03_c4_w5.json 2> gpurun_out/r03_c4_w5.err || exit 1
UVO_DBG_BSTAGE=1 UVO_TRACE=gpurun_out/r03_c4_trace_w40.csv timeout -k 10 200 python bench.py --steps 20 --warmup 40 --timed-only > gpurun_out/r03_c4_w40.json 2> gpurun_out/r03_c4_w40.err || exit 1
cat gpurun_out/r03_c4_w5.json gpurun_out/r03_c4_w40.json
