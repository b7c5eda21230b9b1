cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --timed-only > gpurun_out/r03_c5_w5_$i.json 2> gpurun_out/r03_c5_w5_$i.err || exit 1
  cat gpurun_out/r03_c5_w5_$i.json
done
UVO_TRACE=gpurun_out/r03_c5_trace_w5.csv timeout -k 10 200 python bench.py --steps 20 --warmup 5 --timed-only > gpurun_out/r03_c5_w5_t.json 2> gpurun_out/r03_c5_w5_t.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 40 --timed-only || exit 1
