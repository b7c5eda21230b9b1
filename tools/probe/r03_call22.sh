cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_configs.py > gpurun_out/bench_r03_configs.json 2> gpurun_out/bench_r03_configs.err || { tail -5 gpurun_out/bench_r03_configs.err; exit 1; }
cat gpurun_out/bench_r03_configs.json
