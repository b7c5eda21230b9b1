# Round-3 closing pass after the SIFT work: the whole GPU suite, the two bench forms, the secondary configurations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; tail -4 gpurun_out/t_all.log
python bench.py > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err || exit 1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_20steps.json 2> gpurun_out/bench_r03_20steps.err || exit 1
python tools/bench_configs.py > gpurun_out/bench_r03_configs.json 2> gpurun_out/bench_r03_configs.err || exit 1
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
