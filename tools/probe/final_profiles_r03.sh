# Round-3 measurement pass (run on the GPU box: gpurun -- 'bash tools/probe/final_profiles_r03.sh').  Outputs under gpurun_out/;
# tools/pmc_summary_r03.py turns them into the summaries committed under profiles/.  Every rocprofv3 line has the program directly
# after `--`, counters are collected in passes of their own (kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r03
python bench.py > gpurun_out/bench_${T}_final.json 2> gpurun_out/bench_${T}_final.err
tail -c 400 gpurun_out/bench_${T}_final.json; echo
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${T}_20steps.json 2> gpurun_out/bench_${T}_20steps.err
python bench.py --gpus 1 --backend nccl --force-dist --steps 200 --no-cpu-baseline > gpurun_out/bench_${T}_1rank_rccl.json 2> gpurun_out/bench_${T}_1rank_rccl.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T} -- python3 bench.py --steps 200 --timed-only > gpurun_out/prof_${T}.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${T}_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_${T}_sync.log 2>&1
UVO_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/prof_${T}_roctx -- python3 tools/prof_stereo.py 12 > gpurun_out/prof_${T}_roctx.log 2>&1
echo "kernel stats done"
