# Final measurement pass of a round: bench line, kernel stats (pipelined and synchronous), HBM counters of the roofline kernel,
# the secondary configurations.  Outputs under gpurun_out/; tools/pmc_summary.py copies the summaries into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 200 --no-cpu-baseline > gpurun_out/prof_r01.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01_sync -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_r01_sync.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_write.log 2>&1
python tools/bench_configs.py > gpurun_out/bench_configs.json 2> gpurun_out/bench_configs.err
tail -c 900 gpurun_out/bench_configs.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_mfma -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_mfma.log 2>&1
