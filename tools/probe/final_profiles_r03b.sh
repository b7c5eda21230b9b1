# Round-3 measurement pass, second half (counters, secondary configurations) (run on the GPU box: gpurun -- 'bash tools/probe/final_profiles_r03.sh').  Outputs under gpurun_out/;
# tools/pmc_summary_r03.py turns them into the summaries committed under profiles/.  Every rocprofv3 line has the program directly
# after `--`, counters are collected in passes of their own (kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r03
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${T}_fetch -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${T}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${T}_write -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${T}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_${T}_sq1 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${T}_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_${T}_sq2 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${T}_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_${T}_mfma -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_${T}_mfma.log 2>&1
echo "pmc done"
python tools/bench_configs.py > gpurun_out/bench_${T}_configs.json 2> gpurun_out/bench_${T}_configs.err
tail -c 600 gpurun_out/bench_${T}_configs.json; echo
