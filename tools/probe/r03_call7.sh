cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for part in 1 2; do
UVO_DESC_PART=$part rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_r03a_desc${part}_sq1 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_r03a_desc${part}_sq1.log 2>&1 || exit 1
UVO_DESC_PART=$part rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_r03a_desc${part}_sq2 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_r03a_desc${part}_sq2.log 2>&1 || exit 1
done
echo ok
