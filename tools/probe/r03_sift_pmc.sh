# Counter passes of the SIFT detector (tools/prof_sift.py, 1080p then 640x360), one pass per counter group, kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sift_fetch -- python3 tools/prof_sift.py 4 > gpurun_out/pmc_sift_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_sift_write -- python3 tools/prof_sift.py 4 > gpurun_out/pmc_sift_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_sift_sq1 -- python3 tools/prof_sift.py 4 > gpurun_out/pmc_sift_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sift_sq2 -- python3 tools/prof_sift.py 4 > gpurun_out/pmc_sift_sq2.log 2>&1
echo "sift pmc done"
