#!/bin/bash
# counters of the per-octave detection kernels (UVO_HESSIAN_SPLIT=1), previous build against the current one; separate --pmc passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export UVO_HESSIAN_SPLIT=1
for v in old new; do
  if [ $v = old ]; then export UVO_HIP_LIB=$GRAFT_REPO_ROOT/ergo_uvo_amd/lib_ab/libuvo_hip_old.so; else unset UVO_HIP_LIB; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_ab_${v}_sq1 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_ab_${v}_sq1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_ab_${v}_sq2 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_ab_${v}_sq2.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_ab_${v}_sq3 -- python3 tools/prof_stereo.py 6 > gpurun_out/pmc_ab_${v}_sq3.log 2>&1 || echo "sq3 pass failed"
  for p in sq1 sq2 sq3; do python tools/probe/pmc_quick.py pmc_ab_${v}_${p} "k_hessian_nms_c<0" 2; done
done
