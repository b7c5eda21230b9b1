#!/bin/bash
# per-octave detection kernels (UVO_HESSIAN_SPLIT=1) and the merged launch, previous build (lib_ab/libuvo_hip_old.so) against the current one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in old new; do
  if [ $v = old ]; then export UVO_HIP_LIB=$GRAFT_REPO_ROOT/ergo_uvo_amd/lib_ab/libuvo_hip_old.so; else unset UVO_HIP_LIB; fi
  UVO_HESSIAN_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_${v}_split -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_ab_${v}_split.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_${v}_merged -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_ab_${v}_merged.log 2>&1
  echo "== $v split"; python tools/probe/kstats.py prof_ab_${v}_split 30 | grep -i "hessian"
  echo "== $v merged"; python tools/probe/kstats.py prof_ab_${v}_merged 30 | grep -i "hessian"
done
