#!/bin/bash
# kernel times of a synchronous run under alternative builds of the library (ergo_uvo_amd/lib_ab/libuvo_hip_<name>.so):
#   gpurun -- tools/probe/lib_variants.sh <kernel pattern> <name> [<name> ...]        ("cur" = the current build)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
pat=$1; shift
for v in "$@"; do
  if [ $v = cur ]; then unset UVO_HIP_LIB; else export UVO_HIP_LIB=$PWD/ergo_uvo_amd/lib_ab/libuvo_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lv -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_lv.log 2>&1 || { tail -3 gpurun_out/prof_lv.log; exit 1; }
  echo "$v: $(python tools/probe/kstats.py prof_lv 30 | grep -E "$pat" | awk '{print $NF}' | tr '\n' ' ') us"
  rm -rf gpurun_out/prof_lv
done
