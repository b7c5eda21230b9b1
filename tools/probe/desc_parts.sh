#!/bin/bash
# the descriptor launch's two halves timed alone (UVO_DESC_PART: 1 large windows, 2 small) and the phases of
# its small-window workgroups (UVO_DESC_STAMPS):  gpurun -- tools/probe/desc_parts.sh
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
for part in ${PARTS:-2 1 0}; do
  UVO_DESC_PART=$part rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dp$part -- python3 tools/prof_stereo.py 24 > gpurun_out/prof_dp$part.log 2>&1 || exit 1
  echo "== UVO_DESC_PART=$part"; python tools/probe/kstats.py prof_dp$part 30 | grep -E "descriptor64\("
done
for part in ${SPARTS:-2 0}; do
  echo "== stamps, UVO_DESC_PART=$part"
  UVO_HIP_LIB=$PWD/ergo_uvo_amd/lib_ab/libuvo_hip_stamps.so UVO_DESC_PART=$part UVO_DESC_STAMPS=gpurun_out/desc_stamps_$part.csv python tools/prof_stereo.py 8 > /dev/null 2>&1; python tools/probe/desc_stamps.py gpurun_out/desc_stamps_$part.csv
done
