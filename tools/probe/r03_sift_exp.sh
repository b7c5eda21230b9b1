cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in e1 e2; do
UVO_HIP_LIB=$GRAFT_REPO_ROOT/ergo_uvo_amd/lib/libuvo_hip_$e.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sift_$e -- python3 tools/prof_sift.py 4 > gpurun_out/prof_sift_$e.log 2>&1
done
echo ok
