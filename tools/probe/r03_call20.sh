cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo -n "new  "; timeout -k 10 200 python bench.py --steps 600 --timed-only 2>/dev/null || exit 1
  echo -n "prev "; UVO_HIP_LIB=$GRAFT_REPO_ROOT/ergo_uvo_amd/lib/libuvo_hip_prev.so timeout -k 10 200 python bench.py --steps 600 --timed-only 2>/dev/null || exit 1
done
