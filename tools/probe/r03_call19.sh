cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 600 --timed-only 2>/dev/null || exit 1; done
timeout -k 10 200 python tools/probe/ab_env.py - 2>/dev/null || exit 1
