cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 5 8 11 20 40 5 11 40; do echo -n "w$w "; timeout -k 10 200 python bench.py --steps 20 --warmup $w --timed-only 2>/dev/null || exit 1; done
