#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; echo "-- before"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_periods|nr_throttled|throttled_usec"
python bench.py --blocks 5 --timed-only --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['block_values'], d['collect_gap_ms']['max'], d['collect_gap_ms']['argmax'])"
echo "-- after"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_periods|nr_throttled|throttled_usec"
python bench.py --blocks 5 --timed-only --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['block_values'], d['collect_gap_ms']['max'], d['collect_gap_ms']['argmax'])"
echo "-- after 2"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_periods|nr_throttled|throttled_usec"
uptime
