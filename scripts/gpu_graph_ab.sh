#!/bin/bash
# is the slow-box step host-bound?  eager launches vs hipGraph replay of the same step, plus host info
cd "${GRAFT_REPO_ROOT:-/root/repo}"
grep -m1 "model name" /proc/cpuinfo; nproc; cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null; uptime
for mode in "" "--graph" "" "--graph"; do
python bench.py --no-cpu-baseline --model off $mode 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', l['launch'], l['ms_per_step'], 'eager+events', l['ms_per_step_eager_with_events'])"
done
