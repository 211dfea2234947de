#!/bin/bash
# placement_probe3.py with the clocks / power polled beside it (rocm-smi every ~0.3 s, wall-clock stamps on both sides)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/probe3
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -2 | head -1)"; sleep 0.25; done ) > gpurun_out/probe3/smi.log 2>&1 &
POLL=$!
for p in 1 2; do
  EL_PROBE_STAMP=1 timeout 400 python scripts/placement_probe3.py --tag p$p 2>/dev/null | grep "^{" | tee -a gpurun_out/probe3/trials.log
done
kill $POLL
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -3
wc -l gpurun_out/probe3/smi.log
