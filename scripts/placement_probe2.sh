#!/bin/bash
# second probe: is the 15 % spread of the gather kernels a property of the PROCESS (where its tables land) or of the DEVICE at that
# time?  N short processes with the clocks / power / temperature read between them, then ONE long process (per-10-step means).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/placement
mkdir -p $OUT
cd $R
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)" | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ';'; echo; }
N=${1:-10}
for i in $(seq 1 $N); do
  echo "before $i: $(smi)" >> $OUT/probe2.log
  python scripts/placement_probe.py --steps 20 --tag p$i 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['tag'], {k[6:]: v for k, v in d['ms'].items() if v > 0.05}, d['series'])" >> $OUT/probe2.log
done
echo "before long: $(smi)" >> $OUT/probe2.log
python scripts/placement_probe.py --steps ${2:-400} --tag long 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['tag'], [ (x.get('item_seg'), x.get('user_seg'), x.get('sample')) for x in d['series']])" >> $OUT/probe2.log
echo "after long: $(smi)" >> $OUT/probe2.log
cat $OUT/probe2.log
