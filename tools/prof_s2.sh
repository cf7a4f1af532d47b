#!/bin/bash
# usage: prof_s2.sh <lib> <curves> <tag>
cd /tmp && export TMPDIR=/tmp
export GECM_LIB=$1
rm -rf /tmp/prof_$3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$3 -- python3 $GRAFT_REPO_ROOT/bench.py --curves $2 --b1 100000 --b2 100000000 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > /dev/null 2>&1
f=$(ls /tmp/prof_$3/*/*kernel_stats.csv | head -1)
echo "== $3"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if r[0].startswith('void k_s2') or r[0].startswith('k_s2'): print(r[0][:28], r[1], "%.1f ms" % (float(r[2])/1e6))
PY
