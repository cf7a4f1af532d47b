#!/bin/bash
# usage: prof_s2.sh <lib> <curves> <tag>      per-kernel times of stage 2 (k_s2_*) for one full 1e8 range
set -e
[ $# -eq 3 ] || { echo "usage: $0 <libgecm.so> <curves> <tag>" >&2; exit 1; }
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$3
rm -rf "$out"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export GECM_LIB=$1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --curves "$2" --b1 100000 --b2 100000000 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"
f=$(ls "$out"/*/*kernel_stats.csv | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv under $out (see $out/bench.err)" >&2; exit 1; }
echo "== $3"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if r[0].startswith('void k_s2') or r[0].startswith('k_s2'): print(r[0][:28], r[1], "%.1f ms" % (float(r[2])/1e6))
PY
