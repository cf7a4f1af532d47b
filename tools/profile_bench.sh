#!/bin/bash
# Profile `python3 bench.py <args>` on the GPU box: kernel trace with stats, then three separate PMC passes
# (SQ + GRBM; FETCH_SIZE; WRITE_SIZE — the TCC counters do not fit one pass), and the summary of all four.
#   tools/profile_bench.sh <out-name> [bench.py args ...]     -> gpurun_out/<out-name>/
# Counters are collected in their own runs, never together with a trace domain.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="$@"
# the build the profile is taken on (source hashes of every object in libgecm.so)
python3 -c "import sys; sys.path.insert(0, '$root/avx-ecm_amd'); import pyecm; print(pyecm.lib.gecm_version().decode())" > "$out/build.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/bench_trace.json" 2> "$out/trace.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_sq" -- python3 "$root/bench.py" $args > /dev/null 2> "$out/pmc_sq.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" $args > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" $args > /dev/null 2> "$out/pmc_write.err"
python3 "$root/tools/pmc_summary.py" "$out" "$args"
