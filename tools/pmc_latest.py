#!/usr/bin/env python3
"""profiles/pmc_latest.json from one tools/profile_bench.sh output directory: the headline kernel's counters per launch,
the derived figures bench.py puts on its line (HBM bytes per launch, VALU instructions per wavefront, VALU-active and
issue-stall shares), and the BUILD the profile was taken on (the K/R/D source hashes of gecm_version(), which
profile_bench.sh writes to <dir>/build.txt) — bench.py reports the counters only when its own build matches.
usage: pmc_latest.py <dir> <kernel name as rocprofv3 prints it> <curves> <B1>"""
import json, os, sys

d, kernel, curves, b1 = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
s = json.load(open(os.path.join(d, "pmc_summary.json")))
e = s["kernels"][kernel]
build = open(os.path.join(d, "build.txt")).read().strip()
out = {"command": s["command"], "source": os.path.relpath(os.path.join(d, "pmc_summary.json"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")),
       "build": " ".join(f for f in build.split() if f[:2] in ("K:", "R:", "D:")), "build_full": build,
       "kernel": kernel, "curves": curves, "B1": b1, "counters": e["per_launch"],
       "kernel_ms_avg_in_trace": e["avg_ms"], "launches_in_trace": e["launches_in_trace"]}
for k in ("effective_clock_GHz", "valu_insts_per_wave", "valu_active_share_of_wave_cycles", "issue_stall_share_of_wave_cycles",
          "hbm_bytes_per_launch_corrected", "hbm_GBps"):
    out[k] = e.get(k)
out["note"] = ("FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies a 128-B read request as 64 B), WRITE_SIZE as read. "
               "The bytes counted are mostly the op tape read by every wavefront through scalar loads and served on-die in all but a "
               "few passes.")
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
