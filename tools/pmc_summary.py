#!/usr/bin/env python3
"""Summarise one tools/profile_bench.sh run: per-kernel time from the kernel trace, per-kernel PMC counters from the
three counter passes, FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950.
usage: pmc_summary.py <dir> "<bench args>"  -> <dir>/kernel_stats.csv, <dir>/pmc_summary.json (and prints it)"""
import csv, glob, json, os, re, sys
from collections import defaultdict

d = sys.argv[1]
args = sys.argv[2] if len(sys.argv) > 2 else ""


def rows(pattern):
    for f in glob.glob(os.path.join(d, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


# kernel trace: duration per dispatch
dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(d, "kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms\n")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.3f,%.3f,%.3f,%.3f\n" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6))

# counters: sum over dispatches of each kernel, and the number of dispatches
cnt = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for r in rows(sub + "/**/*counter_collection.csv"):
        k = short(r["Kernel_Name"])
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {"command": "tools/profile_bench.sh: rocprofv3 --kernel-trace --stats, then --pmc passes (SQ+GRBM; FETCH_SIZE; "
                  "WRITE_SIZE), each -- python3 bench.py " + args, "kernels": {}}
for k, c in cnt.items():
    if k not in dur:
        continue
    n = {name: len(ids) for name, ids in disp[k].items()}
    per = {name: v / max(1, n[name]) for name, v in c.items()}           # per launch
    e = {"launches_in_trace": len(dur[k]), "avg_ms": sum(dur[k]) / len(dur[k]) / 1e6, "per_launch": per}
    if "SQ_INSTS_VALU" in per and per.get("SQ_WAVES"):
        e["valu_insts_per_wave"] = per["SQ_INSTS_VALU"] / per["SQ_WAVES"]
    if per.get("SQ_WAVE_CYCLES"):
        e["valu_active_share_of_wave_cycles"] = per.get("SQ_ACTIVE_INST_VALU", 0) / per["SQ_WAVE_CYCLES"]
        e["issue_stall_share_of_wave_cycles"] = per.get("SQ_WAIT_INST_ANY", 0) / per["SQ_WAVE_CYCLES"]
    if "GRBM_GUI_ACTIVE" in per and e["avg_ms"] > 1.0:
        # summed over the 8 XCDs; the trace's average launch time is the time base (the counter passes run the
        # same launches), so this is only meaningful for long kernels
        e["effective_clock_GHz"] = per["GRBM_GUI_ACTIVE"] / 8.0 / (e["avg_ms"] * 1e-3) / 1e9
    if "FETCH_SIZE" in per or "WRITE_SIZE" in per:
        # rocprofv3 reports both in KiB; gfx950 tallies a 128-byte read request as 64 bytes: double FETCH_SIZE
        e["hbm_bytes_per_launch_corrected"] = (2.0 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024.0
        e["hbm_GBps"] = e["hbm_bytes_per_launch_corrected"] / (e["avg_ms"] * 1e-3) / 1e9
    out["kernels"][k] = e
json.dump(out, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches_in_trace"])[:6]:
    print(k, json.dumps({x: y for x, y in e.items() if x != "per_launch"}))
