"""time (and check against the main build) A/B builds of the 32-lane kernel: python3 tools/ab_row_time.py name [name ...]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "main":
        env["GECM_LIB"] = os.path.join(root, "avx-ecm_amd", "libgecm_%s.so" % name)
    for curves in (70, 4096):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "run_small.py"), str(curves), "100000", "32"], env=env,
                             capture_output=True, text=True).stdout.strip()
        print("%-8s %s" % (name, out), flush=True)
