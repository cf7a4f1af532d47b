"""ISA check of the pair-walk kernels that request table rows with hand-placed loads (gecm_stage2.hpp, tb_load_async).
The compiler does not know that a row's registers are pending between the load and the hand-placed s_waitcnt; the only
thing it could put there is a copy (live-range split, loop-carried move) or a spill, since every use in the source
goes through an empty asm placed after the wait.  The check: in k_s2_pairs<NL>, no instruction outside the inline-asm
blocks touches the destination register of an asm load that no later s_waitcnt has covered yet
(walked in layout order, which is execution order inside the unrolled loop body).  Also prints the vmcnt values found.
usage: python tools/check_async_rows.py [NL ...]   (default: every limb count up to GECM_S2_ASYNC_MAXNL)"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "avx-ecm_amd", "csrc", "gecm_kernels.hip")
HDR = open(os.path.join(ROOT, "avx-ecm_amd", "csrc", "gecm_stage2.hpp")).read()
MAXNL = int(re.search(r"#define GECM_S2_ASYNC_MAXNL (\d+)", HDR).group(1))
ALL = [int(x) for x in re.search(r"NLS\s+:=\s+([\d ]+)", open(os.path.join(ROOT, "avx-ecm_amd", "Makefile")).read()).group(1).split()]


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return {"v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    return {tok} if re.fullmatch(r"v\d+", tok) else set()


def check(nl):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DGECM_NL=%d" % nl, "-DGECM_PART=2", "-S",
                        "--cuda-device-only", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r"^_Z10k_s2_pairsILi%dEE\w*:[^\n]*\n(.*?)s_endpgm" % nl, text, re.S | re.M)
    body = m.group(1).split("\n")
    # linear walk in layout order: `out` = row loads not yet covered by a wait, oldest first
    rows, out, bad, waits, in_asm = set(), [], [], {}, False
    for n, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        mm = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
        if mm:
            k = int(mm.group(1))
            waits[k] = waits.get(k, 0) + 1
            out = out[len(out) - k:] if k and k < len(out) else ([] if not k else out)
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip().split(" ")[0] for t in rest.split(",")]
        if in_asm:
            mm = re.match(r"global_load_dword (v\d+), v\d+, s\[", l)
            if mm:
                rows.add(mm.group(1))
                out = [r for r in out if r != mm.group(1)] + [mm.group(1)]
            continue
        touched = set()
        for t in toks:
            touched |= regs(t)
        if touched & set(out):
            bad.append((n, l))
    return nl, len(rows), waits, bad


if __name__ == "__main__":
    nls = [int(a) for a in sys.argv[1:]] or [n for n in sorted(ALL) if n <= MAXNL]
    ok = True
    with ThreadPoolExecutor(max_workers=6) as ex:
        for nl, nrows, waits, bad in ex.map(check, nls):
            print("NL=%2d: %3d row registers, waits %s, %d suspect instructions" % (nl, nrows, dict(sorted(waits.items())), len(bad)), flush=True)
            for n, l in bad[:8]:
                print("     line %d: %s" % (n, l))
            ok &= not bad
    sys.exit(0 if ok else 1)
