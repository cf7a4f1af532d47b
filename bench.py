#!/usr/bin/env python3
"""bench.py — stage-1 curves/sec at B1=1e6 on 416-bit-class N (BASELINE.json metric).

Workload of the headline line = BASELINE.json configs[1]: 4096 curves per GPU, 415-bit random odd N,
B1 = 1e6, reference limb format 52-bit, the library's own choice of lanes per curve.  A "step" is one
pass of the hot path (ecm_stage1: the PRAC ladder over all prime powers < B1, ecm.c:1806-1854) over that
batch, already resident in HBM, plus the device factor scan and the ONE collective of the path (max-reduce
of the found record).  K steps run back to back on the resident points (the output of a stage 1 is a valid
input point of the next).  value = curves processed by all ranks / max-over-ranks time — the figure the
reference prints as "Stage 1 took" (ecm.c:1315-1317), inverted.

Launch:  python bench.py --gpus N --steps K --warmup W
  * N > 1 without WORLD_SIZE in the environment: this process touches no GPU; it starts N ranks through
    torch.distributed.run (one process per GPU, RCCL) as a child process and relays rank 0's line.
  * under torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE set): --gpus must equal WORLD_SIZE.
  * every rank works on its own 4096 curves (weak scaling: N = 8 is BASELINE configs[3]).

Time budget (--budget-s, default 450 s): the step count is derived from the first warm-up step so that
warm-up + timed steps + the CPU baseline stay inside it ("steps_requested" records what was asked for);
the extras run only while budget is left.  Rank 0 prints ONE JSON line on stdout.

Extra objects on the line:
  roofline      VALU-issue roofline of the dominant kernel: achieved integer multiply-adds
                (v_mad_u64_u32, the instruction the multiply is built from) per second against the gfx950
                issue peak; the same in SURVEY.md §8d units (52-bit limb products).  Kernel time from HIP
                events on the kernel's own stream.
  cpu_baseline  the reference's own AVX-512 binary (oracle/_ref, built from /root/reference in the build
                container) timed on this box's host cores on a bounded sample; the scalar-C port
                (oracle/) if the binary is absent or cannot run here.
  saturated     131072 curves in one launch (one curve per lane, 2 wavefronts per SIMD)
  config2 / config4   BASELINE configs[2] (4096 x 831-bit, B1=1e6) and configs[4] (1023-bit, 32-bit
                reference limbs, B1=1e5), one warm-up and one step each
  bits624       the 624-bit class of north_star (4096 x 623-bit, B1=1e6), likewise
  ranks         the world size as the process group reports it, every rank's curve range, device and kernel time, and
                how many found-record all-reduces each rank really ran
"""
import argparse
import ctypes
import json
import os
import random
import re
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))

PEAK_MAD_PER_S = 256 * 4 * 16 * 2.4e9      # CUs x SIMDs x lanes/clk (v_mad_u64_u32: 4 clk per wave64) x Hz
PEAK_FMA64_PER_S = 256 * 4 * 16 * 2.4e9    # v_fma_f64 issues at the same rate (78.6 TFLOP/s datasheet)
KERNEL_NAMES = {1: "k_stage1<%d>", 2: "k_stage1_pair<%d>", 8: "k_stage1_quad<%d>"}


def kernel_name(lanes, dev_limbs):
    """the stage-1 kernel a launch with this layout runs, as rocprofv3 prints it"""
    if lanes == 32:        # templated on limbs per lane and rows of a multiply (one limb more than the buffers hold)
        nq = (dev_limbs + 1 + 15) // 16
        return "k_stage1_row<%d, %d, false>" % (nq, nq * ((dev_limbs + nq) // nq))
    return KERNEL_NAMES.get(lanes, "k_stage1_l%d<%%d>" % lanes) % dev_limbs


def work_per_curve(ptadds, ptdups, nl, n52):
    """(mul, sqr, 28-bit multiply-adds, 52-bit limb products) of one curve's stage 1: point-op counters of
    the reference (ecm.c:441, 455) x 4 mul + 2 sqr per add, 3 mul + 2 sqr per doubling; SURVEY.md §8d"""
    mul = 4 * ptadds + 3 * ptdups
    sqr = 2 * ptadds + 2 * ptdups
    mads = mul * (2 * nl * nl + nl) + sqr * (nl * (nl + 1) // 2 + nl * nl + nl)
    w52 = mul * (2 * n52 * n52 + n52) + sqr * (n52 * (n52 + 1) // 2 + n52 * n52 + n52)
    return mul, sqr, mads, w52


def host_cores():
    """cores this process may really use: cgroup quota if any, else affinity; capped at 64"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline(n, b1, budget_s=30.0):
    """reference AVX-512 binary on the host cores; bounded sample (8 curves on 1 thread, then 8 per core)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "avx-ecm-52")
    ncores = host_cores()
    out = None
    t_start = time.perf_counter()
    if os.path.exists(exe):
        try:
            def run(threads, timeout):
                with tempfile.TemporaryDirectory() as d:
                    p = subprocess.run([exe, str(n), str(8 * threads), str(b1), str(threads), str(b1), "1000"],
                                       cwd=d, capture_output=True, text=True, timeout=timeout)
                m = re.search(r"Stage 1 took ([0-9.]+) seconds", p.stdout)
                if p.returncode != 0 or not m:
                    raise RuntimeError("reference binary failed rc=%d" % p.returncode)
                return float(m.group(1))
            t1 = run(1, max(20.0, 3 * budget_s))
            out = {"value": 8 / t1, "unit": "curves/s", "cores": 1, "kind": "reference",
                   "sample": "oracle/_ref/avx-ecm-52 N 8 %d 1 %d 1000: 8 curves, 1 thread, %.2f s stage 1" % (b1, b1, t1)}
            left = budget_s - (time.perf_counter() - t_start)
            if ncores > 1 and t1 * 2.0 < left:
                tns = [run(ncores, max(20.0, 4 * t1))]
                while len(tns) < 3 and budget_s - (time.perf_counter() - t_start) > 1.5 * tns[-1]:
                    tns.append(run(ncores, max(20.0, 4 * t1)))       # more samples while the budget lasts
                tn = sorted(tns)[len(tns) // 2]
                out = {"value": 8 * ncores / tn, "unit": "curves/s", "cores": ncores, "kind": "reference",
                       "sample": "oracle/_ref/avx-ecm-52 N %d %d %d %d 1000: %d curves on %d threads, median of %d runs "
                                 "(%s s of stage 1); 1 thread, 8 curves: %.3f curves/s"
                                 % (8 * ncores, b1, ncores, b1, 8 * ncores, ncores, len(tns), ", ".join("%.2f" % t for t in tns), 8 / t1),
                       "samples_curves_per_s": [8 * ncores / t for t in tns]}
        except Exception as e:  # SIGILL on a host without AVX-512, missing libgmp, time-out, ...
            if out is None:
                sys.stderr.write("bench: reference binary unusable here (%s); timing the scalar port\n" % e)
    if out is None:
        so = os.path.join(ROOT, "oracle", "libecm_oracle.so")
        L = ctypes.CDLL(so)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.orc_time_stage1.restype = ctypes.c_double
        L.orc_time_stage1.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]
        c = L.orc_create(str(n).encode(), 52)
        t = L.orc_time_stage1(c, 1000, 2, b1)
        out = {"value": 2 / t, "unit": "curves/s", "cores": 1, "kind": "port",
               "sample": "oracle/ecm_oracle.c scalar port, 2 curves, %.2f s" % t}
    return out


def plan_steps(steps, warmup, t_probe, avail_s):
    """(further warm-up steps, timed steps) after the probe step, which is warm-up step 1: what was asked
    for if it fits into avail_s at t_probe seconds per step, else at most one more warm-up step and as many
    timed steps as fit (at least 1)"""
    steps = max(1, steps)
    more_warm = max(0, warmup - 1)
    afford = int(avail_s / max(t_probe, 1e-6))
    if more_warm + steps > afford:
        more_warm = min(more_warm, 1)
        steps = max(1, min(steps, afford - more_warm))
    return more_warm, steps


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv):
    """--gpus N > 1 outside a torch.distributed launch: start the N ranks as a child process.  Nothing in this
    process has touched the GPU (no torch, no libgecm import), so the child launcher is an ordinary spawn."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env)
    return p.returncode


class NoEngine:
    """--no-engine: rehearsal of the launcher and of the collective control flow on a box without a GPU
    (CPU tests).  Computes nothing; the line it produces is marked "rehearsal" and carries no roofline."""

    class _Cfg:
        dev_limbs, nwords, maxbits = 15, 8, 416

    class _St:
        ptadds, ptdups = 1980817, 217929

    cfg = _Cfg()

    def build_curves(self, sig):
        self.batch = len(sig)

    def set_lanes_per_curve(self, lanes):
        pass

    def stage1(self, b1, sync=True):
        time.sleep(0.01)

    def last_kernel_ms(self):
        return 10.0

    def scan_factors(self, stage=1):
        return 0, None

    def stage1_stats(self):
        return self._St()

    def lanes_per_curve(self):
        return 0

    def close(self):
        pass


def timed_pass(eng, b1, warm):
    """one optional warm-up and one stage-1 pass on the resident batch: (wall s, kernel ms)"""
    if warm:
        eng.stage1(b1, sync=True)
    t = time.perf_counter()
    eng.stage1(b1, sync=True)
    return time.perf_counter() - t, eng.last_kernel_ms()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--curves", type=int, default=4096, help="curves per GPU per step (BASELINE configs[1]: 4096)")
    ap.add_argument("--bits", type=int, default=415)
    ap.add_argument("--b1", type=int, default=1000000)
    ap.add_argument("--budget-s", type=float, default=450.0, help="wall-clock budget of the whole run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip saturated / config2 / config4 / special_form")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per curve in stage 1: 0 = library's choice, 1, 2, 8, 32")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank control flow)")
    ap.add_argument("--b2", type=int, default=100000000, help="after the timed stage-1 steps, one stage-2 pass to this B2 on "
                    "every rank's resident batch (BASELINE configs[3]: B2 = 1e8), reported separately as \"stage2\"; "
                    "0 = none.  Not part of the metric, outside the timed region")
    ap.add_argument("--no-engine", action="store_true", help="rehearsal without a GPU: launcher + collectives only")
    a = ap.parse_args()
    t_begin = time.perf_counter()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (a.gpus, world))

    import torch
    dist = None
    if a.no_engine:
        pyecm, devno = None, 0
    else:
        import pyecm
        ndev = max(1, pyecm.device_count())
        devno = local_rank % ndev
    on_gpu = not a.no_engine and (world == 1 or a.backend == "nccl")
    if "WORLD_SIZE" in os.environ:
        # launched as a rank (torch.distributed.run): the process group is made even for a world of one, so that
        # a one-GPU box exercises the same RCCL path (init, barrier, all-reduce on device tensors) as N ranks
        import torch.distributed as dist
        if a.backend == "nccl":
            torch.cuda.set_device(devno)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", devno))
        else:
            dist.init_process_group(backend=a.backend)
    dev = "cuda:%d" % devno if on_gpu else "cpu"

    if a.no_engine:
        sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd", "pyecm"))
        import shard
    else:
        from pyecm import shard
    n = random.Random(a.bits).getrandbits(a.bits) | (1 << (a.bits - 1)) | 1
    eng = NoEngine() if a.no_engine else pyecm.Engine(n, digitbits=52, device=devno)
    # host-side split of the curve batch (pyecm/shard.py): rank g owns the contiguous global curve
    # indices [g*C, (g+1)*C), sigma = 1000 + index
    total = a.curves * world
    lo, hi = shard.shard_bounds(total, rank, world)
    eng.build_curves(shard.shard_sigmas(1000, total, rank, world))
    eng.set_lanes_per_curve(a.lanes)

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def allmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    kernel_ms = []
    found_log = []
    collectives = [0]          # found-record all-reduces that really went through the process group

    prepared = []

    def step():
        if a.b2 > a.b1 and not a.no_engine and not prepared:
            # probe step only: the pair map of [B1, B2) is made on the host while the device runs stage 1
            # (gecm_stage2_prepare; kept in the context for the stage-2 pass after the timed region)
            eng.stage1(a.b1, sync=False)
            eng.stage2_prepare(a.b2)
            eng.sync()
            prepared.append(True)
        else:
            eng.stage1(a.b1, sync=True)
        kernel_ms.append(eng.last_kernel_ms())
        # factor scan of the whole batch on the device (check_factor, ecm.c:2542-2557), then the ONE
        # collective of the path: max-reduce of the found record (lowest global curve with a factor);
        # it replaces the reference's scan + break, ecm.c:1323-1370
        nf, first = eng.scan_factors(1)
        g = shard.allreduce_found(dist, None if first is None else lo + first, total, device=dev)
        found_log.append((nf, g))
        if dist is not None:
            collectives[0] += 1

    # ---- step count from the budget: one probe step (the first warm-up step, or an extra one) ----
    t0 = time.perf_counter()
    step()                                   # warm-up step 1 (run even with --warmup 0: it is the probe)
    t_probe = allmax(time.perf_counter() - t0)
    cpu_reserve = 0.0 if (a.no_cpu_baseline or world > 1) else 40.0
    avail = a.budget_s - (time.perf_counter() - t_begin) - cpu_reserve - 10.0
    # identical on every rank: t_probe is the max over ranks, and rank 0's clock decides
    avail = allmax(avail if rank == 0 else -1e30)
    more_warm, steps = plan_steps(a.steps, a.warmup, t_probe, avail)
    warmup = 1 + more_warm
    for _ in range(more_warm):
        step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = allmax(time.perf_counter() - t0)

    st = eng.stage1_stats()
    cfg = eng.cfg
    lanes = eng.lanes_per_curve()
    kname = kernel_name(lanes, cfg.dev_limbs) if a.no_engine else eng.last_kernel_name()

    # what every rank did, gathered so that the line checks itself: the world size as the process group reports it,
    # each rank's curve range, device and kernel time
    mine = [float(rank), float(lo), float(hi), float(devno), sum(kernel_ms) / max(1, len(kernel_ms)),
            min(kernel_ms) if kernel_ms else 0.0, max(kernel_ms) if kernel_ms else 0.0, float(collectives[0])]
    if dist is not None:
        tm = torch.tensor(mine, dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(tm) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, tm)
        per_rank = [[float(x) for x in p.tolist()] for p in parts]
        world_reported = dist.get_world_size()
    else:
        per_rank = [mine]
        world_reported = None

    stage2 = None
    if a.b2 > a.b1 and not a.no_engine:
        # not part of the metric: one pass of the stage-2 continuation on every rank's resident batch
        barrier()
        t2 = time.perf_counter()
        eng.stage2(a.b2)
        nf2, first2 = eng.scan_factors(2)
        shard.allreduce_found(dist, None if first2 is None else lo + first2, total, device=dev)
        barrier()
        t2 = allmax(time.perf_counter() - t2)
        s2 = eng.stage2_stats()
        stage2 = {"B2": a.b2, "seconds": t2, "curves_per_s": total / t2, "D": s2.D, "U": s2.U,
                  "ptadds": s2.ptadds, "inversions": s2.numinv, "pair_muls": s2.paired,
                  "curves_with_factor_rank0": nf2,
                  "stage1_plus_stage2_curves_per_s": total / (dt / steps + t2),
                  "note": "BASELINE configs[3] per GPU: %d curves, stage 1 to B1=%d then stage 2 to B2=%d on the resident "
                          "batch; not part of the metric" % (a.curves, a.b1, a.b2)}

    line = None
    if rank == 0:
        value = a.curves * world * steps / dt
        mul, sqr, mads, w52 = work_per_curve(st.ptadds, st.ptdups, cfg.dev_limbs, cfg.nwords)
        kms = sum(kernel_ms) / len(kernel_ms)
        achieved = mads * a.curves / (kms * 1e-3)
        # counters of this command from the committed rocprofv3 PMC passes (profiles/pmc_latest.json) — only if they
        # were taken on THIS build: the file carries the source hashes gecm_version() reports (the kernels' K/R/D parts)
        traffic = valu_per_mad = valu_active = None
        build = "" if a.no_engine else pyecm.lib.gecm_version().decode()
        # the part of the build that makes this kernel's code: R (the 32-lane kernels' sources) or K (the others')
        part = "R:" if kname.startswith("k_stage1_row") else "K:"
        pick = lambda b: " ".join(f for f in (b or "").split() if f[:2] == part)
        build_dev = pick(build)
        pmc_state = "no profiles/pmc_latest.json"
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pm.get("kernel") != kname or pm.get("curves") != a.curves or pm.get("B1") != a.b1:
                pmc_state = "committed profile is of another workload (%s, %s curves, B1 %s)" % (pm.get("kernel"), pm.get("curves"), pm.get("B1"))
            elif pick(pm.get("build")) != build_dev:
                pmc_state = "committed profile was taken on another build of this kernel (%s), this is %s" % (pick(pm.get("build")), build_dev)
            else:
                pmc_state = "profiles/pmc_latest.json, same kernel sources (%s)" % build_dev
                traffic = pm["hbm_bytes_per_launch_corrected"]
                waves = pm["counters"]["SQ_WAVES"]
                valu_per_mad = pm["valu_insts_per_wave"] * waves / (mads * a.curves / 64.0)
                valu_active = pm["valu_active_share_of_wave_cycles"]
        except Exception:
            pass
        roof = {
            "bound": "valu", "kernel": kname,
            "achieved": achieved / 1e12, "peak": PEAK_MAD_PER_S / 1e12, "unit": "Tmad/s (v_mad_u64_u32 lane-ops)",
            "frac": achieved / PEAK_MAD_PER_S, "traffic": traffic,
            "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes of this command (profiles/), FETCH_SIZE x2 "
                            "per the gfx950 correction; null unless the committed profile is of this workload AND this build. "
                            "It is ~30x the algorithmic bytes and costs nothing: the op tape (2.1 MB at B1=1e6) is read by "
                            "every one of the 2048 wavefronts through scalar loads, a few of those passes miss L2/MALL; at "
                            "~15 MB/s that is 2e-6 of the HBM peak (the kernel is bound by VALU issue)",
            "pmc": pmc_state,
            "valu_insts_per_algorithmic_mad": valu_per_mad,
            "valu_insts_note": "SQ_INSTS_VALU over all wavefronts / (algorithmic lane multiply-adds / 64): VALU instructions "
                               "issued per wave-level multiply-add the product-scanning count asks for; 1.0 would be a kernel "
                               "of nothing but useful v_mad_u64_u32",
            "valu_active_share_of_wave_cycles": valu_active,
            "kernel_ms_avg": kms, "mads_per_curve": mads,
            "mads_note": "algorithmic: product-scanning multiply 2n^2+n, square n(n+1)/2+n^2+n on n=%d limbs of 28 bits; "
                         "a layout that spends more instructions than that scores lower, not higher" % cfg.dev_limbs,
            "survey_units": {"limb_products_52bit_per_curve": w52,
                             "achieved_T52/s": w52 * a.curves / (kms * 1e-3) / 1e12,
                             "peak_T52/s_fp64_fma_pair": PEAK_FMA64_PER_S / 2 / 1e12,
                             "frac": w52 * a.curves / (kms * 1e-3) / (PEAK_FMA64_PER_S / 2)},
            "hbm_algorithmic_bytes_per_launch": a.curves * cfg.dev_limbs * 4 * 5,
        }
        line = {
            "metric": "stage-1 curves/sec at B1=%d, %d-bit N" % (a.b1, cfg.maxbits), "value": value,
            "unit": "curves/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "steps_requested": a.steps, "warmup_requested": a.warmup, "probe_step_s": t_probe,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 (28-bit limbs, 64-bit accumulate)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d curves per GPU per step, %d-bit random odd N (seed %d), B1=%d, "
                                   "sigma=1000.., stage 1 + device factor scan + found-record all-reduce; reference limb "
                                   "format 52-bit NWORDS=%d" % (a.curves, a.bits, a.bits, a.b1, cfg.nwords),
                       "curves_per_gpu": a.curves, "bits": a.bits, "B1": a.b1, "lanes_per_curve": lanes,
                       "curves_with_factor_last_step": found_log[-1][0],
                       "parallelism": ("curve batch split across %d GPU(s) on the host, no data-path collective, "
                                       "1 all-reduce (%s) of the found record per step"
                                       % (world, "RCCL" if a.backend == "nccl" else a.backend)) if dist is not None else
                                      "one process, one GPU: no collective (single process; launch through "
                                      "torch.distributed.run to make the process group and run the all-reduce)"},
            "ranks": {"process_group": None if dist is None else a.backend,
                      "world_size_reported_by_process_group": world_reported,
                      "found_record_allreduces_per_rank_timed_and_warmup": [int(r[7]) for r in per_rank],
                      "kernel_ms_avg_min_over_ranks": min(r[4] for r in per_rank),
                      "kernel_ms_avg_max_over_ranks": max(r[4] for r in per_rank),
                      "per_rank": [{"rank": int(r[0]), "global_curves": [int(r[1]), int(r[2])], "device": int(r[3]),
                                    "kernel_ms_avg": r[4], "kernel_ms_min": r[5], "kernel_ms_max": r[6]} for r in per_rank],
                      "curves_total": total},
            "roofline": roof,
        }
        if a.no_engine:
            line["rehearsal"] = "no engine: launcher and collectives only, value is meaningless"
            line.pop("roofline")
        if stage2:
            line["stage2"] = stage2
        sys.stderr.write("bench: headline %.1f curves/s (%d steps, %.1f ms/step)\n" % (value, steps, dt / steps * 1e3))

    def left():
        return a.budget_s - (time.perf_counter() - t_begin)

    if rank == 0 and world == 1 and not a.no_engine:
        # ---- rank 0 at N = 1 only: the other ranks would wait at the barrier ----
        if not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(n, a.b1, budget_s=min(30.0, max(8.0, left() - 15.0)))
            except Exception as ex:
                line["cpu_baseline"] = {"error": str(ex)}
        extras = [] if a.no_extras else ["config2", "bits624", "config4", "saturated", "special_form"]
        per_curve_s = dt / steps / a.curves          # of the headline layout; the estimates below are upper bounds
        for name in extras:
            try:
                if name == "config2":            # BASELINE configs[2]: 4096 x 832-bit class
                    if left() < 2 * 4.5 * t_probe + 20:
                        raise TimeoutError("budget")
                    n2 = random.Random(831).getrandbits(831) | (1 << 830) | 1
                    e2 = pyecm.Engine(n2, digitbits=52, device=devno)
                    e2.build_curves(list(range(1000, 1000 + 4096)))
                    w, k = timed_pass(e2, a.b1, True)
                    s2 = e2.stage1_stats()
                    m2 = work_per_curve(s2.ptadds, s2.ptdups, e2.cfg.dev_limbs, e2.cfg.nwords)[2]
                    line[name] = {"workload": "4096 curves, 831-bit random odd N (seed 831), B1=%d" % a.b1,
                                  "value": 4096 / w, "unit": "curves/s", "kernel_ms": k, "lanes_per_curve": e2.lanes_per_curve(),
                                  "dev_limbs": e2.cfg.dev_limbs, "valu_frac": m2 * 4096 / (k * 1e-3) / PEAK_MAD_PER_S}
                    e2.close()
                elif name == "bits624":          # north_star's third size: 624-bit class (623-bit N), same batch and B1
                    if left() < 2 * 2.5 * t_probe + 20:
                        raise TimeoutError("budget")
                    n6 = random.Random(623).getrandbits(623) | (1 << 622) | 1
                    e6 = pyecm.Engine(n6, digitbits=52, device=devno)
                    e6.build_curves(list(range(1000, 1000 + 4096)))
                    w, k = timed_pass(e6, a.b1, True)
                    s6 = e6.stage1_stats()
                    m6 = work_per_curve(s6.ptadds, s6.ptdups, e6.cfg.dev_limbs, e6.cfg.nwords)[2]
                    line[name] = {"workload": "4096 curves, 623-bit random odd N (seed 623), B1=%d" % a.b1,
                                  "value": 4096 / w, "unit": "curves/s", "kernel_ms": k, "lanes_per_curve": e6.lanes_per_curve(),
                                  "dev_limbs": e6.cfg.dev_limbs, "valu_frac": m6 * 4096 / (k * 1e-3) / PEAK_MAD_PER_S}
                    e6.close()
                elif name == "config4":          # BASELINE configs[4]: DIGITBITS=32 boundary, 1024-bit class, B1=1e5
                    if left() < 2 * 1.0 * t_probe + 15:
                        raise TimeoutError("budget")
                    n4 = random.Random(1023).getrandbits(1023) | (1 << 1022) | 1
                    e4 = pyecm.Engine(n4, digitbits=32, device=devno)
                    e4.build_curves(list(range(1000, 1000 + 4096)))
                    w, k = timed_pass(e4, 100000, True)
                    s4 = e4.stage1_stats()
                    m4 = work_per_curve(s4.ptadds, s4.ptdups, e4.cfg.dev_limbs, e4.cfg.nwords)[2]
                    line[name] = {"workload": "4096 curves, 1023-bit random odd N (seed 1023), 32-bit reference limbs, B1=100000",
                                  "value": 4096 / w, "unit": "curves/s", "kernel_ms": k, "lanes_per_curve": e4.lanes_per_curve(),
                                  "dev_limbs": e4.cfg.dev_limbs, "valu_frac": m4 * 4096 / (k * 1e-3) / PEAK_MAD_PER_S}
                    e4.close()
                elif name == "saturated":        # one curve per lane, 2 wavefronts on every SIMD
                    if left() < 2 * 27.0 + 15:
                        raise TimeoutError("budget")
                    eng.build_curves(list(range(1000, 1000 + 131072)))
                    eng.set_lanes_per_curve(0)
                    w, k = timed_pass(eng, a.b1, True)
                    line[name] = {"workload": "131072 curves in one launch, %d-bit N, B1=%d" % (a.bits, a.b1),
                                  "value": 131072 / w, "unit": "curves/s", "kernel_ms": k, "lanes_per_curve": eng.lanes_per_curve(),
                                  "valu_frac": work_per_curve(st.ptadds, st.ptdups, cfg.dev_limbs, cfg.nwords)[2]
                                  * 131072 / (k * 1e-3) / PEAK_MAD_PER_S}
                elif name == "special_form":     # 2^401 - 1 through generic REDC and through the F-form multiply
                    if left() < 25:
                        raise TimeoutError("budget")
                    e3 = pyecm.Engine((1 << 401) - 1, digitbits=52, device=devno)
                    ms = {}
                    for on in (False, True):
                        e3.set_special_form(on)
                        e3.set_lanes_per_curve(1)
                        e3.build_curves(list(range(1000, 1000 + 131072)))
                        e3.stage1(100000, sync=True)
                        ms[on] = e3.last_kernel_ms()
                    line[name] = {"N": "2^401-1", "curves": 131072, "B1": 100000, "generic_redc_kernel_ms": ms[False],
                                  "special_form_kernel_ms": ms[True], "speedup": ms[False] / ms[True]}
                    e3.close()
            except TimeoutError:
                line[name] = {"skipped": "time budget (%.0f s left)" % left()}
            except Exception as ex:                 # an extra must never cost the headline line
                line[name] = {"error": str(ex)}
    if rank == 0:
        line["wall_s"] = time.perf_counter() - t_begin
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
