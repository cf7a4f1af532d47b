#!/usr/bin/env python3
"""bench.py — stage-1 curves/sec at B1=1e6 on 416-bit-class N (BASELINE.json metric).

One process per GPU (torch.distributed over RCCL when --gpus > 1).  A "step" is one pass of the
hot path (ecm_stage1: PRAC ladder over all prime powers < B1) over one batch of curves already
resident in HBM; K steps run back to back on the resident points (the output of a stage 1 is a
valid input point of the next).  value = curves processed by all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      VALU-issue roofline of the dominant kernel (k_stage1): achieved integer
                multiply-adds (v_mad_u64_u32, the instruction the multiply is built from) per second
                against the gfx950 issue peak; plus the same in SURVEY.md §8d units (52-bit limb
                products).  Kernel time from HIP events on the kernel's own stream.
  cpu_baseline  the reference's own AVX-512 binary (oracle/_ref, built from /root/reference in the
                build container) timed on this box's host cores on a bounded sample; falls back to
                the scalar-C port (oracle/) if the binary is absent or cannot run here.
"""
import argparse
import ctypes
import json
import os
import random
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))

# point-op counts per curve at B1=1e6 (reference counters ecm.c:441, 455)
PEAK_MAD_PER_S = 256 * 4 * 16 * 2.4e9      # CUs x SIMDs x lanes/clk (v_mad_u64_u32: 4 clk per wave64) x Hz
PEAK_FMA64_PER_S = 256 * 4 * 16 * 2.4e9    # v_fma_f64 issues at the same rate (78.6 TFLOP/s datasheet)


def work_per_curve(ptadds, ptdups, nl, n52):
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


def cpu_baseline(n, b1, budget_s=25.0):
    """reference AVX-512 binary on the host cores; bounded sample."""
    exe = os.path.join(ROOT, "oracle", "_ref", "avx-ecm-52")
    ncores = host_cores()
    out = None
    if os.path.exists(exe):
        try:
            # 1 thread x 8 curves first (~5 s at B1=1e6 on a 2 GHz core)
            def run(threads):
                with tempfile.TemporaryDirectory() as d:
                    p = subprocess.run([exe, str(n), str(8 * threads), str(b1), str(threads), str(b1), "1000"],
                                       cwd=d, capture_output=True, text=True, timeout=600)
                m = re.search(r"Stage 1 took ([0-9.]+) seconds", p.stdout)
                if p.returncode != 0 or not m:
                    raise RuntimeError("reference binary failed rc=%d" % p.returncode)
                return float(m.group(1))
            t1 = run(1)
            res = {"value": 8 / t1, "unit": "curves/s", "cores": 1, "kind": "reference",
                   "sample": "oracle/_ref/avx-ecm-52 N 8 %d 1 %d 1000: 8 curves, 1 thread, %.2f s stage 1" % (b1, b1, t1)}
            if t1 * 1.5 < budget_s and ncores > 1:
                thr = ncores
                tn = run(thr)
                res = {"value": 8 * thr / tn, "unit": "curves/s", "cores": thr, "kind": "reference",
                       "sample": "oracle/_ref/avx-ecm-52 N %d %d %d %d 1000: %d curves on %d threads, %.2f s stage 1; "
                                 "1 thread: %.3f curves/s" % (8 * thr, b1, thr, b1, 8 * thr, thr, tn, 8 / t1)}
            out = res
        except Exception as e:  # SIGILL on a host without AVX-512, missing libgmp, ...
            out = None
            note = "reference binary unusable here: %s" % e
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--curves", type=int, default=131072, help="curves per GPU per step")
    ap.add_argument("--bits", type=int, default=415)
    ap.add_argument("--b1", type=int, default=1000000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per curve in stage 1: 0 = library's choice, 1, 2")
    ap.add_argument("--no-special-form", action="store_true", help="skip the extra 2^401-1 measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank control flow on fewer GPUs than ranks)")
    ap.add_argument("--b2", type=int, default=0, help="also time one stage-2 pass to this B2 (reported separately)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    import pyecm
    ndev = max(1, pyecm.device_count())
    devno = local_rank % ndev
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            torch.cuda.set_device(devno)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", devno))
        else:
            dist.init_process_group(backend=a.backend)

    from pyecm import shard
    n = random.Random(a.bits).getrandbits(a.bits) | (1 << (a.bits - 1)) | 1
    eng = pyecm.Engine(n, digitbits=52, device=devno)
    # host-side split of the curve batch (pyecm/shard.py): rank g owns the contiguous global curve
    # indices [g*C, (g+1)*C), sigma = 1000 + index
    total = a.curves * world
    lo, hi = shard.shard_bounds(total, rank, world)
    eng.build_curves(shard.shard_sigmas(1000, total, rank, world))
    dev = "cuda:%d" % devno if (world == 1 or a.backend == "nccl") else "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = []

    found_log = []

    eng.set_lanes_per_curve(a.lanes)

    def step():
        eng.stage1(a.b1, sync=True)
        kernel_ms.append(eng.last_kernel_ms())
        # factor scan of the whole batch on the device (check_factor, ecm.c:2542-2557), then the ONE
        # collective of the path: max-reduce of the found record (lowest global curve with a factor)
        nf, first = eng.scan_factors(1)
        g = shard.allreduce_found(dist, None if first is None else lo + first, total, device=dev)
        found_log.append((nf, g))

    for _ in range(a.warmup):
        step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = eng.stage1_stats()
    cfg = eng.cfg
    lanes = eng.lanes_per_curve()
    kname = ("k_stage1<%d>" if lanes == 1 else "k_stage1_pair<%d>") % cfg.dev_limbs
    stage2 = None
    if a.b2 > a.b1:
        # not part of the metric: one pass of the stage-2 continuation on the resident batch
        t2 = time.perf_counter()
        eng.stage2(a.b2)
        nf2, _ = eng.scan_factors(2)
        t2 = time.perf_counter() - t2
        s2 = eng.stage2_stats()
        stage2 = {"B2": a.b2, "seconds": t2, "curves_per_s": a.curves / t2, "D": s2.D, "U": s2.U,
                  "ptadds": s2.ptadds, "inversions": s2.numinv, "pair_muls": s2.paired,
                  "curves_with_factor": nf2}
    small = None
    if world == 1 and a.curves != 4096 and not a.no_small_batch:
        # BASELINE.json configs[1] names a 4096-curve batch: with one curve per lane that is 64
        # wavefronts for 1024 SIMDs, so the library spreads each curve over eight lanes (512 wavefronts:
        # X and Z on two quads, the limbs of a residue over the lanes of its quad).  Measured separately
        # (one pass) and reported next to `value`.
        eng.build_curves(list(range(1000, 1000 + 4096)))
        eng.set_lanes_per_curve(0)
        t1 = time.perf_counter()
        eng.stage1(a.b1, sync=True)
        t1 = time.perf_counter() - t1
        small = {"curves": 4096, "value": 4096 / t1, "unit": "curves/s", "ms_per_step": t1 * 1e3,
                 "kernel_ms": eng.last_kernel_ms(), "lanes_per_curve": eng.lanes_per_curve()}
    special = None
    if world == 1 and not a.no_special_form and not a.no_small_batch:
        # not part of the metric: a Mersenne-form modulus (the reference's isMersenne inputs) through the
        # generic REDC kernel and through the 2^k - 1 multiply, same batch, B1 = 1e5 (parity of the two paths
        # is tests/test_gpu_fform.py's job; this is the timing)
        try:
            e2 = pyecm.Engine((1 << 401) - 1, digitbits=52, device=devno)
            ms = {}
            for on in (False, True):
                e2.set_special_form(on)
                e2.set_lanes_per_curve(1)
                e2.build_curves(list(range(1000, 1000 + a.curves)))
                e2.stage1(100000, sync=True)
                ms[on] = e2.last_kernel_ms()
            special = {"N": "2^401-1", "curves": a.curves, "B1": 100000, "generic_redc_kernel_ms": ms[False],
                       "special_form_kernel_ms": ms[True], "speedup": ms[False] / ms[True]}
            e2.close()
        except Exception as ex:                 # the extra must never cost the headline line
            special = {"error": str(ex)}
    if rank == 0:
        total_curves = a.curves * world * a.steps
        value = total_curves / dt
        mul, sqr, mads, w52 = work_per_curve(st.ptadds, st.ptdups, cfg.dev_limbs, cfg.nwords)
        kms = sum(kernel_ms) / len(kernel_ms)
        mads_per_launch = mads * a.curves
        achieved = mads_per_launch / (kms * 1e-3)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pm.get("kernel") == kname and pm.get("curves") == a.curves and pm.get("B1") == a.b1:
                traffic = pm["hbm_bytes_per_launch_corrected"]
        except Exception:
            pass
        roof = {
            "bound": "valu", "kernel": kname,
            "achieved": achieved / 1e12, "peak": PEAK_MAD_PER_S / 1e12, "unit": "Tmad/s (v_mad_u64_u32 lane-ops)",
            "frac": achieved / PEAK_MAD_PER_S, "traffic": traffic,
            "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes of this command (profiles/), FETCH_SIZE x2 "
                            "per the gfx950 correction; null if no matching profile is committed",
            "kernel_ms_avg": kms, "mads_per_curve": mads,
            "survey_units": {"limb_products_52bit_per_curve": w52,
                             "achieved_T52/s": w52 * a.curves / (kms * 1e-3) / 1e12,
                             "peak_T52/s_fp64_fma_pair": PEAK_FMA64_PER_S / 2 / 1e12,
                             "frac": w52 * a.curves / (kms * 1e-3) / (PEAK_FMA64_PER_S / 2)},
            "hbm_algorithmic_bytes_per_launch": a.curves * cfg.dev_limbs * 4 * 5,
        }
        line = {
            "metric": "stage-1 curves/sec at B1=%d, %d-bit N" % (a.b1, cfg.maxbits), "value": value,
            "unit": "curves/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 (28-bit limbs, 64-bit accumulate)", "data": "synthetic",
            "config": {"workload": "%d curves per GPU per step (= %d sub-batches of the 4096 curves of BASELINE "
                                   "configs[1], in one launch: one curve per lane needs 131072 curves to put 2 "
                                   "wavefronts on each of the 1024 SIMDs), %d-bit random odd N (seed %d), B1=%d, "
                                   "sigma=1000.., stage 1 + device factor scan; reference limb format 52-bit NWORDS=%d"
                                   % (a.curves, a.curves // 4096, a.bits, a.bits, a.b1, cfg.nwords),
                       "curves_per_gpu": a.curves, "bits": a.bits, "B1": a.b1, "lanes_per_curve": lanes,
                       "curves_with_factor_last_step": found_log[-1][0],
                       "parallelism": "curve batch split across %d GPU(s) on the host, no data-path collective, "
                                      "1 all-reduce (RCCL) of the found record per step" % world},
            "roofline": roof,
        }
        if small:
            line["batch_4096"] = small
        if stage2:
            line["stage2"] = stage2
        if special:
            line["special_form"] = special
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only: the other ranks would wait at the barrier
            line["cpu_baseline"] = cpu_baseline(n, a.b1)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
