// gecm_dev.hip — device management for libgecm (gfx950 / MI355X only): buffers, stream, events,
// dispatch to the per-limb-count kernel launchers of gecm_kernels.hip.
#include "gecm_dev.h"
#include "gecm_launch.h"
#include "gecm_tape.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;
extern "C" const char *gecm_dev_error(void) { return g_err.c_str(); }

#define HIPCHK(x)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            char b_[512];                                                                     \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            g_err = b_;                                                                       \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

// ---------------------------------------------------------------- host side
static const int k_supported_nl[] = {
#define X(n) n,
    GECM_NL_LIST(X)
#undef X
    0};
extern "C" const int *gecm_dev_supported_nl(void) { return k_supported_nl; }

#ifndef GECM_S2_WAVES_PER_SIMD
#define GECM_S2_WAVES_PER_SIMD 16         // wavefronts per SIMD a pair-walk launch is cut up for (gecm_dev_s2_init)
#endif
#define GECM_S2_MAX_SLICES 64
#ifndef GECM_ROW_DEFAULT_SMALL
#define GECM_ROW_DEFAULT_SMALL 0   // 32-lane kernel variant for batches up to 2 wavefronts per SIMD (row_a_lds)
#endif
struct gecm_dev {
    int device = 0;
    int nl = 0;
    std::vector<uint32_t> n, kp, one;
    uint32_t rho = 0;
    size_t ncurves = 0, stride = 0;
    uint32_t *dX = nullptr, *dZ = nullptr, *dS = nullptr, *dT0 = nullptr, *dT1 = nullptr;
    uint32_t *dTape = nullptr;
    size_t tape_len = 0, tape_cap = 0;
    std::vector<hipEvent_t> cut_events;   // one per stage-1 launch of the last gecm_dev_stage1 (progress of a long tape)
    size_t cut_launches = 0;
    std::vector<size_t> tape_cuts;   // byte offsets at which a stage-1 launch may start (gecm_dev_set_tape), first = 0, last = tape_len
    // stage 2
    std::vector<uint32_t> r3;
    uint32_t inv_iters = 0;
    uint32_t *dPbX = nullptr, *dBlk = nullptr, *dPd = nullptr, *dAcc = nullptr, *dFail = nullptr, *dKeep = nullptr;
    uint32_t *dPa = nullptr, *dSteps = nullptr, *dFlags = nullptr;
    size_t flags_cap = 0;
    size_t s2_npb = 0, s2_G = 0, s2_ring = 0, s2_stride = 0, steps_cap = 0, keep_cap = 0;
    uint64_t steps_id = 0;    // the kept tape whose copy dSteps holds (0: none), and its length
    uint32_t steps_n = 0;
    uint32_t s2_slices = 1;   // stage-2 accumulators per curve (pair-walk slices), see gecm_dev_s2_init
    uint32_t s2_K = 1;        // sub-sequences per curve for the table build and the giant steps (gecm_dev_s2_subseq)
    uint32_t *dKBlk = nullptr, *dKPa = nullptr, *dPdK = nullptr, *dTgt = nullptr;
    size_t tgt_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    bool timed = false;
    uint32_t *dModQ = nullptr;   // N and K' limbs padded to 40 each, for the eight-lane kernel
    uint32_t *dRowC = nullptr;   // constants of the 32-lane kernel (gecm_dev_set_rowconst), GECM_ROW_KINDS x GECM_ROW_WORDS
    int row_nq = 0, row_rows = 0; // limbs per lane and rows of a multiply there; 0 = not available
    int fform = 0;        // +1 / -1 / 2: modulus is 2^k - 1 / 2^k + 1 / 2^k - c and stage 1 uses the special multiply (gecm_dev_set_fform)
    int cus = 0;          // compute units of the device (4 SIMDs each)
    int last_lanes = 0;   // lanes per curve the last stage-1 launch used
    std::string last_kernel;   // and the kernel's name as rocprofv3 prints it
};

// ---- source manifest (Makefile): "K:<hash of the kernel objects' sources, or MIXED> R:<rowk> D:<this file>"
#ifndef GECM_MANIFEST
#define GECM_MANIFEST "unset"
#endif
extern "C" const char *gecm_manifest_rowk(void);
#define X(n) extern "C" const char *gecm_manifest_k_##n##_p1(void); extern "C" const char *gecm_manifest_k_##n##_p2(void);
GECM_NL_LIST(X)
#undef X
extern "C" const char *gecm_dev_manifest(void)
{
    static std::string m;
    if (m.empty()) {
        std::string k;
        bool mixed = false;
#define X(n)                                                                       \
        for (const char *h : {gecm_manifest_k_##n##_p1(), gecm_manifest_k_##n##_p2()}) { \
            if (k.empty()) k = h;                                                  \
            else if (k != h) mixed = true;                                         \
        }
        GECM_NL_LIST(X)
#undef X
        std::string t = std::string("K:") + (mixed ? "MIXED" : k) + " R:" + gecm_manifest_rowk() + " D:" + GECM_MANIFEST;
#ifdef GECM_DEV_NL15
        t += " DEV-BUILD(416-bit class only)";
#endif
        m = t;
    }
    return m.c_str();
}

extern "C" int gecm_dev_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static gecm_modconst modconst(const gecm_dev *d)
{
    gecm_modconst mc;
    mc.n = d->n.data();
    mc.kp = d->kp.data();
    mc.one = d->one.data();
    mc.r3 = d->r3.empty() ? d->one.data() : d->r3.data();
    mc.rho = d->rho;
    mc.inv_iters = d->inv_iters;
    return mc;
}

extern "C" int gecm_dev_open(gecm_dev **out, int device, int nl, const uint32_t *n, const uint32_t *kp,
                             const uint32_t *one, uint32_t rho)
{
    bool ok = false;
    for (const int *p = k_supported_nl; *p; p++) ok |= (*p == nl);
    if (!ok) {
        g_err = "gecm_dev_open: unsupported limb count " + std::to_string(nl);
        return -2;
    }
    int cnt = 0;
    HIPCHK(hipGetDeviceCount(&cnt));
    if (device < 0 || device >= cnt) {
        g_err = "gecm_dev_open: no such device";
        return -2;
    }
    HIPCHK(hipSetDevice(device));
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    gecm_dev *d = new gecm_dev;
    d->device = device;
    d->cus = cus;
    d->nl = nl;
    d->n.assign(n, n + nl);
    d->kp.assign(kp, kp + nl);
    d->one.assign(one, one + nl);
    d->rho = rho;
    if (nl <= 40) {
        uint32_t h[80] = {0};
        for (int i = 0; i < nl; i++) {
            h[i] = n[i];
            h[40 + i] = kp[i];
        }
        HIPCHK(hipMalloc(&d->dModQ, sizeof h));
        HIPCHK(hipMemcpy(d->dModQ, h, sizeof h, hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&d->ev0));
    HIPCHK(hipEventCreate(&d->ev1));
    *out = d;
    return 0;
}

extern "C" int gecm_dev_set_rowconst(gecm_dev *d, int nq, int rows, const uint32_t *words)
{
    HIPCHK(hipSetDevice(d->device));
    if (nq < 1 || nq > GECM_ROW_MAXNQ || 16 * nq > GECM_ROW_WORDS || rows <= 16 * (nq - 1) || rows > 16 * nq) {
        g_err = "gecm_dev_set_rowconst: limbs per lane out of range";
        return -2;
    }
    if (!d->dRowC) HIPCHK(hipMalloc(&d->dRowC, GECM_ROW_KINDS * GECM_ROW_WORDS * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(d->dRowC, words, GECM_ROW_KINDS * GECM_ROW_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice));
    d->row_nq = nq;
    d->row_rows = rows;
    return 0;
}

static void free_state(gecm_dev *d)
{
    (void)hipFree(d->dX); (void)hipFree(d->dZ); (void)hipFree(d->dS); (void)hipFree(d->dT0); (void)hipFree(d->dT1);
    d->dX = d->dZ = d->dS = d->dT0 = d->dT1 = nullptr;
}

static void free_s2(gecm_dev *d)
{
    (void)hipFree(d->dPbX); (void)hipFree(d->dBlk); (void)hipFree(d->dPd); (void)hipFree(d->dAcc);
    (void)hipFree(d->dFail); (void)hipFree(d->dPa);
    (void)hipFree(d->dKBlk); (void)hipFree(d->dKPa); (void)hipFree(d->dPdK);
    d->dKBlk = d->dKPa = d->dPdK = nullptr;
    d->dPbX = d->dBlk = d->dPd = d->dAcc = d->dFail = d->dPa = nullptr;
    d->s2_npb = d->s2_G = d->s2_ring = d->s2_stride = 0;
}

extern "C" void gecm_dev_close(gecm_dev *d)
{
    if (d) for (hipEvent_t e : d->cut_events) (void)hipEventDestroy(e);
    if (!d) return;
    (void)hipSetDevice(d->device);
    free_state(d);
    free_s2(d);
    (void)hipFree(d->dModQ);
    (void)hipFree(d->dRowC);
    (void)hipFree(d->dKeep);
    (void)hipFree(d->dSteps);
    (void)hipFree(d->dFlags);
    (void)hipFree(d->dTape);
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

extern "C" int gecm_dev_memory(gecm_dev *d, uint64_t *free_bytes, uint64_t *total_bytes)
{
    HIPCHK(hipSetDevice(d->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

extern "C" int gecm_dev_device_name(gecm_dev *d, char *buf, size_t len)
{
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, d->device));
    /* (some boxes report an empty marketing name) */
    snprintf(buf, len, "%s (%s, %d CUs)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName, p.multiProcessorCount);
    return 0;
}

extern "C" size_t gecm_dev_stride(gecm_dev *d) { return d->stride; }

extern "C" int gecm_dev_resize(gecm_dev *d, size_t ncurves)
{
    HIPCHK(hipSetDevice(d->device));
    size_t stride = (ncurves + 63) / 64 * 64;
    if (stride == 0) stride = 64;
    if (stride != d->stride) {
        free_state(d);
        size_t bytes = stride * d->nl * sizeof(uint32_t);
        HIPCHK(hipMalloc(&d->dX, bytes));
        HIPCHK(hipMalloc(&d->dZ, bytes));
        HIPCHK(hipMalloc(&d->dS, bytes));
        HIPCHK(hipMalloc(&d->dT0, bytes));
        HIPCHK(hipMalloc(&d->dT1, bytes));
        d->stride = stride;
    }
    d->ncurves = ncurves;
    return 0;
}

// host [limb][ncurves] -> device [limb][stride], padding lanes zero
static int upload_soa(gecm_dev *d, uint32_t *dst, const uint32_t *src)
{
    HIPCHK(hipMemsetAsync(dst, 0, d->stride * d->nl * sizeof(uint32_t), d->stream));
    if (d->ncurves)
        HIPCHK(hipMemcpy2DAsync(dst, d->stride * 4, src, d->ncurves * 4, d->ncurves * 4, d->nl,
                                hipMemcpyHostToDevice, d->stream));
    return 0;
}

static int download_soa(gecm_dev *d, uint32_t *dst, const uint32_t *src)
{
    if (d->ncurves)
        HIPCHK(hipMemcpy2DAsync(dst, d->ncurves * 4, src, d->stride * 4, d->ncurves * 4, d->nl,
                                hipMemcpyDeviceToHost, d->stream));
    return 0;
}

extern "C" int gecm_dev_upload(gecm_dev *d, const uint32_t *X, const uint32_t *Z, const uint32_t *S)
{
    HIPCHK(hipSetDevice(d->device));
    if (upload_soa(d, d->dX, X)) return -1;
    if (upload_soa(d, d->dZ, Z)) return -1;
    if (upload_soa(d, d->dS, S)) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int gecm_dev_upload_xz(gecm_dev *d, const uint32_t *X, const uint32_t *Z)
{
    HIPCHK(hipSetDevice(d->device));
    if (upload_soa(d, d->dX, X)) return -1;
    if (upload_soa(d, d->dZ, Z)) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int gecm_dev_set_tape(gecm_dev *d, const uint8_t *tape, size_t len)
{
    HIPCHK(hipSetDevice(d->device));
    size_t words = (len + 3) / 4 + 1;
    if (words > d->tape_cap) {
        (void)hipFree(d->dTape);
        d->dTape = nullptr;
        HIPCHK(hipMalloc(&d->dTape, words * 4));
        d->tape_cap = words;
    }
    HIPCHK(hipMemsetAsync(d->dTape, 0, words * 4, d->stream));
    if (len) HIPCHK(hipMemcpyAsync(d->dTape, tape, len, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    d->tape_len = len;
    /* A long tape (B1 in the 1e8s: 200 MB per prime range) runs as several launches.  Between two prac() calls only
     * the point A is live (PRAC_BEGIN sets B = C = A, ecm.c:603-608), and that is what every stage-1 kernel stores at
     * exit and loads at entry, so a launch may start at any PRAC_BEGIN byte — the 2-power doublings are such bytes
     * too.  The kernels read the tape as words: cuts are taken at offsets that are multiples of 4.  GECM_TAPE_CHUNK
     * (bytes) overrides the 16 MB default (tests cut short tapes with it). */
    size_t chunk = (size_t)16 << 20;
    if (const char *e = getenv("GECM_TAPE_CHUNK")) { long v = atol(e); if (v >= 4) chunk = (size_t)v; }
    d->tape_cuts.clear();
    d->tape_cuts.push_back(0);
    for (size_t want = chunk; want < len; ) {
        size_t off = (want + 3) & ~(size_t)3;
        while (off < len && tape[off] != GECM_OP_PRAC_BEGIN) off += 4;
        if (off >= len) break;
        d->tape_cuts.push_back(off);
        want = off + chunk;
    }
    d->tape_cuts.push_back(len);
    return 0;
}

/* Lanes per curve for this batch (tools/lanes_bench.py).  One curve per lane runs in
 * rounds of 2 wavefronts on each SIMD (4 per CU): full = 128 curves x SIMDs.  It is the faster layout
 * (by ~3%) only when its last round is nearly full; whenever that round would leave SIMDs idle or
 * half-occupied, two lanes per curve — twice the wavefronts, each half as long — fills them:
 * 1.83x below a quarter of `full`, 1.05x at half, 1.26x at three quarters.
 * From 26 limbs up (> 640-bit N) the split layout is the faster one at every batch size (all three
 * points stay in registers instead of one being parked in LDS: +1.5% at 26 limbs, +4% at 30, +10% at 37;
 * equal at 23 — tools/lanes_sizes.py, interleaved runs). */
extern "C" int gecm_dev_auto_lanes(gecm_dev *d)
{
    /* a batch that cannot even put one two-lane wavefront on every SIMD: eight lanes per curve (X and Z on two
     * quads, the limbs of a residue spread over the quad, csrc/gecm_quad.hpp) — 1.5x the two-lane layout at 15
     * limbs, 2.2-2.4x at 30-37 limbs, up to 32 curves per CU; from 19 limbs up still 1.2-1.4x at 64 curves
     * per CU (tools/quad_check.py).  Generic moduli only. */
    /* smaller still — at most 4 wavefronts of 2 curves on every SIMD: 32 lanes per curve (X and Z on two DPP rows,
     * the limbs over the 16 lanes of a row, csrc/gecm_row.hpp), the layout that puts BASELINE configs[1]'s 4096
     * curves on every SIMD of the chip twice: 1.8x the eight-lane layout at 4096 curves (415 and 831 bits), 1.3x
     * at 1023 bits, 1.05x at 8192 curves, slower from there on (tools/row_check.py).  Below 10 limbs the
     * 16 lanes of a row are mostly padding and the eight-lane layout wins. */
    /* Above 8192 curves (measured at 15, 30 and 37 limbs, profiles/r02_layouts_mid_batches.txt): the 32-lane kernel's
     * time grows with the batch, the eight- and two-lane kernels' in steps (one more wavefront per SIMD), so the
     * 32-lane layout still wins where the others have just taken a step: one limb per lane (up to 15 limbs) up to 56
     * curves per CU (18.7k against 16.2k curves/s at 12,288 curves, 415 bits; the two-lane layout takes over at
     * 14.4k); more limbs per lane up to 30 curves per CU and from 32 to 50 (831 bits: 31.9k against 29.7k curves/s at
     * 12,288), the eight-lane layout at exactly one or two wavefronts per SIMD in between (8192 curves: 32.7k against
     * 30.6k). */
    if (!d->fform && d->row_nq && d->nl >= 10 && d->stride) {
        const size_t s = d->stride, cu = (size_t)d->cus;
        if (d->row_nq == 1 ? s <= cu * 56 : (s <= cu * 30 || (s > cu * 32 && s <= cu * 50))) return 32;
    }
    /* Below 10 limbs the 32-lane layout still has the shortest chain per curve (9 rows of 6 instructions): while the
     * batch leaves it at one wavefront per SIMD or less (8 curves per CU) it is latency that counts — 8 curves of a
     * 204-bit N at B1 = 3e6: 4.11 s against 7.15 s (eight lanes) and 8.42 s (two), tools/multirange_time.py. */
    if (!d->fform && d->row_nq && d->nl < 10 && d->stride && d->stride <= (size_t)d->cus * 8) return 32;
    if (!d->fform && d->dModQ && d->stride && d->stride <= (size_t)d->cus * (d->nl >= 19 ? 64 : 32)) return 8;
    if (d->nl >= 26) return 2;
    const size_t full = (size_t)d->cus * 4 * 128;
    const size_t r = d->stride % full;
    return (r == 0 || r > full / 4 * 3) ? 1 : 2;
}

/* 32-lane kernel, how the scanned operand of a multiply reaches its row (csrc/gecm_row.hpp): 0 = DPP broadcasts, 1 =
 * ds_swizzle through the LDS crossbar (best from 3 wavefronts per SIMD up, tools/row_check.py), 2 = point forms kept in
 * LDS and read one multiply ahead (k_stage1_rowp: the fewest VALU instructions; what a batch at up to 2 wavefronts per
 * SIMD — BASELINE configs[1] — is short of).  GECM_ROW_ALDS=0/1/2 overrides for experiments. */
static int row_a_lds(const gecm_dev *d)
{
    const char *e = getenv("GECM_ROW_ALDS");
    if (e && e[0] >= '0' && e[0] <= '2' && !e[1]) return e[0] - '0';
    /* two or three limbs per lane: the DPP rows (operand limbs two per v_mov_b64_dpp, gecm_row.hpp) beat the crossbar
     * variant by 8-12 % at every batch size from 4096 to 16,384 curves (profiles/r03/mid_batch_rows.txt); one limb per
     * lane: DPP up to two wavefronts per SIMD (4096 curves: 262 against 284 ms), the crossbar from there (6144: 363
     * against 373; equal from 8192 on) */
    if (d->row_nq >= 2) return 0;
    return d->stride > (size_t)d->cus * 16 ? 1 : GECM_ROW_DEFAULT_SMALL;
}

extern "C" int gecm_dev_last_lanes(gecm_dev *d) { return d->last_lanes; }
extern "C" const char *gecm_dev_last_kernel(gecm_dev *d) { return d->last_kernel.c_str(); }

extern "C" int gecm_dev_fform_generic_limbs(int nl)
{
    switch (nl) {
#define X(n) case n: return gecm_fform_generic_limbs_##n();
        GECM_NL_LIST(X)
#undef X
    }
    return -1;
}

extern "C" void gecm_dev_set_fform(gecm_dev *d, int form) { d->fform = form == 2 ? 2 : form > 0 ? 1 : form < 0 ? -1 : 0; }

extern "C" int gecm_dev_stage1(gecm_dev *d, int lanes_per_curve)
{
    HIPCHK(hipSetDevice(d->device));
    if (!d->stride || !d->dTape) {
        g_err = "gecm_dev_stage1: no curves or no tape";
        return -2;
    }
    if (lanes_per_curve == 0) lanes_per_curve = gecm_dev_auto_lanes(d);
    if (lanes_per_curve != 1 && lanes_per_curve != 2 && lanes_per_curve != 8 && lanes_per_curve != 32) {
        g_err = "gecm_dev_stage1: lanes per curve must be 0 (auto), 1, 2, 8 or 32";
        return -2;
    }
    if (lanes_per_curve == 32 && (d->fform || !d->row_nq)) {
        g_err = "gecm_dev_stage1: no 32-lane kernel for this modulus";
        return -2;
    }
    if (lanes_per_curve == 8 && (d->fform || !d->dModQ)) {
        g_err = "gecm_dev_stage1: no eight-lane kernel for this modulus";
        return -2;
    }
    d->last_lanes = lanes_per_curve;
    {
        char nm[96];
        const int mode = row_a_lds(d);
        if (lanes_per_curve == 32 && mode == 2) snprintf(nm, sizeof nm, "k_stage1_rowp<%d, %d>", d->row_nq, d->row_rows);
        else if (lanes_per_curve == 32) snprintf(nm, sizeof nm, "k_stage1_row<%d, %d, %s>", d->row_nq, d->row_rows, mode ? "true" : "false");
        else if (lanes_per_curve == 8) snprintf(nm, sizeof nm, "k_stage1_quad<%d>", d->nl);
        else if (d->fform) snprintf(nm, sizeof nm, "%s<%d, Mod%c<%d> >", lanes_per_curve == 2 ? "k_stage1_pair_f" : "k_stage1_f", d->nl,
                                    d->fform == 2 ? 'C' : d->fform > 0 ? 'F' : 'P', d->nl);
        else snprintf(nm, sizeof nm, "%s<%d>", lanes_per_curve == 2 ? "k_stage1_pair" : "k_stage1", d->nl);
        d->last_kernel = nm;
    }
    gecm_modconst mc = modconst(d);
    HIPCHK(hipEventRecord(d->ev0, d->stream));
    d->cut_launches = d->tape_cuts.size() - 1;
    while (d->cut_events.size() < d->cut_launches) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->cut_events.push_back(e);
    }
    for (size_t cut = 0; cut + 1 < d->tape_cuts.size(); cut++) {
    const uint32_t *tp = d->dTape + d->tape_cuts[cut] / 4;
    const uint32_t tl = (uint32_t)(d->tape_cuts[cut + 1] - d->tape_cuts[cut]);
    switch (d->nl) {
#define X(n)                                                                                     \
    case n:                                                                                      \
        if (lanes_per_curve == 32) {                                                             \
            if (gecm_launch_stage1_row(d->stream, d->row_nq, d->row_rows, tp, tl,    \
                                       d->dX, d->dZ, d->dS, d->stride, (uint32_t)d->nl,          \
                                       d->dRowC, d->rho,                                 \
                                       row_a_lds(d))) {                                      \
                g_err = "gecm_dev_stage1: no 32-lane kernel for this limb count";                \
                return -2;                                                                       \
            }                                                                                    \
            gecm_launch_canon_##n(d->stream, &mc, d->dX, d->dZ, d->stride);                      \
        } else if (lanes_per_curve == 8) {                                                              \
            if (gecm_launch_stage1_quad_##n(d->stream, &mc, tp, tl,     \
                                            d->dX, d->dZ, d->dS, d->stride, d->dModQ)) {         \
                g_err = "gecm_dev_stage1: no eight-lane kernel for this limb count";             \
                return -2;                                                                       \
            }                                                                                    \
        } else if (d->fform)                                                                     \
            gecm_launch_stage1_f_##n(d->stream, &mc, tp, tl, d->dX,     \
                                     d->dZ, d->dS, d->stride, lanes_per_curve, d->fform);       \
        else if (lanes_per_curve == 2)                                                           \
            gecm_launch_stage1_pair_##n(d->stream, &mc, tp, tl, d->dX,  \
                                        d->dZ, d->dS, d->stride);                                \
        else                                                                                     \
            gecm_launch_stage1_##n(d->stream, &mc, tp, tl, d->dX,       \
                                   d->dZ, d->dS, d->stride);                                     \
        break;
        GECM_NL_LIST(X)
#undef X
    default:
        g_err = "unsupported nl";
        return -2;
    }
    HIPCHK(hipEventRecord(d->cut_events[cut], d->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev1, d->stream));
    d->timed = true;
    return 0;
}

/* launches of the last gecm_dev_stage1 that have finished / that it made (a long tape runs as several) */
extern "C" int gecm_dev_stage1_progress(gecm_dev *d, uint32_t *done, uint32_t *total)
{
    uint32_t n = 0;
    for (size_t i = 0; i < d->cut_launches; i++) {
        if (hipEventQuery(d->cut_events[i]) != hipSuccess) break;    // in order on one stream
        n++;
    }
    (void)hipGetLastError();
    if (done) *done = n;
    if (total) *total = (uint32_t)d->cut_launches;
    return 0;
}

extern "C" int gecm_dev_sync(gecm_dev *d)
{
    HIPCHK(hipSetDevice(d->device));
    HIPCHK(hipStreamSynchronize(d->stream));
    if (d->timed) {
        HIPCHK(hipEventElapsedTime(&d->last_ms, d->ev0, d->ev1));
        d->timed = false;
    }
    return 0;
}

extern "C" float gecm_dev_last_kernel_ms(gecm_dev *d) { return d->last_ms; }

extern "C" int gecm_dev_download_mont(gecm_dev *d, uint32_t *X, uint32_t *Z)
{
    HIPCHK(hipSetDevice(d->device));
    if (download_soa(d, X, d->dX)) return -1;
    if (download_soa(d, Z, d->dZ)) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int gecm_dev_download_plain(gecm_dev *d, uint32_t *x, uint32_t *z)
{
    HIPCHK(hipSetDevice(d->device));
    gecm_modconst mc = modconst(d);
    switch (d->nl) {
#define X(n)                                                                                   \
    case n:                                                                                    \
        gecm_launch_from_mont_##n(d->stream, &mc, d->dX, d->dZ, d->dT0, d->dT1, d->stride);    \
        break;
        GECM_NL_LIST(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    if (download_soa(d, x, d->dT0)) return -1;
    if (download_soa(d, z, d->dT1)) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int gecm_dev_l0(gecm_dev *d, int op, const uint32_t *a, const uint32_t *b, uint32_t *c,
                           uint32_t *dd, size_t count, const uint32_t *fix)
{
    HIPCHK(hipSetDevice(d->device));
    size_t keep = d->ncurves;
    if (gecm_dev_resize(d, count)) return -1;
    // reuse the state buffers: X<-a, Z<-b, outputs T0, T1
    if (upload_soa(d, d->dX, a)) return -1;
    if (upload_soa(d, d->dZ, b ? b : a)) return -1;
    gecm_modconst mc = modconst(d);
    switch (d->nl) {
#define X(n)                                                                                      \
    case n:                                                                                       \
        gecm_launch_l0_##n(d->stream, &mc, op, d->dX, d->dZ, d->dT0, d->dT1, d->stride,           \
                           fix ? fix : d->one.data());                                            \
        break;
        GECM_NL_LIST(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    if (download_soa(d, c, d->dT0)) return -1;
    if (op == GECM_L0_ADDSUB && dd)
        if (download_soa(d, dd, d->dT1)) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    (void)keep;
    return 0;
}

// ---------------------------------------------------------------- stage 2
extern "C" int gecm_dev_set_s2const(gecm_dev *d, const uint32_t *r3, uint32_t inv_iters)
{
    d->r3.assign(r3, r3 + d->nl);
    d->inv_iters = inv_iters;
    return 0;
}

/* Sub-sequences per curve for the stage-2 table build and giant steps: those are one dependent chain per curve, so a
 * batch of a few thousand curves (a few dozen wavefronts) is split into K interleaved chains per curve until there
 * are 2 wavefronts per SIMD (csrc/gecm_stage2.hpp, s2_init_k / giant_chunk_k).  1 for batches that fill the chip.
 * GECM_S2_SUBSEQ=1..32 (a power of two) overrides, for measurements. */
static uint32_t s2_slices_for(const gecm_dev *d, size_t stride);
static uint32_t s2_subseq_for(const gecm_dev *d, size_t stride);

/* Device bytes a batch of `curves` curves takes: the five stage-1 arrays, and with npb != 0 the stage-2 allocations of
 * gecm_dev_s2_init for a table of npb entries, chunks of G giant steps and a ring of ring_size (with the sub-sequence
 * and slice counts a batch of that size gets).  The same sums as the hipMallocs below. */
extern "C" uint64_t gecm_dev_batch_bytes(gecm_dev *d, size_t curves, uint32_t npb, uint32_t G, uint32_t ring_size)
{
    const size_t stride = (curves + 63) / 64 * 64;
    const uint64_t coord = (uint64_t)d->nl * stride * sizeof(uint32_t);
    uint64_t words = 5 + 1;                                   // X, Z, S, two scratch arrays; factor-scan gcds
    if (npb) {
        const uint64_t K = s2_subseq_for(d, stride), slices = s2_slices_for(d, stride);
        words += (uint64_t)npb + 3 * GECM_S2_BLK + 2 + slices + (K > 1 ? K + 1 : 1) + (2 * ((uint64_t)G + 2) + G + ring_size);
        if (K > 1) {
            const uint64_t Gs = G / K + 1;
            words += K * 3 * GECM_S2_BLK + K * (2 * (Gs + 2) + Gs) + 2;
        }
    }
    return words * coord;
}

extern "C" uint32_t gecm_dev_s2_subseq(gecm_dev *d) { return s2_subseq_for(d, d->stride); }

static uint32_t s2_subseq_for(const gecm_dev *d, size_t stride)
{
    // two wavefronts per SIMD is the target (4096 curves, B2 = 1e8: K = 1 1.14 s, 4 0.75 s, 16 and 32 0.64 s)
    const size_t waves = stride / 64, want = (size_t)d->cus * 4 * 2;
    uint32_t k = 1;
    while (k < 32 && waves * (k * 2) <= want) k *= 2;
    if (const char *e = getenv("GECM_S2_SUBSEQ")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= 32 && (v & (v - 1)) == 0) k = (uint32_t)v;
    }
    return k;
}

static uint32_t s2_slices_for(const gecm_dev *d, size_t stride)
{
    const size_t waves = stride / 64, want = (size_t)d->cus * 4 * GECM_S2_WAVES_PER_SIMD;
    size_t p = waves ? (want + waves - 1) / waves : 1;
    uint32_t s = (uint32_t)(p < 1 ? 1 : p > GECM_S2_MAX_SLICES ? GECM_S2_MAX_SLICES : p);
    if (const char *e = getenv("GECM_S2_SLICES")) {      // measurement knob (tools/s2_small.py)
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= GECM_S2_MAX_SLICES) s = (uint32_t)v;
    }
    return s;
}

extern "C" uint32_t gecm_dev_s2_fail_planes(gecm_dev *d) { return d->s2_K > 1 ? d->s2_K + 1 : 1; }

extern "C" int gecm_dev_s2_init(gecm_dev *d, const uint32_t *keep, size_t keep_words, uint32_t umax, uint32_t D,
                                uint32_t npb, uint32_t G, uint32_t ring_size, const uint32_t *tgt, const uint32_t *tgt_off,
                                uint32_t K)
{
    HIPCHK(hipSetDevice(d->device));
    if (!d->stride || d->r3.empty()) {
        g_err = "gecm_dev_s2_init: no curves or stage-2 constants not set";
        return -2;
    }
    const size_t coord = (size_t)d->nl * d->stride * sizeof(uint32_t);
    if (K < 1 || K > 32 || (K & (K - 1)) || (K > 1 && (!tgt || !tgt_off))) {
        g_err = "gecm_dev_s2_init: bad number of sub-sequences";
        return -2;
    }
    if (d->s2_npb != npb || d->s2_G != G || d->s2_ring != ring_size || d->s2_stride != d->stride || d->s2_K != K) {
        free_s2(d);
        d->s2_K = K;
        HIPCHK(hipMalloc(&d->dPbX, coord * npb));
        HIPCHK(hipMalloc(&d->dBlk, coord * 3 * GECM_S2_BLK));    // bx, bz, bp: S2_BLK entries each
        HIPCHK(hipMalloc(&d->dPd, coord * 2));
        // The pair walk is a product, so each run of pairs is walked in `slices` parts (one more grid dimension, one
        // accumulator each; slice 0 is the accumulator everyone else sees) until the launch has GECM_S2_WAVES_PER_SIMD
        // wavefronts for every SIMD.  Two per SIMD is what it takes to issue a multiply-add every 4.7 cycles, but the
        // walk also waits for table rows, and its 160 registers let three wavefronts share a SIMD: the full batch
        // (2048 wavefronts) went from 8.26 s to 7.68 s per 1e8 range with 4 slices, 7.57 s with 8; 32,768 curves from
        // 2.04 s (8 slices) to 1.96 s (32); 4096 curves from 0.380 s (32) to 0.371 s (64)
        // (profiles/r02_stage2_slices.txt).
        d->s2_slices = s2_slices_for(d, d->stride);
        HIPCHK(hipMalloc(&d->dAcc, coord * d->s2_slices));
        HIPCHK(hipMalloc(&d->dFail, coord * (K > 1 ? K + 1 : 1)));       // plane 0 + one per sub-sequence
        if (K > 1) {
            const size_t Gs = G / K + 1;
            HIPCHK(hipMalloc(&d->dKBlk, coord * K * 3 * GECM_S2_BLK));    // kbx, kbz, kbp per (curve block, r)
            HIPCHK(hipMalloc(&d->dKPa, coord * K * (2 * (Gs + 2) + Gs)));  // kgx, kgz (Gs+2 entries), kgp (Gs)
            HIPCHK(hipMalloc(&d->dPdK, coord * 2));
        }
        // giant steps: gx, gz (G+2 entries each), gp (G), ring (ring_size)
        HIPCHK(hipMalloc(&d->dPa, coord * (2 * ((size_t)G + 2) + (size_t)G + (size_t)ring_size)));
        d->s2_npb = npb; d->s2_G = G; d->s2_ring = ring_size; d->s2_stride = d->stride;
    }
    if (keep_words > d->keep_cap) {
        (void)hipFree(d->dKeep);
        d->dKeep = nullptr;
        HIPCHK(hipMalloc(&d->dKeep, keep_words * 4));
        d->keep_cap = keep_words;
    }
    HIPCHK(hipMemcpyAsync(d->dKeep, keep, keep_words * 4, hipMemcpyHostToDevice, d->stream));
    if (K > 1) {
        const size_t tw = (size_t)tgt_off[K] + (K + 1);                 // target lists, then the K + 1 offsets
        if (tw > d->tgt_cap) {
            (void)hipFree(d->dTgt);
            d->dTgt = nullptr;
            HIPCHK(hipMalloc(&d->dTgt, tw * 4));
            d->tgt_cap = tw;
        }
        HIPCHK(hipMemcpyAsync(d->dTgt, tgt, (size_t)tgt_off[K] * 4, hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipMemcpyAsync(d->dTgt + tgt_off[K], tgt_off, (K + 1) * 4, hipMemcpyHostToDevice, d->stream));
    }
    HIPCHK(hipMemsetAsync(d->dFail, 0, coord * (K > 1 ? K + 1 : 1), d->stream));
    HIPCHK(hipMemsetAsync(d->dPbX, 0, coord, d->stream));        // entry 0 (unused) defined
    gecm_s2_init_args a;
    a.X = d->dX; a.Z = d->dZ; a.S = d->dS;
    a.PbX = d->dPbX;
    a.bx = d->dBlk; a.bz = d->dBlk + (coord / 4) * GECM_S2_BLK; a.bp = d->dBlk + (coord / 4) * 2 * GECM_S2_BLK;
    a.PdX = d->dPd; a.PdZ = d->dPd + coord / 4;
    a.acc = d->dAcc; a.fail = d->dFail; a.keep = d->dKeep;
    a.umax = umax; a.D = D; a.npb = npb; a.stride = d->stride;
    a.K = K;
    a.tgt = a.tgt_off = nullptr;
    a.kbx = a.kbz = a.kbp = a.PdKX = a.PdKZ = nullptr;
    if (K > 1) {
        const size_t kw = (coord / 4) * K * GECM_S2_BLK;
        a.tgt = d->dTgt; a.tgt_off = d->dTgt + tgt_off[K];
        a.kbx = d->dKBlk; a.kbz = d->dKBlk + kw; a.kbp = d->dKBlk + 2 * kw;
        a.PdKX = d->dPdK; a.PdKZ = d->dPdK + coord / 4;
    }
    gecm_modconst mc = modconst(d);
    HIPCHK(hipEventRecord(d->ev0, d->stream));
    switch (d->nl) {
#define X(n)                                             \
    case n:                                              \
        gecm_launch_s2_init_##n(d->stream, &mc, &a);     \
        if (d->s2_slices > 1) gecm_launch_s2_acc_init_##n(d->stream, &mc, d->dAcc, d->s2_slices, d->stride); \
        break;
        GECM_NL_LIST(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev1, d->stream));
    d->timed = true;
    return 0;
}

extern "C" int gecm_dev_s2_pair(gecm_dev *d, const uint32_t *steps, uint32_t nsteps, uint32_t D, uint32_t G,
                                uint32_t ring_size, uint64_t A0, uint64_t tape_id)
{
    HIPCHK(hipSetDevice(d->device));
    if (!d->dPbX || d->s2_G != G || d->s2_ring != ring_size || (ring_size & (ring_size - 1))) {
        g_err = "gecm_dev_s2_pair: stage-2 init has not run (or chunk/ring size changed)";
        return -2;
    }
    const size_t coord = (size_t)d->nl * d->stride * sizeof(uint32_t);
    size_t words = (size_t)nsteps * 2 + 2;
    if (words > d->steps_cap) {
        (void)hipFree(d->dSteps);
        d->dSteps = nullptr;
        d->steps_id = 0;
        HIPCHK(hipMalloc(&d->dSteps, words * 4));
        d->steps_cap = words;
    }
    // tape_id != 0 names a tape the host keeps from batch to batch: the device copy of the last one is kept too
    if (nsteps && !(tape_id && tape_id == d->steps_id && nsteps == d->steps_n))
        HIPCHK(hipMemcpyAsync(d->dSteps, steps, (size_t)nsteps * 8, hipMemcpyHostToDevice, d->stream));
    d->steps_id = tape_id;
    d->steps_n = nsteps;
    gecm_s2_pair_args a;
    a.X = d->dX; a.Z = d->dZ; a.S = d->dS; a.PbX = d->dPbX; a.npb = (uint32_t)d->s2_npb;
    a.PdX = d->dPd; a.PdZ = d->dPd + coord / 4;
    const size_t cw = coord / 4;
    a.gx = d->dPa; a.gz = a.gx + cw * ((size_t)G + 2); a.gp = a.gz + cw * ((size_t)G + 2); a.ring = a.gp + cw * (size_t)G;
    a.acc = d->dAcc; a.fail = d->dFail; a.steps = d->dSteps; a.host_steps = steps;
    a.nsteps = nsteps; a.D = D; a.G = G; a.ring_size = ring_size; a.A0 = A0; a.stride = d->stride;
    a.slices = d->s2_slices;
    a.K = d->s2_K; a.Gs = (uint32_t)(G / d->s2_K + 1);
    a.kgx = a.kgz = a.kgp = nullptr; a.PdKX = a.PdKZ = nullptr;
    if (d->s2_K > 1) {
        const size_t kcw = cw * d->s2_K;
        a.kgx = d->dKPa; a.kgz = a.kgx + kcw * ((size_t)a.Gs + 2); a.kgp = a.kgz + kcw * ((size_t)a.Gs + 2);
        a.PdKX = d->dPdK; a.PdKZ = d->dPdK + cw;
    }
    gecm_modconst mc = modconst(d);
    HIPCHK(hipEventRecord(d->ev0, d->stream));
    switch (d->nl) {
#define X(n)                                             \
    case n:                                              \
        gecm_launch_s2_pair_##n(d->stream, &mc, &a);     \
        break;
        GECM_NL_LIST(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev1, d->stream));
    d->timed = true;
    return 0;
}

extern "C" int gecm_dev_s2_download(gecm_dev *d, uint32_t *acc, uint32_t *fail)
{
    HIPCHK(hipSetDevice(d->device));
    if (!d->dAcc) {
        g_err = "gecm_dev_s2_download: no stage-2 state";
        return -2;
    }
    if (download_soa(d, acc, d->dAcc)) return -1;
    if (fail) {          /* gecm_dev_s2_fail_planes() planes of [limb][ncurves] */
        const uint32_t planes = gecm_dev_s2_fail_planes(d);
        for (uint32_t p = 0; p < planes; p++)
            if (download_soa(d, fail + (size_t)p * d->nl * d->ncurves, d->dFail + (size_t)p * d->nl * d->stride)) return -1;
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int gecm_dev_gcd_scan(gecm_dev *d, int which, uint32_t *flags, uint32_t *g)
{
    HIPCHK(hipSetDevice(d->device));
    const uint32_t *src = which == 0 ? d->dZ : d->dAcc;
    if (!src || !d->stride || d->r3.empty()) {
        g_err = "gecm_dev_gcd_scan: nothing to scan";
        return -2;
    }
    if (d->flags_cap < d->stride) {
        (void)hipFree(d->dFlags);
        d->dFlags = nullptr;
        HIPCHK(hipMalloc(&d->dFlags, d->stride * 4));
        d->flags_cap = d->stride;
    }
    gecm_modconst mc = modconst(d);
    switch (d->nl) {
#define X(n)                                                                             \
    case n:                                                                              \
        gecm_launch_gcd_scan_##n(d->stream, &mc, src, d->dT0, d->dFlags, d->stride);     \
        break;
        GECM_NL_LIST(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    if (flags && d->ncurves)
        HIPCHK(hipMemcpyAsync(flags, d->dFlags, d->ncurves * 4, hipMemcpyDeviceToHost, d->stream));
    if (g && download_soa(d, g, d->dT0)) return -1;
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}
