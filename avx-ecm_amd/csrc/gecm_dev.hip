// gecm_dev.hip — device management for libgecm (gfx950 / MI355X only): buffers, stream, events,
// dispatch to the per-limb-count kernel launchers of gecm_kernels.hip.
#include "gecm_dev.h"
#include "gecm_launch.h"
#include <hip/hip_runtime.h>
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

struct gecm_dev {
    int device = 0;
    int nl = 0;
    std::vector<uint32_t> n, kp, one;
    uint32_t rho = 0;
    size_t ncurves = 0, stride = 0;
    uint32_t *dX = nullptr, *dZ = nullptr, *dS = nullptr, *dT0 = nullptr, *dT1 = nullptr;
    uint32_t *dTape = nullptr;
    size_t tape_len = 0, tape_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    bool timed = false;
};

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
    mc.rho = d->rho;
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
    gecm_dev *d = new gecm_dev;
    d->device = device;
    d->nl = nl;
    d->n.assign(n, n + nl);
    d->kp.assign(kp, kp + nl);
    d->one.assign(one, one + nl);
    d->rho = rho;
    HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&d->ev0));
    HIPCHK(hipEventCreate(&d->ev1));
    *out = d;
    return 0;
}

static void free_state(gecm_dev *d)
{
    hipFree(d->dX); hipFree(d->dZ); hipFree(d->dS); hipFree(d->dT0); hipFree(d->dT1);
    d->dX = d->dZ = d->dS = d->dT0 = d->dT1 = nullptr;
}

extern "C" void gecm_dev_close(gecm_dev *d)
{
    if (!d) return;
    hipSetDevice(d->device);
    free_state(d);
    hipFree(d->dTape);
    if (d->ev0) hipEventDestroy(d->ev0);
    if (d->ev1) hipEventDestroy(d->ev1);
    if (d->stream) hipStreamDestroy(d->stream);
    delete d;
}

extern "C" int gecm_dev_device_name(gecm_dev *d, char *buf, size_t len)
{
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, d->device));
    snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
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

extern "C" int gecm_dev_set_tape(gecm_dev *d, const uint8_t *tape, size_t len)
{
    HIPCHK(hipSetDevice(d->device));
    size_t words = (len + 3) / 4 + 1;
    if (words > d->tape_cap) {
        hipFree(d->dTape);
        d->dTape = nullptr;
        HIPCHK(hipMalloc(&d->dTape, words * 4));
        d->tape_cap = words;
    }
    HIPCHK(hipMemsetAsync(d->dTape, 0, words * 4, d->stream));
    if (len) HIPCHK(hipMemcpyAsync(d->dTape, tape, len, hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    d->tape_len = len;
    return 0;
}

extern "C" int gecm_dev_stage1(gecm_dev *d)
{
    HIPCHK(hipSetDevice(d->device));
    if (!d->stride || !d->dTape) {
        g_err = "gecm_dev_stage1: no curves or no tape";
        return -2;
    }
    gecm_modconst mc = modconst(d);
    HIPCHK(hipEventRecord(d->ev0, d->stream));
    switch (d->nl) {
#define X(n)                                                                                     \
    case n:                                                                                      \
        gecm_launch_stage1_##n(d->stream, &mc, d->dTape, (uint32_t)d->tape_len, d->dX, d->dZ,    \
                               d->dS, d->stride);                                                \
        break;
        GECM_NL_LIST(X)
#undef X
    default:
        g_err = "unsupported nl";
        return -2;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev1, d->stream));
    d->timed = true;
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
