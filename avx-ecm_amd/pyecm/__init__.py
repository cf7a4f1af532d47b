"""pyecm — thin ctypes binding of libgecm's C ABI (include/gecm.h) for tests and bench.py.

Plumbing only: every call goes straight to the shared library; there is no Python fallback.
If libgecm.so is missing or fails to load, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GECM_LIB") or os.path.join(os.path.dirname(_HERE), "libgecm.so")

if not os.path.exists(LIB_PATH):
    raise ImportError("libgecm.so not built: run `make -C avx-ecm_amd -j8` (or __graft_entry__.build())")
lib = ctypes.CDLL(LIB_PATH)

c_void_p, c_int, c_size_t, c_char_p, c_u64, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                                         ctypes.c_char_p, ctypes.c_uint64, ctypes.c_double)


class Config(ctypes.Structure):
    _fields_ = [("digitbits", c_int), ("nwords", c_int), ("maxbits", c_int), ("nbits", c_int),
                ("dev_limbs", c_int), ("device", c_int), ("rho", c_u64)]


class Stage1Stats(ctypes.Structure):
    _fields_ = [("ptadds", c_u64), ("ptdups", c_u64), ("last_prime", c_u64), ("tape_len", c_u64)]


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


# every symbol include/gecm.h declares
EXPORTS = ["gecm_last_error", "gecm_device_count", "gecm_version", "gecm_create", "gecm_destroy",
           "gecm_get_config", "gecm_device_name", "gecm_get_one", "gecm_vecmulmod", "gecm_vecsqrmod",
           "gecm_vecaddmod", "gecm_vecsubmod", "gecm_vecaddsubmod", "gecm_build_curves",
           "gecm_upload_points", "gecm_stage1", "gecm_sync", "gecm_last_kernel_ms",
           "gecm_get_stage1_stats", "gecm_download_points", "gecm_download_points_plain",
           "gecm_format_save_line", "gecm_stage1_factor", "gecm_set_lanes_per_curve",
           "gecm_get_lanes_per_curve", "gecm_set_special_form", "gecm_get_special_form"]

_sig("gecm_last_error", c_char_p)
_sig("gecm_device_count", c_int)
_sig("gecm_version", c_char_p)
_sig("gecm_create", c_int, ctypes.POINTER(c_void_p), c_int, c_char_p, c_int)
_sig("gecm_destroy", None, c_void_p)
_sig("gecm_get_config", c_int, c_void_p, ctypes.POINTER(Config))
_sig("gecm_device_name", c_int, c_void_p, c_char_p, c_size_t)
_sig("gecm_get_one", c_int, c_void_p, c_void_p)
_sig("gecm_vecmulmod", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("gecm_vecsqrmod", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("gecm_vecaddmod", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("gecm_vecsubmod", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("gecm_vecaddsubmod", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("gecm_build_curves", c_int, c_void_p, ctypes.POINTER(c_u64), c_size_t)
_sig("gecm_upload_points", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)
_sig("gecm_stage1", c_int, c_void_p, c_u64)
_sig("gecm_sync", c_int, c_void_p)
_sig("gecm_set_lanes_per_curve", c_int, c_void_p, c_int)
_sig("gecm_get_lanes_per_curve", c_int, c_void_p)
_sig("gecm_set_special_form", c_int, c_void_p, c_int)
_sig("gecm_get_special_form", c_int, c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int))
_sig("gecm_last_kernel_ms", c_double, c_void_p)
_sig("gecm_get_stage1_stats", c_int, c_void_p, ctypes.POINTER(Stage1Stats))
_sig("gecm_download_points", c_int, c_void_p, c_void_p, c_void_p)
_sig("gecm_download_points_plain", c_int, c_void_p, c_void_p, c_void_p)
_sig("gecm_format_save_line", c_int, c_void_p, c_size_t, c_char_p, c_size_t)
_sig("gecm_stage1_factor", c_int, c_void_p, c_size_t, c_char_p, c_size_t, ctypes.POINTER(c_int))


class Stage2Stats(ctypes.Structure):
    _fields_ = [("ptadds", c_u64), ("numinv", c_u64), ("paired", c_u64), ("device_inversions", c_u64),
                ("D", ctypes.c_uint32), ("U", ctypes.c_uint32), ("L", ctypes.c_uint32), ("amin_last", ctypes.c_uint32)]


class Pairs(ctypes.Structure):
    _fields_ = [("pairmap_v", ctypes.POINTER(ctypes.c_uint32)), ("pairmap_u", ctypes.POINTER(ctypes.c_uint32)),
                ("steps", ctypes.c_uint32), ("amin", ctypes.c_uint32), ("pairs", ctypes.c_uint32),
                ("primes", ctypes.c_uint32)]


_sig("gecm_stage2_init", c_int, c_void_p, ctypes.c_uint32, ctypes.c_uint32)
_sig("gecm_pair_primes", c_int, ctypes.POINTER(Pairs), c_u64, c_u64, ctypes.c_uint32, ctypes.c_uint32)
_sig("gecm_pairmap_release", None, ctypes.POINTER(Pairs))
_sig("gecm_stage2_pair", c_int, c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
     ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32)
_sig("gecm_stage2", c_int, c_void_p, c_u64, ctypes.c_uint32, ctypes.c_uint32)
_sig("gecm_stage2_prepare", c_int, c_void_p, c_u64, ctypes.c_uint32, ctypes.c_uint32)
_sig("gecm_stage2_pair_prepare", c_int, c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
     ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32)
_sig("gecm_get_stage2_stats", c_int, c_void_p, ctypes.POINTER(Stage2Stats))
_sig("gecm_download_acc", c_int, c_void_p, c_void_p)
_sig("gecm_stage2_factor", c_int, c_void_p, c_size_t, c_char_p, c_size_t, ctypes.POINTER(c_int))
class RangeDesc(ctypes.Structure):
    _fields_ = [("lo", c_u64), ("hi", c_u64), ("nprimes", c_u64), ("first_prime", c_u64), ("last_prime", c_u64),
                ("checkpoint", c_int)]


_sig("gecm_set_report_modulus", c_int, c_void_p, c_char_p)
_sig("gecm_device_memory", c_int, c_void_p, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64))
_sig("gecm_batch_bytes", c_u64, c_void_p, c_size_t, c_int, c_u64, ctypes.c_uint32, ctypes.c_uint32)
EXPORTS += ["gecm_set_report_modulus", "gecm_device_memory", "gecm_batch_bytes"]
_sig("gecm_last_kernel_name", c_int, c_void_p, c_char_p, c_size_t)
_sig("gecm_stage1_progress", c_int, c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32))
EXPORTS += ["gecm_last_kernel_name", "gecm_stage1_progress"]
_sig("gecm_stage1_ranges", c_int, c_u64)
_sig("gecm_stage1_range", c_int, c_void_p, c_u64, ctypes.c_uint32)
_sig("gecm_stage1_describe_range", c_int, c_u64, c_u64, ctypes.c_uint32, ctypes.POINTER(RangeDesc))
_sig("gecm_format_resume_line", c_int, c_void_p, c_size_t, c_u64, c_char_p, c_size_t)
EXPORTS += ["gecm_stage1_ranges", "gecm_stage1_range", "gecm_stage1_describe_range", "gecm_format_resume_line"]
_sig("gecm_scan_factors", c_int, c_void_p, c_int, ctypes.POINTER(c_size_t))
_sig("gecm_curve_flag", c_int, c_void_p, c_int, c_size_t)
EXPORTS += ["gecm_scan_factors", "gecm_curve_flag", "gecm_prepare_input", "gecm_sizeinbase10"]
EXPORTS += ["gecm_stage2_init", "gecm_pair_primes", "gecm_pairmap_release", "gecm_stage2_pair", "gecm_stage2", "gecm_stage2_prepare", "gecm_stage2_pair_prepare",
            "gecm_get_stage2_stats", "gecm_download_acc", "gecm_stage2_factor"]


class GecmError(RuntimeError):
    pass


def _chk(rc, what):
    if rc < 0:
        raise GecmError("%s failed (%d): %s" % (what, rc, lib.gecm_last_error().decode()))
    return rc


def pair_primes(b1, b2, D, U):
    p = Pairs()
    _chk(lib.gecm_pair_primes(ctypes.byref(p), b1, b2, D, U), "gecm_pair_primes")
    return p


def device_count():
    return lib.gecm_device_count()


def stage1_ranges(b1):
    return lib.gecm_stage1_ranges(b1)


def describe_range(b1, b2, r):
    d = RangeDesc()
    _chk(lib.gecm_stage1_describe_range(b1, b2, r, ctypes.byref(d)), "gecm_stage1_describe_range")
    return d


class Engine:
    """One gecm_ctx: N + limb format + device."""

    def __init__(self, n, digitbits=52, device=0):
        self._h = c_void_p()
        _chk(lib.gecm_create(ctypes.byref(self._h), device, str(n).encode(), digitbits), "gecm_create")
        self.cfg = Config()
        _chk(lib.gecm_get_config(self._h, ctypes.byref(self.cfg)), "gecm_get_config")
        self.n = int(str(n), 0) if isinstance(n, str) else int(n)
        self.batch = 0

    def close(self):
        if self._h:
            lib.gecm_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference vec layout helpers: data[lane + limb*batch] ----
    def _ctype(self):
        return ctypes.c_uint64 if self.cfg.digitbits == 52 else ctypes.c_uint32

    def pack(self, values):
        b, nw, db = len(values), self.cfg.nwords, self.cfg.digitbits
        arr = (self._ctype() * (b * nw))()
        mask = (1 << db) - 1
        for lane, v in enumerate(values):
            for j in range(nw):
                arr[lane + j * b] = (v >> (db * j)) & mask
        return arr

    def unpack(self, arr, b):
        nw, db = self.cfg.nwords, self.cfg.digitbits
        return [sum(int(arr[lane + j * b]) << (db * j) for j in range(nw)) for lane in range(b)]

    def empty(self, b):
        return (self._ctype() * (b * self.cfg.nwords))()

    # ---- L0 ----
    def vecmulmod(self, a, b):
        n = len(a)
        out = self.empty(n)
        _chk(lib.gecm_vecmulmod(self._h, self.pack(a), self.pack(b), out, n), "gecm_vecmulmod")
        return self.unpack(out, n)

    def vecsqrmod(self, a):
        n = len(a)
        out = self.empty(n)
        _chk(lib.gecm_vecsqrmod(self._h, self.pack(a), out, n), "gecm_vecsqrmod")
        return self.unpack(out, n)

    def vecaddmod(self, a, b):
        n = len(a)
        out = self.empty(n)
        _chk(lib.gecm_vecaddmod(self._h, self.pack(a), self.pack(b), out, n), "gecm_vecaddmod")
        return self.unpack(out, n)

    def vecsubmod(self, a, b):
        n = len(a)
        out = self.empty(n)
        _chk(lib.gecm_vecsubmod(self._h, self.pack(a), self.pack(b), out, n), "gecm_vecsubmod")
        return self.unpack(out, n)

    def vecaddsubmod(self, a, b):
        n = len(a)
        s, d = self.empty(n), self.empty(n)
        _chk(lib.gecm_vecaddsubmod(self._h, self.pack(a), self.pack(b), s, d, n), "gecm_vecaddsubmod")
        return self.unpack(s, n), self.unpack(d, n)

    # ---- L1 ----
    def build_curves(self, sigmas):
        arr = (c_u64 * len(sigmas))(*sigmas)
        self.batch = len(sigmas)
        return _chk(lib.gecm_build_curves(self._h, arr, len(sigmas)), "gecm_build_curves")

    def upload_points(self, X, Z, s):
        self.batch = len(X)
        _chk(lib.gecm_upload_points(self._h, self.pack(X), self.pack(Z), self.pack(s), len(X)), "gecm_upload_points")

    def stage1(self, b1, sync=True):
        _chk(lib.gecm_stage1(self._h, b1), "gecm_stage1")
        if sync:
            self.sync()

    def stage1_range(self, b1, r, sync=True):
        """one ecm_stage1 call of the reference's loop over prime ranges of 1e8 (ecm.c:1209-1234)"""
        _chk(lib.gecm_stage1_range(self._h, b1, r), "gecm_stage1_range")
        if sync:
            self.sync()

    def resume_line(self, k, b1_field):
        buf = ctypes.create_string_buffer(8192)
        _chk(lib.gecm_format_resume_line(self._h, k, b1_field, buf, len(buf)), "gecm_format_resume_line")
        return buf.value.decode()

    def sync(self):
        _chk(lib.gecm_sync(self._h), "gecm_sync")

    def set_lanes_per_curve(self, lanes):
        """0 = chosen per launch from the batch size, 1 = curve per lane, 2 = X and Z on adjacent lanes,
        8 = X and Z on two quads of lanes with the limbs of each residue spread over the quad"""
        _chk(lib.gecm_set_lanes_per_curve(self._h, lanes), "gecm_set_lanes_per_curve")

    def set_special_form(self, on):
        """use (default) or not the 2^k - 1 multiply for N | 2^k - 1; applies from the next build/upload"""
        _chk(lib.gecm_set_special_form(self._h, 1 if on else 0), "gecm_set_special_form")

    def special_form(self):
        """(enabled, k, limbs)"""
        k, l = c_int(0), c_int(0)
        r = _chk(lib.gecm_get_special_form(self._h, ctypes.byref(k), ctypes.byref(l)), "gecm_get_special_form")
        return bool(r), k.value, l.value

    def special_form_used(self):
        """True if the last stage-1 launch ran with the special multiply"""
        return _chk(lib.gecm_get_special_form(self._h, None, None), "gecm_get_special_form") == 2

    def lanes_per_curve(self):
        """what the last stage-1 launch used"""
        return _chk(lib.gecm_get_lanes_per_curve(self._h), "gecm_get_lanes_per_curve")

    def device_memory(self):
        f, t = c_u64(0), c_u64(0)
        _chk(lib.gecm_device_memory(self._h, ctypes.byref(f), ctypes.byref(t)), "gecm_device_memory")
        return f.value, t.value

    def batch_bytes(self, curves, with_stage2=False, b1=0, D=0, U=0):
        return lib.gecm_batch_bytes(self._h, curves, 1 if with_stage2 else 0, b1, D, U)

    def set_report_modulus(self, n):
        """save lines name n and factors are reported of n (a divisor of the context's modulus): the reference's
        special-form runs, which work modulo 2^k -/+ 1 or 2^k - c and report against the number given"""
        _chk(lib.gecm_set_report_modulus(self._h, None if n is None else str(n).encode()), "gecm_set_report_modulus")

    def stage1_progress(self):
        """(launches finished, launches made) of the stage-1 call in flight"""
        d, t = ctypes.c_uint32(0), ctypes.c_uint32(0)
        _chk(lib.gecm_stage1_progress(self._h, ctypes.byref(d), ctypes.byref(t)), "gecm_stage1_progress")
        return d.value, t.value

    def last_kernel_name(self):
        buf = ctypes.create_string_buffer(128)
        _chk(lib.gecm_last_kernel_name(self._h, buf, len(buf)), "gecm_last_kernel_name")
        return buf.value.decode()

    def last_kernel_ms(self):
        return lib.gecm_last_kernel_ms(self._h)

    def stage1_stats(self):
        st = Stage1Stats()
        _chk(lib.gecm_get_stage1_stats(self._h, ctypes.byref(st)), "gecm_get_stage1_stats")
        return st

    def download_points(self):
        X, Z = self.empty(self.batch), self.empty(self.batch)
        _chk(lib.gecm_download_points(self._h, X, Z), "gecm_download_points")
        return self.unpack(X, self.batch), self.unpack(Z, self.batch)

    def download_points_plain(self):
        X, Z = self.empty(self.batch), self.empty(self.batch)
        _chk(lib.gecm_download_points_plain(self._h, X, Z), "gecm_download_points_plain")
        return self.unpack(X, self.batch), self.unpack(Z, self.batch)

    def save_line(self, k):
        buf = ctypes.create_string_buffer(8192)
        _chk(lib.gecm_format_save_line(self._h, k, buf, len(buf)), "gecm_format_save_line")
        return buf.value.decode()

    def save_lines(self):
        return [self.save_line(k) for k in range(self.batch)]

    def stage1_factor(self, k):
        buf = ctypes.create_string_buffer(2048)
        prp = c_int(0)
        rc = _chk(lib.gecm_stage1_factor(self._h, k, buf, len(buf), ctypes.byref(prp)), "gecm_stage1_factor")
        return (int(buf.value.decode()), bool(prp.value)) if rc == 1 else None

    def scan_factors(self, stage=1):
        """device factor scan: (number of curves with a factor, lowest such index or None)"""
        first = c_size_t(0)
        n = _chk(lib.gecm_scan_factors(self._h, stage, ctypes.byref(first)), "gecm_scan_factors")
        return n, (first.value if n else None)

    def curve_flag(self, stage, k):
        return bool(lib.gecm_curve_flag(self._h, stage, k))

    # ---- stage 2 ----
    def stage2(self, b2, D=0, U=0):
        _chk(lib.gecm_stage2(self._h, b2, D, U), "gecm_stage2")

    def stage2_prepare(self, b2, D=0, U=0):
        """host-side pair map for a later stage2(b2, D, U); callable while an asynchronous stage 1 runs"""
        _chk(lib.gecm_stage2_prepare(self._h, b2, D, U), "gecm_stage2_prepare")

    def stage2_init(self, D=0, U=0, sync=True):
        _chk(lib.gecm_stage2_init(self._h, D, U), "gecm_stage2_init")
        if sync:
            self.sync()

    def stage2_pair(self, pairs, sync=True):
        _chk(lib.gecm_stage2_pair(self._h, pairs.steps, pairs.pairmap_v, pairs.pairmap_u, pairs.amin), "gecm_stage2_pair")
        if sync:
            self.sync()

    def stage2_pair_prepare(self, pairs, D, U):
        """host side of stage2_pair(pairs) ahead of time (the launch tape of that pair map); no device work"""
        _chk(lib.gecm_stage2_pair_prepare(self._h, D, U, pairs.steps, pairs.pairmap_v, pairs.pairmap_u, pairs.amin),
             "gecm_stage2_pair_prepare")

    def stage2_stats(self):
        st = Stage2Stats()
        _chk(lib.gecm_get_stage2_stats(self._h, ctypes.byref(st)), "gecm_get_stage2_stats")
        return st

    def download_acc(self):
        A = self.empty(self.batch)
        _chk(lib.gecm_download_acc(self._h, A), "gecm_download_acc")
        return self.unpack(A, self.batch)

    def stage2_factor(self, k):
        buf = ctypes.create_string_buffer(2048)
        prp = c_int(0)
        rc = _chk(lib.gecm_stage2_factor(self._h, k, buf, len(buf), ctypes.byref(prp)), "gecm_stage2_factor")
        return (int(buf.value.decode()), bool(prp.value)) if rc == 1 else None

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        _chk(lib.gecm_device_name(self._h, buf, len(buf)), "gecm_device_name")
        return buf.value.decode()
