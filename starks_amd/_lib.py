"""ctypes binding of libstarkhip.so (C ABI: include/starkhip.h).

There is NO CPU fallback: if the shared library is missing, or no MI355X is visible, every entry point
raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` (or `make -C starks_amd/csrc`).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STARKHIP_LIB") or os.path.join(_HERE, "libstarkhip.so")

MIMC_P = 2**256 - 2**32 * 351 + 1  # starks/utils.py:22

_lib = None
_ctx = None
_ctx2 = None
_lock = threading.Lock()


class StarkHipError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, _strerror(status), status)
        if detail:
            msg += " -- " + detail
        super().__init__(msg)


def _strerror(status):
    try:
        return lib().sh_strerror(status).decode()
    except Exception:  # pragma: no cover
        return "status %d" % status


def lib():
    """Load the shared library (no device needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "starks_amd: %s is missing -- the HIP extension has not been built. Run "
            "`python -c \"import __graft_entry__ as g; g.build()\"` or `make -C starks_amd/csrc`. "
            "There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c_p, u8p, u64, u32, i32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    pp = ctypes.POINTER(c_p)
    sig = {
        "sh_strerror": (ctypes.c_char_p, [i32]),
        "sh_version": (ctypes.c_char_p, []),
        "sh_ctx_create": (i32, [i32, pp]),
        "sh_ctx_destroy": (None, [c_p]),
        "sh_last_error": (ctypes.c_char_p, [c_p]),
        "sh_device_count": (i32, []),
        "sh_sync": (i32, [c_p]),
        "sh_timer_start": (i32, [c_p]),
        "sh_timer_stop": (i32, [c_p, ctypes.POINTER(ctypes.c_float)]),
        "sh_ntt": (i32, [c_p, c_p, u64, c_p, u64, u8p, i32]),
        "sh_ntt_batch": (i32, [c_p, c_p, u64, c_p, u64, u32, u8p, i32]),
        "sh_mul_polys": (i32, [c_p, u8p, u64, u8p, u64, c_p, u64, u8p]),
        "sh_power_cycle": (i32, [c_p, u8p, u64, c_p]),
        "sh_lde": (i32, [c_p, u8p, c_p, u64, u32, u32, u8p]),
        "sh_merkelize": (i32, [c_p, u8p, u64, c_p]),
        "sh_merkelize_packed": (i32, [c_p, u8p, u64, u32, c_p, c_p]),
        "sh_fri_fold": (i32, [c_p, u8p, u64, u8p, u8p, c_p]),
        "sh_fri_proof_len": (u64, [u64, u64, u32]),
        "sh_fri_prove": (i32, [c_p, u8p, u64, u64, u8p, u64, u32, u32, u32, c_p, u64]),
        "sh_dev_alloc": (i32, [c_p, u64, pp]),
        "sh_dev_free": (i32, [c_p, c_p]),
        "sh_dev_from_wire": (i32, [c_p, u8p, c_p, u64]),
        "sh_dev_to_wire": (i32, [c_p, c_p, c_p, u64]),
        "sh_dev_download": (i32, [c_p, c_p, c_p, u64]),
        "sh_dev_upload": (i32, [c_p, u8p, c_p, u64]),
        "sh_dev_copy": (i32, [c_p, c_p, c_p, u64]),
        "sh_fri_verify": (i32, [u8p, u64, u8p, u64, u8p, u64, u32, u32]),
        "sh_stark_verify": (i32, [u8p, u64, u8p, u8p, u64, u32, u32, u8p, u8p, ctypes.POINTER(u32), u32]),
        "sh_dev_download_async": (i32, [c_p, c_p, c_p, u64]),
        "sh_io_sync": (i32, [c_p]),
        "sh_host_alloc": (i32, [c_p, u64, pp]),
        "sh_host_free": (i32, [c_p, c_p]),
        "sh_ntt_passes": (u32, [u64, u32]),
        "sh_dev_download_2d": (i32, [c_p, c_p, u64, c_p, u64, u64]),
        "sh_dev_fill_seeded": (i32, [c_p, c_p, u64, u64]),
        "sh_dev_ntt": (i32, [c_p, c_p, c_p, u64, u32, u8p, i32]),
        "sh_dev_lde": (i32, [c_p, c_p, c_p, u64, u32, u32, u8p]),
        "sh_dev_merkelize": (i32, [c_p, c_p, u64, u32, c_p]),
        "sh_dev_fri_fold": (i32, [c_p, c_p, c_p, u64, u32, u8p, c_p]),
        "sh_dev_fri_prove": (i32, [c_p, c_p, u64, u8p, u64, u32, u32, u32, c_p]),
        "sh_dev_fri_prove_coeffs": (i32, [c_p, c_p, u64, u64, u8p, u64, u32, u32, u32, c_p]),
        "sh_stark_proof_len": (u64, [u64, u32, u32, u32, u32]),
        "sh_stark_prove": (i32, [c_p, u8p, u8p, u64, u32, u32, u8p, u8p, c_p, u32, u32, c_p, u64]),
        "sh_dev_stark_prove": (i32, [c_p, c_p, c_p, u64, u32, u32, u8p, u8p, c_p, u32, u32, c_p]),
        "sh_stark_status": (i32, [c_p]),
        "sh_stark_status_batch": (i32, [c_p, c_p, u32]),
        "sh_ctx_trim": (i32, [c_p]),
        "sh_ctx_set_plan_budget": (i32, [c_p, u64]),
        "sh_ctx_stats": (i32, [c_p, ctypes.POINTER(u64)]),
        "sh_dev_fill_mimc_units": (i32, [c_p, c_p, c_p, u64, u32, u32, u32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    L._sig = sig
    _lib = L
    return L


EXPORTED = None  # filled lazily by exported_symbols()


def exported_symbols():
    """Names bound above == every function include/starkhip.h declares (checked by tests)."""
    return sorted(lib()._sig.keys())


def default_device():
    for var in ("STARKHIP_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


def ctx():
    """The process-wide context (one process per GPU; device = $STARKHIP_DEVICE or $LOCAL_RANK or 0)."""
    global _ctx
    with _lock:
        if _ctx is None:
            L = lib()
            h = ctypes.c_void_p()
            rc = L.sh_ctx_create(default_device(), ctypes.byref(h))
            if rc != 0:
                raise StarkHipError(rc, "sh_ctx_create", "an MI355X (gfx950) GPU is required; there is no CPU fallback")
            _ctx = h
        return _ctx


def second_ctx():
    """A second context on the same GPU (its own stream, workspaces and plan cache; include/starkhip.h: contexts are
    independent), created on first use and kept for the life of the process: the many-proof driver pipelines two batches."""
    global _ctx2
    ctx()
    with _lock:
        if _ctx2 is None:
            h = ctypes.c_void_p()
            rc = lib().sh_ctx_create(default_device(), ctypes.byref(h))
            if rc != 0:
                raise StarkHipError(rc, "sh_ctx_create", "second context")
            _ctx2 = h
        return _ctx2


def close():
    global _ctx, _ctx2
    with _lock:
        if _ctx2 is not None:
            lib().sh_ctx_destroy(_ctx2)
            _ctx2 = None
        if _ctx is not None:
            lib().sh_ctx_destroy(_ctx)
            _ctx = None


def check(rc, where):
    if rc != 0:
        detail = ""
        if _ctx is not None:
            detail = lib().sh_last_error(_ctx).decode()
        raise StarkHipError(rc, where, detail)


class PinnedBuffer(object):
    """`nbytes` of page-locked host memory (sh_host_alloc): a buffer the host-buffer entry points copy to and from the
    GPU without the staging memcpy.  Usable wherever bytes-like data is passed (`buf.view` is a ctypes char array)."""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        check(lib().sh_host_alloc(ctx(), nbytes, ctypes.byref(self.ptr)), "sh_host_alloc")
        self.nbytes = nbytes
        self.view = (ctypes.c_char * nbytes).from_address(self.ptr.value)

    def close(self):
        """Frees the buffer; `view` is dropped with it.  The page-locked allocation belongs to the process, not to the context
        (sh_ctx_destroy frees no caller buffers): after `_lib.close()` it is released through the context-less form
        sh_host_free(NULL, ptr) instead of being leaked."""
        ptr, self.ptr, self.view = self.ptr, None, None
        if ptr:
            rc = lib().sh_host_free(_ctx, ptr)  # _ctx may be None
            if _ctx is not None:
                check(rc, "sh_host_free")
            elif rc != 0:
                raise StarkHipError(rc, "sh_host_free", "")

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def to_wire(values, modulus=MIMC_P):
    """list of ints / field elements -> concatenated 32-byte big-endian strings (modp.py:94-95).

    Values already in [0, 2^256) are sent as they are (an element built from bytes may be unreduced,
    modp.py:33-34; the device reduces); anything else is reduced first, as `field(int)` would.

    One C-level `int.to_bytes` per value and one join (0.3 us per value; the per-value slice assignment this replaces
    took twice that); plain ints in range -- the common case -- take the first branch without any per-value test."""
    wb = getattr(values, "wire_bytes", None)
    if wb is not None:  # a WireList (wireseq.py): the bytes the device wrote, as they are
        return wb()
    if not isinstance(values, (list, tuple)):
        values = list(values)  # an iterator must not be half consumed by the fast path before the general one starts over
    try:
        return b"".join([v.to_bytes(32, "big") for v in values])
    except (AttributeError, OverflowError, TypeError):
        pass  # field elements (no int.to_bytes signature), negative or >= 2^256 values: the general path
    lim = 1 << 256
    ints = [int(v) for v in values]
    return b"".join([(x if 0 <= x < lim else x % modulus).to_bytes(32, "big") for x in ints])


def from_wire(buf):
    """concatenated 32-byte big-endian strings -> list of ints (0.17 us per value)."""
    mv, conv = memoryview(buf), int.from_bytes
    return [conv(mv[i:i + 32], "big") for i in range(0, len(mv), 32)]


def order_of_root(root, modulus=MIMC_P):
    """Multiplicative order of `root` if it is a power of two <= 2^32 (the 2-adic part of p-1), else None.
    The reference gets the same number by walking the powers until they return to 1 (fft.py:319-321)."""
    t = int(root) % modulus
    n = 1
    for _ in range(33):
        if t == 1:
            return n
        t = t * t % modulus
        n *= 2
    return None
