#!/usr/bin/env python3
"""What pinning a caller's pageable buffer on the fly would cost against the staged copy (GPU box): hipHostRegister / DMA / hipHostUnregister
of 32 MiB and 256 MiB buffers, next to the library's staged sh_dev_upload / sh_dev_download of the same bytes."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from starks_amd import _lib
L, ctx = _lib.lib(), _lib.ctx()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
for mib in (32, 256):
    nb = mib << 20
    d = ctypes.c_void_p()
    _lib.check(L.sh_dev_alloc(ctx, nb, ctypes.byref(d)), "alloc")
    for trial in range(3):
        src = bytes(os.urandom(1 << 20)) * mib          # an immutable bytes object, as the call sites pass
        dst = ctypes.create_string_buffer(nb)            # a fresh, untouched destination
        p = ctypes.cast(ctypes.c_char_p(src), ctypes.c_void_p)
        t0 = time.perf_counter(); rc1 = hip.hipHostRegister(p, nb, 0); t1 = time.perf_counter()
        rc2 = hip.hipMemcpy(d, p, nb, 1); t2 = time.perf_counter()
        rc3 = hip.hipHostUnregister(p); t3 = time.perf_counter()
        q = ctypes.cast(dst, ctypes.c_void_p)
        rc4 = hip.hipHostRegister(q, nb, 0); t4 = time.perf_counter()
        rc5 = hip.hipMemcpy(q, d, nb, 2); t5 = time.perf_counter()
        rc6 = hip.hipHostUnregister(q); t6 = time.perf_counter()
        same = dst.raw == src
        dst2 = ctypes.create_string_buffer(nb)
        t7 = time.perf_counter(); _lib.check(L.sh_dev_upload(ctx, src, d, nb), "ul"); t8 = time.perf_counter()
        _lib.check(L.sh_dev_download(ctx, d, dst2, nb), "dl"); t9 = time.perf_counter()
        print("%4d MiB  up: register %.2f + dma %.2f + unregister %.2f ms (rc %d %d %d) | down: register %.2f + dma %.2f + unregister %.2f ms (rc %d %d %d) same %s"
              " | staged: up %.2f ms, down %.2f ms" % (mib, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, rc1, rc2, rc3, (t4-t3)*1e3, (t5-t4)*1e3, (t6-t5)*1e3,
                                                      rc4, rc5, rc6, same, (t8-t7)*1e3, (t9-t8)*1e3), flush=True)
    L.sh_dev_free(ctx, d)
