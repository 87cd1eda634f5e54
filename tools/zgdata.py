"""Synthetic workloads and a libzstd-backed compressor to turn them into .zst inputs.

The reference's own encoder cannot be built here (Rust) and only implements level "Fastest"; the inputs of the
bench are produced by the C libzstd that ships in the image (the same library the reference's Readme compares
against). libzstd is used ONLY to create inputs and as a context CPU baseline — never on the decode path.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_GEN = None
_ZSTD = None


def _gen():
    global _GEN
    if _GEN is None:
        so = os.path.join(HERE, "libzgdata.so")
        src = os.path.join(HERE, "zgdata.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
        L = C.CDLL(so)
        L.zgdata_iso_like.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.zgdata_text_like.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32]
        _GEN = L
    return _GEN


def text_like(n, seed=0xE9, V=14000):
    buf = C.create_string_buffer(n)
    assert _gen().zgdata_text_like(buf, n, seed, V) == 0
    return buf.raw


def iso_like(n, seed=0x150):
    buf = C.create_string_buffer(n)
    _gen().zgdata_iso_like(buf, n, seed)
    return buf.raw


def libzstd():
    global _ZSTD
    if _ZSTD is None:
        for cand in ("/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so"):
            try:
                # DEEPBIND: under rocprofv3 another libzstd is already loaded; this copy must bind its internals to itself
                L = C.CDLL(cand, mode=os.RTLD_NOW | getattr(os, "RTLD_DEEPBIND", 0))
                break
            except OSError:
                L = None
        if L is None:
            raise RuntimeError("libzstd not found (needed to create .zst inputs)")
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_CCtx_setParameter.restype = C.c_size_t
        L.ZSTD_compress2.restype = C.c_size_t
        L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_decompress.restype = C.c_size_t
        L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_versionNumber.restype = C.c_uint
        _ZSTD = L
    return _ZSTD


def zstd_compress(data, level=3, checksum=True, content_size=True, window_log=0):
    """one frame, single thread (ZSTD_compress2)"""
    L = libzstd()
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, level)              # ZSTD_c_compressionLevel
    L.ZSTD_CCtx_setParameter(cctx, 200, 1 if content_size else 0)  # ZSTD_c_contentSizeFlag
    L.ZSTD_CCtx_setParameter(cctx, 201, 1 if checksum else 0)      # ZSTD_c_checksumFlag
    if window_log:
        L.ZSTD_CCtx_setParameter(cctx, 101, window_log)      # ZSTD_c_windowLog
    cap = L.ZSTD_compressBound(len(data))
    dst = C.create_string_buffer(cap)
    n = L.ZSTD_compress2(cctx, dst, cap, data, len(data))
    L.ZSTD_freeCCtx(cctx)
    if L.ZSTD_isError(n):
        raise RuntimeError("ZSTD_compress2 failed")
    return dst.raw[:n]


def zstd_decompress(z, size):
    L = libzstd()
    dst = C.create_string_buffer(max(size, 1))
    n = L.ZSTD_decompress(dst, size, z, len(z))
    if L.ZSTD_isError(n):
        raise RuntimeError("ZSTD_decompress failed")
    return dst.raw[:n]


def zstd_version():
    v = libzstd().ZSTD_versionNumber()
    return "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100)
