"""ctypes binding of the CPU oracle (oracle/libzstd_oracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

ZOR_OK = 0
ZOR_SKIP_FRAME = 1
ZOR_WINDOW_SIZE_TOO_BIG = 6
ZOR_DICT_NOT_PROVIDED = 7
ZOR_FAILED_READ_BLOCK_BODY = 10
ZOR_TARGET_TOO_SMALL = 12
ZOR_FAILED_SKIP_FRAME = 13
ZOR_REF_PANIC = 90
STRAT_ALL, STRAT_UPTO_BLOCKS, STRAT_UPTO_BYTES = 0, 1, 2


class Sequence(C.Structure):
    _fields_ = [("ll", C.c_uint32), ("ml", C.c_uint32), ("of", C.c_uint32), ("actual_of", C.c_uint32)]


class FseEntry(C.Structure):
    _fields_ = [("base_line", C.c_uint32), ("num_bits", C.c_uint8), ("symbol", C.c_uint8), ("pad", C.c_uint8 * 2)]


class HufEntry(C.Structure):
    _fields_ = [("symbol", C.c_uint8), ("num_bits", C.c_uint8)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "libzstd_oracle.so")
    src = os.path.join(ORACLE_DIR, "zstd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    L = C.CDLL(so)
    u8p, szp, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)
    L.zor_new.restype = C.c_void_p
    L.zor_free.argtypes = [C.c_void_p]
    L.zor_set_max_window_size.argtypes = [C.c_void_p, C.c_uint64]
    L.zor_add_dict.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, u32p]
    L.zor_force_dict.argtypes = [C.c_void_p, C.c_uint32]
    L.zor_init.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, szp, u32p, u32p]
    L.zor_decode_blocks.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, szp, C.c_int, C.c_size_t, C.POINTER(C.c_int)]
    L.zor_can_collect.argtypes = [C.c_void_p]
    L.zor_can_collect.restype = C.c_size_t
    L.zor_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zor_collect.restype = C.c_size_t
    L.zor_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zor_read.restype = C.c_size_t
    L.zor_decode_all.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, szp]
    L.zor_held.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zor_held.restype = C.c_size_t
    L.zor_decode_from_to.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, szp, szp]
    for nm in ("zor_blocks_decoded", "zor_bytes_read_from_source", "zor_content_size", "zor_window_size"):
        getattr(L, nm).argtypes = [C.c_void_p]
        getattr(L, nm).restype = C.c_uint64
    L.zor_is_finished.argtypes = [C.c_void_p]
    L.zor_checksum_from_data.argtypes = [C.c_void_p, u32p]
    L.zor_calculated_checksum.argtypes = [C.c_void_p]
    L.zor_calculated_checksum.restype = C.c_uint32
    L.zor_dict_id.argtypes = [C.c_void_p]
    L.zor_dict_id.restype = C.c_uint32
    L.zor_last_block_type.argtypes = [C.c_void_p]
    L.zor_last_literals.argtypes = [C.c_void_p, szp]
    L.zor_last_literals.restype = C.POINTER(C.c_uint8)
    L.zor_last_sequences.argtypes = [C.c_void_p, szp]
    L.zor_last_sequences.restype = C.POINTER(Sequence)
    L.zor_offset_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 3)]
    L.zor_fse_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(FseEntry)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.zor_fse_table.restype = C.c_size_t
    L.zor_huf_table.argtypes = [C.c_void_p, C.POINTER(C.POINTER(HufEntry)), C.POINTER(C.c_int)]
    L.zor_huf_table.restype = C.c_size_t
    L.zor_fse_build_from_probs.argtypes = [C.c_int, C.POINTER(C.c_int32), C.c_size_t, C.c_int, C.POINTER(FseEntry)]
    L.zor_fse_build_decoder.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(FseEntry), C.POINTER(C.c_int), szp]
    L.zor_huf_build_decoder.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(HufEntry), C.POINTER(C.c_int), u32p]
    L.zor_do_offset_history.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32 * 3)]
    L.zor_do_offset_history.restype = C.c_uint32
    L.zor_revbits_read.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.zor_revbits_read.restype = C.c_int64
    L.zor_fwdbits_read.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.zor_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    L.zor_xxh64.restype = C.c_uint64
    _LIB = L
    return L


class OracleError(Exception):
    def __init__(self, status):
        super().__init__("oracle status %d" % status)
        self.status = status


class FrameDecoder:
    """Mirrors ruzstd's FrameDecoder (decoding/frame_decoder.rs) through the oracle."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.zor_new()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.zor_free(self.h)
            self.h = None

    def set_max_window_size(self, n):
        self.L.zor_set_max_window_size(self.h, n)

    def add_dict(self, raw):
        did = C.c_uint32()
        st = self.L.zor_add_dict(self.h, raw, len(raw), C.byref(did))
        if st:
            raise OracleError(st)
        return did.value

    def force_dict(self, dict_id):
        return self.L.zor_force_dict(self.h, dict_id)

    def init(self, src):
        """returns (status, consumed, skip_magic, skip_len)"""
        c, sm, sl = C.c_size_t(), C.c_uint32(), C.c_uint32()
        st = self.L.zor_init(self.h, src, len(src), C.byref(c), C.byref(sm), C.byref(sl))
        return st, c.value, sm.value, sl.value

    def decode_blocks(self, src, strat=STRAT_ALL, n=0):
        """returns (status, consumed, finished)"""
        c, fin = C.c_size_t(), C.c_int()
        st = self.L.zor_decode_blocks(self.h, src, len(src), C.byref(c), strat, n, C.byref(fin))
        return st, c.value, bool(fin.value)

    def can_collect(self):
        return self.L.zor_can_collect(self.h)

    def collect(self):
        n = self.can_collect()
        buf = C.create_string_buffer(max(n, 1))
        got = self.L.zor_collect(self.h, buf, n)
        return buf.raw[:got]

    def read(self, cap):
        buf = C.create_string_buffer(max(cap, 1))
        got = self.L.zor_read(self.h, buf, cap)
        return buf.raw[:got]

    def held(self, cap=1 << 28):
        """(test accessor) a copy of everything the decode buffer holds, window included; nothing is drained"""
        n = self.L.zor_held(self.h, None, cap)
        buf = C.create_string_buffer(max(n, 1))
        n = self.L.zor_held(self.h, buf, n)
        return buf.raw[:n]

    def decode_from_to(self, src, cap):
        """returns (status, bytes_read, output_bytes) — frame_decoder.rs:439-529"""
        buf = C.create_string_buffer(max(cap, 1))
        r, w = C.c_size_t(), C.c_size_t()
        st = self.L.zor_decode_from_to(self.h, src, len(src), buf, cap, C.byref(r), C.byref(w))
        return st, r.value, buf.raw[:w.value]

    def decode_all(self, src, cap):
        buf = C.create_string_buffer(max(cap, 1))
        w = C.c_size_t()
        st = self.L.zor_decode_all(self.h, src, len(src), buf, cap, C.byref(w))
        return st, buf.raw[:w.value]

    def is_finished(self):
        return bool(self.L.zor_is_finished(self.h))

    def blocks_decoded(self):
        return self.L.zor_blocks_decoded(self.h)

    def bytes_read_from_source(self):
        return self.L.zor_bytes_read_from_source(self.h)

    def content_size(self):
        return self.L.zor_content_size(self.h)

    def window_size(self):
        return self.L.zor_window_size(self.h)

    def checksum_from_data(self):
        v = C.c_uint32()
        return v.value if self.L.zor_checksum_from_data(self.h, C.byref(v)) else None

    def calculated_checksum(self):
        return self.L.zor_calculated_checksum(self.h)

    # --- intermediates of the last decoded block ---
    def last_block_type(self):
        return self.L.zor_last_block_type(self.h)

    def last_literals(self):
        n = C.c_size_t()
        p = self.L.zor_last_literals(self.h, C.byref(n))
        return bytes(bytearray(p[:n.value])) if n.value else b""

    def last_sequences(self):
        n = C.c_size_t()
        p = self.L.zor_last_sequences(self.h, C.byref(n))
        return [(p[i].ll, p[i].ml, p[i].of, p[i].actual_of) for i in range(n.value)]

    def offset_hist(self):
        a = (C.c_uint32 * 3)()
        self.L.zor_offset_hist(self.h, C.byref(a))
        return list(a)

    def fse_table(self, which):
        e, al, rle = C.POINTER(FseEntry)(), C.c_int(), C.c_int()
        n = self.L.zor_fse_table(self.h, which, C.byref(e), C.byref(al), C.byref(rle))
        return [(e[i].base_line, e[i].num_bits, e[i].symbol) for i in range(n)], al.value, rle.value

    def huf_table(self):
        e, mb = C.POINTER(HufEntry)(), C.c_int()
        n = self.L.zor_huf_table(self.h, C.byref(e), C.byref(mb))
        return [(e[i].symbol, e[i].num_bits) for i in range(n)], mb.value


def decode_frame_all(src, dict_raw=None, max_window=None):
    """reset + decode_blocks(All) + collect, as decode_corpus.rs:51-58. Returns (bytes, decoder)."""
    d = FrameDecoder()
    if max_window:
        d.set_max_window_size(max_window)
    if dict_raw is not None:
        d.add_dict(dict_raw)
    st, c, _, _ = d.init(src)
    if st:
        raise OracleError(st)
    st, c2, fin = d.decode_blocks(src[c:], STRAT_ALL)
    if st:
        raise OracleError(st)
    return d.collect(), d
