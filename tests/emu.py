"""ctypes binding of the TEST-ONLY host harness (tests/emu/libzg_emu.so): the engine's lane routines and host parser
run on the CPU. Used by the not-gpu tests to check the decode logic against the oracle and the golden fixtures."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Seq(C.Structure):
    _fields_ = [("of", C.c_uint32), ("ml", C.c_uint32), ("mdst", C.c_uint32), ("lit_start", C.c_uint32)]


def lib():
    global _LIB
    if _LIB is None:
        d = os.path.join(HERE, "emu")
        subprocess.check_call(["make", "-C", d, "-s"])
        L = C.CDLL(os.environ.get("ZG_EMU_LIB") or os.path.join(d, "libzg_emu.so"))   # (ZG_EMU_LIB: e.g. the address-sanitizer build, `make -C tests/emu asan`)
        L.zgemu_decode.restype = C.c_void_p
        L.zgemu_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.zgemu_decode2.restype = C.c_void_p
        L.zgemu_decode2.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int]
        L.zgemu_free.argtypes = [C.c_void_p]
        L.zgemu_parse_status.argtypes = [C.c_void_p]
        L.zgemu_num_frames.argtypes = [C.c_void_p]
        L.zgemu_num_blocks.argtypes = [C.c_void_p]
        L.zgemu_frame.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.zgemu_output.restype = C.POINTER(C.c_uint8)
        L.zgemu_output.argtypes = [C.c_void_p]
        L.zgemu_block_status.argtypes = [C.c_void_p, C.c_uint32]
        L.zgemu_block_status.restype = C.c_uint32
        L.zgemu_block.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 12)]
        L.zgemu_block_literals.restype = C.POINTER(C.c_uint8)
        L.zgemu_block_literals.argtypes = [C.c_void_p, C.c_uint32]
        L.zgemu_block_sequences.restype = C.POINTER(Seq)
        L.zgemu_block_sequences.argtypes = [C.c_void_p, C.c_uint32]
        L.zgemu_block_hist.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 3)]
        L.zgemu_fse_slot.restype = C.POINTER(C.c_uint32)
        L.zgemu_fse_slot.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8 * 4)]
        L.zgemu_huf_slot.restype = C.POINTER(C.c_uint16)
        L.zgemu_huf_slot.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
        L.zgemu_plan.restype = C.c_void_p
        L.zgemu_plan.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32]
        for f in ("zgemu_num_units", "zgemu_num_steps", "zgemu_num_step_units"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_uint32
        L.zgemu_unit.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 4)]
        L.zgemu_step.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 3)]
        L.zgemu_step_unit.argtypes = [C.c_void_p, C.c_uint32]
        L.zgemu_step_unit.restype = C.c_uint32
        L.zgemu_frame_plan.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 7)]
        L.zgemu_seq_block.argtypes = [C.c_void_p, C.c_uint32]
        L.zgemu_seq_block.restype = C.c_uint32
        L.zgemu_exact.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


class Plan:
    """The host's plan for the LZ77 stages of a submit (BatchBuilder::finish): units, sweep steps, per-frame sequence ranges."""

    def __init__(self, src, flat_slots=256, unit_blocks=0):
        L = lib()
        h = L.zgemu_plan(src, len(src), flat_slots, unit_blocks)
        try:
            assert L.zgemu_parse_status(h) == 0
            self.nblocks = L.zgemu_num_blocks(h)
            self.nseq = []
            info = (C.c_uint32 * 12)()
            for b in range(self.nblocks):
                L.zgemu_block(h, b, C.byref(info))
                self.nseq.append(info[5] if info[0] == 2 else 0)          # compressed blocks only (ZG_BT_COMPRESSED == 2)
            u4, s3, f7 = (C.c_uint32 * 4)(), (C.c_uint32 * 3)(), (C.c_uint32 * 7)()
            self.units = []
            for u in range(L.zgemu_num_units(h)):
                L.zgemu_unit(h, u, C.byref(u4))
                self.units.append(tuple(u4))                               # (frame, first_block, nblocks, noseq)
            self.steps = []
            for i in range(L.zgemu_num_steps(h)):
                L.zgemu_step(h, i, C.byref(s3))
                self.steps.append(tuple(s3))                               # (list_off, nunits, max_blocks)
            self.step_units = [L.zgemu_step_unit(h, i) for i in range(L.zgemu_num_step_units(h))]
            self.frames = []
            for f in range(L.zgemu_num_frames(h)):
                L.zgemu_frame_plan(h, f, C.byref(f7))
                self.frames.append(tuple(f7))                              # (first_block, nblocks, first_unit, nunits, seq_first, seq_count, sparse)
            nsb = sum(fr[5] for fr in self.frames)
            self.seq_blocks = [L.zgemu_seq_block(h, i) for i in range(nsb)]
        finally:
            L.zgemu_free(h)


def decode_all_verdict(m, max_window=128 << 20):
    """zgpu_decode_all's verdict rule on the CPU harness: the first failing frame in stream order (zg_k_exact's verdict where it has
    one), else the error of the host walk — what lies in front of the point where the walk stops is decoded first (zg_capi.cpp)"""
    e = EmuBatch(m, max_window=max_window)
    if e.nframes == 0 or e.nblocks == 0:
        return e.parse_status
    ex = e.exact(drain_rule=1)
    for f in range(e.nframes):
        if ex[f][0]:
            return ex[f][0]
        if e.frame(f)[2]:
            return e.frame(f)[2]
    return e.parse_status


class EmuBatch:
    def __init__(self, src, max_window=128 << 20, fast_seq=True):
        self.L = lib()
        self.h = self.L.zgemu_decode2(src, len(src), max_window, 1 if fast_seq else 0)
        self.parse_status = self.L.zgemu_parse_status(self.h)
        self.nframes = self.L.zgemu_num_frames(self.h)
        self.nblocks = self.L.zgemu_num_blocks(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.zgemu_free(self.h)
            self.h = None

    def frame(self, f):
        b, s, st, bb = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        assert self.L.zgemu_frame(self.h, f, C.byref(b), C.byref(s), C.byref(st), C.byref(bb)) == 0
        return b.value, s.value, st.value, bb.value

    def frame_bytes(self, f):
        b, s, st, _ = self.frame(f)
        p = self.L.zgemu_output(self.h)
        return C.string_at(C.addressof(p.contents) + b, s), st

    def exact(self, drain_rule=0, dict_len=0, prior_out=0, prior_reach=0, prior_counted=0):
        """zg_k_exact's source (zg_exact.h) on this submit: [(status, bad_block, counted)] per frame; status 0 = the reference's
        DecodeBuffer bookkeeping has nothing to object to in the blocks the entropy stages accepted"""
        n = self.nframes
        st, bad, cnt = (C.c_uint32 * n)(), (C.c_uint32 * n)(), (C.c_uint64 * n)()
        assert self.L.zgemu_exact(self.h, drain_rule, dict_len, prior_out, prior_reach, prior_counted, st, bad, cnt) == 0
        return [(st[i], bad[i], cnt[i]) for i in range(n)]

    def block(self, b):
        a = (C.c_uint32 * 12)()
        self.L.zgemu_block(self.h, b, C.byref(a))
        keys = ["btype", "lit_type", "nstreams", "seq_modes", "regen_size", "nseq", "frame", "huf_slot", "ll_slot", "of_slot", "ml_slot", "active"]
        d = dict(zip(keys, list(a)))
        for k in ("huf_slot", "ll_slot", "of_slot", "ml_slot"):
            if d[k] >= 1 << 31:
                d[k] -= 1 << 32
        d["status"] = self.L.zgemu_block_status(self.h, b)
        return d

    def block_literals(self, b, n):
        p = self.L.zgemu_block_literals(self.h, b)
        return C.string_at(p, n)

    def block_sequences(self, b, n):
        p = self.L.zgemu_block_sequences(self.h, b)
        return [(p[i].of, p[i].ml, p[i].mdst, p[i].lit_start) for i in range(n)]

    def block_hist(self, b):
        a = (C.c_uint32 * 3)()
        self.L.zgemu_block_hist(self.h, b, C.byref(a))
        return list(a)

    def fse_slot(self, slot):
        lg = (C.c_uint8 * 4)()
        p = self.L.zgemu_fse_slot(self.h, slot, C.byref(lg))
        return p, list(lg)

    def huf_slot(self, slot):
        mb = C.c_int()
        p = self.L.zgemu_huf_slot(self.h, slot, C.byref(mb))
        return p, mb.value
