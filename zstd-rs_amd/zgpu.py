"""ctypes binding of libzgpu.so (include/zgpu.h) + thin Python mirrors of the reference's decoder surface.

Names follow ruzstd (FrameDecoder, StreamingDecoder, decode_all, decode_blocks, collect ...). There is no CPU path:
if libzgpu.so is missing or no gfx950 device is usable, construction raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZGPU_LIB") or os.path.join(HERE, "libzgpu.so")   # ZGPU_LIB: a profiling build (tools/dev)
# the development build (-DZG_DEV_SWITCHES): the only library that reads the ZGPU_* measurement / test switches. Tests that force a path
# and tools/dev ask for it explicitly (Context(dev=True)); the product library ignores the environment.
DEV_LIB_PATH = os.path.join(HERE, "libzgpu_dev.so")
_LIB = None
_DEV_LIB = None

STRAT_ALL, STRAT_UPTO_BLOCKS, STRAT_UPTO_BYTES = 0, 1, 2
E_SKIP_FRAME = 1
E_WINDOW_SIZE_TOO_BIG = 6
E_DICT_NOT_PROVIDED = 7
E_FAILED_READ_BLOCK_HEADER, E_FAILED_READ_BLOCK_BODY, E_FAILED_READ_CHECKSUM = 9, 10, 11
E_TARGET_TOO_SMALL = 12
E_FAILED_SKIP_FRAME = 13
E_RESERVED_BLOCK, E_BLOCK_SIZE_TOO_LARGE = 20, 21
E_UNSUPPORTED = 80
E_HIP = 92


class FrameInfo(C.Structure):
    _fields_ = [("src_begin", C.c_uint64), ("src_end", C.c_uint64), ("window_size", C.c_uint64), ("frame_content_size", C.c_uint64),
                ("out_base", C.c_uint64), ("out_size", C.c_uint64), ("nblocks", C.c_uint32), ("status", C.c_uint32),
                ("bad_block", C.c_uint32), ("has_checksum", C.c_uint32), ("checksum", C.c_uint32), ("pad", C.c_uint32)]


class BlockInfo(C.Structure):
    _fields_ = [("btype", C.c_uint32), ("lit_type", C.c_uint32), ("nstreams", C.c_uint32), ("seq_modes", C.c_uint32),
                ("regen_size", C.c_uint32), ("nseq", C.c_uint32), ("frame", C.c_uint32), ("status", C.c_uint32),
                ("huf_slot", C.c_int32), ("ll_slot", C.c_int32), ("of_slot", C.c_int32), ("ml_slot", C.c_int32),
                ("sum_ll", C.c_uint32), ("sum_ml", C.c_uint32), ("hist_init", C.c_uint32 * 3), ("active", C.c_uint32),
                ("out_base", C.c_uint64)]


class Seq(C.Structure):
    _fields_ = [("of", C.c_uint32), ("ml", C.c_uint32), ("mdst", C.c_uint32), ("lit_start", C.c_uint32)]


class Block(C.Structure):
    """zgpu_block: one host-parsed Block_Header (include/zgpu.h, the thin boundary)"""
    _fields_ = [("src_off", C.c_uint64), ("src_len", C.c_uint32), ("raw_rle_size", C.c_uint32), ("type", C.c_uint8), ("last", C.c_uint8),
                ("pad", C.c_uint8 * 6)]


class StreamOpts(C.Structure):
    """zgpu_stream_opts (include/zgpu.h)"""
    _fields_ = [("read_ahead_bytes", C.c_uint64), ("no_checksum", C.c_uint32), ("copy_threads", C.c_uint32), ("pipe_after_bytes", C.c_uint64),
                ("first_run_blocks", C.c_uint32), ("pad", C.c_uint32)]


NO_READ_AHEAD = 1

EXPORTS = [
    "zgpu_ctx_create", "zgpu_ctx_destroy", "zgpu_set_max_window_size", "zgpu_max_window_size", "zgpu_last_error", "zgpu_status_name",
    "zgpu_decode_all", "zgpu_batch_prepare", "zgpu_batch_run", "zgpu_batch_sync", "zgpu_batch_num_frames", "zgpu_batch_num_blocks",
    "zgpu_batch_compressed_size", "zgpu_batch_frame_info", "zgpu_batch_read", "zgpu_batch_output_device", "zgpu_batch_timings",
    "zgpu_batch_destroy", "zgpu_batch_block_info", "zgpu_batch_block_literals", "zgpu_batch_block_sequences", "zgpu_batch_fse_slot",
    "zgpu_batch_huf_slot", "zgpu_batch_debug_timers", "zgpu_batch_num_units", "zgpu_batch_debug_sweep_mode", "zgpu_batch_unit", "zgpu_batch_debug_scratch", "zgpu_debug_calibrate", "zgpu_add_dict", "zgpu_decoder_force_dict",
    "zgpu_decoder_decode_from_to", "zgpu_decoder_create", "zgpu_decoder_destroy", "zgpu_decoder_init", "zgpu_decoder_decode_blocks",
    "zgpu_decoder_can_collect", "zgpu_decoder_collect", "zgpu_decoder_read", "zgpu_decoder_is_finished", "zgpu_decoder_blocks_decoded",
    "zgpu_decoder_bytes_read_from_source", "zgpu_decoder_content_size", "zgpu_decoder_checksum_from_data",
    "zgpu_decoder_calculated_checksum", "zgpu_decode_all_alloc", "zgpu_free", "zgpu_decoder_collect_to_writer", "zgpu_streaming_create",
    "zgpu_streaming_destroy", "zgpu_streaming_decoder", "zgpu_streaming_read", "zgpu_pool_create", "zgpu_pool_create_on", "zgpu_pool_destroy",
    "zgpu_pool_num_gpus", "zgpu_pool_decode_all", "zgpu_pool_plan", "zgpu_pool_stage", "zgpu_pool_run", "zgpu_pool_frame", "zgpu_pool_read", "zgpu_pool_timings", "zgpu_pool_plan_stats",
    "zgpu_frame_begin", "zgpu_frame_end", "zgpu_blocks_submit", "zgpu_sync", "zgpu_available", "zgpu_read", "zgpu_device_output",
    "zgpu_frame_checksum", "zgpu_frame_blocks_decoded", "zgpu_decoder_device_bytes", "zgpu_debug_tuning",
    "zgpu_streaming_create_ex", "zgpu_streaming_create_slice", "zgpu_streaming_source_position", "zgpu_streaming_copy", "zgpu_streaming_stats",
    "zgpu_decoder_set_hash", "zgpu_decoder_set_read_ahead", "zgpu_release_caches",
]
WRITE_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
READ_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)


def load_library(dev=False):
    """Load libzgpu.so (dev=True: libzgpu_dev.so) and declare the prototypes. Does not touch the GPU."""
    global _LIB, _DEV_LIB
    if dev:
        if _DEV_LIB is None:
            if not os.path.exists(DEV_LIB_PATH):
                raise RuntimeError("libzgpu_dev.so is not built (run __graft_entry__.build())")
            _DEV_LIB = _declare(C.CDLL(DEV_LIB_PATH))
        return _DEV_LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libzgpu.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    _LIB = _declare(C.CDLL(LIB_PATH))
    return _LIB


def _declare(L):
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.c_char_p
    P = C.POINTER
    L.zgpu_ctx_create.argtypes = [C.c_int, P(vp)]
    L.zgpu_ctx_destroy.argtypes = [vp]
    L.zgpu_set_max_window_size.argtypes = [vp, C.c_uint64]
    L.zgpu_max_window_size.argtypes = [vp]
    L.zgpu_max_window_size.restype = C.c_uint64
    L.zgpu_last_error.argtypes = [vp]
    L.zgpu_last_error.restype = C.c_char_p
    L.zgpu_status_name.argtypes = [C.c_int]
    L.zgpu_status_name.restype = C.c_char_p
    L.zgpu_decode_all.argtypes = [vp, u8p, sz, vp, sz, P(sz)]
    L.zgpu_batch_prepare.argtypes = [vp, u8p, sz, P(vp)]
    L.zgpu_batch_run.argtypes = [vp]
    L.zgpu_batch_sync.argtypes = [vp, P(C.c_uint64), P(C.c_uint32), P(C.c_uint32)]
    L.zgpu_batch_num_frames.argtypes = [vp]
    L.zgpu_batch_num_frames.restype = C.c_uint32
    L.zgpu_batch_num_blocks.argtypes = [vp]
    L.zgpu_batch_num_blocks.restype = C.c_uint32
    L.zgpu_batch_compressed_size.argtypes = [vp]
    L.zgpu_batch_compressed_size.restype = C.c_uint64
    L.zgpu_batch_frame_info.argtypes = [vp, C.c_uint32, P(FrameInfo)]
    L.zgpu_batch_read.argtypes = [vp, C.c_uint64, vp, C.c_uint64]
    L.zgpu_batch_output_device.argtypes = [vp]
    L.zgpu_batch_output_device.restype = vp
    L.zgpu_batch_timings.argtypes = [vp, P(C.c_float), C.c_int]
    L.zgpu_batch_destroy.argtypes = [vp]
    L.zgpu_batch_block_info.argtypes = [vp, C.c_uint32, P(BlockInfo)]
    L.zgpu_batch_block_literals.argtypes = [vp, C.c_uint32, vp, sz, P(sz)]
    L.zgpu_batch_block_sequences.argtypes = [vp, C.c_uint32, P(Seq), sz, P(sz)]
    L.zgpu_batch_fse_slot.argtypes = [vp, C.c_uint32, P(C.c_uint32), P(C.c_uint8)]
    L.zgpu_batch_huf_slot.argtypes = [vp, C.c_uint32, P(C.c_uint16), P(C.c_int)]
    L.zgpu_batch_debug_timers.argtypes = [vp, P(C.c_uint64)]
    L.zgpu_batch_num_units.argtypes = [vp]
    L.zgpu_batch_num_units.restype = C.c_uint32
    L.zgpu_batch_debug_sweep_mode.argtypes = [vp]
    L.zgpu_batch_debug_sweep_mode.restype = C.c_uint32
    L.zgpu_batch_unit.argtypes = [vp, C.c_uint32, P(C.c_uint32), P(C.c_uint32), P(C.c_uint64)]
    L.zgpu_batch_debug_scratch.argtypes = [vp, C.c_int, C.c_uint64, vp, C.c_uint64]
    L.zgpu_debug_calibrate.argtypes = [vp, C.c_uint64]
    L.zgpu_frame_begin.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, P(vp)]
    L.zgpu_frame_end.argtypes = [vp]
    L.zgpu_blocks_submit.argtypes = [vp, u8p, sz, P(Block), sz]
    L.zgpu_sync.argtypes = [vp, P(sz), P(C.c_int32)]
    L.zgpu_available.argtypes = [vp, C.c_int]
    L.zgpu_available.restype = sz
    L.zgpu_read.argtypes = [vp, vp, sz, C.c_int, P(sz)]
    L.zgpu_device_output.argtypes = [vp, P(vp), P(sz)]
    L.zgpu_frame_checksum.argtypes = [vp]
    L.zgpu_frame_checksum.restype = C.c_uint32
    L.zgpu_frame_blocks_decoded.argtypes = [vp]
    L.zgpu_frame_blocks_decoded.restype = C.c_uint64
    L.zgpu_decoder_device_bytes.argtypes = [vp]
    L.zgpu_decoder_device_bytes.restype = C.c_uint64
    L.zgpu_decoder_create.argtypes = [vp, P(vp)]
    L.zgpu_add_dict.argtypes = [vp, u8p, sz, P(C.c_uint32)]
    L.zgpu_decoder_force_dict.argtypes = [vp, C.c_uint32]
    L.zgpu_decoder_decode_from_to.argtypes = [vp, u8p, sz, vp, sz, P(sz), P(sz)]
    L.zgpu_decoder_destroy.argtypes = [vp]
    L.zgpu_decoder_init.argtypes = [vp, u8p, sz, P(sz), P(C.c_uint32), P(C.c_uint32)]
    L.zgpu_decoder_decode_blocks.argtypes = [vp, u8p, sz, P(sz), C.c_int, sz, P(C.c_int)]
    L.zgpu_decoder_can_collect.argtypes = [vp]
    L.zgpu_decoder_can_collect.restype = sz
    L.zgpu_decoder_collect.argtypes = [vp, vp, sz]
    L.zgpu_decoder_collect.restype = sz
    L.zgpu_decoder_read.argtypes = [vp, vp, sz]
    L.zgpu_decoder_read.restype = sz
    L.zgpu_decoder_is_finished.argtypes = [vp]
    for nm in ("zgpu_decoder_blocks_decoded", "zgpu_decoder_bytes_read_from_source", "zgpu_decoder_content_size"):
        getattr(L, nm).argtypes = [vp]
        getattr(L, nm).restype = C.c_uint64
    L.zgpu_decoder_checksum_from_data.argtypes = [vp, P(C.c_uint32)]
    L.zgpu_decoder_calculated_checksum.argtypes = [vp]
    L.zgpu_decoder_calculated_checksum.restype = C.c_uint32
    L.zgpu_decode_all_alloc.argtypes = [vp, u8p, sz, P(vp), P(sz)]
    L.zgpu_free.argtypes = [vp]
    L.zgpu_decoder_collect_to_writer.argtypes = [vp, WRITE_FN, vp, P(sz)]
    L.zgpu_streaming_create.argtypes = [vp, READ_FN, vp, P(vp)]
    L.zgpu_streaming_destroy.argtypes = [vp]
    L.zgpu_streaming_decoder.argtypes = [vp]
    L.zgpu_streaming_decoder.restype = vp
    L.zgpu_streaming_read.argtypes = [vp, vp, sz, P(sz)]
    L.zgpu_pool_create.argtypes = [C.c_int, P(vp)]
    L.zgpu_pool_create_on.argtypes = [P(C.c_int), C.c_int, P(vp)]
    L.zgpu_pool_destroy.argtypes = [vp]
    L.zgpu_pool_num_gpus.argtypes = [vp]
    L.zgpu_pool_decode_all.argtypes = [vp, u8p, sz, vp, sz, P(sz)]
    L.zgpu_pool_plan.argtypes = [P(C.c_uint64), C.c_uint32, C.c_uint32, P(C.c_uint32), P(C.c_uint32), P(C.c_uint64)]
    L.zgpu_pool_stage.argtypes = [vp, P(C.c_char_p), P(sz), C.c_uint32]
    L.zgpu_pool_run.argtypes = [vp, P(C.c_float), P(C.c_float)]
    L.zgpu_pool_frame.argtypes = [vp, C.c_uint32, P(C.c_int), P(C.c_uint64), P(C.c_uint32)]
    L.zgpu_pool_read.argtypes = [vp, C.c_uint32, vp, sz, P(sz)]
    L.zgpu_pool_timings.argtypes = [vp, C.c_uint32, P(C.c_float), C.c_int, P(C.c_uint64), P(C.c_uint64), P(C.c_uint32), P(C.c_uint32)]
    L.zgpu_pool_plan_stats.argtypes = [vp, C.c_uint32, P(C.c_uint64), C.c_int]
    L.zgpu_debug_tuning.argtypes = [vp, P(C.c_uint32), C.c_int]
    L.zgpu_streaming_create_ex.argtypes = [vp, READ_FN, vp, P(StreamOpts), P(vp)]
    L.zgpu_streaming_create_slice.argtypes = [vp, vp, sz, P(StreamOpts), P(vp)]
    L.zgpu_streaming_source_position.argtypes = [vp]
    L.zgpu_streaming_source_position.restype = sz
    L.zgpu_streaming_copy.argtypes = [vp, sz, vp, vp, P(C.c_uint64)]
    L.zgpu_streaming_stats.argtypes = [vp, P(C.c_uint64), C.c_int]
    L.zgpu_decoder_set_hash.argtypes = [vp, C.c_int]
    L.zgpu_decoder_set_read_ahead.argtypes = [vp, C.c_uint64]
    for f in ("zgpu_decoder_is_finished", "zgpu_decoder_checksum_from_data"):
        getattr(L, f).argtypes = [vp] if f.endswith("finished") else [vp, P(C.c_uint32)]
    return L


class ZgpuError(Exception):
    def __init__(self, status, what=""):
        self.status = status
        name = load_library().zgpu_status_name(status).decode()
        super().__init__("%s (status %d) %s" % (name, status, what))


class Context:
    """One engine per GPU (zgpu_ctx)."""

    def __init__(self, device=0, dev=False):
        self.L = load_library(dev)
        h = C.c_void_p()
        st = self.L.zgpu_ctx_create(device, C.byref(h))
        if st:
            raise ZgpuError(st, "zgpu_ctx_create: no usable MI355X/HIP device — the engine has no CPU path")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.zgpu_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def tuning(self):
        """the switches this context's engine took when it was created (zgpu_debug_tuning): all defaults in the product library"""
        a = (C.c_uint32 * 8)()
        n = self.L.zgpu_debug_tuning(self.h, a, 8)
        keys = ["dev_build", "unit_blocks", "seq_packed", "flat4", "ramp_percent", "sweep_w", "flat_shape", "force_inorder"]
        d = dict(zip(keys[:n], [int(x) for x in a][:n]))
        for k in ("seq_packed", "flat4"):
            if d.get(k, 0) >= 1 << 31:
                d[k] -= 1 << 32
        return d

    def set_max_window_size(self, n):
        self.L.zgpu_set_max_window_size(self.h, n)

    def max_window_size(self):
        return self.L.zgpu_max_window_size(self.h)

    def add_dict(self, raw):
        """FrameDecoder::add_dict (frame_decoder.rs:224-227); returns the dictionary id"""
        did = C.c_uint32()
        st = self.L.zgpu_add_dict(self.h, raw, len(raw), C.byref(did))
        if st:
            raise ZgpuError(st)
        return did.value

    def decode_all(self, src, cap):
        """FrameDecoder::decode_all (frame_decoder.rs:541-577). Returns the plaintext or raises ZgpuError."""
        w = C.c_size_t()
        try:                                   # no zero-fill, one copy: matters for GB-sized outputs
            import numpy as np
            arr = np.empty(max(cap, 1), dtype=np.uint8)
            st = self.L.zgpu_decode_all(self.h, src, len(src), arr.ctypes.data_as(C.c_void_p), cap, C.byref(w))
            if st:
                raise ZgpuError(st)
            return arr[:w.value].tobytes()
        except ImportError:
            buf = C.create_string_buffer(max(cap, 1))
            st = self.L.zgpu_decode_all(self.h, src, len(src), buf, cap, C.byref(w))
            if st:
                raise ZgpuError(st)
            return buf.raw[:w.value]

    def decode_all_to_vec(self, src):
        """FrameDecoder::decode_all_to_vec (frame_decoder.rs:591-610): the library sizes the output"""
        out, n = C.c_void_p(), C.c_size_t()
        st = self.L.zgpu_decode_all_alloc(self.h, src, len(src), C.byref(out), C.byref(n))
        if st:
            raise ZgpuError(st)
        try:
            return C.string_at(out, n.value)
        finally:
            self.L.zgpu_free(out)

    def prepare(self, src):
        return Batch(self, src)


class Batch:
    """A run of whole frames resident on the device (zgpu_batch)."""

    def __init__(self, ctx, src):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        self.parse_status = self.L.zgpu_batch_prepare(ctx.h, src, len(src), C.byref(h))
        if not h:
            raise ZgpuError(self.parse_status)
        self.h = h
        self.nframes = self.L.zgpu_batch_num_frames(h)
        self.nblocks = self.L.zgpu_batch_num_blocks(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.zgpu_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def run(self):
        st = self.L.zgpu_batch_run(self.h)
        if st:
            raise ZgpuError(st, self.L.zgpu_last_error(self.ctx.h).decode())

    def sync(self):
        tot, bf, bs = C.c_uint64(), C.c_uint32(), C.c_uint32()
        st = self.L.zgpu_batch_sync(self.h, C.byref(tot), C.byref(bf), C.byref(bs))
        if st and st != E_UNSUPPORTED:
            raise ZgpuError(st, self.L.zgpu_last_error(self.ctx.h).decode())
        self.total_out, self.bad_frame, self.bad_status = tot.value, bf.value, bs.value or st
        return self.total_out

    def timings(self):
        a = (C.c_float * 10)()
        self.L.zgpu_batch_timings(self.h, a, 10)
        return dict(zip(["tables", "huf", "seq", "seqpost", "scan", "lit", "flat", "sweep", "lz", "total"], list(a)))

    def debug_timers(self):
        a = (C.c_uint64 * 1024)()
        self.L.zgpu_batch_debug_timers(self.h, a)
        return list(a)

    def sweep_mode(self):
        """after sync: 0 plain chain of sweep steps, 1 split into tails and heads, 2 split and then repeated as a plain chain"""
        return int(self.L.zgpu_batch_debug_sweep_mode(self.h))

    def units(self):
        """[(first_block, nblocks, scratch_base, size, noseq)] — the units zg_k_flatten worked on (size valid after sync); noseq: bit 0: no
        block of the unit has sequences, bit 1: direct unit (resolved to bytes by the flatten itself); either way it has no scratch
        words and no sweep step"""
        out = []
        for u in range(self.L.zgpu_batch_num_units(self.h)):
            fb, nb, base = C.c_uint32(), C.c_uint32(), C.c_uint64()
            assert self.L.zgpu_batch_unit(self.h, u, C.byref(fb), C.byref(nb), C.byref(base)) == 0
            info = (C.c_uint32 * 4)()
            assert self.L.zgpu_batch_debug_scratch(self.h, 1, 16 * u, info, 16) == 0
            out.append((fb.value, nb.value, base.value, info[0], int(info[1])))
        return out

    def scratch_words(self, base, n):
        """n effective offsets (u32) of the flatten scratch starting at word `base`, as a numpy array"""
        import numpy as np
        arr = np.empty(max(n, 1), dtype=np.uint32)
        st = self.L.zgpu_batch_debug_scratch(self.h, 0, 4 * base, arr.ctypes.data_as(C.c_void_p), 4 * n)
        if st:
            raise ZgpuError(st)
        return arr[:n]

    def frame_info(self, f):
        fi = FrameInfo()
        assert self.L.zgpu_batch_frame_info(self.h, f, C.byref(fi)) == 0
        return fi

    def read(self, off, n):
        buf = C.create_string_buffer(max(n, 1))
        st = self.L.zgpu_batch_read(self.h, off, buf, n)
        if st:
            raise ZgpuError(st)
        return buf.raw[:n]

    def frame_bytes(self, f):
        fi = self.frame_info(f)
        return self.read(fi.out_base, fi.out_size)

    def output_device_ptr(self):
        return self.L.zgpu_batch_output_device(self.h)

    def block_info(self, b):
        bi = BlockInfo()
        st = self.L.zgpu_batch_block_info(self.h, b, C.byref(bi))
        if st:
            raise ZgpuError(st)
        return bi

    def block_literals(self, b, n):
        buf = C.create_string_buffer(max(n, 1))
        got = C.c_size_t()
        st = self.L.zgpu_batch_block_literals(self.h, b, buf, n, C.byref(got))
        if st:
            raise ZgpuError(st)
        return buf.raw[:got.value]

    def block_sequences(self, b, n):
        arr = (Seq * max(n, 1))()
        got = C.c_size_t()
        st = self.L.zgpu_batch_block_sequences(self.h, b, arr, n, C.byref(got))
        if st:
            raise ZgpuError(st)
        return [(arr[i].of, arr[i].ml, arr[i].mdst, arr[i].lit_start) for i in range(got.value)]

    def fse_slot(self, slot):
        ent = (C.c_uint32 * 1280)()
        lg = (C.c_uint8 * 4)()
        st = self.L.zgpu_batch_fse_slot(self.h, slot, ent, lg)
        if st:
            raise ZgpuError(st)
        return ent, list(lg)

    def huf_slot(self, slot):
        ent = (C.c_uint16 * 2048)()
        mb = C.c_int()
        st = self.L.zgpu_batch_huf_slot(self.h, slot, ent, C.byref(mb))
        if st:
            raise ZgpuError(st)
        return ent, mb.value


class FrameDecoder:
    """Mirror of ruzstd::decoding::FrameDecoder (frame_decoder.rs:80-627) on the GPU engine."""

    def __init__(self, ctx=None):
        self.ctx = ctx or Context()
        self.L = self.ctx.L
        h = C.c_void_p()
        st = self.L.zgpu_decoder_create(self.ctx.h, C.byref(h))
        if st:
            raise ZgpuError(st)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.zgpu_decoder_destroy(self.h)
            self.h = None

    __del__ = close

    def set_max_window_size(self, n):
        self.ctx.set_max_window_size(n)

    def init(self, src):
        """reset(): returns (status, consumed, skip_magic, skip_len); SkipFrame is a status, as in the reference"""
        c, sm, sl = C.c_size_t(), C.c_uint32(), C.c_uint32()
        st = self.L.zgpu_decoder_init(self.h, src, len(src), C.byref(c), C.byref(sm), C.byref(sl))
        return st, c.value, sm.value, sl.value

    reset = init

    def decode_blocks(self, src, strat=STRAT_ALL, n=0):
        c, fin = C.c_size_t(), C.c_int()
        st = self.L.zgpu_decoder_decode_blocks(self.h, src, len(src), C.byref(c), strat, n, C.byref(fin))
        return st, c.value, bool(fin.value)

    def add_dict(self, raw):
        return self.ctx.add_dict(raw)

    def force_dict(self, dict_id):
        return self.L.zgpu_decoder_force_dict(self.h, dict_id)

    def decode_from_to(self, src, cap):
        """returns (status, bytes_read, output_bytes) — frame_decoder.rs:439-529"""
        buf = C.create_string_buffer(max(cap, 1))
        r, w = C.c_size_t(), C.c_size_t()
        st = self.L.zgpu_decoder_decode_from_to(self.h, src, len(src), buf, cap, C.byref(r), C.byref(w))
        return st, r.value, buf.raw[:w.value]

    def can_collect(self):
        return self.L.zgpu_decoder_can_collect(self.h)

    def collect(self):
        n = self.can_collect()
        buf = C.create_string_buffer(max(n, 1))
        got = self.L.zgpu_decoder_collect(self.h, buf, n)
        return buf.raw[:got]

    def read(self, cap):
        buf = C.create_string_buffer(max(cap, 1))
        got = self.L.zgpu_decoder_read(self.h, buf, cap)
        return buf.raw[:got]

    def is_finished(self):
        return bool(self.L.zgpu_decoder_is_finished(self.h))

    def set_hash(self, on):
        self.L.zgpu_decoder_set_hash(self.h, 1 if on else 0)

    def set_read_ahead(self, n):
        self.L.zgpu_decoder_set_read_ahead(self.h, n)

    def blocks_decoded(self):
        return self.L.zgpu_decoder_blocks_decoded(self.h)

    def bytes_read_from_source(self):
        return self.L.zgpu_decoder_bytes_read_from_source(self.h)

    def content_size(self):
        return self.L.zgpu_decoder_content_size(self.h)

    def get_checksum_from_data(self):
        v = C.c_uint32()
        return v.value if self.L.zgpu_decoder_checksum_from_data(self.h, C.byref(v)) else None

    def get_calculated_checksum(self):
        return self.L.zgpu_decoder_calculated_checksum(self.h)

    def decode_all(self, src, cap):
        return self.ctx.decode_all(src, cap)

    def collect_to_writer(self, writer):
        """collect_to_writer (frame_decoder.rs:395-407); writer.write(bytes) -> number of bytes taken"""
        def wr(_user, data, n):
            return writer.write(C.string_at(data, n))
        cb = WRITE_FN(wr)
        done = C.c_size_t()
        st = self.L.zgpu_decoder_collect_to_writer(self.h, cb, None, C.byref(done))
        if st:
            raise ZgpuError(st)
        return done.value


def plan(costs, n_workers):
    """the work queue's plan, host only: (LPT order, worker of every job, load per worker) — zgpu_pool_plan"""
    L = load_library()
    n = len(costs)
    c = (C.c_uint64 * max(n, 1))(*costs)
    order, worker, load = (C.c_uint32 * max(n, 1))(), (C.c_uint32 * max(n, 1))(), (C.c_uint64 * n_workers)()
    st = L.zgpu_pool_plan(c, n, n_workers, order, worker, load)
    if st:
        raise ZgpuError(st)
    return list(order)[:n], list(worker)[:n], list(load)


class Pool:
    """Frames over the GPUs of one node through the library's work queue (zgpu_pool): one worker thread + engine per GPU."""

    def __init__(self, n_gpus=0, devices=None, dev=False):
        self.L = load_library(dev)
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            st = self.L.zgpu_pool_create_on(arr, len(devices), C.byref(h))
        else:
            st = self.L.zgpu_pool_create(n_gpus, C.byref(h))
        if st:
            raise ZgpuError(st, "zgpu_pool_create: no usable MI355X/HIP device — the engine has no CPU path")
        self.h = h
        self.n_gpus = self.L.zgpu_pool_num_gpus(h)
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.zgpu_pool_destroy(self.h)
            self.h = None

    __del__ = close

    def decode_all(self, src, cap):
        import numpy as np
        w = C.c_size_t()
        arr = np.empty(max(cap, 1), dtype=np.uint8)
        st = self.L.zgpu_pool_decode_all(self.h, src, len(src), arr.ctypes.data_as(C.c_void_p), cap, C.byref(w))
        if st:
            raise ZgpuError(st)
        return arr[:w.value].tobytes()

    def stage(self, frames):
        n = len(frames)
        ptrs = (C.c_char_p * max(n, 1))(*frames)
        lens = (C.c_size_t * max(n, 1))(*[len(f) for f in frames])
        self._keep = frames
        st = self.L.zgpu_pool_stage(self.h, ptrs, lens, n)
        if st:
            raise ZgpuError(st)
        self.nstaged = n

    def run(self):
        """one pass over everything staged; returns (per-GPU kernel ms, wall ms)"""
        g, w = (C.c_float * self.n_gpus)(), C.c_float()
        st = self.L.zgpu_pool_run(self.h, g, C.byref(w))
        if st:
            raise ZgpuError(st)
        return list(g), w.value

    def timings(self, g=0):
        """per-kernel times (ms) of GPU g's last pass + (plaintext bytes, compressed bytes, blocks) of its resident submit"""
        a = (C.c_float * 10)()
        pb, cb, nb, nj = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        st = self.L.zgpu_pool_timings(self.h, g, a, 10, C.byref(pb), C.byref(cb), C.byref(nb), C.byref(nj))
        if st:
            raise ZgpuError(st)
        self.last_njobs = nj.value
        return dict(zip(["tables", "huf", "seq", "seqpost", "scan", "lit", "flat", "sweep", "lz", "total"], list(a))), pb.value, cb.value, nb.value

    def plan_stats(self, g=0):
        """the LZ77 plan of GPU g's resident jobs after a run (zgpu_pool_plan_stats)"""
        a = (C.c_uint64 * 7)()
        st = self.L.zgpu_pool_plan_stats(self.h, g, a, 7)
        if st:
            raise ZgpuError(st)
        return dict(zip(["units", "direct_units", "noseq_units", "pointer_units", "sweep_steps", "pointer_bytes", "direct_bytes"], [int(x) for x in a]))

    def frame(self, i):
        gpu, size, st = C.c_int(), C.c_uint64(), C.c_uint32()
        r = self.L.zgpu_pool_frame(self.h, i, C.byref(gpu), C.byref(size), C.byref(st))
        if r:
            raise ZgpuError(r)
        return gpu.value, size.value, st.value

    def read(self, i, cap):
        import numpy as np
        arr = np.empty(max(cap, 1), dtype=np.uint8)
        w = C.c_size_t()
        st = self.L.zgpu_pool_read(self.h, i, arr.ctypes.data_as(C.c_void_p), cap, C.byref(w))
        if st:
            raise ZgpuError(st)
        return arr[:w.value].tobytes()


class CStreamingDecoder:
    """zgpu_streaming (the C-ABI mirror of StreamingDecoder, streaming_decoder.rs:40-156) over a Python file-like source, or over
    bytes / a numpy array (source=None, data=...: zgpu_streaming_create_slice, nothing is copied on the host).
    read_ahead: bytes decoded ahead of the reader at most (ring size); NO_READ_AHEAD: the reference's block-by-block schedule."""

    def __init__(self, ctx, source=None, data=None, read_ahead=0, checksum=True, pipe_after=0, first_run_blocks=0, copy_threads=0):
        self.L, self.source, self.ctx = ctx.L, source, ctx      # (the context must outlive the stream)
        o = StreamOpts(read_ahead, 0 if checksum else 1, copy_threads, pipe_after, first_run_blocks, 0)
        h = C.c_void_p()
        if source is not None:
            def rd(_user, dst, n):
                b = source.read(n)
                C.memmove(dst, b, len(b))
                return len(b)
            self._cb = READ_FN(rd)
            st = self.L.zgpu_streaming_create_ex(ctx.h, self._cb, None, C.byref(o), C.byref(h))
        else:
            self._keep = data
            if isinstance(data, (bytes, bytearray)):
                self._buf = (C.c_char * len(data)).from_buffer_copy(data) if isinstance(data, bytes) else (C.c_char * len(data)).from_buffer(data)
                ptr, n = C.addressof(self._buf), len(data)
            elif isinstance(data, tuple):                  # (address, length): e.g. pinned memory of a torch tensor
                ptr, n = data
            else:                                          # numpy array
                ptr, n = data.ctypes.data, data.nbytes
            st = self.L.zgpu_streaming_create_slice(ctx.h, ptr, n, C.byref(o), C.byref(h))
        if st:
            raise ZgpuError(st)
        self.h = h

    def read(self, n):
        buf = C.create_string_buffer(max(n, 1))
        got = C.c_size_t()
        st = self.L.zgpu_streaming_read(self.h, buf, n, C.byref(got))
        if st:
            raise ZgpuError(st)
        return buf.raw[:got.value]

    def read_into(self, addr, n):
        """read(&mut buf[..n]) into memory the caller owns; returns the byte count"""
        got = C.c_size_t()
        st = self.L.zgpu_streaming_read(self.h, addr, n, C.byref(got))
        if st:
            raise ZgpuError(st)
        return got.value

    def copy_to_sink(self, buf_size):
        """std::io::copy(&mut decoder, &mut io::sink()) with a buffer of buf_size bytes; returns the bytes copied"""
        total = C.c_uint64()
        st = self.L.zgpu_streaming_copy(self.h, buf_size, None, None, C.byref(total))
        if st:
            raise ZgpuError(st)
        return total.value

    def _dec(self):
        self.L.zgpu_streaming_decoder.restype = C.c_void_p
        self.L.zgpu_streaming_decoder.argtypes = [C.c_void_p]
        return self.L.zgpu_streaming_decoder(self.h)

    def device_bytes(self):
        """device memory the frame holds right now (zgpu_decoder_device_bytes of the decoder behind the stream)"""
        return self.L.zgpu_decoder_device_bytes(self._dec())

    def is_finished(self):
        return bool(self.L.zgpu_decoder_is_finished(self._dec()))

    def get_calculated_checksum(self):
        return self.L.zgpu_decoder_calculated_checksum(self._dec())

    def get_checksum_from_data(self):
        v = C.c_uint32()
        return v.value if self.L.zgpu_decoder_checksum_from_data(self._dec(), C.byref(v)) else None

    def blocks_decoded(self):
        return self.L.zgpu_decoder_blocks_decoded(self._dec())

    def bytes_read_from_source(self):
        return self.L.zgpu_decoder_bytes_read_from_source(self._dec())

    def source_position(self):
        return self.L.zgpu_streaming_source_position(self.h)

    def stats(self):
        a = (C.c_uint64 * 24)()
        self.L.zgpu_streaming_stats(self.h, a, 24)
        return dict(zip(["mode", "runs", "dropped", "host_bytes", "us_worker_idle", "us_run", "us_land", "us_commit", "us_ring_full", "us_reader_wait",
                         "us_reader_copy", "us_pull", "us_prepare", "us_kernels", "k_tables", "k_huf", "k_seq", "k_seqpost", "k_scan", "k_lit", "k_flat",
                         "k_sweep", "k_lz", "k_total"], [int(x) for x in a]))

    def close(self):
        if getattr(self, "h", None):
            self.L.zgpu_streaming_destroy(self.h)
            self.h = None

    __del__ = close


class StreamingDecoder:
    """Mirror of ruzstd::decoding::StreamingDecoder (streaming_decoder.rs:40-156): an io-style reader over ONE frame.

    source: a binary file-like object positioned at the frame header. The decoder reads exactly the bytes the
    reference would: the frame header, then whole blocks as read() needs them."""

    def __init__(self, source, decoder=None, ctx=None, max_window_size=None):
        self.source = source
        self.decoder = decoder or FrameDecoder(ctx)
        if max_window_size is not None:
            self.decoder.set_max_window_size(max_window_size)
        # frame header: 4 magic + 1 descriptor tell how long the rest is (frame.rs:6-85)
        head = source.read(5)
        if len(head) == 5 and head[:4] == bytes([0x28, 0xB5, 0x2F, 0xFD]):
            desc = head[4]
            single = (desc >> 5) & 1
            extra = (0 if single else 1) + (0, 1, 2, 4)[desc & 3] + ((1 if single else 0), 2, 4, 8)[desc >> 6]
            head += source.read(extra)
        else:
            head += source.read(3)
        self._cs = len(head) >= 5 and bool((head[4] >> 2) & 1)
        st, used, _, _ = self.decoder.reset(head)
        if st:
            raise ZgpuError(st)
        assert used == len(head)

    def _read_blocks(self, n):
        """n whole blocks (or up to the last block) from the source, as one byte string"""
        out = []
        for _ in range(n):
            hdr = self.source.read(3)
            out.append(hdr)
            if len(hdr) < 3:
                break
            btype = (hdr[0] >> 1) & 3
            size = (hdr[0] >> 3) | (hdr[1] << 5) | (hdr[2] << 13)
            out.append(self.source.read(1 if btype == 1 else size))
            if hdr[0] & 1:
                if (self.decoder_checksum_flag):
                    out.append(self.source.read(4))
                break
        return b"".join(out)

    @property
    def decoder_checksum_flag(self):
        return self._cs

    def read(self, n=-1):
        """impl Read (streaming_decoder.rs:119-155)"""
        d = self.decoder
        if n is None or n < 0:
            chunks = []
            while True:
                c = self.read(1 << 20)
                if not c:
                    return b"".join(chunks)
                chunks.append(c)
        if d.is_finished() and d.can_collect() == 0:
            return b""
        while d.can_collect() < n and not d.is_finished():
            need = n - d.can_collect()
            m = max(1, (need + (128 << 10) - 1) // (128 << 10))      # UptoBytes(need) never stops before ceil(need / 128 KiB) blocks
            data = self._read_blocks(m)
            st, used, fin = d.decode_blocks(data, STRAT_UPTO_BLOCKS, m)
            if st:
                raise ZgpuError(st)
        return d.read(n)

    def into_frame_decoder(self):
        return self.decoder


class BlockFrame:
    """The thin boundary (include/zgpu.h: zgpu_frame_begin / zgpu_blocks_submit / zgpu_sync / zgpu_read): the caller parses the
    frame header and the 3-byte block headers itself (as ruzstd's FrameDecoder does before it calls decode_block_content,
    frame_decoder.rs:319-375) and submits block tables."""

    def __init__(self, ctx, window_size, content_size=0, dict_id=0):
        self.L = ctx.L
        self.h = C.c_void_p()
        st = self.L.zgpu_frame_begin(ctx.h, window_size, content_size, dict_id, C.byref(self.h))
        if st:
            self.h = None
            raise ZgpuError(st)

    def close(self):
        if self.h:
            self.L.zgpu_frame_end(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def submit(self, src, blocks):
        """blocks: [(src_off, src_len, type, last, raw_rle_size)] offsets into src"""
        arr = (Block * max(len(blocks), 1))()
        for i, (off, ln, ty, last, sz) in enumerate(blocks):
            arr[i].src_off, arr[i].src_len, arr[i].type, arr[i].last, arr[i].raw_rle_size = off, ln, ty, last, sz
        st = self.L.zgpu_blocks_submit(self.h, src, len(src), arr, len(blocks))
        if st:
            raise ZgpuError(st)

    def sync(self):
        """returns (first_bad_block or None, its status)"""
        bad, st = C.c_size_t(), C.c_int32()
        r = self.L.zgpu_sync(self.h, C.byref(bad), C.byref(st))
        if r:
            raise ZgpuError(r)
        return (None if bad.value == C.c_size_t(-1).value else bad.value), st.value

    def available(self, finished):
        return self.L.zgpu_available(self.h, 1 if finished else 0)

    def read(self, cap, finished):
        buf = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t()
        st = self.L.zgpu_read(self.h, buf, cap, 1 if finished else 0, C.byref(n))
        if st:
            raise ZgpuError(st)
        return buf.raw[:n.value]

    def device_output(self):
        p, n = C.c_void_p(), C.c_size_t()
        st = self.L.zgpu_device_output(self.h, C.byref(p), C.byref(n))
        if st:
            raise ZgpuError(st)
        return p.value, n.value

    def checksum(self):
        return self.L.zgpu_frame_checksum(self.h)

    def blocks_decoded(self):
        return self.L.zgpu_frame_blocks_decoded(self.h)
