// zg_capi.cpp — the extern "C" boundary declared in include/zgpu.h, plus the host-side mirror of the reference's
// FrameDecoder (ruzstd/src/decoding/frame_decoder.rs) built on the engine.
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "zg_capi_int.h"
#include "zg_dev.h"
#include "zg_stream.h"

using namespace zg;

struct zgpu_batch {
  zgpu_ctx* ctx = nullptr;
  Batch* b = nullptr;
};

extern "C" {

int zgpu_ctx_create(int device_id, zgpu_ctx** out) {
  if (!out) return ZGPU_E_BAD_ARG;
  Engine* e = nullptr;
  int st = Engine::create(device_id, &e);
  if (st) return st;
  zgpu_ctx* c = new (std::nothrow) zgpu_ctx();
  if (!c) { delete e; return ZGPU_E_NOMEM; }
  c->eng = e;
  *out = c;
  return ZGPU_OK;
}
void zgpu_ctx_destroy(zgpu_ctx* c) {
  if (!c) return;
  delete c->eng;
  delete c;
}
void zgpu_set_max_window_size(zgpu_ctx* c, uint64_t m) { c->eng->max_window = m < kMaxWindow ? m : kMaxWindow; }
uint64_t zgpu_max_window_size(const zgpu_ctx* c) { return c->eng->max_window; }
const char* zgpu_last_error(const zgpu_ctx* c) { return c->eng->last_error.c_str(); }

const char* zgpu_status_name(int s) {
  switch (s) {
    case ZGPU_OK: return "Ok";
    case ZGPU_E_SKIP_FRAME: return "SkipFrame";
    case ZGPU_E_BAD_MAGIC: return "BadMagicNumber";
    case ZGPU_E_HEADER_READ: return "FrameHeaderReadError";
    case ZGPU_E_WINDOW_TOO_BIG_SPEC: return "WindowTooBig";
    case ZGPU_E_WINDOW_TOO_SMALL: return "WindowTooSmall";
    case ZGPU_E_WINDOW_SIZE_TOO_BIG: return "WindowSizeTooBig";
    case ZGPU_E_DICT_NOT_PROVIDED: return "DictNotProvided";
    case ZGPU_E_NOT_INITIALIZED: return "NotYetInitialized";
    case ZGPU_E_FAILED_READ_BLOCK_HEADER: return "FailedToReadBlockHeader";
    case ZGPU_E_FAILED_READ_BLOCK_BODY: return "FailedToReadBlockBody";
    case ZGPU_E_FAILED_READ_CHECKSUM: return "FailedToReadChecksum";
    case ZGPU_E_TARGET_TOO_SMALL: return "TargetTooSmall";
    case ZGPU_E_FAILED_SKIP_FRAME: return "FailedToSkipFrame";
    case ZGPU_E_RESERVED_BLOCK: return "FoundReservedBlock";
    case ZGPU_E_BLOCK_SIZE_TOO_LARGE: return "BlockSizeTooLarge";
    case ZGPU_E_MALFORMED_SECTION_HEADER: return "MalformedSectionHeader";
    case ZGPU_E_LITERALS_HEADER: return "LiteralsSectionParseError";
    case ZGPU_E_SEQUENCES_HEADER: return "SequencesHeaderParseError";
    case ZGPU_E_LIT_UNINIT_HUF: return "UninitializedHuffmanTable";
    case ZGPU_E_LIT_MISSING_JUMP: return "MissingBytesForJumpHeader";
    case ZGPU_E_LIT_MISSING_BYTES: return "MissingBytesForLiterals";
    case ZGPU_E_LIT_EXTRA_PADDING: return "ExtraPadding(literals)";
    case ZGPU_E_LIT_BITSTREAM_MISMATCH: return "BitstreamReadMismatch";
    case ZGPU_E_LIT_COUNT_MISMATCH: return "DecodedLiteralCountMismatch";
    case ZGPU_E_HUF_TABLE: return "HuffmanTableError";
    case ZGPU_E_FSE_TABLE: return "FSETableError";
    case ZGPU_E_FSE_UNINIT: return "TableIsUninitialized";
    case ZGPU_E_SEQ_MISSING_MODE: return "MissingCompressionMode";
    case ZGPU_E_SEQ_RLE_BYTE: return "MissingByteForRleTable";
    case ZGPU_E_SEQ_EXTRA_PADDING: return "ExtraPadding(sequences)";
    case ZGPU_E_SEQ_UNSUPPORTED_OFFSET: return "UnsupportedOffset";
    case ZGPU_E_SEQ_NOT_ENOUGH_BYTES: return "NotEnoughBytesForNumSequences";
    case ZGPU_E_SEQ_EXTRA_BITS: return "ExtraBits";
    case ZGPU_E_EXE_NOT_ENOUGH_LITERALS: return "NotEnoughBytesForSequence";
    case ZGPU_E_EXE_ZERO_OFFSET: return "ZeroOffset";
    case ZGPU_E_EXE_OFFSET_TOO_BIG: return "OffsetTooBig";
    case ZGPU_E_EXE_DICT_TOO_SMALL: return "NotEnoughBytesInDictionary";
    case ZGPU_E_DICT_DECODE: return "DictionaryDecodeError";
    case ZGPU_E_UNSUPPORTED: return "Unsupported";
    case ZGPU_E_INTERNAL: return "Internal";
    case ZGPU_E_NOMEM: return "OutOfMemory";
    case ZGPU_E_HIP: return "HipError";
    case ZGPU_E_BAD_ARG: return "BadArgument";
    default: return "Unknown";
  }
}

// ---- staged batch ----------------------------------------------------------------------------------------
int zgpu_batch_prepare(zgpu_ctx* c, const uint8_t* src, size_t len, zgpu_batch** out) {
  if (!c || !out || (!src && len)) return ZGPU_E_BAD_ARG;
  Batch* b = nullptr;
  int st = c->eng->prepare(src, len, &b);
  if (st) return st;
  zgpu_batch* zb = new (std::nothrow) zgpu_batch();
  if (!zb) { delete b; return ZGPU_E_NOMEM; }
  zb->ctx = c;
  zb->b = b;
  *out = zb;
  return b->parse_status;
}
int zgpu_batch_run(zgpu_batch* zb) { return zb->b->run(); }

static void first_bad(const Batch* b, uint32_t* frame, uint32_t* status) {
  *frame = UINT32_MAX;
  *status = 0;
  for (uint32_t f = 0; f < b->frame_out.size(); f++)
    if (b->frame_out[f].status) { *frame = f; *status = b->frame_out[f].status; return; }
}
int zgpu_batch_sync(zgpu_batch* zb, uint64_t* total_out, uint32_t* bad_frame, uint32_t* its_status) {
  int st = zb->b->sync();
  if (st) return st;
  uint32_t bf, bs;
  first_bad(zb->b, &bf, &bs);
  if (total_out) *total_out = zb->b->total_out;
  if (bad_frame) *bad_frame = bf;
  if (its_status) *its_status = bs;
  return ZGPU_OK;
}
uint32_t zgpu_batch_num_frames(const zgpu_batch* zb) { return (uint32_t)zb->b->bb.frames.size(); }
uint32_t zgpu_batch_num_blocks(const zgpu_batch* zb) { return (uint32_t)zb->b->bb.blocks.size(); }
uint64_t zgpu_batch_compressed_size(const zgpu_batch* zb) { return zb->b->src_len; }
int zgpu_batch_frame_info(const zgpu_batch* zb, uint32_t f, zgpu_frame_info* o) {
  const Batch* b = zb->b;
  if (f >= b->info.size() || !o) return ZGPU_E_BAD_ARG;
  memset(o, 0, sizeof *o);
  const FrameInfo& fi = b->info[f];
  o->src_begin = fi.src_begin; o->src_end = fi.src_end; o->window_size = fi.window_size;
  o->frame_content_size = fi.header.frame_content_size; o->nblocks = fi.nblocks;
  o->has_checksum = fi.has_checksum; o->checksum = fi.checksum;
  o->status = fi.host_status;
  if (f < b->frame_out.size()) {
    o->out_base = b->frame_out[f].out_base; o->out_size = b->frame_out[f].out_size;
    if (b->frame_out[f].status) { o->status = b->frame_out[f].status; o->bad_block = b->frame_out[f].bad_block; }
    else if (fi.host_status) o->bad_block = fi.nblocks;   // the walk stopped behind the frame's blocks, all of which decoded: the block it could not read
  }
  return ZGPU_OK;
}
int zgpu_batch_read(zgpu_batch* zb, uint64_t off, uint8_t* dst, uint64_t n) { return zb->b->read_output(off, dst, n); }
const void* zgpu_batch_output_device(const zgpu_batch* zb) { return zb->b->device_output(); }
int zgpu_batch_timings(const zgpu_batch* zb, float* ms, int n) {
  int k = n < ZG_T_COUNT ? n : ZG_T_COUNT;
  for (int i = 0; i < k; i++) ms[i] = zb->b->ms[i];
  return k;
}
void zgpu_batch_destroy(zgpu_batch* zb) {
  if (!zb) return;
  delete zb->b;
  delete zb;
}

int zgpu_batch_block_info(zgpu_batch* zb, uint32_t i, zgpu_block_info* o) {
  Batch* b = zb->b;
  if (i >= b->bb.blocks.size() || !o) return ZGPU_E_BAD_ARG;
  const ZgBlock& k = b->bb.blocks[i];
  memset(o, 0, sizeof *o);
  o->btype = k.btype; o->lit_type = k.lit_type; o->nstreams = k.nstreams; o->seq_modes = k.seq_modes;
  o->regen_size = k.regen_size; o->nseq = k.nseq; o->frame = k.frame;
  o->huf_slot = k.huf_slot; o->ll_slot = k.ll_slot; o->of_slot = k.of_slot; o->ml_slot = k.ml_slot;
  std::vector<uint32_t> st;
  int r = b->read_block_status(&st);
  if (r) return r;
  o->status = k.host_status ? k.host_status : st[i];
  std::vector<ZgSeq> dummy;
  ZgBlockSeqOut so;
  ZgBlockPos pos;
  memset(&so, 0, sizeof so);
  // read only the per-block records (no sequences)
  ZgBlock saved = b->bb.blocks[i];
  b->bb.blocks[i].nseq = 0;
  r = b->read_sequences(i, &dummy, &so, &pos);
  b->bb.blocks[i] = saved;
  if (r) return r;
  if (k.btype == ZG_BT_COMPRESSED && k.nseq) { o->sum_ll = so.sum_ll; o->sum_ml = so.sum_ml; }
  o->hist_init[0] = pos.hist_init[0]; o->hist_init[1] = pos.hist_init[1]; o->hist_init[2] = pos.hist_init[2];
  o->active = pos.active; o->out_base = pos.out_base;
  return ZGPU_OK;
}
int zgpu_batch_block_literals(zgpu_batch* zb, uint32_t i, uint8_t* dst, size_t cap, size_t* n) {
  std::vector<uint8_t> v;
  int r = zb->b->read_literals(i, &v);
  if (r) return r;
  *n = v.size();
  if (v.size() > cap) return ZGPU_E_TARGET_TOO_SMALL;
  if (!v.empty()) memcpy(dst, v.data(), v.size());
  return ZGPU_OK;
}
int zgpu_batch_block_sequences(zgpu_batch* zb, uint32_t i, zgpu_seq* dst, size_t cap, size_t* n) {
  std::vector<ZgSeq> v;
  int r = zb->b->read_sequences(i, &v, nullptr, nullptr);
  if (r) return r;
  *n = v.size();
  if (v.size() > cap) return ZGPU_E_TARGET_TOO_SMALL;
  for (size_t k = 0; k < v.size(); k++) { dst[k].of = v[k].of; dst[k].ml = ZG_SEQ_ML(v[k]); dst[k].mdst = ZG_SEQ_MDST(v[k]); dst[k].lit_start = ZG_SEQ_LIT(v[k]); }
  return ZGPU_OK;
}
int zgpu_batch_debug_timers(zgpu_batch* zb, uint64_t out[1024]) { return zb->b->read_debug(out); }
int zgpu_batch_debug_scratch(zgpu_batch* zb, int what, uint64_t off, void* dst, uint64_t n) { return zb->b->read_scratch(what, off, dst, n); }
uint32_t zgpu_batch_num_units(const zgpu_batch* zb) { return (uint32_t)zb->b->bb.units.size(); }
uint32_t zgpu_batch_debug_sweep_mode(const zgpu_batch* zb) { return zb->b->sweep_mode; }
int zgpu_batch_unit(zgpu_batch* zb, uint32_t u, uint32_t* first_block, uint32_t* nblocks, uint64_t* scratch_base) {
  if (u >= zb->b->bb.units.size()) return ZGPU_E_BAD_ARG;
  const ZgUnit& x = zb->b->bb.units[u];
  *first_block = x.first_block; *nblocks = x.nblocks;
  return zb->b->unit_scratch_base(u, scratch_base);
}
int zgpu_debug_calibrate(zgpu_ctx* c, uint64_t bytes) {
  // profiler calibration: known traffic per access pattern — a 16 B/lane and a 4 B/lane copy of exactly `bytes` read + `bytes`
  // written, and bytes / 64 random 4-byte and 8-byte reads spread over `bytes` (every one a cache miss)
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return ZGPU_E_NOMEM;
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 2, bytes);
  zg_launch_calib(a, b, bytes, c->eng->stream());
  hipError_t e = hipStreamSynchronize(c->eng->stream());
  (void)hipFree(a); (void)hipFree(b);
  return e == hipSuccess ? ZGPU_OK : ZGPU_E_HIP;
}
int zgpu_debug_tuning(const zgpu_ctx* c, uint32_t* out, int n) {
  if (!c || !out || n <= 0) return 0;
  const Tuning& t = c->eng->tuning();
  const uint32_t v[8] = {t.dev_build ? 1u : 0u, t.unit_blocks, (uint32_t)t.seq_packed, (uint32_t)t.flat4, t.ramp_percent, t.sweep_w, (uint32_t)t.flat_shape, t.force_inorder ? 1u : 0u};
  const int k = n < 8 ? n : 8;
  for (int i = 0; i < k; i++) out[i] = v[i];
  return k;
}
int zgpu_batch_fse_slot(zgpu_batch* zb, uint32_t slot, uint32_t* entries, uint8_t logs[4]) {
  std::vector<uint32_t> v;
  int r = zb->b->read_fse_slot(slot, &v, logs);
  if (r) return r;
  memcpy(entries, v.data(), v.size() * 4);
  return ZGPU_OK;
}
int zgpu_batch_huf_slot(zgpu_batch* zb, uint32_t slot, uint16_t* entries, int* max_bits) {
  std::vector<uint16_t> v;
  int r = zb->b->read_huf_slot(slot, &v, max_bits);
  if (r) return r;
  memcpy(entries, v.data(), v.size() * 2);
  return ZGPU_OK;
}

// ---- decode_all --------------------------------------------------------------------------------------------
static int decode_all_per_frame(zgpu_ctx* c, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* written);
int zgpu_decode_all(zgpu_ctx* c, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* written) {
  if (!c || !written || (!src && len) || (!dst && cap)) return ZGPU_E_BAD_ARG;
  *written = 0;
  zgpu_batch* zb = nullptr;
  int st = zgpu_batch_prepare(c, src, len, &zb);
  if (st == ZGPU_E_DICT_NOT_PROVIDED && !c->dicts.empty()) {
    // frames that need a dictionary go through the FrameDecoder mirror one by one, like the reference's own loop (:541-577)
    zgpu_batch_destroy(zb);
    return decode_all_per_frame(c, src, len, dst, cap, written);
  }
  // The walk stopped at a header it could not read (a truncated block, a reserved block type, the next frame's magic ...): the
  // reference meets that error only after it has decoded everything in front of it (decode_blocks reads and decodes block by
  // block, frame_decoder.rs:319-375), so an error INSIDE an earlier block comes first. The blocks in front of the stop are in the
  // batch: they are decoded, and the walk's error is the answer only if none of them fails.
  int walk = 0;
  if (st) {
    if (!zb || zb->b->bb.blocks.empty()) { zgpu_batch_destroy(zb); return st; }
    walk = st;
  }
  zb->b->drain_rule = ZG_DRAIN_DECODE_ALL;          // decode_all drains its DecodeBuffer every MiB (frame_decoder.rs:560-563): zg_exact.h
  uint64_t total = 0;
  uint32_t bf = 0, bs = 0;
  if ((st = zgpu_batch_run(zb)) || (st = zgpu_batch_sync(zb, &total, &bf, &bs))) { zgpu_batch_destroy(zb); return st; }
  if (bs) { zgpu_batch_destroy(zb); return (int)bs; }
  if (walk) { zgpu_batch_destroy(zb); return walk; }
  if (total > cap) { zgpu_batch_destroy(zb); return ZGPU_E_TARGET_TOO_SMALL; }  // frame_decoder.rs:567-569
  st = zgpu_batch_read(zb, 0, dst, total);
  zgpu_batch_destroy(zb);
  if (st) return st;
  *written = (size_t)total;
  return ZGPU_OK;
}

// FrameDecoder::decode_all_to_vec (frame_decoder.rs:591-610): the output buffer is sized by the decoder, here exactly (the
// size of every frame is known on the host before the LZ77 stages run). *out is malloc'ed; release it with zgpu_free.
int zgpu_decode_all_alloc(zgpu_ctx* c, const uint8_t* src, size_t len, uint8_t** out, size_t* written) {
  if (!c || !out || !written || (!src && len)) return ZGPU_E_BAD_ARG;
  *out = nullptr; *written = 0;
  zgpu_batch* zb = nullptr;
  int st = zgpu_batch_prepare(c, src, len, &zb);
  if (st == ZGPU_E_DICT_NOT_PROVIDED && !c->dicts.empty()) {
    // dictionary frames go one by one through the FrameDecoder mirror: grow like Vec does
    zgpu_batch_destroy(zb);
    size_t cap = len * 4 + (1u << 20);
    for (;;) {
      uint8_t* buf = (uint8_t*)malloc(cap);
      if (!buf) return ZGPU_E_NOMEM;
      st = decode_all_per_frame(c, src, len, buf, cap, written);
      if (st == ZGPU_E_TARGET_TOO_SMALL && cap < ((size_t)1 << 40)) { free(buf); cap *= 2; continue; }
      if (st) { free(buf); return st; }
      *out = buf;
      return ZGPU_OK;
    }
  }
  int walk = 0;                                      // (as in zgpu_decode_all: what lies in front of the point where the walk stopped is decoded first)
  if (st) {
    if (!zb || zb->b->bb.blocks.empty()) { zgpu_batch_destroy(zb); return st; }
    walk = st;
  }
  zb->b->drain_rule = ZG_DRAIN_DECODE_ALL;          // decode_all_to_vec runs the same loop as decode_all (:580-591)
  uint64_t total = 0;
  uint32_t bf = 0, bs = 0;
  if ((st = zgpu_batch_run(zb)) || (st = zgpu_batch_sync(zb, &total, &bf, &bs))) { zgpu_batch_destroy(zb); return st; }
  if (bs) { zgpu_batch_destroy(zb); return (int)bs; }
  if (walk) { zgpu_batch_destroy(zb); return walk; }
  uint8_t* buf = (uint8_t*)malloc(total ? total : 1);
  if (!buf) { zgpu_batch_destroy(zb); return ZGPU_E_NOMEM; }
  st = zgpu_batch_read(zb, 0, buf, total);
  zgpu_batch_destroy(zb);
  if (st) { free(buf); return st; }
  *out = buf; *written = (size_t)total;
  return ZGPU_OK;
}
void zgpu_free(void* p) { free(p); }

}  // extern "C"

// ---- dictionaries (decoding/dictionary.rs:45-126) -------------------------------------------------------------------------
// Parsed on the host with the same lane routines the kernels use (zg_dev.h is host-callable); the tables travel to the
// device as the frame's carried tables (scratch.rs:70-78 init_from_dict).
static int parse_dict(const uint8_t* raw, size_t len, ZgDict* d) {
  static const uint8_t kMagic[4] = {0x37, 0xA4, 0x30, 0xEC};   // dictionary.rs:39
  if (len < 8) return ZG_DICT_DECODE;
  if (memcmp(raw, kMagic, 4)) return ZG_DICT_DECODE;
  memcpy(&d->id, raw + 4, 4);
  d->fse.assign(ZG_FSE_SLOT_U32, 0);
  d->huf.assign(ZG_HUF_SLOT_U16, 0);
  const uint8_t* t = raw + 8;
  size_t tl = len - 8;
  int16_t probs[256];
  uint16_t counter[256];
  uint8_t weights[264];
  uint32_t fsew[64], used = 0;
  int nw = 0, mb = 0;
  // the routines may look 8 bytes past a position: work on a padded copy
  std::vector<uint8_t> pad(t, t + tl);
  pad.resize(tl + 64, 0);
  t = pad.data();
  if (zg_huf_read_weights(t, (uint32_t)tl, weights, &nw, &used, fsew, probs, counter)) return ZG_DICT_DECODE;
  if (zg_huf_build(weights, nw, d->huf.data(), &mb)) return ZG_DICT_DECODE;
  d->huf_maxbits = (uint8_t)mb;
  if (tl < used) return ZG_DICT_DECODE;
  t += used; tl -= used;
  // order OF, ML, LL (dictionary.rs:74-97); logs are kept in the order LL, OF, ML like the arena
  const int kinds[3] = {ZG_KIND_OF, ZG_KIND_ML, ZG_KIND_LL}, maxlog[3] = {8, 9, 9}, maxsym[3] = {31, 52, 35}, logidx[3] = {1, 2, 0};
  const uint32_t offs[3] = {ZG_FSE_OF_OFF, ZG_FSE_ML_OFF, ZG_FSE_LL_OFF};
  for (int k = 0; k < 3; k++) {
    int np = 0, al = 0;
    if (zg_fse_read_probs(t, (uint32_t)tl, maxlog[k], maxsym[k], probs, &np, &al, &used)) return ZG_DICT_DECODE;
    if (zg_fse_build(probs, np, al, kinds[k], d->fse.data() + offs[k], counter)) return ZG_DICT_DECODE;
    d->logs[logidx[k]] = (uint8_t)al;
    if (tl < used) return ZG_DICT_DECODE;
    t += used; tl -= used;
  }
  d->logs[3] = 0;
  if (tl < 12) return ZG_DICT_DECODE;
  memcpy(d->hist, t, 12);
  d->content.assign(t + 12, t + tl);
  return ZG_OK;
}

extern "C" int zgpu_add_dict(zgpu_ctx* c, const uint8_t* raw, size_t len, uint32_t* id_out) {
  if (!c || (!raw && len)) return ZGPU_E_BAD_ARG;
  ZgDict d;
  int st = parse_dict(raw, len, &d);
  if (st) return st;
  if (id_out) *id_out = d.id;
  c->dicts[d.id] = std::move(d);   // BTreeMap::insert: a dictionary with the same id is replaced (frame_decoder.rs:224-227)
  return ZGPU_OK;
}

// ---- FrameDecoder mirror (frame_decoder.rs:80-627): struct zgpu_decoder is in zg_capi_int.h ------------------------------------
size_t zg_dec_drain(zgpu_decoder* d, size_t n, uint8_t* dst) {  // DecodeBuffer::drain_to decode_buffer.rs:256-314
  if (!n) return 0;
  if (dst) memcpy(dst, d->buf.data() + d->head, n);
  if (d->hash_on) d->hash.update(d->buf.data() + d->head, n);
  d->head += n;
  if (d->head == d->buf.size()) { d->buf.clear(); d->head = 0; }
  else if (d->head > (1u << 22) && d->head > d->held()) { d->buf.erase(d->buf.begin(), d->buf.begin() + d->head); d->head = 0; }
  return n;
}

int zg_apply_dict(zgpu_decoder* d, const ZgDict& dict) {   // DecoderScratch::init_from_dict scratch.rs:70-78
  // Tables, offset history and dictionary content are replaced, whenever it is called (force_dict may come after blocks
  // were decoded, frame_decoder.rs:229-243): the device window is rebuilt as [new dictionary content][frame bytes so far].
  FrameState& fs = d->fs;
  int st;
  if ((st = fs.d_fse.reserve(ZG_FSE_SLOT_U32 * 4)) || (st = fs.d_huf.reserve(ZG_HUF_SLOT_U16 * 2))) return st;
  const uint64_t keep = fs.have;
  DevBuf nb;
  if ((st = nb.reserve(kOutFront + dict.content.size() + keep + 256))) return st;
  uint8_t* np = (uint8_t*)nb.p + kOutFront;
  if (hipMemcpy(fs.d_fse.p, dict.fse.data(), ZG_FSE_SLOT_U32 * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(fs.d_huf.p, dict.huf.data(), ZG_HUF_SLOT_U16 * 2, hipMemcpyHostToDevice) != hipSuccess ||
      (dict.content.size() && hipMemcpy(np, dict.content.data(), dict.content.size(), hipMemcpyHostToDevice) != hipSuccess) ||
      (keep && hipMemcpy(np + dict.content.size(), fs.out_ptr() + fs.base, keep, hipMemcpyDeviceToDevice) != hipSuccess)) {
    nb.release();
    return ZGPU_E_HIP;
  }
  fs.d_out.release();
  fs.d_out = nb;
  memcpy(fs.logs, dict.logs, 4);
  fs.huf_maxbits = dict.huf_maxbits;
  fs.carry_mask = 0xF;
  memcpy(fs.hist, dict.hist, 12);
  fs.base = dict.content.size();
  d->using_dict = dict.id;
  return ZGPU_OK;
}

// decode up to max_blocks blocks (0 = to the end of the frame) of the current frame from src
static int decode_run(zgpu_decoder* d, const uint8_t* src, size_t len, uint32_t max_blocks, size_t* consumed, bool* saw_last = nullptr) {
  Engine* eng = d->ctx->eng;
  if (saw_last) *saw_last = false;
  Batch* b = nullptr;
  size_t used = 0;
  *consumed = 0;
  int st = eng->prepare_run(src, len, &d->fs, d->fh.content_checksum(), max_blocks, d->held(), &b, &used);
  if (st) return st;
  const int parse_status = b->parse_status;
  const size_t nb = b->bb.blocks.size();
  if (nb == 0) {
    if (parse_status == ZG_FAILED_READ_BLOCK_BODY) d->bytes_read += 3;   // (the header of the block whose body is not there was read and counted, :325-328)
    delete b;
    return parse_status ? parse_status : ZGPU_E_INTERNAL;
  }
  b->drain_rule = d->drain_rule;
  if ((st = b->run()) || (st = b->sync())) { delete b; return st; }
  if (b->frame_out.empty()) { delete b; return ZGPU_E_INTERNAL; }
  const ZgFrameOut fo = b->frame_out[0];   // (a failed frame: out_size ends with its last good block, Batch::sync)
  if ((st = b->commit(&d->fs))) { delete b; return st; }
  // bring the new bytes to the host buffer the collect/read calls drain
  const size_t old = d->buf.size();
  d->buf.resize(old + fo.out_size);
  if (fo.out_size && hipMemcpy(d->buf.data() + old, (const uint8_t*)d->fs.out_ptr() + fo.out_base, fo.out_size, hipMemcpyDeviceToHost) != hipSuccess) {
    delete b;
    return ZGPU_E_HIP;
  }
  // counters (frame_decoder.rs:328,341,343): blocks that decoded count; a failing block does not
  const uint32_t good = fo.status ? fo.good_blocks : (uint32_t)nb - (b->bb.blocks.back().host_status ? 1u : 0u);
  size_t bytes = 0;
  for (uint32_t i = 0; i < good && i < nb; i++) bytes += 3 + b->bb.blocks[i].src_len;
  d->block_counter += good;
  int result = fo.status ? (int)fo.status : parse_status;
  // the last block decoded: the frame counts as finished before the checksum is read (frame_decoder.rs:347-357) — read() hands out
  // everything from then on, also when the four bytes never come
  if (!fo.status && b->saw_last_block && (parse_status == 0 || parse_status == ZG_FAILED_READ_CHECKSUM)) {
    d->frame_finished = true;
    if (saw_last) *saw_last = true;
    if (!result && !b->info.empty() && b->info[0].has_checksum) { d->has_checksum = true; d->checksum = b->info[0].checksum; bytes += 4; }
  }
  // a block whose body fails had its header read and counted (:325-341): everything but the three header-level errors
  if (result && result != ZG_FAILED_READ_BLOCK_HEADER && result != ZG_RESERVED_BLOCK && result != ZG_BLOCK_SIZE_TOO_LARGE && result != ZG_FAILED_READ_CHECKSUM)
    bytes += 3;
  d->bytes_read += bytes;
  *consumed = bytes;
  delete b;
  return result;
}

extern "C" {

int zgpu_decoder_create(zgpu_ctx* c, zgpu_decoder** out) {
  if (!c || !out) return ZGPU_E_BAD_ARG;
  zgpu_decoder* d = new (std::nothrow) zgpu_decoder();
  if (!d) return ZGPU_E_NOMEM;
  d->ctx = c;
  *out = d;
  return ZGPU_OK;
}
void zgpu_decoder_destroy(zgpu_decoder* d) {
  if (!d) return;
  d->fs.release();
  delete d;
}

int zgpu_decoder_init(zgpu_decoder* d, const uint8_t* src, size_t len, size_t* consumed, uint32_t* skip_magic, uint32_t* skip_len) {
  // FrameDecoder::reset (frame_decoder.rs:200-221), FrameDecoderState::new/reset (:103-134)
  FrameHeader h;
  size_t c = 0;
  if (consumed) *consumed = 0;
  int st = read_frame_header(src, len, &h, &c, skip_magic, skip_len);
  if (st) { if (st == ZG_SKIP_FRAME && consumed) *consumed = c; return st; }
  uint64_t w;
  if ((st = frame_window_size(h, &w))) return st;
  if (w > d->ctx->eng->max_window) return ZGPU_E_WINDOW_SIZE_TOO_BIG;
  d->has_state = true; d->fh = h; d->window_size = w; d->frame_finished = false; d->block_counter = 0;
  d->bytes_read = c; d->has_checksum = false; d->checksum = 0; d->using_dict = 0;
  d->buf.clear(); d->head = 0; d->hash.reset(0);
  d->fs.reset();
  d->fs.window_size = w;
  if (consumed) *consumed = c;
  if (h.has_dict_id) {   // :212-219
    auto it = d->ctx->dicts.find(h.dict_id);
    if (it == d->ctx->dicts.end()) return ZGPU_E_DICT_NOT_PROVIDED;
    return zg_apply_dict(d, it->second);
  }
  return ZGPU_OK;
}

int zgpu_decoder_force_dict(zgpu_decoder* d, uint32_t dict_id) {   // frame_decoder.rs:229-243
  if (!d->has_state) return ZGPU_E_NOT_INITIALIZED;
  if (d->stream && zg_stream_blocks_decoded(d->stream)) return ZGPU_E_BAD_ARG;   // (behind a stream: only before its first read)
  auto it = d->ctx->dicts.find(dict_id);
  if (it == d->ctx->dicts.end()) return ZGPU_E_DICT_NOT_PROVIDED;
  return zg_apply_dict(d, it->second);
}

int zgpu_decoder_decode_blocks(zgpu_decoder* d, const uint8_t* src, size_t len, size_t* consumed, int strat, size_t n, int* frame_finished) {
  // FrameDecoder::decode_blocks (frame_decoder.rs:309-377)
  if (consumed) *consumed = 0;
  if (frame_finished) *frame_finished = 0;
  if (!d->has_state) return ZGPU_E_NOT_INITIALIZED;
  if (d->stream) return ZGPU_E_BAD_ARG;            // the streaming decoder that owns this decoder feeds it (zgpu_streaming_decoder: accessors and read / collect only)
  size_t p = 0;
  int st = ZGPU_OK;
  if (strat == ZGPU_STRAT_ALL || strat == ZGPU_STRAT_UPTO_BLOCKS) {
    size_t used = 0;
    st = decode_run(d, src, len, strat == ZGPU_STRAT_ALL ? 0u : (uint32_t)(n ? n : 1), &used);   // UptoBlocks(0) still decodes one block
    p += used;
  } else {
    // UptoBytes(n): stop after the first block that brings the growth to n. A block regenerates at most 128 KiB, so
    // ceil(missing / 128 KiB) blocks can never overshoot that block; repeat until the growth is reached.
    if (d->read_ahead > n) n = (size_t)d->read_ahead;
    // (the loop ends with a last block of THIS call, frame_decoder.rs:347-359: a caller that goes on after the frame's end — the flag is
    //  already set then — has its bytes read as further blocks by the reference, as many as the strategy asks for: tools/dev/soak_api.py)
    const size_t before = d->buf.size();
    bool last_now = false;
    do {
      const size_t growth = d->buf.size() - before;
      const size_t missing = n > growth ? n - growth : 0;
      uint32_t m = (uint32_t)((missing + kMaxBlockSize - 1) / kMaxBlockSize);
      if (m == 0) m = 1;
      size_t used = 0;
      st = decode_run(d, src + p, len - p, m, &used, &last_now);
      p += used;
    } while (!st && !last_now && d->buf.size() - before < n);
  }
  if (consumed) *consumed = p;
  if (frame_finished) *frame_finished = d->frame_finished ? 1 : 0;
  return st;
}

int zgpu_decoder_is_finished(const zgpu_decoder* d) {
  if (!d->has_state) return 1;
  if (d->stream) return zg_stream_is_finished(d->stream) ? 1 : 0;
  if (d->fh.content_checksum()) return d->frame_finished && d->has_checksum;
  return d->frame_finished;
}
size_t zgpu_decoder_can_collect(const zgpu_decoder* d) {
  if (!d->has_state) return 0;
  if (d->stream) return zg_stream_can_collect(d->stream);
  if (zgpu_decoder_is_finished(d)) return d->held();
  return d->held() > d->window_size ? d->held() - (size_t)d->window_size : 0;  // decode_buffer.rs:182-188
}
size_t zgpu_decoder_collect(zgpu_decoder* d, uint8_t* dst, size_t cap) {
  size_t n = zgpu_decoder_can_collect(d);
  if (n > cap) n = cap;
  if (d->stream) return zg_stream_take(d->stream, dst, n);   // (what is collectable is buffered already: nothing is decoded or pulled from the source)
  return zg_dec_drain(d, n, dst);
}
size_t zgpu_decoder_read(zgpu_decoder* d, uint8_t* dst, size_t cap) {
  if (!d->has_state) return 0;
  if (d->stream) { size_t n = zgpu_decoder_can_collect(d); if (n > cap) n = cap; return zg_stream_take(d->stream, dst, n); }
  size_t n = d->frame_finished ? d->held() : (d->held() > d->window_size ? d->held() - (size_t)d->window_size : 0);
  if (n > cap) n = cap;
  return zg_dec_drain(d, n, dst);
}

int zgpu_decoder_decode_from_to(zgpu_decoder* d, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* read_out, size_t* written_out) {
  // FrameDecoder::decode_from_to (frame_decoder.rs:439-529): only whole blocks are consumed; the checksum may arrive later
  if (read_out) *read_out = 0;
  if (written_out) *written_out = 0;
  if (d->stream) return ZGPU_E_BAD_ARG;
  const uint64_t at_start = d->has_state ? d->bytes_read : 0;
  if (!zgpu_decoder_is_finished(d) || !d->has_state) {
    size_t p = 0;
    if (!d->has_state) {
      size_t c = 0;
      int st = zgpu_decoder_init(d, src, len, &c, nullptr, nullptr);
      if (st) return st;
      p = c;
    }
    if (d->fh.content_checksum() && d->frame_finished && !d->has_checksum) {
      // the checksum was the only thing missing after the previous call (:465-477)
      if (len - p >= 4) { memcpy(&d->checksum, src + p, 4); d->has_checksum = true; d->bytes_read += 4; }
      if (read_out) *read_out = 4;
      return ZGPU_OK;
    }
    // count the complete blocks the source holds (:479-492)
    size_t q = p;
    uint32_t nblocks = 0;
    bool last = false;
    int hdr_err = 0;
    while (!last && len - q >= 3) {
      BlockHeader bh;
      int st = read_block_header(src + q, &bh);
      if (st) { if (nblocks == 0) return st; hdr_err = st; break; }   // (the blocks in front are decoded first, then this is the answer: :484-486)
      if (len - q - 3 < bh.content_size) break;
      q += 3 + bh.content_size;
      nblocks++;
      last = bh.last;
    }
    if (nblocks) {
      // the run must not swallow a checksum that is only partly there: hand over exactly the blocks (+ checksum if whole)
      size_t avail = q - p;
      if (last && d->fh.content_checksum() && len - q >= 4) avail += 4;
      const bool cs_missing = last && d->fh.content_checksum() && len - q < 4;
      size_t used = 0;
      int st;
      if (cs_missing) {
        // decode the blocks now; the checksum is read by a later call: temporarily treat the frame as checksum-less
        const uint8_t saved = d->fh.descriptor;
        d->fh.descriptor &= (uint8_t)~0x04u;
        st = decode_run(d, src + p, avail, nblocks, &used);
        d->fh.descriptor = saved;
      } else st = decode_run(d, src + p, avail, nblocks, &used);
      if (st) return st;
    }
    if (hdr_err) return hdr_err;
  }
  const size_t w = zgpu_decoder_read(d, dst, cap);
  if (read_out) *read_out = (size_t)(d->bytes_read - at_start);
  if (written_out) *written_out = w;
  return ZGPU_OK;
}

// (behind a streaming decoder that reads ahead these two count what has been DECODED, which runs ahead of what the reader was given)
uint64_t zgpu_decoder_blocks_decoded(const zgpu_decoder* d) { return !d->has_state ? 0 : d->stream ? zg_stream_blocks_decoded(d->stream) : d->block_counter; }
uint64_t zgpu_decoder_bytes_read_from_source(const zgpu_decoder* d) { return !d->has_state ? 0 : d->stream ? zg_stream_bytes_read(d->stream) : d->bytes_read; }
uint64_t zgpu_decoder_content_size(const zgpu_decoder* d) { return d->has_state ? d->fh.frame_content_size : 0; }
int zgpu_decoder_checksum_from_data(const zgpu_decoder* d, uint32_t* out) {
  if (d->has_state && d->stream) return zg_stream_checksum_from_data(d->stream, out) ? 1 : 0;
  if (!d->has_state || !d->has_checksum) return 0;
  *out = d->checksum;
  return 1;
}
uint32_t zgpu_decoder_calculated_checksum(const zgpu_decoder* d) { return d->stream ? zg_stream_calculated_checksum(d->stream) : (uint32_t)d->hash.digest(); }
// ruzstd's `hash` cargo feature (default on): without it the decoder keeps no XXH64 of what it hands out (decode_buffer.rs:42,223-227)
void zgpu_decoder_set_hash(zgpu_decoder* d, int on) { if (d) d->hash_on = on != 0; }
// decode_blocks(UptoBytes(n)) decodes at least this many bytes per call: a submit costs as long as ONE block's sequence chain whatever it
// holds, so a caller that asks for 8 KiB at a time (one block per submit) gets the same bytes several hundred times faster by letting the
// decoder run ahead. What it observes is what UptoBytes(max(n, bytes)) would give in the reference. 0 (default): exactly n.
void zgpu_decoder_set_read_ahead(zgpu_decoder* d, uint64_t bytes) { if (d) d->read_ahead = bytes; }
// device memory the frame holds right now: window (dictionary + undrained and recent bytes) + carried tables. Stays bounded by
// the window size however long the frame is (FrameState::make_room).
uint64_t zgpu_decoder_device_bytes(const zgpu_decoder* d) { return d ? (uint64_t)(d->fs.d_out.cap + d->fs.d_fse.cap + d->fs.d_huf.cap + d->fs.d_tmp.cap) : 0; }

}  // extern "C"

// ---- the thin boundary: the caller keeps the frame / block header parse (ruzstd: frame.rs:6-85, block_decoder.rs:201-247) and
// hands over host-parsed block tables; this is the seam BlockDecoder::decode_block_content (block_decoder.rs:39-95) sits at,
// called from FrameDecoder::decode_blocks (frame_decoder.rs:319-375). Built on the same FrameState as the FrameDecoder mirror.
struct zgpu_frame {
  zgpu_decoder dec;              // device window + carried tables + the host buffer read() drains (no frame header of its own)
  Batch* pending = nullptr;      // the submit in flight
  uint64_t submitted = 0;        // blocks handed over since zgpu_frame_begin
  uint64_t pending_first = 0;    // frame-relative index of the pending submit's first block
  uint64_t content_size = 0;
  bool saw_last = false;
  int sticky = 0;                // first error of the frame
  uint64_t bad_block = ~0ull;
};

static int frame_finish_pending(zgpu_frame* f) {
  Batch* b = f->pending;
  if (!b) return ZGPU_OK;
  f->pending = nullptr;
  zgpu_decoder* d = &f->dec;
  int st = b->sync();
  if (!st && b->frame_out.empty()) st = ZGPU_E_INTERNAL;
  const ZgFrameOut fo = st ? ZgFrameOut() : b->frame_out[0];   // (a failed frame: out_size ends with its last good block, Batch::sync)
  if (!st) st = b->commit(&d->fs);
  if (st) {   // the engine failed: the frame cannot go on (a later submit must not continue with these blocks missing)
    if (!f->sticky) { f->sticky = st; f->bad_block = f->pending_first; }
    delete b;
    return st;
  }
  const size_t old = d->buf.size();
  d->buf.resize(old + fo.out_size);
  if (fo.out_size && hipMemcpy(d->buf.data() + old, (const uint8_t*)d->fs.out_ptr() + fo.out_base, fo.out_size, hipMemcpyDeviceToHost) != hipSuccess) {
    delete b;
    return ZGPU_E_HIP;
  }
  const size_t nb = b->bb.blocks.size();
  const uint32_t good = fo.status ? fo.good_blocks : (uint32_t)nb - ((nb && b->bb.blocks.back().host_status) ? 1u : 0u);
  d->block_counter += good;
  const int result = fo.status ? (int)fo.status : b->parse_status;
  if (result && !f->sticky) { f->sticky = result; f->bad_block = f->pending_first + good; }
  if (!result && b->saw_last_block) { d->frame_finished = true; f->saw_last = true; }
  delete b;
  return ZGPU_OK;
}

extern "C" {

int zgpu_frame_begin(zgpu_ctx* c, uint64_t window_size, uint64_t content_size_or_0, uint32_t dict_id_or_0, zgpu_frame** out) {
  // FrameDecoderState::new / reset (frame_decoder.rs:103-134) with the header fields the caller parsed; DecoderScratch::reset
  // (scratch.rs:48-68), then init_from_dict (:70-78) when the header names a dictionary (frame_decoder.rs:212-219)
  if (!c || !out) return ZGPU_E_BAD_ARG;
  if (window_size > c->eng->max_window) return ZGPU_E_WINDOW_SIZE_TOO_BIG;           // frame_decoder.rs:137-145
  zgpu_frame* f = new (std::nothrow) zgpu_frame();
  if (!f) return ZGPU_E_NOMEM;
  zgpu_decoder* d = &f->dec;
  d->ctx = c; d->has_state = true; d->window_size = window_size; d->hash.reset(0);
  d->fs.reset();
  d->fs.window_size = window_size;
  f->content_size = content_size_or_0;
  if (dict_id_or_0) {
    auto it = c->dicts.find(dict_id_or_0);
    if (it == c->dicts.end()) { delete f; return ZGPU_E_DICT_NOT_PROVIDED; }
    const int st = zg_apply_dict(d, it->second);
    if (st) { d->fs.release(); delete f; return st; }
  }
  *out = f;
  return ZGPU_OK;
}

void zgpu_frame_end(zgpu_frame* f) {
  if (!f) return;
  delete f->pending;
  f->dec.fs.release();
  delete f;
}

int zgpu_blocks_submit(zgpu_frame* f, const uint8_t* src, size_t src_len, const zgpu_block* blocks, size_t nblocks) {
  // decode_block_content x nblocks (block_decoder.rs:39-95), as one batched submit. src is copied to the device before
  // this returns; the kernels run on the context's streams (zgpu_sync waits for them).
  if (!f || (!src && src_len) || (!blocks && nblocks)) return ZGPU_E_BAD_ARG;
  int st = frame_finish_pending(f);            // blocks of one frame depend on each other: one submit in flight
  if (st) return st;
  if (f->sticky || nblocks == 0) return ZGPU_OK;   // the frame failed already: zgpu_sync reports where
  if (f->saw_last) return ZGPU_E_BAD_ARG;          // blocks behind the frame's last block
  zgpu_decoder* d = &f->dec;
  std::vector<Engine::HostBlock> hb(nblocks);
  for (size_t i = 0; i < nblocks; i++) {
    hb[i].src_off = blocks[i].src_off; hb[i].src_len = blocks[i].src_len; hb[i].raw_rle_size = blocks[i].raw_rle_size;
    hb[i].type = blocks[i].type; hb[i].last = blocks[i].last;
  }
  Batch* b = nullptr;
  st = d->ctx->eng->prepare_blocks(src, src_len, hb.data(), nblocks, &d->fs, d->held(), &b);
  if (st) return st;
  if (b->bb.blocks.empty()) {
    // the first block of the submit failed the host's checks (reserved type, size beyond 128 KiB, body out of range): the same
    // sticky verdict a later block of a submit gets, reported by zgpu_sync
    const int ps = b->parse_status;
    delete b;
    if (!ps) return ZGPU_E_INTERNAL;
    f->sticky = ps; f->bad_block = f->submitted;
    return ZGPU_OK;
  }
  if ((st = b->run())) { delete b; return st; }
  f->pending = b;
  f->pending_first = f->submitted;
  f->submitted += b->bb.blocks.size();
  return ZGPU_OK;
}

int zgpu_sync(zgpu_frame* f, size_t* first_bad_block, int32_t* its_status) {
  if (!f) return ZGPU_E_BAD_ARG;
  const int st = frame_finish_pending(f);
  if (first_bad_block) *first_bad_block = f->sticky ? (size_t)f->bad_block : (size_t)-1;
  if (its_status) *its_status = f->sticky;
  return st;
}

size_t zgpu_available(const zgpu_frame* f, int frame_finished) {
  // DecodeBuffer::can_drain_to_window_size / can_drain (decode_buffer.rs:182-219): while the frame is unfinished the last
  // window_size bytes stay back (a later match may still copy from them)
  if (!f || f->pending) return 0;
  const zgpu_decoder* d = &f->dec;
  if (frame_finished) return d->held();
  return d->held() > d->window_size ? d->held() - (size_t)d->window_size : 0;
}

int zgpu_read(zgpu_frame* f, uint8_t* dst, size_t cap, int frame_finished, size_t* n) {
  if (!f || !n || (!dst && cap)) return ZGPU_E_BAD_ARG;
  const int st = frame_finish_pending(f);
  if (st) return st;
  size_t k = zgpu_available(f, frame_finished);
  if (k > cap) k = cap;
  *n = zg_dec_drain(&f->dec, k, dst);
  return ZGPU_OK;
}

int zgpu_device_output(zgpu_frame* f, const void** dptr, size_t* len) {
  if (!f || !dptr || !len) return ZGPU_E_BAD_ARG;
  const int st = frame_finish_pending(f);
  if (st) return st;
  const FrameState& fs = f->dec.fs;
  *dptr = fs.d_out.p ? (const void*)(fs.out_ptr() + fs.base) : nullptr;   // the most recent fs.have bytes of the frame, in HBM
  *len = (size_t)fs.have;
  return ZGPU_OK;
}

uint32_t zgpu_frame_checksum(const zgpu_frame* f) { return f ? (uint32_t)f->dec.hash.digest() : 0u; }   // XXH64 (seed 0) of the bytes read so far, low 32 bits
uint64_t zgpu_frame_blocks_decoded(const zgpu_frame* f) { return f ? f->dec.block_counter : 0; }

}  // extern "C"

// ---- collect_to_writer (frame_decoder.rs:395-407); the StreamingDecoder mirror is in zg_stream.cpp --------------------------------
extern "C" {

int zgpu_decoder_collect_to_writer(zgpu_decoder* d, zgpu_write_fn write, void* user, size_t* written) {
  // drains what collect() would return into the writer, in chunks; a short write ends the call (io::Write::write semantics)
  if (!d || !write) return ZGPU_E_BAD_ARG;
  size_t n = zgpu_decoder_can_collect(d), done = 0;
  while (done < n) {
    const size_t chunk = n - done < (1u << 20) ? n - done : (1u << 20);
    const size_t w = write(user, d->buf.data() + d->head, chunk);
    if (w > chunk) return ZGPU_E_BAD_ARG;
    zg_dec_drain(d, w, nullptr);
    done += w;
    if (w < chunk) break;
  }
  if (written) *written = done;
  return ZGPU_OK;
}

}  // extern "C"

static int decode_all_per_frame(zgpu_ctx* c, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* written) {
  // FrameDecoder::decode_all (frame_decoder.rs:541-577) frame by frame
  zgpu_decoder* d = nullptr;
  int st = zgpu_decoder_create(c, &d);
  if (st) return st;
  d->drain_rule = ZG_DRAIN_DECODE_ALL;   // one decode_blocks(All) per frame stands for the reference's rounds of UptoBytes(1 MiB) + read(): same verdicts
  size_t p = 0, total = 0;
  while (p < len) {
    size_t used = 0;
    uint32_t sm = 0, sl = 0;
    st = zgpu_decoder_init(d, src + p, len - p, &used, &sm, &sl);
    if (st == ZGPU_E_SKIP_FRAME) {
      p += used;
      if ((size_t)sl > len - p) { st = ZGPU_E_FAILED_SKIP_FRAME; break; }
      p += sl;
      st = ZGPU_OK;
      continue;
    }
    if (st) break;
    const size_t hdr = used, frame_at = p;
    p += used;
    int fin = 0;
    st = zgpu_decoder_decode_blocks(d, src + p, len - p, &used, ZGPU_STRAT_ALL, 0, &fin);
    if (st == ZGPU_E_UNSUPPORTED) {
      // The one thing a single submit cannot serve (zg_exact.h: a match that starts in the dictionary behind bytes the reference has
      // drained INSIDE this call): the frame again, on the reference's own schedule — rounds of UptoBytes(1 MiB), each followed by read()
      // (frame_decoder.rs:560-573) — so that the drains fall between submits, where the device window is laid out like the reference's
      // buffer (FrameState::make_room). Slow (a submit per MiB) and only ever taken by frames no encoder makes.
      st = zgpu_decoder_init(d, src + frame_at, len - frame_at, &used, &sm, &sl);
      if (st) break;
      d->drain_rule = ZG_DRAIN_NONE;
      p = frame_at + hdr;
      for (;;) {
        st = zgpu_decoder_decode_blocks(d, src + p, len - p, &used, ZGPU_STRAT_UPTO_BYTES, (size_t)1 << 20, &fin);
        p += used;
        if (st) break;
        total += zgpu_decoder_read(d, dst + total, cap - total);
        if (zgpu_decoder_can_collect(d) != 0) { st = ZGPU_E_TARGET_TOO_SMALL; break; }
        if (zgpu_decoder_is_finished(d)) break;
      }
      d->drain_rule = ZG_DRAIN_DECODE_ALL;
      if (st) break;
      continue;
    }
    p += used;
    if (st) break;
    total += zgpu_decoder_read(d, dst + total, cap - total);
    if (zgpu_decoder_can_collect(d) != 0) { st = ZGPU_E_TARGET_TOO_SMALL; break; }
  }
  zgpu_decoder_destroy(d);
  if (!st) *written = total;
  return st;
}
