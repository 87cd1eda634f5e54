// zg_capi.cpp — the extern "C" boundary declared in include/zgpu.h, plus the host-side mirror of the reference's
// FrameDecoder (ruzstd/src/decoding/frame_decoder.rs) built on the engine.
#include <string.h>
#include <new>
#include <vector>
#include "../../include/zgpu.h"
#include "zg_engine.h"
#include "zg_xxh64.h"

using namespace zg;

struct zgpu_ctx {
  Engine* eng = nullptr;
  std::string err;
};
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
  if (zb->b->overflow && !bs) return ZGPU_E_UNSUPPORTED;  // a block regenerated more than 128 KiB (non-conforming)
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
  static_assert(sizeof(zgpu_seq) == sizeof(ZgSeq), "layout");
  if (!v.empty()) memcpy(dst, v.data(), v.size() * sizeof(ZgSeq));
  return ZGPU_OK;
}
int zgpu_batch_debug_timers(zgpu_batch* zb, uint64_t out[8]) { return zb->b->read_debug(out); }
int zgpu_debug_calibrate(zgpu_ctx* c, uint64_t bytes) {
  // profiler calibration: one device-to-device copy kernel of exactly `bytes` read + `bytes` written
  void *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return ZGPU_E_NOMEM;
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 2, bytes);
  zg_launch_calib(a, b, bytes, c->eng->stream());
  zg_launch_calib(a, b, bytes, c->eng->stream());
  hipError_t e = hipStreamSynchronize(c->eng->stream());
  (void)hipFree(a); (void)hipFree(b);
  return e == hipSuccess ? ZGPU_OK : ZGPU_E_HIP;
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
int zgpu_decode_all(zgpu_ctx* c, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* written) {
  if (!c || !written || (!src && len) || (!dst && cap)) return ZGPU_E_BAD_ARG;
  *written = 0;
  zgpu_batch* zb = nullptr;
  int st = zgpu_batch_prepare(c, src, len, &zb);
  if (st) { zgpu_batch_destroy(zb); return st; }   // the reference returns the first error of the walk
  uint64_t total = 0;
  uint32_t bf = 0, bs = 0;
  if ((st = zgpu_batch_run(zb)) || (st = zgpu_batch_sync(zb, &total, &bf, &bs))) { zgpu_batch_destroy(zb); return st; }
  if (bs) { zgpu_batch_destroy(zb); return (int)bs; }
  if (total > cap) { zgpu_batch_destroy(zb); return ZGPU_E_TARGET_TOO_SMALL; }  // frame_decoder.rs:567-569
  st = zgpu_batch_read(zb, 0, dst, total);
  zgpu_batch_destroy(zb);
  if (st) return st;
  *written = (size_t)total;
  return ZGPU_OK;
}

}  // extern "C"

// ---- FrameDecoder mirror ------------------------------------------------------------------------------------
// Round-1 scope: init/reset + decode_blocks(All) + collect/read with the reference's window-retention rule and
// counters. The UptoBlocks/UptoBytes strategies need a persistent device window across submits.
struct zgpu_decoder {
  zgpu_ctx* ctx = nullptr;
  bool has_state = false;
  FrameHeader fh;
  uint64_t window_size = 0;
  bool frame_finished = false;
  uint64_t block_counter = 0, bytes_read = 0;
  bool has_checksum = false;
  uint32_t checksum = 0;
  std::vector<uint8_t> buf;   // decoded, not yet drained bytes (DecodeBuffer, decode_buffer.rs:9-17)
  size_t head = 0;
  Xxh64 hash;
  size_t held() const { return buf.size() - head; }
};

static size_t dec_drain(zgpu_decoder* d, size_t n, uint8_t* dst) {  // DecodeBuffer::drain_to decode_buffer.rs:256-314
  if (!n) return 0;
  if (dst) memcpy(dst, d->buf.data() + d->head, n);
  d->hash.update(d->buf.data() + d->head, n);
  d->head += n;
  if (d->head == d->buf.size()) { d->buf.clear(); d->head = 0; }
  return n;
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
void zgpu_decoder_destroy(zgpu_decoder* d) { delete d; }

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
  d->bytes_read = c; d->has_checksum = false; d->checksum = 0;
  d->buf.clear(); d->head = 0; d->hash.reset(0);
  if (consumed) *consumed = c;
  if (h.has_dict_id) return ZGPU_E_DICT_NOT_PROVIDED;
  return ZGPU_OK;
}

int zgpu_decoder_decode_blocks(zgpu_decoder* d, const uint8_t* src, size_t len, size_t* consumed, int strat, size_t n, int* frame_finished) {
  (void)n;
  if (consumed) *consumed = 0;
  if (!d->has_state) return ZGPU_E_NOT_INITIALIZED;
  if (strat != ZGPU_STRAT_ALL) return ZGPU_E_UNSUPPORTED;
  // the block loop (frame_decoder.rs:319-375): the engine walks the remaining block headers of the frame and
  // decodes the whole run on the device
  Engine* eng = d->ctx->eng;
  Batch* b = nullptr;
  size_t p = 0;
  int st = eng->prepare_run(src, len, d->window_size, d->fh.content_checksum(), &b, &p);
  if (st) return st;
  const uint64_t nblocks = b->info.empty() ? 0 : b->info[0].nblocks;
  const bool has_ck = !b->info.empty() && b->info[0].has_checksum;
  const uint32_t ck = has_ck ? b->info[0].checksum : 0;
  if (b->parse_status) { st = b->parse_status; delete b; return st; }
  if ((st = b->run()) || (st = b->sync())) { delete b; return st; }
  if (b->frame_out.empty()) { delete b; return ZGPU_E_INTERNAL; }
  if (b->frame_out[0].status) { st = (int)b->frame_out[0].status; delete b; return st; }
  if (b->overflow) { delete b; return ZGPU_E_UNSUPPORTED; }
  uint64_t total = b->total_out;
  size_t old = d->buf.size();
  d->buf.resize(old + total);
  st = b->read_output(0, d->buf.data() + old, total);
  delete b;
  if (st) return st;
  d->block_counter += nblocks;
  d->bytes_read += p;
  d->frame_finished = true;
  if (has_ck) { d->has_checksum = true; d->checksum = ck; }
  if (consumed) *consumed = p;
  if (frame_finished) *frame_finished = 1;
  return ZGPU_OK;
}

int zgpu_decoder_is_finished(const zgpu_decoder* d) {
  if (!d->has_state) return 1;
  if (d->fh.content_checksum()) return d->frame_finished && d->has_checksum;
  return d->frame_finished;
}
size_t zgpu_decoder_can_collect(const zgpu_decoder* d) {
  if (!d->has_state) return 0;
  if (zgpu_decoder_is_finished(d)) return d->held();
  return d->held() > d->window_size ? d->held() - (size_t)d->window_size : 0;  // decode_buffer.rs:182-188
}
size_t zgpu_decoder_collect(zgpu_decoder* d, uint8_t* dst, size_t cap) {
  size_t n = zgpu_decoder_can_collect(d);
  if (n > cap) n = cap;
  return dec_drain(d, n, dst);
}
size_t zgpu_decoder_read(zgpu_decoder* d, uint8_t* dst, size_t cap) {
  if (!d->has_state) return 0;
  size_t n = d->frame_finished ? d->held() : (d->held() > d->window_size ? d->held() - (size_t)d->window_size : 0);
  if (n > cap) n = cap;
  return dec_drain(d, n, dst);
}
uint64_t zgpu_decoder_blocks_decoded(const zgpu_decoder* d) { return d->has_state ? d->block_counter : 0; }
uint64_t zgpu_decoder_bytes_read_from_source(const zgpu_decoder* d) { return d->has_state ? d->bytes_read : 0; }
uint64_t zgpu_decoder_content_size(const zgpu_decoder* d) { return d->has_state ? d->fh.frame_content_size : 0; }
int zgpu_decoder_checksum_from_data(const zgpu_decoder* d, uint32_t* out) {
  if (!d->has_state || !d->has_checksum) return 0;
  *out = d->checksum;
  return 1;
}
uint32_t zgpu_decoder_calculated_checksum(const zgpu_decoder* d) { return (uint32_t)d->hash.digest(); }

}  // extern "C"
