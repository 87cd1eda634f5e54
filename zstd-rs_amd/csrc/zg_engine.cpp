// zg_engine.cpp — see zg_engine.h. Compiled by hipcc as host code.
#include "zg_engine.h"
#include <stdlib.h>
#include <string.h>

namespace zg {

#define ZG_HIP(call)                                  \
  do {                                                \
    hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return eng->fail(e_, #call); \
  } while (0)

int DevBuf::reserve(size_t n, bool keep, hipStream_t s) {
  if (n <= cap && p) return 0;
  size_t ncap = n < 256 ? 256 : n;
  // a buffer that grows again: amortise, but by no more than 256 MiB (the large ones are sized exactly by their callers)
  if (p) { const size_t slack = cap / 2 < (256u << 20) ? cap / 2 : (256u << 20); if (ncap < cap + slack) ncap = cap + slack; }
  if (p && !(keep && cap)) { (void)hipFree(p); p = nullptr; cap = 0; }   // nothing to keep: free first (old + new need not fit together)
  void* np = nullptr;
  if (hipMalloc(&np, ncap) != hipSuccess) { (void)hipGetLastError(); return ZG_NOMEM; }   // (the failure must not stay behind as the runtime's last error: the next launch check would report it)
  if (p) {
    if (hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      (void)hipFree(np);
      return ZG_HIP_ERROR;
    }
    (void)hipFree(p);
  }
  p = np;
  cap = ncap;
  return 0;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

int Scratch::init_events() {
  if (have_events) return 0;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return ZG_HIP_ERROR;
  for (auto& e : ev_huf)
    if (hipEventCreate(&e) != hipSuccess) return ZG_HIP_ERROR;
  if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return ZG_HIP_ERROR;
  if (hipEventCreateWithFlags(&ev_fork3, hipEventDisableTiming) != hipSuccess) return ZG_HIP_ERROR;
  if (hipEventCreateWithFlags(&ev_up, hipEventDisableTiming) != hipSuccess) return ZG_HIP_ERROR;
  for (auto& e : ev_sw)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return ZG_HIP_ERROR;
  for (auto& e : ev_flat)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return ZG_HIP_ERROR;
  have_events = true;
  return 0;
}
void Scratch::release_but_output() {
  DevBuf* all[] = {&d_src, &d_aux, &d_fparsed, &d_slot_log, &d_fse, &d_huf, &d_hufmax, &d_lit, &d_seq, &d_og, &d_raw};
  for (DevBuf* b : all) b->release();
}
void Scratch::release() {
  DevBuf* all[] = {&d_src, &d_blocks, &d_frames, &d_aux, &d_fparsed, &d_slot_log, &d_fse, &d_huf, &d_hufmax, &d_status, &d_lit, &d_seq, &d_seqout, &d_pos,
                   &d_frameout, &d_dst, &d_seqblocks, &d_hufitems, &d_hufgroups, &d_totals, &d_og, &d_units, &d_unitinfo, &d_stepunits, &d_swdesc,
                   &d_dbg, &d_raw, &d_unitlist};
  for (DevBuf* b : all) b->release();
  for (auto& e : ev)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : ev_huf)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
  if (ev_fork3) { (void)hipEventDestroy(ev_fork3); ev_fork3 = nullptr; }
  if (ev_up) { (void)hipEventDestroy(ev_up); ev_up = nullptr; }
  for (auto& e : ev_sw)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto& e : ev_flat)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
  have_events = false;
}

int Engine::fail(hipError_t e, const char* what) {
  last_error = std::string(what) + ": " + hipGetErrorString(e);
  return ZG_HIP_ERROR;
}

Tuning Tuning::from_env() {
  Tuning t;
#ifdef ZG_DEV_SWITCHES   // the development build only (libzgpu_dev.so): the product library never looks at the environment
  t.dev_build = true;
  // (a negative value is ignored, not wrapped into a huge unsigned one: ADVICE r5)
  auto num = [](const char* name, uint32_t* v, bool positive) { const char* e = getenv(name); if (e && atoi(e) >= (positive ? 1 : 0)) *v = (uint32_t)atoi(e); return e != nullptr && atoi(e) >= 0; };
  auto is0 = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
  num("ZGPU_UNIT_BLOCKS", &t.unit_blocks, true);
  t.direct = !is0("ZGPU_DIRECT");
  num("ZGPU_RAMP", &t.ramp_percent, false);
  t.sparse_set = num("ZGPU_SPARSE_MAX", &t.sparse_max, false);
  if (is0("ZGPU_LIT_DIRECT")) t.lit_direct = 0;
  { uint32_t v = 0; if (num("ZGPU_DIRECT_SHARE", &v, true) && v >= 10) t.direct_share10 = v; }
  t.debug_timers = getenv("ZGPU_DEBUG_TIMERS") != nullptr;
  { const char* e = getenv("ZGPU_FORCE_INORDER"); t.force_inorder = e && e[0] == '1'; }
  num("ZGPU_FLAT_MODE", &t.flat_mode, false);
  num("ZGPU_SWEEP_W", &t.sweep_w, true);
  t.overlap = !is0("ZGPU_OVERLAP");
  t.no_sweep = getenv("ZGPU_DEBUG_NO_SWEEP") != nullptr;
  t.sweep_split = !is0("ZGPU_SWEEP_SPLIT");
  t.no_exact = getenv("ZGPU_DEBUG_NO_EXACT") != nullptr;
  t.no_presize = is0("ZGPU_PRESIZE");
  { uint32_t v = 0; if (num("ZGPU_FLAT4", &v, false)) t.flat4 = (int)v; }
  { uint32_t v = 0; if (num("ZGPU_SEQ_PACKED", &v, false)) t.seq_packed = (int)v; }
  { const char* e = getenv("ZGPU_FLAT_T"); t.flat_shape = (e && atoi(e) == 512) ? 1 : 0; }   // "512": 512 threads x 8 KiB tiles, two workgroups per CU; else the default
  num("ZGPU_SWEEP_MODE", &t.sweep.mode, false);
  num("ZGPU_SWEEP_NB", &t.sweep.nbatch, true);
  num("ZGPU_SWEEP_GROUP", &t.sweep.group, true);
  num("ZGPU_SWEEP_HEAD_LDS", &t.sweep.head_lds, false);
  num("ZGPU_SWEEP_HEAD_NB", &t.sweep.head_nbatch, true);
#endif
  return t;
}

int Engine::create(int device, Engine** out) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) return ZG_HIP_ERROR;  // no GPU: fail loudly, no CPU path
  if (hipSetDevice(device) != hipSuccess) return ZG_HIP_ERROR;
  Engine* e = new Engine();
  e->device_ = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->cus_ = prop.multiProcessorCount;
    e->tn_ = Tuning::from_env();                        // the only place the engine looks at the environment
    e->flat_shape_ = e->tn_.flat_shape;
    e->no_presize_ = e->tn_.no_presize;
  }
  // two streams: the sequences chain is the critical one (its kernels last as long as one block's serial chain), so its
  // workgroups are dispatched first; the literals chain fills what is left
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&e->stream_, hipStreamNonBlocking, prio_hi) != hipSuccess) { delete e; return ZG_HIP_ERROR; }
  if (hipStreamCreateWithPriority(&e->stream2_, hipStreamNonBlocking, prio_lo) != hipSuccess) { delete e; return ZG_HIP_ERROR; }
  if (hipStreamCreateWithPriority(&e->stream3_, hipStreamNonBlocking, prio_hi) != hipSuccess) { delete e; return ZG_HIP_ERROR; }
  if (hipStreamCreateWithPriority(&e->stream4_, hipStreamNonBlocking, prio_lo) != hipSuccess) { delete e; return ZG_HIP_ERROR; }
  if (hipStreamCreateWithPriority(&e->stream5_, hipStreamNonBlocking, prio_hi) != hipSuccess) { delete e; return ZG_HIP_ERROR; }
  *out = e;
  return ZG_OK;
}
Engine::~Engine() {
  (void)hipSetDevice(device_);
  for (Scratch* s : free_) { s->release(); delete s; }
  if (stream_) (void)hipStreamDestroy(stream_);
  if (stream2_) (void)hipStreamDestroy(stream2_);
  if (stream3_) (void)hipStreamDestroy(stream3_);
  if (stream4_) (void)hipStreamDestroy(stream4_);
  if (stream5_) (void)hipStreamDestroy(stream5_);
}
Scratch* Engine::acquire() {
  if (!free_.empty()) { Scratch* s = free_.back(); free_.pop_back(); return s; }
  return new Scratch();
}
void Engine::recycle(Scratch* s) {
  if (free_.size() < 2) free_.push_back(s);   // what a streaming decoder or a queue worker alternates between
  else { s->release(); delete s; }
}

int parse_frames(const uint8_t* src, size_t len, uint64_t max_window, BatchBuilder* bb, std::vector<FrameInfo>* info) {
  // FrameDecoder::decode_all (frame_decoder.rs:541-577): concatenated frames, skippable frames skipped;
  // the first error ends the walk (the reference returns it).
  size_t p = 0;
  static const uint32_t kHist[3] = {1, 4, 8};  // scratch.rs:44
  while (p < len) {
    FrameHeader h;
    size_t c;
    uint32_t sm = 0, sl = 0;
    int st = read_frame_header(src + p, len - p, &h, &c, &sm, &sl);
    if (st == ZG_SKIP_FRAME) {
      p += c;
      if ((size_t)sl > len - p) return ZG_FAILED_SKIP_FRAME;  // :550-556
      p += sl;
      continue;
    }
    if (st) return st;
    uint64_t w;
    if ((st = frame_window_size(h, &w))) return st;
    if (w > max_window) return ZG_WINDOW_SIZE_TOO_BIG;          // frame_decoder.rs:137-145
    if (h.has_dict_id) return ZG_DICT_NOT_PROVIDED;             // :212-217 (dictionaries: see DESIGN.md "next")
    FrameInfo fi;
    fi.header = h;
    fi.window_size = w;
    fi.src_begin = p;
    p += c;
    bb->begin_frame(w, kHist, 0);
    for (;;) {
      if (len - p < 3) { st = ZG_FAILED_READ_BLOCK_HEADER; break; }
      BlockHeader bh;
      if ((st = read_block_header(src + p, &bh))) break;
      p += 3;
      if (len - p < bh.content_size) { st = ZG_FAILED_READ_BLOCK_BODY; break; }
      st = bb->add_block(bh, src + p, p);
      p += bh.content_size;
      fi.nblocks++;
      if (st) break;
      if (bh.last) {
        if (h.content_checksum()) {
          if (len - p < 4) { st = ZG_FAILED_READ_CHECKSUM; break; }
          memcpy(&fi.checksum, src + p, 4);
          fi.has_checksum = true;
          p += 4;
        }
        break;
      }
    }
    fi.src_end = p;
    fi.host_status = st;
    info->push_back(fi);
    if (st) return st;
  }
  return ZG_OK;
}


int split_frames(const uint8_t* src, size_t len, std::vector<FrameSpan>* out) {
  // the walk of parse_frames without the section headers: frame header, then block headers up to the last block
  size_t p = 0;
  while (p < len) {
    FrameHeader h;
    size_t c;
    uint32_t sm = 0, sl = 0;
    FrameSpan sp;
    sp.begin = p; sp.content_size = 0; sp.has_content_size = false; sp.skippable = false;
    int st = read_frame_header(src + p, len - p, &h, &c, &sm, &sl);
    if (st == ZG_SKIP_FRAME) {
      p += c;
      if ((size_t)sl > len - p) return ZG_FAILED_SKIP_FRAME;
      p += sl;
      sp.end = p; sp.skippable = true;
      out->push_back(sp);
      continue;
    }
    if (st) return st;
    p += c;
    for (;;) {
      if (len - p < 3) return ZG_FAILED_READ_BLOCK_HEADER;
      BlockHeader bh;
      if ((st = read_block_header(src + p, &bh))) return st;
      p += 3;
      if (len - p < bh.content_size) return ZG_FAILED_READ_BLOCK_BODY;
      p += bh.content_size;
      if (bh.last) {
        if (h.content_checksum()) {
          if (len - p < 4) return ZG_FAILED_READ_CHECKSUM;
          p += 4;
        }
        break;
      }
    }
    sp.end = p; sp.content_size = h.frame_content_size; sp.has_content_size = h.has_fcs();
    out->push_back(sp);
  }
  return ZG_OK;
}

Batch::~Batch() {
  // a run uses all three of the engine's streams (Huffman chain and sweep heads on the second, the ramped chain on the third) and
  // joins them by events only when it reaches its end: nothing may be in flight on any of them when the buffers go back
  if (sc && eng) {
    (void)hipSetDevice(eng->device_);
    (void)hipStreamSynchronize(eng->stream_); (void)hipStreamSynchronize(eng->stream2_); (void)hipStreamSynchronize(eng->stream3_);
    if (wait_upload) (void)hipStreamSynchronize(eng->stream4_);   // (prepared beside another submit and never run)
    (void)hipStreamSynchronize(eng->stream5_);                      // (the second half of a download)
    eng->recycle(sc);
  }
}

void FrameState::reset() {
  base = 0; have = 0; produced = 0; counted = 0; hist[0] = 1; hist[1] = 4; hist[2] = 8;
  logs[0] = logs[1] = logs[2] = logs[3] = 0; huf_maxbits = 0; carry_mask = 0; window_size = 0;
}
void FrameState::release() { d_out.release(); d_fse.release(); d_huf.release(); d_tmp.release(); }

int FrameState::make_room(uint64_t extra, uint64_t keep, hipStream_t s) {
  if (keep > have) keep = have;
  // With a dictionary in front, the window is kept the way the reference keeps it: DecodeBuffer holds the undrained bytes and nothing older,
  // and a match that starts in front of them is served from the dictionary's tail and then from the OLDEST undrained byte
  // (repeat_from_dict, decode_buffer.rs:144-179) — bytes that have been drained are not in between. So they are not in between here either:
  // [dictionary content][undrained bytes], and every kernel's copy "offset bytes back" lands where the reference's does. (Only such a
  // non-conforming match can tell the difference; a conforming one stays inside the window, which is never drained.)
  if (base && keep < have && d_out.p) {
    uint8_t* dst = out_ptr() + base;
    const uint8_t* src = dst + (have - keep);
    hipError_t e = hipSuccess;
    if (keep && have - keep >= keep) e = hipMemcpyAsync(dst, src, keep, hipMemcpyDeviceToDevice, s);   // the ranges do not overlap
    else if (keep) {
      if (d_tmp.reserve(keep)) return ZG_NOMEM;
      e = hipMemcpyAsync(d_tmp.p, src, keep, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipMemcpyAsync(dst, d_tmp.p, keep, hipMemcpyDeviceToDevice, s);
    }
    if (e != hipSuccess) return ZG_HIP_ERROR;
    have = keep;
  }
  const uint64_t need = kOutFront + base + have + extra + 64;
  if (d_out.p && need <= d_out.cap) return 0;
  // In place: the buffer is large enough for what must stay plus the new bytes once the dropped ones are gone (a streaming decoder that
  // reads ahead reserves its window once, reserve_window, and then only ever gets here: no allocation on the submit path). The kept tail
  // moves to the front — through the staging buffer when the two ranges overlap.
  if (d_out.p && !base && kOutFront + keep + extra + 64 <= d_out.cap) {
    uint8_t* dst = out_ptr();
    const uint8_t* src = dst + (have - keep);
    hipError_t e = hipSuccess;
    if (keep && have - keep >= keep) e = hipMemcpyAsync(dst, src, keep, hipMemcpyDeviceToDevice, s);
    else if (keep) {
      if (d_tmp.reserve(keep)) return ZG_NOMEM;
      e = hipMemcpyAsync(d_tmp.p, src, keep, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipMemcpyAsync(dst, d_tmp.p, keep, hipMemcpyDeviceToDevice, s);
    }
    if (e != hipSuccess) return ZG_HIP_ERROR;
    have = keep;
    return 0;
  }
  // rebuild: [dictionary][the last `keep` frame bytes] move to a new buffer with room to grow; what the caller has drained
  // and no match can reach any more (decode_buffer.rs:182-219) is dropped here, so the device window stays bounded
  const uint64_t live = base + keep;
  uint64_t ncap = kOutFront + live + extra + 64;
  const uint64_t slack = live + extra > (8u << 20) ? (live + extra) / 2 : (4u << 20);
  ncap += slack;
  void* np = nullptr;
  if (hipMalloc(&np, ncap) != hipSuccess) { (void)hipGetLastError(); return ZG_NOMEM; }   // (the failure must not stay behind as the runtime's last error: the next launch check would report it)
  if (d_out.p) {
    hipError_t e = hipSuccess;
    if (base) e = hipMemcpyAsync((uint8_t*)np + kOutFront, out_ptr(), base, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && keep) e = hipMemcpyAsync((uint8_t*)np + kOutFront + base, out_ptr() + base + (have - keep), keep, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(np); return ZG_HIP_ERROR; }
    (void)hipFree(d_out.p);
  }
  d_out.p = np; d_out.cap = ncap;
  have = keep;
  return 0;
}

// Room for `payload` bytes behind the front pad ([dictionary][frame bytes]), what exists is kept: a streaming decoder that reads ahead
// reserves the window + two runs once, so that make_room never allocates while a run's plaintext is still travelling to the host.
int FrameState::reserve_window(uint64_t payload, hipStream_t s) {
  const uint64_t ncap = kOutFront + payload + 64;
  if (d_out.p && d_out.cap >= ncap) return 0;
  void* np = nullptr;
  if (hipMalloc(&np, ncap) != hipSuccess) { (void)hipGetLastError(); return ZG_NOMEM; }
  if (d_out.p) {
    hipError_t e = hipSuccess;
    if (base + have) e = hipMemcpyAsync((uint8_t*)np + kOutFront, out_ptr(), base + have, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(np); return ZG_HIP_ERROR; }
    (void)hipFree(d_out.p);
  }
  d_out.p = np; d_out.cap = ncap;
  return 0;
}

int parse_block_run(const uint8_t* src, size_t len, uint64_t window, bool has_checksum, const uint32_t hist[3], uint32_t carry_mask,
                    uint32_t max_blocks, BatchBuilder* bb, std::vector<FrameInfo>* info, size_t* consumed, bool* saw_last) {
  // the block loop of FrameDecoder::decode_blocks (frame_decoder.rs:319-375) on a run that starts at a block header.
  // Blocks in front of a truncated one stay in the run (the reference decodes them before it fails).
  FrameInfo fi;
  fi.window_size = window;
  bb->begin_frame(window, hist, carry_mask);
  size_t p = 0;
  int st = ZG_OK;
  *saw_last = false;
  for (;;) {
    if (len - p < 3) { st = ZG_FAILED_READ_BLOCK_HEADER; break; }
    BlockHeader bh;
    if ((st = read_block_header(src + p, &bh))) break;
    if (len - p - 3 < bh.content_size) { st = ZG_FAILED_READ_BLOCK_BODY; break; }
    p += 3;
    st = bb->add_block(bh, src + p, p);
    p += bh.content_size;
    fi.nblocks++;
    if (st) break;
    if (bh.last) {
      *saw_last = true;
      if (has_checksum) {
        if (len - p < 4) { st = ZG_FAILED_READ_CHECKSUM; break; }
        memcpy(&fi.checksum, src + p, 4);
        fi.has_checksum = true;
        p += 4;
      }
      break;
    }
    if (max_blocks && fi.nblocks >= max_blocks) break;
  }
  fi.src_end = p;
  fi.host_status = st;
  info->push_back(fi);
  *consumed = p;
  return st;
}


int Engine::prepare(const uint8_t* src, size_t len, Batch** out) {
  Batch* b = new Batch();
  b->eng = this;
  b->src_len = len;
  b->parse_status = parse_frames(src, len, max_window, &b->bb, &b->info);
  b->all_declared = !b->info.empty();
  for (const FrameInfo& fi : b->info) {
    if (!fi.header.has_fcs()) { b->all_declared = false; break; }
    b->declared_total += fi.header.frame_content_size;
  }
  if (!b->all_declared || b->declared_total > (1ull << 40)) { b->all_declared = false; b->declared_total = 0; }
  return upload(b, src, len, out);
}

int Engine::prepare_run(const uint8_t* src, size_t len, FrameState* fs, bool has_checksum, uint32_t max_blocks, uint64_t keep, Batch** out, size_t* consumed,
                        bool side, const uint32_t* carry_mask_now) {
  Batch* b = new Batch();
  b->eng = this;
  b->fs = fs;
  b->keep_bytes = keep;
  // (carry_mask_now: the run in front has not been folded into fs yet — its caller says which tables will exist when this one starts; the
  //  offset history is taken when the run is launched, Batch::run)
  b->carry_mask_in = carry_mask_now ? *carry_mask_now : fs->carry_mask;
  b->parse_status = parse_block_run(src, len, fs->window_size, has_checksum, fs->hist, b->carry_mask_in, max_blocks, &b->bb, &b->info, consumed,
                                    &b->saw_last_block);
  b->src_len = *consumed;
  b->bb.frames[0].fixed_base = 1;   // where its bytes go is decided in run(), when the size of the run is known
  int st;
  if ((st = fs->d_fse.reserve(ZG_FSE_SLOT_U32 * 4)) || (st = fs->d_huf.reserve(ZG_HUF_SLOT_U16 * 2))) { delete b; return st; }   // the tables it will carry on
  return upload(b, src, *consumed, out, side);
}

int Engine::prepare_blocks(const uint8_t* src, size_t len, const HostBlock* hb, size_t n, FrameState* fs, uint64_t keep, Batch** out) {
  // the block loop of FrameDecoder::decode_blocks (frame_decoder.rs:319-375) with the headers already read by the caller
  Batch* b = new Batch();
  b->eng = this;
  b->fs = fs;
  b->keep_bytes = keep;
  b->src_len = len;
  FrameInfo fi;
  fi.window_size = fs->window_size;
  b->bb.begin_frame(fs->window_size, fs->hist, fs->carry_mask);
  int st = ZG_OK;
  for (size_t i = 0; i < n && !st; i++) {
    const HostBlock& h = hb[i];
    BlockHeader bh;
    bh.last = h.last != 0; bh.type = h.type;
    if (h.type > ZG_BT_COMPRESSED) { st = ZG_RESERVED_BLOCK; b->bb.fail_frame(st); break; }                 // block_decoder.rs:226-228
    if ((h.type == ZG_BT_COMPRESSED ? h.src_len : h.raw_rle_size) > kMaxBlockSize) { st = ZG_BLOCK_SIZE_TOO_LARGE; b->bb.fail_frame(st); break; }   // :259-266
    if (h.src_off > len || h.src_len > len - h.src_off) { st = ZG_FAILED_READ_BLOCK_BODY; b->bb.fail_frame(st); break; }
    bh.decompressed_size = h.type == ZG_BT_COMPRESSED ? 0 : h.raw_rle_size;
    bh.content_size = h.type == ZG_BT_RLE ? 1 : h.src_len;
    if (h.type == ZG_BT_RAW && h.src_len != h.raw_rle_size) { st = ZG_FAILED_READ_BLOCK_BODY; b->bb.fail_frame(st); break; }
    if (h.type == ZG_BT_RLE && h.src_len < 1) { st = ZG_FAILED_READ_BLOCK_BODY; b->bb.fail_frame(st); break; }
    st = b->bb.add_block(bh, src + h.src_off, h.src_off);
    fi.nblocks++;
    if (bh.last) { b->saw_last_block = true; break; }
  }
  fi.host_status = st;
  b->info.push_back(fi);
  b->parse_status = st;
  b->bb.frames[0].fixed_base = 1;
  if ((st = fs->d_fse.reserve(ZG_FSE_SLOT_U32 * 4)) || (st = fs->d_huf.reserve(ZG_HUF_SLOT_U16 * 2))) { delete b; return st; }
  return upload(b, src, len, out);
}

int Engine::upload(Batch* b, const uint8_t* src, size_t len, Batch** out, bool side) {
  Engine* eng = this;
  ZG_HIP(hipSetDevice(device_));
  // side: the submit in front is still running on the main stream — everything this one brings to the device travels on the upload stream
  // beside it, and its run() waits for the event instead of the host waiting here
  hipStream_t us = side ? stream4_ : stream_;
  // a frame-layer error after some good frames: the good frames are still uploaded (callers such as the
  // FrameDecoder mirror may want them); decode_all reports parse_status.
  const Tuning& tn = tn_;                                      // (measurement / test switches, read when the engine was created)
  if (tn.unit_blocks) b->bb.unit_blocks = tn.unit_blocks;
  if (!tn.direct) b->bb.direct_units = false;
  b->bb.ramp_percent = tn.ramp_percent;                        // N > 0: one long frame in units growing by +-N %, the sweep chain beside the flatten
  if (tn.sparse_set) { b->bb.sparse_max = tn.sparse_max; b->bb.sparse_per_block = 1u << 20; }   // (tests) sequences per frame up to which zg_k_sparse replaces the sweep, whatever their density; 0: never
  b->bb.flat_slots = (uint32_t)cus_ * (flat_shape_ == 0 ? 1u : 2u);   // zg_k_flatten: workgroups the device holds at once
  b->bb.chain_slots = (uint32_t)cus_ * 32u;                            // zg_k_seq: blocks whose chains run at once
  if (tn.lit_direct == 0) b->bb.lit_direct_allowed = false;
  b->bb.seq_packed_force = tn.seq_packed;
  if (tn.direct_share10) b->bb.direct_share10 = tn.direct_share10;   // (measurement) tenths
  b->bb.finish();
  BatchBuilder& bb = b->bb;
  Scratch* sc = b->sc = acquire();
  const uint32_t nb = (uint32_t)bb.blocks.size(), nf = (uint32_t)bb.frames.size();
  auto up = [&](DevBuf& d, const void* h, size_t bytes) -> int {
    int st = d.reserve(bytes ? bytes : 16);
    if (st) return st;
    if (bytes && hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, us) != hipSuccess) return ZG_HIP_ERROR;
    return 0;
  };
  int st = 0;
  // compressed bytes, padded so that 8-byte bit-window loads near the end stay inside the allocation
  // ... and 64 bytes in front for the 16-byte windows of the sequence decoder
  if ((st = sc->d_src.reserve(len + 128))) { delete b; return st; }
  (void)hipMemsetAsync(sc->d_src.p, 0, 64, us);
  if (len && hipMemcpyAsync((uint8_t*)sc->d_src.p + 64, src, len, hipMemcpyHostToDevice, us) != hipSuccess) { delete b; return ZG_HIP_ERROR; }
  (void)hipMemsetAsync((uint8_t*)sc->d_src.p + 64 + len, 0, 64, us);
  const uint32_t nslots = bb.nslots();
  if ((st = up(sc->d_blocks, bb.blocks.data(), nb * sizeof(ZgBlock))) || (st = up(sc->d_frames, bb.frames.data(), nf * sizeof(ZgFrame))) ||
      (st = up(sc->d_seqblocks, bb.seq_blocks.data(), bb.seq_blocks.size() * 4)) ||
      (st = up(sc->d_hufitems, bb.huf_items.data(), bb.huf_items.size() * 4)) ||
      (st = up(sc->d_hufgroups, bb.huf_groups.data(), bb.huf_groups.size() * sizeof(ZgHufGroup))) ||
      (st = up(sc->d_units, bb.units.data(), bb.units.size() * sizeof(ZgUnit))) ||
      (st = up(sc->d_stepunits, bb.step_units.data(), bb.step_units.size() * 4)) ||
      (st = up(sc->d_unitlist, bb.unit_list.data(), bb.unit_list.size() * 4)) ||
      (st = sc->d_aux.reserve((size_t)nb * sizeof(ZgBlockAux) + 16)) || (st = sc->d_fparsed.reserve((size_t)nb * sizeof(ZgFtabParsed) + 16)) || (st = sc->d_slot_log.reserve((size_t)nslots * 4)) ||
      (st = sc->d_fse.reserve((size_t)nslots * ZG_FSE_SLOT_U32 * 4)) || (st = sc->d_huf.reserve((size_t)(bb.nhuf_slots + 1) * ZG_HUF_SLOT_U16 * 2)) ||
      (st = sc->d_hufmax.reserve(bb.nhuf_slots + 16)) || (st = sc->d_status.reserve(7 * ((size_t)nb * 4 + 16))) ||
      (st = sc->d_lit.reserve(bb.lit_bytes + 128)) || (st = sc->d_seq.reserve((bb.seq_count + 2) * sizeof(ZgSeq))) ||
      (st = sc->d_raw.reserve((bb.seq_count + 8) * 4)) ||
      (st = sc->d_seqout.reserve((size_t)nb * sizeof(ZgBlockSeqOut) + 16)) || (st = sc->d_pos.reserve((size_t)nb * sizeof(ZgBlockPos) + 16)) ||
      (st = sc->d_frameout.reserve((size_t)nf * sizeof(ZgFrameOut) + 16)) || (st = sc->d_totals.reserve(64)) ||
      (st = sc->d_swdesc.reserve(bb.step_units.size() * sizeof(ZgSweepDesc) + 32)) ||
      (st = sc->d_unitinfo.reserve(bb.units.size() * sizeof(ZgUnitInfo) + 16)) || (st = sc->d_dbg.reserve(8192)) || (st = sc->init_events())) {
    delete b;
    return st;
  }
  // the output and the flatten scratch are sized in run(), when the exact output size of every frame is known
  ZgBatchDev& d = b->dev;
  d.src = sc->d_src.as<uint8_t>() + 64; d.src_len = len;
  d.blocks = sc->d_blocks.as<ZgBlock>(); d.nblocks = nb;
  d.frames = sc->d_frames.as<ZgFrame>(); d.nframes = nf;
  d.nslots = nslots; d.nhuf_slots = bb.nhuf_slots;
  d.aux = sc->d_aux.as<ZgBlockAux>(); d.ftab_parsed = sc->d_fparsed.as<ZgFtabParsed>(); d.slot_log = sc->d_slot_log.as<uint8_t>();
  d.fse_arena = sc->d_fse.as<uint32_t>(); d.huf_arena = sc->d_huf.as<uint16_t>(); d.huf_maxbits = sc->d_hufmax.as<uint8_t>();
  d.status = sc->d_status.as<uint32_t>(); d.tab_status = d.status + nb + 4; d.lit_status = d.tab_status + nb + 4;
  d.lit_counts = d.lit_status + nb + 4;     // [4 * nblocks], written by zg_k_huf (no reset needed)
  d.lit_arena = sc->d_lit.as<uint8_t>() + 64;   // (zg_k_flatten reads literal windows that start up to 7 bytes in front of a block's literals)
  d.seq_arena = sc->d_seq.as<ZgSeq>(); d.raw_arena = sc->d_raw.as<uint32_t>();
  d.seq_out = sc->d_seqout.as<ZgBlockSeqOut>(); d.pos = sc->d_pos.as<ZgBlockPos>(); d.frame_out = sc->d_frameout.as<ZgFrameOut>();
  d.dst = nullptr; d.dst_cap = 0; d.og = nullptr;
  d.dict = nullptr;
  d.seq_blocks = sc->d_seqblocks.as<uint32_t>(); d.nseq_blocks = (uint32_t)bb.seq_blocks.size();
  d.huf_items = sc->d_hufitems.as<uint32_t>(); d.huf_groups = sc->d_hufgroups.as<ZgHufGroup>(); d.nhuf_groups = (uint32_t)bb.huf_groups.size();
  d.totals = sc->d_totals.as<uint32_t>();
  d.units = sc->d_units.as<ZgUnit>(); d.nunits = (uint32_t)bb.units.size(); d.unit_info = sc->d_unitinfo.as<ZgUnitInfo>();
  d.step_units = sc->d_stepunits.as<uint32_t>();
  d.unit_list = sc->d_unitlist.as<uint32_t>(); d.ndirect = bb.n_direct;
  d.sweep_desc = sc->d_swdesc.as<ZgSweepDesc>();
  b->sweep_steps.clear();
  for (const ZgStepRange& r : bb.steps) {
    ZgSweepStep ss;
    ss.list_off = r.list_off; ss.nunits = r.nunits; ss.slices = r.max_blocks * (kMaxBlockSize / ZG_SW_BATCH); ss.pad = 0;   // zg_k_sweep: ZG_SW_BATCH bytes of output per workgroup
    b->sweep_steps.push_back(ss);
  }
  d.dbg = tn.debug_timers ? sc->d_dbg.as<unsigned long long>() : nullptr;
  d.flags = tn.force_inorder ? 1u : 0u;
  d.flags |= (uint32_t)flat_shape_ << 2;
  d.flags |= ((tn.flat_mode & 3u) << 4) | ((tn.flat_mode & 4u) << 5);   // (timing experiments) zg_k_flatten without scratch stores / gathers
  d.sweep_window = tn.sweep_w;
  if (side) {
    if (hipEventRecord(sc->ev_up, us) != hipSuccess) { delete b; return ZG_HIP_ERROR; }
    b->wait_upload = true;
  } else if (hipStreamSynchronize(stream_) != hipSuccess) { delete b; return ZG_HIP_ERROR; }
  *out = b;
  return ZG_OK;
}

int Batch::run() {
  ZG_HIP(hipSetDevice(eng->device_));
  hipStream_t s = eng->stream_;
  ZgBatchDev& d = dev;
  if (d.nframes == 0) { ran = true; total_out = 0; return ZG_OK; }
  if (d.nblocks == 0) {
    // frames without a single block (the walk stopped at the first block header of each: parse_status / FrameInfo::host_status say why):
    // nothing to launch — a grid of zero workgroups is an error to the runtime (tools/dev/soak_batch.py) — and nothing comes out
    frame_out.assign(d.nframes, ZgFrameOut{});
    for (ZgFrameOut& fo : frame_out) fo.err_packed = 0xFFFFFFFFu;
    total_out = 0; ran = false; synced = true;
    return ZG_OK;
  }
  hipEvent_t* ev = sc->ev;
  // every frame declares its content size: output and flatten scratch are sized now, and nothing below waits for the host
  presized = false;
  d.dst_cap_pre = 0;
  // ... and the declared sizes are possible at all: a block regenerates at most 128 KiB on this path, and Frame_Content_Size is a field
  // the reference never believes (a corrupted one must not become a 600 GB allocation: the sizes are then taken from the scan)
  // (bb.out_bound: what the host already knows the blocks can produce at most — exact for raw and RLE blocks, 128 KiB per compressed block.
  //  A declared size beyond it is a lie whatever else is true — 500,000 empty raw blocks behind an FCS of 64 GiB must not become a 320 GiB
  //  allocation — and a reserve that fails all the same is no error either: the reference never looks at the field, so the
  //  submit falls back to sizes taken from the scan.)
  // (not with ramped units, a measurement switch: their sweep chain polls flags that the flatten of a no-op run never raises — ADVICE r4)
  // (not for a declared total of 0: the device's "sized in advance" marker is a capacity > 0, and a submit that declares nothing but produces
  //  one byte kept its promise by that marker while the host's capacity said 0 — zgpu_batch_read refused the byte, tools/dev/soak_batch.py)
  if (!fs && all_declared && declared_total && !eng->no_presize_ && declared_total <= bb.out_bound && !bb.ramped) {
    // (the scratch first: it is four times the output. When either reserve fails nothing of the attempt stays allocated — the sizes taken
    //  from the scan may well fit where the declared ones did not: ADVICE r5)
    if (sc->d_og.reserve(declared_total * 4 + 64) == 0 && sc->d_dst.reserve(kOutFront + declared_total + 64) == 0) {
      d.dst = sc->d_dst.as<uint8_t>() + kOutFront; d.dst_cap = declared_total;
      d.og = sc->d_og.as<uint32_t>(); d.og_words = og_words = declared_total;
      d.dst_cap_pre = declared_total;                        // (0 means "not sized in advance")
      presized = true;
    } else { sc->d_og.release(); sc->d_dst.release(); }
  }
  if (wait_upload) { ZG_HIP(hipStreamWaitEvent(s, sc->ev_up, 0)); wait_upload = false; }
  ZG_HIP(hipEventRecord(ev[0], s));
  if (fs) {
    // a continued frame: the offset history is the one the run in front has left (that run may have been folded into fs after this one
    // was prepared: the streaming decoder prepares run k + 1 while run k is on the GPU)
    ZgFrame& fr0 = bb.frames[0];
    fr0.hist_init[0] = fs->hist[0]; fr0.hist_init[1] = fs->hist[1]; fr0.hist_init[2] = fs->hist[2];
    ZG_HIP(hipMemcpyAsync((void*)d.frames, bb.frames.data(), sizeof(ZgFrame), hipMemcpyHostToDevice, s));
  }
  ZG_HIP(hipMemsetAsync(d.status, 0, 3 * ((size_t)d.nblocks * 4 + 16), s));
  ZG_HIP(hipMemsetAsync(d.huf_maxbits, 0, d.nhuf_slots + 16, s));
  ZG_HIP(hipMemsetAsync(d.slot_log, 0, (size_t)d.nslots * 4, s));
  ZG_HIP(hipMemsetAsync(d.totals, 0, 64, s));
  if (d.dbg) ZG_HIP(hipMemsetAsync(d.dbg, 0, 8192, s));
  if (fs && fs->carry_mask) {   // tables carried into this run: the frame's carry slots of the two arenas
    const ZgFrame& fr = bb.frames[0];
    ZG_HIP(hipMemcpyAsync(d.fse_arena + (size_t)fr.carry_slot * ZG_FSE_SLOT_U32, fs->d_fse.p, ZG_FSE_SLOT_U32 * 4, hipMemcpyDeviceToDevice, s));
    ZG_HIP(hipMemcpyAsync(d.slot_log + (size_t)fr.carry_slot * 4, fs->logs, 4, hipMemcpyHostToDevice, s));
    ZG_HIP(hipMemcpyAsync(d.huf_arena + (size_t)fr.carry_huf_slot * ZG_HUF_SLOT_U16, fs->d_huf.p, ZG_HUF_SLOT_U16 * 2, hipMemcpyDeviceToDevice, s));
    ZG_HIP(hipMemcpyAsync(d.huf_maxbits + fr.carry_huf_slot, &fs->huf_maxbits, 1, hipMemcpyHostToDevice, s));
  }
  // ---- phase 1: entropy stages + scan. Literals (Huffman tree descriptions -> streams) and sequences (FSE descriptions ->
  // state chains -> post-pass) of a block are independent, and all of these kernels are latency-bound chains that leave most
  // of the chip idle: the two chains run side by side on two streams and meet again in zg_k_merge.
  hipStream_t s2 = eng->stream2_;
  ZG_HIP(hipMemsetAsync(d.totals + 3, 0, 4, s));   // "a match reaches beyond its frame's window" (zg_k_seqpost)
  zg_launch_tables(d, s, 1);
  ZG_HIP(hipEventRecord(ev[1], s));
  // The literals chain starts when the FSE tables are done, i.e. together with zg_k_seq: the dispatcher then places the
  // high-priority zg_k_seq workgroups first (all of them must be resident at once: the kernel lasts as long as one
  // block's chain) and the Huffman workgroups fill the LDS that is left. Started earlier, they would sit in the CUs
  // when zg_k_seq arrives and push half of its workgroups into a second round.
  ZG_HIP(hipEventRecord(sc->ev_fork, s));
  ZG_HIP(hipStreamWaitEvent(s2, sc->ev_fork, 0));
  // (literal-heavy submits, BatchBuilder::finish: only the tree descriptions here; the streams are decoded after the scan, when
  //  the blocks without sequences know their place in the output — phase 2)
  const bool lit_direct = bb.lit_direct;
  d.flags &= ~ZG_FLAG_LIT_DIRECT;
  ZG_HIP(hipEventRecord(sc->ev_huf[0], s2));
  zg_launch_tables(d, s2, 0);
  if (!lit_direct) zg_launch_huf(d, s2);
  ZG_HIP(hipEventRecord(sc->ev_huf[1], s2));
  ZG_HIP(hipEventRecord(ev[2], s));
  // (more blocks with sequences than chains the device runs at once — CUs x 32 —: the kernel is then bound by chains per unit of time, not by
  //  the length of one chain, and the packed-entry form holds half as many again per CU)
  zg_launch_seq(d, s, bb.seq_packed);   // (BatchBuilder::finish decided: more blocks with sequences than one round holds -> the packed form)
  ZG_HIP(hipEventRecord(ev[3], s));
  zg_launch_seqpost(d, s);
  ZG_HIP(hipStreamWaitEvent(s, sc->ev_huf[1], 0));
  zg_launch_merge(d, s);
  ZG_HIP(hipEventRecord(ev[4], s));
  { uint32_t mb = 0; for (const ZgFrame& fr : bb.frames) mb = fr.nblocks > mb ? fr.nblocks : mb; zg_launch_scan(d, s, mb); }
  ZG_HIP(hipEventRecord(ev[5], s));
  // ---- the output buffer and the flatten scratch are sized to the frames' exact sizes, which the scan has just computed: one host
  // round trip — unless every frame declares its content size: then both were sized before the run (above), the LZ77 stages go
  // behind the scan right away, and the device checks the promise (zg_k_scanf; a frame that produces more than it declared turns
  // them into no-ops and sync() repeats them the slow way: Frame_Content_Size is never checked by the reference, frame_decoder.rs:541-577).
  if (!presized) { const int st = size_output(); if (st) return st; }
  const int st2 = launch_phase2();
  if (st2) return st2;
  ZG_HIP(hipGetLastError());
  ran = true;
  return ZG_OK;
}

int Batch::size_output() {
  hipStream_t s = eng->stream_;
  ZgBatchDev& d = dev;
  frame_out.resize(d.nframes);
  uint32_t totals[4] = {0, 0, 0, 0};
  ZG_HIP(hipMemcpyAsync(frame_out.data(), d.frame_out, (size_t)d.nframes * sizeof(ZgFrameOut), hipMemcpyDeviceToHost, s));
  ZG_HIP(hipMemcpyAsync(totals, d.totals, 16, hipMemcpyDeviceToHost, s));
  ZG_HIP(hipStreamSynchronize(s));
  total_out = (uint64_t)totals[0] | ((uint64_t)totals[1] << 32);
  far_seen = totals[3] != 0;
  int st = 0;
  bool any_fast = false;
  for (const ZgFrameOut& fo : frame_out) any_fast |= fo.fast != 0;
  if (fs) {
    const uint64_t add = frame_out[0].out_size;
    if ((st = fs->make_room(add + kMaxBlockSize, keep_bytes, s))) return st;   // (+ a block: what a block that fails in sequence execution leaves behind the run, zg_k_partial)
    // this run continues the frame: its bytes go right behind what exists, and matches may reach back into it — as far as
    // the caller still holds bytes (DecodeBuffer semantics: repeat() sees only what has not been drained)
    ZgFrame& fr = bb.frames[0];
    fr.prior_out = fs->produced;
    fr.prior_reach = keep_bytes < fs->have ? keep_bytes : fs->have;
    fr.prior_counted = fs->counted;
    fr.dict_len = fs->base;
    fr.out_base_fixed = fs->base + fs->have;
    ZG_HIP(hipMemcpyAsync((void*)d.frames, bb.frames.data(), sizeof(ZgFrame), hipMemcpyHostToDevice, s));
    ZG_HIP(hipMemcpyAsync(&d.frame_out[0].out_base, &fr.out_base_fixed, 8, hipMemcpyHostToDevice, s));
    frame_out[0].out_base = fr.out_base_fixed;
    d.dst = fs->out_ptr(); d.dst_cap = fs->base + fs->have + add;
    og_words = add;
  } else {
    if ((st = sc->d_dst.reserve(kOutFront + total_out + 64))) return st;
    d.dst = sc->d_dst.as<uint8_t>() + kOutFront; d.dst_cap = total_out;
    og_words = total_out;
  }
  if (any_fast && (st = sc->d_og.reserve(og_words * 4 + 64))) return st;
  d.og = sc->d_og.as<uint32_t>();
  d.og_words = og_words;
  return ZG_OK;
}

int Batch::launch_phase2() {
  hipStream_t s = eng->stream_;
  ZgBatchDev& d = dev;
  hipEvent_t* ev = sc->ev;
  // ---- phase 2: LZ77 execution (and, for literal-heavy submits, the Huffman streams in front of it)
  if (bb.lit_direct) {
    d.flags |= ZG_FLAG_LIT_DIRECT;
    ZG_HIP(hipEventRecord(sc->ev_huf[0], s));
    zg_launch_huf(d, s);
    zg_launch_litfix(d, s);
    ZG_HIP(hipEventRecord(sc->ev_huf[1], s));
  }
  zg_launch_lit(d, s);
  ZG_HIP(hipEventRecord(ev[6], s));
  sweep_mode = 0; synced = false;
  // The direct units (a frame's first unit, resolved to bytes: zg_flat4.h) have a kernel of their own, and with it their own occupancy.
  // ONE unit is flattened fastest by 1024 threads on 16 KiB tiles (the pointer-mode units' kernel, one workgroup per CU); a submit of
  // many direct units — single-block frames, BASELINE config 4b — is flattened fastest by small workgroups, eight to a CU, which overlap
  // each other's barriers (measured on 16384 single-block frames, flatten ms: one kernel 5.23; zg_k_flatten4 with 1024 / 512 / 256 threads
  // and 8 / 4 / 2 KiB tiles 4.23 / 3.84 / 3.59; on 64 x (1 direct + 1 pointer unit) the one kernel is the fastest: 15.3 against 15.7 - 16.9).
  int flat4 = eng->tn_.flat4;
  if (flat4 < 0) {
    const uint32_t cus = (uint32_t)eng->cus_, nd = d.ndirect;
    flat4 = nd >= 16u * cus ? 6 : nd >= 8u * cus ? 4 : nd >= 4u * cus ? 1 : 0;
  }
  // One long frame in ramped units (BatchBuilder::finish): the flatten goes to its own stream and the sweep chain starts at once;
  // a step waits for its unit's flag (zg_k_flatten sets it, zg_k_sweep polls it). ZGPU_OVERLAP=0: one after the other.
  const bool overlap = bb.ramped && eng->tn_.overlap && !eng->tn_.no_sweep;
  d.overlap_epoch = overlap ? ++epoch_ : 0u;
  if (overlap) {
    // the flatten stays on the main stream and is enqueued FIRST; the chain goes to the third stream. (Should the two streams
    // share a hardware queue, the chain then simply runs behind the flatten; the other way round its first step would sit in
    // front of the kernel it waits for.)
    hipStream_t s3 = eng->stream3_;
    ZG_HIP(hipMemsetAsync(d.unit_info, 0, (size_t)d.nunits * sizeof(ZgUnitInfo), s));   // the flags of earlier users of this memory
    ZG_HIP(hipEventRecord(sc->ev_fork3, s));
    zg_launch_flat(d, s, eng->stream2_, sc->ev_flat, flat4);
    ZG_HIP(hipEventRecord(ev[7], s));
    ZG_HIP(hipStreamWaitEvent(s3, sc->ev_fork3, 0));
    launch_sweep(true, s3);
    ZG_HIP(hipEventRecord(sc->ev_fork3, s3));
    ZG_HIP(hipStreamWaitEvent(s, sc->ev_fork3, 0));
  } else {
    zg_launch_flat(d, s, eng->stream2_, sc->ev_flat, flat4);
    { bool any = false; for (const ZgFrame& fr : bb.frames) any = any || fr.sparse; if (any) zg_launch_sparse(d, s); }
    ZG_HIP(hipEventRecord(ev[7], s));
    if (!eng->tn_.no_sweep) launch_sweep(true);
  }
  ZG_HIP(hipEventRecord(ev[8], s));
  zg_launch_lz(d, s);   // only frames that left the flatten path (a block regenerating > 128 KiB)
  ZG_HIP(hipEventRecord(ev[9], s));
  return ZG_OK;
}

// The sweep. split: tails on the main stream, heads beside them on the second one, which is right as long as no match
// reaches further back than its frame's window (zg_k_seqpost reports one that does: sync() then repeats the sweep the plain way).
void Batch::launch_sweep(bool split, hipStream_t main) {
  if (!main) main = eng->stream_;
  if (!eng->tn_.sweep_split) split = false;
  uint64_t wmax = 0, wmin = ~0ull;
  for (const ZgFrame& fr : bb.frames) {
    wmax = fr.window_size > wmax ? fr.window_size : wmax;
    wmin = fr.window_size < wmin ? fr.window_size : wmin;
    // an initial history other than the format's (dictionary, continued frame) may hold offsets nobody checked against the window
    if (fr.hist_init[0] != 1 || fr.hist_init[1] != 4 || fr.hist_init[2] != 8 || fr.dict_len || fr.prior_out) split = false;
  }
  if (dev.sweep_window) wmax = wmin = dev.sweep_window;
  if (wmax > 0x7FFFFFFFull) wmax = 0x7FFFFFFFull;
  if (wmin > wmax) wmin = wmax;
  split_sweep = zg_launch_sweep(dev, main, sweep_steps.data(), (uint32_t)sweep_steps.size(), eng->stream2_, sc->ev_sw, split ? 80u : 0u,
                                bb.unit_blocks_used * kMaxBlockSize, (uint32_t)wmax, (uint32_t)wmin, eng->tn_.sweep);
  sweep_mode = split_sweep ? 1u : sweep_mode;
}

void Batch::release_scratch() {
  if (!sc || !eng || !synced) return;
  (void)hipSetDevice(eng->device_);
  sc->release_but_output();
  dev.og = nullptr; dev.src = nullptr; dev.lit_arena = nullptr; dev.seq_arena = nullptr; dev.raw_arena = nullptr;
  ran = false;                     // (the submit cannot be run again; reading its output stays possible)
}

int Batch::sync() {
  ZG_HIP(hipSetDevice(eng->device_));
  ZG_HIP(hipStreamSynchronize(eng->stream_));
  if (dev.nframes == 0 || !ran) return ZG_OK;
  if (presized) {
    // what run() did not wait for: the frames' sizes, and whether they kept their promise
    uint32_t totals[4] = {0, 0, 0, 0};
    ZG_HIP(hipMemcpy(totals, dev.totals, 16, hipMemcpyDeviceToHost));
    total_out = (uint64_t)totals[0] | ((uint64_t)totals[1] << 32);
    far_seen = totals[3] != 0;
    frame_out.resize(dev.nframes);
    if (totals[2]) {
      // a frame produced more than it declared: the LZ77 stages did nothing (every kernel checks the flag). Size the buffers from
      // what the scan found and run them now, the way a submit without declared sizes goes.
      presized = false;
      dev.dst_cap_pre = 0;
      uint32_t zero = 0;
      ZG_HIP(hipMemcpy(dev.totals + 2, &zero, 4, hipMemcpyHostToDevice));
      int st = size_output();
      if (!st) st = launch_phase2();
      if (st) return st;
      ZG_HIP(hipStreamSynchronize(eng->stream_));
    }
  }
  if (split_sweep) {
    uint32_t far = 0;
    ZG_HIP(hipMemcpy(&far, dev.totals + 3, 4, hipMemcpyDeviceToHost));
    if (far) {   // a match longer than the window: the heads may have copied bytes that were not final yet
      launch_sweep(false);
      ZG_HIP(hipStreamSynchronize(eng->stream_));
      sweep_mode = 2;
    }
    split_sweep = false;
  }
  ZG_HIP(hipMemcpy(frame_out.data(), dev.frame_out, (size_t)dev.nframes * sizeof(ZgFrameOut), hipMemcpyDeviceToHost));
  {
    // The fast path treats every byte the frame has produced as reachable and approximates which of the two "offset too far"
    // errors applies; the reference's DecodeBuffer is stricter (zg_exact.h). Where that can change a verdict — an offset beyond
    // the window was seen, a frame ended with one of those errors, a dictionary is involved — the exact bookkeeping is replayed.
    bool need = far_seen;
    for (uint32_t f = 0; f < dev.nframes && !need; f++) {
      const ZgFrame& fr = bb.frames[f];
      const uint32_t st = frame_out[f].status;
      need = st == (uint32_t)ZG_EXE_OFFSET_TOO_BIG || st == (uint32_t)ZG_EXE_DICT_TOO_SMALL || fr.dict_len != 0 ||
             st == (uint32_t)ZG_EXE_NOT_ENOUGH_LITERALS || st == (uint32_t)ZG_EXE_ZERO_OFFSET ||   // (a sequence in front of the rejected one may reach too far: that comes first)
             fr.hist_init[0] > fr.window_size || fr.hist_init[1] > fr.window_size || fr.hist_init[2] > fr.window_size;
    }
    if (need && !eng->tn_.no_exact) {
      zg_launch_exact(dev, eng->stream_, drain_rule);
      ZG_HIP(hipStreamSynchronize(eng->stream_));
      ZG_HIP(hipMemcpy(frame_out.data(), dev.frame_out, (size_t)dev.nframes * sizeof(ZgFrameOut), hipMemcpyDeviceToHost));
      exact_ran = true;
    }
  }
  // a frame that failed in the execution stage (zg_k_flatten / zg_k_exact / zg_k_lz) still carries the size of all its blocks:
  // what it produced ends with its last good block, as for the entropy errors zg_k_scan trims itself
  // (host-side only: total_out and the device copy of frame_out keep the untrimmed sizes. One copy per failed frame while they are few; a
  //  submit of many corrupt frames fetches the block positions once instead of paying a round trip per frame — ADVICE r4)
  {
    uint32_t nfail = 0;
    for (uint32_t f = 0; f < dev.nframes; f++) nfail += frame_out[f].status && frame_out[f].good_blocks < bb.frames[f].nblocks;
    std::vector<ZgBlockPos> all_pos;
    if (nfail > 8) { all_pos.resize(dev.nblocks); ZG_HIP(hipMemcpy(all_pos.data(), dev.pos, (size_t)dev.nblocks * sizeof(ZgBlockPos), hipMemcpyDeviceToHost)); }
    for (uint32_t f = 0; f < dev.nframes && nfail; f++) {
      ZgFrameOut& fo = frame_out[f];
      const ZgFrame& fr = bb.frames[f];
      if (!fo.status || fo.good_blocks >= fr.nblocks) continue;
      ZgBlockPos p;
      if (!all_pos.empty()) p = all_pos[fr.first_block + fo.good_blocks];
      else ZG_HIP(hipMemcpy(&p, dev.pos + fr.first_block + fo.good_blocks, sizeof p, hipMemcpyDeviceToHost));
      if (p.out_base < fo.out_size) fo.out_size = p.out_base;
    }
  }
  // A run of ONE frame that is decoded run by run (the FrameDecoder mirror, the thin boundary, the streaming decoder: fs) and failed in
  // sequence execution: the reference's decode buffer also holds what the failing block wrote before it failed — the output of the
  // sequences in front of the one that could not be executed, and that one's literals unless it was the literals that ran out
  // (sequence_execution.rs:6-52) — and hands it out like any other byte (collect / read after the Err). zg_k_partial produces exactly
  // that behind the good blocks' bytes. (Frames of a many-frame submit end with their last good block: include/zgpu.h.)
  if (fs && dev.nframes == 1 && !frame_out.empty()) {
    ZgFrameOut& fo = frame_out[0];
    const ZgFrame& fr = bb.frames[0];
    const uint32_t st = fo.status;
    const bool exe = st == (uint32_t)ZG_EXE_NOT_ENOUGH_LITERALS || st == (uint32_t)ZG_EXE_ZERO_OFFSET || st == (uint32_t)ZG_EXE_OFFSET_TOO_BIG ||
                     st == (uint32_t)ZG_EXE_DICT_TOO_SMALL;
    if (exe && fo.good_blocks < fr.nblocks) {
      const uint32_t b = fr.first_block + fo.good_blocks;
      const ZgBlock& blk = bb.blocks[b];
      ZgBlockSeqOut so;
      memset(&so, 0, sizeof so);
      if (blk.btype == ZG_BT_COMPRESSED && blk.nseq && !blk.host_status)
        ZG_HIP(hipMemcpy(&so, dev.seq_out + b, sizeof so, hipMemcpyDeviceToHost));
      if (so.pad && so.pad <= blk.nseq) {                       // 1 + the sequence that failed (zg_k_seqpost, or zg_k_exact where it decided)
        const uint32_t j = so.pad - 1u;
        const bool lits = st != (uint32_t)ZG_EXE_NOT_ENOUGH_LITERALS;   // (:14-19 come before the push; the offset checks :28-38 behind it)
        // room behind the good blocks: the block of slack size_output reserved — and, when the scan still counted the failing block (an
        // error found while executing: the in-order path, the flatten), that block's own place
        const uint64_t used = kOutFront + fs->base + fs->have + fo.out_size;
        const uint64_t room = fs->d_out.cap > used + 64 ? fs->d_out.cap - used - 64 : 0;
        const uint32_t limit = room > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)room;
        zg_launch_partial(dev, eng->stream_, 0, b, j, lits, limit);
        ZG_HIP(hipStreamSynchronize(eng->stream_));
        uint32_t psize = 0;
        ZG_HIP(hipMemcpy(&psize, dev.totals + 5, 4, hipMemcpyDeviceToHost));
        if (psize != 0xFFFFFFFFu) {                             // (else: more than there is room for — a block beyond 128 KiB rejected by zg_k_seqpost: left out)
          fo.out_size += psize;
          dev.dst_cap += psize;
        }
      }
    }
  }
  hipEvent_t* ev = sc->ev;
  for (int i = 0; i < ZG_T_TOTAL; i++) {
    float m = 0;
    if (hipEventElapsedTime(&m, ev[i], ev[i + 1]) == hipSuccess) ms[i] = m;
  }
  float m = 0;
  if (hipEventElapsedTime(&m, sc->ev_huf[0], sc->ev_huf[1]) == hipSuccess) ms[ZG_T_HUF] = m;   // beside seq + seqpost, not in line ...
  if (dev.flags & ZG_FLAG_LIT_DIRECT) ms[ZG_T_LIT] = ms[ZG_T_LIT] > ms[ZG_T_HUF] ? ms[ZG_T_LIT] - ms[ZG_T_HUF] : 0.f;   // ... or, after the scan, in front of zg_k_lit
  if (hipEventElapsedTime(&m, ev[0], ev[ZG_T_TOTAL]) == hipSuccess) ms[ZG_T_TOTAL] = m;
  synced = true;
  return ZG_OK;
}

// which tables exist once this run has been folded into the frame's state (what commit() will leave in FrameState::carry_mask): the
// streaming decoder prepares the NEXT run while this one is still on the GPU
uint32_t Batch::carry_mask_after() const {
  uint32_t m = carry_mask_in;
  if (bb.frames.empty()) return m;
  const Lineage fl = bb.final_lineage();
  const int32_t sl[3] = {fl.ll, fl.of, fl.ml};
  const int32_t carry = (int32_t)bb.frames[0].carry_slot;
  for (int k = 0; k < 3; k++) if (sl[k] >= 0 && sl[k] != carry) m |= 2u << k;
  if (fl.huf >= 0 && fl.huf != (int32_t)bb.frames[0].carry_huf_slot) m |= 1u;
  return m;
}

int Batch::commit(FrameState* st) {
  // fold a finished streaming run into the frame state: DecoderScratch after decode_block_content
  if (dev.nframes != 1 || frame_out.size() != 1) return ZG_BAD_ARG;
  const ZgFrameOut& fo = frame_out[0];
  st->produced += fo.out_size;
  st->have += fo.out_size;
  st->hist[0] = fo.hist_end[0]; st->hist[1] = fo.hist_end[1]; st->hist[2] = fo.hist_end[2];
  st->counted += fo.counted;
  if (fo.status) return ZG_OK;   // the frame failed: the caller reports it; tables are not needed any more
  const Lineage fl = bb.final_lineage();
  hipStream_t s = eng->stream_;
  const int32_t sl[3] = {fl.ll, fl.of, fl.ml};
  const uint32_t offs[3] = {ZG_FSE_LL_OFF, ZG_FSE_OF_OFF, ZG_FSE_ML_OFF}, nent[3] = {512, 256, 512};
  const int32_t carry = (int32_t)bb.frames[0].carry_slot;
  for (int k = 0; k < 3; k++) {
    if (sl[k] < 0 || sl[k] == carry) continue;   // never set, or still the carried table
    ZG_HIP(hipMemcpyAsync(st->d_fse.as<uint32_t>() + offs[k], dev.fse_arena + (size_t)sl[k] * ZG_FSE_SLOT_U32 + offs[k], nent[k] * 4,
                          hipMemcpyDeviceToDevice, s));
    ZG_HIP(hipMemcpyAsync(&st->logs[k], dev.slot_log + (size_t)sl[k] * 4 + k, 1, hipMemcpyDeviceToHost, s));
    st->carry_mask |= 2u << k;
  }
  if (fl.huf >= 0 && fl.huf != bb.frames[0].carry_huf_slot) {
    ZG_HIP(hipMemcpyAsync(st->d_huf.p, dev.huf_arena + (size_t)fl.huf * ZG_HUF_SLOT_U16, ZG_HUF_SLOT_U16 * 2, hipMemcpyDeviceToDevice, s));
    ZG_HIP(hipMemcpyAsync(&st->huf_maxbits, dev.huf_maxbits + fl.huf, 1, hipMemcpyDeviceToHost, s));
    st->carry_mask |= 1u;
  }
  ZG_HIP(hipStreamSynchronize(s));
  return ZG_OK;
}

int Batch::read_output(uint64_t off, uint8_t* dst, uint64_t n) {
  if (ran && !synced) { const int st = sync(); if (st) return st; }   // (sync() may still have to repeat the sweep)
  if (off + n > dev.dst_cap) return ZG_BAD_ARG;
  if (n) {   // on the engine's second stream (idle once the run is synced): several engines' downloads and uploads overlap
    // (a large download as two halves on two streams: a single copy may get one copy engine, ~28 GB/s, where two copies get one each — what a
    //  process has done on the device before decides, LABNOTES round 6)
    const uint64_t h = n >= (8u << 20) ? (n / 2 + 4095) & ~4095ull : n;
    ZG_HIP(hipMemcpyAsync(dst, dev.dst + off, h, hipMemcpyDeviceToHost, eng->stream2_));
    if (n > h) ZG_HIP(hipMemcpyAsync(dst + h, dev.dst + off + h, n - h, hipMemcpyDeviceToHost, eng->stream5_));
    ZG_HIP(hipStreamSynchronize(eng->stream2_));
    if (n > h) ZG_HIP(hipStreamSynchronize(eng->stream5_));
  }
  return ZG_OK;
}
int Batch::read_output_async(uint64_t off, uint8_t* dst, uint64_t n, hipStream_t s) {
  if (off + n > dev.dst_cap) return ZG_BAD_ARG;
  if (n) ZG_HIP(hipMemcpyAsync(dst, dev.dst + off, n, hipMemcpyDeviceToHost, s));
  return ZG_OK;
}
int Batch::read_block_status(std::vector<uint32_t>* out) {
  out->resize(dev.nblocks);
  if (dev.nblocks) ZG_HIP(hipMemcpy(out->data(), dev.status, (size_t)dev.nblocks * 4, hipMemcpyDeviceToHost));
  return ZG_OK;
}
int Batch::read_literals(uint32_t block, std::vector<uint8_t>* out) {
  if (block >= dev.nblocks) return ZG_BAD_ARG;
  const ZgBlock& b = bb.blocks[block];
  out->clear();
  if (b.btype != ZG_BT_COMPRESSED || b.lit_type < ZG_LT_COMPRESSED) return ZG_OK;
  out->resize(b.regen_size);
  const uint8_t* at = dev.lit_arena + b.lit_base;
  if ((dev.flags & ZG_FLAG_LIT_DIRECT) && b.nseq == 0) {   // zg_k_huf wrote them straight to the block's place in the output
    ZgBlockPos p;
    ZG_HIP(hipMemcpy(&p, dev.pos + block, sizeof p, hipMemcpyDeviceToHost));
    if (b.frame >= frame_out.size() || !p.active) { out->clear(); return ZG_OK; }
    at = dev.dst + frame_out[b.frame].out_base + p.out_base;
  }
  if (b.regen_size) ZG_HIP(hipMemcpy(out->data(), at, b.regen_size, hipMemcpyDeviceToHost));
  return ZG_OK;
}
int Batch::read_sequences(uint32_t block, std::vector<ZgSeq>* seqs, ZgBlockSeqOut* so, ZgBlockPos* pos) {
  if (block >= dev.nblocks) return ZG_BAD_ARG;
  const ZgBlock& b = bb.blocks[block];
  seqs->resize(b.btype == ZG_BT_COMPRESSED ? b.nseq : 0);
  if (!seqs->empty()) ZG_HIP(hipMemcpy(seqs->data(), dev.seq_arena + b.seq_base, seqs->size() * sizeof(ZgSeq), hipMemcpyDeviceToHost));
  if (so) ZG_HIP(hipMemcpy(so, dev.seq_out + block, sizeof *so, hipMemcpyDeviceToHost));
  if (pos) ZG_HIP(hipMemcpy(pos, dev.pos + block, sizeof *pos, hipMemcpyDeviceToHost));
  return ZG_OK;
}
int Batch::read_fse_slot(uint32_t slot, std::vector<uint32_t>* entries, uint8_t logs[4]) {
  if (slot >= dev.nslots) return ZG_BAD_ARG;
  entries->resize(ZG_FSE_SLOT_U32);
  ZG_HIP(hipMemcpy(entries->data(), dev.fse_arena + (size_t)slot * ZG_FSE_SLOT_U32, ZG_FSE_SLOT_U32 * 4, hipMemcpyDeviceToHost));
  ZG_HIP(hipMemcpy(logs, dev.slot_log + (size_t)slot * 4, 4, hipMemcpyDeviceToHost));
  return ZG_OK;
}
int Batch::read_scratch(int what, uint64_t off, void* dst, uint64_t n) {
  const uint8_t* base = what == 0 ? (const uint8_t*)dev.og : what == 1 ? (const uint8_t*)dev.unit_info : nullptr;
  const uint64_t cap = what == 0 ? og_words * 4 : bb.units.size() * sizeof(ZgUnitInfo);
  if (!base || off + n > cap) return ZG_BAD_ARG;
  if (n) ZG_HIP(hipMemcpy(dst, base + off, n, hipMemcpyDeviceToHost));
  return ZG_OK;
}
int Batch::unit_scratch_base(uint32_t unit, uint64_t* base) {
  if (unit >= bb.units.size()) return ZG_BAD_ARG;
  const ZgUnit& u = bb.units[unit];
  ZgBlockPos p;
  ZG_HIP(hipMemcpy(&p, dev.pos + u.first_block, sizeof p, hipMemcpyDeviceToHost));
  if (u.frame >= frame_out.size()) return ZG_BAD_ARG;
  *base = frame_out[u.frame].og_base + p.out_base;
  return ZG_OK;
}
int Batch::read_debug(uint64_t out[1024]) {
  for (int i = 0; i < 1024; i++) out[i] = 0;
  if (dev.dbg) ZG_HIP(hipMemcpy(out, dev.dbg, 8192, hipMemcpyDeviceToHost));
  return ZG_OK;
}
int Batch::read_huf_slot(uint32_t slot, std::vector<uint16_t>* entries, int* max_bits) {
  if (slot >= dev.nhuf_slots) return ZG_BAD_ARG;
  entries->resize(ZG_HUF_SLOT_U16);
  uint8_t mb = 0;
  ZG_HIP(hipMemcpy(entries->data(), dev.huf_arena + (size_t)slot * ZG_HUF_SLOT_U16, ZG_HUF_SLOT_U16 * 2, hipMemcpyDeviceToHost));
  ZG_HIP(hipMemcpy(&mb, dev.huf_maxbits + slot, 1, hipMemcpyDeviceToHost));
  *max_bits = mb;
  return ZG_OK;
}

}  // namespace zg
