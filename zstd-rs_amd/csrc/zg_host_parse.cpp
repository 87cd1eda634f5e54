// zg_host_parse.cpp — see zg_host_parse.h.
#include "zg_host_parse.h"
#include <math.h>
#include <string.h>

namespace zg {

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

int read_frame_header(const uint8_t* src, size_t len, FrameHeader* h, size_t* consumed, uint32_t* skip_magic, uint32_t* skip_len) {
  // ruzstd decoding/frame.rs:6-85
  *consumed = 0;
  if (len < 4) return ZG_HEADER_READ;
  uint32_t magic = rd32(src);
  size_t p = 4;
  if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {  // skippable frame, reported like the reference does (:15-23)
    if (len < 8) return ZG_HEADER_READ;
    if (skip_magic) *skip_magic = magic;
    if (skip_len) *skip_len = rd32(src + 4);
    *consumed = 8;
    return ZG_SKIP_FRAME;
  }
  if (magic != kMagic) return ZG_BAD_MAGIC;
  if (len < p + 1) return ZG_HEADER_READ;
  *h = FrameHeader();
  h->descriptor = src[p++];
  if (!h->single_segment()) {
    if (len < p + 1) return ZG_HEADER_READ;
    h->window_descriptor = src[p++];
  }
  static const unsigned kDidLen[4] = {0, 1, 2, 4};  // frame.rs:231-239
  unsigned dl = kDidLen[h->descriptor & 3];
  if (dl) {
    if (len < p + dl) return ZG_HEADER_READ;
    uint32_t id = 0;
    for (unsigned i = 0; i < dl; i++) id += (uint32_t)src[p + i] << (8 * i);
    p += dl;
    if (id != 0) { h->has_dict_id = true; h->dict_id = id; }  // dict id 0 means "none" (:60-62)
  }
  unsigned fl;  // frame.rs:212-226
  switch (h->descriptor >> 6) {
    case 0: fl = h->single_segment() ? 1 : 0; break;
    case 1: fl = 2; break;
    case 2: fl = 4; break;
    default: fl = 8; break;
  }
  if (fl) {
    if (len < p + fl) return ZG_HEADER_READ;
    uint64_t fcs = 0;
    for (unsigned i = 0; i < fl; i++) fcs += (uint64_t)src[p + i] << (8 * i);
    if (fl == 2) fcs += 256;  // :78-80
    h->frame_content_size = fcs;
    p += fl;
  }
  h->header_size = (uint32_t)p;
  *consumed = p;
  return ZG_OK;
}

int frame_window_size(const FrameHeader& h, uint64_t* out) {
  // frame.rs:116-139: single-segment frames use the content size, with no minimum
  if (h.single_segment()) { *out = h.frame_content_size; return ZG_OK; }
  unsigned exp = h.window_descriptor >> 3, mant = h.window_descriptor & 7;
  uint64_t base = 1ull << (10 + exp), w = base + (base / 8) * mant;
  if (w < kMinWindow) return ZG_WINDOW_TOO_SMALL;
  if (w >= kMaxWindow) return ZG_WINDOW_TOO_BIG_SPEC;
  *out = w;
  return ZG_OK;
}

int read_block_header(const uint8_t* p, BlockHeader* h) {
  // block_decoder.rs:201-247, :249-283
  h->last = p[0] & 1;
  h->type = (p[0] >> 1) & 3;
  if (h->type == 3) return ZG_RESERVED_BLOCK;
  uint32_t size = (uint32_t)(p[0] >> 3) | ((uint32_t)p[1] << 5) | ((uint32_t)p[2] << 13);
  if (size > kMaxBlockSize) return ZG_BLOCK_SIZE_TOO_LARGE;
  h->decompressed_size = (h->type == ZG_BT_COMPRESSED) ? 0 : size;
  h->content_size = (h->type == ZG_BT_RLE) ? 1 : size;
  return ZG_OK;
}

uint32_t BatchBuilder::begin_frame(uint64_t window_size, const uint32_t hist[3], uint32_t carry_mask) {
  ZgFrame f;
  memset(&f, 0, sizeof f);
  f.first_block = (uint32_t)blocks.size();
  f.nblocks = 0;
  f.window_size = window_size;
  f.hist_init[0] = hist[0]; f.hist_init[1] = hist[1]; f.hist_init[2] = hist[2];
  f.carry_huf_slot = ZG_REF_UNINIT;
  frames.push_back(f);
  cur_ = Lineage();
  if (carry_mask & 1u) cur_.huf = kCarryHuf;
  if (carry_mask & 2u) cur_.ll = kCarry;
  if (carry_mask & 4u) cur_.of = kCarry;
  if (carry_mask & 8u) cur_.ml = kCarry;
  frame_failed_ = false;
  return (uint32_t)frames.size() - 1;
}

void BatchBuilder::fail_frame(int status) {
  // a zero-size raw block carrying the error stops the frame at this point
  ZgBlock b;
  memset(&b, 0, sizeof b);
  b.btype = ZG_BT_RAW;
  b.frame = (uint32_t)frames.size() - 1;
  b.huf_slot = b.ll_slot = b.of_slot = b.ml_slot = ZG_REF_UNINIT;
  b.host_status = (uint32_t)status;
  blocks.push_back(b);
  frames.back().nblocks++;
  frame_failed_ = true;
}

int BatchBuilder::add_block(const BlockHeader& bh, const uint8_t* body, uint64_t src_off) {
  ZgBlock b;
  memset(&b, 0, sizeof b);
  b.src_off = src_off;
  b.src_len = bh.content_size;
  b.btype = bh.type;
  b.frame = (uint32_t)frames.size() - 1;
  b.huf_slot = b.ll_slot = b.of_slot = b.ml_slot = ZG_REF_UNINIT;
  const uint32_t bidx = (uint32_t)blocks.size();
  int st = ZG_OK, seq_st = ZG_OK;   // seq_st: found in the sequences section header, i.e. behind the literals (ZgBlock::seq_host_status)
  if (bh.type != ZG_BT_COMPRESSED) {
    b.regen_size = bh.decompressed_size;
    out_bound += b.regen_size;
  } else {
    // decompress_block (block_decoder.rs:97-197): header of the literals section
    const uint32_t n = bh.content_size;
    out_bound += kMaxBlockSize;
    do {
      if (n == 0) { st = ZG_LITERALS_HEADER; break; }               // literals_section.rs:119 (no bits to read)
      b.lit_type = body[0] & 3;
      const unsigned sf = (body[0] >> 2) & 3;
      unsigned need;
      if (b.lit_type == ZG_LT_RAW || b.lit_type == ZG_LT_RLE) need = (sf == 0 || sf == 2) ? 1 : (sf == 1 ? 2 : 3);
      else need = sf <= 1 ? 3 : (sf == 2 ? 4 : 5);
      if (n < need) { st = ZG_LITERALS_HEADER; break; }             // NotEnoughBytes :124-129
      uint32_t upper;
      if (b.lit_type == ZG_LT_RAW || b.lit_type == ZG_LT_RLE) {     // :132-159
        if (sf == 0 || sf == 2) b.regen_size = body[0] >> 3;
        else if (sf == 1) b.regen_size = (body[0] >> 4) + ((uint32_t)body[1] << 4);
        else b.regen_size = (body[0] >> 4) + ((uint32_t)body[1] << 4) + ((uint32_t)body[2] << 12);
        upper = b.lit_type == ZG_LT_RLE ? 1 : b.regen_size;          // block_decoder.rs:120-127
      } else {                                                      // :161-221
        b.nstreams = sf == 0 ? 1 : 4;
        if (sf <= 1) {
          b.regen_size = (body[0] >> 4) + (((uint32_t)body[1] & 0x3f) << 4);
          b.lit_comp_size = (body[1] >> 6) + ((uint32_t)body[2] << 2);
        } else if (sf == 2) {
          b.regen_size = (body[0] >> 4) + ((uint32_t)body[1] << 4) + (((uint32_t)body[2] & 0x3) << 12);
          b.lit_comp_size = (body[2] >> 2) + ((uint32_t)body[3] << 6);
        } else {
          b.regen_size = (body[0] >> 4) + ((uint32_t)body[1] << 4) + (((uint32_t)body[2] & 0x3F) << 12);
          b.lit_comp_size = (body[2] >> 6) + ((uint32_t)body[3] << 2) + ((uint32_t)body[4] << 10);
        }
        upper = b.lit_comp_size;
      }
      b.lit_off = need;
      if (n - need < upper) { st = ZG_MALFORMED_SECTION_HEADER; break; }  // block_decoder.rs:129-134
      if (b.lit_type == ZG_LT_COMPRESSED) {
        cur_.huf = (int32_t)nhuf_slots++;                            // this block defines the Huffman table from now on
      } else if (b.lit_type == ZG_LT_TREELESS && cur_.huf == ZG_REF_UNINIT) {
        st = ZG_LIT_UNINIT_HUF; break;                               // literals_section_decoder.rs:60-63
      }
      if (b.lit_type >= ZG_LT_COMPRESSED) b.huf_slot = cur_.huf;
      // sequences section header (sequence_section.rs:108-167)
      const uint8_t* s = body + need + upper;
      const uint32_t rem = n - need - upper;
      if (rem == 0) { seq_st = ZG_SEQUENCES_HEADER; break; }
      unsigned shl;
      bool has_modes = false;
      if (s[0] == 0) { b.nseq = 0; shl = 1; }
      else if (s[0] < 128) {
        if (rem < 2) { seq_st = ZG_SEQUENCES_HEADER; break; }
        b.nseq = s[0]; b.seq_modes = s[1]; has_modes = true; shl = 2;
      } else if (s[0] < 255) {
        if (rem < 2) { seq_st = ZG_SEQUENCES_HEADER; break; }
        b.nseq = (((uint32_t)s[0] - 128) << 8) + s[1]; shl = 2;
        if (b.nseq != 0) {
          if (rem < 3) { seq_st = ZG_SEQUENCES_HEADER; break; }
          b.seq_modes = s[2]; has_modes = true; shl = 3;
        }
      } else {
        if (rem < 4) { seq_st = ZG_SEQUENCES_HEADER; break; }
        b.nseq = (uint32_t)s[1] + ((uint32_t)s[2] << 8) + 0x7F00; b.seq_modes = s[3]; has_modes = true; shl = 4;
      }
      b.seq_off = need + upper + shl;
      if (b.nseq == 0) {
        if (rem != shl) { seq_st = ZG_SEQ_EXTRA_BITS; break; }           // block_decoder.rs:184-190
      } else {
        if (!has_modes) { seq_st = ZG_SEQ_MISSING_MODE; break; }
        // maybe_update_fse_tables lineage (sequence_section_decoder.rs:294-410): LL, OF, ML
        int32_t* cur[3] = {&cur_.ll, &cur_.of, &cur_.ml};
        const int modes[3] = {b.seq_modes >> 6, (b.seq_modes >> 4) & 3, (b.seq_modes >> 2) & 3};
        for (int k = 0; k < 3; k++) {
          if (modes[k] == ZG_MODE_PREDEFINED) *cur[k] = kPredef;
          else if (modes[k] == ZG_MODE_RLE || modes[k] == ZG_MODE_FSE) *cur[k] = (int32_t)bidx;
        }
        b.ll_slot = cur_.ll; b.of_slot = cur_.of; b.ml_slot = cur_.ml;
        // a Repeat mode with nothing to repeat fails in FSEDecoder::init_state (fse_decoder.rs:33-35); the device
        // reports it (ZG_FSE_UNINIT) after the table descriptions of the other streams were checked, like the reference
      }
    } while (0);
  }
  if (frame_failed_ && !st) st = ZG_INTERNAL;
  if (seq_st) { b.nseq = 0; b.seq_modes = 0; b.seq_off = 0; }   // the block's literals are decoded, its sequences never looked at
  b.host_status = (uint32_t)st;
  b.seq_host_status = st ? 0u : (uint32_t)seq_st;
  if (!st && bh.type == ZG_BT_COMPRESSED) {
    if (b.lit_type >= ZG_LT_COMPRESSED) { b.lit_base = lit_bytes; lit_bytes += b.regen_size; }
    if (b.nseq) { b.seq_base = seq_count; seq_count += b.nseq; }
  }
  blocks.push_back(b);
  frames.back().nblocks++;
  if (st || seq_st) frame_failed_ = true;
  return st ? st : seq_st;
}

void BatchBuilder::finish() {
  const uint32_t nb = (uint32_t)blocks.size();
  const uint32_t block_huf_slots = nhuf_slots;
  nhuf_slots += (uint32_t)frames.size();
  for (uint32_t f = 0; f < frames.size(); f++) {
    frames[f].carry_slot = nb + 1 + f;
    frames[f].carry_huf_slot = (int32_t)(block_huf_slots + f);
  }
  // lineage of the last frame after its last block (streaming: what the next submit of this frame starts from)
  final_ = cur_;
  if (!frames.empty()) {
    const uint32_t lf = (uint32_t)frames.size() - 1;
    int32_t* fl[3] = {&final_.ll, &final_.of, &final_.ml};
    for (int k = 0; k < 3; k++) {
      if (*fl[k] == kPredef) *fl[k] = (int32_t)nb;
      else if (*fl[k] == kCarry) *fl[k] = (int32_t)frames[lf].carry_slot;
    }
    if (final_.huf == kCarryHuf) final_.huf = frames[lf].carry_huf_slot;
  }
  seq_blocks.clear(); huf_items.clear(); huf_groups.clear(); units.clear(); step_units.clear(); steps.clear(); unit_list.clear();
  ramped = false;
  // blocks per unit: zg_k_flatten runs flat_slots workgroups at once, one per unit, so aim at ~flat_slots units over the whole
  // submit (more blocks per unit = fewer sweep steps, but less parallelism in the flatten pass). What costs there are the
  // blocks that have sequences: a literal-heavy frame gets smaller units in proportion, so that few literal-only blocks share a
  // unit — scratch words, a sweep step — with a block that needs them.
  uint32_t ub = unit_blocks;
  if (ub == 0) {
    const uint32_t slots = flat_slots ? flat_slots : 1;
    ub = (nb + slots - 1) / slots;
    // (round 5) ... but no larger than the submit needs. The flatten lasts as long as its largest unit (~137 us per block of text), the sweep
    // as many steps as the longest frame has units (~10 us each): units of sqrt(0.0745 x blocks of the longest frame) blocks balance the two,
    // and only the blocks that HAVE sequences (in frames that are swept at all) need a workgroup. Twelve Silesia-sized frames, a third of
    // them literal-heavy: 1617 blocks / 256 = 7 blocks per unit left half of the CUs without a unit; 5 (the balance for its 316-block
    // text frame, and still one round of workgroups for its 935 blocks with sequences): flatten 0.97 -> 0.65 ms, 64 -> 70 GB/s. One long
    // frame (7630 blocks -> 30) and submits of many frames (-> 256) are bound by the workgroups the device holds, as before.
    uint64_t nbs_all = 0, nbs_longest = 0;
    for (const ZgFrame& fr : frames) {
      uint64_t nsq = 0, nbs = 0;
      for (uint32_t i = 0; i < fr.nblocks; i++) { const ZgBlock& bk = blocks[fr.first_block + i]; if (bk.btype == ZG_BT_COMPRESSED && bk.nseq) { nbs++; nsq += bk.nseq; } }
      if (sparse_max && nsq <= sparse_max && (uint64_t)fr.nblocks * sparse_per_block >= nsq) continue;   // (zg_k_sparse finishes it: no units to speak of)
      nbs_all += nbs;
      if (nbs > nbs_longest) nbs_longest = nbs;
    }
    const uint32_t bal = (uint32_t)(sqrt(0.0745 * (double)nbs_longest) + 0.5);
    const uint32_t need = (uint32_t)((nbs_all + slots - 1) / slots);
    if (bal < ub) ub = bal > need ? bal : (need < ub ? need : ub);
    if (ub < 4) ub = 4;
    if (ub > 256) ub = 256;   // unit-relative positions stay far below 2^30
  }
  unit_blocks_used = ub;
  uint32_t max_units = 0;
  for (uint32_t f = 0; f < frames.size(); f++) {
    ZgFrame& fr = frames[f];
    fr.first_unit = (uint32_t)units.size();
    uint32_t ubf = ub, nbs = 0;
    uint64_t nsq = 0;
    for (uint32_t i = 0; i < fr.nblocks; i++) {
      const ZgBlock& bk = blocks[fr.first_block + i];
      if (bk.btype == ZG_BT_COMPRESSED && bk.nseq) { nbs++; nsq += bk.nseq; }
    }
    // few sequences in a frame of many blocks: one wave copies its matches in order faster than a chain of sweep launches runs
    fr.sparse = (sparse_max && nsq <= sparse_max && (uint64_t)fr.nblocks * sparse_per_block >= nsq) ? 1u : 0u;
    fr.seq_first = fr.seq_count = 0; fr.pad2 = 0;
    if (unit_blocks == 0 && fr.nblocks) {
      ubf = (uint32_t)(((uint64_t)ub * nbs + fr.nblocks - 1) / fr.nblocks);
      if (ubf < 4) ubf = 4;
    }
    // One long frame alone in the submit: all its units are flattened side by side and swept one after the other. With units
    // that grow along the frame the flatten finishes them in frame order, and the sweep of unit k (a chain of short launches
    // that leaves most of the chip idle) runs while the flatten is still busy with the larger units behind it.
    const bool ramp = ramp_percent && unit_blocks == 0 && frames.size() == 1 && fr.nblocks >= 64 * ubf && !fr.sparse;
    const uint32_t nun = (fr.nblocks + ubf - 1) / ubf;
    ramped = ramped || ramp;
    // A frame whose first unit is resolved to bytes by the flatten itself (direct unit, below) and that has more units behind it:
    // the direct body moves ~30 % more bytes per unit of time than the pointer-mode body (no scratch words to write and gather,
    // measured in round 4: 1.55 against 1.19 MB/ms per workgroup), and the units of a submit are flattened side by side — so the
    // first unit gets ~1.3 shares of the frame's blocks and the others one each: they finish together, and less of the frame goes
    // through the scratch and the sweep.
    // (round 5: only where units are large — with units of a few blocks the share rounds to 1.4 - 1.5 and the direct unit becomes the longest
    //  of the submit: twelve Silesia-sized frames in units of 5 blocks, flatten 0.82 ms with the share, 0.65 ms without)
    const bool direct_frame = unit_blocks == 0 && !ramp && direct_units && !fr.fixed_base && !fr.sparse && nun >= 2 && nun <= direct_max_units && ubf >= 16;
    uint32_t first_take = 0, rest_take = 0;
    if (direct_frame) {
      first_take = (uint32_t)(((uint64_t)fr.nblocks * direct_share10 + (10ull * (nun - 1) + direct_share10) - 1) / (10ull * (nun - 1) + direct_share10));
      if (first_take > 384) first_take = 384;
      if (first_take < ubf) first_take = ubf;
      if (first_take >= fr.nblocks) first_take = fr.nblocks;
      rest_take = fr.nblocks > first_take ? (fr.nblocks - first_take + (nun - 1) - 1) / (nun - 1) : 0;
      if (rest_take > 256) rest_take = 256;   // (more units than planned, then)
    }
    uint32_t done_blocks = 0, ui = 0;
    for (uint32_t i = 0; i < fr.nblocks; ui++) {
      uint32_t take = direct_frame ? (i == 0 ? first_take : (rest_take ? rest_take : ubf)) : ubf;
      if (ramp) {
        // cumulative target: blocks in front of unit ui + 1 = integral of the linear ramp
        const double x = (double)(ui + 1) / nun, r = ramp_percent / 100.0;
        const double cum = fr.nblocks * (x * (1.0 - r) + r * x * x);
        uint32_t upto = ui + 1 == nun ? fr.nblocks : (uint32_t)(cum + 0.5);
        if (upto <= done_blocks) upto = done_blocks + 1;
        if (upto > fr.nblocks) upto = fr.nblocks;
        take = upto - done_blocks;
        if (take > 256) take = 256;
      }
      ZgUnit u;
      u.frame = f; u.first_block = fr.first_block + i; u.desc = 0xFFFFFFFFu; u.pad = 0;
      u.nblocks = fr.nblocks - i < take ? fr.nblocks - i : take; u.noseq = 1;
      for (uint32_t k = 0; k < u.nblocks; k++) {
        const ZgBlock& bk = blocks[u.first_block + k];
        if (bk.btype == ZG_BT_COMPRESSED && bk.nseq) u.noseq = 0;
      }
      // the first unit of a frame that starts from nothing (no dictionary, no earlier submit: the engine marks those frames
      // fixed_base before finish()) copies from nothing outside itself: the flatten resolves it to bytes right away (direct unit, zg_flat4.h)
      // ... when that saves a noticeable share of the frame's sweep steps: a direct unit takes ~10 % longer to flatten than a
      // pointer-mode one, and all units of a submit are flattened side by side (a frame of hundreds of units gains nothing)
      if (i == 0 && !u.noseq && direct_units && !fr.fixed_base && !fr.sparse && (fr.nblocks + ubf - 1) / ubf <= direct_max_units) u.noseq = ZG_UNIT_DIRECT;
      units.push_back(u);
      i += u.nblocks; done_blocks += u.nblocks;
    }
    fr.nunits = (uint32_t)units.size() - fr.first_unit;
    if (fr.nunits > max_units) max_units = fr.nunits;
  }
  // launch order of the flatten: the pointer-mode body and the direct body are kernels of their own (zg_k_flatten, zg_k_flatten4)
  unit_list.clear(); n_direct = 0;
  for (uint32_t u = 0; u < units.size(); u++) if (!(units[u].noseq & ZG_UNIT_DIRECT)) unit_list.push_back(u);
  for (uint32_t u = 0; u < units.size(); u++) if (units[u].noseq & ZG_UNIT_DIRECT) { unit_list.push_back(u); n_direct++; }
  // sweep steps: step s takes unit s of every frame that has one (frames are independent; units of a frame go in order).
  // Units without sequences need no step, and a step nobody needs is not launched (literal-heavy frames: most of them).
  for (uint32_t s = 0; s < max_units; s++) {
    ZgStepRange r;
    r.list_off = (uint32_t)step_units.size(); r.nunits = 0; r.max_blocks = 0;
    for (uint32_t f = 0; f < frames.size(); f++) {
      if (frames[f].nunits <= s) continue;
      const uint32_t u = frames[f].first_unit + s;
      if (units[u].noseq || frames[f].sparse) continue;
      units[u].desc = (uint32_t)step_units.size();
      step_units.push_back(u);
      r.nunits++;
      if (units[u].nblocks > r.max_blocks) r.max_blocks = units[u].nblocks;
    }
    if (r.nunits) steps.push_back(r);
  }
  // literals after the scan? What it saves: the copy of the literals of blocks without sequences (read + write, ~2.5 bytes per
  // nanosecond); what it costs: the literals chain no longer hides beside the sequences chain, which lasts as long as its longest
  // block (~0.12 us per sequence) times the rounds the blocks with sequences need.
  {
    uint64_t huf_noseq = 0, max_nseq = 0, nsb = 0;
    for (const ZgBlock& b : blocks) {
      if (b.btype != ZG_BT_COMPRESSED || b.host_status) continue;
      if (b.nseq) { nsb++; if (b.nseq > max_nseq) max_nseq = b.nseq; }
      else if (b.lit_type >= ZG_LT_COMPRESSED) huf_noseq += b.regen_size;
    }
    // (more blocks than one round holds: zg_k_seq runs in its packed-entry form, half as many chains again per CU. Decided HERE, once — the
    //  launch, Batch::run, takes the same flag: ADVICE r5)
    seq_packed = seq_packed_force < 0 ? (chain_slots && nsb > chain_slots) : seq_packed_force != 0;
    const uint64_t slots = chain_slots ? (seq_packed ? chain_slots + chain_slots / 2 : chain_slots) : 1;
    const uint64_t rounds = (nsb + slots - 1) / slots;
    const double gain_us = (double)huf_noseq / 2.5e6, loss_us = 0.12 * (double)max_nseq * (double)(rounds ? rounds : 1);
    lit_direct = lit_direct_allowed && gain_us > 1.5 * loss_us + 20.0;
  }
  for (uint32_t i = 0; i < nb; i++) {
    ZgBlock& b = blocks[i];
    if (b.btype != ZG_BT_COMPRESSED || b.host_status) continue;
    int32_t* sl[3] = {&b.ll_slot, &b.of_slot, &b.ml_slot};
    for (int k = 0; k < 3; k++) {
      if (*sl[k] == kPredef) *sl[k] = (int32_t)nb;
      else if (*sl[k] == kCarry) *sl[k] = (int32_t)frames[b.frame].carry_slot;
    }
    if (b.huf_slot == kCarryHuf) b.huf_slot = frames[b.frame].carry_huf_slot;
    if (b.nseq) {
      ZgFrame& fr = frames[b.frame];
      if (!fr.seq_count) fr.seq_first = (uint32_t)seq_blocks.size();
      fr.seq_count++;
      b.seq_idx = (uint32_t)seq_blocks.size(); seq_blocks.push_back(i);
    }
    if (b.lit_type >= ZG_LT_COMPRESSED) {
      for (uint32_t k = 0; k < b.nstreams; k++) {
        if (huf_groups.empty() || huf_groups.back().slot != b.huf_slot || huf_groups.back().nitems >= ZG_HUF_GROUP) {
          ZgHufGroup g;
          g.slot = b.huf_slot; g.first_item = (uint32_t)huf_items.size(); g.nitems = 0; g.pad = 0;
          huf_groups.push_back(g);
        }
        huf_items.push_back((i << 2) | k);
        huf_groups.back().nitems++;
      }
    }
  }
}

}  // namespace zg
