// zg_emu.cpp — TEST-ONLY host harness: runs the engine's host parser (zg_host_parse.cpp), the table routines and the
// symbolic offset history of zstd-rs_amd/csrc/zg_dev.h (the code the kernels call) plus a serial model of the
// wave-cooperative decoders (zg_emu_serial.h) on the CPU, in the order of the HIP pipeline, so that this logic
// can be checked against the oracle and the golden fixtures without a GPU. It is NOT a fallback of the product:
// libzgpu.so neither contains nor loads it, and it lives under tests/.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "zg_emu_serial.h"
#include "../../zstd-rs_amd/csrc/zg_host_parse.h"

using namespace zg;

namespace zg {
int parse_frames(const uint8_t* src, size_t len, uint64_t max_window, BatchBuilder* bb, std::vector<struct FrameInfoLite>* info);
}

#include "zg_emu_batch.h"

static void set_status(EmuBatch& e, uint32_t b, int st) { if (st && !e.status[b]) e.status[b] = (uint32_t)st; }

// decode_all-style walk, same as zg::parse_frames in zg_engine.cpp (which needs HIP headers, so restated here)
static int walk(const uint8_t* src, size_t len, uint64_t max_window, BatchBuilder* bb) {
  size_t p = 0;
  static const uint32_t kHist[3] = {1, 4, 8};
  while (p < len) {
    FrameHeader h; size_t c; uint32_t sm = 0, sl = 0;
    int st = read_frame_header(src + p, len - p, &h, &c, &sm, &sl);
    if (st == ZG_SKIP_FRAME) { p += c; if ((size_t)sl > len - p) return ZG_FAILED_SKIP_FRAME; p += sl; continue; }
    if (st) return st;
    uint64_t w;
    if ((st = frame_window_size(h, &w))) return st;
    if (w > max_window) return ZG_WINDOW_SIZE_TOO_BIG;
    if (h.has_dict_id) return ZG_DICT_NOT_PROVIDED;
    p += c;
    bb->begin_frame(w, kHist, 0);
    for (;;) {
      if (len - p < 3) return ZG_FAILED_READ_BLOCK_HEADER;
      BlockHeader bh;
      if ((st = read_block_header(src + p, &bh))) return st;
      p += 3;
      if (len - p < bh.content_size) return ZG_FAILED_READ_BLOCK_BODY;
      st = bb->add_block(bh, src + p, p);
      p += bh.content_size;
      if (st) return st;
      if (bh.last) { if (h.content_checksum()) { if (len - p < 4) return ZG_FAILED_READ_CHECKSUM; p += 4; } break; }
    }
  }
  return ZG_OK;
}

static void k_tables(EmuBatch& e) {
  const uint32_t nb = (uint32_t)e.bb.blocks.size();
  int16_t probs[256]; uint16_t counter[256];
  {
    uint32_t* slot = e.fse.data() + (size_t)nb * ZG_FSE_SLOT_U32;
    for (int i = 0; i < 36; i++) probs[i] = ZG_LL_DEFAULT[i];
    zg_fse_build(probs, 36, 6, ZG_KIND_LL, slot + ZG_FSE_LL_OFF, counter);
    for (int i = 0; i < 29; i++) probs[i] = ZG_OF_DEFAULT[i];
    zg_fse_build(probs, 29, 5, ZG_KIND_OF, slot + ZG_FSE_OF_OFF, counter);
    for (int i = 0; i < 53; i++) probs[i] = ZG_ML_DEFAULT[i];
    zg_fse_build(probs, 53, 6, ZG_KIND_ML, slot + ZG_FSE_ML_OFF, counter);
    uint8_t* lg = e.slot_log.data() + (size_t)nb * 4; lg[0] = 6; lg[1] = 5; lg[2] = 6;
  }
  for (uint32_t b = 0; b < nb; b++) {
    const ZgBlock blk = e.bb.blocks[b];
    if (blk.host_status || blk.btype != ZG_BT_COMPRESSED) continue;
    const uint8_t* body = e.src + blk.src_off;
    ZgBlockAux aux; memset(&aux, 0, sizeof aux); aux.seq_bits_off = blk.seq_off;
    int st = ZG_OK;
    if (blk.lit_type == ZG_LT_COMPRESSED) {
      uint8_t weights[264]; uint32_t fsew[64]; int nw = 0, mb = 0; uint32_t used = 0;
      st = zg_huf_read_weights(body + blk.lit_off, blk.lit_comp_size, weights, &nw, &used, fsew, probs, counter);
      if (!st) st = zg_huf_build(weights, nw, e.huf.data() + (size_t)blk.huf_slot * ZG_HUF_SLOT_U16, &mb);
      if (!st) { e.hufmax[blk.huf_slot] = (uint8_t)mb; aux.huf_desc_bytes = used; }
    }
    int fst = ZG_OK;                         // the sequences' table descriptions: their verdict is kept apart (the literal streams' comes first)
    if (!st && blk.nseq > 0) {
      int& st = fst;
      const uint8_t* p = body + blk.seq_off; uint32_t rem = blk.src_len - blk.seq_off;
      uint32_t* slot = e.fse.data() + (size_t)b * ZG_FSE_SLOT_U32;
      const int kinds[3] = {ZG_KIND_LL, ZG_KIND_OF, ZG_KIND_ML};
      const int modes[3] = {blk.seq_modes >> 6, (blk.seq_modes >> 4) & 3, (blk.seq_modes >> 2) & 3};
      const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
      const uint32_t offs[3] = {ZG_FSE_LL_OFF, ZG_FSE_OF_OFF, ZG_FSE_ML_OFF};
      for (int k = 0; k < 3 && !st; k++) {
        if (modes[k] == ZG_MODE_FSE) {
          int np, al; uint32_t used;
          st = zg_fse_read_probs(p, rem, max_log[k], max_sym[k], probs, &np, &al, &used);
          if (!st) st = zg_fse_build(probs, np, al, kinds[k], slot + offs[k], counter);
          if (!st) { aux.log[k] = (uint8_t)al; p += used; rem -= used; }
        } else if (modes[k] == ZG_MODE_RLE) {
          if (rem == 0) st = ZG_SEQ_RLE_BYTE;
          else if (p[0] > max_sym[k]) st = ZG_SEQ_RLE_BYTE;
          else { slot[offs[k]] = zg_fse_pack(kinds[k], 0, 0, p[0]); aux.log[k] = 0; p += 1; rem -= 1; }
        }
      }
      aux.seq_bits_off = (uint32_t)(p - body);
      uint8_t* lg = e.slot_log.data() + (size_t)b * 4; lg[0] = aux.log[0]; lg[1] = aux.log[1]; lg[2] = aux.log[2];
    }
    e.aux[b] = aux;
    set_status(e, b, st);
    e.fse_status[b] = (uint32_t)fst;
  }
}

static void k_huf(EmuBatch& e) {
  for (const ZgHufGroup& grp : e.bb.huf_groups) {
    unsigned max_bits = grp.slot >= 0 ? e.hufmax[grp.slot] : 0;
    if (max_bits > 11) max_bits = 0;
    const uint16_t* tab = e.huf.data() + (size_t)(grp.slot >= 0 ? grp.slot : 0) * ZG_HUF_SLOT_U16;
    for (uint32_t t = 0; t < grp.nitems; t++) {
      uint32_t item = e.bb.huf_items[grp.first_item + t], b = item >> 2, k = item & 3;
      const ZgBlock blk = e.bb.blocks[b];
      if (max_bits == 0) { set_status(e, b, ZG_LIT_UNINIT_HUF); continue; }
      if (e.status[b]) continue;
      uint32_t desc = blk.lit_type == ZG_LT_COMPRESSED ? e.aux[b].huf_desc_bytes : 0;
      if (desc > blk.lit_comp_size) { set_status(e, b, ZG_INTERNAL); continue; }
      const uint8_t* pay = e.src + blk.src_off + blk.lit_off + desc;
      uint32_t total = blk.lit_comp_size - desc, regen = blk.regen_size;
      uint8_t* lit = e.lit.data() + blk.lit_base;
      const uint8_t* sp; uint32_t slen, doff, cap;
      if (blk.nstreams == 4) {
        if (total < 6) { set_status(e, b, ZG_LIT_MISSING_JUMP); continue; }
        uint32_t j1 = zg_ld16(pay), j2 = j1 + zg_ld16(pay + 2), j3 = j2 + zg_ld16(pay + 4), rest = total - 6;
        if (rest < j3) { set_status(e, b, ZG_LIT_MISSING_BYTES); continue; }
        uint32_t start = k == 0 ? 0 : k == 1 ? j1 : k == 2 ? j2 : j3, end = k == 0 ? j1 : k == 1 ? j2 : k == 2 ? j3 : rest;
        sp = pay + 6 + start; slen = end - start;
        uint32_t seg = (regen + 3) / 4;
        doff = k * seg; if (doff > regen) doff = regen;
        cap = k < 3 ? seg : regen - doff;
        if (cap > regen - doff) cap = regen - doff;
      } else { sp = pay; slen = total; doff = 0; cap = regen; }
      uint32_t count = 0; int32_t endbits = 0;
      int st = zg_huf_decode_stream(sp, slen, tab, max_bits, lit + doff, cap, &count, &endbits);
      if (!st && blk.nstreams == 4 && count <= cap && endbits != -(int32_t)max_bits) st = ZG_LIT_BITSTREAM_MISMATCH;   // (count > cap: the routine stopped early; the pass below looks at the stream's end)
      if (!st && count != cap) st = ZG_LIT_COUNT_MISMATCH;
      set_status(e, b, st);
    }
  }
  // zg_k_huf_uneven: four streams that end on their last bit and add up to the section's size, split differently from the
  // format's (regen + 3) / 4: valid for the reference (it compares the total only, literals_section_decoder.rs:150-155)
  for (uint32_t b = 0; b < e.bb.blocks.size(); b++) {
    const ZgBlock blk = e.bb.blocks[b];
    if (blk.nstreams != 4 || blk.lit_type < ZG_LT_COMPRESSED || blk.huf_slot < 0 || e.status[b] != (uint32_t)ZG_LIT_COUNT_MISMATCH) continue;
    unsigned max_bits = e.hufmax[blk.huf_slot];
    if (max_bits == 0 || max_bits > 11) continue;
    const uint16_t* tab = e.huf.data() + (size_t)blk.huf_slot * ZG_HUF_SLOT_U16;
    uint32_t desc = blk.lit_type == ZG_LT_COMPRESSED ? e.aux[b].huf_desc_bytes : 0;
    const uint8_t* pay = e.src + blk.src_off + blk.lit_off + desc;
    uint32_t total = blk.lit_comp_size - desc, regen = blk.regen_size;
    uint32_t j[5] = {0, zg_ld16(pay), 0, 0, total - 6};
    j[2] = j[1] + zg_ld16(pay + 2); j[3] = j[2] + zg_ld16(pay + 4);
    std::vector<uint8_t> tmp[4];
    uint64_t sum = 0;
    bool clean = true;
    int bad_st = 0;
    for (int k = 0; k < 4 && clean; k++) {
      // (a stream may hold more symbols than the whole section regenerates — the reference decodes it to its end all the same and
      //  only then compares the total: room for one symbol per bit)
      const uint32_t room = 8u * (j[k + 1] - j[k]) + 16u;
      tmp[k].assign(room + 16, 0);
      uint32_t count = 0; int32_t endbits = 0;
      int st = zg_huf_decode_stream(pay + 6 + j[k], j[k + 1] - j[k], tab, max_bits, tmp[k].data(), room, &count, &endbits);
      if (st == ZG_LIT_EXTRA_PADDING) { bad_st = st; clean = false; }             // the first stream, in order, that fails decides
      else if (st || endbits != -(int32_t)max_bits || count > room) clean = false;
      tmp[k].resize(count);
      sum += count;
    }
    if (!clean) { e.status[b] = bad_st ? (uint32_t)bad_st : (uint32_t)ZG_LIT_BITSTREAM_MISMATCH; continue; }   // a stream's end outranks any count (:116-121 comes first)
    if (sum != regen) continue;
    uint8_t* lit = e.lit.data() + blk.lit_base;
    for (int k = 0; k < 4; k++) { memcpy(lit, tmp[k].data(), tmp[k].size()); lit += tmp[k].size(); }
    e.status[b] = 0;
  }
}

static void k_seq(EmuBatch& e) {
  const uint32_t nb = (uint32_t)e.bb.blocks.size();
  // (zg_k_merge: the table descriptions of the sequences, and what the host found in a sequences section header, come behind the literals' verdicts)
  for (uint32_t b = 0; b < nb; b++) set_status(e, b, (int)e.fse_status[b]);
  for (uint32_t b = 0; b < nb; b++) if (e.bb.blocks[b].seq_host_status && !e.bb.blocks[b].host_status) set_status(e, b, (int)e.bb.blocks[b].seq_host_status);
  for (uint32_t b : e.bb.seq_blocks) {
    const ZgBlock blk = e.bb.blocks[b];
    int32_t sl[3] = {blk.ll_slot, blk.of_slot, blk.ml_slot};
    const uint32_t offs[3] = {ZG_FSE_LL_OFF, ZG_FSE_OF_OFF, ZG_FSE_ML_OFF};
    bool ok = e.status[b] == 0;
    const uint32_t* tp[3] = {nullptr, nullptr, nullptr}; unsigned lg[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
      if (sl[k] < 0) { ok = false; continue; }
      if ((uint32_t)sl[k] < nb && e.status[sl[k]] != 0) { ok = false; continue; }
      lg[k] = e.slot_log[(size_t)sl[k] * 4 + k];
      if (lg[k] > 9) { ok = false; continue; }
      tp[k] = e.fse.data() + (size_t)sl[k] * ZG_FSE_SLOT_U32 + offs[k];
    }
    if (!ok) {   // (the padding of the bitstream is looked at before the states are initialised: zg_k_seq)
      const uint32_t bo = e.aux[b].seq_bits_off;
      const bool nopad = bo <= blk.src_len && (blk.src_len == bo || e.src[blk.src_off + blk.src_len - 1u] == 0);
      set_status(e, b, nopad ? ZG_SEQ_EXTRA_PADDING : ZG_FSE_UNINIT);
      continue;
    }
    uint32_t bits_off = e.aux[b].seq_bits_off;
    if (bits_off > blk.src_len) { set_status(e, b, ZG_INTERNAL); continue; }
    const uint8_t* bs = e.src + blk.src_off + bits_off;
    ZgBlockSeqOut so;
    int st = e.use_fast ? zg_seq_decode_block_fast(bs, blk.src_len - bits_off, blk.nseq, tp[0], lg[0], tp[1], lg[1], tp[2], lg[2], blk.regen_size,
                                 e.seq.data() + blk.seq_base, &so)
                        : zg_seq_decode_block(bs, blk.src_len - bits_off, blk.nseq, tp[0], lg[0], tp[1], lg[1], tp[2], lg[2], blk.regen_size,
                                 e.seq.data() + blk.seq_base, &so);
    e.seqout[b] = so;
    set_status(e, b, st);
  }
}

static void k_scan(EmuBatch& e) {  // serial statement of zg_k_scan / zg_k_scanf
  uint64_t base = 0;
  for (uint32_t f = 0; f < e.bb.frames.size(); f++) {
    const ZgFrame fr = e.bb.frames[f];
    uint32_t h[3] = {fr.hist_init[0], fr.hist_init[1], fr.hist_init[2]};
    uint64_t pos = 0; uint32_t good = fr.nblocks, bad_status = 0;
    for (uint32_t i = 0; i < fr.nblocks; i++) {
      uint32_t b = fr.first_block + i;
      const ZgBlock& blk = e.bb.blocks[b];
      uint32_t st = blk.host_status ? blk.host_status : e.status[b];
      if (st) {
        good = i; bad_status = st;
        for (uint32_t j = i; j < fr.nblocks; j++) e.pos[fr.first_block + j].active = 0;
        // (like zg_k_scan: the failing block still learns where it would start and with which history — zg_k_exact may look at its first sequences)
        ZgBlockPos p; p.out_base = pos; p.hist_init[0] = h[0]; p.hist_init[1] = h[1]; p.hist_init[2] = h[2]; p.active = 0;
        e.pos[b] = p;
        break;
      }
      ZgBlockPos p; p.out_base = pos; p.hist_init[0] = h[0]; p.hist_init[1] = h[1]; p.hist_init[2] = h[2]; p.active = 1;
      e.pos[b] = p;
      if (blk.btype == ZG_BT_COMPRESSED && blk.nseq) {
        const ZgBlockSeqOut& so = e.seqout[b];
        pos += (uint64_t)blk.regen_size + so.sum_ml;
        uint32_t n0 = zg_sym_resolve(so.hist_end[0], h), n1 = zg_sym_resolve(so.hist_end[1], h), n2 = zg_sym_resolve(so.hist_end[2], h);
        h[0] = n0; h[1] = n1; h[2] = n2;
      } else pos += blk.regen_size;
    }
    ZgFrameOut fo; memset(&fo, 0, sizeof fo);
    fo.out_base = base; fo.out_size = pos; fo.status = bad_status; fo.bad_block = good; fo.good_blocks = good;
    fo.hist_end[0] = h[0]; fo.hist_end[1] = h[1]; fo.hist_end[2] = h[2];
    e.fout[f] = fo;
    base += pos;
  }
  e.dst.assign(base + 64, 0);
}

static void k_exec(EmuBatch& e) {  // zg_k_lit + zg_k_lz, serial
  for (uint32_t f = 0; f < e.bb.frames.size(); f++) {
    const ZgFrame fr = e.bb.frames[f];
    ZgFrameOut& fo = e.fout[f];
    uint8_t* fbase = e.dst.data() + fo.out_base;
    for (uint32_t i = 0; i < fo.good_blocks; i++) {
      uint32_t b = fr.first_block + i;
      const ZgBlock& blk = e.bb.blocks[b];
      const ZgBlockPos& p = e.pos[b];
      uint8_t* out = fbase + p.out_base;
      const uint8_t* body = e.src + blk.src_off;
      if (blk.btype == ZG_BT_RAW) { memcpy(out, body, blk.regen_size); continue; }
      if (blk.btype == ZG_BT_RLE) { memset(out, body[0], blk.regen_size); continue; }
      const bool rle = blk.lit_type == ZG_LT_RLE;
      const uint8_t* lit = blk.lit_type <= ZG_LT_RLE ? body + blk.lit_off : e.lit.data() + blk.lit_base;
      uint32_t sum_ll = 0, sum_ml = 0;
      int err = 0;
      if (blk.nseq) {
        const ZgBlockSeqOut& so = e.seqout[b];
        sum_ll = so.sum_ll; sum_ml = so.sum_ml;
        const EmuSeq* sq = e.seq.data() + blk.seq_base;
        for (uint32_t s = 0; s < blk.nseq && !err; s++) {
          const EmuSeq q = sq[s];
          uint32_t next = s + 1 < blk.nseq ? sq[s + 1].lit_start : sum_ll, ll = next - q.lit_start;
          uint8_t* o = out + (q.mdst - ll);
          for (uint32_t k = 0; k < ll; k++) o[k] = rle ? lit[0] : lit[q.lit_start + k];
          uint32_t off = zg_sym_resolve(q.of, p.hist_init);
          uint64_t dpos = p.out_base + q.mdst;
          if (off == 0) err = ZG_EXE_ZERO_OFFSET;
          else if (off > dpos) err = ZG_EXE_OFFSET_TOO_BIG;
          else { uint8_t* d = out + q.mdst; for (uint32_t k = 0; k < q.ml; k++) d[k] = d[(int64_t)k - off]; }
        }
      }
      if (err) { fo.status = err; fo.bad_block = i; fo.good_blocks = i; break; }
      uint32_t rest = blk.regen_size - sum_ll;
      uint8_t* o = out + ((uint64_t)sum_ll + sum_ml);
      for (uint32_t k = 0; k < rest; k++) o[k] = rle ? lit[0] : lit[sum_ll + k];
    }
  }
}

extern "C" {

void* zgemu_decode3(const uint8_t* src, size_t len, uint64_t max_window, int use_fast, uint32_t unit_blocks, uint32_t flat_slots) {
  EmuBatch* e = new EmuBatch();
  e->use_fast = use_fast;
  e->bb.unit_blocks = unit_blocks;        // 0: as the engine chooses from the submit size and flat_slots
  if (flat_slots) e->bb.flat_slots = flat_slots;
  e->src_store.assign(len + 128, 0);
  memcpy(e->src_store.data() + 64, src, len);
  e->src = e->src_store.data() + 64;
  e->parse_status = walk(e->src, len, max_window, &e->bb);
  e->bb.finish();
  const uint32_t nb = (uint32_t)e->bb.blocks.size(), nf = (uint32_t)e->bb.frames.size();
  e->aux.resize(nb + 1); e->slot_log.assign((size_t)e->bb.nslots() * 4, 0);
  e->fse.assign((size_t)e->bb.nslots() * ZG_FSE_SLOT_U32, 0xDEADBEEF);
  e->huf.assign((size_t)(e->bb.nhuf_slots + 1) * ZG_HUF_SLOT_U16, 0xFFFF);
  e->hufmax.assign(e->bb.nhuf_slots + 1, 0);
  e->status.assign(nb + 1, 0);
  e->fse_status.assign(nb + 1, 0);
  e->lit.assign(e->bb.lit_bytes + 64, 0xEE);
  e->seq.resize(e->bb.seq_count + 1);
  e->seqout.resize(nb + 1); e->pos.resize(nb + 1); e->fout.resize(nf + 1);
  k_tables(*e); k_huf(*e); k_seq(*e); k_scan(*e); k_exec(*e);
  return e;
}
void* zgemu_decode2(const uint8_t* src, size_t len, uint64_t max_window, int use_fast) { return zgemu_decode3(src, len, max_window, use_fast, 0, 0); }
void* zgemu_decode(const uint8_t* src, size_t len, uint64_t max_window) { return zgemu_decode2(src, len, max_window, 1); }
void zgemu_free(void* h) { delete (EmuBatch*)h; }
int zgemu_parse_status(void* h) { return ((EmuBatch*)h)->parse_status; }
uint32_t zgemu_num_frames(void* h) { return (uint32_t)((EmuBatch*)h)->bb.frames.size(); }
uint32_t zgemu_num_blocks(void* h) { return (uint32_t)((EmuBatch*)h)->bb.blocks.size(); }
int zgemu_frame(void* h, uint32_t f, uint64_t* base, uint64_t* size, uint32_t* status, uint32_t* bad_block) {
  EmuBatch* e = (EmuBatch*)h;
  if (f >= e->bb.frames.size()) return -1;
  *base = e->fout[f].out_base; *size = e->fout[f].out_size; *status = e->fout[f].status; *bad_block = e->fout[f].bad_block;
  return 0;
}
const uint8_t* zgemu_output(void* h) { return ((EmuBatch*)h)->dst.data(); }
uint32_t zgemu_block_status(void* h, uint32_t b) { EmuBatch* e = (EmuBatch*)h; return e->bb.blocks[b].host_status ? e->bb.blocks[b].host_status : e->status[b]; }
// per-block intermediates, same layout as the C ABI accessors
int zgemu_block(void* h, uint32_t b, uint32_t* info /*[12]*/) {
  EmuBatch* e = (EmuBatch*)h;
  const ZgBlock& k = e->bb.blocks[b];
  info[0] = k.btype; info[1] = k.lit_type; info[2] = k.nstreams; info[3] = k.seq_modes; info[4] = k.regen_size; info[5] = k.nseq;
  info[6] = k.frame; info[7] = (uint32_t)k.huf_slot; info[8] = (uint32_t)k.ll_slot; info[9] = (uint32_t)k.of_slot; info[10] = (uint32_t)k.ml_slot;
  info[11] = e->pos[b].active;
  return 0;
}
// the sequence bitstream of a block (after the table descriptions): pointer into the harness's copy of the input + its length
// (tools/dev/seq_sync.py: can a second decoder started mid-stream fall onto the true chain?)
const uint8_t* zgemu_block_seq_bits(void* h, uint32_t b, uint32_t* len) {
  EmuBatch* e = (EmuBatch*)h;
  const ZgBlock& k = e->bb.blocks[b];
  const uint32_t off = e->aux[b].seq_bits_off;
  *len = off <= k.src_len ? k.src_len - off : 0u;
  return e->src + k.src_off + off;
}
const uint8_t* zgemu_block_literals(void* h, uint32_t b) { EmuBatch* e = (EmuBatch*)h; return e->lit.data() + e->bb.blocks[b].lit_base; }
const EmuSeq* zgemu_block_sequences(void* h, uint32_t b) { EmuBatch* e = (EmuBatch*)h; return e->seq.data() + e->bb.blocks[b].seq_base; }
void zgemu_block_hist(void* h, uint32_t b, uint32_t* out3) { EmuBatch* e = (EmuBatch*)h; memcpy(out3, e->pos[b].hist_init, 12); }
const uint32_t* zgemu_fse_slot(void* h, uint32_t slot, uint8_t* logs) {
  EmuBatch* e = (EmuBatch*)h; memcpy(logs, e->slot_log.data() + (size_t)slot * 4, 4); return e->fse.data() + (size_t)slot * ZG_FSE_SLOT_U32;
}
// the host's plan for the LZ77 stages (zg_host_parse.cpp, finish()): units, sweep steps, per-frame sequence ranges. No decode:
// only the block walk. flat_slots as the engine would pass it (workgroups zg_k_flatten can hold at once).
void* zgemu_plan(const uint8_t* src, size_t len, uint32_t flat_slots, uint32_t unit_blocks) {
  EmuBatch* e = new EmuBatch();
  e->src_store.assign(len + 128, 0);
  memcpy(e->src_store.data() + 64, src, len);
  e->src = e->src_store.data() + 64;
  e->parse_status = walk(e->src, len, 1ull << 31, &e->bb);
  e->bb.flat_slots = flat_slots; e->bb.unit_blocks = unit_blocks;
  e->bb.finish();
  e->pos.resize(e->bb.blocks.size() + 1);       // (zgemu_block reports a block's "active" flag from here)
  return e;
}
uint32_t zgemu_num_units(void* h) { return (uint32_t)((EmuBatch*)h)->bb.units.size(); }
void zgemu_unit(void* h, uint32_t u, uint32_t* out4) { const ZgUnit& x = ((EmuBatch*)h)->bb.units[u]; out4[0] = x.frame; out4[1] = x.first_block; out4[2] = x.nblocks; out4[3] = x.noseq; }
uint32_t zgemu_num_steps(void* h) { return (uint32_t)((EmuBatch*)h)->bb.steps.size(); }
void zgemu_step(void* h, uint32_t i, uint32_t* out3) { const ZgStepRange& r = ((EmuBatch*)h)->bb.steps[i]; out3[0] = r.list_off; out3[1] = r.nunits; out3[2] = r.max_blocks; }
uint32_t zgemu_step_unit(void* h, uint32_t i) { return ((EmuBatch*)h)->bb.step_units[i]; }
uint32_t zgemu_num_step_units(void* h) { return (uint32_t)((EmuBatch*)h)->bb.step_units.size(); }
void zgemu_frame_plan(void* h, uint32_t f, uint32_t* out7) {
  const ZgFrame& fr = ((EmuBatch*)h)->bb.frames[f];
  out7[0] = fr.first_block; out7[1] = fr.nblocks; out7[2] = fr.first_unit; out7[3] = fr.nunits; out7[4] = fr.seq_first; out7[5] = fr.seq_count; out7[6] = fr.sparse;
}
uint32_t zgemu_seq_block(void* h, uint32_t i) { return ((EmuBatch*)h)->bb.seq_blocks[i]; }
const uint16_t* zgemu_huf_slot(void* h, uint32_t slot, int* max_bits) {
  EmuBatch* e = (EmuBatch*)h; *max_bits = e->hufmax[slot]; return e->huf.data() + (size_t)slot * ZG_HUF_SLOT_U16;
}
}
