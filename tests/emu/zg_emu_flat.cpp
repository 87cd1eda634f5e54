// zg_emu_flat.cpp — TEST-ONLY: runs the SOURCE of the direct-unit flatten (zstd-rs_amd/csrc/zg_flat4.h, a body of zg_k_flatten)
// on the CPU through the SIMT emulator of zg_simt.h, on the intermediates the harness of zg_emu.cpp produced for a submit
// (host parser + table routines + serial model of the entropy stages). What comes out — the plaintext of the direct units — is
// compared with the oracle by tests/test_flat4_cpu.py. Not part of the product; nothing here is linked into libzgpu.so.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../zstd-rs_amd/csrc/zg_types.h"
#include "zg_simt.h"
#include "../../zstd-rs_amd/csrc/zg_flat4.h"
#include "../../zstd-rs_amd/csrc/zg_flat1.h"
#include "zg_emu_batch.h"

namespace {

template <int T, int TS, int SPT>
void run_unit(const ZgBatchDev& d, uint32_t ui) {
  static ZgFlat4Lds<T, TS, SPT> L;
  simt::run(T, [&]() { zg_flat4_unit<T, TS, SPT>(d, ui, L); });
}

template <int T, int TS, int SPT>
void run_unit1(const ZgBatchDev& d, uint32_t ui) {
  static ZgFlat1Lds<T, TS, SPT> L;
  simt::run(T, [&]() { zg_flat1_unit<T, TS, SPT>(d, ui, L); });
}

}  // namespace

extern "C" {

// Both bodies of zg_k_flatten on every unit of the plan, then a plain model of zg_k_sweep (units in frame order: every byte
// with a nonzero scratch word copies from that many bytes back): direct units through zg_flat4_unit, pointer-mode units through
// zg_flat1_unit (zg_flat1.h). shape as for zgemu_flat4.
//   dst_out   [total output bytes] the plaintext after the sweep
//   og_out    [total output bytes] the flatten scratch (effective offsets; untouched words: 0xEEEEEEEE)
//   unit_mode [units] 0 pointer, 1 no sequences, 2 direct; may be null
// Frames marked sparse (no scratch): zg_flat1_unit places their literal runs, a model of zg_k_sparse copies their matches in order.
int zgemu_flatten(void* h, int shape, uint8_t* dst_out, uint32_t* og_out, uint32_t* unit_mode) {
  EmuBatch* e = (EmuBatch*)h;
  const zg::BatchBuilder& bb = e->bb;
  const uint32_t nb = (uint32_t)bb.blocks.size(), nf = (uint32_t)bb.frames.size(), nu = (uint32_t)bb.units.size();
  uint64_t total = 0;
  for (uint32_t f = 0; f < nf; f++) total = e->fout[f].out_base + e->fout[f].out_size > total ? e->fout[f].out_base + e->fout[f].out_size : total;
  std::vector<uint8_t> dst(256 + total + 64, 0xAA), lit(64 + e->lit.size() + 64, 0);
  std::vector<uint32_t> og(total + 64, 0xEEEEEEEEu);
  memcpy(lit.data() + 64, e->lit.data(), e->lit.size());
  std::vector<ZgSeq> seqs(e->seq.size() + 2);
  for (size_t i = 0; i < e->seq.size(); i++) {
    const EmuSeq& q = e->seq[i];
    seqs[i].of = q.of; seqs[i].w1 = ZG_SEQ_W1(q.mdst, q.ml); seqs[i].w2 = ZG_SEQ_W2(q.lit_start, q.ml);
  }
  std::vector<ZgUnitInfo> uinfo(nu + 1);
  std::vector<ZgFrameOut> fout(e->fout.begin(), e->fout.begin() + nf);
  for (ZgFrameOut& fo : fout) { fo.fast = 1; fo.err_packed = 0xFFFFFFFFu; fo.og_base = fo.out_base; }
  uint32_t totals[4] = {0, 0, 0, 0};
  ZgBatchDev d;
  memset(&d, 0, sizeof d);
  d.src = e->src; d.blocks = bb.blocks.data(); d.nblocks = nb; d.frames = bb.frames.data(); d.nframes = nf;
  d.lit_arena = lit.data() + 64; d.seq_arena = seqs.data(); d.seq_out = e->seqout.data(); d.pos = e->pos.data();
  d.frame_out = fout.data(); d.dst = dst.data() + 256; d.dst_cap = total; d.totals = totals;
  d.units = bb.units.data(); d.nunits = nu; d.unit_info = uinfo.data();
  d.og = og.data(); d.og_words = total;
  // zg_k_lit: raw and RLE blocks and blocks without sequences are final before the flatten runs
  for (uint32_t b = 0; b < nb; b++) {
    const ZgBlock& blk = bb.blocks[b];
    if (!e->pos[b].active || (blk.btype == ZG_BT_COMPRESSED && blk.nseq)) continue;
    const uint64_t at = e->fout[blk.frame].out_base + e->pos[b].out_base;
    memcpy(dst.data() + 256 + at, e->dst.data() + at, blk.regen_size);
  }
  for (uint32_t u = 0; u < nu; u++) {
    const ZgUnit& un = bb.units[u];
    if (unit_mode) unit_mode[u] = un.noseq;
    if (!e->pos[un.first_block].active) continue;
    if (un.noseq & ZG_UNIT_DIRECT) {
      if (shape == 0) run_unit<256, 4096, 2>(d, u);
      else if (shape == 1) run_unit<512, 8192, 2>(d, u);
      else if (shape == 2) run_unit<1024, 16384, 2>(d, u);
      else if (shape == 3) run_unit<1024, 8192, 1>(d, u);      // zg_k_flatten4's default shape (two workgroups per CU)
      else if (shape == 4) run_unit<512, 4096, 1>(d, u);
      else if (shape == 5) run_unit<256, 2048, 1>(d, u);
      else if (shape == 6) run_unit<512, 2048, 1>(d, u);
      else return -1;
    } else {
      if (shape == 0) run_unit1<256, 4096, 2>(d, u);
      else if (shape == 1) run_unit1<512, 8192, 2>(d, u);
      else if (shape >= 2 && shape <= 6) run_unit1<1024, 16384, 2>(d, u);
      else return -1;
    }
  }
  // the sweep, unit after unit (frames are independent; units are listed frame by frame, in order)
  for (uint32_t u = 0; u < nu; u++) {
    const ZgUnit& un = bb.units[u];
    if (un.noseq || !e->pos[un.first_block].active) continue;
    const uint64_t at = e->fout[un.frame].out_base + e->pos[un.first_block].out_base;
    const uint64_t size = uinfo[u].size;
    if (fout[un.frame].err_packed != 0xFFFFFFFFu) { memcpy(dst.data() + 256 + at, e->dst.data() + at, size); continue; }
    uint8_t* o = dst.data() + 256 + at;
    if (bb.frames[un.frame].sparse) {
      // model of zg_k_sparse: zg_flat1_unit has put the literal runs in place; the matches follow in order
      for (uint32_t k = 0; k < un.nblocks; k++) {
        const uint32_t b = un.first_block + k;
        const ZgBlock& blk = bb.blocks[b];
        if (!e->pos[b].active) break;
        if (blk.btype != ZG_BT_COMPRESSED || !blk.nseq) continue;
        uint8_t* ob = dst.data() + 256 + e->fout[un.frame].out_base + e->pos[b].out_base;
        for (uint32_t i = 0; i < blk.nseq; i++) {
          const ZgSeq& q = seqs[blk.seq_base + i];
          const uint32_t off = zg_sym_resolve(q.of, e->pos[b].hist_init), ml = ZG_SEQ_ML(q), md = ZG_SEQ_MDST(q);
          for (uint32_t x = 0; x < ml; x++) ob[md + x] = *(ob + md + x - (int64_t)off);
        }
      }
      continue;
    }
    const uint32_t* w = og.data() + fout[un.frame].og_base + e->pos[un.first_block].out_base;
    for (uint64_t x = 0; x < size; x++) if (w[x]) o[x] = *(o + x - (int64_t)w[x]);
  }
  int first_status = 0;
  for (uint32_t f = 0; f < nf; f++) {
    if (fout[f].err_packed != 0xFFFFFFFFu) { if (!first_status) first_status = (int)(fout[f].err_packed & 0xFF); }
    else if (fout[f].status && !first_status) first_status = (int)fout[f].status;
  }
  memcpy(dst_out, dst.data() + 256, total);
  if (og_out) memcpy(og_out, og.data(), total * sizeof(uint32_t));
  return first_status;
}


// h: an EmuBatch after zgemu_decode* (every stage up to the in-order execution has run; e->dst holds the plaintext the serial
// model produced). Runs zg_flat4_unit on every DIRECT unit of the plan (shape: 0 = 256 threads x 4 KiB tiles, 1 = 512 x 8 KiB,
// 2 = 1024 x 16 KiB as on the GPU); the bytes of all other units are taken from the serial model (on the GPU they come from
// zg_flat1_unit + zg_k_sweep, which the GPU tests cover).
//   dst_out   [total output bytes]
//   unit_mode [units] 0 pointer, 1 no sequences, 2 direct; may be null
// Returns the first frame status found (0 = all frames fine), or -1 for an unknown shape.
int zgemu_flat4(void* h, int shape, uint8_t* dst_out, uint32_t* unit_mode) {
  EmuBatch* e = (EmuBatch*)h;
  const zg::BatchBuilder& bb = e->bb;
  const uint32_t nb = (uint32_t)bb.blocks.size(), nf = (uint32_t)bb.frames.size(), nu = (uint32_t)bb.units.size();
  uint64_t total = 0;
  for (uint32_t f = 0; f < nf; f++) total = e->fout[f].out_base + e->fout[f].out_size > total ? e->fout[f].out_base + e->fout[f].out_size : total;
  // device-side buffers as the engine lays them out: front pads in front of the output and the literals
  std::vector<uint8_t> dst(256 + total + 64, 0xAA), lit(64 + e->lit.size() + 64, 0);
  memcpy(lit.data() + 64, e->lit.data(), e->lit.size());
  std::vector<ZgSeq> seqs(e->seq.size() + 2);
  for (size_t i = 0; i < e->seq.size(); i++) {
    const EmuSeq& q = e->seq[i];
    seqs[i].of = q.of; seqs[i].w1 = ZG_SEQ_W1(q.mdst, q.ml); seqs[i].w2 = ZG_SEQ_W2(q.lit_start, q.ml);
  }
  std::vector<ZgUnitInfo> uinfo(nu + 1);
  std::vector<ZgFrameOut> fout(e->fout.begin(), e->fout.begin() + nf);
  for (ZgFrameOut& fo : fout) { fo.fast = 1; fo.err_packed = 0xFFFFFFFFu; fo.og_base = fo.out_base; }
  uint32_t totals[4] = {0, 0, 0, 0};
  ZgBatchDev d;
  memset(&d, 0, sizeof d);
  d.src = e->src; d.blocks = bb.blocks.data(); d.nblocks = nb; d.frames = bb.frames.data(); d.nframes = nf;
  d.lit_arena = lit.data() + 64; d.seq_arena = seqs.data(); d.seq_out = e->seqout.data(); d.pos = e->pos.data();
  d.frame_out = fout.data(); d.dst = dst.data() + 256; d.dst_cap = total; d.totals = totals;
  d.units = bb.units.data(); d.nunits = nu; d.unit_info = uinfo.data();
  for (uint32_t u = 0; u < nu; u++) {
    const ZgUnit& un = bb.units[u];
    if (unit_mode) unit_mode[u] = un.noseq;
    if (un.noseq & ZG_UNIT_DIRECT) {
      // zg_k_lit: raw and RLE blocks and blocks without sequences are final before the flatten runs
      for (uint32_t b = un.first_block; b < un.first_block + un.nblocks; b++) {
        const ZgBlock& blk = bb.blocks[b];
        if (!e->pos[b].active || (blk.btype == ZG_BT_COMPRESSED && blk.nseq)) continue;
        const uint64_t at = e->fout[blk.frame].out_base + e->pos[b].out_base;
        memcpy(dst.data() + 256 + at, e->dst.data() + at, blk.regen_size);
      }
      if (shape == 0) run_unit<256, 4096, 2>(d, u);
      else if (shape == 1) run_unit<512, 8192, 2>(d, u);
      else if (shape == 2) run_unit<1024, 16384, 2>(d, u);
      else if (shape == 3) run_unit<1024, 8192, 1>(d, u);      // zg_k_flatten4's default shape (two workgroups per CU)
      else if (shape == 4) run_unit<512, 4096, 1>(d, u);
      else if (shape == 5) run_unit<256, 2048, 1>(d, u);
      else if (shape == 6) run_unit<512, 2048, 1>(d, u);
      else return -1;
    } else {
      // every other unit: the serial model's bytes (what zg_flat1_unit + zg_k_sweep produce on the GPU)
      if (!e->pos[un.first_block].active) continue;
      const uint64_t at = e->fout[un.frame].out_base + e->pos[un.first_block].out_base;
      uint64_t end = e->fout[un.frame].out_base + e->fout[un.frame].out_size;
      if (u + 1 < nu && bb.units[u + 1].frame == un.frame && e->pos[bb.units[u + 1].first_block].active)
        end = e->fout[un.frame].out_base + e->pos[bb.units[u + 1].first_block].out_base;
      memcpy(dst.data() + 256 + at, e->dst.data() + at, end - at);
    }
  }
  int first_status = 0;
  for (uint32_t f = 0; f < nf; f++) {
    if (fout[f].err_packed != 0xFFFFFFFFu) { if (!first_status) first_status = (int)(fout[f].err_packed & 0xFF); }
    else if (fout[f].status && !first_status) first_status = (int)fout[f].status;
  }
  memcpy(dst_out, dst.data() + 256, total);
  return first_status;
}

}  // extern "C"
