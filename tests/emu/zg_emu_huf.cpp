// zg_emu_huf.cpp — TEST-ONLY: runs the SOURCE of zg_k_huf (zstd-rs_amd/csrc/zg_huf.h: Huffman literal streams, one wave per stream,
// self-synchronising chunks, the bit window held in registers) on the CPU through the SIMT emulator of zg_simt.h, on the
// intermediates the harness of zg_emu.cpp produced for a submit (host parser, Huffman tables, positions). What comes out — the
// literals arena, or for ZG_FLAG_LIT_DIRECT the output bytes of the blocks without sequences — is compared with the serial model
// and the oracle by tests/test_huf_cpu.py. Not part of the product; nothing here is linked into libzgpu.so.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../zstd-rs_amd/csrc/zg_types.h"
#include "zg_simt.h"
#include "../../zstd-rs_amd/csrc/zg_huf.h"
#include "zg_emu_batch.h"

extern "C" {

uint64_t zgemu_lit_bytes(void* h) { return ((EmuBatch*)h)->bb.lit_bytes; }
uint64_t zgemu_block_lit_base(void* h, uint32_t b) { return ((EmuBatch*)h)->bb.blocks[b].lit_base; }
uint64_t zgemu_block_out_base(void* h, uint32_t b) { EmuBatch* e = (EmuBatch*)h; return e->fout[e->bb.blocks[b].frame].out_base + e->pos[b].out_base; }

// h: an EmuBatch after zgemu_decode* (e->lit: the serial model's literals, e->dst: its plaintext, e->status: its verdicts).
//   lit_out     [bb.lit_bytes] the arena zg_k_huf's source fills
//   dst_out     [total output] only with direct != 0: what it wrote straight into the output (0xAA elsewhere)
//   lit_status  [blocks] rank << 8 | status, as the kernel leaves it (zg_k_merge strips the rank)
//   lit_counts  [4 * blocks]
// Blocks whose tables or headers the earlier stages rejected are skipped like on the GPU (tab_status).
int zgemu_huf(void* h, int direct, uint8_t* lit_out, uint8_t* dst_out, uint32_t* lit_status, uint32_t* lit_counts) {
  EmuBatch* e = (EmuBatch*)h;
  const zg::BatchBuilder& bb = e->bb;
  const uint32_t nb = (uint32_t)bb.blocks.size(), nf = (uint32_t)bb.frames.size();
  uint64_t total = 0;
  for (uint32_t f = 0; f < nf; f++) total = e->fout[f].out_base + e->fout[f].out_size > total ? e->fout[f].out_base + e->fout[f].out_size : total;
  std::vector<uint8_t> dst(256 + total + 64, 0xAA), lit(64 + bb.lit_bytes + 128, 0x55);
  std::vector<uint32_t> tab_status(nb + 1, 0), lstat(nb + 1, 0), lcnt(4 * (size_t)nb + 4, 0);
  // what zg_k_tables leaves: a tree description that failed keeps its block out of zg_k_huf. The harness folds every stage's
  // verdict into one word per block; statuses of the table stage are the HUF_TABLE / FSE ones, found before any stream is read
  for (uint32_t b = 0; b < nb; b++) {
    const uint32_t st = e->status[b];
    if (st == (uint32_t)ZG_HUF_TABLE || st == (uint32_t)ZG_INTERNAL) tab_status[b] = st;
  }
  ZgBatchDev d;
  memset(&d, 0, sizeof d);
  d.src = e->src; d.blocks = bb.blocks.data(); d.nblocks = nb; d.frames = bb.frames.data(); d.nframes = nf;
  d.aux = e->aux.data(); d.huf_arena = e->huf.data(); d.huf_maxbits = e->hufmax.data();
  d.tab_status = tab_status.data(); d.lit_status = lstat.data(); d.lit_counts = lcnt.data();
  d.lit_arena = lit.data() + 64; d.pos = e->pos.data(); d.frame_out = e->fout.data();
  d.dst = dst.data() + 256; d.dst_cap = total;
  d.huf_items = bb.huf_items.data(); d.huf_groups = bb.huf_groups.data(); d.nhuf_groups = (uint32_t)bb.huf_groups.size();
  d.flags = direct ? ZG_FLAG_LIT_DIRECT : 0u;
  uint32_t totals[4] = {0, 0, 0, 0};
  d.totals = totals;
  static ZgHufLds<ZG_HUF_GROUP> L;
  for (uint32_t g = 0; g < d.nhuf_groups; g++) simt::run(64 * ZG_HUF_GROUP, [&]() { zg_huf_group<ZG_HUF_GROUP>(d, g, L); });
  memcpy(lit_out, lit.data() + 64, bb.lit_bytes);
  if (dst_out) memcpy(dst_out, dst.data() + 256, total);
  memcpy(lit_status, lstat.data(), nb * sizeof(uint32_t));
  memcpy(lit_counts, lcnt.data(), 4 * (size_t)nb * sizeof(uint32_t));
  return 0;
}

}  // extern "C"
