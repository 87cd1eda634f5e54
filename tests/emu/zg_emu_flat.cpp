// zg_emu_flat.cpp — TEST-ONLY: runs the SOURCE of the flatten kernel (zstd-rs_amd/csrc/zg_flat4.h, the body of zg_k_flat4)
// on the CPU through the SIMT emulator of zg_simt.h, on the intermediates the harness of zg_emu.cpp produced for a submit
// (host parser + table routines + serial model of the entropy stages), followed by a serial statement of zg_k_sweep.
// What comes out — the flatten scratch and the plaintext — is compared with tests/lz_model.py and the oracle by
// tests/test_flat4_cpu.py. Not part of the product; nothing here is linked into libzgpu.so.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../zstd-rs_amd/csrc/zg_types.h"
#include "zg_simt.h"
#include "../../zstd-rs_amd/csrc/zg_flat4.h"
#include "zg_emu_batch.h"

namespace {

template <int T, int TS, int SPT>
void run_unit(const ZgBatchDev& d, uint32_t ui, bool direct) {
  if (direct) {
    static ZgFlat4Lds<T, TS, SPT, true> L;
    simt::run(T, [&]() { zg_flat4_unit<T, TS, SPT, true>(d, ui, L); });
  } else {
    static ZgFlat4Lds<T, TS, SPT, false> L;
    simt::run(T, [&]() { zg_flat4_unit<T, TS, SPT, false>(d, ui, L); });
  }
}

}  // namespace

extern "C" {

// h: an EmuBatch after zgemu_decode* (every stage up to the in-order execution has run; e->dst holds the plaintext the serial
// model produced). Runs zg_flat4_unit on every unit (shape: 0 = 256 threads x 4 KiB tiles, 1 = 512 x 8 KiB, 2 = 1024 x 16 KiB as on
// the GPU; force_pointer: ignore the plan's direct units), then the sweep, unit after unit.
//   dst_out   [total output bytes] what flatten + sweep produce
//   og_out    [total output bytes] the scratch words (0 where a unit wrote none), may be null
//   unit_mode [units] 0 pointer, 1 no sequences, 2 direct; may be null
// Returns the first frame status found (0 = all frames fine), or -1 for an unknown shape.
int zgemu_flat4(void* h, int shape, int force_pointer, uint8_t* dst_out, uint32_t* og_out, uint32_t* unit_mode) {
  EmuBatch* e = (EmuBatch*)h;
  const zg::BatchBuilder& bb = e->bb;
  const uint32_t nb = (uint32_t)bb.blocks.size(), nf = (uint32_t)bb.frames.size(), nu = (uint32_t)bb.units.size();
  uint64_t total = 0;
  for (uint32_t f = 0; f < nf; f++) total = e->fout[f].out_base + e->fout[f].out_size > total ? e->fout[f].out_base + e->fout[f].out_size : total;
  // device-side buffers as the engine lays them out: front pads in front of the output, the literals and the scratch
  std::vector<uint8_t> dst(256 + total + 64, 0xAA), lit(64 + e->lit.size() + 64, 0);
  std::vector<uint32_t> og(16 + total + 16, 0xDEADBEEFu);
  memcpy(lit.data() + 64, e->lit.data(), e->lit.size());
  std::vector<ZgSeq> seqs(e->seq.size() + 2);
  for (size_t i = 0; i < e->seq.size(); i++) {
    const EmuSeq& q = e->seq[i];
    seqs[i].of = q.of; seqs[i].w1 = ZG_SEQ_W1(q.mdst, q.ml); seqs[i].w2 = ZG_SEQ_W2(q.lit_start, q.ml);
  }
  std::vector<ZgUnit> units(bb.units);
  std::vector<ZgUnitInfo> uinfo(nu + 1);
  std::vector<ZgFrameOut> fout(e->fout.begin(), e->fout.begin() + nf);
  for (ZgFrameOut& fo : fout) { fo.fast = 1; fo.err_packed = 0xFFFFFFFFu; fo.og_base = fo.out_base; }
  for (ZgUnit& u : units) if (force_pointer && (u.noseq & ZG_UNIT_DIRECT)) u.noseq = 0;
  uint32_t totals[4] = {0, 0, 0, 0};
  ZgBatchDev d;
  memset(&d, 0, sizeof d);
  d.src = e->src; d.blocks = bb.blocks.data(); d.nblocks = nb; d.frames = bb.frames.data(); d.nframes = nf;
  d.lit_arena = lit.data() + 64; d.seq_arena = seqs.data(); d.seq_out = e->seqout.data(); d.pos = e->pos.data();
  d.frame_out = fout.data(); d.dst = dst.data() + 256; d.dst_cap = total; d.totals = totals;
  d.og = og.data() + 16; d.og_words = total; d.units = units.data(); d.nunits = nu; d.unit_info = uinfo.data();
  // zg_k_lit: raw and RLE blocks and blocks without sequences are final before the flatten runs
  for (uint32_t b = 0; b < nb; b++) {
    const ZgBlock& blk = bb.blocks[b];
    if (!e->pos[b].active) continue;
    if (blk.btype == ZG_BT_COMPRESSED && blk.nseq) continue;
    const uint64_t at = e->fout[blk.frame].out_base + e->pos[b].out_base;
    memcpy(dst.data() + 256 + at, e->dst.data() + at, blk.regen_size);
  }
  for (uint32_t u = 0; u < nu; u++) {
    const bool direct = (units[u].noseq & ZG_UNIT_DIRECT) != 0;
    if (unit_mode) unit_mode[u] = units[u].noseq;
    if (shape == 0) run_unit<256, 4096, 2>(d, u, direct);
    else if (shape == 1) run_unit<512, 8192, 2>(d, u, direct);
    else if (shape == 2) run_unit<1024, 16384, 2>(d, u, direct);
    else return -1;
  }
  // zg_k_sweep + zg_k_fin, serially: the units of a frame in order, every match byte of a pointer-mode unit from finished output
  int first_status = 0;
  for (uint32_t f = 0; f < nf; f++) {
    const ZgFrame& fr = bb.frames[f];
    ZgFrameOut& fo = fout[f];
    if (fo.err_packed != 0xFFFFFFFFu) { if (!first_status) first_status = (int)(fo.err_packed & 0xFF); continue; }
    if (fo.status) { if (!first_status) first_status = (int)fo.status; }
    for (uint32_t k = 0; k < fr.nunits; k++) {
      const ZgUnit& un = units[fr.first_unit + k];
      if (un.noseq) continue;
      if (!e->pos[un.first_block].active) break;
      const uint64_t at = fo.out_base + e->pos[un.first_block].out_base;
      const uint32_t size = uinfo[fr.first_unit + k].size;
      uint8_t* o = dst.data() + 256 + at;
      const uint32_t* g = og.data() + 16 + at;
      // zg_k_lit places the literal bytes of pointer-mode units (scratch word 0): here they come from the serial model's output
      for (uint32_t i = 0; i < size; i++) if (!g[i]) o[i] = e->dst[at + i];
      for (uint32_t i = 0; i < size; i++) if (g[i]) o[i] = o[(int64_t)i - (int64_t)g[i]];
    }
  }
  memcpy(dst_out, dst.data() + 256, total);
  if (og_out) for (uint64_t i = 0; i < total; i++) og_out[i] = og[16 + i];
  return first_status;
}

}  // extern "C"
