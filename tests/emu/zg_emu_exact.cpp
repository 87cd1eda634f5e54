// zg_emu_exact.cpp — TEST-ONLY: runs the SOURCE of zg_k_exact (zstd-rs_amd/csrc/zg_exact.h: the reference's DecodeBuffer
// bookkeeping, replayed exactly) on the CPU through the SIMT emulator of zg_simt.h, on the intermediates the harness of
// zg_emu.cpp produced for a submit. tests/test_exact_cpu.py compares its verdicts with the oracle's FrameDecoder on frames
// whose matches reach beyond the window, into a dictionary, or across the reference's drain points.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../zstd-rs_amd/csrc/zg_types.h"
#include "zg_simt.h"
#include "../../zstd-rs_amd/csrc/zg_exact.h"
#include "zg_emu_batch.h"

extern "C" {

// h: an EmuBatch after zgemu_decode*. Every frame of the submit is treated as starting with `dict_len` bytes of dictionary in
// front of it, `prior_out` bytes decoded by earlier submits of which the caller still holds `prior_reach`, and a
// total_output_counter of `prior_counted`. The serial model's own execution errors are ignored: the walk covers every block the
// entropy stages accepted. Per frame: status_out (0 = nothing to object to), bad_out (frame-relative block), counted_out.
int zgemu_exact(void* h, uint32_t drain_rule, uint64_t dict_len, uint64_t prior_out, uint64_t prior_reach, uint64_t prior_counted,
                uint32_t* status_out, uint32_t* bad_out, uint64_t* counted_out) {
  EmuBatch* e = (EmuBatch*)h;
  const zg::BatchBuilder& bb = e->bb;
  const uint32_t nb = (uint32_t)bb.blocks.size(), nf = (uint32_t)bb.frames.size();
  std::vector<ZgSeq> seqs(e->seq.size() + 2);
  for (size_t i = 0; i < e->seq.size(); i++) {
    const EmuSeq& q = e->seq[i];
    seqs[i].of = q.of; seqs[i].w1 = ZG_SEQ_W1(q.mdst, q.ml); seqs[i].w2 = ZG_SEQ_W2(q.lit_start, q.ml);
  }
  std::vector<ZgFrame> frames(bb.frames.begin(), bb.frames.end());
  std::vector<ZgFrameOut> fout(e->fout.begin(), e->fout.begin() + nf);
  for (uint32_t f = 0; f < nf; f++) {
    ZgFrame& fr = frames[f];
    fr.dict_len = dict_len; fr.prior_out = prior_out; fr.prior_reach = prior_reach; fr.prior_counted = prior_counted;
    // blocks the entropy stages (and the parser) accepted
    uint32_t good = 0;
    while (good < fr.nblocks && !bb.blocks[fr.first_block + good].host_status && !e->status[fr.first_block + good]) good++;
    ZgFrameOut& fo = fout[f];
    // (a block that regenerates more than 128 KiB sends its frame to the in-order path, zg_k_scan: its records' position fields wrap)
    bool slow = false;
    for (uint32_t i = 0; i < good; i++) {
      const ZgBlock& bk = bb.blocks[fr.first_block + i];
      if (bk.btype == ZG_BT_COMPRESSED && bk.nseq && (uint64_t)bk.regen_size + e->seqout[fr.first_block + i].sum_ml > ZG_FLAT_MAX) slow = true;
    }
    fo.fast = slow ? 0u : 1u; fo.err_packed = 0xFFFFFFFFu; fo.good_blocks = good; fo.counted = 0;
    if (fo.status >= ZG_EXE_NOT_ENOUGH_LITERALS && fo.status <= ZG_EXE_DICT_TOO_SMALL) fo.status = 0;   // the serial model's execution verdict: not wanted here
    // ... except what zg_k_seqpost finds itself (a sequence without literals left, an offset of 0): the frame stops at that block with that status
    if (good < fr.nblocks && !bb.blocks[fr.first_block + good].host_status &&
        (e->status[fr.first_block + good] == (uint32_t)ZG_EXE_NOT_ENOUGH_LITERALS || e->status[fr.first_block + good] == (uint32_t)ZG_EXE_ZERO_OFFSET)) {
      fo.status = e->status[fr.first_block + good]; fo.bad_block = good;
    }
  }
  uint32_t totals[4] = {0, 0, 0, 0};
  ZgBatchDev d;
  memset(&d, 0, sizeof d);
  d.src = e->src; d.blocks = bb.blocks.data(); d.nblocks = nb; d.frames = frames.data(); d.nframes = nf;
  d.seq_arena = seqs.data(); d.seq_out = e->seqout.data(); d.pos = e->pos.data(); d.frame_out = fout.data(); d.totals = totals;
  for (uint32_t f = 0; f < nf; f++) {
    static ZgExactLds<256> L;
    simt::run(256, [&]() { zg_exact_frame<256>(d, f, drain_rule, L); });
    status_out[f] = fout[f].status; bad_out[f] = fout[f].bad_block; counted_out[f] = fout[f].counted;
  }
  return 0;
}

}  // extern "C"
