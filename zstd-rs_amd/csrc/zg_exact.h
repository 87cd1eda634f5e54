// zg_exact.h — body of zg_k_exact: the reference's DecodeBuffer bookkeeping, replayed exactly, for the submits where it can
// change a verdict. The fast path (zg_k_flatten / zg_k_sweep) executes a frame as if every byte the frame has produced were
// still in reach; ruzstd's DecodeBuffer (decode_buffer.rs) is stricter and quirkier:
//   * repeat() (:79-111) copies from what is still IN the buffer: what the caller has drained is gone. FrameDecoder::decode_all
//     (frame_decoder.rs:541-577) drains down to window_size after every round of >= 1 MiB of new output, so a match that
//     reaches beyond window_size + what the round has produced fails there although the bytes were once decoded;
//   * a match that starts in front of the buffer goes to repeat_from_dict (:144-179): NotEnoughBytesInDictionary or a copy
//     from the dictionary while total_output_counter <= window_size, OffsetTooBig after that — and total_output_counter
//     counts the output of compressed blocks only (push :74-77, repeat :108; raw and RLE blocks go through extend_* and are
//     not counted, :62-72), and not a match that is served by the dictionary alone (:164-172).
// None of this matters for frames a conforming encoder made (offsets stay inside the window), so the host runs this kernel
// only when a submit could be affected: a sequence set an offset beyond its frame's window (zg_k_seqpost's flag), a frame
// ended with one of the two "offset too far" verdicts, or a frame has a dictionary (its counter feeds later submits).
//
// One workgroup per frame walks the frame's good blocks in order; inside a block the sequences are checked T at a time (a
// prefix sum over the dictionary-only matches gives every sequence its counter value). The first failing sequence in frame
// order decides, exactly like the serial reference. Written against the zx_* primitives like zg_flat4.h: the same source
// runs under the CPU emulator in tests/test_exact_cpu.py against the oracle.
#pragma once
#include <stdint.h>
#include "zg_types.h"
#include "zg_dev.h"

template <int T>
struct ZgExactLds {
  unsigned long long bad;       // first failing sequence of the block: index << 8 | status
  uint32_t wsum[T / 64];        // per wave: dictionary-only match bytes of the chunk
  uint32_t wpos[T / 64];        // per wave: output bytes (literals + match) of the chunk's sequences
};

template <int T>
ZX_DEV void zg_exact_frame(const ZgBatchDev& d, const uint32_t f, const uint32_t drain_rule, ZgExactLds<T>& L) {
  const uint32_t t = zx_tid(), lane = t & 63u, wv = t >> 6;
  const ZgFrame fr = d.frames[f];
  const ZgFrameOut fo = d.frame_out[f];
  if (d.totals[2]) return;
  // blocks whose sequences exist: everything in front of the first block the entropy stages or the parser rejected; when the
  // flatten (or, for a frame on the in-order path, zg_k_lz: it leaves the verdict in the status) found a sequence it could not
  // execute, that block is the last one to look at
  const bool exec_err = fo.err_packed != 0xFFFFFFFFu;
  const bool lz_err = !fo.fast && (fo.status == (uint32_t)ZG_EXE_ZERO_OFFSET || fo.status == (uint32_t)ZG_EXE_OFFSET_TOO_BIG || fo.status == (uint32_t)ZG_EXE_DICT_TOO_SMALL);
  // zg_k_seqpost rejected a sequence (it asks for more literals than the block has, or its offset is 0: sequence_execution.rs:14-30):
  // the reference executes the sequences in front of it first, and one of THOSE may reach too far — that error comes first. The
  // block's records exist up to the rejected sequence (ZgBlockSeqOut::pad): they are walked like any other block's.
  const bool sp_err = !exec_err && !lz_err && (fo.status == (uint32_t)ZG_EXE_NOT_ENOUGH_LITERALS || fo.status == (uint32_t)ZG_EXE_ZERO_OFFSET) &&
                      fo.good_blocks < fr.nblocks && d.seq_out[fr.first_block + fo.good_blocks].pad != 0u &&
                      d.blocks[fr.first_block + fo.good_blocks].host_status == 0u;
  uint32_t nwalk = exec_err ? (fo.err_packed >> 8) + 1u : (lz_err || sp_err) ? fo.good_blocks + 1u : fo.good_blocks;
  if (nwalk > fr.nblocks) nwalk = fr.nblocks;
  uint64_t buf = fr.prior_reach;         // DecodeBuffer::len(): undrained bytes
  uint64_t cnt = fr.prior_counted;       // total_output_counter
  uint64_t round0 = buf;                 // decode_blocks' buffer_size_before (frame_decoder.rs:321-323)
  bool cut_here = false;                 // the drain rule has dropped bytes INSIDE this submit (the device still holds them in place)
  for (uint32_t i = 0; i < nwalk; i++) {
    const uint32_t b = fr.first_block + i;
    const ZgBlock blk = d.blocks[b];
    const bool cut = sp_err && i == fo.good_blocks;               // only the sequences in front of the rejected one
    const uint32_t nseq = cut ? d.seq_out[b].pad - 1u : blk.nseq;
    const bool seqs = blk.btype == ZG_BT_COMPRESSED && nseq;
    const uint64_t size = seqs ? (uint64_t)blk.regen_size + d.seq_out[b].sum_ml : blk.regen_size;
    uint64_t dict_only = 0;              // bytes of the block's matches that came from the dictionary alone
    if (seqs) {
      const ZgBlockPos p = d.pos[b];
      const ZgSeq* sq = d.seq_arena + blk.seq_base;
      if (t == 0) L.bad = ~0ull;
      zx_barrier();
      const uint32_t sum_ll = d.seq_out[b].sum_ll;
      uint32_t pos_carry = 0;                                      // output bytes of the block's sequences in front of the chunk
      for (uint32_t j0 = 0; j0 < nseq; j0 += T) {
        const uint32_t j = j0 + t;
        const bool have = j < nseq;
        uint32_t off = 0, m0 = 0, ml = 0, ll = 0;
        if (have) {
          const ZgSeq q = sq[j];
          off = zg_sym_resolve(q.of, p.hist_init); ml = ZG_SEQ_ML(q);
          // where the match starts: the record's position field wraps in a block beyond 128 KiB (frames on the in-order path),
          // ml and ll = (next literal index - this one) mod 2^17 stay exact (zg_types.h): positions are rebuilt from them
          const uint32_t nx = (j + 1 < nseq || cut) ? ZG_SEQ_LIT(sq[j + 1]) : sum_ll;   // (a cut block: the rejected sequence's record follows)
          ll = (nx - ZG_SEQ_LIT(q)) & 0x1FFFFu;
        }
        {
          uint32_t sp = ll + ml;
          for (int o = 1; o < 64; o <<= 1) { const uint32_t pv = zx_shfl_up(sp, o); if ((int)lane >= o) sp += pv; }
          if (lane == 63u) L.wpos[wv] = sp;
          zx_barrier();
          uint32_t before = sp - (ll + ml), all = 0;
          for (uint32_t w = 0; w < (uint32_t)T / 64u; w++) { const uint32_t x = L.wpos[w]; if (w < wv) before += x; all += x; }
          m0 = pos_carry + before + ll;
          pos_carry += all;
        }
        const uint64_t at = buf + m0;                    // buffer.len() when repeat() is called for this match
        const bool outside = have && (uint64_t)off > at; // :80
        const uint64_t need = outside ? (uint64_t)off - at : 0ull;   // bytes_from_dict :150
        const uint32_t v = (outside && need >= ml) ? ml : 0u;        // the match lies in the dictionary alone: not counted (:164-172)
        // exclusive prefix sum of v over the chunk
        uint32_t sx = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t pv = zx_shfl_up(sx, o); if ((int)lane >= o) sx += pv; }
        if (lane == 63u) L.wsum[wv] = sx;
        zx_barrier();
        uint32_t before = sx - v, all = 0;
        for (uint32_t w = 0; w < (uint32_t)T / 64u; w++) { const uint32_t x = L.wsum[w]; if (w < wv) before += x; all += x; }
        if (have && off == 0) zx_min_lds64(&L.bad, ((unsigned long long)j << 8) | (uint32_t)ZG_EXE_ZERO_OFFSET);   // sequence_execution.rs:28-30 (comes before repeat())
        if (have && off >= ZG_OFF_HUGE - 2u && !outside)               // an offset >= 2^30 (zg_k_seqpost) with 1 GiB held undrained: the one case the engine
          zx_min_lds64(&L.bad, ((unsigned long long)j << 8) | (uint32_t)ZG_UNSUPPORTED);   // cannot decide like the reference (its value is gone)
        if (outside) {
          const uint64_t c = cnt + m0 - (dict_only + before);        // total_output_counter at this repeat(): literals of this sequence included
          uint32_t st = 0;
          if (c > fr.window_size) st = ZG_EXE_OFFSET_TOO_BIG;                       // :173-178
          else if (need > fr.dict_len) st = ZG_EXE_DICT_TOO_SMALL;                  // :152-157
          // (at < prior_out + produced + m0 — bytes of the frame have been drained, and the match starts in front of what is left: the
          //  reference then splices the dictionary's tail with the OLDEST byte it still holds, decode_buffer.rs:159-163. Since round 5 the
          //  engine's window is laid out exactly like that when a dictionary is in front — [dictionary content][undrained bytes], the drained
          //  ones dropped: FrameState::make_room — so the copy with the sequence's own offset yields the reference's bytes: no verdict here.
          //  That holds for drains BETWEEN submits. A drain the rule models inside this submit — decode_all's rounds of 1 MiB — leaves the
          //  drained bytes in place on the device, where the copy would read them instead of the dictionary's tail: the one splice the
          //  engine cannot serve, ADVICE r5. It needs a dictionary, more than 1 MiB of uncounted raw / RLE output in ONE decode_all frame
          //  and an offset beyond the window)
          else if (cut_here) st = ZG_UNSUPPORTED;
          if (st) zx_min_lds64(&L.bad, ((unsigned long long)j << 8) | st);
        }
        dict_only += all;
        zx_barrier();
        if (L.bad != ~0ull) break;
      }
      const unsigned long long bad = L.bad;
      zx_barrier();
      if (cut && bad == ~0ull) return;                              // nothing in front of the rejected sequence fails: zg_k_seqpost's verdict stands
      if (bad != ~0ull) {
        if (t == 0) {
          ZgFrameOut* o = &d.frame_out[f];
          o->status = (uint32_t)bad & 0xFFu; o->bad_block = i; o->good_blocks = i;
          o->err_packed = (i << 8) | ((uint32_t)bad & 0xFFu);
          d.seq_out[b].pad = (uint32_t)(bad >> 8) + 1u;             // 1 + the sequence that failed, as zg_k_seqpost leaves it for its own verdicts (Batch::sync: zg_k_partial)
        }
        return;
      }
    }
    buf += size;
    if (blk.btype == ZG_BT_COMPRESSED) cnt += size - dict_only;
    if (drain_rule == ZG_DRAIN_DECODE_ALL && buf - round0 >= (1u << 20)) {   // UptoBytes(1 MiB) is reached after this block (:364-375); read() then drains
      if (buf > fr.window_size) { buf = fr.window_size; cut_here = true; }   // can_drain_to_window_size (decode_buffer.rs:182-188)
      round0 = buf;
    }
  }
  if (t == 0) d.frame_out[f].counted = cnt - fr.prior_counted;
}
