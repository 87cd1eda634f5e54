// zg_emu_serial.h — TEST-ONLY serial model of the wave-cooperative decoders of zg_kernels.hip (zg_k_huf, zg_k_seq +
// zg_k_seqpost): one Huffman stream and one sequences section decoded the plain way, one field after the other. It is a
// separate restatement, NOT the code the GPU runs: the kernels are checked against the oracle on the GPU
// (tests/test_gpu_parity.py); this model lets the CPU tests exercise the host parser, the table routines of zg_dev.h
// (which the kernels do call) and the symbolic offset history end to end without a GPU.
#pragma once
#include "../../zstd-rs_amd/csrc/zg_dev.h"

struct EmuSeq { uint32_t of, ml, mdst, lit_start; };   // of: resolved offset or symbolic (zg_dev.h)

// ---- one Huffman stream (literals_section_decoder.rs:94-122 / :128-147; huff0_decoder.rs:25-53) -------------
// Decodes into dst[0..cap). The stream ends by bit exhaustion: symbols are emitted while the reader has more
// than -max_bits bits left. Returns the number of symbols the stream holds (may exceed cap: nothing is written
// past cap) and the final bits_remaining through *end_bits.
template <typename TabPtr>
ZG_HD int zg_huf_decode_stream(const uint8_t* s, uint32_t len, TabPtr table, unsigned max_bits, uint8_t* dst, uint32_t cap, uint32_t* count_out,
                               int32_t* end_bits) {
  if (len == 0) return ZG_LIT_EXTRA_PADDING;
  uint32_t lastb = s[len - 1];
  if (lastb == 0) return ZG_LIT_EXTRA_PADDING;
  int32_t P = (int32_t)(len - 1) * 8 + (int32_t)(zg_hbit(lastb) - 1);
  const uint32_t mask = (1u << max_bits) - 1u;
  const int32_t lim = -(int32_t)max_bits;
  P -= max_bits;
  uint32_t state = zg_bits_at_z(s, P, max_bits);
  uint32_t n = 0;
  // fast part: every read lies fully inside the stream
  while (P >= 16 && n < cap) {
    uint32_t e = table[state];
    dst[n++] = (uint8_t)e;
    unsigned nb = e >> 8;
    P -= nb;
    state = ((state << nb) & mask) | zg_bits_at(s, P, nb);
  }
  while (P > lim) {
    uint32_t e = table[state];
    if (n < cap) dst[n] = (uint8_t)e;
    n++;
    unsigned nb = e >> 8;
    P -= nb;
    state = ((state << nb) & mask) | zg_bits_at_z(s, P, nb);
    if (n > cap + 8u) break;  // more symbols than the section can hold: the caller reports the mismatch
  }
  *count_out = n;
  *end_bits = P;
  return ZG_OK;
}

// ---- the sequence section of one block (sequence_section_decoder.rs:14-221) --------------------------------------
// bs[0..bs_len) is the reversed bitstream (after the table descriptions). Tables are packed entries; a log of 0
// means a one-entry table (RLE mode, or a carried RLE symbol). Writes nseq EmuSeq and the block summary.
template <typename TabPtr>
ZG_HD int zg_seq_decode_block(const uint8_t* bs, uint32_t bs_len, uint32_t nseq, TabPtr t_ll, unsigned ll_log, TabPtr t_of, unsigned of_log,
                              TabPtr t_ml, unsigned ml_log, uint32_t regen_size, EmuSeq* out, ZgBlockSeqOut* sum) {
  if (bs_len == 0) return ZG_SEQ_EXTRA_PADDING;
  uint32_t lastb = bs[bs_len - 1];
  if (lastb == 0) return ZG_SEQ_EXTRA_PADDING;
  int32_t P = (int32_t)(bs_len - 1) * 8 + (int32_t)(zg_hbit(lastb) - 1);
  // init order LL, OF, ML (:164-166)
  P -= ll_log; uint32_t e_ll = t_ll[zg_bits_at_z(bs, P, ll_log)];
  P -= of_log; uint32_t e_of = t_of[zg_bits_at_z(bs, P, of_log)];
  P -= ml_log; uint32_t e_ml = t_ml[zg_bits_at_z(bs, P, ml_log)];
  uint32_t h0 = (1u << 30) | 0u, h1 = (2u << 30) | 0u, h2 = (3u << 30) | 0u;
  uint32_t lit_pos = 0, out_pos = 0, sum_ml = 0;
  int status = ZG_OK, exe_status = ZG_OK;
  uint32_t emitted = 0;
  for (uint32_t i = 0; i < nseq; i++) {
    unsigned of_code = ZG_FSE_SYM(e_of), ml_code = ZG_FSE_SYM(e_ml), ll_code = ZG_FSE_SYM(e_ll);
    unsigned xb_of = ZG_FSE_XB(e_of), xb_ml = ZG_FSE_XB(e_ml), xb_ll = ZG_FSE_XB(e_ll);
    // extra bits in the order OF, ML, LL (:185; get_bits_triple bit_reader_reverse.rs:151-162)
    P -= xb_of; uint32_t obits = zg_bits_at_z(bs, P, xb_of);
    P -= xb_ml; uint32_t ml_add = zg_bits_at_z(bs, P, xb_ml);
    P -= xb_ll; uint32_t ll_add = zg_bits_at_z(bs, P, xb_ll);
    uint32_t of = obits + (1u << of_code);
    uint32_t ml = ZG_ML_BASE[ml_code] + ml_add;
    uint32_t ll = ZG_LL_BASE[ll_code] + ll_add;
    if (i + 1 < nseq) {  // state update order LL, ML, OF (:204-206)
      unsigned nb;
      nb = ZG_FSE_NB(e_ll); P -= nb; e_ll = t_ll[ZG_FSE_BL(e_ll) + zg_bits_at_z(bs, P, nb)];
      nb = ZG_FSE_NB(e_ml); P -= nb; e_ml = t_ml[ZG_FSE_BL(e_ml) + zg_bits_at_z(bs, P, nb)];
      nb = ZG_FSE_NB(e_of); P -= nb; e_of = t_of[ZG_FSE_BL(e_of) + zg_bits_at_z(bs, P, nb)];
    }
    if (P < 0) { status = ZG_SEQ_NOT_ENOUGH_BYTES; break; }  // :209-211
    // execution bookkeeping (sequence_execution.rs:10-39), done here because this lane walks the block in order.
    // The reference decodes the whole section before executing, so a bitstream error outranks these.
    if (exe_status == ZG_OK) {
      const bool big = of > 3u && of - 3u >= (1u << 30);   // offsets >= 2^30 would collide with the symbolic tags: they travel as ZG_OFF_HUGE (zg_k_seqpost)
      uint32_t actual = zg_hist_step(big ? ZG_OFF_HUGE + 3u : of, ll, h0, h1, h2);
      if (actual == 0) exe_status = ZG_EXE_ZERO_OFFSET;
      else if ((uint64_t)lit_pos + ll > regen_size) exe_status = ZG_EXE_NOT_ENOUGH_LITERALS;
      else if ((uint64_t)out_pos + ll + ml >= (1ull << 31)) exe_status = ZG_UNSUPPORTED;
      EmuSeq q;                                            // (the rejected sequence leaves a record too, like zg_k_seqpost's: zg_k_exact reads its literal index)
      q.of = actual; q.ml = ml; q.mdst = out_pos + ll; q.lit_start = lit_pos;
      out[i] = q;
      if (exe_status == ZG_OK) { lit_pos += ll; out_pos += ll + ml; sum_ml += ml; emitted++; }
    }
  }
  if (status == ZG_OK && P > 0) status = ZG_SEQ_EXTRA_BITS;  // :214-220
  if (status == ZG_OK) status = exe_status;
  sum->sum_ll = lit_pos; sum->sum_ml = sum_ml;
  sum->hist_end[0] = h0; sum->hist_end[1] = h1; sum->hist_end[2] = h2;
  sum->pad = exe_status ? emitted + 1u : 0u;            // like zg_k_seqpost: 1 + the sequence that cannot be executed
  return status;
}

// ---- 128-bit bit window for the sequence decoder -----------------------------------------------------------------
// One sequence reads at most 31+16+16 extra bits and 9+9+8 state bits = 89 bits, all directly below the current bit
// position P. One unaligned 16-byte load at byte (P/8 - 15) covers bits [P - 120 - P%8, P + 8 - P%8): the whole
// sequence. The load for the next sequence is issued as soon as this sequence's bit count is known, so its latency
// overlaps the extraction work (the engine keeps 16 bytes of padding in front of the compressed buffer).
struct __attribute__((packed)) zg_u128u { uint64_t lo, hi; };
struct ZgWin { uint64_t lo, hi; int32_t base; };   // base = bit index of bit 0 of lo
ZG_HD ZgWin zg_win_load(const uint8_t* s, int32_t P) {
  int32_t by = (P >> 3);
  if (by < -1) by = -1;                     // positions below the stream start only occur on the error path
  const int32_t kb = by - 15;
  const zg_u128u* p = (const zg_u128u*)(s + kb);
  ZgWin w;
  w.lo = p->lo; w.hi = p->hi; w.base = kb * 8;
  return w;
}
// bits [q, q+n) of the stream, n <= 32, taken from the window (q >= w.base, q + n <= w.base + 128)
ZG_HD uint32_t zg_win_bits(const ZgWin& w, int32_t q, unsigned n) {
  const unsigned r = (unsigned)(q - w.base) & 127u;
  uint64_t v;
  if (r >= 64) v = w.hi >> (r - 64);
  else v = r ? ((w.lo >> r) | (w.hi << (64 - r))) : w.lo;
  return (uint32_t)(v & ((1ull << n) - 1ull));
}

// ---- sequence decode, one step at a time ---------------------------------------------------------------------------
// The kernel (zg_k_seq) drives these with windows read from an LDS ring; zg_seq_decode_block_fast below drives them
// with windows read straight from memory (host harness, and the readable statement of what the kernel does).
struct EmuSeqState {
  uint32_t e_ll, e_of, e_ml;     // current table entries of the three FSE states
  int32_t P;                     // bits_remaining (bit_reader_reverse.rs:27-29)
  uint32_t h0, h1, h2;           // symbolic offset history
  uint32_t lit_pos, out_pos, sum_ml, emitted;
  int status, exe_status;
};
// skip the padding of the last byte; false = ExtraPadding (sequence_section_decoder.rs:29-40)
ZG_HD bool zg_seq_begin(EmuSeqState& st, uint32_t bs_len, uint32_t lastb) {
  st.h0 = 1u << 30; st.h1 = 2u << 30; st.h2 = 3u << 30;
  st.lit_pos = st.out_pos = st.sum_ml = st.emitted = 0;
  st.status = st.exe_status = ZG_OK;
  st.e_ll = st.e_of = st.e_ml = 0;
  st.P = 0;
  if (bs_len == 0 || lastb == 0) return false;
  st.P = (int32_t)(bs_len - 1) * 8 + (int32_t)(zg_hbit(lastb) - 1);
  return true;
}
// initial states, order LL, OF, ML (:164-166); w must cover the 26 bits below st.P
template <typename TabPtr>
ZG_HD void zg_seq_init_states(EmuSeqState& st, const ZgWin& w, TabPtr t_ll, unsigned ll_log, TabPtr t_of, unsigned of_log, TabPtr t_ml,
                              unsigned ml_log) {
  int32_t P = st.P;
  P -= ll_log; st.e_ll = t_ll[P >= 0 ? zg_win_bits(w, P, ll_log) : 0];
  P -= of_log; st.e_of = t_of[P >= 0 ? zg_win_bits(w, P, of_log) : 0];
  P -= ml_log; st.e_ml = t_ml[P >= 0 ? zg_win_bits(w, P, ml_log) : 0];
  st.P = P;   // may be negative: reported after the first sequence, like the reference
}
// bit position after the sequence the states currently describe (known before its bits are looked at)
ZG_HD int32_t zg_seq_next_pos(const EmuSeqState& st, bool last) {
  int32_t n = (int32_t)(ZG_FSE_XB(st.e_of) + ZG_FSE_XB(st.e_ml) + ZG_FSE_XB(st.e_ll));
  if (!last) n += (int32_t)(ZG_FSE_NB(st.e_ll) + ZG_FSE_NB(st.e_ml) + ZG_FSE_NB(st.e_of));
  return st.P - n;
}
// one sequence: cur covers the (up to 89) bits below st.P. Returns false when decoding must stop (bitstream error).
template <typename TabPtr, typename BasePtr>
ZG_HD bool zg_seq_step(EmuSeqState& st, const ZgWin& cur, bool last, TabPtr t_ll, TabPtr t_of, TabPtr t_ml, BasePtr ll_base, BasePtr ml_base,
                      uint32_t regen_size, EmuSeq* slot) {
  const uint32_t e_ll = st.e_ll, e_ml = st.e_ml, e_of = st.e_of;
  const unsigned of_code = ZG_FSE_SYM(e_of), ml_code = ZG_FSE_SYM(e_ml), ll_code = ZG_FSE_SYM(e_ll);
  const unsigned xb_of = ZG_FSE_XB(e_of), xb_ml = ZG_FSE_XB(e_ml), xb_ll = ZG_FSE_XB(e_ll);
  const unsigned nb_ll = last ? 0 : ZG_FSE_NB(e_ll), nb_ml = last ? 0 : ZG_FSE_NB(e_ml), nb_of = last ? 0 : ZG_FSE_NB(e_of);
  // positions of the six fields below P, in stream order OF, ML, LL extra bits (:185) then LL, ML, OF state bits (:204-206)
  const int32_t q_of = st.P - (int32_t)xb_of, q_ml = q_of - (int32_t)xb_ml, q_ll = q_ml - (int32_t)xb_ll;
  const int32_t q_sll = q_ll - (int32_t)nb_ll, q_sml = q_sll - (int32_t)nb_ml, q_sof = q_sml - (int32_t)nb_of;
  if (q_sof < 0) { st.status = ZG_SEQ_NOT_ENOUGH_BYTES; return false; }  // :209-211 (bits_remaining went negative)
  const uint32_t obits = zg_win_bits(cur, q_of, xb_of), ml_add = zg_win_bits(cur, q_ml, xb_ml), ll_add = zg_win_bits(cur, q_ll, xb_ll);
  if (!last) {
    st.e_ll = t_ll[ZG_FSE_BL(e_ll) + zg_win_bits(cur, q_sll, nb_ll)];
    st.e_ml = t_ml[ZG_FSE_BL(e_ml) + zg_win_bits(cur, q_sml, nb_ml)];
    st.e_of = t_of[ZG_FSE_BL(e_of) + zg_win_bits(cur, q_sof, nb_of)];
  }
  st.P = q_sof;
  const uint32_t of = obits + (1u << of_code);
  const uint32_t ml = ml_base[ml_code] + ml_add;
  const uint32_t ll = ll_base[ll_code] + ll_add;
  // execution bookkeeping (sequence_execution.rs:10-39). The reference decodes the whole section before executing,
  // so a bitstream error outranks these: keep decoding after the first execution error.
  if (st.exe_status == ZG_OK) {
    const bool big = of > 3u && of - 3u >= (1u << 30);
    uint32_t actual = zg_hist_step(big ? ZG_OFF_HUGE + 3u : of, ll, st.h0, st.h1, st.h2);
    if (actual == 0) st.exe_status = ZG_EXE_ZERO_OFFSET;
    else if ((uint64_t)st.lit_pos + ll > regen_size) st.exe_status = ZG_EXE_NOT_ENOUGH_LITERALS;
    else if ((uint64_t)st.out_pos + ll + ml >= (1ull << 31)) st.exe_status = ZG_UNSUPPORTED;
    EmuSeq q;
    q.of = actual; q.ml = ml; q.mdst = st.out_pos + ll; q.lit_start = st.lit_pos;
    *slot = q;
    if (st.exe_status == ZG_OK) { st.lit_pos += ll; st.out_pos += ll + ml; st.sum_ml += ml; st.emitted++; }
  }
  return true;
}
ZG_HD int zg_seq_finish(const EmuSeqState& st, ZgBlockSeqOut* sum) {
  int status = st.status;
  if (status == ZG_OK && st.P > 0) status = ZG_SEQ_EXTRA_BITS;  // :214-220
  if (status == ZG_OK) status = st.exe_status;
  sum->sum_ll = st.lit_pos; sum->sum_ml = st.sum_ml;
  sum->hist_end[0] = st.h0; sum->hist_end[1] = st.h1; sum->hist_end[2] = st.h2;
  sum->pad = st.exe_status ? st.emitted + 1u : 0u;      // like zg_k_seqpost: 1 + the sequence that cannot be executed
  return status;
}

// Same results as zg_seq_decode_block, built from the steps above with windows loaded straight from memory.
template <typename TabPtr>
ZG_HD int zg_seq_decode_block_fast(const uint8_t* bs, uint32_t bs_len, uint32_t nseq, TabPtr t_ll, unsigned ll_log, TabPtr t_of,
                                   unsigned of_log, TabPtr t_ml, unsigned ml_log, uint32_t regen_size, EmuSeq* out, ZgBlockSeqOut* sum) {
  EmuSeqState st;
  if (!zg_seq_begin(st, bs_len, bs_len ? bs[bs_len - 1] : 0)) return ZG_SEQ_EXTRA_PADDING;
  ZgWin w = zg_win_load(bs, st.P);
  zg_seq_init_states(st, w, t_ll, ll_log, t_of, of_log, t_ml, ml_log);
  w = zg_win_load(bs, st.P);
  for (uint32_t i = 0; i < nseq; i++) {
    const bool last = i + 1 == nseq;
    const ZgWin cur = w;
    w = zg_win_load(bs, zg_seq_next_pos(st, last));   // next sequence's window: its latency overlaps the work below
    if (!zg_seq_step(st, cur, last, t_ll, t_of, t_ml, (const uint32_t*)ZG_LL_BASE, (const uint32_t*)ZG_ML_BASE, regen_size, out + st.emitted)) break;
  }
  return zg_seq_finish(st, sum);
}
