// zg_dev.h — lane-level decode routines of the zgpu engine (one GPU lane runs one of these).
//
// Every routine is a plain function over pointers so that the HIP kernels (zg_kernels.hip) call them
// with LDS / global pointers on gfx950, and the CPU test harness (tests/emu) can run the very same
// code lane-by-lane on the host. Results must equal the reference's (ruzstd 0.9.1) for:
//   FSE table description + build   ruzstd/src/fse/fse_decoder.rs:116-366
//   Huffman weights + table build    ruzstd/src/huff0/huff0_decoder.rs:117-377
//   Huffman stream decode            ruzstd/src/decoding/literals_section_decoder.rs:94-147
//   sequence decode                  ruzstd/src/decoding/sequence_section_decoder.rs:14-221
//   offset history                   ruzstd/src/decoding/sequence_execution.rs:59-118
//   reversed bit reader              ruzstd/src/bit_io/bit_reader_reverse.rs:27-162
#pragma once
#include <stdint.h>
#include "zg_types.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZG_HD __host__ __device__ __forceinline__
#else
#define ZG_HD static inline __attribute__((always_inline))
#endif

// ---- value tables of the format (sequence_section_decoder.rs:227-284) -------------------------------------
static constexpr uint32_t ZG_LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                            48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static constexpr uint8_t ZG_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3,
                                           4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static constexpr uint32_t ZG_ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                            30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099,
                                            8195, 16387, 32771, 65539};
static constexpr uint8_t ZG_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                           0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
// predefined distributions (sequence_section_decoder.rs:413-442)
static constexpr int16_t ZG_LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static constexpr int16_t ZG_ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                              1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static constexpr int16_t ZG_OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

enum { ZG_KIND_LL = 0, ZG_KIND_OF = 1, ZG_KIND_ML = 2, ZG_KIND_HUFW = 3 };

// ---- raw memory access ------------------------------------------------------------------------------------
struct __attribute__((packed)) zg_u64u { uint64_t v; };
struct __attribute__((packed)) zg_u32u { uint32_t v; };
struct __attribute__((packed)) zg_u16u { uint16_t v; };
ZG_HD uint64_t zg_ld64(const uint8_t* p) { return ((const zg_u64u*)p)->v; }
ZG_HD uint32_t zg_ld32(const uint8_t* p) { return ((const zg_u32u*)p)->v; }
ZG_HD uint32_t zg_ld16(const uint8_t* p) { return ((const zg_u16u*)p)->v; }
ZG_HD unsigned zg_hbit(uint32_t x) { return 32u - (unsigned)__builtin_clz(x); }  // highest_bit_set, x > 0

// bits [lo, lo+n) of the stream seen as one little-endian integer, n <= 32, lo >= 0.
// Loads 8 bytes at s + lo/8: the engine pads the compressed buffer so this never leaves the allocation.
ZG_HD uint32_t zg_bits_at(const uint8_t* s, int32_t lo, unsigned n) {
  uint64_t w = zg_ld64(s + (lo >> 3)) >> (unsigned)(lo & 7);
  return (uint32_t)(w & ((1ull << n) - 1ull));
}
// same, but bits below index 0 read as zero (bit_reader_reverse.rs:66-86), lo may be negative
ZG_HD uint32_t zg_bits_at_z(const uint8_t* s, int32_t lo, unsigned n) {
  if (lo >= 0) return zg_bits_at(s, lo, n);
  int32_t m = (int32_t)n + lo;  // real bits available
  if (m <= 0) return 0;
  return zg_bits_at(s, 0, (unsigned)m) << (unsigned)(-lo);
}

// ---- forward bit reader for table descriptions (bit_io/bit_reader.rs:28-91) -----------------------------------
struct ZgFwdBits {
  const uint8_t* s;
  uint32_t nbits;  // 8 * len
  uint32_t idx;
};
ZG_HD bool zg_fb_get(ZgFwdBits& b, unsigned n, uint32_t& v) {  // n <= 16
  if (b.nbits - b.idx < n) return false;                      // NotEnoughRemainingBits
  // assemble from single bytes: the description may end flush with the compressed buffer
  uint32_t byte = b.idx >> 3, sh = b.idx & 7, w = 0, have = 0;
  uint32_t nbytes = (b.nbits >> 3);
  for (unsigned k = 0; k < 3; k++) {
    if (byte + k < nbytes) w |= (uint32_t)b.s[byte + k] << (8 * k);
    have += 8;
    if (have >= sh + n) break;
  }
  v = (w >> sh) & ((1u << n) - 1u);
  b.idx += n;
  return true;
}

// ---- FSE table description → probabilities (fse_decoder.rs:224-307) ------------------------------------------
// probs must hold 256 entries. Returns ZgStatus.
ZG_HD int zg_fse_read_probs(const uint8_t* src, uint32_t len, int max_log, int max_symbol, int16_t* probs, int* nprobs_out,
                            int* acc_log_out, uint32_t* bytes_used) {
  ZgFwdBits br{src, len * 8u, 0};
  uint32_t v;
  if (!zg_fb_get(br, 4, v)) return ZG_FSE_TABLE;
  int acc_log = 5 + (int)v;
  if (acc_log > max_log) return ZG_FSE_TABLE;  // AccLogTooBig
  uint32_t sum = 1u << acc_log, counter = 0;
  int n = 0;
  while (counter < sum) {
    uint32_t max_remaining = sum - counter + 1;
    unsigned bits = zg_hbit(max_remaining);
    if (!zg_fb_get(br, bits, v)) return ZG_FSE_TABLE;
    uint32_t low_threshold = ((1u << bits) - 1) - max_remaining;
    uint32_t mask = (1u << (bits - 1)) - 1;
    uint32_t small = v & mask, value;
    if (small < low_threshold) { br.idx -= 1; value = small; }
    else if (v > mask) value = v - low_threshold;
    else value = v;
    int prob = (int)value - 1;
    if (n > max_symbol) return ZG_FSE_TABLE;  // ends as TooManySymbols in the reference (:294-298)
    probs[n++] = (int16_t)prob;
    if (prob != 0) counter += prob > 0 ? (uint32_t)prob : 1u;
    else {
      for (;;) {
        if (!zg_fb_get(br, 2, v)) return ZG_FSE_TABLE;
        if (n + (int)v > max_symbol + 1) return ZG_FSE_TABLE;
        for (uint32_t i = 0; i < v; i++) probs[n++] = 0;
        if (v != 3) break;
      }
    }
  }
  if (counter != sum) return ZG_FSE_TABLE;  // ProbabilityCounterMismatch
  *nprobs_out = n;
  *acc_log_out = acc_log;
  *bytes_used = (br.idx + 7) >> 3;
  return ZG_OK;
}

ZG_HD uint32_t zg_fse_xbits(int kind, unsigned sym) {
  return kind == ZG_KIND_LL ? ZG_LL_BITS[sym < 36 ? sym : 0] : kind == ZG_KIND_ML ? ZG_ML_BITS[sym < 53 ? sym : 0] : kind == ZG_KIND_OF ? sym : 0u;
}
ZG_HD uint32_t zg_fse_pack(int kind, uint32_t bl, uint32_t nb, unsigned sym) {
  return kind == ZG_KIND_HUFW ? ZG_FSEW_PACK(bl, nb, sym) : ZG_FSE_PACK(bl, nb, sym, zg_fse_xbits(kind, sym));
}
ZG_HD unsigned zg_fse_sym_of(int kind, uint32_t e) { return kind == ZG_KIND_HUFW ? ZG_FSEW_SYM(e) : ZG_FSE_SYM(e); }

// ---- probabilities → decode table (fse_decoder.rs:141-220, :334-366) -------------------------------------------
// out holds 1<<acc_log packed entries; counter is scratch for max_symbol+1 u16.
ZG_HD int zg_fse_build(const int16_t* probs, int nprobs, int acc_log, int kind, uint32_t* out, uint16_t* counter) {
  uint32_t size = 1u << acc_log, neg = size;
  for (int s = 0; s < nprobs; s++) {
    counter[s] = 0;
    if (probs[s] == -1) {
      if (neg == 0) return ZG_INTERNAL;
      out[--neg] = (uint32_t)s;
    }
  }
  uint32_t pos = 0, step = (size >> 1) + (size >> 3) + 3, placed = 0;
  for (int s = 0; s < nprobs; s++) {
    if (probs[s] <= 0) continue;
    for (int i = 0; i < probs[s]; i++) {
      if (++placed > neg) return ZG_INTERNAL;
      out[pos] = (uint32_t)s;
      pos = (pos + step) & (size - 1);
      uint32_t guard = 0;
      while (pos >= neg) {
        pos = (pos + step) & (size - 1);
        if (++guard > size) return ZG_INTERNAL;
      }
    }
  }
  if (placed != neg) return ZG_INTERNAL;
  for (uint32_t i = 0; i < neg; i++) {
    unsigned s = out[i];
    uint32_t p = (uint32_t)probs[s], k = counter[s]++;
    unsigned hb = zg_hbit(p);
    uint32_t slices = ((1u << (hb - 1)) == p) ? p : (1u << hb);
    uint32_t dbl = slices - p, single = p - dbl, width = size / slices;
    uint32_t nb = zg_hbit(width) - 1, bl;
    if (k < dbl) { bl = single * width + k * width * 2; nb += 1; }
    else bl = (k - dbl) * width;
    out[i] = zg_fse_pack(kind, bl, nb, s);
  }
  for (uint32_t i = neg; i < size; i++) out[i] = zg_fse_pack(kind, 0, (uint32_t)acc_log, out[i]);
  return ZG_OK;
}

// ---- Huffman weights (huff0_decoder.rs:132-278) ---------------------------------------------------------------
// weights must hold 260 bytes; fse_tab is scratch for 64 packed entries; probs/counter scratch 256 each.
ZG_HD int zg_huf_read_weights(const uint8_t* src, uint32_t len, uint8_t* weights, int* nweights, uint32_t* bytes_used, uint32_t* fse_tab,
                              int16_t* probs, uint16_t* counter) {
  if (len == 0) return ZG_HUF_TABLE;
  uint32_t header = src[0];
  uint32_t bits_read = 8;
  if (header < 128) {
    const uint8_t* fs = src + 1;
    uint32_t flen = len - 1;
    if (header > flen) return ZG_HUF_TABLE;
    int np, al;
    uint32_t used;
    int st = zg_fse_read_probs(fs, flen, 6, 255, probs, &np, &al, &used);
    if (st) return ZG_HUF_TABLE;
    st = zg_fse_build(probs, np, al, ZG_KIND_HUFW, fse_tab, counter);
    if (st) return st == ZG_INTERNAL ? st : ZG_HUF_TABLE;
    if (used > header) return ZG_HUF_TABLE;
    uint32_t clen = header - used;
    const uint8_t* cs = fs + used;
    bits_read += (used + clen) * 8;
    // reversed reader over cs[0..clen): skip padding
    if (clen == 0) return ZG_HUF_TABLE;  // ExtraPadding: 9 zero bits read from an empty stream
    uint32_t lastb = cs[clen - 1];
    if (lastb == 0) return ZG_HUF_TABLE;
    int32_t P = (int32_t)(clen - 1) * 8 + (int32_t)(zg_hbit(lastb) - 1);
    uint32_t e1, e2;
    P -= al; e1 = fse_tab[zg_bits_at_z(cs, P, (unsigned)al)];
    P -= al; e2 = fse_tab[zg_bits_at_z(cs, P, (unsigned)al)];
    int n = 0;
    for (;;) {
      weights[n++] = (uint8_t)ZG_FSEW_SYM(e1);
      { unsigned nb = ZG_FSE_NB(e1); P -= nb; e1 = fse_tab[ZG_FSE_BL(e1) + zg_bits_at_z(cs, P, nb)]; }
      if (P <= -1) { weights[n++] = (uint8_t)ZG_FSEW_SYM(e2); break; }
      weights[n++] = (uint8_t)ZG_FSEW_SYM(e2);
      { unsigned nb = ZG_FSE_NB(e2); P -= nb; e2 = fse_tab[ZG_FSE_BL(e2) + zg_bits_at_z(cs, P, nb)]; }
      if (P <= -1) { weights[n++] = (uint8_t)ZG_FSEW_SYM(e1); break; }
      if (n > 255) return ZG_HUF_TABLE;  // TooManyWeights
    }
    *nweights = n;
  } else {
    uint32_t n = header - 127, need = (n + 1) / 2;
    if (len - 1 < need) return ZG_HUF_TABLE;
    for (uint32_t i = 0; i < n; i++) {
      uint8_t b = src[1 + i / 2];
      weights[i] = (i & 1) ? (b & 0xF) : (b >> 4);
      bits_read += 4;
    }
    *nweights = (int)n;
  }
  *bytes_used = (bits_read + 7) >> 3;
  return ZG_OK;
}

// ---- weights → Huffman decode table (huff0_decoder.rs:284-377) ------------------------------------------------
// out holds 2048 u16. Returns status; *max_bits_out is set as the reference sets max_num_bits.
ZG_HD int zg_huf_build(const uint8_t* weights, int nweights, uint16_t* out, int* max_bits_out) {
  uint32_t weight_sum = 0;
  for (int i = 0; i < nweights; i++) {
    unsigned w = weights[i];
    if (w > 11) return ZG_HUF_TABLE;
    weight_sum += w ? (1u << (w - 1)) : 0u;
  }
  if (weight_sum == 0) return ZG_HUF_TABLE;
  unsigned max_bits = zg_hbit(weight_sum);
  uint32_t left = (1u << max_bits) - weight_sum;
  if (left & (left - 1)) return ZG_HUF_TABLE;  // LeftoverIsNotAPowerOf2 (left >= 1 always)
  unsigned last_weight = zg_hbit(left);
  if (max_bits > 11) return ZG_HUF_TABLE;      // MaxBitsTooHigh
  *max_bits_out = (int)max_bits;
  uint32_t rank_count[13];
  for (int i = 0; i < 13; i++) rank_count[i] = 0;
  for (int s = 0; s <= nweights; s++) {
    unsigned w = s < nweights ? weights[s] : last_weight;
    unsigned b = w ? max_bits + 1 - w : 0;
    rank_count[b]++;
  }
  uint32_t rank_idx[13];
  rank_idx[max_bits] = 0;
  for (unsigned b = max_bits; b >= 1; b--) rank_idx[b - 1] = rank_idx[b] + rank_count[b] * (1u << (max_bits - b));
  uint32_t size = 1u << max_bits;
  if (rank_idx[0] != size) return ZG_INTERNAL;  // assert :353-358
  for (int s = 0; s <= nweights; s++) {
    unsigned w = s < nweights ? weights[s] : last_weight;
    if (!w) continue;
    unsigned b = max_bits + 1 - w;
    uint32_t base = rank_idx[b], n = 1u << (max_bits - b);
    rank_idx[b] += n;
    uint16_t e = ZG_HUF_PACK((unsigned)(s & 255), b);
    for (uint32_t i = 0; i < n; i++) out[base + i] = e;
  }
  return ZG_OK;
}

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

// ---- offset history, symbolic form ---------------------------------------------------------------------------------
// A history slot / resolved offset is a u32: top two bits 0 → a concrete offset (< 2^30); top two bits t in 1..3 →
// "slot t-1 of the block's initial history, minus k (saturating)" with k in the low 30 bits. Blocks decode their
// sequences in parallel before the previous block's final history is known; the scan kernel resolves the symbols.
#define ZG_SYM_TAG(v) ((v) >> 30)
#define ZG_SYM_K(v) ((v) & 0x3FFFFFFFu)
ZG_HD uint32_t zg_sym_dec(uint32_t v) {  // saturating "minus one" (sequence_execution.rs:74)
  if (ZG_SYM_TAG(v)) return v + 1;
  return v ? v - 1 : 0;
}
ZG_HD uint32_t zg_sym_resolve(uint32_t v, const uint32_t* h) {
  uint32_t t = ZG_SYM_TAG(v);
  if (!t) return v;
  uint32_t x = h[t - 1], k = ZG_SYM_K(v);
  return x > k ? x - k : 0;
}
// do_offset_history (sequence_execution.rs:59-118) on symbolic slots h0..h2; returns the (symbolic) actual offset.
ZG_HD uint32_t zg_hist_step(uint32_t of, uint32_t ll, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
  uint32_t actual;
  if (ll > 0) {
    if (of == 1) return h0;
    if (of == 2) { actual = h1; h1 = h0; h0 = actual; return actual; }
    actual = of == 3 ? h2 : of - 3;
  } else {
    if (of == 1) { actual = h1; h1 = h0; h0 = actual; return actual; }
    actual = of == 2 ? h2 : of == 3 ? zg_sym_dec(h0) : of - 3;
  }
  h2 = h1; h1 = h0; h0 = actual;
  return actual;
}

// ---- the sequence section of one block (sequence_section_decoder.rs:14-221) --------------------------------------
// bs[0..bs_len) is the reversed bitstream (after the table descriptions). Tables are packed entries; a log of 0
// means a one-entry table (RLE mode, or a carried RLE symbol). Writes nseq ZgSeq and the block summary.
template <typename TabPtr>
ZG_HD int zg_seq_decode_block(const uint8_t* bs, uint32_t bs_len, uint32_t nseq, TabPtr t_ll, unsigned ll_log, TabPtr t_of, unsigned of_log,
                              TabPtr t_ml, unsigned ml_log, uint32_t regen_size, ZgSeq* out, ZgBlockSeqOut* sum) {
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
      uint32_t actual = zg_hist_step(of, ll, h0, h1, h2);
      if (actual == 0) exe_status = ZG_EXE_ZERO_OFFSET;
      else if (!ZG_SYM_TAG(actual) && actual >= (1u << 30)) exe_status = ZG_EXE_OFFSET_TOO_BIG;
      else if ((uint64_t)lit_pos + ll > regen_size) exe_status = ZG_EXE_NOT_ENOUGH_LITERALS;
      else if ((uint64_t)out_pos + ll + ml >= (1ull << 31)) exe_status = ZG_UNSUPPORTED;
      else {
        ZgSeq q;
        q.of = actual; q.ml = ml; q.mdst = out_pos + ll; q.lit_start = lit_pos;
        out[i] = q;
        lit_pos += ll; out_pos += ll + ml; sum_ml += ml;
      }
    }
  }
  if (status == ZG_OK && P > 0) status = ZG_SEQ_EXTRA_BITS;  // :214-220
  if (status == ZG_OK) status = exe_status;
  sum->sum_ll = lit_pos; sum->sum_ml = sum_ml;
  sum->hist_end[0] = h0; sum->hist_end[1] = h1; sum->hist_end[2] = h2;
  sum->pad = 0;
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
struct ZgSeqState {
  uint32_t e_ll, e_of, e_ml;     // current table entries of the three FSE states
  int32_t P;                     // bits_remaining (bit_reader_reverse.rs:27-29)
  uint32_t h0, h1, h2;           // symbolic offset history
  uint32_t lit_pos, out_pos, sum_ml, emitted;
  int status, exe_status;
};
// skip the padding of the last byte; false = ExtraPadding (sequence_section_decoder.rs:29-40)
ZG_HD bool zg_seq_begin(ZgSeqState& st, uint32_t bs_len, uint32_t lastb) {
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
ZG_HD void zg_seq_init_states(ZgSeqState& st, const ZgWin& w, TabPtr t_ll, unsigned ll_log, TabPtr t_of, unsigned of_log, TabPtr t_ml,
                              unsigned ml_log) {
  int32_t P = st.P;
  P -= ll_log; st.e_ll = t_ll[P >= 0 ? zg_win_bits(w, P, ll_log) : 0];
  P -= of_log; st.e_of = t_of[P >= 0 ? zg_win_bits(w, P, of_log) : 0];
  P -= ml_log; st.e_ml = t_ml[P >= 0 ? zg_win_bits(w, P, ml_log) : 0];
  st.P = P;   // may be negative: reported after the first sequence, like the reference
}
// bit position after the sequence the states currently describe (known before its bits are looked at)
ZG_HD int32_t zg_seq_next_pos(const ZgSeqState& st, bool last) {
  int32_t n = (int32_t)(ZG_FSE_XB(st.e_of) + ZG_FSE_XB(st.e_ml) + ZG_FSE_XB(st.e_ll));
  if (!last) n += (int32_t)(ZG_FSE_NB(st.e_ll) + ZG_FSE_NB(st.e_ml) + ZG_FSE_NB(st.e_of));
  return st.P - n;
}
// one sequence: cur covers the (up to 89) bits below st.P. Returns false when decoding must stop (bitstream error).
template <typename TabPtr, typename BasePtr>
ZG_HD bool zg_seq_step(ZgSeqState& st, const ZgWin& cur, bool last, TabPtr t_ll, TabPtr t_of, TabPtr t_ml, BasePtr ll_base, BasePtr ml_base,
                      uint32_t regen_size, ZgSeq* slot) {
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
    uint32_t actual = zg_hist_step(of, ll, st.h0, st.h1, st.h2);
    if (actual == 0) st.exe_status = ZG_EXE_ZERO_OFFSET;
    else if (!ZG_SYM_TAG(actual) && actual >= (1u << 30)) st.exe_status = ZG_EXE_OFFSET_TOO_BIG;
    else if ((uint64_t)st.lit_pos + ll > regen_size) st.exe_status = ZG_EXE_NOT_ENOUGH_LITERALS;
    else if ((uint64_t)st.out_pos + ll + ml >= (1ull << 31)) st.exe_status = ZG_UNSUPPORTED;
    else {
      ZgSeq q;
      q.of = actual; q.ml = ml; q.mdst = st.out_pos + ll; q.lit_start = st.lit_pos;
      *slot = q;
      st.lit_pos += ll; st.out_pos += ll + ml; st.sum_ml += ml; st.emitted++;
    }
  }
  return true;
}
ZG_HD int zg_seq_finish(const ZgSeqState& st, ZgBlockSeqOut* sum) {
  int status = st.status;
  if (status == ZG_OK && st.P > 0) status = ZG_SEQ_EXTRA_BITS;  // :214-220
  if (status == ZG_OK) status = st.exe_status;
  sum->sum_ll = st.lit_pos; sum->sum_ml = st.sum_ml;
  sum->hist_end[0] = st.h0; sum->hist_end[1] = st.h1; sum->hist_end[2] = st.h2;
  sum->pad = 0;
  return status;
}

// Same results as zg_seq_decode_block, built from the steps above with windows loaded straight from memory.
template <typename TabPtr>
ZG_HD int zg_seq_decode_block_fast(const uint8_t* bs, uint32_t bs_len, uint32_t nseq, TabPtr t_ll, unsigned ll_log, TabPtr t_of,
                                   unsigned of_log, TabPtr t_ml, unsigned ml_log, uint32_t regen_size, ZgSeq* out, ZgBlockSeqOut* sum) {
  ZgSeqState st;
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
