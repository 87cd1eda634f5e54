// zg_dev.h — lane-level decode routines of the zgpu engine (one GPU lane runs one of these).
//
// Every routine is a plain function over pointers so that the HIP kernels (zg_kernels.hip) call them
// with LDS / global pointers on gfx950, the dictionary parser (zg_capi.cpp) calls them on the host, and the CPU
// test harness (tests/emu) can run the very same code on the host. Everything in this file is called by the
// product. Results must equal the reference's (ruzstd 0.9.1) for:
//   FSE table description + build   ruzstd/src/fse/fse_decoder.rs:116-366
//   Huffman weights + table build    ruzstd/src/huff0/huff0_decoder.rs:117-377
//   offset history                   ruzstd/src/decoding/sequence_execution.rs:59-118
//   forward bit reader               ruzstd/src/bit_io/bit_reader.rs:28-91
// (The wave-cooperative stream and sequence decoders live in zg_kernels.hip; tests/emu/zg_emu_serial.h holds a
// serial model of them for the CPU tests.)
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

// ---- offset history, symbolic form ---------------------------------------------------------------------------------
// A history slot / resolved offset is a u32: top two bits 0 → a concrete offset (< 2^30); top two bits t in 1..3 →
// "slot t-1 of the block's initial history, minus k (saturating)" with k in the low 30 bits. Blocks decode their
// sequences in parallel before the previous block's final history is known; the scan kernel resolves the symbols.
#define ZG_OFF_HUGE 0x3FFFFFFFu   // stands for every offset >= 2^30 (the largest real one below that is 2^30 - 4): see zg_k_seqpost
#define ZG_SYM_TAG(v) ((v) >> 30)
#define ZG_SYM_K(v) ((v) & 0x3FFFFFFFu)
ZG_HD uint32_t zg_sym_dec(uint32_t v) {  // saturating "minus one" (sequence_execution.rs:74); selects only (the callers are wave code)
  const uint32_t c = v ? v - 1 : 0u;
  return ZG_SYM_TAG(v) ? v + 1 : c;
}
ZG_HD uint32_t zg_sym_resolve(uint32_t v, const uint32_t* h) {
  uint32_t t = ZG_SYM_TAG(v);
  if (!t) return v;
  const uint32_t x = t == 1 ? h[0] : t == 2 ? h[1] : h[2], k = ZG_SYM_K(v);   // selects: a dynamic index would put h into scratch memory
  return x > k ? x - k : 0;
}
// do_offset_history (sequence_execution.rs:59-118) on symbolic slots h0..h2; returns the (symbolic) actual offset.
ZG_HD uint32_t zg_hist_step(uint32_t of, uint32_t ll, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
  // Without literals the repeat codes shift by one (:68-81): c is the code as if literals were present, 4 = "slot 0 minus one".
  // Written with selects: every lane of a wave takes the same instructions.
  const bool isnew = of > 3u;
  const uint32_t c = ll ? of : of + 1u;
  const uint32_t rep = c == 1u ? h0 : c == 2u ? h1 : c == 3u ? h2 : zg_sym_dec(h0);
  const uint32_t actual = isnew ? of - 3u : rep;
  const bool rot = isnew || c >= 3u;                  // slots move down by one, the offset enters at the front
  const bool moved = rot || c == 2u;                  // c == 2: slots 0 and 1 swap; c == 1: nothing changes (actual == h0)
  h2 = rot ? h1 : h2;
  h1 = moved ? h0 : h1;
  h0 = actual;
  return actual;
}
