/*
 * zstd_oracle.c — CPU oracle: a plain-C restatement of ruzstd 0.9.1's decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see zstd_oracle.h). Not part of the product; never linked
 * into libzgpu.so. Parity status: PINNED against the reference's golden fixtures.
 *
 * Citations are `file:line` relative to /root/reference/ruzstd/src.
 * Where the reference would panic (assert!/unreachable!/slice index), this oracle
 * returns ZOR_REF_PANIC instead of crashing.
 */
#include "zstd_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------------- */
#define MAGIC_NUM 0xFD2FB528u                 /* common/mod.rs:6 */
#define MIN_WINDOW_SIZE 1024ull               /* common/mod.rs:10 */
#define MAX_WINDOW_SIZE ((1ull << 41) + 7ull * (1ull << 38)) /* common/mod.rs:14 */
#define MAX_BLOCK_SIZE (128u * 1024u)         /* common/mod.rs:21 */
#define DEFAULT_MAX_WINDOW_SIZE (1024ull * 1024ull * 128ull) /* frame_decoder.rs:25 */
#define MAX_LITERAL_LENGTH_CODE 35            /* blocks/sequence_section.rs:6 */
#define MAX_MATCH_LENGTH_CODE 52              /* :7 */
#define MAX_OFFSET_CODE 31                    /* :8 */
#define MAX_MAX_NUM_BITS 11                   /* huff0/huff0_decoder.rs:9 */
#define LL_MAX_LOG 9                          /* decoding/sequence_section_decoder.rs:288 */
#define ML_MAX_LOG 9                          /* :290 */
#define OF_MAX_LOG 8                          /* :292 */

static unsigned highest_bit_set(uint32_t x) { /* fse_decoder.rs:326-329 (x > 0) */
  return 32u - (unsigned)__builtin_clz(x);
}

/* ------------------------------------------------------------------------------------------
 * XXH64 — public algorithm; the reference uses twox_hash::XxHash64::with_seed(0)
 * (decode_buffer.rs:16,42,54) and feeds drained bytes (:223-227,290,301).
 * Streaming state so bytes can be fed as they are drained.
 * ---------------------------------------------------------------------------------------- */
#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
static uint64_t xmerge(uint64_t acc, uint64_t v) { v = xround(0, v); acc ^= v; return acc * XP1 + XP4; }
typedef struct { uint64_t v[4]; uint8_t mem[32]; unsigned memsize; uint64_t total; uint64_t seed; } xxh64_state;
static void xxh64_reset(xxh64_state* s, uint64_t seed) {
  s->seed = seed; s->v[0] = seed + XP1 + XP2; s->v[1] = seed + XP2; s->v[2] = seed; s->v[3] = seed - XP1;
  s->memsize = 0; s->total = 0;
}
static void xxh64_update(xxh64_state* s, const uint8_t* p, size_t len) {
  s->total += len;
  if (s->memsize + len < 32) { memcpy(s->mem + s->memsize, p, len); s->memsize += (unsigned)len; return; }
  const uint8_t* end = p + len;
  if (s->memsize) {
    size_t fill = 32 - s->memsize; memcpy(s->mem + s->memsize, p, fill);
    for (int i = 0; i < 4; i++) s->v[i] = xround(s->v[i], rd64(s->mem + 8 * i));
    p += fill; s->memsize = 0;
  }
  while (p + 32 <= end) { for (int i = 0; i < 4; i++) s->v[i] = xround(s->v[i], rd64(p + 8 * i)); p += 32; }
  if (p < end) { memcpy(s->mem, p, (size_t)(end - p)); s->memsize = (unsigned)(end - p); }
}
static uint64_t xxh64_digest(const xxh64_state* s) {
  uint64_t h;
  if (s->total >= 32) {
    h = rotl64(s->v[0], 1) + rotl64(s->v[1], 7) + rotl64(s->v[2], 12) + rotl64(s->v[3], 18);
    for (int i = 0; i < 4; i++) h = xmerge(h, s->v[i]);
  } else h = s->seed + XP5;
  h += s->total;
  const uint8_t* p = s->mem; const uint8_t* end = p + s->memsize;
  while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
  while (p < end) { h ^= (*p) * XP5; h = rotl64(h, 11) * XP1; p++; }
  h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
  return h;
}
uint64_t zor_xxh64(const uint8_t* p, size_t len, uint64_t seed) {
  xxh64_state s; xxh64_reset(&s, seed); xxh64_update(&s, p, len); return xxh64_digest(&s);
}

/* ------------------------------------------------------------------------------------------
 * BitReaderReversed — bit_io/bit_reader_reverse.rs
 * Restated on the conceptual model the container code implements: the stream is
 * one little-endian integer; bits are taken from the top; reading past bit 0 yields zeros and
 * is counted, so bits_remaining() (:27-29) goes negative.
 * ---------------------------------------------------------------------------------------- */
typedef struct { const uint8_t* src; size_t len; int64_t pos; /* == bits_remaining() */ } revbits;
static void rb_init(revbits* r, const uint8_t* src, size_t len) { /* new :31-39 */
  r->src = src; r->len = len; r->pos = (int64_t)len * 8;
}
static uint64_t rb_load64(const revbits* r, size_t byte) { /* zero-padded LE load (refill :57-65) */
  uint64_t v = 0;
  if (byte + 8 <= r->len) memcpy(&v, r->src + byte, 8);
  else if (byte < r->len) memcpy(&v, r->src + byte, r->len - byte);
  return v;
}
static uint64_t rb_get(revbits* r, unsigned n) { /* get_bits :92-100, peek_bits :105-113 ; n <= 56 */
  if (n == 0) return 0;
  int64_t P = r->pos, lo = P - (int64_t)n;
  r->pos = lo;
  uint64_t mask = (1ull << n) - 1ull;
  if (lo >= 0) return (rb_load64(r, (size_t)(lo >> 3)) >> (lo & 7)) & mask;
  if (P <= 0) return 0;                       /* only zeros left (refill :76-86) */
  uint64_t v = rb_load64(r, 0) & ((1ull << P) - 1ull); /* P < n <= 56 */
  return (v << (unsigned)(-lo)) & mask;
}
static int64_t rb_remaining(const revbits* r) { return r->pos; } /* bits_remaining :27-29 */

int64_t zor_revbits_read(const uint8_t* src, size_t len, const uint8_t* widths, size_t n, uint64_t* values) {
  revbits r; rb_init(&r, src, len);
  for (size_t i = 0; i < n; i++) values[i] = rb_get(&r, widths[i]);
  return rb_remaining(&r);
}

/* ------------------------------------------------------------------------------------------
 * BitReader (forward, LSB first, hard bounds) — bit_io/bit_reader.rs:28-91
 * ---------------------------------------------------------------------------------------- */
typedef struct { const uint8_t* src; size_t len; size_t idx; } fwdbits;
static int fb_get(fwdbits* b, unsigned n, uint64_t* out) { /* get_bits :28-91 */
  if (n > 64) return -1;
  if (b->len * 8 - b->idx < n) return -1;  /* NotEnoughRemainingBits :35-40 */
  uint64_t v = 0;
  for (unsigned i = 0; i < n; i++) {
    size_t bit = b->idx + i;
    v |= (uint64_t)((b->src[bit >> 3] >> (bit & 7)) & 1u) << i;
  }
  b->idx += n; *out = v; return 0;
}
int zor_fwdbits_read(const uint8_t* src, size_t len, const uint8_t* widths, size_t n, uint64_t* values) {
  fwdbits b = {src, len, 0};
  for (size_t i = 0; i < n; i++) if (fb_get(&b, widths[i], &values[i])) return -1;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * FSE table — fse/fse_decoder.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int max_symbol;                /* :61 */
  zor_fse_entry decode[512];     /* :64 (≤ 1<<9) */
  size_t decode_len;
  int accuracy_log;              /* :68 */
  int32_t probs[512];            /* symbol_probabilities :79 (bounded: see read_probabilities) */
  size_t nprobs;
} fse_table;

static void fse_reset(fse_table* t) { t->decode_len = 0; t->accuracy_log = 0; t->nprobs = 0; } /* reset :108-113 */

static void calc_baseline_and_numbits(uint32_t total, uint32_t nsym, uint32_t k, uint32_t* bl, uint8_t* nb) {
  /* fse_decoder.rs:340-366 */
  if (nsym == 0) { *bl = 0; *nb = 0; return; }
  uint32_t slices = ((1u << (highest_bit_set(nsym) - 1)) == nsym) ? nsym : (1u << highest_bit_set(nsym));
  uint32_t dbl = slices - nsym, single = nsym - dbl, width = total / slices;
  uint32_t bits = highest_bit_set(width) - 1;
  if (k < dbl) { *bl = single * width + k * width * 2; *nb = (uint8_t)(bits + 1); }
  else { *bl = (k - dbl) * width; *nb = (uint8_t)bits; }
}

static int fse_build_decoding_table(fse_table* t) { /* build_decoding_table :141-220 */
  if (t->nprobs > (size_t)t->max_symbol + 1) return ZOR_FSE_TABLE;  /* TooManySymbols :142-146 */
  size_t size = (size_t)1 << t->accuracy_log;
  if (size > 512) return ZOR_REF_PANIC;
  memset(t->decode, 0, sizeof(zor_fse_entry) * size);
  t->decode_len = size;
  size_t negative_idx = size;
  for (size_t s = 0; s < t->nprobs; s++) {   /* -1 symbols at the top :167-175 */
    if (t->probs[s] == -1) {
      if (negative_idx == 0) return ZOR_REF_PANIC;
      negative_idx--;
      t->decode[negative_idx].symbol = (uint8_t)s;
      t->decode[negative_idx].base_line = 0;
      t->decode[negative_idx].num_bits = (uint8_t)t->accuracy_log;
    }
  }
  size_t position = 0;                       /* spread :178-197 */
  size_t step = (size >> 1) + (size >> 3) + 3; /* next_position :334-338 */
  size_t placed = 0;
  for (size_t s = 0; s < t->nprobs; s++) {
    if (t->probs[s] <= 0) continue;
    for (int32_t i = 0; i < t->probs[s]; i++) {
      if (position >= size || ++placed > negative_idx) return ZOR_REF_PANIC; /* malformed distributions only */
      t->decode[position].symbol = (uint8_t)s;
      position = (position + step) & (size - 1);
      size_t guard = 0;
      while (position >= negative_idx) {
        position = (position + step) & (size - 1);
        if (++guard > size) return ZOR_REF_PANIC; /* reference would spin forever */
      }
    }
  }
  uint32_t counter[512]; memset(counter, 0, sizeof(uint32_t) * (t->nprobs ? t->nprobs : 1)); /* :200-203 */
  for (size_t i = 0; i < negative_idx; i++) { /* :204-218 */
    uint8_t sym = t->decode[i].symbol;
    if (sym >= t->nprobs) return ZOR_REF_PANIC;
    int32_t prob = t->probs[sym];
    uint32_t bl; uint8_t nb;
    calc_baseline_and_numbits((uint32_t)size, (uint32_t)prob, counter[sym], &bl, &nb);
    if (nb > t->accuracy_log) return ZOR_REF_PANIC; /* assert :213 */
    counter[sym]++;
    t->decode[i].base_line = bl; t->decode[i].num_bits = nb;
  }
  return ZOR_OK;
}

static int fse_read_probabilities(fse_table* t, const uint8_t* src, size_t len, int max_log, size_t* bytes_read) {
  /* read_probabilities :224-307 */
  t->nprobs = 0;
  fwdbits br = {src, len, 0};
  uint64_t v;
  if (fb_get(&br, 4, &v)) return ZOR_FSE_TABLE;
  t->accuracy_log = 5 + (int)v;                       /* ACC_LOG_OFFSET :324 */
  if (t->accuracy_log > max_log) return ZOR_FSE_TABLE; /* AccLogTooBig :229-234 */
  if (t->accuracy_log == 0) return ZOR_FSE_TABLE;
  uint32_t sum = 1u << t->accuracy_log, counter = 0;
  while (counter < sum) {                              /* :242-286 */
    uint32_t max_remaining = sum - counter + 1;
    unsigned bits = highest_bit_set(max_remaining);
    if (fb_get(&br, bits, &v)) return ZOR_FSE_TABLE;
    uint32_t unchecked = (uint32_t)v;
    uint32_t low_threshold = ((1u << bits) - 1) - max_remaining;
    uint32_t mask = (1u << (bits - 1)) - 1;
    uint32_t small = unchecked & mask;
    uint32_t value;
    if (small < low_threshold) { br.idx -= 1; value = small; }   /* return_bits(1) :252-254 */
    else if (unchecked > mask) value = unchecked - low_threshold;
    else value = unchecked;
    int32_t prob = (int32_t)value - 1;
    if (t->nprobs >= 512) return ZOR_FSE_TABLE;        /* would end as TooManySymbols :294-298 */
    t->probs[t->nprobs++] = prob;
    if (prob != 0) {
      if (prob > 0) counter += (uint32_t)prob; else counter += 1;  /* :266-272 */
    } else {
      for (;;) {                                        /* zero-run flags :274-284 */
        if (fb_get(&br, 2, &v)) return ZOR_FSE_TABLE;
        size_t skip = (size_t)v;
        if (t->nprobs + skip > 512) return ZOR_FSE_TABLE;
        for (size_t i = 0; i < skip; i++) t->probs[t->nprobs++] = 0;
        if (skip != 3) break;
      }
    }
  }
  if (counter != sum) return ZOR_FSE_TABLE;            /* ProbabilityCounterMismatch :288-293 */
  if (t->nprobs > (size_t)t->max_symbol + 1) return ZOR_FSE_TABLE; /* TooManySymbols :294-298 */
  *bytes_read = (br.idx + 7) / 8;                      /* :300-304 */
  return ZOR_OK;
}

static int fse_build_decoder(fse_table* t, const uint8_t* src, size_t len, int max_log, size_t* bytes_read) {
  /* build_decoder :116-124 */
  t->accuracy_log = 0;
  int st = fse_read_probabilities(t, src, len, max_log, bytes_read);
  if (st) return st;
  return fse_build_decoding_table(t);
}
static int fse_build_from_probabilities(fse_table* t, int acc_log, const int32_t* probs, size_t n) {
  /* build_from_probabilities :126-139 */
  if (acc_log == 0) return ZOR_FSE_TABLE;
  if (n > 512) return ZOR_FSE_TABLE;
  memcpy(t->probs, probs, n * sizeof(int32_t)); t->nprobs = n;
  t->accuracy_log = acc_log;
  return fse_build_decoding_table(t);
}

/* FSEDecoder — fse_decoder.rs:5-52 */
typedef struct { zor_fse_entry state; const fse_table* table; } fse_decoder;
static void fsed_new(fse_decoder* d, const fse_table* t) { /* new :14-24 */
  d->table = t;
  if (t->decode_len) d->state = t->decode[0]; else memset(&d->state, 0, sizeof d->state);
}
static int fsed_init_state(fse_decoder* d, revbits* br) { /* init_state :32-40 */
  if (d->table->accuracy_log == 0) return ZOR_FSE_UNINIT;
  uint64_t s = rb_get(br, (unsigned)d->table->accuracy_log);
  if (s >= d->table->decode_len) return ZOR_REF_PANIC;
  d->state = d->table->decode[s];
  return ZOR_OK;
}
static int fsed_update_state(fse_decoder* d, revbits* br) { /* update_state :43-51 */
  uint64_t add = rb_get(br, d->state.num_bits);
  uint64_t ns = (uint64_t)d->state.base_line + add;
  if (ns >= d->table->decode_len) return ZOR_REF_PANIC;
  d->state = d->table->decode[ns];
  return ZOR_OK;
}

int zor_fse_build_from_probs(int acc_log, const int32_t* probs, size_t nprobs, int max_symbol, zor_fse_entry* out) {
  fse_table t; fse_reset(&t); t.max_symbol = max_symbol;
  int st = fse_build_from_probabilities(&t, acc_log, probs, nprobs);
  if (st) return st;
  memcpy(out, t.decode, sizeof(zor_fse_entry) * t.decode_len);
  return ZOR_OK;
}
int zor_fse_build_decoder(const uint8_t* src, size_t len, int max_log, int max_symbol,
                          zor_fse_entry* out, int* acc_log, size_t* bytes_read) {
  fse_table t; fse_reset(&t); t.max_symbol = max_symbol;
  int st = fse_build_decoder(&t, src, len, max_log, bytes_read);
  if (st) return st;
  memcpy(out, t.decode, sizeof(zor_fse_entry) * t.decode_len);
  *acc_log = t.accuracy_log;
  return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * Huffman table — huff0/huff0_decoder.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  zor_huf_entry decode[1 << MAX_MAX_NUM_BITS]; size_t decode_len;   /* :58 */
  uint8_t weights[260]; size_t nweights;                            /* :63 */
  int max_num_bits;                                                 /* :68 */
  fse_table fse;                                                    /* :73 */
} huf_table;
static void huf_reset(huf_table* h) { h->decode_len = 0; h->nweights = 0; h->max_num_bits = 0; fse_reset(&h->fse); h->fse.max_symbol = 255; }

static int huf_read_weights(huf_table* h, const uint8_t* src, size_t len, uint32_t* bytes_read) {
  /* read_weights :132-278 */
  if (len == 0) return ZOR_HUF_TABLE;                  /* SourceIsEmpty :135-137 */
  uint8_t header = src[0];
  size_t bits_read = 8;
  if (header < 128) {                                  /* FSE-compressed weights :145-242 */
    const uint8_t* fse_stream = src + 1; size_t fse_len = len - 1;
    if (header > fse_len) return ZOR_HUF_TABLE;        /* NotEnoughBytesForWeights :147-152 */
    size_t used;
    int st = fse_build_decoder(&h->fse, fse_stream, fse_len, 6, &used); /* :154 */
    if (st) return st == ZOR_REF_PANIC ? st : ZOR_HUF_TABLE;
    if (used > header) return ZOR_HUF_TABLE;           /* FSETableUsedTooManyBytes :156-161 */
    fse_decoder dec1, dec2; fsed_new(&dec1, &h->fse); fsed_new(&dec2, &h->fse); /* :170-171 */
    size_t clen = (size_t)header - used;               /* :174 */
    if (fse_len - used < clen) return ZOR_HUF_TABLE;   /* :177-182 */
    revbits br; rb_init(&br, fse_stream + used, clen); /* :183-184 */
    bits_read += (used + clen) * 8;                    /* :186 */
    int skipped = 0;                                   /* padding skip :189-200 */
    for (;;) { uint64_t v = rb_get(&br, 1); skipped++; if (v == 1 || skipped > 8) break; }
    if (skipped > 8) return ZOR_HUF_TABLE;             /* ExtraPadding */
    if ((st = fsed_init_state(&dec1, &br))) return st == ZOR_REF_PANIC ? st : ZOR_HUF_TABLE; /* :202 */
    if ((st = fsed_init_state(&dec2, &br))) return st == ZOR_REF_PANIC ? st : ZOR_HUF_TABLE; /* :203 */
    h->nweights = 0;                                   /* :205 */
    for (;;) {                                         /* :208-241 */
      h->weights[h->nweights++] = dec1.state.symbol;
      if ((st = fsed_update_state(&dec1, &br))) return st;
      if (rb_remaining(&br) <= -1) { h->weights[h->nweights++] = dec2.state.symbol; break; }
      h->weights[h->nweights++] = dec2.state.symbol;
      if ((st = fsed_update_state(&dec2, &br))) return st;
      if (rb_remaining(&br) <= -1) { h->weights[h->nweights++] = dec1.state.symbol; break; }
      if (h->nweights > 255) return ZOR_HUF_TABLE;     /* TooManyWeights :236-240 */
    }
  } else {                                             /* direct 4-bit weights :250-269 */
    const uint8_t* raw = src + 1; size_t rawlen = len - 1;
    unsigned n = (unsigned)header - 127;
    size_t need = (n + 1) / 2;
    if (rawlen < need) return ZOR_HUF_TABLE;           /* NotEnoughBytesInSource :256-261 */
    h->nweights = n;
    for (unsigned i = 0; i < n; i++) {
      h->weights[i] = (i % 2 == 0) ? (raw[i / 2] >> 4) : (raw[i / 2] & 0xF);
      bits_read += 4;
    }
  }
  *bytes_read = (uint32_t)((bits_read + 7) / 8);       /* :272-277 */
  return ZOR_OK;
}

static int huf_build_table_from_weights(huf_table* h) { /* build_table_from_weights :284-377 */
  uint8_t bits[262]; size_t nbits = h->nweights + 1;
  memset(bits, 0, nbits);
  uint32_t weight_sum = 0;
  for (size_t i = 0; i < h->nweights; i++) {
    uint8_t w = h->weights[i];
    if (w > MAX_MAX_NUM_BITS) return ZOR_HUF_TABLE;    /* WeightBiggerThanMaxNumBits :292-294 */
    weight_sum += w > 0 ? (1u << (w - 1)) : 0;
  }
  if (weight_sum == 0) return ZOR_HUF_TABLE;           /* MissingWeights :298-300 */
  unsigned max_bits = highest_bit_set(weight_sum);
  uint32_t left_over = (1u << max_bits) - weight_sum;
  if (left_over == 0 || (left_over & (left_over - 1))) return ZOR_HUF_TABLE; /* LeftoverIsNotAPowerOf2 :306-308 */
  unsigned last_weight = highest_bit_set(left_over);
  for (size_t s = 0; s < h->nweights; s++)
    bits[s] = h->weights[s] > 0 ? (uint8_t)(max_bits + 1 - h->weights[s]) : 0;
  bits[h->nweights] = (uint8_t)(max_bits + 1 - last_weight);
  h->max_num_bits = (int)max_bits;                     /* :322 (set before the check, as the reference does) */
  if (max_bits > MAX_MAX_NUM_BITS) return ZOR_HUF_TABLE; /* MaxBitsTooHigh :324-326 */
  uint32_t bit_ranks[MAX_MAX_NUM_BITS + 2]; memset(bit_ranks, 0, sizeof bit_ranks);
  for (size_t i = 0; i < nbits; i++) {
    if (bits[i] > max_bits) return ZOR_REF_PANIC;      /* index out of bounds in the reference */
    bit_ranks[bits[i]]++;
  }
  size_t size = (size_t)1 << max_bits;
  /* decode.resize(size): the vector was cleared by build_decoder :118 → zero-filled */
  memset(h->decode, 0, sizeof(zor_huf_entry) * size);
  h->decode_len = size;
  size_t rank_idx[MAX_MAX_NUM_BITS + 2]; memset(rank_idx, 0, sizeof rank_idx);
  rank_idx[max_bits] = 0;                              /* :344-351 */
  for (unsigned b = max_bits; b >= 1; b--)
    rank_idx[b - 1] = rank_idx[b] + (size_t)bit_ranks[b] * ((size_t)1 << (max_bits - b));
  if (rank_idx[0] != size) return ZOR_REF_PANIC;       /* assert :353-358 */
  for (size_t s = 0; s < nbits; s++) {                 /* :360-374 */
    unsigned b = bits[s];
    if (b != 0) {
      size_t base = rank_idx[b], n = (size_t)1 << (max_bits - b);
      rank_idx[b] += n;
      if (base + n > size) return ZOR_REF_PANIC;
      for (size_t i = 0; i < n; i++) { h->decode[base + i].symbol = (uint8_t)s; h->decode[base + i].num_bits = (uint8_t)b; }
    }
  }
  return ZOR_OK;
}

static int huf_build_decoder(huf_table* h, const uint8_t* src, size_t len, uint32_t* bytes_used) {
  /* build_decoder :117-124 */
  h->decode_len = 0;
  int st = huf_read_weights(h, src, len, bytes_used);
  if (st) return st;
  return huf_build_table_from_weights(h);
}
int zor_huf_build_decoder(const uint8_t* src, size_t len, zor_huf_entry* out, int* max_bits, uint32_t* bytes_read) {
  huf_table* h = (huf_table*)malloc(sizeof *h); if (!h) return ZOR_NOMEM;
  huf_reset(h);
  int st = huf_build_decoder(h, src, len, bytes_read);
  if (!st) { memcpy(out, h->decode, sizeof(zor_huf_entry) * h->decode_len); *max_bits = h->max_num_bits; }
  free(h); return st;
}

/* ------------------------------------------------------------------------------------------
 * growable byte / sequence vectors
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t* p; size_t len, cap; } bytevec;
static int bv_reserve(bytevec* v, size_t extra) {
  if (v->len + extra <= v->cap) return 0;
  size_t nc = v->cap ? v->cap : 4096; while (nc < v->len + extra) nc *= 2;
  uint8_t* np = (uint8_t*)realloc(v->p, nc); if (!np) return -1;
  v->p = np; v->cap = nc; return 0;
}
static int bv_push(bytevec* v, const uint8_t* d, size_t n) {
  if (bv_reserve(v, n)) return -1; if (n) memcpy(v->p + v->len, d, n); v->len += n; return 0;
}

/* ------------------------------------------------------------------------------------------
 * DecodeBuffer — decoding/decode_buffer.rs (the ring buffer is replaced by a flat vector
 * with a drained-prefix index; `len` below is RingBuffer::len(), i.e. undrained bytes)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  bytevec buf; size_t head;          /* bytes [head, buf.len) are held */
  bytevec dict_content;              /* :11 */
  size_t window_size;                /* :13 */
  uint64_t total_output_counter;     /* :14 */
  xxh64_state hash;                  /* :16 */
} decode_buffer;
static size_t db_len(const decode_buffer* b) { return b->buf.len - b->head; } /* len :58 */
static void db_reset(decode_buffer* b, size_t window_size) { /* reset :46-56 */
  b->window_size = window_size; b->buf.len = 0; b->head = 0; b->dict_content.len = 0;
  b->total_output_counter = 0; xxh64_reset(&b->hash, 0);
}
static int db_push(decode_buffer* b, const uint8_t* d, size_t n) { /* push :74-77 */
  if (bv_push(&b->buf, d, n)) return ZOR_NOMEM; b->total_output_counter += n; return ZOR_OK;
}
static int db_extend_raw(decode_buffer* b, const uint8_t* d, size_t n) { /* extend_from_reader :66-72 (no counter) */
  return bv_push(&b->buf, d, n) ? ZOR_NOMEM : ZOR_OK;
}
static int db_extend_fill(decode_buffer* b, uint8_t byte, size_t n) { /* extend_and_fill :62-64 (no counter) */
  if (bv_reserve(&b->buf, n)) return ZOR_NOMEM; memset(b->buf.p + b->buf.len, byte, n); b->buf.len += n; return ZOR_OK;
}
static int db_repeat(decode_buffer* b, size_t offset, size_t match_length); /* fwd */
static int db_repeat_from_dict(decode_buffer* b, size_t offset, size_t match_length) { /* repeat_from_dict :144-179 */
  if (b->total_output_counter <= (uint64_t)b->window_size) {
    size_t bytes_from_dict = offset - db_len(b);
    if (bytes_from_dict > b->dict_content.len) return ZOR_EXE_DICT_TOO_SMALL; /* :152-157 */
    if (bytes_from_dict < match_length) {
      if (bv_push(&b->buf, b->dict_content.p + (b->dict_content.len - bytes_from_dict), bytes_from_dict)) return ZOR_NOMEM;
      b->total_output_counter += bytes_from_dict;      /* :163 */
      return db_repeat(b, db_len(b), match_length - bytes_from_dict); /* :164 */
    } else {
      size_t low = b->dict_content.len - bytes_from_dict;
      if (bv_push(&b->buf, b->dict_content.p + low, match_length)) return ZOR_NOMEM; /* :166-170, counter untouched */
    }
    return ZOR_OK;
  }
  return ZOR_EXE_OFFSET_TOO_BIG;                        /* :173-177 */
}
static int db_repeat(decode_buffer* b, size_t offset, size_t match_length) { /* repeat :79-111 */
  if (offset > db_len(b)) return db_repeat_from_dict(b, offset, match_length);
  if (bv_reserve(&b->buf, match_length)) return ZOR_NOMEM;
  size_t start = b->buf.len - offset;
  uint8_t* p = b->buf.p;
  if (offset >= match_length) memcpy(p + b->buf.len, p + start, match_length);
  else for (size_t i = 0; i < match_length; i++) p[b->buf.len + i] = p[start + i]; /* repeat_in_chunks :113-141 */
  b->buf.len += match_length;
  b->total_output_counter += match_length;              /* :108 */
  return ZOR_OK;
}
static size_t db_can_drain_to_window(const decode_buffer* b) { /* can_drain_to_window_size :182-188 (None → 0) */
  return db_len(b) > b->window_size ? db_len(b) - b->window_size : 0;
}
static size_t db_drain_to(decode_buffer* b, size_t amount, uint8_t* dst) { /* drain_to :256-314 */
  if (amount == 0) return 0;
  if (dst) memcpy(dst, b->buf.p + b->head, amount);
  xxh64_update(&b->hash, b->buf.p + b->head, amount);
  b->head += amount;
  if (b->head == b->buf.len) { b->head = 0; b->buf.len = 0; }
  else if (b->head > (1u << 22) && b->head > db_len(b)) { /* compact; invisible to semantics */
    size_t l = db_len(b); memmove(b->buf.p, b->buf.p + b->head, l); b->head = 0; b->buf.len = l;
  }
  return amount;
}

/* ------------------------------------------------------------------------------------------
 * scratch — decoding/scratch.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct { fse_table offsets, literal_lengths, match_lengths; int of_rle, ll_rle, ml_rle; /* -1 = None */ } fse_scratch; /* :99-106 */
typedef struct { zor_sequence* p; size_t len, cap; } seqvec;
typedef struct {
  huf_table huf;              /* :17 */
  fse_scratch fse;            /* :19 */
  decode_buffer buffer;       /* :21 */
  uint32_t offset_hist[3];    /* :22 */
  bytevec literals_buffer;    /* :24 */
  seqvec sequences;           /* :25 */
} decoder_scratch;

static void fse_scratch_init(fse_scratch* f) { /* FSEScratch::new :109-118 */
  fse_reset(&f->offsets); f->offsets.max_symbol = MAX_OFFSET_CODE;
  fse_reset(&f->literal_lengths); f->literal_lengths.max_symbol = MAX_LITERAL_LENGTH_CODE;
  fse_reset(&f->match_lengths); f->match_lengths.max_symbol = MAX_MATCH_LENGTH_CODE;
  f->of_rle = f->ll_rle = f->ml_rle = -1;
}
static void scratch_reset(decoder_scratch* s, size_t window_size) { /* reset :52-68 */
  s->offset_hist[0] = 1; s->offset_hist[1] = 4; s->offset_hist[2] = 8;
  s->literals_buffer.len = 0; s->sequences.len = 0;
  db_reset(&s->buffer, window_size);
  fse_scratch_init(&s->fse);
  huf_reset(&s->huf);
}

/* ------------------------------------------------------------------------------------------
 * Dictionary — decoding/dictionary.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct dict { uint32_t id; fse_scratch fse; huf_table huf; bytevec content; uint32_t offset_hist[3]; struct dict* next; } dict;

static int dict_decode(const uint8_t* raw, size_t len, dict** out) { /* decode_dict :45-126 */
  static const uint8_t DMAGIC[4] = {0x37, 0xA4, 0x30, 0xEC}; /* :39 */
  if (len < 8) return ZOR_DICT_DECODE;
  dict* d = (dict*)calloc(1, sizeof *d); if (!d) return ZOR_NOMEM;
  fse_scratch_init(&d->fse); huf_reset(&d->huf);
  d->offset_hist[0] = 2; d->offset_hist[1] = 4; d->offset_hist[2] = 8;
  int st = ZOR_DICT_DECODE;
  if (memcmp(raw, DMAGIC, 4)) goto fail;
  d->id = rd32(raw + 4);
  const uint8_t* t = raw + 8; size_t tl = len - 8;
  uint32_t huf_size; size_t n;
  if ((st = huf_build_decoder(&d->huf, t, tl, &huf_size))) { if (st != ZOR_REF_PANIC) st = ZOR_DICT_DECODE; goto fail; }
  st = ZOR_DICT_DECODE;
  if (tl < huf_size) goto fail; t += huf_size; tl -= huf_size;
  if (fse_build_decoder(&d->fse.offsets, t, tl, OF_MAX_LOG, &n)) goto fail;          /* :74-77 */
  if (tl < n) goto fail; t += n; tl -= n;
  if (fse_build_decoder(&d->fse.match_lengths, t, tl, ML_MAX_LOG, &n)) goto fail;    /* :84-87 */
  if (tl < n) goto fail; t += n; tl -= n;
  if (fse_build_decoder(&d->fse.literal_lengths, t, tl, LL_MAX_LOG, &n)) goto fail;  /* :94-97 */
  if (tl < n) goto fail; t += n; tl -= n;
  if (tl < 12) goto fail;
  d->offset_hist[0] = rd32(t); d->offset_hist[1] = rd32(t + 4); d->offset_hist[2] = rd32(t + 8);
  if (bv_push(&d->content, t + 12, tl - 12)) { st = ZOR_NOMEM; goto fail; }
  *out = d; return ZOR_OK;
fail:
  free(d->content.p); free(d); return st;
}

/* ------------------------------------------------------------------------------------------
 * literals section header — blocks/literals_section.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t regenerated_size; int has_compressed_size; uint32_t compressed_size; int num_streams; int ls_type; } literals_section;
static int literals_parse_header(literals_section* s, const uint8_t* raw, size_t len, unsigned* hdr) {
  /* parse_from_header :117-223, header_bytes_needed :66-114 */
  if (len == 0) return ZOR_LITERALS_HEADER;            /* GetBitsError from br.get_bits(2) :119 */
  s->ls_type = raw[0] & 3; unsigned sf = (raw[0] >> 2) & 3;
  s->has_compressed_size = 0; s->num_streams = 0;
  unsigned need;
  if (s->ls_type == 0 || s->ls_type == 1) need = (sf == 0 || sf == 2) ? 1 : (sf == 1 ? 2 : 3);
  else need = (sf <= 1) ? 3 : (sf == 2 ? 4 : 5);
  if (len < need) return ZOR_LITERALS_HEADER;          /* NotEnoughBytes :124-129 */
  if (s->ls_type == 0 || s->ls_type == 1) {
    if (sf == 0 || sf == 2) s->regenerated_size = raw[0] >> 3;
    else if (sf == 1) s->regenerated_size = (raw[0] >> 4) + ((uint32_t)raw[1] << 4);
    else s->regenerated_size = (raw[0] >> 4) + ((uint32_t)raw[1] << 4) + ((uint32_t)raw[2] << 12);
  } else {
    s->num_streams = sf == 0 ? 1 : 4; s->has_compressed_size = 1;
    if (sf <= 1) {
      s->regenerated_size = (raw[0] >> 4) + (((uint32_t)raw[1] & 0x3f) << 4);
      s->compressed_size = (raw[1] >> 6) + ((uint32_t)raw[2] << 2);
    } else if (sf == 2) {
      s->regenerated_size = (raw[0] >> 4) + ((uint32_t)raw[1] << 4) + (((uint32_t)raw[2] & 0x3) << 12);
      s->compressed_size = (raw[2] >> 2) + ((uint32_t)raw[3] << 6);
    } else {
      s->regenerated_size = (raw[0] >> 4) + ((uint32_t)raw[1] << 4) + (((uint32_t)raw[2] & 0x3F) << 12);
      s->compressed_size = (raw[2] >> 6) + ((uint32_t)raw[3] << 2) + ((uint32_t)raw[4] << 10);
    }
  }
  *hdr = need; return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * sequences section header — blocks/sequence_section.rs:108-167
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t num_sequences; int has_modes; uint8_t modes; } sequences_header;
static int sequences_parse_header(sequences_header* h, const uint8_t* src, size_t len, unsigned* hdr) {
  h->num_sequences = 0; h->has_modes = 0; h->modes = 0;
  if (len == 0) return ZOR_SEQUENCES_HEADER;
  uint8_t b0 = src[0];
  if (b0 == 0) { *hdr = 1; return ZOR_OK; }
  if (b0 < 128) {
    if (len < 2) return ZOR_SEQUENCES_HEADER;
    h->num_sequences = b0; h->has_modes = 1; h->modes = src[1]; *hdr = 2; return ZOR_OK;
  }
  if (b0 < 255) {
    if (len < 2) return ZOR_SEQUENCES_HEADER;
    h->num_sequences = (((uint32_t)b0 - 128) << 8) + src[1]; *hdr = 2;
    if (h->num_sequences != 0) {
      if (len < 3) return ZOR_SEQUENCES_HEADER;
      h->has_modes = 1; h->modes = src[2]; *hdr = 3;
    }
    return ZOR_OK;
  }
  if (len < 4) return ZOR_SEQUENCES_HEADER;
  h->num_sequences = (uint32_t)src[1] + ((uint32_t)src[2] << 8) + 0x7F00;
  h->has_modes = 1; h->modes = src[3]; *hdr = 4; return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * literals section decode — decoding/literals_section_decoder.rs
 * ---------------------------------------------------------------------------------------- */
static int skip_padding(revbits* br) { /* the padding loop, e.g. literals_section_decoder.rs:98-109 */
  int skipped = 0;
  for (;;) { uint64_t v = rb_get(br, 1); skipped++; if (v == 1 || skipped > 8) break; }
  return skipped > 8 ? -1 : 0;
}
static int huf_decode_stream(const huf_table* h, const uint8_t* s, size_t n, bytevec* target, int check_end) {
  /* one stream: literals_section_decoder.rs:94-122 (4-stream) / :128-147 (single) */
  revbits br; rb_init(&br, s, n);
  if (skip_padding(&br)) return ZOR_LIT_EXTRA_PADDING;
  uint64_t state = rb_get(&br, (unsigned)h->max_num_bits);      /* init_state huff0_decoder.rs:32-37 */
  int64_t lim = -(int64_t)h->max_num_bits;
  while (rb_remaining(&br) > lim) {
    if (bv_reserve(target, 1)) return ZOR_NOMEM;
    target->p[target->len++] = h->decode[state].symbol;          /* decode_symbol :25-27 */
    unsigned nb = h->decode[state].num_bits;                      /* next_state :41-53 */
    uint64_t nbits = rb_get(&br, nb);
    state = ((state << nb) & (uint64_t)(h->decode_len - 1)) | nbits;
  }
  if (check_end && rb_remaining(&br) != lim) return ZOR_LIT_BITSTREAM_MISMATCH; /* :116-121 */
  return ZOR_OK;
}
static int decompress_literals(const literals_section* sec, huf_table* huf, const uint8_t* source, bytevec* target, uint32_t* bytes_read_out) {
  /* decompress_literals :40-158 (source is already limited to compressed_size by the caller, block_decoder.rs:136) */
  size_t slen = sec->compressed_size;
  uint32_t bytes_read = 0;
  if (sec->ls_type == 2) {                               /* Compressed: table description :55-59 */
    int st = huf_build_decoder(huf, source, slen, &bytes_read);
    if (st) return st;
  } else if (huf->max_num_bits == 0) return ZOR_LIT_UNINIT_HUF; /* Treeless :60-63 */
  if (bytes_read > slen) return ZOR_REF_PANIC;
  source += bytes_read; slen -= bytes_read;
  if (sec->num_streams == 4) {
    if (slen < 6) return ZOR_LIT_MISSING_JUMP;           /* :72-74 */
    size_t j1 = source[0] + ((size_t)source[1] << 8);
    size_t j2 = j1 + source[2] + ((size_t)source[3] << 8);
    size_t j3 = j2 + source[4] + ((size_t)source[5] << 8);
    bytes_read += 6; source += 6; slen -= 6;
    if (slen < j3) return ZOR_LIT_MISSING_BYTES;         /* :81-86 */
    const uint8_t* st_[4] = {source, source + j1, source + j2, source + j3};
    size_t ln_[4] = {j1, j2 - j1, j3 - j2, slen - j3};
    for (int i = 0; i < 4; i++) { int st = huf_decode_stream(huf, st_[i], ln_[i], target, 1); if (st) return st; }
    bytes_read += (uint32_t)slen;                        /* :124 */
  } else {
    int st = huf_decode_stream(huf, source, slen, target, 0);
    if (st) return st;
    bytes_read += (uint32_t)slen;                        /* :148 */
  }
  if (target->len != sec->regenerated_size) return ZOR_LIT_COUNT_MISMATCH; /* :150-155 */
  *bytes_read_out = bytes_read; return ZOR_OK;
}
static int decode_literals(const literals_section* sec, huf_table* huf, const uint8_t* source, size_t slen, bytevec* target, uint32_t* used) {
  /* decode_literals :12-34 */
  (void)slen;
  if (sec->ls_type == 0) { if (bv_push(target, source, sec->regenerated_size)) return ZOR_NOMEM; *used = sec->regenerated_size; return ZOR_OK; }
  if (sec->ls_type == 1) {
    if (bv_reserve(target, sec->regenerated_size)) return ZOR_NOMEM;
    memset(target->p + target->len, source[0], sec->regenerated_size); target->len += sec->regenerated_size;
    *used = 1; return ZOR_OK;
  }
  return decompress_literals(sec, huf, source, target, used);
}

/* ------------------------------------------------------------------------------------------
 * sequence section decode — decoding/sequence_section_decoder.rs
 * ---------------------------------------------------------------------------------------- */
static const int32_t LL_DEFAULT[36] = { /* :420-423 */
  4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int32_t ML_DEFAULT[53] = { /* :431-434 */
  1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int32_t OF_DEFAULT[29] = { /* :442-444 */
  1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

static int lookup_ll_code(uint8_t code, uint32_t* value, uint8_t* nbits) { /* :227-252 */
  static const uint32_t base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                    48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
  static const uint8_t bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  if (code > 35) return -1; *value = base[code]; *nbits = bits[code]; return 0;
}
static int lookup_ml_code(uint8_t code, uint32_t* value, uint8_t* nbits) { /* :258-284 */
  static const uint32_t base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                                    35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
  static const uint8_t bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                   1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  if (code > 52) return -1; *value = base[code]; *nbits = bits[code]; return 0;
}

static int update_one_table(int mode, fse_table* tab, int* rle, const uint8_t* src, size_t len, int max_log,
                            int max_code, const int32_t* dflt, size_t ndflt, int dflt_log, size_t* used) {
  /* one arm of maybe_update_fse_tables :303-408 */
  *used = 0;
  switch (mode) {
    case 2: { int st = fse_build_decoder(tab, src, len, max_log, used); if (st) return st; *rle = -1; return ZOR_OK; }
    case 1:
      if (len == 0) return ZOR_SEQ_RLE_BYTE;
      *used = 1;
      if (src[0] > max_code) return ZOR_SEQ_RLE_BYTE;   /* reported as MissingByteForRleMlTable :320-322 */
      *rle = src[0]; return ZOR_OK;
    case 0: { int st = fse_build_from_probabilities(tab, dflt_log, dflt, ndflt); if (st) return st; *rle = -1; return ZOR_OK; }
    default: return ZOR_OK;                             /* Repeat: nothing */
  }
}
static int maybe_update_fse_tables(const sequences_header* sec, const uint8_t* src, size_t len, fse_scratch* sc, size_t* bytes_read) {
  /* :294-410 — order LL, OF, ML */
  if (!sec->has_modes) return ZOR_SEQ_MISSING_MODE;
  size_t rd = 0, used; int st;
  st = update_one_table(sec->modes >> 6, &sc->literal_lengths, &sc->ll_rle, src, len, LL_MAX_LOG, MAX_LITERAL_LENGTH_CODE, LL_DEFAULT, 36, 6, &used);
  if (st) return st; rd += used;
  st = update_one_table((sec->modes >> 4) & 3, &sc->offsets, &sc->of_rle, src + rd, len - rd, OF_MAX_LOG, MAX_OFFSET_CODE, OF_DEFAULT, 29, 5, &used);
  if (st) return st; rd += used;
  st = update_one_table((sec->modes >> 2) & 3, &sc->match_lengths, &sc->ml_rle, src + rd, len - rd, ML_MAX_LOG, MAX_MATCH_LENGTH_CODE, ML_DEFAULT, 53, 6, &used);
  if (st) return st; rd += used;
  *bytes_read = rd; return ZOR_OK;
}

static int seq_push(seqvec* v, zor_sequence s) {
  if (v->len == v->cap) { size_t nc = v->cap ? v->cap * 2 : 1024; zor_sequence* np = (zor_sequence*)realloc(v->p, nc * sizeof *np); if (!np) return -1; v->p = np; v->cap = nc; }
  v->p[v->len++] = s; return 0;
}

static int decode_sequences(const sequences_header* sec, const uint8_t* source, size_t slen, fse_scratch* sc, seqvec* target) {
  /* decode_sequences :14-47 + decode_sequences_with_rle :49-152 / _without_rle :154-221 (same loop) */
  size_t bytes_read; int st = maybe_update_fse_tables(sec, source, slen, sc, &bytes_read);
  if (st) return st;
  revbits br; rb_init(&br, source + bytes_read, slen - bytes_read);
  if (skip_padding(&br)) return ZOR_SEQ_EXTRA_PADDING;   /* :29-40 */
  fse_decoder ll, ml, of;
  fsed_new(&ll, &sc->literal_lengths); fsed_new(&ml, &sc->match_lengths); fsed_new(&of, &sc->offsets);
  if (sc->ll_rle < 0 && (st = fsed_init_state(&ll, &br))) return st;  /* order LL, OF, ML :59-67 / :164-166 */
  if (sc->of_rle < 0 && (st = fsed_init_state(&of, &br))) return st;
  if (sc->ml_rle < 0 && (st = fsed_init_state(&ml, &br))) return st;
  target->len = 0;
  for (uint32_t i = 0; i < sec->num_sequences; i++) {
    uint8_t ll_code = sc->ll_rle >= 0 ? (uint8_t)sc->ll_rle : ll.state.symbol;
    uint8_t ml_code = sc->ml_rle >= 0 ? (uint8_t)sc->ml_rle : ml.state.symbol;
    uint8_t of_code = sc->of_rle >= 0 ? (uint8_t)sc->of_rle : of.state.symbol;
    uint32_t ll_value, ml_value; uint8_t ll_bits, ml_bits;
    if (lookup_ll_code(ll_code, &ll_value, &ll_bits)) return ZOR_REF_PANIC;   /* unreachable! :250 */
    if (lookup_ml_code(ml_code, &ml_value, &ml_bits)) return ZOR_REF_PANIC;   /* unreachable! :282 */
    if (of_code > MAX_OFFSET_CODE) return ZOR_SEQ_UNSUPPORTED_OFFSET;          /* :103-107 */
    /* get_bits_triple(of_code, ml_bits, ll_bits) bit_reader_reverse.rs:151-162 == three get_bits */
    uint64_t obits = rb_get(&br, of_code), ml_add = rb_get(&br, ml_bits), ll_add = rb_get(&br, ll_bits);
    uint32_t offset = (uint32_t)obits + (1u << of_code);
    zor_sequence s = {ll_value + (uint32_t)ll_add, ml_value + (uint32_t)ml_add, offset, 0};
    if (seq_push(target, s)) return ZOR_NOMEM;
    if (target->len < sec->num_sequences) {              /* update order LL, ML, OF :127-137 / :204-206 */
      if (sc->ll_rle < 0 && (st = fsed_update_state(&ll, &br))) return st;
      if (sc->ml_rle < 0 && (st = fsed_update_state(&ml, &br))) return st;
      if (sc->of_rle < 0 && (st = fsed_update_state(&of, &br))) return st;
    }
    if (rb_remaining(&br) < 0) return ZOR_SEQ_NOT_ENOUGH_BYTES; /* :140-142 */
  }
  if (rb_remaining(&br) > 0) return ZOR_SEQ_EXTRA_BITS;  /* :145-151 */
  return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * sequence execution — decoding/sequence_execution.rs
 * ---------------------------------------------------------------------------------------- */
uint32_t zor_do_offset_history(uint32_t offset_value, uint32_t lit_len, uint32_t h[3]) { /* do_offset_history :59-118 */
  uint32_t actual;
  if (lit_len > 0) {
    if (offset_value >= 1 && offset_value <= 3) actual = h[offset_value - 1]; else actual = offset_value - 3;
  } else {
    if (offset_value == 1 || offset_value == 2) actual = h[offset_value];
    else if (offset_value == 3) actual = h[0] ? h[0] - 1 : 0;   /* saturating_sub :74 */
    else actual = offset_value - 3;
  }
  if (lit_len > 0) {
    if (offset_value == 1) { }
    else if (offset_value == 2) { h[1] = h[0]; h[0] = actual; }
    else { h[2] = h[1]; h[1] = h[0]; h[0] = actual; }
  } else {
    if (offset_value == 1) { h[1] = h[0]; h[0] = actual; }
    else { h[2] = h[1]; h[1] = h[0]; h[0] = actual; }
  }
  return actual;
}
static int execute_sequences(decoder_scratch* s) { /* execute_sequences :5-54 */
  size_t lit_counter = 0;
  for (size_t i = 0; i < s->sequences.len; i++) {
    zor_sequence* q = &s->sequences.p[i];
    if (q->ll > 0) {
      size_t high = lit_counter + q->ll;
      if (high > s->literals_buffer.len) return ZOR_EXE_NOT_ENOUGH_LITERALS; /* :14-19 */
      int st = db_push(&s->buffer, s->literals_buffer.p + lit_counter, q->ll); if (st) return st;
      lit_counter += q->ll;
    }
    uint32_t actual = zor_do_offset_history(q->of, q->ll, s->offset_hist);
    q->actual_of = actual;
    if (actual == 0) return ZOR_EXE_ZERO_OFFSET;         /* :28-30 */
    if (q->ml > 0) { int st = db_repeat(&s->buffer, actual, q->ml); if (st) return st; }
  }
  if (lit_counter < s->literals_buffer.len) {            /* :40-44 */
    int st = db_push(&s->buffer, s->literals_buffer.p + lit_counter, s->literals_buffer.len - lit_counter); if (st) return st;
  }
  return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * frame header — decoding/frame.rs
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t descriptor; uint8_t window_descriptor; int has_dict_id; uint32_t dict_id; uint64_t fcs; } frame_header;
static int fh_single_segment(const frame_header* h) { return (h->descriptor >> 5) & 1; }   /* :189-191 */
static int fh_checksum_flag(const frame_header* h) { return (h->descriptor >> 2) & 1; }    /* :194-196 */

struct zor_decoder {
  int has_state;                    /* state: Option<FrameDecoderState> frame_decoder.rs:81 */
  frame_header fh;
  decoder_scratch scratch;
  int frame_finished; uint64_t block_counter, bytes_read_counter;
  int has_checksum; uint32_t check_sum;
  uint32_t using_dict;
  dict* dicts;
  uint64_t max_window_size;
  int last_block_type;
};

static int read_frame_header(const uint8_t* src, size_t len, frame_header* h, size_t* consumed, uint32_t* skip_magic, uint32_t* skip_len) {
  /* read_frame_header frame.rs:6-85 */
  size_t p = 0;
  if (len < 4) return ZOR_HEADER_READ;
  uint32_t magic = rd32(src); p = 4;
  if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {   /* :15-23 */
    if (len < 8) return ZOR_HEADER_READ;
    if (skip_magic) *skip_magic = magic; if (skip_len) *skip_len = rd32(src + 4);
    *consumed = 8; return ZOR_SKIP_FRAME;
  }
  if (magic != MAGIC_NUM) return ZOR_BAD_MAGIC;
  if (len < p + 1) return ZOR_HEADER_READ;
  memset(h, 0, sizeof *h);
  h->descriptor = src[p++];
  if (!fh_single_segment(h)) { if (len < p + 1) return ZOR_HEADER_READ; h->window_descriptor = src[p++]; }
  static const unsigned didlen[4] = {0, 1, 2, 4};       /* dictionary_id_bytes :231-239 */
  unsigned dl = didlen[h->descriptor & 3];
  if (dl) {
    if (len < p + dl) return ZOR_HEADER_READ;
    uint32_t id = 0; for (unsigned i = 0; i < dl; i++) id += (uint32_t)src[p + i] << (8 * i);
    p += dl;
    if (id != 0) { h->has_dict_id = 1; h->dict_id = id; }  /* :60-62 */
  }
  unsigned fl;                                           /* frame_content_size_bytes :212-226 */
  switch (h->descriptor >> 6) { case 0: fl = fh_single_segment(h) ? 1 : 0; break; case 1: fl = 2; break; case 2: fl = 4; break; default: fl = 8; }
  if (fl) {
    if (len < p + fl) return ZOR_HEADER_READ;
    uint64_t fcs = 0; for (unsigned i = 0; i < fl; i++) fcs += (uint64_t)src[p + i] << (8 * i);
    if (fl == 2) fcs += 256;                             /* :78-80 */
    h->fcs = fcs; p += fl;
  }
  *consumed = p; return ZOR_OK;
}
static int fh_window_size(const frame_header* h, uint64_t* out) { /* window_size frame.rs:116-139 */
  if (fh_single_segment(h)) { *out = h->fcs; return ZOR_OK; }
  unsigned exp = h->window_descriptor >> 3, mant = h->window_descriptor & 7;
  uint64_t base = 1ull << (10 + exp), w = base + (base / 8) * mant;
  if (w < MIN_WINDOW_SIZE) return ZOR_WINDOW_TOO_SMALL;
  if (w >= MAX_WINDOW_SIZE) return ZOR_WINDOW_TOO_BIG_SPEC;
  *out = w; return ZOR_OK;
}

/* ------------------------------------------------------------------------------------------
 * FrameDecoder — decoding/frame_decoder.rs
 * ---------------------------------------------------------------------------------------- */
zor_decoder* zor_new(void) { /* new :158-164 */
  zor_decoder* d = (zor_decoder*)calloc(1, sizeof *d); if (!d) return NULL;
  d->max_window_size = DEFAULT_MAX_WINDOW_SIZE;
  scratch_reset(&d->scratch, 0);
  return d;
}
void zor_free(zor_decoder* d) {
  if (!d) return;
  free(d->scratch.buffer.buf.p); free(d->scratch.buffer.dict_content.p);
  free(d->scratch.literals_buffer.p); free(d->scratch.sequences.p);
  for (dict* x = d->dicts; x;) { dict* n = x->next; free(x->content.p); free(x); x = n; }
  free(d);
}
void zor_set_max_window_size(zor_decoder* d, uint64_t m) { d->max_window_size = m < MAX_WINDOW_SIZE ? m : MAX_WINDOW_SIZE; } /* :175-177 */

static dict* find_dict(zor_decoder* d, uint32_t id) { for (dict* x = d->dicts; x; x = x->next) if (x->id == id) return x; return NULL; }
int zor_add_dict(zor_decoder* d, const uint8_t* raw, size_t len, uint32_t* id_out) { /* add_dict :224-227 */
  dict* x; int st = dict_decode(raw, len, &x); if (st) return st;
  dict** pp = &d->dicts;                                 /* BTreeMap::insert replaces an existing id */
  while (*pp) { if ((*pp)->id == x->id) { dict* old = *pp; *pp = old->next; free(old->content.p); free(old); break; } pp = &(*pp)->next; }
  x->next = d->dicts; d->dicts = x;
  if (id_out) *id_out = x->id; return ZOR_OK;
}
static void init_from_dict(decoder_scratch* s, const dict* x) { /* scratch.rs:70-78 */
  s->fse = x->fse; s->huf = x->huf;
  memcpy(s->offset_hist, x->offset_hist, sizeof s->offset_hist);
  s->buffer.dict_content.len = 0; bv_push(&s->buffer.dict_content, x->content.p, x->content.len);
}
int zor_force_dict(zor_decoder* d, uint32_t id) { /* force_dict :229-243 */
  if (!d->has_state) return ZOR_NOT_INITIALIZED;
  dict* x = find_dict(d, id); if (!x) return ZOR_DICT_NOT_PROVIDED;
  init_from_dict(&d->scratch, x); d->using_dict = id; return ZOR_OK;
}

int zor_init(zor_decoder* d, const uint8_t* src, size_t len, size_t* consumed, uint32_t* skip_magic, uint32_t* skip_len) {
  /* init/reset :190-221; FrameDecoderState::new/reset :103-134 */
  frame_header fh; size_t hs = 0;
  *consumed = 0;
  int st = read_frame_header(src, len, &fh, &hs, skip_magic, skip_len);
  if (st) { if (st == ZOR_SKIP_FRAME) *consumed = hs; return st; }
  uint64_t w; if ((st = fh_window_size(&fh, &w))) return st;
  if (w > d->max_window_size) return ZOR_WINDOW_SIZE_TOO_BIG;   /* check_window_size :137-145 */
  d->has_state = 1; d->fh = fh; d->frame_finished = 0; d->block_counter = 0;
  scratch_reset(&d->scratch, (size_t)w);
  d->bytes_read_counter = hs; d->has_checksum = 0; d->check_sum = 0; d->using_dict = 0;
  *consumed = hs;
  if (fh.has_dict_id) {                                           /* :212-219 */
    dict* x = find_dict(d, fh.dict_id); if (!x) return ZOR_DICT_NOT_PROVIDED;
    init_from_dict(&d->scratch, x); d->using_dict = fh.dict_id;
  }
  return ZOR_OK;
}

/* decompress_block — decoding/block_decoder.rs:97-197 */
static int decompress_block(decoder_scratch* ws, const uint8_t* raw, size_t content_size) {
  literals_section sec; unsigned lh;
  int st = literals_parse_header(&sec, raw, content_size, &lh); if (st) return st; /* :110-111 */
  const uint8_t* p = raw + lh; size_t rem = content_size - lh;
  size_t upper = sec.has_compressed_size ? sec.compressed_size : (sec.ls_type == 1 ? 1 : sec.regenerated_size); /* :120-127 */
  if (rem < upper) return ZOR_MALFORMED_SECTION_HEADER;  /* :129-134 */
  ws->literals_buffer.len = 0;                           /* :139 */
  uint32_t used = 0;
  if ((st = decode_literals(&sec, &ws->huf, p, upper, &ws->literals_buffer, &used))) return st; /* :140-145 */
  if (sec.regenerated_size != ws->literals_buffer.len) return ZOR_REF_PANIC; /* assert :146-151 */
  if (used != upper) return ZOR_REF_PANIC;               /* assert :152 */
  p += upper; rem -= upper;
  sequences_header sh; unsigned shl;
  if ((st = sequences_parse_header(&sh, p, rem, &shl))) return st; /* :157-158 */
  p += shl; rem -= shl;
  if (sh.num_sequences != 0) {
    if ((st = decode_sequences(&sh, p, rem, &ws->fse, &ws->sequences))) return st; /* :175-180 */
    return execute_sequences(ws);                        /* :182 */
  }
  if (rem != 0) return ZOR_SEQ_EXTRA_BITS;               /* :184-190 */
  ws->sequences.len = 0;
  return db_push(&ws->buffer, ws->literals_buffer.p, ws->literals_buffer.len); /* :192-193 */
}

int zor_decode_blocks(zor_decoder* d, const uint8_t* src, size_t len, size_t* consumed, int strat, size_t n, int* frame_finished) {
  /* decode_blocks :309-377 */
  *consumed = 0;
  if (!d->has_state) return ZOR_NOT_INITIALIZED;
  size_t p = 0;
  size_t buffer_size_before = db_len(&d->scratch.buffer);
  uint64_t block_counter_before = d->block_counter;
  for (;;) {
    /* read_block_header block_decoder.rs:201-247 */
    if (len - p < 3) { *consumed = len; return ZOR_FAILED_READ_BLOCK_HEADER; }
    const uint8_t* hb = src + p;
    int last = hb[0] & 1; unsigned btype = (hb[0] >> 1) & 3;
    if (btype == 3) { *consumed = p + 3; return ZOR_RESERVED_BLOCK; }           /* :213-216 */
    uint32_t bsize = (uint32_t)(hb[0] >> 3) | ((uint32_t)hb[1] << 5) | ((uint32_t)hb[2] << 13); /* :279-283 */
    if (bsize > MAX_BLOCK_SIZE) { *consumed = p + 3; return ZOR_BLOCK_SIZE_TOO_LARGE; } /* :270-277 */
    p += 3; d->bytes_read_counter += 3;                   /* frame_decoder.rs:328 */
    uint32_t content_size = btype == 1 ? 1 : bsize;       /* :229-234 */
    if (len - p < content_size) { *consumed = len; return ZOR_FAILED_READ_BLOCK_BODY; }
    int st;
    d->last_block_type = (int)btype;
    if (btype == 1) st = db_extend_fill(&d->scratch.buffer, src[p], bsize);            /* block_decoder.rs:55-70 */
    else if (btype == 0) st = db_extend_raw(&d->scratch.buffer, src + p, bsize);       /* :71-82 */
    else st = decompress_block(&d->scratch, src + p, content_size);                    /* :88-93 */
    if (btype != 2) { d->scratch.sequences.len = 0; }
    if (st) { *consumed = p + content_size; return st; }
    p += content_size; d->bytes_read_counter += content_size;                          /* :341 */
    d->block_counter++;
    if (last) {                                           /* :347-359 */
      d->frame_finished = 1;
      if (fh_checksum_flag(&d->fh)) {
        if (len - p < 4) { *consumed = len; return ZOR_FAILED_READ_CHECKSUM; }
        d->check_sum = rd32(src + p); d->has_checksum = 1; p += 4; d->bytes_read_counter += 4;
      }
      break;
    }
    if (strat == ZOR_STRAT_UPTO_BLOCKS) { if (d->block_counter - block_counter_before >= n) break; }
    else if (strat == ZOR_STRAT_UPTO_BYTES) { if (db_len(&d->scratch.buffer) - buffer_size_before >= n) break; }
  }
  *consumed = p; if (frame_finished) *frame_finished = d->frame_finished;
  return ZOR_OK;
}

int zor_is_finished(const zor_decoder* d) { /* :284-294 */
  if (!d->has_state) return 1;
  if (fh_checksum_flag(&d->fh)) return d->frame_finished && d->has_checksum;
  return d->frame_finished;
}
size_t zor_can_collect(const zor_decoder* d) { /* :410-424 */
  if (!d->has_state) return 0;
  return zor_is_finished(d) ? db_len(&d->scratch.buffer) : db_can_drain_to_window(&d->scratch.buffer);
}
size_t zor_collect(zor_decoder* d, uint8_t* dst, size_t cap) { /* collect :381-389 */
  if (!d->has_state) return 0;
  size_t n = zor_can_collect(d); if (n > cap) n = cap;
  return db_drain_to(&d->scratch.buffer, n, dst);
}
size_t zor_read(zor_decoder* d, uint8_t* dst, size_t cap) { /* impl Read :615-627; decode_buffer.rs:19-32, 241-254 */
  if (!d->has_state) return 0;
  size_t n = d->frame_finished ? db_len(&d->scratch.buffer) : db_can_drain_to_window(&d->scratch.buffer);
  if (n > cap) n = cap;
  return db_drain_to(&d->scratch.buffer, n, dst);
}
size_t zor_held(const zor_decoder* d, uint8_t* dst, size_t cap) { /* (test accessor, no counterpart: what the decode buffer holds, undrained) */
  if (!d->has_state) return 0;
  size_t n = db_len(&d->scratch.buffer); if (n > cap) n = cap;
  if (dst && n) memcpy(dst, d->scratch.buffer.buf.p + d->scratch.buffer.head, n);
  return n;
}
int zor_decode_from_to(zor_decoder* d, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* read_out, size_t* written_out) {
  /* decode_from_to :439-529: whole blocks only, a header whose body is not all there is "never read"; the checksum may come alone */
  *read_out = 0; *written_out = 0;
  const uint64_t at_start = d->has_state ? d->bytes_read_counter : 0;            /* :445-448 */
  if (!zor_is_finished(d) || !d->has_state) {                                     /* :450 */
    size_t p = 0;
    if (!d->has_state) {                                                          /* :453-455 */
      size_t c = 0; uint32_t sm = 0, sl = 0;
      int st = zor_init(d, src, len, &c, &sm, &sl);
      if (st) return st;
      p = c;
    }
    if (fh_checksum_flag(&d->fh) && d->frame_finished && !d->has_checksum) {      /* :465-477 */
      if (len - p >= 4) { d->check_sum = rd32(src + p); d->has_checksum = 1; d->bytes_read_counter += 4; }
      *read_out = 4;                                                              /* Ok((4, 0)) whether or not the bytes were there */
      return ZOR_OK;
    }
    for (;;) {
      if (len - p < 3) break;                                                     /* :481-483 */
      const uint8_t* hb = src + p;
      int last = hb[0] & 1; unsigned btype = (hb[0] >> 1) & 3;
      if (btype == 3) return ZOR_RESERVED_BLOCK;                                  /* :484-486 (read_block_header) */
      uint32_t bsize = (uint32_t)(hb[0] >> 3) | ((uint32_t)hb[1] << 5) | ((uint32_t)hb[2] << 13);
      if (bsize > MAX_BLOCK_SIZE) return ZOR_BLOCK_SIZE_TOO_LARGE;
      uint32_t content_size = btype == 1 ? 1 : bsize;
      if (len - p - 3 < content_size) break;                                      /* :490-492 */
      p += 3; d->bytes_read_counter += 3;                                         /* :493 */
      int st;
      d->last_block_type = (int)btype;
      if (btype == 1) st = db_extend_fill(&d->scratch.buffer, src[p], bsize);
      else if (btype == 0) st = db_extend_raw(&d->scratch.buffer, src + p, bsize);
      else st = decompress_block(&d->scratch, src + p, content_size);
      if (btype != 2) { d->scratch.sequences.len = 0; }
      if (st) return st;                                                          /* :495-501 */
      p += content_size; d->bytes_read_counter += content_size; d->block_counter++;   /* :502-503 */
      if (last) {                                                                 /* :505-517 */
        d->frame_finished = 1;
        if (fh_checksum_flag(&d->fh) && len - p >= 4) { d->check_sum = rd32(src + p); d->has_checksum = 1; p += 4; d->bytes_read_counter += 4; }
        break;
      }
    }
  }
  *written_out = zor_read(d, dst, cap);                                           /* :522 */
  *read_out = (size_t)(d->bytes_read_counter - at_start);                         /* :523-528 */
  return ZOR_OK;
}
int zor_decode_all(zor_decoder* d, const uint8_t* in, size_t inlen, uint8_t* out, size_t outcap, size_t* written) {
  /* decode_all :541-577 */
  size_t total = 0, p = 0;
  *written = 0;
  while (p < inlen) {
    size_t c; uint32_t sm, sl;
    int st = zor_init(d, in + p, inlen - p, &c, &sm, &sl);
    if (st == ZOR_SKIP_FRAME) {
      p += c;
      if ((size_t)sl > inlen - p) return ZOR_FAILED_SKIP_FRAME;  /* :550-556 */
      p += sl; continue;
    }
    if (st) return st;
    p += c;
    for (;;) {
      int fin;
      st = zor_decode_blocks(d, in + p, inlen - p, &c, ZOR_STRAT_UPTO_BYTES, 1024 * 1024, &fin);
      p += c;
      if (st) return st;
      size_t w = zor_read(d, out + total, outcap - total);
      total += w;
      if (zor_can_collect(d) != 0) return ZOR_TARGET_TOO_SMALL;  /* :567-569 */
      if (zor_is_finished(d)) break;
    }
  }
  *written = total; return ZOR_OK;
}

uint64_t zor_blocks_decoded(const zor_decoder* d) { return d->has_state ? d->block_counter : 0; }
uint64_t zor_bytes_read_from_source(const zor_decoder* d) { return d->has_state ? d->bytes_read_counter : 0; }
uint64_t zor_content_size(const zor_decoder* d) { return d->has_state ? d->fh.fcs : 0; }
uint64_t zor_window_size(const zor_decoder* d) { return d->has_state ? d->scratch.buffer.window_size : 0; }
int zor_checksum_from_data(const zor_decoder* d, uint32_t* out) { if (!d->has_state || !d->has_checksum) return 0; *out = d->check_sum; return 1; }
uint32_t zor_calculated_checksum(const zor_decoder* d) { return (uint32_t)xxh64_digest(&d->scratch.buffer.hash); }
uint32_t zor_dict_id(const zor_decoder* d) { return d->has_state && d->fh.has_dict_id ? d->fh.dict_id : 0; }

int zor_last_block_type(const zor_decoder* d) { return d->last_block_type; }
const uint8_t* zor_last_literals(const zor_decoder* d, size_t* len) { *len = d->scratch.literals_buffer.len; return d->scratch.literals_buffer.p; }
const zor_sequence* zor_last_sequences(const zor_decoder* d, size_t* n) { *n = d->scratch.sequences.len; return d->scratch.sequences.p; }
void zor_offset_hist(const zor_decoder* d, uint32_t out[3]) { memcpy(out, d->scratch.offset_hist, 12); }
size_t zor_fse_table(const zor_decoder* d, int which, const zor_fse_entry** entries, int* acc_log, int* rle) {
  const fse_table* t = which == 0 ? &d->scratch.fse.literal_lengths : which == 1 ? &d->scratch.fse.offsets : &d->scratch.fse.match_lengths;
  *rle = which == 0 ? d->scratch.fse.ll_rle : which == 1 ? d->scratch.fse.of_rle : d->scratch.fse.ml_rle;
  *entries = t->decode; *acc_log = t->accuracy_log; return t->decode_len;
}
size_t zor_huf_table(const zor_decoder* d, const zor_huf_entry** entries, int* max_bits) {
  *entries = d->scratch.huf.decode; *max_bits = d->scratch.huf.max_num_bits; return d->scratch.huf.decode_len;
}
