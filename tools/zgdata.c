/* zgdata.c — synthetic workload generators, integer-only and portable.
 * text_like: Zipf-distributed words over a splitmix64 vocabulary (the enwik-like stand-in, ratio ~3.2 at zstd -3);
 * iso_like: low-compressibility bytes (ratio ~1.18), Huffman-literal dominated.
 * Bench/test infrastructure: built by __graft_entry__.build() into tools/libzgdata.so. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t sm64(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void zgdata_iso_like(uint8_t* out, size_t n, uint64_t seed) {
  for (size_t i = 0; i < n; i++) {
    uint64_t r = sm64(seed, i);
    unsigned a = r & 255, b = (r >> 8) & 255, c = (r >> 16) & 255, sel = (r >> 24) & 7, u = (r >> 32) & 255;
    out[i] = sel == 0 ? (uint8_t)u : (uint8_t)((a * b * c) >> 16);
  }
}

static const unsigned LETTER_W[26] = {82, 15, 28, 43, 127, 22, 20, 61, 70, 2, 8, 40, 24, 67, 75, 19, 1, 60, 63, 91, 28, 10, 24, 2, 20, 1};

int zgdata_text_like(uint8_t* out, size_t n, uint64_t seed, uint32_t V) {
  unsigned cum[26], acc = 0;
  for (int i = 0; i < 26; i++) { acc += LETTER_W[i]; cum[i] = acc; }  /* 1003 */
  uint8_t* words = (uint8_t*)malloc((size_t)V * 12);
  uint8_t* wlen = (uint8_t*)malloc(V);
  uint64_t* zc = (uint64_t*)malloc((size_t)V * 8);
  if (!words || !wlen || !zc) { free(words); free(wlen); free(zc); return -1; }
  uint64_t vs = seed ^ 0x5EEDull, ztot = 0;
  for (uint32_t k = 0; k < V; k++) {
    uint64_t d0 = sm64(vs, 13ull * k);
    unsigned len = 2 + (unsigned)(d0 % 10);
    wlen[k] = (uint8_t)len;
    for (unsigned j = 0; j < len; j++) {
      unsigned v = (unsigned)(sm64(vs, 13ull * k + 1 + j) % 1003), c = 0;
      while (cum[c] <= v) c++;
      words[(size_t)k * 12 + j] = (uint8_t)('a' + c);
    }
    ztot += (uint64_t)(4294967296.0 / (double)(k + 1));
    zc[k] = ztot;
  }
  /* exact integer weights floor(2^32/(k+1)) */
  ztot = 0;
  for (uint32_t k = 0; k < V; k++) { ztot += 4294967296ull / (k + 1); zc[k] = ztot; }
  size_t pos = 0;
  for (uint64_t chunk = 0; pos < n; chunk++) {
    uint64_t cs = seed + chunk * 0x1000003ull;
    for (uint64_t t = 0; t < 200000 && pos < n; t++) {
      uint64_t rw = sm64(cs, 2 * t) % ztot, rs = sm64(cs, 2 * t + 1) % 100;
      uint32_t lo = 0, hi = V - 1;  /* smallest index whose cumulative weight exceeds rw */
      while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (zc[mid] > rw) hi = mid; else lo = mid + 1; }
      const uint8_t* w = words + (size_t)lo * 12;
      for (unsigned j = 0; j < wlen[lo] && pos < n; j++) out[pos++] = w[j];
      const char* sep = rs < 86 ? " " : rs < 92 ? ", " : rs < 97 ? ". " : "\n";
      for (const char* s = sep; *s && pos < n; s++) out[pos++] = (uint8_t)*s;
    }
  }
  free(words); free(wlen); free(zc);
  return 0;
}
