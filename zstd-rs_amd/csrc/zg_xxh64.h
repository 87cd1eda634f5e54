// zg_xxh64.h — XXH64 (seed 0 in practice), the content checksum of the zstd frame format. The reference feeds the
// bytes it drains to twox_hash::XxHash64 (ruzstd/src/decoding/decode_buffer.rs:16,42,223-227,290,301) and reports
// the low 32 bits (frame_decoder.rs:263-270). Host-side streaming implementation of the public algorithm.
#pragma once
#include <stdint.h>
#include <string.h>
namespace zg {
class Xxh64 {
 public:
  Xxh64() { reset(0); }
  void reset(uint64_t seed) {
    seed_ = seed; v_[0] = seed + P1 + P2; v_[1] = seed + P2; v_[2] = seed; v_[3] = seed - P1; n_ = 0; total_ = 0;
  }
  void update(const uint8_t* p, size_t len) {
    total_ += len;
    if (n_ + len < 32) { memcpy(mem_ + n_, p, len); n_ += (unsigned)len; return; }
    const uint8_t* end = p + len;
    if (n_) {
      size_t fill = 32 - n_;
      memcpy(mem_ + n_, p, fill);
      for (int i = 0; i < 4; i++) v_[i] = round(v_[i], rd(mem_ + 8 * i));
      p += fill; n_ = 0;
    }
    {  // the four lanes in locals: the stripe loop keeps them in registers (as members they are stored and reloaded every stripe,
       // a byte pointer may alias them) — the streaming decoder's hasher thread runs at this loop's speed
      uint64_t a = v_[0], b = v_[1], c = v_[2], d = v_[3];
      while (p + 32 <= end) { a = round(a, rd(p)); b = round(b, rd(p + 8)); c = round(c, rd(p + 16)); d = round(d, rd(p + 24)); p += 32; }
      v_[0] = a; v_[1] = b; v_[2] = c; v_[3] = d;
    }
    if (p < end) { memcpy(mem_, p, (size_t)(end - p)); n_ = (unsigned)(end - p); }
  }
  uint64_t digest() const {
    uint64_t h;
    if (total_ >= 32) {
      h = rotl(v_[0], 1) + rotl(v_[1], 7) + rotl(v_[2], 12) + rotl(v_[3], 18);
      for (int i = 0; i < 4; i++) { h ^= round(0, v_[i]); h = h * P1 + P4; }
    } else h = seed_ + P5;
    h += total_;
    const uint8_t* p = mem_; const uint8_t* end = p + n_;
    while (p + 8 <= end) { h ^= round(0, rd(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { uint32_t w; memcpy(&w, p, 4); h ^= (uint64_t)w * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
  }
 private:
  static constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                            P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
  static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
  static uint64_t rd(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
  static uint64_t round(uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; }
  uint64_t v_[4], total_, seed_;
  uint8_t mem_[32];
  unsigned n_;
};
}  // namespace zg
