// zg_simt.h — TEST-ONLY: a small SIMT emulator for kernels written against the zx_* primitives (zg_flat4.h). One workgroup
// at a time; every GPU thread is a fiber (ucontext) that runs the kernel body verbatim and yields at workgroup barriers
// and wave collectives; lanes of a wave are 64 consecutive threads. Deterministic (threads run in index order between
// synchronisation points), so a data race does not show as flakiness here — what it checks is the kernel's logic, against
// the oracle, before the same source is compiled for gfx950. Not part of the product.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <functional>
#include <vector>

namespace simt {

enum { RUN = 0, WAVE_WAIT = 1, WG_WAIT = 2, DONE = 3 };
struct Lane { ucontext_t ctx; char* stack = nullptr; int state = RUN; };
struct Machine {
  std::vector<Lane> lanes;
  ucontext_t sched;
  uint32_t cur = 0, T = 0;
  std::function<void()> body;
  std::vector<uint64_t> slot;      // collectives: one value per lane
  uint64_t barriers = 0;
};
inline Machine*& M() { static Machine* m = nullptr; return m; }

inline void trampoline() {
  Machine* m = M();
  m->body();
  m->lanes[m->cur].state = DONE;
  m->slot[m->cur] = 0;
  swapcontext(&m->lanes[m->cur].ctx, &m->sched);
}
inline void yield(int state) {
  Machine* m = M();
  m->lanes[m->cur].state = state;
  swapcontext(&m->lanes[m->cur].ctx, &m->sched);
}
// run `body` on T threads of one workgroup
inline void run(uint32_t T, const std::function<void()>& body) {
  Machine m;
  M() = &m;
  m.T = T; m.body = body;
  m.lanes.resize(T); m.slot.assign(T, 0);
  const size_t STK = 256 << 10;
  for (uint32_t i = 0; i < T; i++) {
    Lane& l = m.lanes[i];
    l.stack = (char*)malloc(STK);
    getcontext(&l.ctx);
    l.ctx.uc_stack.ss_sp = l.stack; l.ctx.uc_stack.ss_size = STK; l.ctx.uc_link = nullptr;
    makecontext(&l.ctx, (void (*)())trampoline, 0);
  }
  for (;;) {
    bool ran = false;
    for (uint32_t i = 0; i < T; i++)
      if (m.lanes[i].state == RUN) { m.cur = i; swapcontext(&m.sched, &m.lanes[i].ctx); ran = true; }
    bool released = false, all_done = true, all_wg = true, any_wg = false;
    for (uint32_t w = 0; w * 64 < T; w++) {
      bool ready = true, any = false;
      for (uint32_t l = w * 64; l < T && l < w * 64 + 64; l++) {
        const int s = m.lanes[l].state;
        if (s == WAVE_WAIT) any = true; else if (s != DONE) ready = false;
      }
      if (ready && any) { for (uint32_t l = w * 64; l < T && l < w * 64 + 64; l++) if (m.lanes[l].state == WAVE_WAIT) m.lanes[l].state = RUN; released = true; }
    }
    for (uint32_t i = 0; i < T; i++) {
      const int s = m.lanes[i].state;
      if (s != DONE) all_done = false;
      if (s == WG_WAIT) any_wg = true; else if (s != DONE) all_wg = false;
    }
    if (all_done) break;
    if (all_wg && any_wg) { for (uint32_t i = 0; i < T; i++) if (m.lanes[i].state == WG_WAIT) m.lanes[i].state = RUN; released = true; m.barriers++; }
    if (!ran && !released) { fprintf(stderr, "simt: deadlock (divergent barrier or collective)\n"); abort(); }
  }
  for (Lane& l : m.lanes) free(l.stack);
  M() = nullptr;
}

}  // namespace simt

// ---- the zx_* primitives on the emulator -----------------------------------------------------------------------------
#define ZX_DEV static inline
#define ZX_OOB 0x80000000u   // as on the GPU (zg_kernels.hip): far outside every resource, and adding a few bytes to it does not wrap into one
#define ZX_FRESH(v) (v)
struct ZxBuf { uint8_t* base; uint32_t bytes; };
static inline uint32_t zx_tid() { return simt::M()->cur; }
static inline void zx_barrier() { simt::yield(simt::WG_WAIT); }
static inline void zx_barrier_vm() { simt::yield(simt::WG_WAIT); }
static inline unsigned long long zx_ballot(bool p) {
  simt::Machine* m = simt::M();
  const uint32_t me = m->cur, w0 = me & ~63u;
  m->slot[me] = p ? 1 : 0;
  simt::yield(simt::WAVE_WAIT);
  unsigned long long r = 0;
  for (uint32_t l = 0; l < 64 && w0 + l < m->T; l++) if (m->lanes[w0 + l].state != simt::DONE && m->slot[w0 + l]) r |= 1ull << l;
  simt::yield(simt::WAVE_WAIT);
  return r;
}
static inline uint32_t zx_shfl_up(uint32_t v, int o) {
  simt::Machine* m = simt::M();
  const uint32_t me = m->cur, lane = me & 63u;
  m->slot[me] = v;
  simt::yield(simt::WAVE_WAIT);
  const uint32_t r = (int)lane >= o ? (uint32_t)m->slot[me - o] : v;
  simt::yield(simt::WAVE_WAIT);
  return r;
}
// the value lane l of the caller's wave holds (every lane of the wave calls it)
static inline uint32_t zx_shfl(uint32_t v, int l) {
  simt::Machine* m = simt::M();
  const uint32_t me = m->cur, w0 = me & ~63u;
  m->slot[me] = v;
  simt::yield(simt::WAVE_WAIT);
  const uint32_t r = (uint32_t)m->slot[w0 + ((uint32_t)l & 63u)];
  simt::yield(simt::WAVE_WAIT);
  return r;
}
static inline bool zx_any(bool p) { return zx_ballot(p) != 0ull; }
// the lanes of a wave run one after the other here: what they wrote to LDS is complete for all of them behind this point
static inline void zx_wave_sync() { (void)zx_ballot(true); }
static inline void zx_max_glb(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
static inline void zx_gst128(void* p, const ZxU4& v) {
  if ((uintptr_t)p % 16) { fprintf(stderr, "simt: misaligned 16-byte global store\n"); abort(); }
  memcpy(p, &v, 16);
}
static inline void zx_gst32u(void* p, uint32_t v) { memcpy(p, &v, 4); }
static inline ZxU4 zx_gld128(const void* p) {
  if ((uintptr_t)p % 16) { fprintf(stderr, "simt: misaligned 16-byte global load\n"); abort(); }
  ZxU4 r; memcpy(&r, p, 16); return r;
}
static inline void zx_or_lds(uint32_t* p, uint32_t v) { *p |= v; }
static inline void zx_min_lds(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
static inline void zx_min_lds64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
static inline void zx_min_glb(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
template <typename P> static inline ZxBuf zx_buf(P* base, uint32_t bytes) { ZxBuf b; b.base = (uint8_t*)base; b.bytes = bytes; return b; }
static inline ZxBuf zx_buf(decltype(nullptr), uint32_t) { ZxBuf b; b.base = nullptr; b.bytes = 0; return b; }
// raw buffer semantics: every dword (byte) of an access is range-checked by itself; out of range reads 0, writes nothing
// (offsets are 32-bit on the hardware: an offset near 2^32 wraps around for the later dwords of a wide access)
static inline uint32_t zx__dw(const ZxBuf& b, uint64_t off64) { const uint32_t off = (uint32_t)off64; uint32_t v = 0; if ((uint64_t)off + 4 <= b.bytes) memcpy(&v, b.base + off, 4); return v; }
static inline uint32_t zx_ld32(const ZxBuf& b, uint32_t off) { return zx__dw(b, off); }
#define ZX_ALIGNED(off, a) do { if ((off) != ZX_OOB && (off) < b.bytes && (((uintptr_t)b.base + (off)) % (a))) { fprintf(stderr, "simt: misaligned buffer access %u %% %d at %s:%d\n", (unsigned)(off), (int)(a), __FILE__, __LINE__); abort(); } } while (0)
static inline ZxU2 zx_ld64(const ZxBuf& b, uint32_t off) { ZX_ALIGNED(off, 4); ZxU2 r; r.x = zx__dw(b, off); r.y = zx__dw(b, (uint64_t)off + 4); return r; }
static inline ZxU3 zx_ld96(const ZxBuf& b, uint32_t off) { ZX_ALIGNED(off, 4); ZxU3 r; r.x = zx__dw(b, off); r.y = zx__dw(b, (uint64_t)off + 4); r.z = zx__dw(b, (uint64_t)off + 8); return r; }
static inline uint32_t zx_ld8(const ZxBuf& b, uint32_t off) { return off < b.bytes ? b.base[off] : 0u; }
static inline void zx_add_lds(uint32_t* p, uint32_t v) { *p += v; }
static inline void zx_st8(const ZxBuf& b, uint32_t off, uint32_t v) { if (off < b.bytes) b.base[off] = (uint8_t)v; }
static inline void zx_st32(const ZxBuf& b, uint32_t off, uint32_t v) { ZX_ALIGNED(off, 4); if ((uint64_t)off + 4 <= b.bytes) memcpy(b.base + off, &v, 4); }
static inline uint32_t zx_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u)); }
// packed 16-bit lanes: a - b per lane; 0xFFFF per lane whose signed value is negative
static inline uint32_t zx_pksub16(uint32_t a, uint32_t b) { return ((a - b) & 0xFFFFu) | (((a >> 16) - (b >> 16)) << 16); }
static inline uint32_t zx_pksign16(uint32_t a) { return ((a & 0x8000u) ? 0xFFFFu : 0u) | ((a & 0x80000000u) ? 0xFFFF0000u : 0u); }
