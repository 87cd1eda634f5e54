// zg_huf.h — body of zg_k_huf: Huffman literal streams (decompress_literals, literals_section_decoder.rs:40-158; HuffmanDecoder,
// huff0_decoder.rs:25-53).
//
// A stream is a serial chain (peek max_bits bits -> table -> consume num_bits), up to 32 K symbols long, but Huffman codes
// SELF-SYNCHRONISE: a decoder started at a wrong bit position falls onto the true code boundaries after a few symbols. So one
// WAVE decodes one stream, 64 x 128 bits at a time (a window): lane l starts 32 bits above its chunk (warm-up, not recorded),
// decodes the chunk and reports where it left it; then every lane whose entry differs from its upper neighbour's exit decodes
// again from that exit, until all agree — lane 0 starts at the true position, so by induction all of them are then on the true
// path. Symbols are staged per lane in LDS, counted, prefix-summed and written out; the exit of the last lane is the next
// window's true entry. One workgroup = up to GROUP streams that share a table, staged once per workgroup.
//
// The step of a lane's chain is ONE dependent LDS round trip: the lane keeps the 64 staged bits around its position in two
// registers (refilled a dword at a time from the staged window, one dword ahead: the refill is never waited for), so the
// only load a symbol waits for is its table entry. (Round 3 read two window dwords from LDS per symbol, then the entry: two
// dependent trips, ~360 cycles per symbol with 26 waves per CU contending for the LDS.)
//
// Where the symbols go: the literals arena — or, when the submit runs its literals AFTER the position scan
// (ZG_FLAG_LIT_DIRECT: literal-heavy input, chosen by the host), for a block without sequences straight to the block's place in
// the output: the literals ARE the block (block_decoder.rs:184-194), and the arena -> output copy of zg_k_lit goes away.
//
// Written against the zx_* primitives (zg_kernels.hip maps them onto gfx950 builtins, tests/emu/zg_simt.h onto the CPU
// emulator): tests/test_huf_cpu.py runs this source against the serial model and the oracle's literals.
#pragma once
#include <stdint.h>
#include "zg_types.h"
#include "zg_dev.h"

#define ZG_HP_CB 128                        // bits per lane and window ...
#define ZG_HP_CB_DENSE 32                   // ... or this many, once a chunk held more than ZG_HP_ROWS symbols
#define ZG_HP_ROWS 48                       // symbols a lane can record per chunk (LDS, and with it the number of streams a CU decodes at once)
#ifndef ZG_HP_WARM
#define ZG_HP_WARM 32                       // bits a lane starts above its chunk, to be on a code boundary when it enters it (<= ZG_HP_CB: they are staged)
#endif
#define ZG_HP_WBYTES (64 * ZG_HP_CB / 8)    // stream bytes covered by a window
#define ZG_HP_LOW 96                        // staged bits below the window's lowest chunk: an 11-bit peek below a chunk's end + the two dwords the register window holds below it
#define ZG_HP_STAGE (ZG_HP_WBYTES + 64)     // staged: the window, ZG_HP_LOW bits below it, alignment slack, the dword above the entry position

// (measurement on the CPU emulator only, tools/dev/huf_passes.py: windows, passes, wave steps = the busiest lane's per pass, lane steps)
#ifdef ZG_HUF_STATS
extern "C" unsigned long long zg_huf_stats[8];
#define ZG_HUF_STAT(x) x
#else
#define ZG_HUF_STAT(x)
#endif

template <int GROUP>
struct ZgHufLds {
  uint16_t tab[ZG_HUF_SLOT_U16];
  __attribute__((aligned(16))) uint8_t win[GROUP][ZG_HP_STAGE];
  // A lane's symbols, four to a dword: symbol i of lane l is byte i % 4 of dword [i / 4][l]. A lane collects four symbols in a
  // register and stores them at once (all lanes of a wave are at the same symbol count while they decode), and the output loop
  // moves a dword per iteration. A chunk of cb bits holds at most cb symbols, but 128 rows per wave would be most of the kernel's
  // LDS for a case that needs codes of < 3 bits on average: a window in which a chunk overflows ZG_HP_ROWS is decoded again, and
  // the rest of the stream with it, in chunks of ZG_HP_CB_DENSE bits (which cannot overflow).
  uint32_t sym[GROUP][ZG_HP_ROWS / 4][64];
};
static_assert(ZG_HP_WARM <= ZG_HP_CB && ZG_HP_CB_DENSE <= ZG_HP_ROWS && ZG_HP_CB_DENSE <= ZG_HP_CB && ZG_HP_ROWS % 4 == 0, "the dense chunk size must fit the rows");
static_assert(ZG_HP_WBYTES + (ZG_HP_LOW + 7) / 8 + 15 + 8 <= ZG_HP_STAGE, "staged bytes: window + low margin + alignment + the entry's dword");

// The streams of a block's literals run in different waves: which error is reported must not depend on who is first.
// rank 0 is the most significant; the word keeps (255 - rank) << 8 | status, the largest wins (zg_k_merge strips the rank).
ZX_DEV void zg_huf_set_status(uint32_t* status, uint32_t b, uint32_t rank, int st) {
  if (st) zx_max_glb(&status[b], ((255u - rank) << 8) | (uint32_t)st);
}

template <int GROUP>
ZX_DEV void zg_huf_group(const ZgBatchDev& d, const uint32_t gi, ZgHufLds<GROUP>& S) {
  if (d.totals[2]) return;                            // (the output was sized in advance and the frames produce more: Batch::sync() repeats this stage)
  const ZgHufGroup grp = d.huf_groups[gi];
  const uint32_t t = zx_tid(), wv = t >> 6, lane = t & 63;
  unsigned max_bits = grp.slot >= 0 ? d.huf_maxbits[grp.slot] : 0;
  if (max_bits > 11) max_bits = 0;
  if (max_bits) {
    const uint16_t* g = d.huf_arena + (uint64_t)grp.slot * ZG_HUF_SLOT_U16;
    for (uint32_t i = t; i < (1u << max_bits); i += 64 * GROUP) S.tab[i] = g[i];
  }
  zx_barrier_vm();
  if (wv >= grp.nitems) return;                       // whole waves leave: no workgroup barrier below
  const uint32_t item = d.huf_items[grp.first_item + wv];
  const uint32_t b = item >> 2, k = item & 3;
  const ZgBlock blk = d.blocks[b];
  // literals of a block without sequences go straight to its place in the output when the submit decodes them after the scan
  bool direct = (d.flags & ZG_FLAG_LIT_DIRECT) != 0u && blk.nseq == 0u;
  if (direct && !d.pos[b].active) {
    // behind its frame's first failing block: the reference never gets there. The failing block itself, when what stops it is its
    // sequences section header (the host found that; ZgBlock::seq_host_status): its literals are decoded all the same — their verdict
    // comes first (zg_k_litfix) — but into the arena: the block has no place in the output
    if (!blk.seq_host_status || (uint32_t)b != d.frames[blk.frame].first_block + d.frame_out[blk.frame].good_blocks) return;
    direct = false;
  }
  // the checks of the stream header: every lane computes the same, lane 0 reports
  int hst = ZG_OK;
  const uint8_t* sp = nullptr;
  uint32_t slen = 0, doff = 0, cap = 0;
  if (max_bits == 0) hst = ZG_LIT_UNINIT_HUF;                                     // literals_section_decoder.rs:60-63
  else if (d.tab_status[b]) return;                                                // its own tree description failed
  else {
    const uint32_t desc = blk.lit_type == ZG_LT_COMPRESSED ? d.aux[b].huf_desc_bytes : 0;
    if (desc > blk.lit_comp_size) hst = ZG_INTERNAL;
    else {
      const uint8_t* pay = d.src + blk.src_off + blk.lit_off + desc;
      const uint32_t total = blk.lit_comp_size - desc, regen = blk.regen_size;
      if (blk.nstreams == 4) {
        if (total < 6) hst = ZG_LIT_MISSING_JUMP;
        else {
          const uint32_t j1 = zg_ld16(pay), j2 = j1 + zg_ld16(pay + 2), j3 = j2 + zg_ld16(pay + 4);
          const uint32_t rest = total - 6;
          if (rest < j3) hst = ZG_LIT_MISSING_BYTES;
          else {
            const uint32_t start = k == 0 ? 0 : k == 1 ? j1 : k == 2 ? j2 : j3;
            const uint32_t end = k == 0 ? j1 : k == 1 ? j2 : k == 2 ? j3 : rest;
            sp = pay + 6 + start; slen = end - start;
            const uint32_t seg = (regen + 3) / 4;
            doff = k * seg; if (doff > regen) doff = regen;
            cap = k < 3 ? seg : regen - doff;
            if (cap > regen - doff) cap = regen - doff;
          }
        }
      } else { sp = pay; slen = total; doff = 0; cap = regen; }
    }
  }
  uint32_t lastb = 0;
  if (!hst) {
    lastb = slen ? sp[slen - 1] : 0;
    if (slen == 0 || lastb == 0) hst = ZG_LIT_EXTRA_PADDING;                       // :98-109
  }
  // (what is wrong with the section as a whole — jump table, sizes, no table — is found before any stream is looked at: rank 0. The
  //  padding of a stream's last byte is checked when THAT stream's turn comes, literals_section_decoder.rs:98-109: it ranks with its
  //  stream, behind a bitstream mismatch of an earlier one)
  if (hst) { if (lane == 0) zg_huf_set_status(d.lit_status, b, hst == ZG_LIT_EXTRA_PADDING ? (uint32_t)k : 0u, hst); return; }
  const uint32_t hb = zg_hbit(lastb) - 1;                     // payload bits of the last byte (below the marker)
  const int32_t T = (int32_t)((slen - 1) * 8 + hb);           // bits of the stream; position P = bits not yet consumed
  const int64_t A = (int64_t)(uintptr_t)sp;                   // address of stream bit 0
  uint8_t* dst = direct ? d.dst + d.frame_out[blk.frame].out_base + d.pos[b].out_base + doff : d.lit_arena + blk.lit_base + doff;
  uint8_t* win = S.win[wv];
  uint32_t* sym = &S.sym[wv][0][lane];                        // the lane's column: dword [j] at sym[64 j]
  const uint32_t psh = 32u - max_bits;
  int32_t top = T;                                            // true entry position of the window
  uint32_t ndone = 0;
  bool overflow = false;
  int32_t cb = ZG_HP_CB;                                      // bits per lane in this window
  while (top > 0) {
    // ---- stage the bytes that hold bits [top - 64 cb - ZG_HP_LOW, top + 32): 16-byte pieces, zeros below the stream start
    const int64_t lowbit = (int64_t)top - 64 * ZG_HP_CB - ZG_HP_LOW;
    const int64_t wb0 = (A + (lowbit >> 3)) & ~15ll;          // address of staged byte 0 (may lie below the stream)
    for (uint32_t pc = lane; pc < ZG_HP_STAGE / 16; pc += 64) {
      const int64_t addr = wb0 + 16 * (int64_t)pc;
      ZxU4 v; v.x = v.y = v.z = v.w = 0;
      if (addr + 16 > A) {
        v = zx_gld128((const void*)(uintptr_t)addr);
        if (addr < A) {                                       // piece straddles the stream start: zero the bytes below it
          const uint32_t zb = (uint32_t)(A - addr);           // 1..15
          uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; i++) { const uint32_t lo = 4u * i; q[i] = zb >= lo + 4 ? 0u : zb > lo ? q[i] & (0xFFFFFFFFu << (8 * (zb - lo))) : q[i]; }
          v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3];
        }
      }
      *(ZxU4*)(win + 16 * pc) = v;
    }
    zx_wave_sync();                                           // (LDS of one wave is in order: this costs nothing on the GPU)
    const int32_t wq0 = (int32_t)((wb0 - A) * 8);             // stream bit index of staged bit 0
    const uint32_t* win32 = (const uint32_t*)win;
    const int32_t U = top - (int32_t)lane * cb, L = U - cb > 0 ? U - cb : 0;
    bool spill = false;                                         // more symbols in the chunk than rows
    uint32_t n = 0;
    int32_t entry = U;                                          // where the lane's recorded symbols start
    // one pass over the lane's chunk from position `from` (> L): symbols to S.sym, returns the exit position.
    // Positions are counted in STAGED bits here (ps = position - wq0 > 0). Register window: hi:lo = the staged dwords that hold
    // bits [32 (ps / 32 - 1), 32 (ps / 32 + 1)): the 32 bits below the position are alignbit(hi, lo, ps) whatever ps is (the
    // instruction looks at ps % 32 only), and the position has left the upper dword when ps / 32 changes — a code has at most
    // 11 bits, so that is (ps ^ ps') > 31. The staged dword below lo is what a refill shifts in: requested with every table
    // entry, used after that entry has arrived, never waited for.
    const int32_t Us = U - wq0, Ls = L - wq0;
    auto pass = [&](int32_t from) -> int32_t {
      uint32_t ps = (uint32_t)(from - wq0);                     // >= ZG_HP_LOW + 1: the three staged dwords exist
      const uint32_t* qa = win32 + ((ps >> 5) - 2u);            // the dword below lo
      uint32_t lo = qa[1], hi = qa[2];
      uint32_t acc = 0;                                         // the last (up to four) symbols, newest in the top byte
#define ZG_HUF_STEP(BODY)                                                                               \
      {                                                                                                  \
        const uint32_t nxt = *qa;                                                                        \
        const uint32_t e = S.tab[zx_alignbit(hi, lo, ps) >> psh];                                        \
        BODY                                                                                             \
        uint32_t nb = e >> 8;                                                                            \
        nb = nb > 1u ? nb : 1u;   /* every code has >= 1 bit; the max keeps a corrupted entry from stalling the loop */ \
        const uint32_t pn = ps - nb;                                                                     \
        const bool need = (ps ^ pn) > 31u;                      /* shift the window down by one dword */ \
        ps = pn;                                                                                         \
        hi = need ? lo : hi; lo = need ? nxt : lo;                                                       \
        qa -= need ? 1 : 0;                                                                              \
      }
      // the warm-up above the chunk: nothing is recorded (a loop of its own: its step is a third shorter than a recording one)
      ZG_HUF_STAT(unsigned long long steps_ = 0;)
      while ((int32_t)ps > Us) ZG_HUF_STEP(ZG_HUF_STAT(steps_++;))
      entry = (int32_t)ps + wq0;                                // where the lane's recorded symbols start
      n = 0;
      // the chunk. Every lane that is still decoding is at the same symbol count n: every fourth step they store a dword together;
      // a chunk with more symbols than rows keeps writing its last row and is found out by its count. (With a byte store per
      // symbol the step is two instructions shorter — measured 4 % faster on its own — but the output loop below then moves a
      // byte per iteration, and that loop was a fifth of the kernel's instructions.)
      while ((int32_t)ps > Ls)
        ZG_HUF_STEP(acc = zx_alignbit(e, acc, 8u); if ((n & 3u) == 3u) sym[64u * (n / 4u < ZG_HP_ROWS / 4u - 1u ? n / 4u : ZG_HP_ROWS / 4u - 1u)] = acc; n++; ZG_HUF_STAT(steps_++;))
#undef ZG_HUF_STEP
      // the symbols behind the lane's last full dword (they sit in the top n % 4 bytes)
      if (n & 3u) sym[64u * (n / 4u < ZG_HP_ROWS / 4u - 1u ? n / 4u : ZG_HP_ROWS / 4u - 1u)] = acc >> (8u * (4u - (n & 3u)));
      ZG_HUF_STAT(zg_huf_stats[3] += steps_; if (steps_ > zg_huf_stats[7]) zg_huf_stats[7] = steps_;)
      spill = spill || n > ZG_HP_ROWS;
      return (int32_t)ps + wq0;
    };
    const bool active = U > 0;
    int32_t E = U;
    if (active) E = pass(lane ? U + (cb < ZG_HP_WARM ? cb : ZG_HP_WARM) : U);   // lane 0 starts at the true position; nobody above the window's entry
    for (int round = 0; round < 64; round++) {
      const int32_t pe = (int32_t)zx_shfl_up((uint32_t)E, 1);
      const bool need = active && lane > 0 && pe != entry;
      ZG_HUF_STAT(if (lane == 0) { zg_huf_stats[1]++; zg_huf_stats[2] += zg_huf_stats[7]; zg_huf_stats[7] = 0; if (round == 0) zg_huf_stats[0]++; } if (need) zg_huf_stats[4]++;)
      if (!zx_any(need)) break;
      if (need) E = pass(pe);
    }
    if (zx_any(spill)) {                                         // only possible with cb == ZG_HP_CB
      cb = ZG_HP_CB_DENSE;
      continue;                                                 // the same window again (the staged bytes cover the smaller one)
    }
    // ---- all lanes are on the true path: count, place, write
    uint32_t incl = active ? n : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = zx_shfl_up(incl, o); if ((int)lane >= o) incl += v; }
    const uint32_t wtot = zx_shfl(incl, 63);
    // (measured in round 4: packing the window's symbols in LDS and storing them as 16-byte pieces instead of these byte stores
    //  made the kernel 6 % SLOWER — it is bound by the instructions of the decode passes, not by its stores, and the staging cost
    //  LDS, i.e. waves per CU)
    const uint32_t at = ndone + incl - (active ? n : 0u);
    if (active) {
      // four symbols per iteration: one (unaligned) dword store where all four are symbols and inside the stream's share
      const uint32_t room = at < cap ? cap - at : 0u, m = n < room ? n : room;
      uint32_t i = 0;
      for (; i + 4u <= m; i += 4u) zx_gst32u(dst + at + i, sym[16u * i]);
      if (i < m) {
        const uint32_t v = sym[16u * i];
        dst[at + i] = (uint8_t)v;                                // (one to three symbols; written out, or the compiler builds a vector loop for them)
        if (i + 1u < m) dst[at + i + 1u] = (uint8_t)(v >> 8);
        if (i + 2u < m) dst[at + i + 2u] = (uint8_t)(v >> 16);
      }
    }
    if (ndone + wtot > cap) overflow = true;                   // more symbols than its share of the section holds
    ndone += wtot;
    // the last active lane's exit is the next entry; it is <= 0 when that lane's chunk reaches the stream start
    const uint32_t nact = (uint32_t)((top + cb - 1) / cb);
    top = (int32_t)zx_shfl((uint32_t)E, (int)(nact < 64u ? nact - 1u : 63u));
    // (one of four streams: the reference only compares the TOTAL with the section's size, literals_section_decoder.rs:150-155 —
    //  the count goes on, without writes, so that zg_k_huf_uneven can tell a different split from a wrong total)
    if (overflow && blk.nstreams != 4) break;
  }
  if (blk.nstreams == 4 && lane == 0) d.lit_counts[4u * b + k] = ndone;
  int st = ZG_OK;
  if (blk.nstreams == 4 && top != 0) st = ZG_LIT_BITSTREAM_MISMATCH;        // bits_remaining != -max_bits (:116-121)
  else if (overflow || ndone != cap) st = ZG_LIT_COUNT_MISMATCH;            // :150-155 (per stream, the format's split: zg_k_huf_uneven looks at the total)
  // The reference decodes the streams in order and checks each one's end as it goes (:116-121); the symbol count is compared
  // once, after the last stream (:150-155): a stream's BitstreamReadMismatch outranks any count mismatch, an earlier stream a
  // later one.
  if (lane == 0) zg_huf_set_status(d.lit_status, b, st == ZG_LIT_BITSTREAM_MISMATCH ? k : 8u, st);
}
