// zg_flat4.h — body of the DIRECT units of zg_k_flatten: LZ77 execution (execute_sequences, sequence_execution.rs:5-54;
// DecodeBuffer::push / repeat, decode_buffer.rs:74-141) of a frame's FIRST unit (a unit = a run of consecutive blocks of one
// frame, one workgroup each), at DWORD granularity: a lane owns groups of four consecutive output bytes through every phase of
// a tile, so the rank query, the record fetches, the gathers of what lies in front of the tile and the stores are paid once per
// group instead of once per byte.
//
// The first unit of a frame that starts from nothing (no dictionary, no earlier submit) copies from nothing outside itself:
// everything a match can reach is final once its tile is done. So the pointer machinery of the flatten (tile bytes point to
// their parents, pointer jumping finds every byte's root: a literal byte, or a match byte whose parent lies before the tile)
// resolves byte VALUES here, the plaintext is written right away, and the unit needs neither scratch words nor a sweep
// step. Units further back in a frame go through zg_flat1_unit (zg_kernels.hip): effective offsets + zg_k_sweep. (A
// pointer-mode variant of this body existed during round 3 and measured 18 % slower than zg_flat1_unit; see DESIGN.md.)
//
// The file is written against a small set of primitives (zx_*): zg_kernels.hip maps them onto gfx950 builtins,
// tests/emu/zg_emu_flat.cpp onto a fiber-based SIMT emulator, so the very same source runs on the CPU in the
// not-gpu tests (against the oracle's plaintext) before it ever meets a GPU.
//
// Tile geometry: T threads, TS bytes, GPT = TS / (4 T) groups per thread; thread t owns groups t, t + T, ... (adjacent
// lanes = adjacent groups: output stores coalesce, LDS accesses are conflict-free). A tile starts at a position whose
// output address is a multiple of four: up to three DEAD bytes in front of its first live byte.
#pragma once
#include <stdint.h>
#include "zg_types.h"
#include "zg_dev.h"
#include "zg_flat1.h"



// four 16-bit lanes (a01 = lanes 0 and 1, a23 = lanes 2 and 3) -> one byte each: the lanes' low / high bytes
ZX_DEV uint32_t zg_lanes_lo(uint32_t a23, uint32_t a01) { return (a01 & 0xFFu) | ((a01 >> 8) & 0xFF00u) | ((a23 << 16) & 0xFF0000u) | ((a23 << 8) & 0xFF000000u); }
ZX_DEV uint32_t zg_lanes_hi(uint32_t a23, uint32_t a01) { return ((a01 >> 8) & 0xFFu) | ((a01 >> 16) & 0xFF00u) | ((a23 << 8) & 0xFF0000u) | (a23 & 0xFF000000u); }
// (zx_bfi(m, a, b) = (a & m) | (b & ~m): zg_flat1.h)

template <int T, int TS, int SPT>
struct ZgFlat4Lds {
  static constexpr int NW = TS / 32, SOFF = SPT * T;
  ZxU2 rec[SOFF];                                        // per sequence of the tile: {offset, first match byte (tile-relative, 15 bits) | literal index of tile byte 0 (mod 2^17) << 15}.
                                                         // (8 bytes since round 5 — 16 before: with 8 KiB tiles the unit's LDS is 34 KB, two 1024-thread workgroups per CU)
  __attribute__((aligned(16))) uint16_t par[TS + 8];     // 0xFFFF literal, 0x8000 match byte with its parent before the tile, else tile-relative parent; [TS]: a dummy byte that is always a root
  __attribute__((aligned(16))) uint8_t val[TS];          // a root's byte value
  uint32_t bits[NW];                                     // marks: the first tile byte of every sequence
  uint16_t cnt[NW];                                      // marks before each word of bits
  uint32_t wtot[NW / 64];
  uint32_t next, cut, err;
  unsigned long long bad;                                // first failing sequence of the block: index << 32 | match position << 8 | provisional status
};

template <int T, int TS, int SPT>
ZX_DEV void zg_flat4_unit(const ZgBatchDev& d, const uint32_t ui, ZgFlat4Lds<T, TS, SPT>& L) {
  constexpr int GPT = TS / (4 * T);            // groups per thread
  constexpr int SOFF = SPT * T;                // sequences a tile takes; a denser tile is cut short
  constexpr int NW = TS / 32;                  // words of the mark bitmap
  static_assert(GPT * 4 * T == TS && GPT >= 1 && GPT <= 4 && NW <= T && (NW % 64) == 0 && SPT >= 1 && SPT <= 2 && TS <= 0x4000, "shape");
  static_assert(sizeof(ZgFlat4Lds<T, TS, SPT>) <= 81920 || TS > 8192, "an 8 KiB-tile shape is meant to fit twice into a CU's LDS");
  const uint32_t t = zx_tid();
  const ZgUnit un = d.units[ui];
  if (d.totals[2]) return;
  const ZgFrameOut fo = d.frame_out[un.frame];
  if (!fo.fast) return;
  const uint64_t unit_abs0 = d.pos[un.first_block].out_base;     // frame-relative position of the unit's first byte
  uint8_t* out_u = d.dst + fo.out_base + unit_abs0;
  const uint32_t ualign = (uint32_t)(fo.out_base + unit_abs0) & 3u;   // groups are aligned in the output
  // what the unit can hold at most — and never more than what is left of the frame: tile bytes behind a block's end are
  // classified (and their windows requested) like live ones, and the output allocation ends with the last frame
  const uint64_t fleft = fo.out_size - unit_abs0;
  const uint32_t ucap = (uint64_t)un.nblocks * ZG_FLAT_MAX < fleft ? un.nblocks * ZG_FLAT_MAX : (uint32_t)fleft;
  // output byte u: offset u + ualign + 4 (the base is dword-aligned; the engine keeps 256 bytes in front of every output)
  const ZxBuf out_rs = zx_buf(out_u - ualign - 4, ucap + ualign + 4u);
  if (t == 0) { L.err = 0; L.bad = ~0ull; L.par[TS] = (uint16_t)ZG_PAR_LIT; }
  uint32_t unit_size = 0;
  zx_barrier_vm();
  for (uint32_t bi = 0; bi < un.nblocks; bi++) {
    const uint32_t b = un.first_block + bi;
    const ZgBlockPos p = d.pos[b];
    if (!p.active) break;
    const ZgBlock blk = d.blocks[b];
    const uint32_t bu0 = (uint32_t)(p.out_base - unit_abs0);      // unit-relative position of the block
    if (blk.btype != ZG_BT_COMPRESSED || blk.nseq == 0) {        // all of it is final already (zg_k_lit)
      unit_size = bu0 + blk.regen_size;
      continue;
    }
    const ZgBlockSeqOut so = d.seq_out[b];
    const uint32_t S = blk.regen_size + so.sum_ml;               // <= ZG_FLAT_MAX on this path
    unit_size = bu0 + S;
    const uint32_t nseq = blk.nseq;
    const uint8_t* body = d.src + blk.src_off;
    const bool lit_rle = blk.lit_type == ZG_LT_RLE;
    const uint8_t* lit = blk.lit_type <= ZG_LT_RLE ? body + blk.lit_off : d.lit_arena + blk.lit_base;
    const uint32_t fill4 = lit_rle ? 0x01010101u * lit[0] : 0u;
    // literal k of the block: offset k + lit_lo + 4 of a dword-aligned resource that starts 4 bytes in front of the literals'
    // dword (a group's window may start up to 3 bytes before its first literal); RLE literals: nothing is fetched, fill4 is the value
    const uint32_t lit_lo = (uint32_t)((uint64_t)lit & 3u);
    const ZxBuf lit_rs = zx_buf(lit - lit_lo - 4, lit_rle ? 0u : ((blk.regen_size + lit_lo + 4u + 7u) & ~3u));
    const ZxBuf seq_rs = zx_buf(d.seq_arena + blk.seq_base, nseq * 12u);
    // bytes of the frame (and dictionary) that exist before this block: the farthest a match may reach. Offsets are < 2^30
    // and positions in the block < 2^17: once 2^31 bytes exist every offset is in reach, else 32-bit arithmetic decides.
    const uint64_t reach = p.out_base + d.frames[un.frame].prior_reach + d.frames[un.frame].dict_len;
    const bool reach_all = reach >= 0x80000000ull;
    const uint32_t reach32 = (uint32_t)reach;
    // the sequences a thread places per tile travel in registers: they are requested one tile ahead
    ZxU3 q[SPT];
    uint32_t qn[SPT];                               // third word of the record behind q (its literal index)
#define ZG_F4_FETCH(i0)                                                                        \
    _Pragma("unroll") for (int s = 0; s < SPT; s++) {                                            \
      const uint32_t i_ = (i0) + t + s * T;                                                      \
      q[s] = zx_ld96(seq_rs, i_ < nseq ? 12u * i_ : ZX_OOB);                                     \
      qn[s] = zx_ld32(seq_rs, i_ + 1 < nseq ? 12u * i_ + 20u : ZX_OOB);                          \
    }
    ZG_F4_FETCH(0)
    uint32_t i_start = 0;
    for (uint32_t t0 = 0; t0 < S;) {
      // the tile: block positions [t0a, t0a + TS), the first `lead` of them dead (they belong to the previous tile or block)
      const uint32_t lead = (ualign + bu0 + t0) & 3u;
      const uint32_t t0a = t0 - lead;                              // (wraps below zero for the first tile of an unaligned block: all arithmetic on it is modular)
      const uint32_t t1o = t0a + TS < S ? t0a + TS : S;            // where the tile ends unless it holds too many sequences (t0a + TS > 0: no wrap)
      if (t == 0) { L.next = 0xFFFFFFFFu; L.cut = 0xFFFFFFFFu; }
      if (t < NW) L.bits[t] = 0;
      zx_barrier();
      // ---- S1a: one thread per sequence i (index nseq stands for the trailing literals). It covers [a, m0) with literals
      // and [m0, m1) with its match; the part inside the tile is described by one record and one mark at its first tile byte.
#pragma unroll
      for (int s = 0; s < SPT; s++) {
        const uint32_t j = t + s * T, i = i_start + j;
        const uint32_t qx = q[s].x, qy = q[s].y, qz = q[s].z, next = i + 1 < nseq ? qn[s] & 0x1FFFFu : so.sum_ll;
        const bool valid = i <= nseq;
        uint32_t a = 0, m0 = 0, m1 = 0, lstart = 0, off = 0;
        if (i < nseq) {
          lstart = qz & 0x1FFFFu; m0 = qy & 0x1FFFFu; m1 = m0 + ((qy >> 17) | (((qz >> 17) & 7u) << 15));
          a = m0 - ((next - lstart) & 0x1FFFFu);
          off = zg_sym_resolve(qx, p.hist_init);
          // the first failing sequence (in order) decides, like the reference's in-order execution; which of the two
          // "too far" errors it is (repeat_from_dict, decode_buffer.rs:144-179) is worked out off the hot path
          if (off == 0) zx_min_lds64(&L.bad, ((unsigned long long)i << 32) | (m0 << 8) | (uint32_t)ZG_EXE_ZERO_OFFSET);              // sequence_execution.rs:28-30
          else if ((!reach_all && off > reach32 + m0) || off >= ZG_OFF_HUGE - 2u) zx_min_lds64(&L.bad, ((unsigned long long)i << 32) | (m0 << 8) | (uint32_t)ZG_EXE_OFFSET_TOO_BIG);
        } else if (i == nseq) {
          lstart = so.sum_ll; a = so.sum_ll + so.sum_ml; m0 = m1 = S;
          off = 0x3FFFFFFFu;          // (tile bytes behind the block's end are classified like any others: as roots, not as their own parents)
        }
        // first sequence that reaches beyond this tile starts the next one: sequences are in order along the lanes, so the
        // lowest lane of a wave that sees one speaks for the wave (one LDS atomic per wave, not one per sequence)
        const bool beyond = valid && (m1 > t1o || a >= t1o);
        const unsigned long long bm = zx_ballot(beyond);
        if (beyond && (t & 63u) == (uint32_t)__builtin_ctzll(bm)) zx_min_lds(&L.next, i);
        if (valid && a < t1o) {
          // (the tile's first sequence owns the dead bytes too: a mark at tile byte 0 keeps every rank query in range)
          const uint32_t st = j == 0 ? 0u : (a > t0 ? a : t0) - t0a;
          const uint32_t mr = (m0 > t0 ? (m0 < t1o ? m0 : t1o) : t0) - t0a;
          // (the literal a tile byte x of this sequence stands for is lstart + t0a - a + x: an index into the block's literals, below 2^17)
          ZxU2 r; r.x = off; r.y = mr | (((lstart + t0a - a) & 0x1FFFFu) << 15);
          L.rec[j] = r;
          zx_or_lds(&L.bits[st >> 5], 1u << (st & 31u));
          // the last sequence the tile has room for, and more follow: the tile ends with this one
          if (j == SOFF - 1 && i < nseq && m1 <= t1o) L.cut = m1;
        }
      }
      zx_barrier();
      // ---- S1b: marks before every word (prefix sum over the words)
      {
        const uint32_t tb = ZX_FRESH(t);
        uint32_t c = 0, sc = 0;
        if (tb < NW) {
          c = (uint32_t)__builtin_popcount(L.bits[tb]);
          sc = c;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const uint32_t v = zx_shfl_up(sc, o); if ((int)(tb & 63) >= o) sc += v; }
          if ((tb & 63) == 63) L.wtot[tb >> 6] = sc;
        }
        zx_barrier();
        if (tb < NW) {
          uint32_t before = sc - c;
          for (uint32_t w = 0; w < (tb >> 6); w++) before += L.wtot[w];
          L.cnt[tb] = (uint16_t)before;
        }
      }
      // every wave's stores of the previous tile (scratch words / output bytes) have reached memory before any wave gathers from them
      zx_barrier_vm();
      const uint32_t cut = L.cut;
      const uint32_t t1 = cut != 0xFFFFFFFFu ? cut : t1o;
      const uint32_t i_next = cut != 0xFFFFFFFFu ? i_start + SOFF : (L.next == 0xFFFFFFFFu ? nseq + 1 : L.next);
      if (L.bad != ~0ull) break;
      const uint32_t n = t1 - t0a;                                 // tile bytes [lead, n) are live
      const uint32_t tu0a = bu0 + t0a;                             // unit-relative position of tile byte 0 (modular)
      if (t1 < S) { ZG_F4_FETCH(i_next) }                          // next tile's sequences: in flight behind this tile's work
      // ---- S1c: every group finds its (at most two: matches are >= 3 bytes long) sequences by rank of the marks, and every byte
      // becomes a literal, a match byte with its parent inside the tile (pointer), or a root: a match byte whose parent lies
      // before the tile. The four bytes are classified together, as two pairs of 16-bit lanes (packed arithmetic): the kernel is
      // bound by the instructions it issues. What the roots of a group need from before the tile — the parents' byte values —
      // is consecutive in memory per sequence: one window load per sequence of the group, requested here and consumed after
      // the pointer jumping (the round trip hides behind it). A window is requested whether or not its bytes turn out to be
      // roots: what a non-root byte receives is never looked at.
      ZxU2 VA[GPT], VB[GPT], LW[GPT];
      uint32_t meta[GPT], litl[GPT];   // meta: [2:0] first byte of the second sequence, [12:8] / [20:16] / [28:24] funnel shifts of the literal / A / B windows; litl: literal bytes (byte mask)
      {
        const uint32_t tc = ZX_FRESH(t);
        const uint32_t lead2 = lead * 0x10001u;
        uint32_t wordv[GPT], cntv[GPT];
#pragma unroll
        for (int k = 0; k < GPT; k++) { const uint32_t xw = (tc + k * T) >> 3; wordv[k] = L.bits[xw]; cntv[k] = L.cnt[xw]; }
        ZxU2 rA[GPT], rB[GPT];
        uint32_t fbv[GPT];
#pragma unroll
        for (int k = 0; k < GPT; k++) {
          const uint32_t x0 = 4u * (tc + k * T), sh = x0 & 31u;
          const uint32_t ra = cntv[k] + (uint32_t)__builtin_popcount(wordv[k] & (0xFFFFFFFFu >> (31u - sh))) - 1u;   // marks up to and including x0, minus one
          const uint32_t mk = (wordv[k] >> sh) & 0xEu;                 // a second sequence starts at byte 1, 2 or 3 of the group
          fbv[k] = mk ? (uint32_t)__builtin_ctz(mk) : 4u;
          rA[k] = L.rec[ra]; rB[k] = L.rec[ra + (mk ? 1u : 0u)];
        }
#pragma unroll
        for (int k = 0; k < GPT; k++) {
          const uint32_t x0 = 4u * (tc + k * T), fb = fbv[k];
          // lanes: bytes (0, 1) and (2, 3) of the group; mB: the lanes of the second sequence
          const uint32_t x01 = x0 * 0x10001u + 0x10000u, x23 = x01 + 0x20002u;
          const uint32_t mB01 = fb == 1u ? 0xFFFF0000u : 0u, mB23 = fb <= 2u ? 0xFFFFFFFFu : (fb == 3u ? 0xFFFF0000u : 0u);
          const uint32_t oa = rA[k].x < 0x7FF0u ? rA[k].x : 0x7FF0u, ob = rB[k].x < 0x7FF0u ? rB[k].x : 0x7FF0u;   // (a tile is at most 2^14 bytes: a larger offset leads in front of it all the same, and the lanes stay inside 16 signed bits)
          const uint32_t oa2 = oa * 0x10001u, ob2 = ob * 0x10001u, ma2 = (rA[k].y & 0x7FFFu) * 0x10001u, mb2 = (rB[k].y & 0x7FFFu) * 0x10001u;
          const uint32_t p01 = zx_pksub16(x01, zx_bfi(mB01, ob2, oa2)), p23 = zx_pksub16(x23, zx_bfi(mB23, ob2, oa2));   // tile-relative parents
          const uint32_t e01 = zx_pksign16(zx_pksub16(p01, lead2)), e23 = zx_pksign16(zx_pksub16(p23, lead2));           // lanes whose parent lies before the tile's live bytes
          const uint32_t l01 = zx_pksign16(zx_pksub16(x01, zx_bfi(mB01, mb2, ma2))), l23 = zx_pksign16(zx_pksub16(x23, zx_bfi(mB23, mb2, ma2)));   // literal lanes (x < first match byte)
          ZxU2 pp;
          pp.x = zx_bfi(e01, ZG_PAR_EXIT * 0x10001u, p01) | l01;        // literal: 0xFFFF, root: 0x8000, else the parent
          pp.y = zx_bfi(e23, ZG_PAR_EXIT * 0x10001u, p23) | l23;
          *(ZxU2*)&L.par[x0] = pp;
          uint32_t m = fb;
          const int32_t uA = (int32_t)(tu0a + x0 - rA[k].x), uB = (int32_t)(tu0a + x0 - rB[k].x);   // unit-relative positions where the parents' windows start
          // the literal bytes of a group belong to one sequence (a second literal run would need a whole match between them):
          // one 8-byte window that starts at the first of them
          // literal bytes as a byte mask (the tile's dead bytes are literals by class: not these)
          const uint32_t lb = zg_lanes_lo(l23, l01) & (x0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8u * lead));
          const uint32_t ilit = lb ? (uint32_t)__builtin_ctz(lb) >> 3 : 0u;
          const uint32_t zl = (ilit < fb ? rA[k].y : rB[k].y) >> 15;
          const uint32_t ol = ((zl + x0 + ilit) & 0x1FFFFu) + lit_lo + 4u - ilit;   // offset of the byte group byte 0 would stand for
          LW[k] = zx_ld64(lit_rs, lb ? ol & ~3u : ZX_OOB);
          litl[k] = lb;
          const uint32_t oA = (uint32_t)uA + ualign + 4u, oB = (uint32_t)uB + ualign + 4u;
          // (a window may start up to three bytes before the frame: its first bytes are then literals or belong to the other sequence)
          VA[k] = zx_ld64(out_rs, uA >= -3 ? oA & ~3u : ZX_OOB);
          VB[k] = zx_ld64(out_rs, (uB >= -3 && fb < 4u) ? oB & ~3u : ZX_OOB);
          m |= ((ol & 3u) << 11) | ((oA & 3u) << 19) | ((oB & 3u) << 27);
          meta[k] = m;
        }
      }
      zx_barrier();
      // ---- S2: asynchronous pointer jumping. A byte's pointer only ever moves to another of its ancestors, so stale reads
      // are harmless and no barrier is needed between visits; a byte is done when its pointer's pointer is a root marker.
      // BYTE-STRIDED here (thread t visits tile bytes t, t + T, ...: the ownership of zg_flat1.h), not group-wise like the phases
      // around it: the four bytes of a group share their fate — same chain, same depth — and a wave lasts as long as its busiest
      // lane, so group-wise jumping (round 3: two hops per visit, branch-free, 93 vector instructions per round of four visits)
      // cost twice what the strided loop costs. Which of its bytes still have a parent inside the tile a thread reads off the
      // pointers themselves (S1c's bit mask belongs to the thread that classified the group).
      {
        const uint32_t t2 = ZX_FRESH(t);
        uint32_t open = 0;
        constexpr int PERB = TS / T;
#pragma unroll
        for (int k = 0; k < PERB; k++) open |= L.par[t2 + (uint32_t)k * T] < ZG_PAR_EXIT ? 1u << k : 0u;
        for (uint32_t guard = 0; open && guard < (1u << 16); guard++) {
          uint32_t m = open, kk[4], pp[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { kk[j] = m ? (uint32_t)__builtin_ctz(m) : 32u; m &= m - 1; }
#pragma unroll
          for (int j = 0; j < 4; j++) pp[j] = kk[j] < 32u ? L.par[t2 + kk[j] * T] : (uint32_t)TS;
#pragma unroll
          for (int j = 0; j < 4; j++) pp[j] = L.par[pp[j]];               // ([TS]: the dummy root)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (kk[j] < 32u) {
              if (pp[j] >= ZG_PAR_EXIT) open &= ~(1u << kk[j]);           // its pointer is the root
              else L.par[t2 + kk[j] * T] = (uint16_t)pp[j];               // u16 stores are atomic
            }
          }
        }
        if (open) L.err = ZG_INTERNAL;   // cannot happen: every visit moves a pointer up its chain (seen by everybody behind the next barrier)
      }
      // ---- S3a: the windows requested in S1c have arrived: every root's value is published
      {
        const uint32_t t3 = ZX_FRESH(t);
#pragma unroll
        for (int k = 0; k < GPT; k++) {
          const uint32_t x0 = 4u * (t3 + k * T), m = meta[k], fb = m & 7u;
          const uint32_t vA = zx_alignbit(VA[k].y, VA[k].x, (m >> 16) & 31u), vB = zx_alignbit(VB[k].y, VB[k].x, (m >> 24) & 31u);
          const uint32_t l4 = zx_alignbit(LW[k].y, LW[k].x, (m >> 8) & 31u) | fill4;
          const uint32_t mB = fb >= 4u ? 0u : 0xFFFFFFFFu << (8u * fb);                       // bytes of the second sequence
          *(uint32_t*)&L.val[x0] = zx_bfi(litl[k], l4, zx_bfi(mB, vB, vA));
        }
      }
      zx_barrier();
      if (L.err) break;
      // ---- S3b: every byte takes the value its root holds -> output, 4 bytes per group
      {
        const uint32_t t4 = ZX_FRESH(t);
#pragma unroll
        for (int k = 0; k < GPT; k++) {
          const uint32_t x0 = 4u * (t4 + k * T);
          const ZxU2 pp = *(const ZxU2*)&L.par[x0];
          const uint32_t pr[4] = {pp.x & 0xFFFFu, pp.x >> 16, pp.y & 0xFFFFu, pp.y >> 16};
          const bool full = x0 >= lead && x0 + 4u <= n;
          uint32_t r[4];
#pragma unroll
          for (int i = 0; i < 4; i++) r[i] = pr[i] >= ZG_PAR_EXIT ? x0 + i : pr[i];
          uint32_t vb[4];
#pragma unroll
          for (int i = 3; i >= 0; i--) vb[i] = L.val[r[i]];
          const uint32_t v = vb[0] | (vb[1] << 8) | (vb[2] << 16) | (vb[3] << 24);
          const uint32_t o = tu0a + x0 + ualign + 4u;
          zx_st32(out_rs, full ? o : ZX_OOB, v);
          if (!full) {
#pragma unroll
            for (int i = 0; i < 4; i++) zx_st8(out_rs, (x0 + i - lead < n - lead) ? o + i : ZX_OOB, vb[i]);
          }
        }
      }
      zx_barrier();  // par / val / the records are reused by the next tile
      t0 = t1;
      i_start = i_next;
    }
#undef ZG_F4_FETCH
    if (L.err || L.bad != ~0ull) {
      if (t == 0) {
        const ZgFrame fr = d.frames[un.frame];
        uint32_t st = L.err;
        if (!st) {
          const unsigned long long bad = L.bad;
          const uint32_t m0 = ((uint32_t)bad >> 8) & 0x1FFFFu;
          st = (uint32_t)bad & 0xFFu;
          if (st == (uint32_t)ZG_EXE_OFFSET_TOO_BIG && p.out_base + fr.prior_out + m0 <= fr.window_size) st = ZG_EXE_DICT_TOO_SMALL;
        }
        zx_min_glb(&d.frame_out[un.frame].err_packed, ((b - fr.first_block) << 8) | st);
      }
      break;
    }
  }
  zx_barrier_vm();
  if (t == 0) { ZgUnitInfo ui2; ui2.size = unit_size; ui2.noseq = un.noseq; d.unit_info[ui] = ui2; }
}
