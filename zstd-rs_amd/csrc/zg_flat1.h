// zg_flat1.h — body of the POINTER-MODE units of zg_k_flatten: LZ77 execution (execute_sequences, sequence_execution.rs:5-54;
// DecodeBuffer::push / repeat, decode_buffer.rs:74-141) of a unit (a run of consecutive blocks of one frame, one workgroup each)
// that may copy from in front of itself: every byte is resolved to its EFFECTIVE OFFSET e (byte[pos] = byte[pos - e], where
// pos - e is a literal byte or lies in front of the unit; 0 for a literal byte) and stored in the flatten scratch; literal bytes
// go to the output right away; zg_k_sweep resolves the offsets unit after unit. (A frame's first unit is resolved to bytes by
// zg_flat4.h instead.) Byte-granular: a thread owns the tile bytes t, t + T, t + 2T ... through all phases.
//
// Written against the zx_* primitives (zg_kernels.hip maps them onto gfx950 builtins, tests/emu/zg_simt.h onto the CPU
// emulator): tests/test_flat1_cpu.py runs this source on the CPU — its scratch words against the numpy model of
// tests/lz_model.py, the swept plaintext against the oracle — before it meets a GPU.
#pragma once
#include <stdint.h>
#include "zg_types.h"
#include "zg_dev.h"

// (a & m) | (b & ~m)
ZX_DEV uint32_t zx_bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

// zg_k_flatten<T, TS, SPT>: T threads resolve TS-byte tiles; a thread owns the tile bytes t, t + T, t + 2T ... through all phases
// (consecutive lanes = consecutive bytes: LDS accesses are conflict-free and, above all, the scratch gathers of adjacent
// lanes fall into the same cache lines — bytes of one match have adjacent parents). SPT = sequences a thread places per
// tile; a tile that would hold more than SPT * T sequences is cut short. No phase has a data-dependent branch: loads and
// stores that depend on the data go through buffer resources with an out-of-range offset for "not needed" (no traffic).
template <int T, int TS, int SPT>
struct ZgFlat1Lds {
  static constexpr int NW = TS / 32, SOFF = SPT * T;
  uint16_t par[TS];                // 0xFFFF literal, 0x8000 match byte with its parent before the tile, else tile-relative parent
  uint32_t word[TS];               // a root's effective offset; a literal: tag + index of its value in the block's literals
  uint32_t bits[NW];               // marks: the first tile byte of every sequence
  uint16_t cnt[NW];                // marks before each word of bits
  ZxU4 rec[SOFF];   // per sequence of the tile, as S1c wants it: {offset, first match byte, 2^31 + literal index of tile byte 0, 4 * offset}
  uint32_t wtot[NW / 64];
  uint32_t next, cut, err;
  unsigned long long bad;          // first failing sequence of the block: index << 32 | match position << 8 | provisional status
};
template <int T, int TS, int SPT>
ZX_DEV void zg_flat1_unit(const ZgBatchDev& d, const uint32_t ui, ZgFlat1Lds<T, TS, SPT>& L) {
  constexpr int PER = TS / T;                   // tile bytes per thread
  constexpr int SOFF = SPT * T;                 // sequences a tile takes; a denser tile is cut short
  constexpr int NW = TS / 32;                   // words of the mark bitmap
  static_assert(PER * T == TS && PER <= 16 && NW <= T && (NW % 64) == 0 && SPT >= 1 && SPT <= 2, "shape");
  const uint32_t t = zx_tid();
  const ZgUnit un = d.units[ui];
  if (d.totals[2]) return;
  const ZgFrameOut fo = d.frame_out[un.frame];
  if (!fo.fast) return;
  const uint64_t unit_abs0 = d.pos[un.first_block].out_base;     // frame-relative position of the unit's first byte
  uint8_t* out_u = d.dst + fo.out_base + unit_abs0;
  uint32_t* og = d.og + fo.og_base + unit_abs0;
  // (a frame whose few matches zg_k_sparse copies in order has no sweep step either: nobody reads its scratch words, so they are
  //  not written — an empty resource turns the stores into no-ops; on literal-heavy data they were most of the kernel's traffic)
  const bool no_scratch = d.frames[un.frame].sparse != 0u;
  // (timing experiments, ZGPU_FLAT_MODE: bit 0 drops the scratch stores, bit 1 the cross-tile scratch gathers — wrong results, what is
  //  left is what the kernel costs without that traffic)
  const uint32_t fdbg = ZG_DEVSW(((d.flags >> 4) & 3u) | ((d.flags >> 5) & 4u));   // (bit 2, flag bit 7: S3b without the literal bytes' loads and stores)
  if (t == 0) { L.err = 0; L.bad = ~0ull; }
  uint32_t unit_size = 0;
#if defined(ZG_PROFILE_FLAT) && defined(__HIPCC__)   // per-phase cycle counters (tools/dev/flat_phases.py); costs registers, off in the product build
  unsigned long long ptc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define ZG_TICK(i) { const unsigned long long n_ = clock64(); ptc[i] += n_ - tlast; tlast = n_; }
#else
#define ZG_TICK(i)
#endif
  zx_barrier_vm();
  for (uint32_t bi = 0; bi < un.nblocks; bi++) {
    const uint32_t b = un.first_block + bi;
    const ZgBlockPos p = d.pos[b];
    if (!p.active) break;
    const ZgBlock blk = d.blocks[b];
    const uint32_t bu0 = (uint32_t)(p.out_base - unit_abs0);      // unit-relative position of the block
    if (blk.btype != ZG_BT_COMPRESSED || blk.nseq == 0) {        // all of it is final already (zg_k_lit): effective offset 0
      const uint32_t n = blk.regen_size;
      if (!un.noseq && !no_scratch) for (uint32_t i = t; i < n; i += T) og[bu0 + i] = 0u;   // (a unit without sequences has no sweep step: nobody reads its scratch)
      unit_size = bu0 + n;
      continue;
    }
    const ZgBlockSeqOut so = d.seq_out[b];
    const uint32_t S = blk.regen_size + so.sum_ml;               // <= ZG_FLAT_MAX on this path
    unit_size = bu0 + S;
    const uint32_t nseq = blk.nseq;
    const uint8_t* body = d.src + blk.src_off;
    const bool lit_rle = blk.lit_type == ZG_LT_RLE;
    const uint8_t* lit = blk.lit_type <= ZG_LT_RLE ? body + blk.lit_off : d.lit_arena + blk.lit_base;
    const uint32_t lit_fill = lit_rle ? lit[0] : 0u;
    const ZxBuf lit_rs = zx_buf(lit, lit_rle ? 0u : blk.regen_size);    // RLE literals: nothing is fetched (0), lit_fill is the value
    const ZxBuf seq_rs = zx_buf(d.seq_arena + blk.seq_base, nseq * 12u);
    // What S3b stores goes through resources that END WITH THE BLOCK: tile bytes behind a tile's end are processed like live
    // ones (no predicate per byte) — behind the block's end the stores fall out of range, inside the block they leave values
    // that the next tile, which owns those bytes, overwrites (its stores come after this tile's: zx_barrier_vm in between)
    const ZxBuf out_rs = zx_buf(out_u, bu0 + S);
    const ZxBuf og_rs = zx_buf(og, (no_scratch || (fdbg & 1u)) ? 0u : 4u * (bu0 + S));
    // bytes of the frame (and dictionary) that exist before this block: the farthest a match may reach. Offsets are < 2^30
    // and positions in the block < 2^17: once 2^31 bytes exist every offset is in reach, else 32-bit arithmetic decides.
    // (once the caller has drained bytes the dictionary is out of reach: DecodeBuffer holds nothing older than what is undrained, and
    //  total_output_counter has passed window_size by then — repeat_from_dict, decode_buffer.rs:144-179, answers OffsetTooBig)
    const ZgFrame& frr = d.frames[un.frame];
    // (round 5: with a dictionary in front the device window is [dictionary content][undrained bytes] whatever has been drained — the
    //  reference's DecodeBuffer + dict_content, FrameState::make_room —, so the dictionary is in reach of the copy as long as the reference
    //  would serve it; whether it still WOULD, total_output_counter <= window_size, is zg_k_exact's verdict: it runs for every frame with a
    //  dictionary)
    const uint64_t reach = p.out_base + frr.prior_reach + frr.dict_len;
    const bool reach_all = reach >= 0x80000000ull;
    const uint32_t reach32 = (uint32_t)reach;
    if (no_scratch) {
      // A frame whose matches zg_k_sparse copies in order (literal-heavy data: a sequence or two in one block out of ten). What is
      // left to do here is to check the offsets and to put the literal runs in place: plain copies, no tiles (through the tile
      // machinery a block costs 8 x 13 us whatever it holds; measured on 128 x 64 MiB iso-like frames, 2.4 of the pass's 17 ms).
      // Runs are few and long: the whole workgroup copies one run after the other; a block of many runs (the frame is sparse on
      // average only) spreads them over its waves. Run i = the literals in front of sequence i; run nseq = the trailing ones.
      const bool wide = nseq + 1u <= 2u * (T / 64);
      const uint32_t x0 = wide ? t : (t & 63u), step = wide ? (uint32_t)T : 64u;
      for (uint32_t i = wide ? 0u : (t >> 6); i <= nseq; i += wide ? 1u : (uint32_t)(T / 64)) {
        uint32_t lstart = so.sum_ll, a = so.sum_ll + so.sum_ml, m0 = S;
        if (i < nseq) {
          const ZxU3 r = zx_ld96(seq_rs, 12u * i);
          const uint32_t next = i + 1 < nseq ? zx_ld32(seq_rs, 12u * i + 20u) & 0x1FFFFu : so.sum_ll;
          lstart = r.z & 0x1FFFFu; m0 = r.y & 0x1FFFFu;
          a = m0 - ((next - lstart) & 0x1FFFFu);
          const uint32_t off = zg_sym_resolve(r.x, p.hist_init);
          if (x0 == 0) {                             // the same verdicts as S1a below
            if (off == 0) zx_min_lds64(&L.bad, ((unsigned long long)i << 32) | (m0 << 8) | (uint32_t)ZG_EXE_ZERO_OFFSET);
            else if ((!reach_all && off > reach32 + m0) || off >= ZG_OFF_HUGE - 2u) zx_min_lds64(&L.bad, ((unsigned long long)i << 32) | (m0 << 8) | (uint32_t)ZG_EXE_OFFSET_TOO_BIG);
          }
        }
        const uint32_t len = m0 - a;
        for (uint32_t x = x0; x < len; x += 8u * step) {   // eight bytes per thread in flight
          uint32_t v[8];
#pragma unroll
          for (int h = 0; h < 8; h++) v[h] = zx_ld8(lit_rs, x + h * step < len ? lstart + x + h * step : ZX_OOB);
#pragma unroll
          for (int h = 0; h < 8; h++) zx_st8(out_rs, x + h * step < len ? bu0 + a + x + h * step : ZX_OOB, (uint8_t)(v[h] | lit_fill));
        }
      }
      zx_barrier();
    }
    // the sequences a thread places per tile travel in registers: they are requested one tile ahead
    ZxU3 q[SPT];
    uint32_t qn[SPT];                               // third word of the record behind q (its literal index)
    auto fetch = [&](uint32_t i0) {
#pragma unroll
      for (int s = 0; s < SPT; s++) {
        const uint32_t i = i0 + t + s * T;
        q[s] = zx_ld96(seq_rs, i < nseq ? 12u * i : ZX_OOB);
        qn[s] = zx_ld32(seq_rs, i + 1 < nseq ? 12u * i + 20u : ZX_OOB);
      }
    };
    fetch(0);
    uint32_t i_start = 0;
    for (uint32_t t0 = no_scratch ? S : 0u; t0 < S;) {   // (a sparse frame's block is done: only the verdict below is left)
      const uint32_t t1o = t0 + TS < S ? t0 + TS : S;            // where the tile ends unless it holds too many sequences
      if (t == 0) { L.next = 0xFFFFFFFFu; L.cut = 0xFFFFFFFFu; }
      if (t < NW) L.bits[t] = 0;
      zx_barrier();
      ZG_TICK(0)
      // ---- S1a: one thread per sequence i (index nseq stands for the trailing literals). It covers [a, m0) with literals
      // and [m0, m1) with its match; the part inside the tile is described by one record and one mark at its first tile byte.
#pragma unroll
      for (int s = 0; s < SPT; s++) {
        const uint32_t j = t + s * T, i = i_start + j;
        // (the prefetched registers are read on every path: a load the compiler sees unconsumed on some path makes it wait
        // for everything in flight — vmcnt is in order — when the register is reused)
        const uint32_t qx = q[s].x, qy = q[s].y, qz = q[s].z, next = i + 1 < nseq ? qn[s] & 0x1FFFFu : so.sum_ll;
        const bool valid = i <= nseq;               // (no early exit: the ballot below is taken by whole waves)
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
        }
        // first sequence that reaches beyond this tile starts the next one: sequences are in order along the lanes, so the
        // lowest lane of a wave that sees one speaks for the wave (one LDS atomic per wave, not one per sequence)
        const bool beyond = valid && (m1 > t1o || a >= t1o);
        const unsigned long long bm = zx_ballot(beyond);
        if (beyond && (t & 63u) == (uint32_t)__builtin_ctzll(bm)) zx_min_lds(&L.next, i);
        if (!valid || a >= t1o) continue;
        const uint32_t st = (a > t0 ? a : t0) - t0;
        const uint32_t mr = (m0 > t0 ? (m0 < t1o ? m0 : t1o) : t0) - t0;
        // (z: the literal a tile byte x of this sequence stands for is z + x - 2^31; it only has to be right for x >= st)
        // (w: 4 * (unit position of the tile - offset), modulo 2^32: S1c adds 4 x and has the byte offset of the parent's scratch word —
        //  or, for a parent in front of the unit, a value beyond 2^31: offsets are below 2^30 and units far below 2^29 bytes)
        { ZxU4 rr; rr.x = off; rr.y = mr; rr.z = 0x80000000u + lstart + (a > t0 ? 0u : t0 - a) - st; rr.w = 4u * (bu0 + t0) - 4u * off; L.rec[j] = rr; }
        zx_or_lds(&L.bits[st >> 5], 1u << (st & 31u));
        // the last sequence the tile has room for, and more follow: the tile ends with this one
        if (j == SOFF - 1 && i < nseq && m1 <= t1o) L.cut = m1;
      }
      zx_barrier();
      // ---- S1b: marks before every word (prefix sum over the words)
      {
        const uint32_t tb = ZX_FRESH(t);   // (every phase derives its LDS addresses from its own copy of the thread index: held
                                           //  across the tile loop they do not fit the register budget, and a spilled one comes back
                                           //  through a scratch load whose wait also covers the records requested for the next tile)
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
      // every wave's scratch stores of the previous tile have reached memory before any wave gathers from them. (The builtin,
      // not inline asm: the compiler then knows that nothing is in flight here, and does not protect registers of earlier
      // loads with waits that would also cover the loads issued below.)
      zx_barrier_vm();
      ZG_TICK(1)
      const uint32_t cut = L.cut;
      const uint32_t t1 = cut != 0xFFFFFFFFu ? cut : t1o;
      const uint32_t i_next = cut != 0xFFFFFFFFu ? i_start + SOFF : (L.next == 0xFFFFFFFFu ? nseq + 1 : L.next);
      if (L.bad != ~0ull) break;
      const uint32_t n = t1 - t0;
      const uint32_t tu0 = bu0 + t0;                             // unit-relative position of the tile
      if (t1 < S) fetch(i_next);                                  // next tile's sequences: in flight behind this tile's work
      // the scratch words of the unit's EARLIER tiles, and nothing else: a parent inside this tile or in front of the unit is out
      // of this resource's range by its position alone (no fetch, no select)
      const ZxBuf og_prev = zx_buf(og, (no_scratch || (fdbg & 2u)) ? 0u : 4u * tu0);
      // ---- S1c: every byte finds its sequence (rank of the marks up to it) and becomes a literal, a match byte with its
      // parent inside the tile (pointer), or a root: a match byte whose parent lies before the tile. A root's effective
      // offset is its sequence's offset if the parent lies before the unit, else offset + e[parent]: the parent's scratch
      // word is requested here and added after the pointer jumping (the round trip hides behind it).
      uint32_t wadd[PER], unresolved = 0;
      const uint32_t tc = ZX_FRESH(t);
      constexpr int G = 2;                       // bytes worked on together: their LDS round trips overlap
#pragma unroll
      for (int k0 = 0; k0 < PER; k0 += G) {
        uint32_t word[G], cnt[G];
        ZxU4 rec[G];
#pragma unroll
        for (int g = 0; g < G; g++) { const uint32_t xw = (tc + (k0 + g) * T) >> 5; word[g] = L.bits[xw]; cnt[g] = L.cnt[xw]; }
#pragma unroll
        for (int g = 0; g < G; g++) {
          const uint32_t x = tc + (k0 + g) * T;
          // marks up to and including x, minus one (the tile's first byte carries a mark); bytes behind the tile's end get the last sequence
          rec[g] = L.rec[cnt[g] + (uint32_t)__builtin_popcount(word[g] & (0xFFFFFFFFu >> (31u - (x & 31u)))) - 1u];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
          const int k = k0 + g;
          const uint32_t x = tc + k * T;
          // (selects nested one condition at a time: combined conditions cost a scalar instruction each, and this kernel is bound by
          //  the number of instructions of any kind it issues)
          const uint32_t off = rec[g].x, m0 = rec[g].y;
          const bool c_live = x < n, c_lit = x < m0, c_in = off <= x;      // c_in (when not a literal): parent inside the tile
          // (the literal / match distinction as a MASK and bit arithmetic, not as selects between values that come from loads: the
          //  compiler turns those into branches with the loads — and their waits — inside)
          const uint32_t lm = c_lit ? 0xFFFFFFFFu : 0u;
          // the parent's scratch word is wanted for a root whose parent lies in an earlier tile of the unit: 4 * its unit position is in
          // og_prev's range exactly then. A literal byte has no parent: bit 31 puts it out of range. (Bytes behind the tile's end
          // fetch and write too: their slots are not used by anything.)
          wadd[k] = zx_ld32(og_prev, (rec[g].w + 4u * x) | (lm & ZX_OOB));
          const uint32_t par = (c_in ? x - off : (uint32_t)ZG_PAR_EXIT) | lm;   // (as u16: 0xFFFF = ZG_PAR_LIT for a literal)
          L.par[x] = (uint16_t)par;
          L.word[x] = zx_bfi(lm, rec[g].z + x, off);                      // (the word of a byte whose parent lies in the tile is never looked at: it is no root)
          uint32_t ub = c_in ? 1u << k : 0u;
          ub &= ~lm;
          ub = c_live ? ub : 0u;
          unresolved |= ub;
        }
      }
      zx_barrier();
      ZG_TICK(2)
      // ---- S2: asynchronous pointer jumping. A byte's pointer only ever moves to another of its ancestors, so stale reads
      // are harmless and no barrier is needed between rounds; a byte is done when its pointer's pointer is a root marker.
      // Each thread visits just its still-unresolved bytes (few: most parents are before the tile), four per step.
      {
        const uint32_t t2 = ZX_FRESH(t);
        for (uint32_t guard = 0; unresolved && guard < (1u << 16); guard++) {
          uint32_t m = unresolved, kk[4], pp[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { kk[j] = m ? (uint32_t)__builtin_ctz(m) : 32u; m &= m - 1; }
#pragma unroll
          for (int j = 0; j < 4; j++) pp[j] = kk[j] < 32u ? L.par[t2 + kk[j] * T] : 0u;
#pragma unroll
          for (int j = 0; j < 4; j++) pp[j] = L.par[pp[j]];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (kk[j] < 32u) {
              if (pp[j] >= ZG_PAR_EXIT) unresolved &= ~(1u << kk[j]);     // its pointer is the root
              else L.par[t2 + kk[j] * T] = (uint16_t)pp[j];               // u16 stores are atomic
            }
          }
        }
        if (unresolved) L.err = ZG_INTERNAL;   // cannot happen: every step moves a pointer up its chain (seen by everybody behind the next barrier)
      }
      ZG_TICK(3)
      // ---- S3a: the scratch words requested in S1c have arrived: the roots' effective offsets are completed in LDS
      const uint32_t t3 = ZX_FRESH(t);
#pragma unroll
      for (int k = 0; k < PER; k++) zx_add_lds(&L.word[t3 + k * T], wadd[k]);   // ds_add_u32; 0 where nothing was requested
      zx_barrier();
      ZG_TICK(4)
      if (L.err) break;
      // ---- S3b: every byte's effective offset = its root's + the distance to the root (a literal root counts 0) -> scratch;
      // the tile's literal bytes are fetched and go to the output
      constexpr int H = PER < 8 ? PER : 8;                        // bytes per batch (their loads are in flight together)
#pragma unroll
      for (int k0 = 0; k0 < PER; k0 += H) {
        uint32_t lb[H], da[H], pr[H], w[H];
        // (three passes over the batch, so that its LDS reads go out together: one round trip for the pointers, one for the words)
#pragma unroll
        for (int h = 0; h < H; h++) pr[h] = L.par[t3 + (k0 + h) * T];
        // (each of these loops takes its inputs last first: the one wait in front of the first use then covers the whole batch,
        //  where first-to-last order costs a wait instruction per element; the kernel is bound by instructions issued)
#pragma unroll
        for (int h = H - 1; h >= 0; h--) { const uint32_t x = t3 + (k0 + h) * T; w[h] = L.word[pr[h] >= ZG_PAR_EXIT ? x : pr[h]]; }
#pragma unroll
        for (int h = H - 1; h >= 0; h--) {
          const uint32_t x = t3 + (k0 + h) * T, ux = tu0 + x;
          const uint32_t r = pr[h] >= ZG_PAR_EXIT ? x : pr[h];
          const uint32_t e = ((w[h] >> 31) ? 0u : w[h]) + (x - r);
          const bool isl = pr[h] == ZG_PAR_LIT;                           // a literal byte: w carries where its value is
          lb[h] = (fdbg & 4u) ? 0u : zx_ld8(lit_rs, isl ? w[h] & 0x7FFFFFFFu : ZX_OOB);
          da[h] = isl ? ux : ZX_OOB;                                      // where its value goes
          zx_st32(og_rs, 4u * ux, e);
        }
#pragma unroll
        for (int h = 0; h < H; h++)   // (lb[0] was requested last)
          if (!(fdbg & 4u)) zx_st8(out_rs, da[h], (uint8_t)(lb[h] | lit_fill));
      }
      zx_barrier();  // L.par / L.word / the records are reused by the next tile
      ZG_TICK(5)
      t0 = t1;
      i_start = i_next;
    }
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
      unit_size = bu0;     // the unit ends in front of the block that failed: its scratch is complete up to there, and the sweep resolves the good blocks
      break;
    }
  }
  zx_barrier_vm();
  if (t == 0) { d.unit_info[ui].size = unit_size; d.unit_info[ui].noseq = un.noseq; }
#if defined(ZG_PROFILE_FLAT) && defined(__HIPCC__)
  if (t == 0 && d.dbg) { for (int i = 0; i < 8; i++) atomicAdd(&d.dbg[i], ptc[i]); }
#endif
#undef ZG_TICK
}

