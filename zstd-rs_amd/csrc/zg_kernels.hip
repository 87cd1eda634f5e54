// zg_kernels.hip — gfx950 (MI355X, CDNA4, wave64) kernels of the zstd block-decode engine.
//
// Pipeline of one submit (two HIP streams; the host reads the frame sizes once, between zg_k_scanf and zg_k_lit):
//   sequences chain (high-priority stream)
//     zg_k_ftab     one wave per block         FSE table descriptions -> FSE arena (+ the predefined tables)
//     zg_k_seq      decoder + mover wave per 16 blocks, four lanes per block
//                                              the three FSE state chains of a block -> the states of every sequence (8 B)
//     zg_k_seqpost  one workgroup per block    symbols, extra bits, values, positions, offset history (scans) -> ZgSeq[] (12 B)
//   literals chain (low-priority stream, beside zg_k_seq)
//     zg_k_tables   one wave per block         Huffman tree descriptions -> Huffman arena
//     zg_k_huf      one wave per stream        Huffman literal streams, self-synchronising -> literals arena
//     zg_k_huf_uneven  one wave per block      (rare) four streams split differently from the format's rule: placed again by their counts
//   then, on the first stream
//     zg_k_merge    one thread per block       literals errors into the block status
//     zg_k_scan     one workgroup per frame    block output positions + offset-history resolution (function-composition scan)
//     zg_k_scanf    one workgroup              frame output positions and scratch bases
//     zg_k_lit      one workgroup per block    raw and RLE blocks, blocks without sequences -> output
//     zg_k_flatten  one workgroup per unit     every byte of a run of blocks -> its effective offset (to a literal byte, or to a byte in
//                                              front of the unit); literals -> output. A frame's FIRST unit instead becomes plaintext
//                                              right here (zg_flat4.h: nothing in front of it to copy from): no scratch, no sweep step
//     zg_k_swprep   one thread per unit        sweep descriptors
//     zg_k_sweep    one launch per unit index  the units' tails, unit after unit: match bytes gathered from finished output;
//                   + one per 16 indices       their heads beside the chain on the second stream (split sweep)
//     zg_k_sparse   one wave per frame         frames with hardly any sequences: their matches in order, instead of sweep steps
//     zg_k_fin      one thread per frame       execution errors -> frame status
//     zg_k_lz       one workgroup per frame    in-order fallback (blocks regenerating > 128 KiB)
//   and, from Batch::sync(), only for submits it can matter to
//     zg_k_exact    one workgroup per frame    the reference's DecodeBuffer bookkeeping replayed exactly (zg_exact.h): which bytes are
//                                              still in reach, which of the two "offset too far" errors applies
//
// The reference functions each kernel reproduces are cited at the lane routines in zg_dev.h.
#include <stdlib.h>
#include "zg_kernels.h"
#include "zg_dev.h"

#define ZG_SEQ_G 16       // blocks per workgroup (four lanes per block in either wave) in zg_k_seq: 16 x (2.5 KiB + 1.25 KiB tables + ring + records) in LDS -> 2 workgroups per CU
#define ZG_LZ_T 256       // threads per frame in zg_k_lz

typedef uint32_t zg_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) zg_v4u zg_gv4u;   // 16 bytes in global memory: global_load/store, not flat
typedef uint32_t zg_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) zg_v2u zg_gv2u;
typedef uint32_t zg_v3u __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(1))) zg_v3u zg_gv3u;
// four bytes at any address through one dword-aligned 8-byte load and a funnel shift (a misaligned dword load is split by the hardware)
__device__ __forceinline__ uint32_t zg_ld32_fun(const uint8_t* p) {
  const uint64_t a = (uint64_t)p;
  const zg_v2u v = *(const zg_gv2u*)(a & ~3ull);
  return __builtin_amdgcn_alignbit(v.y, v.x, ((uint32_t)a & 3u) * 8u);
}

__device__ __forceinline__ void zg_set_status(uint32_t* status, uint32_t b, int st) {
  if (st) atomicCAS(&status[b], 0u, (uint32_t)st);
}
// how far back a match of the frame may reach as far as the split sweep is concerned (zg_k_seqpost reports a longer one)
__device__ __forceinline__ uint32_t zg_sweep_window(const ZgBatchDev& d, const ZgFrame& fr) {
  return d.sweep_window ? d.sweep_window : (fr.window_size > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)fr.window_size);
}

// What one lane of a wave wrote to LDS inside a divergent branch is read by the other lanes afterwards: the compiler
// reasons per thread and may otherwise move those reads ahead of a branch the reading lanes do not take.
__device__ __forceinline__ void zg_wave_publish() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that orders LDS only: unlike __syncthreads() it does not wait for this wave's global loads/stores
__device__ __forceinline__ void zg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Buffer resource over [p, p + bytes): loads through it with an offset >= bytes return 0 and cause no traffic. The inputs
// are wave-uniform; passing them through readfirstlane makes that provable to the compiler, which otherwise wraps every
// buffer load into a "waterfall" loop (one iteration, but it serialises the loads).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t zg_make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
#define ZG_OOB 0xFFFFFFFFu   // an offset no buffer resource covers
// the same value, but new to the compiler: what is computed from it is computed again here instead of being kept in a register
#define ZG_FRESH(v) ({ uint32_t v_ = (v); asm volatile("" : "+v"(v_)); v_; })

// ------------------------------------------------------------------------------------------------------------
// The zx_* primitives: kernel bodies that also run under the CPU emulator (tests/emu/zg_simt.h) are written against them —
// zg_huf.h (zg_k_huf), zg_flat4.h (direct units of zg_k_flatten), zg_exact.h (zg_k_exact).
// ------------------------------------------------------------------------------------------------------------
#define ZX_DEV __device__ __forceinline__
// "not needed": an offset no resource of the flatten covers (they are all far below 2^31 bytes). Not 0xFFFFFFFF: the compiler narrows
// a wide load whose first dwords are unused by ADDING to the offset, and 0xFFFFFFFF + 4 is 3 — inside every resource.
#define ZX_OOB 0x80000000u
#define ZX_FRESH(v) ZG_FRESH(v)
typedef __amdgpu_buffer_rsrc_t ZxBuf;
ZX_DEV uint32_t zx_tid() { return threadIdx.x; }
ZX_DEV void zx_barrier() { zg_lds_barrier(); }
// every wave's global stores have reached memory, then the barrier (the builtin, not inline asm: the compiler then knows
// that nothing is in flight here and does not protect registers of earlier loads with waits that also cover later ones)
ZX_DEV void zx_barrier_vm() { __builtin_amdgcn_s_waitcnt(0x0F70); zg_lds_barrier(); }
ZX_DEV unsigned long long zx_ballot(bool p) { return __ballot(p); }
ZX_DEV uint32_t zx_shfl_up(uint32_t v, int o) { return __shfl_up(v, o, 64); }
ZX_DEV void zx_or_lds(uint32_t* p, uint32_t v) { atomicOr(p, v); }
ZX_DEV void zx_min_lds(uint32_t* p, uint32_t v) { atomicMin(p, v); }
ZX_DEV void zx_min_lds64(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }
ZX_DEV void zx_min_glb(uint32_t* p, uint32_t v) { atomicMin(p, v); }
ZX_DEV ZxBuf zx_buf(const void* base, uint32_t bytes) { return zg_make_rsrc(base, bytes); }
ZX_DEV uint32_t zx_ld32(ZxBuf b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(b, off, 0, 0); }
ZX_DEV ZxU2 zx_ld64(ZxBuf b, uint32_t off) { const zg_v2u v = __builtin_amdgcn_raw_buffer_load_b64(b, off, 0, 0); ZxU2 r; r.x = v.x; r.y = v.y; return r; }
ZX_DEV ZxU3 zx_ld96(ZxBuf b, uint32_t off) { const zg_v3u v = __builtin_amdgcn_raw_buffer_load_b96(b, off, 0, 0); ZxU3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }
ZX_DEV uint32_t zx_ld8(ZxBuf b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b8(b, off, 0, 0); }
ZX_DEV void zx_add_lds(uint32_t* p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_add_u32
ZX_DEV void zx_st8(ZxBuf b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, b, off, 0, 0); }
ZX_DEV void zx_st32(ZxBuf b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, b, off, 0, 0); }
ZX_DEV uint32_t zx_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
typedef short zg_v2s __attribute__((ext_vector_type(2)));
// packed 16-bit lanes (v_pk_sub_i16, v_pk_ashrrev_i16): a - b per lane; 0xFFFF per lane whose signed value is negative
ZX_DEV uint32_t zx_pksub16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (zg_v2s)(__builtin_bit_cast(zg_v2s, a) - __builtin_bit_cast(zg_v2s, b))); }
// (as an instruction: written as a shift the compiler turns what is done with the result — lane masks for v_bfi — into a compare and a select per lane)
ZX_DEV uint32_t zx_pksign16(uint32_t a) { uint32_t r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
ZX_DEV uint32_t zx_shfl(uint32_t v, int l) { return (uint32_t)__shfl((int)v, l, 64); }
ZX_DEV bool zx_any(bool p) { return __any(p) != 0; }
ZX_DEV void zx_max_glb(uint32_t* p, uint32_t v) { atomicMax(p, v); }
ZX_DEV void zx_gst128(void* p, const ZxU4& v) { const zg_v4u w = {v.x, v.y, v.z, v.w}; *(zg_gv4u*)p = w; }   // 16 bytes to global memory, 16-byte aligned
// a dword to global memory at any byte address: ONE store (gfx950 runs in unaligned access mode; told about the alignment, the
// compiler takes the dword apart into bytes and reassembles them)
// (round 4 used __builtin_nontemporal_store here: zg_k_huf's lanes store dwords 16-48 bytes apart, and as streaming "nt" stores those partial
//  lines went to HBM one by one — 27.9 GB written for 8.6 GB of literals on the iso-like frames; plain stores meet again in the L2)
struct __attribute__((packed)) ZxU32Unaligned { uint32_t v; };
#ifdef ZG_HUF_ST_NT
ZX_DEV void zx_gst32u(void* p, uint32_t v) { __builtin_nontemporal_store(v, (uint32_t*)p); }
#else
ZX_DEV void zx_gst32u(void* p, uint32_t v) { ((ZxU32Unaligned*)p)->v = v; }
#endif
ZX_DEV ZxU4 zx_gld128(const void* p) { const zg_v4u v = *(const zg_gv4u*)p; ZxU4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }   // 16 bytes of global memory, 16-byte aligned
// what the lanes of a wave wrote to LDS is read by the other lanes afterwards (a wave runs in lockstep: no hardware barrier, but the
// compiler must not move the reads ahead)
ZX_DEV void zx_wave_sync() { zg_wave_publish(); }

// ------------------------------------------------------------------------------------------------------------
// zg_k_tables: Huffman tree descriptions of the literals sections (HuffmanTable::build_decoder, huff0_decoder.rs:117-124
// with :232-403; the FSE tables of the sequences chain are zg_k_ftab's). One WAVE per block:
//   weights     lane 0 reads them (direct nibbles, or FSE-compressed: read_probabilities, a 64-entry table and two
//               interleaved state chains, all bit-serial) with everything it touches in LDS: on gfx950 a load behind a
//               global store waits for that store, so nothing is written out before the table is complete;
//   build       all 64 lanes, four symbols each: weight sum and checks (:283-325), symbols per code length, a symbol's
//               rank among the symbols of its length (ballots, symbol order), and the table itself: a symbol with code
//               length b owns 2^(max_bits - b) consecutive entries (:327-403). Long runs are written by the whole wave,
//               short ones by their lane, into LDS; the finished table leaves with 16-byte stores.
// ------------------------------------------------------------------------------------------------------------
#define ZG_HT_W 2      // blocks (waves) per workgroup: 11.8 KB of LDS, so that a workgroup fits beside two zg_k_seq workgroups (147,472 of the CU's 163,840 bytes)
#define ZG_TAB_HDR 160 // bytes of a literals section staged for parsing (a tree description is at most 129 bytes)
struct ZgHufTabLds {
  int16_t probs[256];
  uint16_t counter[256];
  uint32_t fsew[64];
  uint8_t weights[264];
  uint32_t cls[16];                                       // symbols per code length
  int32_t res[4];                                         // lane 0 -> wave: status, number of weights, description bytes
  __attribute__((aligned(16))) uint8_t hdr[ZG_TAB_HDR + 32];
  __attribute__((aligned(16))) uint16_t table[ZG_HUF_SLOT_U16];
};
__global__ void __launch_bounds__(64 * ZG_HT_W) zg_k_tables(ZgBatchDev d) {
  __shared__ ZgHufTabLds s_l[ZG_HT_W];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t b = blockIdx.x * ZG_HT_W + wv;
  if (b >= d.nblocks) return;
  const ZgBlock blk = d.blocks[b];
  if (blk.host_status || blk.btype != ZG_BT_COMPRESSED) return;
  ZgHufTabLds& L = s_l[wv];
  uint32_t desc_bytes = 0;
  int st = ZG_OK;
  if (blk.lit_type == ZG_LT_COMPRESSED) {
    // the description (+8 bytes for the bit windows): dword-aligned 16-byte loads, one per lane
    const uint32_t hl = blk.lit_comp_size < 136u ? blk.lit_comp_size : 136u;   // a tree description is at most 129 bytes
    const uint8_t* g = d.src + blk.src_off + blk.lit_off;
    const uint64_t ga = (uint64_t)g & ~3ull;
    const uint32_t sh = (uint32_t)((uint64_t)g & 3u), nv = (hl + 8 + sh + 15) / 16;
    if (lane < nv) { const zg_v4u r = *(const zg_gv4u*)(ga + 16ull * lane); *(zg_v4u*)(L.hdr + 16 * lane) = r; }
    if (lane < 16) L.cls[lane] = 0;
    zg_wave_publish();
    if (lane == 0) {
      int nw = 0;
      uint32_t used = 0;
      L.res[0] = zg_huf_read_weights(L.hdr + sh, hl, L.weights, &nw, &used, L.fsew, L.probs, L.counter);
      L.res[1] = nw; L.res[2] = (int32_t)used;
    }
    zg_wave_publish();
    st = L.res[0];
    const int nw = L.res[1];
    if (!st && nw > 255) {
      // 256 weights (the two-decoder loop may end with one more than it checks for, :207-234): the implied symbol is "256",
      // stored as byte 0. Too rare for a wave version: lane 0 builds the table the serial way.
      if (lane == 0) {
        int mb = 0;
        L.res[0] = zg_huf_build(L.weights, nw, d.huf_arena + (uint64_t)blk.huf_slot * ZG_HUF_SLOT_U16, &mb);
        if (!L.res[0]) d.huf_maxbits[blk.huf_slot] = (uint8_t)mb;
      }
      zg_wave_publish();
      st = L.res[0];
      if (!st) desc_bytes = (uint32_t)L.res[2];
    } else if (!st) {
      // ---- weight sum, max_bits, the implicit last weight (:283-325)
      uint32_t w[4], sum = 0;
      bool bad = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int sy = (int)lane + 64 * k;
        w[k] = sy < nw ? L.weights[sy] : 0u;
        bad = bad || w[k] > 11u;
        sum += w[k] ? 1u << (w[k] - 1) : 0u;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      const bool anybad = __any(bad);
      uint32_t max_bits = 0, last_weight = 0;
      if (anybad || sum == 0) st = ZG_HUF_TABLE;
      else {
        max_bits = zg_hbit(sum);
        const uint32_t left = (1u << max_bits) - sum;
        if (left & (left - 1)) st = ZG_HUF_TABLE;             // LeftoverIsNotAPowerOf2 (left >= 1 always)
        else if (max_bits > 11) st = ZG_HUF_TABLE;            // MaxBitsTooHigh
        else last_weight = zg_hbit(left);
      }
      if (!st) {
        // ---- code lengths; symbols per length
        uint32_t bl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int sy = (int)lane + 64 * k;
          if (sy == nw) w[k] = last_weight;                   // the last symbol's weight is implied (:309-316)
          bl[k] = w[k] ? max_bits + 1 - w[k] : 0u;
          if (bl[k]) atomicAdd(&L.cls[bl[k]], 1u);
        }
        zg_wave_publish();
        // first entry of every length: lengths in descending order from entry 0 (:339-351)
        uint32_t first[12];
        {
          uint32_t acc = 0;
#pragma unroll
          for (int q = 11; q >= 1; q--) {
            first[q] = acc;
            if ((uint32_t)q <= max_bits) acc += L.cls[q] << (max_bits - (uint32_t)q);
          }
          if (acc != (1u << max_bits)) st = ZG_INTERNAL;      // assert :353-358
        }
        if (!st) {
          // ---- a symbol's rank among the symbols of its length, in symbol order (symbol = lane + 64 k)
          const uint64_t lt = (1ull << lane) - 1ull;
          uint32_t rank[4] = {0, 0, 0, 0};
          for (uint32_t q = 1; q <= max_bits; q++) {          // (uniform trip count)
            uint32_t before = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const uint64_t m = __ballot(bl[k] == q);
              if (bl[k] == q) rank[k] = before + (uint32_t)__popcll(m & lt);
              before += (uint32_t)__popcll(m);
            }
          }
          // ---- the table: symbol (lane, k) owns entries [base, base + n)
          uint32_t base[4], n[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            n[k] = bl[k] ? 1u << (max_bits - bl[k]) : 0u;
            uint32_t f = 0;
#pragma unroll
            for (int q = 1; q <= 11; q++) f = bl[k] == (uint32_t)q ? first[q] : f;   // (selects: a dynamic index would put first[] into scratch memory)
            base[k] = f + rank[k] * n[k];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint32_t e = ZG_HUF_PACK((lane + 64u * (uint32_t)k) & 255u, bl[k]);
            // runs of 64 entries and more: the whole wave writes them, one after the other
            uint64_t m = __ballot(n[k] >= 64u);
            while (m) {
              const int j = __builtin_ctzll(m);
              m &= m - 1;
              const uint32_t bj = __shfl(base[k], j, 64), nj = __shfl(n[k], j, 64), ej = __shfl(e, j, 64);
              for (uint32_t i = lane; i < nj; i += 64) L.table[bj + i] = (uint16_t)ej;
            }
            if (n[k] < 64u) for (uint32_t i = 0; i < n[k]; i++) L.table[base[k] + i] = (uint16_t)e;
          }
          zg_wave_publish();
          uint16_t* out = d.huf_arena + (uint64_t)blk.huf_slot * ZG_HUF_SLOT_U16;
          for (uint32_t i = lane * 8; i < (1u << max_bits); i += 64 * 8) {
            if (i + 8 <= (1u << max_bits)) *(zg_gv4u*)(out + i) = *(const zg_v4u*)(L.table + i);
            else for (uint32_t j = i; j < (1u << max_bits); j++) out[j] = L.table[j];
          }
          if (lane == 0) { d.huf_maxbits[blk.huf_slot] = (uint8_t)max_bits; }
          desc_bytes = (uint32_t)L.res[2];
        }
      }
    }
  }
  if (lane == 0) {
    d.aux[b].huf_desc_bytes = desc_bytes;
    d.tab_status[b] = (uint32_t)st;              // a bad tree description: zg_k_huf leaves the block alone, zg_k_merge reports it
  }
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_ftab: FSE table descriptions of the sequences sections (maybe_update_fse_tables,
// sequence_section_decoder.rs:294-410) and the predefined tables. One WAVE per block: lane 0 parses a description
// (read_probabilities is a bit-serial chain, fse_decoder.rs:264-332), then all 64 lanes build the table
// (build_decoder, fse_decoder.rs:141-262):
//   spreading   cell c of the symbol-ordered list goes to the c-th position of the walk pos -> (pos + step) & mask that
//               is below the low-probability area; the walk is i*step & mask in closed form, so a ballot/popcount
//               prefix over i gives c, and a search in the cumulated probabilities gives the symbol;
//   numbering   the k-th entry (by position) of a symbol gets state prob + k: k is a running count per symbol plus the
//               rank among equal symbols inside the current 64 positions (one ballot per distinct symbol);
//   entries     num_bits / base_line from (prob, k) as in the serial version (zg_fse_build, zg_dev.h), written coalesced.
// ------------------------------------------------------------------------------------------------------------
#define ZG_FT_W 4         // blocks (waves) per workgroup
struct ZgFtabLds {
  int16_t probs[64];
  uint16_t cum[64];
  uint16_t base[64];
  uint8_t symat[512];
  uint8_t xb_ll[36], xb_ml[53];
};
__device__ __forceinline__ int zg_fse_build_wave(ZgFtabLds& L, int np, int al, int kind, uint32_t* out_g, uint32_t lane) {
  const uint32_t N = 1u << al, mask = N - 1, step = (N >> 1) + (N >> 3) + 3;
  const uint64_t lt = (1ull << lane) - 1ull;
  // low-probability symbols take the top positions, in symbol order (fse_decoder.rs:153-164)
  const int p = (int)lane < np ? (int)L.probs[lane] : 0;
  const bool isneg = p == -1;
  const uint64_t nm = __ballot(isneg);
  const uint32_t negcnt = (uint32_t)__popcll(nm);
  if (negcnt > N) return ZG_INTERNAL;
  const uint32_t neg = N - negcnt;
  if (isneg) L.symat[N - 1 - (uint32_t)__popcll(nm & lt)] = (uint8_t)lane;
  // cumulated positive probabilities
  const uint32_t pp = p > 0 ? (uint32_t)p : 0u;
  uint32_t incl = pp;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
  L.cum[lane] = (uint16_t)(incl - pp);
  L.base[lane] = 0;
  if ((uint32_t)__shfl(incl, 63, 64) != neg) return ZG_INTERNAL;     // probabilities do not fill the table
  // spreading (:166-186)
  uint32_t cnt = 0;
  for (uint32_t i0 = 0; i0 < N; i0 += 64) {
    const uint32_t i = i0 + lane;
    const uint32_t ps = (i * step) & mask;
    const bool valid = i < N && ps < neg;
    const uint64_t vm = __ballot(valid);
    if (valid) {
      const uint32_t c = cnt + (uint32_t)__popcll(vm & lt);
      uint32_t lo = 0, hi = (uint32_t)np;                 // largest s with cum[s] <= c (its interval is not empty)
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.cum[mid] <= c) lo = mid; else hi = mid; }
      L.symat[ps] = (uint8_t)lo;
    }
    cnt += (uint32_t)__popcll(vm);
  }
  // numbering + entries (:188-262)
  for (uint32_t p0 = 0; p0 < N; p0 += 64) {
    const uint32_t pos = p0 + lane;
    const bool in = pos < N, low = in && pos < neg;
    const uint32_t sy = in ? (uint32_t)L.symat[pos] : 0xFFFFu;
    uint32_t k = 0;
    uint64_t todo = __ballot(low);
    while (todo) {
      const int l0 = __ffsll((unsigned long long)todo) - 1;
      const uint32_t s0 = (uint32_t)__shfl((int)sy, l0, 64);
      const bool mine = low && sy == s0;
      const uint64_t mm = __ballot(mine);
      if (mine) k = (uint32_t)L.base[s0] + (uint32_t)__popcll(mm & lt);
      if ((int)lane == l0) L.base[s0] = (uint16_t)(L.base[s0] + (uint32_t)__popcll(mm));
      todo &= ~mm;
    }
    if (in) {
      uint32_t e;
      const uint32_t xb = kind == ZG_KIND_LL ? L.xb_ll[sy < 36u ? sy : 0u] : kind == ZG_KIND_ML ? L.xb_ml[sy < 53u ? sy : 0u] : sy;
      if (low) {
        const uint32_t pr = (uint32_t)L.probs[sy];
        const uint32_t hb = zg_hbit(pr);
        const uint32_t sl = ((1u << (hb - 1)) == pr) ? hb - 1 : hb;       // log2 of the number of slices
        const uint32_t slices = 1u << sl, dbl = slices - pr, single = pr - dbl, width = N >> sl;
        uint32_t nb = (uint32_t)al - sl, bl;
        if (k < dbl) { bl = single * width + k * width * 2; nb += 1; }
        else bl = (k - dbl) * width;
        e = ZG_FSE_PACK(bl, nb, sy, xb);
      } else e = ZG_FSE_PACK(0u, (uint32_t)al, sy, xb);
      out_g[pos] = e;
    }
  }
  return ZG_OK;
}

// zg_k_fparse: the table descriptions of a block's sequences section, ONE LANE PER BLOCK (round 6). read_probabilities is a bit-serial chain
// with nothing for a wave to share, and zg_k_ftab — one wave per block, lane 0 reading, 64 lanes building — spent 60 % of its instructions
// on it (15 K wave instructions per block, the kernel is bound by their number). Here 64 blocks are read by one wave at the price of one.
#define ZG_FP_STAGE 336      // bytes of a section's start a lane stages in LDS (three descriptions are < 300 bytes); a multiple of 16
__global__ void __launch_bounds__(64) zg_k_fparse(ZgBatchDev d) {
  // bit reads are dependent loads: each lane brings the start of its section to LDS with 16-byte loads that are all in flight together
  // (one memory round trip instead of three per symbol), and reads the descriptions there; rows are 340 bytes apart (85 dwords: odd, so
  // the lanes' rows start in different banks)
  __shared__ __attribute__((aligned(16))) uint8_t s_hdr[64][ZG_FP_STAGE + 4];
  const uint32_t lane = threadIdx.x, b = blockIdx.x * 64u + lane;
  if (b >= d.nblocks) return;
  const ZgBlock blk = d.blocks[b];
  if (blk.host_status || blk.btype != ZG_BT_COMPRESSED || blk.nseq == 0) return;
  ZgFtabParsed* P = &d.ftab_parsed[b];
  const uint8_t* body = d.src + blk.src_off + blk.seq_off;
  const uint32_t rem_all = blk.src_len - blk.seq_off;
  const uint32_t sh = (uint32_t)((uint64_t)body & 3u);
  const uint32_t lim = rem_all + sh > (uint32_t)ZG_FP_STAGE - 8u ? (uint32_t)ZG_FP_STAGE - 8u - sh : rem_all;   // bytes of the section that are staged
  {
    const uint64_t ga = (uint64_t)body & ~3ull;            // (the submit's bytes have 64 bytes of padding behind them: a row may end up to 11 bytes behind the section)
    uint32_t* row = (uint32_t*)&s_hdr[lane][0];
    const uint32_t nd = (lim + sh + 3u) >> 2;
#pragma unroll 4
    for (uint32_t i = 0; i < nd; i++) row[i] = *(const __attribute__((address_space(1))) uint32_t*)(ga + 4ull * i);
  }
  const int modes[3] = {blk.seq_modes >> 6, (blk.seq_modes >> 4) & 3, (blk.seq_modes >> 2) & 3};
  const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
  int st = ZG_OK, k = 0;
  uint32_t off = 0;
  auto parse = [&](const uint8_t* pb, uint32_t plim) {   // inlined twice: on the staged bytes, and (a section whose descriptions are longer) on the section itself
    uint32_t rem = plim;
    st = ZG_OK; off = 0;
    for (k = 0; k < 3; k++) {
      if (modes[k] == ZG_MODE_FSE) {
        int np = 0, al = 0;
        uint32_t used = 0;
        st = zg_fse_read_probs(pb + off, rem, max_log[k], max_sym[k], P->probs[k], &np, &al, &used);
        if (st) break;
        for (int i = np; i < 64; i++) P->probs[k][i] = 0;
        P->np[k] = (uint8_t)np; P->al[k] = (uint8_t)al;
        off += used; rem -= used;
      } else if (modes[k] == ZG_MODE_RLE) {
        if (rem == 0 || pb[off] > max_sym[k]) { st = ZG_SEQ_RLE_BYTE; break; }
        P->rle[k] = pb[off];
        off += 1; rem -= 1;
      }
    }
  };
  parse(&s_hdr[lane][sh], lim);
  if (st && lim != rem_all) parse(body, rem_all);
  P->nparsed = (uint8_t)k; P->status = (uint32_t)st; P->done = off;
}

__global__ void __launch_bounds__(64 * ZG_FT_W) zg_k_ftab(ZgBatchDev d) {
  __shared__ ZgFtabLds s_l[ZG_FT_W];
  const uint32_t t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const uint32_t b = blockIdx.x * ZG_FT_W + wv;
  ZgFtabLds& L = s_l[wv];
  if (lane < 36) L.xb_ll[lane] = ZG_LL_BITS[lane];
  if (lane < 53) L.xb_ml[lane] = ZG_ML_BITS[lane];
  zg_wave_publish();
  if (b > d.nblocks) return;
  if (b == d.nblocks) {  // predefined tables (acc logs 6/5/6)
    uint32_t* slot = d.fse_arena + (uint64_t)d.nblocks * ZG_FSE_SLOT_U32;
    L.probs[lane] = lane < 36 ? ZG_LL_DEFAULT[lane] : (int16_t)0;
    zg_wave_publish();
    zg_fse_build_wave(L, 36, 6, ZG_KIND_LL, slot + ZG_FSE_LL_OFF, lane);
    zg_wave_publish();
    L.probs[lane] = lane < 29 ? ZG_OF_DEFAULT[lane] : (int16_t)0;
    zg_wave_publish();
    zg_fse_build_wave(L, 29, 5, ZG_KIND_OF, slot + ZG_FSE_OF_OFF, lane);
    zg_wave_publish();
    L.probs[lane] = lane < 53 ? ZG_ML_DEFAULT[lane] : (int16_t)0;
    zg_wave_publish();
    zg_fse_build_wave(L, 53, 6, ZG_KIND_ML, slot + ZG_FSE_ML_OFF, lane);
    if (lane == 0) { uint8_t* lg = d.slot_log + (uint64_t)d.nblocks * 4; lg[0] = 6; lg[1] = 5; lg[2] = 6; lg[3] = 0; }
    return;
  }
  const ZgBlock blk = d.blocks[b];
  if (blk.host_status || blk.btype != ZG_BT_COMPRESSED) return;
  uint32_t seq_bits_off = blk.seq_off;
  uint8_t logs[3] = {0, 0, 0};
  int st = ZG_OK;
  if (blk.nseq > 0) {
    uint32_t* slot = d.fse_arena + (uint64_t)b * ZG_FSE_SLOT_U32;
    // order LL, OF, ML (sequence_section_decoder.rs:305,341,376). The descriptions were read by zg_k_fparse (one lane per block); a table
    // is built here, by the whole wave, when everything in front of it was fine: the first failing step in section order decides
    const ZgFtabParsed* P = &d.ftab_parsed[b];
    const int kinds[3] = {ZG_KIND_LL, ZG_KIND_OF, ZG_KIND_ML};
    const int modes[3] = {blk.seq_modes >> 6, (blk.seq_modes >> 4) & 3, (blk.seq_modes >> 2) & 3};
    const uint32_t offs[3] = {ZG_FSE_LL_OFF, ZG_FSE_OF_OFF, ZG_FSE_ML_OFF};
    const uint32_t nparsed = P->nparsed;
    for (int k = 0; k < 3 && !st; k++) {
      if ((uint32_t)k >= nparsed) { st = (int)P->status; break; }
      if (modes[k] == ZG_MODE_FSE) {
        const int np = P->np[k], al = P->al[k];
        zg_wave_publish();                                   // (the previous table's build still reads L.probs)
        L.probs[lane] = P->probs[k][lane];
        zg_wave_publish();
        st = zg_fse_build_wave(L, np, al, kinds[k], slot + offs[k], lane);
        if (!st) logs[k] = (uint8_t)al;
      } else if (modes[k] == ZG_MODE_RLE) {
        if (lane == 0) slot[offs[k]] = zg_fse_pack(kinds[k], 0, 0, P->rle[k]);
        logs[k] = 0;
      }
    }
    if (!st && nparsed < 3u) st = (int)P->status;
    seq_bits_off = blk.seq_off + P->done;
    if (lane == 0) { uint8_t* lg = d.slot_log + (uint64_t)b * 4; lg[0] = logs[0]; lg[1] = logs[1]; lg[2] = logs[2]; lg[3] = 0; }
  }
  if (lane == 0) {
    d.aux[b].seq_bits_off = seq_bits_off;
    d.aux[b].log[0] = logs[0]; d.aux[b].log[1] = logs[1]; d.aux[b].log[2] = logs[2];
    zg_set_status(d.status, b, st);
  }
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_huf: Huffman literal streams, one wave per stream, self-synchronising chunks — the body is zg_huf.h (zx_* primitives:
// the same source runs under the CPU emulator)
// ------------------------------------------------------------------------------------------------------------
#include "zg_huf.h"
#define ZG_HUF_T (64 * ZG_HUF_GROUP)
__global__ void __launch_bounds__(ZG_HUF_T) zg_k_huf(ZgBatchDev d) {
  __shared__ ZgHufLds<ZG_HUF_GROUP> s_l;
  zg_huf_group<ZG_HUF_GROUP>(d, blockIdx.x, s_l);
}

// zg_k_huf_uneven: four streams whose symbols add up to the section's size but are not split the way the format says
// ((regen + 3) / 4 each, the rest in the fourth). ruzstd decodes the streams one after the other into one buffer and
// compares only the total (literals_section_decoder.rs:94-155), so such a section is valid for it; zg_k_huf wrote every
// stream where the format's split puts it and reported a count mismatch. No encoder emits this: the repair is a plain
// serial decode, one lane per stream, with the counts zg_k_huf left behind. One wave per block; a block that is not
// affected costs its wave one load.
__global__ void __launch_bounds__(64) zg_k_huf_uneven(ZgBatchDev d) {
  // (one wave and one table per workgroup: 4 KB of LDS, which fits beside two zg_k_seq workgroups; with four tables per workgroup
  //  this kernel — one load per block on conforming input — waited for the sequences chain to leave the CUs)
  __shared__ uint16_t s_tab[1][ZG_HUF_SLOT_U16];
  const uint32_t wv = 0, lane = threadIdx.x & 63u, b = blockIdx.x;
  if (b >= d.nblocks || d.totals[2]) return;
  if (d.lit_status[b] != (((255u - 8u) << 8) | (uint32_t)ZG_LIT_COUNT_MISMATCH)) return;   // every stream ended on its last bit, some count differs
  const ZgBlock blk = d.blocks[b];
  if (blk.nstreams != 4 || blk.lit_type < ZG_LT_COMPRESSED || blk.huf_slot < 0) return;
  const uint32_t regen = blk.regen_size;
  const uint32_t c0 = d.lit_counts[4u * b], c1 = d.lit_counts[4u * b + 1], c2 = d.lit_counts[4u * b + 2], c3 = d.lit_counts[4u * b + 3];
  if ((uint64_t)c0 + c1 + c2 + c3 != regen) return;            // DecodedLiteralCountMismatch stands
  // (where zg_k_huf put them: the arena, or the block's place in the output — zg_huf.h)
  bool direct = (d.flags & ZG_FLAG_LIT_DIRECT) != 0u && blk.nseq == 0u;
  if (direct && !d.pos[b].active) {                            // (the same rule as zg_huf.h: the block its frame stops at because of its sequences header)
    if (!blk.seq_host_status || b != d.frames[blk.frame].first_block + d.frame_out[blk.frame].good_blocks) return;
    direct = false;
  }
  uint8_t* lits = direct ? d.dst + d.frame_out[blk.frame].out_base + d.pos[b].out_base : d.lit_arena + blk.lit_base;
  const unsigned max_bits = d.huf_maxbits[blk.huf_slot];
  if (max_bits == 0 || max_bits > 11) return;
  const uint16_t* g = d.huf_arena + (uint64_t)blk.huf_slot * ZG_HUF_SLOT_U16;
  for (uint32_t i = lane; i < (1u << max_bits); i += 64) s_tab[wv][i] = g[i];
  zg_wave_publish();
  if (lane < 4) {
    const uint32_t k = lane;
    const uint32_t desc = blk.lit_type == ZG_LT_COMPRESSED ? d.aux[b].huf_desc_bytes : 0;
    const uint8_t* pay = d.src + blk.src_off + blk.lit_off + desc;
    const uint32_t total = blk.lit_comp_size - desc;
    const uint32_t j1 = zg_ld16(pay), j2 = j1 + zg_ld16(pay + 2), j3 = j2 + zg_ld16(pay + 4);
    const uint32_t start = k == 0 ? 0 : k == 1 ? j1 : k == 2 ? j2 : j3;
    const uint32_t end = k == 0 ? j1 : k == 1 ? j2 : k == 2 ? j3 : total - 6;
    const uint8_t* sp = pay + 6 + start;
    const int32_t slen = (int32_t)(end - start);
    const uint32_t cnt = k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3;
    uint8_t* dst = lits + (k == 0 ? 0u : k == 1 ? c0 : k == 2 ? c0 + c1 : c0 + c1 + c2);
    const uint32_t pmask = (1u << max_bits) - 1u;
    int32_t P = (slen - 1) * 8 + (int32_t)zg_hbit(sp[slen - 1]) - 1;   // bits of the stream (zg_k_huf checked the final-bit marker)
    for (uint32_t n = 0; n < cnt && P > 0; n++) {
      const int32_t q = P - (int32_t)max_bits;                 // the code's bits: [q, P); below the stream's first bit: zeros
      uint32_t v = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) { const int32_t at = (q >> 3) + i; if (at >= 0 && at < slen) v |= (uint32_t)sp[at] << (8 * i); }
      const uint32_t e = s_tab[wv][(v >> (q & 7)) & pmask];
      dst[n] = (uint8_t)e;
      const int32_t nb = (int32_t)(e >> 8);
      P -= nb > 1 ? nb : 1;
    }
  }
  zg_wave_publish();
  if (lane == 0) d.lit_status[b] = 0;
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_seq: sequence sections (decode_sequences, sequence_section_decoder.rs:14-221). The bitstream of a block is
// one serial chain, so the kernel is bound by the length of one step of that chain: a wave issues one instruction
// every four cycles whatever the number of active lanes, and a block has ~13 K steps. Everything that is not the
// chain itself is therefore moved out of the step: zg_k_seq only follows the three FSE state chains and records the
// three STATES of every sequence (8 bytes); symbols, extra bits, values and positions are rebuilt from them in
// parallel by zg_k_seqpost. Four lanes serve one block (one per chain plus one spare), ZG_SEQ_G blocks per wave,
// tables staged in LDS as {u16 baseline << 4 | state bits, u8 all bits the state's symbol takes}.
//
// The chain never touches global memory. A workgroup is two waves: the DECODER wave works in LDS only (tables,
// a 512-byte ring of each block's bitstream, the recorded states), ZG_SEQ_CH sequences per block and phase; the MOVER
// wave, one barrier behind, flushes the states of the phase just finished with 8-byte stores and extends every ring
// downwards with 16-byte loads that it lands one phase later, so neither wave ever waits for a load it just issued and
// the decoder only meets the mover at the barrier that ends a phase.
//
// A decode phase comes in two forms. While more than ZG_SEQ_CH sequences are left, none of them is the block's last
// one (which takes no state bits, :203) and a stream that runs out of bits is corrupt, so the FAST form checks nothing
// per step: the position only decreases, one sign test after the phase is enough, and a chain that ran past the
// start of a corrupt stream reads ring and table addresses that stay in range by construction. The block's last
// <= ZG_SEQ_CH sequences go through the CAREFUL form, which freezes a finished block and applies :203 and :209-211
// per step.
// ------------------------------------------------------------------------------------------------------------
#define ZG_SEQ_CH 12                                  // sequences per block between two mover phases
#define ZG_SEQ_CMAX 144                               // >= bytes ZG_SEQ_CH sequences can consume (12 x 89 bits)
// Bytes of bitstream kept resident below the position a request is based on. A piece requested after phase k (position
// P_k) lands during phase k+2's decode and is first read in phase k+3, which reaches down to P_k - 3 CMAX - 16; and what it
// overwrites, 512 bytes further up, must be above everything phase k+2 reads (<= P_k + 8): MARGIN + 16 + 8 < 512.
#define ZG_SEQ_MARGIN (3 * ZG_SEQ_CMAX + 32)
#define ZG_SEQ_RING 512                               // per-block ring, indexed by the low bits of the global address
#define ZG_SEQ_PIECES 10                              // 16-byte pieces one mover phase can add per block (>= CMAX/16 + 1)
#define ZG_SEQ_PREG ((ZG_SEQ_PIECES + 3) / 4)         // piece requests per lane and phase: a block's four mover lanes share them
static_assert(ZG_SEQ_MARGIN + 16 + 8 < ZG_SEQ_RING, "ring too small for the margin");
#define ZG_SEQ_PRO ((ZG_SEQ_MARGIN + 16 + 8 + 15 + 15) / 16 + 1)   // pieces of the prologue fill

// Ring storage: 16 bytes that mirror the ring's END, then the ring. The four dwords a step looks at are the ones
// [a - 12, a + 4) around the dword a that holds the current position, so with the mirror in FRONT their address is
// storage + 4 + a without a wrap.
#define ZG_SEQ_RSTORE (ZG_SEQ_RING + 16)
__device__ __forceinline__ void zg_ring_put(uint8_t* store, uint64_t addr, const zg_v4u& v) {
  const uint32_t ro = (uint32_t)(addr & (ZG_SEQ_RING - 1));
  *(zg_v4u*)(store + 16 + ro) = v;
  if (ro == ZG_SEQ_RING - 16) *(zg_v4u*)store = v;
}
// bits [pr, pr+n) of the ring (pr = stream bit position + the ring phase of the stream's first byte), n <= 31
__device__ __forceinline__ uint32_t zg_ring_bits(const uint8_t* store, int32_t pr, uint32_t n) {
  const uint32_t a = (((uint32_t)pr >> 5) & (ZG_SEQ_RING / 4 - 1)) * 4u;
  const uint32_t d0 = *(const uint32_t*)(store + 16 + a), d1 = *(const uint32_t*)(store + 16 + ((a + 4) & (ZG_SEQ_RING - 1)));
  return __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(d1, d0, (uint32_t)pr & 31u), 0u, n);
}

struct ZgSeqChain {        // one lane's view of its block's chain
  int32_t pr;              // bit position + ring phase
  uint32_t e, s;           // the lane's table entry: baseline << 4 | state bits; all bits its symbol takes
  uint32_t st;             // the lane's state (what gets recorded)
  uint32_t w0, w1, w2, w3, wlo;   // the window below the position and the bit address of its first dword plus 2^32 - 96... see zg_seq_window
};
// request the four dwords around the position and remember where they start
__device__ __forceinline__ void zg_seq_window(ZgSeqChain& c, const uint8_t* store4) {
  const uint32_t a = __builtin_amdgcn_ubfe((uint32_t)c.pr, 5u, 7u);
  const uint32_t* w = (const uint32_t*)(store4 + a * 4u);
  c.w0 = w[0]; c.w1 = w[1]; c.w2 = w[2]; c.w3 = w[3];
  c.wlo = ((uint32_t)c.pr & ~31u) - 96u;
}
// One step of the chain. FAST: no checks (see above). Otherwise `act`, `left`, `cnt` are maintained and a finished or
// failed block keeps its state.
// PK (round 5): ONE 16-bit entry per state instead of a u16 and a u8 — [9:0] (baseline << 1) | (1 << state bits): a baseline is a multiple of
// 2^(state bits), so the lowest set bit says how many there are; [15:10] all bits the symbol takes. Four instructions more in the step and
// a third less LDS per block: three workgroups per CU instead of two. The kernel lasts as long as ONE chain when all blocks of the submit are
// resident at once (the 7630 blocks of the 1e9-byte frame: the unpacked form), and as long as blocks x chain / resident chains when they
// are not (65536 single-block frames, 128 x 64 MiB frames: eight rounds -> 5.3: the packed form). zg_launch_seq picks.
// (measurement variant, tools/dev/mkvariant.sh ILV -DZG_SEQ_ILV: the tables of a workgroup's 16 blocks interleaved at 16-byte granularity, so
//  that block g's look-ups always land in the four LDS banks 4g .. 4g + 3 and never collide with another block's — VERDICT r5 item 2a. A
//  lane's `tab` / `xtab` are then BYTE bases (block granule + the table's offset), the address of entry i is computed per look-up.)
#ifdef ZG_SEQ_ILV
#define ZG_SEQ_ILVA(o) ((((o) >> 4) << 8) | ((o) & 15u))
#define ZG_SEQ_LD16(tab, i) (*(const uint16_t*)((const uint8_t*)(tab) + ZG_SEQ_ILVA(to2 + 2u * (i))))
#define ZG_SEQ_LD8(xtab, i) (*((const uint8_t*)(xtab) + ZG_SEQ_ILVA(to1 + (i))))
#else
#define ZG_SEQ_LD16(tab, i) ((tab)[i])
#define ZG_SEQ_LD8(xtab, i) ((xtab)[i])
#endif
#define ZG_SEQ_PK(bl, nb, all) ((uint16_t)((((uint32_t)(bl) << 1) | (1u << (nb))) | ((uint32_t)(all) << 10)))
template <bool FAST, bool PK>
__device__ __forceinline__ void zg_seq_step(ZgSeqChain& c, const uint16_t* tab, const uint8_t* xtab, const uint8_t* store4, uint16_t* rec,
                                            int32_t rbits, bool& act, uint32_t& left, uint32_t& cnt, const uint32_t to2 = 0, const uint32_t to1 = 0) {
  const uint32_t nb0 = PK ? (uint32_t)__builtin_ctz(c.e) : c.e & 15u;
  if (PK) c.s = c.e >> 10;
  uint32_t nb = nb0, pk = c.s | (nb0 << 8);          // pk: [7:0] all bits this lane's symbol takes, [15:8] its state bits
  if (!FAST) { const bool last = left == 1u; nb = last ? 0u : nb0; pk = last ? c.s - nb0 : pk; }   // no state update after the last sequence (:203)
  // quad prefix sums, lanes in stream order from the low end: OF state, ML state, LL state (the extra bits above them are skipped as one count)
  const uint32_t i1 = pk + (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0x93, 0xF, 0xF, true);     // quad_perm [3,0,1,2]
  const uint32_t incl = i1 + (uint32_t)__builtin_amdgcn_mov_dpp((int)pk, 0x4F, 0xF, 0xF, true);   // quad_perm [3,3,0,1]
  const uint32_t tot = (uint32_t)__builtin_amdgcn_mov_dpp((int)incl, 0xAA, 0xF, 0xF, true) & 255u; // quad_perm [2,2,2,2]
  const int32_t q = c.pr - (int32_t)tot;              // where the sequence ends
  // this lane's nb state bits start (incl - pk) >> 8 bits above q; q >= position - 89, so they are inside the window
  const uint32_t rel = (uint32_t)q - c.wlo + ((incl - pk) >> 8);
  const bool up = rel >= 64u, odd = (rel & 32u) != 0u;
  const uint32_t a0 = up ? c.w2 : c.w0, a1 = up ? c.w3 : c.w1, a2 = up ? c.w3 : c.w2;
  const uint32_t d0 = odd ? a1 : a0, d1 = odd ? a2 : a1;
  const uint32_t bits = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(d1, d0, rel), 0u, nb);
  const uint32_t st2 = PK ? ((c.e & 0x3FFu) >> 1) + bits - ((1u << nb0) >> 1) : (c.e >> 4) + bits;
  if (FAST) {
    *rec = (uint16_t)c.st;
    c.pr = q; c.st = st2;
    c.e = ZG_SEQ_LD16(tab, st2); if (!PK) c.s = ZG_SEQ_LD8(xtab, st2);
    zg_seq_window(c, store4);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    const bool ok = act && q >= rbits;                // :209-211
    rec[cnt * 4u] = (uint16_t)c.st;                   // (not recorded unless cnt advances)
    cnt += ok ? 1u : 0u;
    const bool was = act;
    left -= ok ? 1u : 0u;                             // stops at the sequence that ran out of bits
    act = ok && left != 0u;
    ZgSeqChain n = c;
    n.pr = q; n.st = st2; n.e = ZG_SEQ_LD16(tab, st2); if (!PK) n.s = ZG_SEQ_LD8(xtab, st2);
    zg_seq_window(n, store4);
    __builtin_amdgcn_sched_barrier(0);
    if (was) { c.pr = n.pr; c.wlo = n.wlo; c.w0 = n.w0; c.w1 = n.w1; c.w2 = n.w2; c.w3 = n.w3; }
    if (act) { c.st = n.st; c.e = n.e; c.s = n.s; }
  }
}

template <bool PK>
__global__ void __launch_bounds__(128) zg_k_seq(ZgBatchDev d) {
#ifdef ZG_SEQ_ILV
  __shared__ __attribute__((aligned(16))) uint8_t s_tabi[((ZG_FSE_SLOT_U32 + 2) * 2 + 15) / 16 * 256];
  __shared__ __attribute__((aligned(16))) uint8_t s_xbi[PK ? 16 : (ZG_FSE_SLOT_U32 + 4 + 15) / 16 * 256];
#define ZG_SEQ_ST16(g, i, v) (*(uint16_t*)(s_tabi + (ZG_SEQ_ILVA(2u * (i)) | ((g) << 4))) = (v))
#define ZG_SEQ_ST8(g, i, v) (s_xbi[ZG_SEQ_ILVA((uint32_t)(i)) | ((g) << 4)] = (v))
#else
  __shared__ uint16_t s_tab[ZG_SEQ_G][ZG_FSE_SLOT_U32 + 2];   // + the one-entry dummy table of the spare lane
  __shared__ uint8_t s_xb[PK ? 1 : ZG_SEQ_G][PK ? 4 : ZG_FSE_SLOT_U32 + 4];
#define ZG_SEQ_ST16(g, i, v) (s_tab[g][i] = (v))
#define ZG_SEQ_ST8(g, i, v) (s_xb[g][i] = (v))
#endif
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[ZG_SEQ_G][ZG_SEQ_RSTORE];
  __shared__ __attribute__((aligned(16))) uint2 s_out[2][ZG_SEQ_G][ZG_SEQ_CH];   // records, 4 x u16: states {OF, ML, LL, 0}; one buffer per phase parity
  __shared__ int32_t s_pos[2][ZG_SEQ_G];            // decoder -> mover, per phase parity: the block's position after the phase,
  __shared__ uint32_t s_cnt[2][ZG_SEQ_G];           //   records of the phase | still active << 8
  __shared__ uint32_t s_more[2];                    //   any block of the workgroup still active
  __shared__ uint64_t s_fetch_hi[ZG_SEQ_G], s_fetch_lo[ZG_SEQ_G];   // prologue: the part of each block's stream to load, [lo, hi)
  __shared__ uint8_t s_log[ZG_SEQ_G][4];
  __shared__ int s_ok[ZG_SEQ_G];
  const uint32_t base = blockIdx.x * ZG_SEQ_G, t = threadIdx.x;
  const bool mover = t >= 64u;
  for (uint32_t g = 0; g < ZG_SEQ_G; g++) {
    uint32_t idx = base + g;
    if (idx >= d.nseq_blocks) break;
    uint32_t b = d.seq_blocks[idx];
    const ZgBlock* blk = &d.blocks[b];
    int32_t sl[3] = {blk->ll_slot, blk->of_slot, blk->ml_slot};
    const uint32_t offs[3] = {ZG_FSE_LL_OFF, ZG_FSE_OF_OFF, ZG_FSE_ML_OFF};
    bool ok = d.status[b] == 0;
    for (int k = 0; k < 3; k++) {
      if (sl[k] < 0) { ok = false; continue; }
      if ((uint32_t)sl[k] < d.nblocks && d.status[sl[k]] != 0) { ok = false; continue; }  // defining block failed
      unsigned lg = d.slot_log[(uint64_t)sl[k] * 4 + k];
      if (lg > 9) { ok = false; continue; }
      const uint32_t* g_t = d.fse_arena + (uint64_t)sl[k] * ZG_FSE_SLOT_U32 + offs[k];
      {  // <= 512 entries: every lane loads its (up to 4) entries first, then stores them
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = t + 128 * j; v[j] = i < (1u << lg) ? g_t[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t i = t + 128 * j;
          if (i < (1u << lg)) {
            const uint32_t nb = ZG_FSE_NB(v[j]);
            const uint32_t all = nb + (k == 1 ? ZG_FSE_SYM(v[j]) : v[j] >> 26);                // OF: the code is the number of extra bits
            if (PK) ZG_SEQ_ST16(g, offs[k] + i, ZG_SEQ_PK(ZG_FSE_BL(v[j]), nb, all));
            else { ZG_SEQ_ST16(g, offs[k] + i, (uint16_t)((ZG_FSE_BL(v[j]) << 4) | nb)); ZG_SEQ_ST8(g, offs[k] + i, (uint8_t)all); }
          }
        }
      }
      if (t == 0) s_log[g][k] = (uint8_t)lg;
    }
    if (t == 0) { s_ok[g] = ok ? 1 : 0; ZG_SEQ_ST16(g, ZG_FSE_SLOT_U32, PK ? ZG_SEQ_PK(0, 0, 0) : 0); if (!PK) ZG_SEQ_ST8(g, ZG_FSE_SLOT_U32, 0); }
  }
  __syncthreads();
  // ---- per-lane setup: four lanes per block in either wave. Decoder: lane role 0 follows the OF chain, 1 the ML chain,
  // 2 the LL chain; role 3 is the spare (its "table" is the one-entry dummy: no bits). All four hold the block's scalars.
  const uint32_t g = (t & 63u) >> 2, role = t & 3u;
  const bool owner = role == 3u;
  bool act = base + g < d.nseq_blocks;
  bool have = false;
  uint32_t b = 0, nseq = 0, left = 0;
  int32_t rbits = 0;
  uint64_t bsA = 0, floorA = 0, lo = 0, dstp = 0;
  const uint32_t toff = role == 0u ? ZG_FSE_OF_OFF : role == 1u ? ZG_FSE_ML_OFF : role == 2u ? ZG_FSE_LL_OFF : ZG_FSE_SLOT_U32;
#ifdef ZG_SEQ_ILV
  const uint16_t* tab = (const uint16_t*)(s_tabi + (g << 4));
  const uint8_t* xtab = PK ? s_xbi : s_xbi + (g << 4);
  const uint32_t to2 = toff * 2u, to1 = toff;
#else
  const uint16_t* tab = &s_tab[g][toff];
  const uint8_t* xtab = PK ? &s_xb[0][0] : &s_xb[g][toff];
  const uint32_t to2 = 0, to1 = 0;
#endif
  uint8_t* const store = s_ring[g];
  const uint8_t* const store4 = store + 4;
  int32_t P = 0;
  int status = ZG_OK;
  if (act) {
    b = d.seq_blocks[base + g];
    const ZgBlock blk = d.blocks[b];
    if (!s_ok[g]) {
      // FSEDecoder::init_state on a table that was never set (fse_decoder.rs:33-35), or an upstream failure (whose status stands: the first
      // one set is kept). The reference looks at the bitstream's padding BEFORE it initialises the states (sequence_section_decoder.rs:29-40
      // in front of :52-54 / :130-132): a block with both defects answers ExtraPadding (tools/dev/soak.py seed 204)
      const uint32_t bo = d.aux[b].seq_bits_off;
      const bool nopad = bo <= blk.src_len && (blk.src_len == bo || d.src[blk.src_off + blk.src_len - 1u] == 0);
      if (owner && !mover) zg_set_status(d.status, b, nopad ? ZG_SEQ_EXTRA_PADDING : ZG_FSE_UNINIT);
      act = false;
    } else {
      const uint32_t bits_off = d.aux[b].seq_bits_off;
      if (bits_off > blk.src_len) { if (owner && !mover) zg_set_status(d.status, b, ZG_INTERNAL); act = false; }
      else {
        const uint8_t* bs = d.src + blk.src_off + bits_off;
        const uint32_t bs_len = blk.src_len - bits_off;
        nseq = blk.nseq;
        const uint32_t lastb = bs_len ? bs[bs_len - 1] : 0;
        if (bs_len == 0 || lastb == 0) { if (owner && !mover) zg_set_status(d.status, b, ZG_SEQ_EXTRA_PADDING); act = false; }  // :29-40
        else {
          P = (int32_t)(bs_len - 1) * 8 + (int32_t)(zg_hbit(lastb) - 1);
          bsA = (uint64_t)bs;
          rbits = (int32_t)((uint32_t)(bsA & (ZG_SEQ_RING - 1)) * 8u);
          floorA = (bsA & ~15ull) - 16;                 // the engine keeps 64 bytes of padding in front of the buffer
          dstp = (uint64_t)(d.raw_arena + blk.seq_base);
          have = true;
        }
      }
    }
  }
  // ---- prologue: fill the rings from the top of each stream downwards
  {
    uint64_t top = 0, want = 0;
    if (act) {
      top = (bsA + (uint64_t)(P >> 3) + 8 + 15) & ~15ull;       // a read at bit P touches bytes up to P/8 + 7
      const uint64_t p0 = bsA + (uint64_t)(P >> 3);
      want = p0 > ZG_SEQ_MARGIN + 16 ? (p0 - ZG_SEQ_MARGIN - 16) & ~15ull : 0;
      if (want < floorA) want = floorA;
      lo = want;
    }
    if (owner && !mover) { s_fetch_hi[g] = act ? top : 0; s_fetch_lo[g] = act ? want : 0; }
    __syncthreads();
    for (uint32_t j = t; j < ZG_SEQ_G * ZG_SEQ_PRO; j += 128) {
      const uint32_t gg = j / ZG_SEQ_PRO, k = j % ZG_SEQ_PRO;
      const uint64_t hi = s_fetch_hi[gg], addr = hi - 16ull * (k + 1);
      if (hi && addr >= s_fetch_lo[gg] && addr < hi) { const zg_v4u v = *(const zg_gv4u*)addr; zg_ring_put(s_ring[gg], addr, v); }
    }
    __syncthreads();
  }
  ZgSeqChain c = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (act) {  // initial states, order LL, OF, ML (:164-166); a negative position is reported after the first sequence
    const uint32_t ll_log = s_log[g][0], of_log = s_log[g][1], ml_log = s_log[g][2];
    const uint32_t lg = role == 0u ? of_log : role == 1u ? ml_log : role == 2u ? ll_log : 0u;
    const int32_t q = P - (int32_t)(role == 2u ? ll_log : role == 0u ? ll_log + of_log : ll_log + of_log + ml_log);
    c.st = (q >= 0 && role != 3u) ? zg_ring_bits(store, q + rbits, lg) : 0u;
    c.e = ZG_SEQ_LD16(tab, c.st); c.s = PK ? 0u : ZG_SEQ_LD8(xtab, c.st);
    P -= (int32_t)(ll_log + of_log + ml_log);
    if (owner && !mover) d.seq_out[b].pad = (uint32_t)P;   // where the first sequence starts: zg_k_seqpost rebuilds the positions from the bit counts
    left = nseq;
  }
  c.pr = P + rbits;
#ifdef ZG_PROFILE_SEQ
  unsigned long long tcs[3] = {0, 0, 0}, tl_ = clock64();
#define ZG_QTICK(i) { const unsigned long long n_ = clock64(); tcs[i] += n_ - tl_; tl_ = n_; }
#else
#define ZG_QTICK(i)
#endif
  if (!mover) {
    if (act) zg_seq_window(c, store4);   // every step requests the next one's window; the mover only writes far below it, so it stays valid across phases
    // ---- DECODER wave: LDS only, and one LDS round trip per sequence: the lane's next table entry, its bit count and the
    // 128 bits of stream below the next position are requested together; everything between is 32-bit ALU, and the
    // three chains of a block run in three lanes that exchange their bit counts through DPP quad permutes
    // (FSEDecoder::update_state, fse_decoder.rs:40-48, order LL, ML, OF :204-206).
    for (uint32_t par = 0;; par ^= 1u) {
      uint16_t* const out_base = (uint16_t*)&s_out[par][g][0] + role;   // record {OF, ML, LL state, spare}: one u16 per lane
      uint32_t cnt = 0;
      if (act && left > ZG_SEQ_CH) {
#pragma unroll
        for (int k = 0; k < ZG_SEQ_CH; k++) zg_seq_step<true, PK>(c, tab, xtab, store4, out_base + 4 * k, rbits, act, left, cnt, to2, to1);
        cnt = ZG_SEQ_CH; left -= ZG_SEQ_CH;
        if (c.pr < rbits) act = false;                             // ran out of bits with sequences left (:209-211)
      } else if (act) {
#pragma unroll
        for (int k = 0; k < ZG_SEQ_CH; k++) zg_seq_step<false, PK>(c, tab, xtab, store4, out_base, rbits, act, left, cnt, to2, to1);
      }
      P = c.pr - rbits;
      const bool more = __any(act);
      if (owner) { s_pos[par][g] = P; s_cnt[par][g] = cnt | (act ? 256u : 0u); }
      if (t == 0) s_more[par] = more ? 1u : 0u;
      ZG_QTICK(0)
      __syncthreads();
      ZG_QTICK(1)
      if (!more) break;
    }
    if (have && owner) {
      if (left != 0) status = ZG_SEQ_NOT_ENOUGH_BYTES;        // the loop stopped at a sequence that ran out of bits (:209-211)
      else if (P > 0) status = ZG_SEQ_EXTRA_BITS;             // :214-220
      zg_set_status(d.status, b, status);
    }
  } else {
    // ---- MOVER wave: every quad serves its own block (ring bounds and output pointer are held by all four lanes)
    zg_v4u piece[ZG_SEQ_PREG];
    uint64_t piece_addr[ZG_SEQ_PREG];
    bool piece_ok[ZG_SEQ_PREG];
#pragma unroll
    for (int pi = 0; pi < ZG_SEQ_PREG; pi++) { piece[pi] = zg_v4u{0, 0, 0, 0}; piece_addr[pi] = 0; piece_ok[pi] = false; }
    for (uint32_t par = 0;; par ^= 1u) {
      __syncthreads();                                             // the phase's records and positions are in LDS
      const uint32_t cw = s_cnt[par][g], cnt = cw & 255u;
      const bool live = have && (cw >> 8) != 0u;
      const int32_t pos = s_pos[par][g];
      const bool more = s_more[par] != 0u;
      // (1) the records of the phase: the quad's lanes take every fourth one (8 bytes each)
      zg_v2u rec[(ZG_SEQ_CH + 3) / 4];
#pragma unroll
      for (int i = 0; i < (ZG_SEQ_CH + 3) / 4; i++) rec[i] = *(const zg_v2u*)&s_out[par][g][role + 4u * (uint32_t)i];
      // (2) land the pieces requested one phase ago (they are older than every store below)
#pragma unroll
      for (int pi = 0; pi < ZG_SEQ_PREG; pi++) if (piece_ok[pi]) zg_ring_put(store, piece_addr[pi], piece[pi]);
      // (3) request the next pieces: [want, hi) just below what the ring holds or has been promised
      {
        uint64_t hi = 0, want = 0;
        if (live) {
          const uint64_t p0 = bsA + (uint64_t)(pos >> 3);
          want = p0 > ZG_SEQ_MARGIN ? (p0 - ZG_SEQ_MARGIN) & ~15ull : 0;
          if (want < floorA) want = floorA;
          if (want < lo) { hi = lo; lo = want; }
        }
#pragma unroll
        for (int pi = 0; pi < ZG_SEQ_PREG; pi++) {
          const uint64_t addr = hi - 16ull * (role + 4u * (uint32_t)pi + 1u);
          piece_ok[pi] = hi && addr >= want && addr < hi;
          if (piece_ok[pi]) { piece[pi] = *(const zg_gv4u*)addr; piece_addr[pi] = addr; }
        }
      }
      // (4) flush: the three states of a sequence fit 26 bits — 4 bytes per record in memory (zg_k_seqpost is bound by what it reads)
#pragma unroll
      for (int i = 0; i < (ZG_SEQ_CH + 3) / 4; i++) {
        const uint32_t k = role + 4u * (uint32_t)i;
        if (k < cnt) ((__attribute__((address_space(1))) uint32_t*)dstp)[k] = ZG_RAW_PACK(rec[i].x & 0xFFu, (rec[i].x >> 16) & 0x1FFu, rec[i].y & 0x1FFu);
      }
      dstp += (uint64_t)cnt * 4u;
      if (!more) break;
    }
  }
#ifdef ZG_PROFILE_SEQ
  if (t == 0 && d.dbg) { atomicAdd(&d.dbg[16], tcs[0]); atomicAdd(&d.dbg[17], tcs[1]); atomicAdd(&d.dbg[18], 1ull); }
#endif
#undef ZG_QTICK
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_seqpost: everything about a block's sequences that is not the serial state chain, done in parallel over the
// states zg_k_seq recorded: symbols and bit counts (the block's three tables, staged in LDS), extra bits read from the
// bitstream (get_bits_triple, bit_reader_reverse.rs:151-162), values
// (lookup_ll_code / lookup_ml_code, sequence_section_decoder.rs:227-284), the output position of every sequence
// (prefix sums, sequence_execution.rs:6-39) and the offset history (do_offset_history :59-118) as a scan of symbolic
// maps: the history after sequence i is the composition of the maps of sequences 0..i, and the actual offset of
// sequence i is slot 0 of it. One workgroup per block, 256 sequences per pass.
// ------------------------------------------------------------------------------------------------------------
#define ZG_SP_T 256
#define ZG_SP_S 8           // consecutive sequences per thread: the history inside them is stepped directly, only thread totals are scanned
struct ZgHistMap { uint32_t s[3]; };
__device__ __forceinline__ ZgHistMap zg_map_identity() { return {{1u << 30, 2u << 30, 3u << 30}}; }
// v (a slot value relative to map A's output) expressed relative to A's input
__device__ __forceinline__ uint32_t zg_map_apply(const ZgHistMap& A, uint32_t v) {
  // (branches on purpose: almost every slot is a constant, and a wave whose lanes all hold constants skips the rest;
  //  a select-only version was measured slower in zg_k_seqpost and zg_k_scan)
  const uint32_t t = ZG_SYM_TAG(v);
  if (!t) return v;
  const uint32_t a = t == 1 ? A.s[0] : t == 2 ? A.s[1] : A.s[2], k = ZG_SYM_K(v);
  if (ZG_SYM_TAG(a)) {
    uint32_t kk = ZG_SYM_K(a) + k;
    if (kk > 0x3FFFFFFFu) kk = 0x3FFFFFFFu;
    return (a & 0xC0000000u) | kk;
  }
  return a > k ? a - k : 0;
}
// apply A first, then B
__device__ __forceinline__ ZgHistMap zg_map_compose(const ZgHistMap& A, const ZgHistMap& B) {
  ZgHistMap r;
  r.s[0] = zg_map_apply(A, B.s[0]); r.s[1] = zg_map_apply(A, B.s[1]); r.s[2] = zg_map_apply(A, B.s[2]);
  return r;
}

__global__ void __launch_bounds__(ZG_SP_T, 4) zg_k_seqpost(ZgBatchDev d) {
  __shared__ ZgHistMap s_wm[ZG_SP_T / 64];
  __shared__ uint32_t s_wl[ZG_SP_T / 64], s_wo[ZG_SP_T / 64], s_wx[ZG_SP_T / 64];
  __shared__ uint32_t s_err;
  __shared__ uint32_t s_llb[36], s_mlb[53];     // base | extra bits << 24 (a constant-memory lookup is a global load here)
  __shared__ uint16_t s_t[ZG_FSE_SLOT_U32];      // per state: symbol << 4 | state bits
  __shared__ uint32_t s_tr[ZG_SP_T / 64][64 * 25]; // a wave's records on their way out (see below)
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t b = d.seq_blocks[blockIdx.x];
  if (d.status[b]) return;                       // the bitstream (or a table) failed: nothing to post-process
  const ZgBlock blk = d.blocks[b];
  {
    const int32_t sl[3] = {blk.ll_slot, blk.ml_slot, blk.of_slot};          // in the order of the slot layout (LL, ML, OF)
    const uint32_t offs[4] = {ZG_FSE_LL_OFF, ZG_FSE_ML_OFF, ZG_FSE_OF_OFF, ZG_FSE_SLOT_U32};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const uint32_t* g_t = d.fse_arena + (uint64_t)sl[k] * ZG_FSE_SLOT_U32;   // (the block passed zg_k_seq: all three slots are set)
      for (uint32_t i = offs[k] + t; i < offs[k + 1]; i += ZG_SP_T) { const uint32_t v = g_t[i]; s_t[i] = (uint16_t)((ZG_FSE_SYM(v) << 4) | ZG_FSE_NB(v)); }
    }
  }
  const uint32_t nseq = blk.nseq, regen = blk.regen_size;
  const uint32_t wlim = zg_sweep_window(d, d.frames[blk.frame]);
  const uint8_t* bs = d.src + blk.src_off + d.aux[b].seq_bits_off;
  const uint32_t* raw = d.raw_arena + blk.seq_base;
  ZgSeq* out = d.seq_arena + blk.seq_base;
  if (t == 0) s_err = 0xFFFFFFFFu;
  if (t < 36) s_llb[t] = ZG_LL_BASE[t] | ((uint32_t)ZG_LL_BITS[t] << 24);
  if (t < 53) s_mlb[t] = ZG_ML_BASE[t] | ((uint32_t)ZG_ML_BITS[t] << 24);
  ZgHistMap carry = zg_map_identity();
  uint32_t lit_carry = 0, out_carry = 0;
  uint32_t pos_carry = d.seq_out[b].pad;          // bit position where the next pass's first sequence starts (left by zg_k_seq)
  uint32_t err_i0 = 0;                            // the pass in which a sequence was rejected starts here (s_err holds its index in the pass)
  __syncthreads();
  for (uint32_t i0 = 0; i0 < nseq; i0 += ZG_SP_T * ZG_SP_S) {
    const uint32_t ib = i0 + t * ZG_SP_S;
    const uint32_t n = ib < nseq ? (nseq - ib < ZG_SP_S ? nseq - ib : ZG_SP_S) : 0u;
    uint32_t r[ZG_SP_S];
    if (n == ZG_SP_S) {
#pragma unroll
      for (int j = 0; j < ZG_SP_S; j += 4) {
        const zg_v4u v = *(const zg_gv4u*)(raw + ib + j);
        r[j] = v.x; r[j + 1] = v.y; r[j + 2] = v.z; r[j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < ZG_SP_S; j++) r[j] = (uint32_t)j < n ? raw[ib + j] : 0u;
    }
    // symbols and bit counts from the recorded states: a sequence takes its three codes' extra bits and, unless it is the
    // block's last one, the three state updates (:203-206)
    uint32_t codes[ZG_SP_S];                       // of | ml << 8 | ll << 16
    uint32_t P[ZG_SP_S];
    {
      uint32_t tx = 0;
#pragma unroll
      for (int j = 0; j < ZG_SP_S; j++) {
        const uint32_t e_of = s_t[ZG_FSE_OF_OFF + ZG_RAW_OF(r[j])], e_ml = s_t[ZG_FSE_ML_OFF + ZG_RAW_ML(r[j])], e_ll = s_t[ZG_FSE_LL_OFF + ZG_RAW_LL(r[j])];
        const uint32_t of_code = (e_of >> 4) & 31u, ml_code = e_ml >> 4, ll_code = e_ll >> 4;
        const uint32_t vl = s_llb[ll_code < 36 ? ll_code : 0], vm = s_mlb[ml_code < 53 ? ml_code : 0];
        const uint32_t upd = ib + (uint32_t)j + 1u == nseq ? 0u : (e_of & 15u) + (e_ml & 15u) + (e_ll & 15u);
        const bool have = (uint32_t)j < n;             // (a slot past the end takes no bits: its reads below stay at the last position)
        codes[j] = have ? of_code | (ml_code << 8) | (ll_code << 16) : 0u;
        P[j] = tx;
        tx += have ? of_code + (vl >> 24) + (vm >> 24) + upd : 0u;
      }
      uint32_t sx = tx;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const uint32_t px = __shfl_up(sx, off, 64); if ((int)lane >= off) sx += px; }
      if (lane == 63) s_wx[wv] = sx;
      __syncthreads();
      uint32_t before = sx - tx, all = 0;
#pragma unroll
      for (uint32_t w = 0; w < ZG_SP_T / 64; w++) { const uint32_t wx = s_wx[w]; if (w < wv) before += wx; all += wx; }
#pragma unroll
      for (int j = 0; j < ZG_SP_S; j++) P[j] = pos_carry - before - P[j];
      pos_carry -= all;
    }
    // values: the three extra-bit fields of a sequence are adjacent, [q_ll, P), at most 63 bits: three dwords cover them
    uint32_t ll[ZG_SP_S], ml[ZG_SP_S], of[ZG_SP_S];
#pragma unroll
    for (int j = 0; j < ZG_SP_S; j++) {
      const uint32_t of_code = codes[j] & 255u, ml_code = (codes[j] >> 8) & 255u, ll_code = codes[j] >> 16;
      const uint32_t vl = s_llb[ll_code < 36 ? ll_code : 0], vm = s_mlb[ml_code < 53 ? ml_code : 0];
      const uint32_t xb_ll = vl >> 24, xb_ml = vm >> 24;
      const uint32_t q_ll = P[j] - of_code - xb_ml - xb_ll;              // >= 0 for every record zg_k_seq emitted
      const uint64_t pa = (uint64_t)(bs + (q_ll >> 3));
      const zg_v3u wv = *(const zg_gv3u*)(pa & ~3ull);                    // dword-aligned: a misaligned load is split by the hardware
      const uint32_t sh = ((uint32_t)pa & 3u) * 8u + (q_ll & 7u);         // <= 31
      const uint32_t lo = __builtin_amdgcn_alignbit(wv.y, wv.x, sh), hi = __builtin_amdgcn_alignbit(wv.z, wv.y, sh);
      const uint32_t ll_add = __builtin_amdgcn_ubfe(lo, 0u, xb_ll);
      const uint32_t ml_add = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hi, lo, xb_ll), 0u, xb_ml);
      const uint32_t so = xb_ll + xb_ml;                                  // <= 32
      const uint32_t ov = so >= 32u ? hi : __builtin_amdgcn_alignbit(hi, lo, so);
      const bool have = (uint32_t)j < n;
      ll[j] = have ? (vl & 0xFFFFFFu) + ll_add : 0u;
      ml[j] = have ? (vm & 0xFFFFFFu) + ml_add : 0u;
      of[j] = have ? __builtin_amdgcn_ubfe(ov, 0u, of_code) + (1u << of_code) : 0u;   // 0 marks "no sequence"
    }
    // history inside the thread's run, relative to its start; thread totals
    uint32_t h0 = 1u << 30, h1 = 2u << 30, h2 = 3u << 30, tl = 0, to = 0, act[ZG_SP_S];
    bool far = false;
#pragma unroll
    for (int j = 0; j < ZG_SP_S; j++) {
      act[j] = 1;
      if (of[j]) {
        // offsets >= 2^30 (offset codes 30 and 31) would collide with the symbolic slots: they travel as ZG_OFF_HUGE, which no
        // real offset equals and which is out of reach as long as fewer than 2^30 bytes are held — the case in which the
        // reference rejects them too (zg_k_exact picks its error; with 1 GiB held undrained it answers ZG_UNSUPPORTED)
        const bool tb = of[j] > 3u && of[j] - 3u >= (1u << 30);
        far = far || tb || (of[j] > 3u && of[j] - 3u > wlim);
        act[j] = zg_hist_step(tb ? ZG_OFF_HUGE + 3u : of[j], ll[j], h0, h1, h2);
      }
      tl += ll[j]; to += ll[j] + ml[j];
    }
    // inclusive scans over the wave of the thread totals
    uint32_t sl = tl, so = to;
    ZgHistMap sc = {{h0, h1, h2}};
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t pl = __shfl_up(sl, off, 64), po = __shfl_up(so, off, 64);
      if ((int)lane >= off) { sl += pl; so += po; }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      ZgHistMap pm;
      pm.s[0] = __shfl_up(sc.s[0], off, 64); pm.s[1] = __shfl_up(sc.s[1], off, 64); pm.s[2] = __shfl_up(sc.s[2], off, 64);
      if ((int)lane >= off) sc = zg_map_compose(pm, sc);
    }
    if (lane == 63) { s_wm[wv] = sc; s_wl[wv] = sl; s_wo[wv] = so; }
    // exclusive: what the lanes before this one did
    ZgHistMap ex;
    ex.s[0] = __shfl_up(sc.s[0], 1, 64); ex.s[1] = __shfl_up(sc.s[1], 1, 64); ex.s[2] = __shfl_up(sc.s[2], 1, 64);
    if (lane == 0) ex = zg_map_identity();
    __syncthreads();
    ZgHistMap pre = carry, tot = carry;           // tot runs through the waves' maps; pre is its value in front of this wave
    uint32_t pl = lit_carry, po = out_carry, totl = lit_carry, toto = out_carry;
#pragma unroll
    for (uint32_t w = 0; w < ZG_SP_T / 64; w++) {
      const ZgHistMap wm = s_wm[w];
      const uint32_t wl = s_wl[w], wo = s_wo[w];
      if (w == wv) pre = tot;
      if (w < wv) { pl += wl; po += wo; }
      tot = zg_map_compose(tot, wm); totl += wl; toto += wo;
    }
    pre = zg_map_compose(pre, ex);                 // history before this thread's first sequence, relative to the block start
    uint32_t lit_pos = pl + sl - tl, out_pos = po + so - to;
    {
      // The wave's 512 records are 6 KiB in a row; written thread by thread they would be dword stores 96 bytes apart (24
      // store instructions, each touching 48 cache lines). A full wave passes them through LDS instead (25-dword rows: no
      // bank conflicts) and stores 16 bytes per lane; the last, partial pass of a block stores record by record.
      const bool full = __all(n == ZG_SP_S);
      uint32_t* row = &s_tr[wv][lane * 25u];
      uint32_t bad = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < ZG_SP_S; j++) {
        if ((uint32_t)j < n) {
          const uint32_t actual = zg_map_apply(pre, act[j]);
          uint32_t e = ZG_OK;
          if ((uint64_t)out_pos + ll[j] + ml[j] >= (1ull << 31)) e = ZG_UNSUPPORTED;
          if ((uint64_t)lit_pos + ll[j] > regen) e = ZG_EXE_NOT_ENOUGH_LITERALS;   // sequence_execution.rs:14-19
          if (actual == 0) e = ZG_EXE_ZERO_OFFSET;                                  // :28-30
          if (e && bad == 0xFFFFFFFFu) bad = ((t * ZG_SP_S + (uint32_t)j) << 8) | e;
          const uint32_t mdst = out_pos + ll[j];
          const uint32_t w1 = ZG_SEQ_W1(mdst, ml[j]), w2 = ZG_SEQ_W2(lit_pos, ml[j]);
          if (full) { row[3 * j] = actual; row[3 * j + 1] = w1; row[3 * j + 2] = w2; }
          else {
            // (a three-element vector type would be stored as four dwords and clobber the next record's first field)
            __attribute__((address_space(1))) uint32_t* qo = (__attribute__((address_space(1))) uint32_t*)(out + ib + j);
            qo[0] = actual; qo[1] = w1; qo[2] = w2;
          }
          lit_pos += ll[j]; out_pos += ll[j] + ml[j];
        }
      }
      if (bad != 0xFFFFFFFFu) atomicMin(&s_err, bad);                               // the first failing sequence decides
      if (full) {
        zg_wave_publish();
        __attribute__((address_space(1))) uint32_t* wo = (__attribute__((address_space(1))) uint32_t*)(out + (ib - lane * ZG_SP_S));   // the wave's first record
#pragma unroll
        for (int q0 = 0; q0 < ZG_SP_S * 3 * 64 / 4; q0 += 64) {                    // 4 dwords per lane and step: they never straddle two rows (24 % 4 == 0)
          const uint32_t m = 4u * ((uint32_t)q0 + lane);
          const uint32_t* src4 = &s_tr[wv][(m / 24u) * 25u + m % 24u];
          const zg_v4u v = {src4[0], src4[1], src4[2], src4[3]};
          *(__attribute__((address_space(1))) zg_v4u*)(wo + m) = v;
        }
      }
    }
    // a new offset beyond the frame's window: legal here, but no encoder emits one and the split sweep relies on their absence
    // (repeat codes only repeat what a new offset set, or the frame's initial history, which the host checks)
    if (far) d.totals[3] = 1u;
    carry = tot; lit_carry = totl; out_carry = toto;
    __syncthreads();
    if (s_err != 0xFFFFFFFFu) { err_i0 = i0; break; }
  }
  if (t == 0) {
    ZgBlockSeqOut so;
    so.sum_ll = lit_carry; so.sum_ml = out_carry - lit_carry;
    so.hist_end[0] = carry.s[0]; so.hist_end[1] = carry.s[1]; so.hist_end[2] = carry.s[2];
    so.pad = s_err != 0xFFFFFFFFu ? err_i0 + (s_err >> 8) + 1u : 0u;   // 1 + the sequence that cannot be executed (no literals left, offset 0): zg_k_exact looks at the ones in front of it
                                                                       // (its index in the BLOCK: until late in round 6 the pass's start was missing, and a block whose rejected sequence lay behind the
                                                                       //  first 2048 hid an offset error in front of it from zg_k_exact — tools/dev/soak.py seed 41)
    d.seq_out[b] = so;
    if (s_err != 0xFFFFFFFFu) zg_set_status(d.status, b, (int)(s_err & 0xFFu));
  }
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_scan: per frame — output position of every block and the offset history at every block start.
// The history recurrence (sequence_execution.rs:59-118; never reset between blocks, scratch.rs:22) is a
// composition of per-block maps on three symbolic slots, so it is scanned like a prefix sum.
// ------------------------------------------------------------------------------------------------------------
// ZG_SCAN_T threads per frame: 1024 for frames of many blocks, one wave for a submit of short frames (16384 single-block frames:
// 16384 workgroups of 1024 threads that hold one block each took 0.69 ms)
// Round 6: ZG_SCAN_I consecutive blocks per thread (8 in the 1024-thread form: the 7630 blocks of the 1e9-byte frame are ONE pass — their
// loads are requested together, a thread composes its own blocks serially, one wave scan, one cross-wave step — where eight passes of one
// block per thread, each with its own memory round trip and four barriers, took 0.10 ms).
template <int ZG_SCAN_T, int ZG_SCAN_I>
__global__ void __launch_bounds__(ZG_SCAN_T) zg_k_scan(ZgBatchDev d) {
  __shared__ uint64_t s_size[ZG_SCAN_T / 64];
  __shared__ ZgHistMap s_map[ZG_SCAN_T / 64];
  __shared__ uint32_t s_bad;       // chunk-local index of the first failing block
  __shared__ uint32_t s_badst;
  __shared__ uint32_t s_slow;      // some block regenerates more than 128 KiB (non-conforming): in-order fallback
  const uint32_t f = blockIdx.x, t = threadIdx.x;
  const ZgFrame fr = d.frames[f];
  uint64_t carry_size = 0;
  ZgHistMap carry_map = zg_map_identity();
  uint32_t good = fr.nblocks, bad_status = 0;
  __shared__ unsigned long long s_counted;   // output of the good compressed blocks (what DecodeBuffer counts, decode_buffer.rs:74-77,108)
  uint64_t my_counted = 0;
  if (t == 0) { s_slow = 0; s_counted = 0; }
  constexpr uint32_t CH = ZG_SCAN_T * ZG_SCAN_I;
  for (uint32_t c0 = 0; c0 < fr.nblocks; c0 += CH) {
    uint64_t size[ZG_SCAN_I];
    ZgHistMap m[ZG_SCAN_I];
    uint32_t st[ZG_SCAN_I];
    bool comp[ZG_SCAN_I];
    if (t == 0) { s_bad = 0xFFFFFFFFu; s_badst = 0; }
    __syncthreads();
    // every load of the thread's blocks is requested before any is used (no load behind a branch on another load's value: one round trip)
    uint32_t f_host[ZG_SCAN_I], f_stat[ZG_SCAN_I], f_regen[ZG_SCAN_I], f_nseq[ZG_SCAN_I], f_bt[ZG_SCAN_I];
    ZgBlockSeqOut f_so[ZG_SCAN_I];
#pragma unroll
    for (int j = 0; j < ZG_SCAN_I; j++) {
      const uint32_t i = c0 + t * ZG_SCAN_I + (uint32_t)j;
      const uint32_t b = fr.first_block + (i < fr.nblocks ? i : fr.nblocks - 1u);     // (clamped: the last block again)
      const ZgBlock* blk = &d.blocks[b];
      f_host[j] = blk->host_status; f_regen[j] = blk->regen_size; f_nseq[j] = blk->nseq; f_bt[j] = blk->btype;
      f_stat[j] = d.status[b];
      f_so[j] = d.seq_out[b];
    }
#pragma unroll
    for (int j = 0; j < ZG_SCAN_I; j++) {
      const uint32_t i = c0 + t * ZG_SCAN_I + (uint32_t)j;
      size[j] = 0; m[j] = zg_map_identity(); st[j] = 0; comp[j] = false;
      if (i < fr.nblocks) {
        st[j] = f_host[j] ? f_host[j] : f_stat[j];
        comp[j] = f_bt[j] == ZG_BT_COMPRESSED;
        if (!comp[j] || f_nseq[j] == 0) size[j] = f_regen[j];
        else {
          size[j] = (uint64_t)f_regen[j] + f_so[j].sum_ml;
          if (size[j] > ZG_FLAT_MAX) s_slow = 1;
          m[j].s[0] = f_so[j].hist_end[0]; m[j].s[1] = f_so[j].hist_end[1]; m[j].s[2] = f_so[j].hist_end[2];
        }
        if (st[j]) atomicMin(&s_bad, t * ZG_SCAN_I + (uint32_t)j);
      }
    }
    __syncthreads();
    const uint32_t bad = s_bad;
    // the failing block and everything after it produce nothing; the thread's own blocks are composed serially
    uint64_t tsum = 0;
    ZgHistMap tmapl = zg_map_identity();
#pragma unroll
    for (int j = 0; j < ZG_SCAN_I; j++) {
      if (t * ZG_SCAN_I + (uint32_t)j >= bad) { size[j] = 0; m[j] = zg_map_identity(); }
      if (comp[j]) my_counted += size[j];
      tsum += size[j];
      tmapl = zg_map_compose(tmapl, m[j]);
    }
    // inclusive scans over the threads of the chunk: inside the wave with shuffles, across the waves through LDS
    uint64_t isz = tsum;
    ZgHistMap im = tmapl;
    {
      const uint32_t lane = t & 63;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)isz, off, 64), hi = __shfl_up((uint32_t)(isz >> 32), off, 64);
        ZgHistMap pm;
        pm.s[0] = __shfl_up(im.s[0], off, 64); pm.s[1] = __shfl_up(im.s[1], off, 64); pm.s[2] = __shfl_up(im.s[2], off, 64);
        if ((int)lane >= off) { isz += ((uint64_t)hi << 32) | lo; im = zg_map_compose(pm, im); }
      }
      if (lane == 63) { s_size[t >> 6] = isz; s_map[t >> 6] = im; }
    }
    __syncthreads();
    uint64_t wsz = 0, tsz = 0;                    // what the earlier waves of the chunk add; the whole chunk
    ZgHistMap wmap = zg_map_identity(), tmap = zg_map_identity();
    for (uint32_t w = 0; w < ZG_SCAN_T / 64; w++) {
      if (w < (t >> 6)) { wsz += s_size[w]; wmap = zg_map_compose(wmap, s_map[w]); }
      tsz += s_size[w]; tmap = zg_map_compose(tmap, s_map[w]);
    }
    // history before a thread's first block: everything before the chunk, the earlier waves, the earlier lanes of its wave
    ZgHistMap prev;
    prev.s[0] = __shfl_up(im.s[0], 1, 64); prev.s[1] = __shfl_up(im.s[1], 1, 64); prev.s[2] = __shfl_up(im.s[2], 1, 64);
    if ((t & 63) == 0) prev = zg_map_identity();
    {
      uint64_t excl = carry_size + wsz + isz - tsum;                             // sizes before the thread's first block
      ZgHistMap pre = zg_map_compose(zg_map_compose(carry_map, wmap), prev);
#pragma unroll
      for (int j = 0; j < ZG_SCAN_I; j++) {
        const uint32_t li = t * ZG_SCAN_I + (uint32_t)j, i = c0 + li;
        if (i < fr.nblocks) {
          ZgBlockPos p;
          p.out_base = excl;
          p.hist_init[0] = zg_sym_resolve(pre.s[0], fr.hist_init);
          p.hist_init[1] = zg_sym_resolve(pre.s[1], fr.hist_init);
          p.hist_init[2] = zg_sym_resolve(pre.s[2], fr.hist_init);
          p.active = li < bad ? 1u : 0u;
          d.pos[fr.first_block + i] = p;
          if (li == bad) s_badst = st[j];
        }
        excl += size[j];
        pre = zg_map_compose(pre, m[j]);
      }
    }
    __syncthreads();
    if (bad != 0xFFFFFFFFu) {  // sizes / maps of blocks from the failing one on are zero / identity
      good = c0 + bad;
      bad_status = s_badst;
      carry_size += tsz;
      carry_map = zg_map_compose(carry_map, tmap);
      // blocks after this chunk are inactive
      for (uint32_t j = c0 + CH + t; j < fr.nblocks; j += ZG_SCAN_T) d.pos[fr.first_block + j].active = 0;
      break;
    }
    carry_size += tsz;
    carry_map = zg_map_compose(carry_map, tmap);
    __syncthreads();
  }
  if (my_counted) atomicAdd(&s_counted, (unsigned long long)my_counted);
  __syncthreads();
  if (t == 0) {
    ZgFrameOut fo;
    fo.counted = s_counted;
    fo.out_base = 0; fo.out_size = carry_size; fo.status = bad_status; fo.bad_block = good;
    fo.hist_end[0] = zg_sym_resolve(carry_map.s[0], fr.hist_init);
    fo.hist_end[1] = zg_sym_resolve(carry_map.s[1], fr.hist_init);
    fo.hist_end[2] = zg_sym_resolve(carry_map.s[2], fr.hist_init);
    fo.good_blocks = good;
    fo.fast = (s_slow || (d.flags & 1u)) ? 0u : 1u;
    fo.err_packed = 0xFFFFFFFFu;
    fo.og_base = 0;
    if (d.nframes == 1u) {   // a submit of ONE frame needs no scan over the frames (zg_k_scanf is not launched): its place and the totals, here
      fo.out_base = fr.fixed_base ? fr.out_base_fixed : 0ull;
      const uint64_t tot = fr.fixed_base ? 0ull : carry_size;
      d.totals[0] = (uint32_t)tot; d.totals[1] = (uint32_t)(tot >> 32); d.totals[2] = (d.dst_cap_pre && tot > d.dst_cap_pre) ? 1u : 0u;
    }
    d.frame_out[f] = fo;
  }
}

// frames are laid out back to back in the output, like FrameDecoder::decode_all (frame_decoder.rs:541-577). The host reads
// the sizes after this kernel and allocates the output and the flatten scratch exactly (the scratch is indexed by output
// position: one word per output byte).
__global__ void __launch_bounds__(1024) zg_k_scanf(ZgBatchDev d) {
  __shared__ uint64_t s_v[1024];
  const uint32_t t = threadIdx.x;
  uint64_t carry = 0;
  for (uint32_t c0 = 0; c0 < d.nframes; c0 += 1024) {
    uint32_t f = c0 + t;
    s_v[t] = (f < d.nframes && !d.frames[f].fixed_base) ? d.frame_out[f].out_size : 0;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      uint64_t v = t >= off ? s_v[t - off] : 0;
      __syncthreads();
      s_v[t] += v;
      __syncthreads();
    }
    if (f < d.nframes) {
      const uint64_t rel = carry + (t ? s_v[t - 1] : 0);
      d.frame_out[f].out_base = d.frames[f].fixed_base ? d.frames[f].out_base_fixed : rel;
      d.frame_out[f].og_base = d.frames[f].fixed_base ? 0 : rel;      // (a streaming run holds exactly one frame)
    }
    carry += s_v[1023];
    __syncthreads();
  }
  if (t == 0) { d.totals[0] = (uint32_t)carry; d.totals[1] = (uint32_t)(carry >> 32); d.totals[2] = (d.dst_cap_pre && carry > d.dst_cap_pre) ? 1u : 0u; }
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_lit: blocks that do not depend on earlier output at all — raw blocks, RLE blocks (block_decoder.rs:55-82) and
// compressed blocks without sequences (block_decoder.rs:184-194: the literals are the block). Literal runs of blocks
// WITH sequences (DecodeBuffer::push, decode_buffer.rs:74-77) are placed by zg_k_flatten (flatten path) or zg_k_lz.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void zg_wg_copy(uint8_t* dst, const uint8_t* src, uint64_t n, uint32_t t, uint32_t T) {
  uint64_t n8 = n >> 3;
  for (uint64_t i = t; i < n8; i += T) ((zg_u64u*)(dst + i * 8))->v = zg_ld64(src + i * 8);
  for (uint64_t i = (n8 << 3) + t; i < n; i += T) dst[i] = src[i];
}
__device__ __forceinline__ void zg_wg_fill(uint8_t* dst, uint8_t byte, uint64_t n, uint32_t t, uint32_t T) {
  uint64_t v = 0x0101010101010101ull * byte, n8 = n >> 3;
  for (uint64_t i = t; i < n8; i += T) ((zg_u64u*)(dst + i * 8))->v = v;
  for (uint64_t i = (n8 << 3) + t; i < n; i += T) dst[i] = byte;
}

__global__ void __launch_bounds__(256) zg_k_lit(ZgBatchDev d) {
  if (d.totals[2]) return;
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  const ZgBlockPos p = d.pos[b];
  if (!p.active) return;
  const ZgBlock blk = d.blocks[b];
  uint8_t* out = d.dst + d.frame_out[blk.frame].out_base + p.out_base;
  const uint8_t* body = d.src + blk.src_off;
  if (blk.btype == ZG_BT_RAW) { zg_wg_copy(out, body, blk.regen_size, t, 256); return; }
  if (blk.btype == ZG_BT_RLE) { zg_wg_fill(out, body[0], blk.regen_size, t, 256); return; }
  if (blk.nseq) return;
  if (blk.lit_type == ZG_LT_RLE) zg_wg_fill(out, body[blk.lit_off], blk.regen_size, t, 256);
  else if (blk.lit_type >= ZG_LT_COMPRESSED && (d.flags & ZG_FLAG_LIT_DIRECT)) return;   // zg_k_huf wrote them here itself (zg_huf.h)
  else zg_wg_copy(out, blk.lit_type == ZG_LT_RAW ? body + blk.lit_off : d.lit_arena + blk.lit_base, blk.regen_size, t, 256);
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_flatten + zg_k_sweep: LZ77 execution (execute_sequences, sequence_execution.rs:5-54; DecodeBuffer::repeat,
// decode_buffer.rs:79-141) without walking the frame's sequences one after the other.
//
// Chains of matches (a match copying the output of an earlier match ...) are what makes in-order execution slow on
// a GPU: every hop is a memory round trip, and text has chains millions of hops deep. Instead every MATCH byte is
// resolved to an "effective offset" e: byte[pos] = byte[pos - e], where pos - e is a byte that is final before the
// byte's unit is swept — a literal byte anywhere, or any byte before the unit (a unit = a run of consecutive blocks of
// one frame). Literal bytes have e = 0. No byte VALUES travel through this stage: only offsets.
//   zg_k_flatten  (all units at once, one workgroup per unit) walks the unit in tiles held in LDS. A tile byte points to
//              its parent byte (position - offset); pointer jumping inside the tile (u16 pointers) shortens every chain
//              to its tile root: a literal byte, or a match byte whose parent lies before the tile. Such a parent is in
//              an earlier tile of the unit (its e is final: one gather from the scratch, e = offset + e[parent]) or
//              before the unit (e = offset). Literal bytes are copied to the output on the way.
//   zg_k_sweep one launch per unit index, in frame order (a kernel boundary is the cheapest chip-wide barrier there is,
//              and needs no co-residency assumption): every match byte of the unit is one independent gather from
//              bytes that are final by then.
// ------------------------------------------------------------------------------------------------------------

// zg_flat1_unit: the pointer-mode body of zg_k_flatten (zg_flat1.h, written against the zx_* primitives like zg_flat4.h)
#include "zg_flat1.h"

// ------------------------------------------------------------------------------------------------------------
// zg_flat4_unit: the flatten of a frame's first unit at dword granularity, resolved to byte values (direct unit) — the body is
// zg_flat4.h, written against the zx_* primitives above so that tests/emu runs the same source on the CPU.
// ------------------------------------------------------------------------------------------------------------
#include "zg_flat4.h"

// everything a sweep workgroup needs to know about its unit, in one 32-byte descriptor
__device__ __forceinline__ ZgSweepDesc zg_sweep_desc(const ZgBatchDev& d, uint32_t u, uint32_t size) {
  const ZgUnit un = d.units[u];
  const ZgFrameOut fo = d.frame_out[un.frame];
  ZgSweepDesc sd;
  sd.out = (uint64_t)(d.dst + fo.out_base + d.pos[un.first_block].out_base);
  sd.og = (uint64_t)(d.og + fo.og_base + d.pos[un.first_block].out_base);
  sd.size = size;
  const uint32_t w = zg_sweep_window(d, d.frames[un.frame]);
  sd.head = sd.size > w ? (sd.size - w) / (4u * ZG_SW_T * ZG_SW_B) : 0u;
  // (a frame one of whose units failed in the flatten is swept like any other: a unit's size ends in front of the block that failed —
  //  zg_flat1_unit — so what the good blocks in front of it copy is resolved, and what lies behind it is nobody's output)
  sd.live = (!d.totals[2] && fo.fast) ? 1u : 0u;
  sd.pad = 0;
  return sd;
}

// zg_k_flatten: one workgroup per unit. A frame's first unit (nothing in front of it to copy from) is resolved to bytes by the
// dword-granular body (zg_flat4_unit: no scratch, no sweep step); every other unit gets its effective offsets from the
// byte-granular body above (zg_flat1_unit) — measured on the 1e9-byte frame, the dword-granular body in pointer mode issued 23 %
// more vector instructions than this one (its pointer jumping works on clumps of four bytes that share their fate, and a wave
// lasts as long as its busiest lane), so it is used where it wins: where values can flow.
// both != 0: one workgroup per unit of the submit, either body (round 4's form: ZGPU_FLAT4=0, the 512-thread shape, ramped units);
// both == 0: the first nunits - ndirect entries of unit_list — the direct units have a kernel of their own (zg_k_flatten4 below)
template <int T, int TS, int SPT>
__global__ void __launch_bounds__(T, 4) zg_k_flatten(ZgBatchDev d, uint32_t both) {
  __shared__ union { ZgFlat1Lds<T, TS, SPT> p; ZgFlat4Lds<T, TS, SPT> v; } s_u;
  const uint32_t ui = both ? blockIdx.x : d.unit_list[blockIdx.x];
  if (threadIdx.x == 0) { d.unit_info[ui].size = 0; d.unit_info[ui].noseq = 0; }
  if (d.units[ui].noseq & ZG_UNIT_DIRECT) zg_flat4_unit<T, TS, SPT>(d, ui, s_u.v);
  else zg_flat1_unit<T, TS, SPT>(d, ui, s_u.p);
  // The sweep chain runs beside this kernel (one long frame in units that grow along it: see BatchBuilder::finish): the unit is
  // handed to its sweep step here — descriptor, then everything this workgroup wrote made visible to the whole device (release),
  // then the flag the step polls. (Producer recipe of the hardware guide: plain stores -> workgroup barrier -> one lane's
  // agent-scope release -> vmcnt(0) -> relaxed agent-scope flag store.)
  if (ZG_DEVSW(d.overlap_epoch)) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t de = d.units[ui].desc;
      if (de != 0xFFFFFFFFu) d.sweep_desc[de] = zg_sweep_desc(d, ui, d.unit_info[ui].size);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&d.unit_info[ui].done, d.overlap_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// zg_k_flatten4: the direct units of a submit (a frame's first unit: resolved to bytes, zg_flat4.h) as a kernel of their own, so that they
// get their own occupancy: with 8 KiB tiles and 8-byte sequence records the body needs 34 KB of LDS and 64 registers — TWO 1024-thread
// workgroups per CU, where the pointer-mode body's 134 KB (which the direct units inherited while both bodies were one kernel) allows
// one. A submit of single-block frames (BASELINE config 4b) or of Silesia-sized frames is direct units only. WPE: waves per SIMD asked for.
template <int T, int TS, int SPT, int WPE>
__global__ void __launch_bounds__(T, WPE) zg_k_flatten4(ZgBatchDev d) {
  __shared__ ZgFlat4Lds<T, TS, SPT> s_v;
  const uint32_t ui = d.unit_list[d.nunits - d.ndirect + blockIdx.x];
  if (threadIdx.x == 0) { d.unit_info[ui].size = 0; d.unit_info[ui].noseq = 0; }
  zg_flat4_unit<T, TS, SPT>(d, ui, s_v);
}

// zg_k_exact: the reference's DecodeBuffer bookkeeping replayed exactly (zg_exact.h), one workgroup per frame; launched by the
// host only for submits it can matter to (Batch::sync)
#include "zg_exact.h"
__global__ void __launch_bounds__(256) zg_k_exact(ZgBatchDev d, uint32_t drain_rule) {
  __shared__ ZgExactLds<256> s_l;
  zg_exact_frame<256>(d, blockIdx.x, drain_rule, s_l);
}

// zg_k_swprep: one thread per (step, unit) entry: everything a sweep workgroup needs about its unit in one 32-byte
// descriptor, so that a sweep launch starts with ONE dependent scalar load instead of a chain of five.
__global__ void __launch_bounds__(256) zg_k_swprep(ZgBatchDev d, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t u = d.step_units[i];
  d.sweep_desc[i] = zg_sweep_desc(d, u, d.unit_info[u].size);
}

// zg_k_sweep: one launch per step; step s fills unit s of every frame that has one (blockIdx.y picks the unit from the
// step's list, blockIdx.x a ZG_SW_BATCH-byte slice of it). A group of four output bytes at w: byte i comes from (w + i) - e_i =
// byte i of the dword at w - e_i, so one load per DISTINCT offset of the group serves it (usually one or two: a match
// boundary); literal bytes (e = 0) are already in place and come from the dword at w itself. The loads go through a
// buffer resource: a load that is not needed gets an out-of-range offset (no traffic, no branch), so all loads of a
// thread are in flight together.
//
// The split sweep (part 1 / 2): a match reaches at most `window` bytes back, so everything a later unit can copy from is
// the TAIL of the units in front of it. Only the tails (part 1) form the chain of steps; the heads (part 2) are filled
// beside it, many units per launch, on the engine's second stream. part 0 is the whole unit.
// OGM (timing experiments only, ZGPU_SWEEP_MODE 5..8: wrong results): how many bytes of scratch a group reads — 5: 8, 6: 4, 8: 12,
// 7: 8 and then 8 more at an address that depends on the first (what a directory + entries format would cost a step)
template <int OGM>
__device__ __forceinline__ void zg_sweep_body(const ZgBatchDev& d, uint32_t list_off, uint32_t nbatch, uint32_t dbgmode_arg, uint32_t part) {
  const uint32_t dbgmode = ZG_DEVSW(dbgmode_arg);             // (the product build: constant 0, the timing modes below fold away)
  if (ZG_DEVSW(d.overlap_epoch)) {
    // the flatten may still be at this unit (it runs beside the chain): one lane polls the unit's flag. A step that finds it set
    // — the usual case: units finish in frame order, ahead of the chain — reads data that was released before this launch began;
    // one that had to wait acquires (consumer recipe of the hardware guide). The wait is bounded: the flatten does not depend
    // on this kernel, but a hang must never be the way a mistake shows.
    if (threadIdx.x == 0) {
      const uint32_t u = d.step_units[list_off + blockIdx.y];
      uint32_t spins = 0;
      bool waited = false;
      while (__hip_atomic_load(&d.unit_info[u].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != d.overlap_epoch) {
        __builtin_amdgcn_s_sleep(16);
        waited = true;
        if (++spins > (1u << 24)) { atomicCAS(&d.frame_out[d.units[u].frame].status, 0u, (uint32_t)ZG_INTERNAL); break; }
      }
      if (waited) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  const ZgSweepDesc sd = d.sweep_desc[list_off + blockIdx.y];
  const uint32_t t = threadIdx.x, size = sd.size;
  constexpr uint32_t BG = ZG_SW_T * ZG_SW_B;                  // groups per batch (ZG_SW_BATCH bytes of output)
  const uint32_t b0 = blockIdx.x * nbatch + (part == 1u ? sd.head : 0u);   // the workgroup's batches: b0 .. b0 + nbatch - 1, one after the other
  if (!sd.live || 4ull * b0 * BG >= size) return;
  if (part == 2u && b0 >= sd.head) return;
  if (dbgmode == 1) return;                                   // timing experiments (ZGPU_SWEEP_MODE): launch floor
  typedef __attribute__((address_space(1))) uint8_t zg_gu8;
  typedef __attribute__((address_space(1))) uint32_t zg_gu32;
  zg_gu8* out = (zg_gu8*)sd.out;                               // global, not flat, accesses
  const zg_gu32* og = (const zg_gu32*)sd.og;
  const uint32_t n4 = size >> 2;
  const uint32_t nb_all = part == 2u ? sd.head : (n4 + BG - 1) / BG;   // batches of the unit (part 2: of its head)
  const uint32_t nb = b0 + nbatch <= nb_all ? nbatch : (nb_all > b0 ? nb_all - b0 : 0u);
  // sources lie at most 2^31 bytes before the unit's first byte (offsets < 2^30 + a unit): a resource that starts there
  const uint32_t lowb = (uint32_t)sd.out & 3u;
  const uint32_t rel0 = 0x80000000u + lowb;                    // resource offset of the unit's first byte
  const __amdgpu_buffer_rsrc_t rs = zg_make_rsrc((const void*)(sd.out - lowb - 0x80000000ull), rel0 + size + 8u);
  auto load_og = [&](uint32_t bt, uint4 (&o)[ZG_SW_B]) {      // clamped, not branched: the loads of a batch overlap
#pragma unroll
    for (int k = 0; k < ZG_SW_B; k++) {
      const uint32_t g = bt * BG + t + k * ZG_SW_T;
      const uint64_t gi = g < n4 ? g : 0u;
      if (OGM == 0) { const zg_v4u v = *(const zg_gv4u*)(og + 4 * gi); o[k] = make_uint4(v.x, v.y, v.z, v.w); }
      else if (OGM == 6) { const uint32_t v = og[gi]; o[k] = make_uint4(v, v, v, v); }
      else if (OGM == 8) { const zg_v3u v = *(const zg_gv3u*)(og + 3 * gi); o[k] = make_uint4(v.x, v.y, v.z, v.x); }
      else {
        const zg_v2u v = *(const zg_gv2u*)(og + 2 * gi);
        o[k] = make_uint4(v.x, v.y, v.x, v.y);
        if (OGM == 7) { const zg_v2u w = *(const zg_gv2u*)(og + 2 * ((gi & ~63ull) + (v.x & 63u))); o[k].z = w.x; o[k].w = w.y; }
      }
    }
  };
  uint4 o[ZG_SW_B], onx[ZG_SW_B];
  load_og(b0, o);
  // Software pipeline over the workgroup's batches: the gathers of batch i are issued, then the scratch words of batch i + 1
  // are requested (they stream from HBM while the gathers come back from L2), then batch i is finished.
  for (uint32_t i = 0; i < nb; i++) {
    const uint32_t gb = (b0 + i) * BG;
#pragma unroll
    for (int k = 0; k < ZG_SW_B; k++) if (gb + t + k * ZG_SW_T >= n4) o[k] = make_uint4(0, 0, 0, 0);
    zg_v2u rA[ZG_SW_B], rB[ZG_SW_B], rC[ZG_SW_B], rD[ZG_SW_B], rW[ZG_SW_B];
#pragma unroll
    for (int k = 0; k < ZG_SW_B; k++) {
      const uint4 q = o[k];
      const uint32_t wrel = rel0 + 4u * (gb + t + k * ZG_SW_T);  // resource offset of the group
      const bool ux = q.x != 0, uy = q.y != 0, uz = q.z != 0, uw = q.w != 0;
      const bool any = ux || uy || uz || uw, all = ux && uy && uz && uw;
      const bool nD = uw && !(ux && q.w == q.x);
      const bool nB = uy && !(ux && q.y == q.x) && !(uw && q.y == q.w);
      const bool nC = uz && !(ux && q.z == q.x) && !(uw && q.z == q.w) && !(uy && q.z == q.y);
      rA[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (dbgmode != 2 && ux) ? (wrel - q.x) & ~3u : ZG_OOB, 0, 0);
      rD[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (dbgmode != 2 && dbgmode < 3 && nD) ? (wrel - q.w) & ~3u : ZG_OOB, 0, 0);
      rB[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (dbgmode != 2 && dbgmode < 3 && nB) ? (wrel - q.y) & ~3u : ZG_OOB, 0, 0);
      rC[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (dbgmode != 2 && dbgmode < 3 && nC) ? (wrel - q.z) & ~3u : ZG_OOB, 0, 0);
      rW[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, (dbgmode != 2 && dbgmode < 4 && any && !all) ? wrel & ~3u : ZG_OOB, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);                          // keep the issue order: all gathers, then the next scratch words, then the uses
    load_og(i + 1 < nb ? b0 + i + 1 : b0 + i, onx);            // (the last batch is simply requested again: no branch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < ZG_SW_B; k++) {
      const uint4 q = o[k];
      const uint32_t g = gb + t + k * ZG_SW_T;
      const bool ux = q.x != 0, uy = q.y != 0, uz = q.z != 0, uw = q.w != 0;
      auto fun = [&](const zg_v2u r, uint32_t e) { return __builtin_amdgcn_alignbit(r.y, r.x, ((lowb - e) & 3u) * 8u); };
      const uint32_t lA = fun(rA[k], q.x), lD = fun(rD[k], q.w), lB = fun(rB[k], q.y), lC = fun(rC[k], q.z), lW = fun(rW[k], 0u);
      const uint32_t sw = (ux && q.w == q.x) ? lA : lD;
      const uint32_t sy = (ux && q.y == q.x) ? lA : (uw && q.y == q.w) ? sw : lB;
      const uint32_t sz = (ux && q.z == q.x) ? lA : (uw && q.z == q.w) ? sw : (uy && q.z == q.y) ? sy : lC;
      const uint32_t v = ((ux ? lA : lW) & 0x000000FFu) | ((uy ? sy : lW) & 0x0000FF00u) | ((uz ? sz : lW) & 0x00FF0000u) |
                         ((uw ? sw : lW) & 0xFF000000u);
      // (a group without match bytes, also one behind the unit's end, stores nowhere; no branch: a branch would pull loads into it)
      __builtin_amdgcn_raw_buffer_store_b32(v, rs, (ux || uy || uz || uw) ? rel0 + 4u * g : ZG_OOB, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < ZG_SW_B; k++) o[k] = onx[k];
  }
  // tail bytes of the unit (size not a multiple of four): by the workgroup that would hold their group
  if (part != 2u && (size & 3u) && n4 >= b0 * BG && n4 < (b0 + nbatch) * BG && t < (size & 3u)) {
    const uint32_t x = (n4 << 2) + t;
    const uint32_t e = og[x];
    if (e) out[x] = out[(int64_t)x - (int64_t)e];
  }
}
// the steps of a chain, and the heads beside them: at most three waves per SIMD (a step is a short launch that should find free slots, and
// the heads must not fill the CUs)
template <int OGM>
__global__ void __launch_bounds__(ZG_SW_T) __attribute__((amdgpu_waves_per_eu(2, 3))) zg_k_sweep(ZgBatchDev d, uint32_t list_off, uint32_t nbatch, uint32_t dbgmode, uint32_t part) {
  zg_sweep_body<OGM>(d, list_off, nbatch, dbgmode, part);
}
// (round 5, measured on 64 x 64 MiB frames whose sweep is ONE step of 64 units: eight waves per SIMD instead of three change nothing,
//  10.18 ms either way; four batches per workgroup, software-pipelined, do: 8.95 ms — zg_launch_sweep)

// after the last sweep step: an execution error found by zg_k_flatten becomes the frame's status
__global__ void __launch_bounds__(256) zg_k_fin(ZgBatchDev d) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= d.nframes || d.totals[2]) return;
  if (!d.frame_out[f].fast) return;
  const uint32_t ep = d.frame_out[f].err_packed;
  if (ep != 0xFFFFFFFFu) {
    d.frame_out[f].status = ep & 0xFF;
    d.frame_out[f].bad_block = ep >> 8;
    d.frame_out[f].good_blocks = ep >> 8;
  }
}

// ------------------------------------------------------------------------------------------------------------
// zg_k_lz: in-order execution (execute_sequences, sequence_execution.rs:5-54; DecodeBuffer::push / repeat,
// decode_buffer.rs:74-141) for frames that left the flatten path (a block regenerating more than 128 KiB: not
// conforming). One workgroup walks its frame. Positions are rebuilt here from the exact fields of the records
// (ml, ll): a batch of ZG_LZ_T consecutive sequences is scanned, every lane places its literal run, then the matches
// are resolved in rounds: a match is copied once all of its source bytes lie below the high-water mark (the
// destination of the first match of the batch that is still pending); the first pending match always qualifies.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void zg_lane_match_copy(uint8_t* dst, uint32_t off, uint32_t ml) {
  const uint8_t* src = dst - off;
  uint32_t k = 0;
  if (off >= 8) {
    for (; k + 8 <= ml; k += 8) ((zg_u64u*)(dst + k))->v = zg_ld64(src + k);
  }
  for (; k < ml; k++) dst[k] = src[k];  // also the overlapping case (offset < match length): periodic extension
}

__global__ void __launch_bounds__(ZG_LZ_T) zg_k_lz(ZgBatchDev d) {
  __shared__ uint32_t s_min[ZG_LZ_T / 64];
  __shared__ uint32_t s_so[ZG_LZ_T / 64], s_sl[ZG_LZ_T / 64];
  __shared__ uint32_t s_err;
  __shared__ uint32_t s_errblk;
  if (d.totals[2]) return;
  const uint32_t f = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const ZgFrame fr = d.frames[f];
  const ZgFrameOut fo = d.frame_out[f];
  if (fo.fast) return;
  uint8_t* frame_out = d.dst + fo.out_base;
  if (t == 0) { s_err = 0; s_errblk = 0; }
  __syncthreads();
  for (uint32_t bi = 0; bi < fo.good_blocks; bi++) {
    const uint32_t b = fr.first_block + bi;
    const ZgBlock* blk = &d.blocks[b];
    if (blk->btype != ZG_BT_COMPRESSED || blk->nseq == 0) continue;
    const uint32_t nseq = blk->nseq;
    const ZgBlockPos p = d.pos[b];
    const ZgBlockSeqOut so = d.seq_out[b];
    const ZgSeq* sq = d.seq_arena + blk->seq_base;
    const uint8_t* body = d.src + blk->src_off;
    const bool lit_rle = blk->lit_type == ZG_LT_RLE;
    const uint8_t* lit = blk->lit_type <= ZG_LT_RLE ? body + blk->lit_off : d.lit_arena + blk->lit_base;
    uint32_t carry_out = 0, carry_lit = 0;      // block-relative output position / literal index before the batch
    for (uint32_t s0 = 0; s0 < nseq; s0 += ZG_LZ_T) {
      const uint32_t i = s0 + t;
      bool pending = false;
      uint32_t off = 0, ml = 0, ll = 0, mdst = 0xFFFFFFFFu;
      if (i < nseq) {
        const ZgSeq q = sq[i];
        const uint32_t nx = i + 1 < nseq ? ZG_SEQ_LIT(sq[i + 1]) : so.sum_ll;
        ml = ZG_SEQ_ML(q); ll = (nx - ZG_SEQ_LIT(q)) & 0x1FFFFu;
        off = zg_sym_resolve(q.of, p.hist_init);
      }
      // exclusive scans of ll + ml and ll over the batch
      uint32_t io = ll + ml, il = ll;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t vo = __shfl_up(io, o, 64), vl = __shfl_up(il, o, 64);
        if ((int)lane >= o) { io += vo; il += vl; }
      }
      if (lane == 63) { s_so[wv] = io; s_sl[wv] = il; }
      __syncthreads();
      uint32_t bo = carry_out, bl = carry_lit, to = carry_out, tl = carry_lit;
      for (uint32_t w = 0; w < ZG_LZ_T / 64; w++) { if (w < wv) { bo += s_so[w]; bl += s_sl[w]; } to += s_so[w]; tl += s_sl[w]; }
      const uint32_t lit_start = bl + il - ll;
      uint64_t dpos = 0;  // frame-relative position of the match destination
      if (i < nseq) {
        mdst = bo + io - ml;
        dpos = p.out_base + mdst;
        uint8_t* o = frame_out + dpos - ll;
        if (lit_rle) { const uint8_t v = lit[0]; for (uint32_t k = 0; k < ll; k++) o[k] = v; }
        else { const uint8_t* s = lit + lit_start; for (uint32_t k = 0; k < ll; k++) o[k] = s[k]; }
        if (off == 0) { atomicCAS(&s_err, 0u, (uint32_t)ZG_EXE_ZERO_OFFSET); }
        else if ((uint64_t)off > dpos + fr.prior_reach + fr.dict_len || off >= ZG_OFF_HUGE - 2u) { atomicCAS(&s_err, 0u, (uint32_t)(dpos + fr.prior_out <= fr.window_size ? ZG_EXE_DICT_TOO_SMALL : ZG_EXE_OFFSET_TOO_BIG)); }
        else pending = ml > 0;
      }
      carry_out = to; carry_lit = tl;
      __syncthreads();
      if (s_err) break;
      for (uint32_t guard = 0;; guard++) {                       // (every round retires the first pending match at least: <= ZG_LZ_T rounds)
        if (guard > ZG_LZ_T) { if (t == 0) s_err = ZG_INTERNAL; __syncthreads(); break; }
        // high-water mark = smallest destination among pending matches of this batch
        uint32_t m = pending ? mdst : 0xFFFFFFFFu;
        for (int sh = 32; sh >= 1; sh >>= 1) { uint32_t o = __shfl_xor(m, sh, 64); m = o < m ? o : m; }
        if (lane == 0) s_min[wv] = m;
        __syncthreads();
        uint32_t hwm = s_min[0];
        for (int w = 1; w < ZG_LZ_T / 64; w++) hwm = s_min[w] < hwm ? s_min[w] : hwm;
        if (hwm == 0xFFFFFFFFu) break;
        if (pending) {
          // source bytes that must already exist: [dpos - off, min(dpos - off + ml, dpos))
          const int64_t src_end = (int64_t)dpos - (int64_t)off + (int64_t)ml;   // (<= 0: all of the source lies in front of the frame)
          const uint64_t need_end = ml < off ? (src_end > 0 ? (uint64_t)src_end : 0ull) : dpos;
          if (need_end <= p.out_base + hwm) {
            zg_lane_match_copy(frame_out + dpos, off, ml);
            pending = false;
          }
        }
        __syncthreads();  // makes the copies visible to the other waves of this workgroup (same CU, shared L1)
      }
      __syncthreads();
    }
    if (s_err) { if (t == 0) s_errblk = bi; break; }
    // trailing literals (sequence_execution.rs:40-44)
    {
      const uint32_t rest = blk->regen_size - so.sum_ll;
      uint8_t* o = frame_out + p.out_base + ((uint64_t)so.sum_ll + so.sum_ml);
      if (lit_rle) zg_wg_fill(o, lit[0], rest, t, ZG_LZ_T);
      else zg_wg_copy(o, lit + so.sum_ll, rest, t, ZG_LZ_T);
      __syncthreads();
    }
  }
  __syncthreads();
  if (t == 0 && s_err) {
    d.frame_out[f].status = s_err;
    d.frame_out[f].bad_block = s_errblk;
    d.frame_out[f].good_blocks = s_errblk;
  }
}

// Known-traffic kernels that calibrate the profiler's HBM byte counters per access pattern (tools/dev/profile.sh): the
// engine's kernels read with wide coalesced loads (16 B per lane), narrow coalesced loads (4 B per lane), random 4- and
// 8-byte gathers, and write 4 B and 16 B per lane; FETCH_SIZE / WRITE_SIZE are only documented for the first.
__global__ void __launch_bounds__(256) zg_k_calib_copy(const uint4* src, uint4* dst, uint64_t n16) {        // 16 B per lane, read + write
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) zg_k_calib_copy4(const uint32_t* src, uint32_t* dst, uint64_t n4) {   // 4 B per lane, read + write
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}
// n random reads of W bytes each (W = 4 or 8, W-aligned) spread over `span` bytes (much more than all caches): every one a miss
template <typename T>
__global__ void __launch_bounds__(256) zg_k_calib_gather(const T* src, uint32_t* sink, uint64_t span_elems, uint64_t n) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    const T v = src[z % span_elems];
    acc ^= (uint32_t)v ^ (uint32_t)((uint64_t)v >> 32);
  }
  if (acc == 0x12345678u) sink[0] = acc;       // keeps the loads alive
}
void zg_launch_calib(const void* src, void* dst, uint64_t bytes, hipStream_t s) {
  hipLaunchKernelGGL(zg_k_calib_copy, dim3(2048), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, bytes / 16);
  hipLaunchKernelGGL(zg_k_calib_copy4, dim3(2048), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, bytes / 4);
  hipLaunchKernelGGL(zg_k_calib_gather<uint32_t>, dim3(2048), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, bytes / 4, bytes / 64);
  hipLaunchKernelGGL(zg_k_calib_gather<uint64_t>, dim3(2048), dim3(256), 0, s, (const uint64_t*)src, (uint32_t*)dst, bytes / 8, bytes / 64);
}

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
void zg_launch_tables(const ZgBatchDev& d, hipStream_t s, int part) {
  uint32_t n = d.nblocks + 1;
  if (part == 0) hipLaunchKernelGGL(zg_k_tables, dim3((n + ZG_HT_W - 1) / ZG_HT_W), dim3(64 * ZG_HT_W), 0, s, d);
  else {
    hipLaunchKernelGGL(zg_k_fparse, dim3((d.nblocks + 63) / 64), dim3(64), 0, s, d);
    hipLaunchKernelGGL(zg_k_ftab, dim3((n + ZG_FT_W - 1) / ZG_FT_W), dim3(64 * ZG_FT_W), 0, s, d);
  }
}
void zg_launch_huf(const ZgBatchDev& d, hipStream_t s) {
  if (!d.nhuf_groups) return;
  hipLaunchKernelGGL(zg_k_huf, dim3(d.nhuf_groups), dim3(ZG_HUF_T), 0, s, d);
  hipLaunchKernelGGL(zg_k_huf_uneven, dim3(d.nblocks), dim3(64), 0, s, d);
}
// packed: the 16-bit-entry form (three workgroups per CU): for submits with more blocks than the device holds chains at once
void zg_launch_seq(const ZgBatchDev& d, hipStream_t s, bool packed) {
  if (!d.nseq_blocks) return;
  if (packed) hipLaunchKernelGGL(zg_k_seq<true>, dim3((d.nseq_blocks + ZG_SEQ_G - 1) / ZG_SEQ_G), dim3(128), 0, s, d);
  else hipLaunchKernelGGL(zg_k_seq<false>, dim3((d.nseq_blocks + ZG_SEQ_G - 1) / ZG_SEQ_G), dim3(128), 0, s, d);
}
// The literals chain (Huffman tree descriptions, zg_k_huf) ran beside the sequences chain on a second stream: fold its
// errors into the block status. The literals section is decoded first (block_decoder.rs:131-150), so its errors —
// the tree description's, then the streams' — outrank those of the sequences section.
__global__ void __launch_bounds__(256) zg_k_merge(ZgBatchDev d) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= d.nblocks) return;
  if (d.tab_status[b]) d.status[b] = d.tab_status[b];
  else if (d.lit_status[b]) d.status[b] = d.lit_status[b] & 0xFFu;
  else if (d.blocks[b].seq_host_status && !d.status[b]) d.status[b] = d.blocks[b].seq_host_status;   // the host found it, but it comes behind the literals
}
void zg_launch_merge(const ZgBatchDev& d, hipStream_t s) {
  if (d.nblocks) hipLaunchKernelGGL(zg_k_merge, dim3((d.nblocks + 255) / 256), dim3(256), 0, s, d);
}
// ZG_FLAG_LIT_DIRECT: the literal streams were decoded AFTER the scan had laid the frames out, so a literals error is found
// late. One workgroup per frame: the first block (in frame order) whose literals failed — in front of, or at, the block the frame
// stopped at so far: a block's literals are decoded before its sequences (block_decoder.rs:131-150), so their error outranks a
// sequences error of the same block, but not a header or tree-description error — becomes the frame's failing block; it and
// everything behind it is not executed. (zg_k_merge, launched again in front of this kernel, has folded the literal verdicts into
// the block statuses.)
__global__ void __launch_bounds__(256) zg_k_litfix(ZgBatchDev d) {
  __shared__ uint32_t s_first;
  const uint32_t f = blockIdx.x, t = threadIdx.x;
  const ZgFrame fr = d.frames[f];
  const ZgFrameOut fo = d.frame_out[f];
  if (t == 0) s_first = 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t lim = fo.good_blocks < fr.nblocks ? fo.good_blocks + 1u : fr.nblocks;
  for (uint32_t i = t; i < lim; i += 256) {
    const uint32_t b = fr.first_block + i;
    if (d.lit_status[b] && !d.blocks[b].host_status && !d.tab_status[b]) atomicMin(&s_first, i);
  }
  __syncthreads();
  const uint32_t first = s_first;
  if (first == 0xFFFFFFFFu) return;
  for (uint32_t i = first + t; i < fr.nblocks; i += 256) d.pos[fr.first_block + i].active = 0u;
  if (t == 0) {
    ZgFrameOut* o = &d.frame_out[f];
    o->status = d.lit_status[fr.first_block + first] & 0xFFu;
    o->bad_block = first; o->good_blocks = first;
  }
}
void zg_launch_litfix(const ZgBatchDev& d, hipStream_t s) {
  zg_launch_merge(d, s);
  if (d.nframes) hipLaunchKernelGGL(zg_k_litfix, dim3(d.nframes), dim3(256), 0, s, d);
}
void zg_launch_seqpost(const ZgBatchDev& d, hipStream_t s) {
  if (d.nseq_blocks) hipLaunchKernelGGL(zg_k_seqpost, dim3(d.nseq_blocks), dim3(ZG_SP_T), 0, s, d);
}
void zg_launch_scan(const ZgBatchDev& d, hipStream_t s, uint32_t max_frame_blocks) {
  if (max_frame_blocks <= 64u) hipLaunchKernelGGL((zg_k_scan<64, 1>), dim3(d.nframes), dim3(64), 0, s, d);
  else if (max_frame_blocks <= 1024u) hipLaunchKernelGGL((zg_k_scan<1024, 1>), dim3(d.nframes), dim3(1024), 0, s, d);
  else hipLaunchKernelGGL((zg_k_scan<1024, 8>), dim3(d.nframes), dim3(1024), 0, s, d);
  if (d.nframes != 1u) hipLaunchKernelGGL(zg_k_scanf, dim3(1), dim3(1024), 0, s, d);
}
void zg_launch_lit(const ZgBatchDev& d, hipStream_t s) {
  if (d.nblocks) hipLaunchKernelGGL(zg_k_lit, dim3(d.nblocks), dim3(256), 0, s, d);
}
// ------------------------------------------------------------------------------------------------------------
// zg_k_sparse: the matches of a frame that has hardly any (literal-heavy data: a sequence or two in one block out of twenty).
// zg_k_flatten has placed the literals and checked the offsets; what is left is a few hundred short copies per frame, which one
// wave does in order (a match may copy from an earlier one) in the time of a few sweep launches — of which the frame would
// need one per unit. One wave per frame, 64 sequences at a time, the same "copy what no pending match can still write"
// rule as zg_k_lz.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) zg_k_sparse(ZgBatchDev d) {
  if (d.totals[2]) return;
  const uint32_t f = blockIdx.x, lane = threadIdx.x;
  const ZgFrame fr = d.frames[f];
  if (!fr.sparse) return;
  const ZgFrameOut fo = d.frame_out[f];
  if (!fo.fast) return;                                        // the in-order path has it
  const uint32_t stop = fo.err_packed == 0xFFFFFFFFu ? 0xFFFFFFFFu : fo.err_packed >> 8;   // zg_k_flatten found a sequence that cannot be executed: the blocks in front of its block are
  uint8_t* frame_out = d.dst + fo.out_base;
  for (uint32_t e = 0; e < fr.seq_count; e++) {
    const uint32_t b = d.seq_blocks[fr.seq_first + e];
    const ZgBlockPos p = d.pos[b];
    if (!p.active || b - fr.first_block >= stop) break;
    const ZgBlock* blk = &d.blocks[b];
    const uint32_t nseq = blk->nseq;
    const ZgSeq* sq = d.seq_arena + blk->seq_base;
    for (uint32_t s0 = 0; s0 < nseq; s0 += 64) {
      const uint32_t i = s0 + lane;
      bool pending = false;
      uint32_t off = 0, ml = 0, mdst = 0xFFFFFFFFu;
      uint64_t dpos = 0;
      if (i < nseq) {
        const ZgSeq q = sq[i];
        ml = ZG_SEQ_ML(q); mdst = ZG_SEQ_MDST(q);
        off = zg_sym_resolve(q.of, p.hist_init);
        dpos = p.out_base + mdst;
        pending = ml > 0;
      }
      for (uint32_t guard = 0;; guard++) {                       // (every round retires the first pending match at least: <= 64 rounds)
        if (guard > 64u) { if (lane == 0) d.frame_out[f].status = ZG_INTERNAL; return; }
        uint32_t hwm = pending ? mdst : 0xFFFFFFFFu;            // the lowest destination a pending match of the batch still has to write
        for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = __shfl_xor(hwm, sh, 64); hwm = o < hwm ? o : hwm; }
        if (hwm == 0xFFFFFFFFu) break;
        if (pending) {
          // source bytes that must exist: [dpos - off, need_end). What lies in front of the frame (dictionary, earlier submits) exists.
          const int64_t src_end = (int64_t)dpos - (int64_t)off + (int64_t)ml;
          const uint64_t need_end = ml < off ? (src_end > 0 ? (uint64_t)src_end : 0ull) : dpos;
          if (need_end <= p.out_base + hwm) { zg_lane_match_copy(frame_out + dpos, off, ml); pending = false; }
        }
        __threadfence_block();                                   // the copies are visible to the lanes that copy from them next
      }
    }
  }
}
// ------------------------------------------------------------------------------------------------------------
// zg_k_partial: what the reference's decode buffer holds of a block whose sequence EXECUTION failed. execute_sequences
// (sequence_execution.rs:6-52) pushes a sequence's literals, then its match, one sequence after the other, and returns at the first one it
// cannot execute: the output of the sequences in front of it stays in the buffer (and that sequence's literals, unless it was the
// literals that ran out), where collect() / read() still find it after the Err. The fast path produces a block as a whole or not at
// all; for the ONE block that failed, of a frame that is decoded run by run (FrameDecoder mirror, thin boundary, streaming decoder),
// Batch::sync() runs this kernel: sequences [0, nexec) in order behind the bytes of the good blocks, plus the literals of sequence
// nexec when lits_of_next is set. One wave, 64 sequences per round (zg_k_sparse's scheme: a match is copied once everything below the
// lowest pending destination is final); an error path, at most 128 KiB.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) zg_k_partial(ZgBatchDev d, uint32_t f, uint32_t b, uint32_t nexec, uint32_t lits_of_next, uint32_t limit) {
  const uint32_t lane = threadIdx.x;
  const ZgFrameOut fo = d.frame_out[f];
  const ZgBlockPos p = d.pos[b];
  const ZgBlock* blk = &d.blocks[b];
  uint8_t* frame_out = d.dst + fo.out_base;
  const ZgSeq* sq = d.seq_arena + blk->seq_base;
  const uint8_t* body = d.src + blk->src_off;
  const bool lit_rle = blk->lit_type == ZG_LT_RLE;
  const uint8_t* lit = blk->lit_type <= ZG_LT_RLE ? body + blk->lit_off : d.lit_arena + blk->lit_base;
  const uint32_t total = nexec + (lits_of_next ? 1u : 0u);          // sequences whose literals go out
  // Positions are rebuilt from the exact fields of the records, as in zg_k_lz (a block beyond 128 KiB wraps the position fields): the
  // literal run of sequence i is what lies between the end of sequence i - 1 and its match, (mdst_i - mdst_(i-1) - ml_(i-1)) mod 2^17.
  // pass 0 only measures (the host has reserved `limit` bytes behind the run: more than that is not written at all), pass 1 executes.
  for (int pass = 0; pass < 2; pass++) {
    uint32_t carry = 0;                                              // block-relative end of the sequences in front of the batch
    for (uint32_t s0 = 0; s0 < total; s0 += 64) {
      const uint32_t i = s0 + lane;
      bool pending = false;
      uint32_t off = 0, ml = 0, ll = 0, lp = 0;
      if (i < total) {
        const ZgSeq q = sq[i];
        uint32_t prev_end = 0;
        if (i) { const ZgSeq pq = sq[i - 1]; prev_end = ZG_SEQ_MDST(pq) + ZG_SEQ_ML(pq); }
        ll = (ZG_SEQ_MDST(q) - prev_end) & 0x1FFFFu;
        lp = ZG_SEQ_LIT(q);
        if (i < nexec) { ml = ZG_SEQ_ML(q); off = zg_sym_resolve(q.of, p.hist_init); }
      }
      uint32_t io = ll + ml;                                         // inclusive scan over the batch
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(io, o, 64); if ((int)lane >= o) io += v; }
      const uint32_t start = carry + io - (ll + ml);                 // where this sequence's literals start
      const uint32_t mdst = start + ll;                              // ... and its match
      carry += __shfl(io, 63, 64);
      if (pass == 0) continue;
      uint64_t dpos = 0;
      if (i < total) {
        uint8_t* o = frame_out + p.out_base + start;
        if (lit_rle) { const uint8_t v = lit[0]; for (uint32_t k = 0; k < ll; k++) o[k] = v; }
        else for (uint32_t k = 0; k < ll; k++) o[k] = lit[lp + k];
        dpos = p.out_base + mdst;
        pending = i < nexec && ml > 0;
      }
      __threadfence_block();                                         // the literals are in place for the matches that copy from them
      for (uint32_t guard = 0; guard <= 64u; guard++) {              // (every round retires the first pending match at least)
        uint32_t hwm = pending ? mdst : 0xFFFFFFFFu;
        for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o2 = __shfl_xor(hwm, sh, 64); hwm = o2 < hwm ? o2 : hwm; }
        if (hwm == 0xFFFFFFFFu) break;
        if (pending) {
          const int64_t src_end = (int64_t)dpos - (int64_t)off + (int64_t)ml;
          const uint64_t need_end = ml < off ? (src_end > 0 ? (uint64_t)src_end : 0ull) : dpos;
          if (need_end <= p.out_base + hwm) { zg_lane_match_copy(frame_out + dpos, off, ml); pending = false; }
        }
        __threadfence_block();
      }
    }
    if (pass == 0) {
      if (lane == 0) d.totals[5] = carry <= limit ? carry : 0xFFFFFFFFu;   // what the block leaves behind; 0xFFFFFFFF: more than was reserved, nothing written
      if (carry > limit) return;
    }
  }
}
void zg_launch_partial(const ZgBatchDev& d, hipStream_t s, uint32_t frame, uint32_t block, uint32_t nexec, bool lits_of_next, uint32_t limit) {
  hipLaunchKernelGGL(zg_k_partial, dim3(1), dim3(64), 0, s, d, frame, block, nexec, lits_of_next ? 1u : 0u, limit);
}
void zg_launch_sparse(const ZgBatchDev& d, hipStream_t s) {
  if (d.nframes) hipLaunchKernelGGL(zg_k_sparse, dim3(d.nframes), dim3(64), 0, s, d);
}
// flat4: how the direct units are flattened (zg::Tuning::flat4) — 0: by zg_k_flatten itself; 1 / 2: by zg_k_flatten4 (8 / 16 KiB tiles), on s2 beside
// the pointer-mode units when the submit has both (ev[0]: fork, ev[1]: join)
void zg_launch_flat(const ZgBatchDev& d, hipStream_t s, hipStream_t s2, hipEvent_t* ev, int flat4) {
  if (!d.nunits) return;
  const uint32_t shape = (d.flags >> 2) & 3u;
  if (shape == 1) { hipLaunchKernelGGL((zg_k_flatten<512, 8192, 2>), dim3(d.nunits), dim3(512), 0, s, d, 1u); return; }
  if (flat4 == 0 || d.ndirect == 0 || d.overlap_epoch) { hipLaunchKernelGGL((zg_k_flatten<1024, 16384, 2>), dim3(d.nunits), dim3(1024), 0, s, d, 1u); return; }
  const uint32_t nptr = d.nunits - d.ndirect;
  hipStream_t sd = s;
  if (nptr) { (void)hipEventRecord(ev[0], s); (void)hipStreamWaitEvent(s2, ev[0], 0); sd = s2; }
  if (flat4 == 2) hipLaunchKernelGGL((zg_k_flatten4<1024, 16384, 2, 4>), dim3(d.ndirect), dim3(1024), 0, sd, d);
  else if (flat4 == 3) hipLaunchKernelGGL((zg_k_flatten4<512, 8192, 2, 4>), dim3(d.ndirect), dim3(512), 0, sd, d);      // (measurement) two 512-thread workgroups per CU
  else if (flat4 == 4) hipLaunchKernelGGL((zg_k_flatten4<512, 4096, 1, 8>), dim3(d.ndirect), dim3(512), 0, sd, d);      // (measurement) four
  else if (flat4 == 5) hipLaunchKernelGGL((zg_k_flatten4<256, 4096, 2, 8>), dim3(d.ndirect), dim3(256), 0, sd, d);
  else if (flat4 == 6) hipLaunchKernelGGL((zg_k_flatten4<256, 2048, 1, 8>), dim3(d.ndirect), dim3(256), 0, sd, d);
  else if (flat4 == 7) hipLaunchKernelGGL((zg_k_flatten4<512, 2048, 1, 8>), dim3(d.ndirect), dim3(512), 0, sd, d);
  else hipLaunchKernelGGL((zg_k_flatten4<1024, 8192, 1, 8>), dim3(d.ndirect), dim3(1024), 0, sd, d);
  if (nptr) {
    hipLaunchKernelGGL((zg_k_flatten<1024, 16384, 2>), dim3(nptr), dim3(1024), 0, s, d, 0u);
    (void)hipEventRecord(ev[1], s2); (void)hipStreamWaitEvent(s, ev[1], 0);
  }
}
bool zg_launch_sweep(const ZgBatchDev& d, hipStream_t s, const ZgSweepStep* steps, uint32_t nsteps, hipStream_t s2, hipEvent_t* evs, uint32_t nev,
                     uint32_t unit_bytes, uint32_t window_max, uint32_t window_min, const ZgSweepTuning& tn) {
  uint32_t dbgmode = tn.mode;                                 // timing experiments only (ZGPU_SWEEP_MODE)
  void (*kern)(ZgBatchDev, uint32_t, uint32_t, uint32_t, uint32_t) = zg_k_sweep<0>;
#ifdef ZG_DEV_SWITCHES
  if (dbgmode >= 5u) { kern = dbgmode == 5u ? zg_k_sweep<5> : dbgmode == 6u ? zg_k_sweep<6> : dbgmode == 7u ? zg_k_sweep<7> : zg_k_sweep<8>; dbgmode = 0u; }
#else
  dbgmode = 0u;
#endif
  // batches per workgroup, software-pipelined (the scratch words of batch i + 1 are requested behind the gathers of batch i): in a chain of
  // steps more than one did not pay (a step is over when its slowest workgroup is); a sweep that is ONE large launch gains 12 % with four
  const uint32_t nbatch = tn.nbatch ? tn.nbatch : (nsteps == 1 ? 4u : 1u);
  constexpr uint32_t BB = 4u * ZG_SW_T * ZG_SW_B;             // bytes per batch
  uint32_t n = 0;
  bool contiguous = true;
  for (uint32_t i = 0; i < nsteps; i++) { if (steps[i].list_off != steps[0].list_off + n) contiguous = false; n += steps[i].nunits; }
  if (n && !d.overlap_epoch) hipLaunchKernelGGL(zg_k_swprep, dim3((n + 255) / 256), dim3(256), 0, s, d, n);   // (beside the flatten: every unit's flatten writes its own descriptor)
  // The tails are worth a chain of their own when they are clearly shorter than the units (else: the plain chain).
  const uint32_t tail_batches = window_max / BB + 2u;
  const bool split = nev >= 3 && contiguous && nsteps > 1 && (uint64_t)tail_batches * BB + 65536u < unit_bytes;
  if (!split) {
    for (uint32_t i = 0; i < nsteps; i++)
      hipLaunchKernelGGL(kern, dim3((steps[i].slices + nbatch - 1) / nbatch, steps[i].nunits), dim3(ZG_SW_T), 0, s, d, steps[i].list_off, nbatch, dbgmode, 0u);
  } else {
    // stream s: tail of step 0, 1, 2, ...; stream s2: the heads of steps [g0, g1) in one launch as soon as the tails of step g1 - 2
    // are done (the heads of step i copy from the tails of steps < i).
    const uint32_t gmin = tn.group ? tn.group : 16u;          // steps whose heads share a launch
    const uint32_t gs = (nsteps + (nev - 2) - 1) / (nev - 2) > gmin ? (nsteps + (nev - 2) - 1) / (nev - 2) : gmin;
    uint32_t g0 = 0, e = 0;
    const uint32_t head_lds = tn.head_lds;
    auto heads = [&]() {
      const uint32_t g1 = g0 + gs < nsteps ? g0 + gs : nsteps;
      uint32_t units = 0, slices = 0;
      for (uint32_t i = g0; i < g1; i++) { units += steps[i].nunits; slices = steps[i].slices > slices ? steps[i].slices : slices; }
      slices = slices > window_min / BB ? slices - window_min / BB : 1u;   // a head is its unit without the last window (workgroups beyond it would only come and go)
      (void)hipEventRecord(evs[e], s);
      (void)hipStreamWaitEvent(s2, evs[e], 0);
      e++;
      // (the heads are bulk work beside a chain of short launches: an unused LDS allocation keeps them to a few workgroups per
      //  CU, so that a tail step always finds free wave slots instead of waiting for head workgroups to retire)
      const uint32_t hnb = tn.head_nbatch ? tn.head_nbatch : 1u;
      hipLaunchKernelGGL(kern, dim3((slices + hnb - 1) / hnb, units), dim3(ZG_SW_T), head_lds, s2, d, steps[g0].list_off, hnb, dbgmode, 2u);
      g0 = g1;
    };
    auto need = [&]() { return (int64_t)(g0 + gs < nsteps ? g0 + gs : nsteps) - 2; };   // the last tail step the next group of heads waits for
    while (g0 < nsteps && need() < 0) heads();
    for (uint32_t i = 0; i < nsteps; i++) {
      hipLaunchKernelGGL(kern, dim3((tail_batches + nbatch - 1) / nbatch + 1u, steps[i].nunits), dim3(ZG_SW_T), 0, s, d, steps[i].list_off, nbatch, dbgmode, 1u);
      while (g0 < nsteps && need() <= (int64_t)i) heads();
    }
    (void)hipEventRecord(evs[e], s2);
    (void)hipStreamWaitEvent(s, evs[e], 0);
  }
  if (d.nframes) hipLaunchKernelGGL(zg_k_fin, dim3((d.nframes + 255) / 256), dim3(256), 0, s, d);
  return split;
}
void zg_launch_exact(const ZgBatchDev& d, hipStream_t s, uint32_t drain_rule) {
  if (d.nframes) hipLaunchKernelGGL(zg_k_exact, dim3(d.nframes), dim3(256), 0, s, d, drain_rule);
}
void zg_launch_lz(const ZgBatchDev& d, hipStream_t s) {
  hipLaunchKernelGGL(zg_k_lz, dim3(d.nframes), dim3(ZG_LZ_T), 0, s, d);
}
