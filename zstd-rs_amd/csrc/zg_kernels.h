// zg_kernels.h — launch interface of the gfx950 kernels (implemented in zg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zg_types.h"

// Device-side view of one submit (all pointers are device pointers).
struct ZgBatchDev {
  const uint8_t* src;          // compressed bytes of the whole submit (padded by >= 16 bytes at the end)
  uint64_t src_len;
  const ZgBlock* blocks;
  uint32_t nblocks;
  const ZgFrame* frames;
  uint32_t nframes;
  uint32_t nslots;             // FSE arena slots = nblocks + 1 (predefined) + nframes (carry)
  uint32_t nhuf_slots;         // Huffman arena slots
  ZgBlockAux* aux;             // [nblocks]
  uint8_t* slot_log;           // [nslots][4]: accuracy logs LL, OF, ML of the tables held by each FSE slot
  uint32_t* fse_arena;         // [nslots][ZG_FSE_SLOT_U32]
  uint16_t* huf_arena;         // [nhuf_slots][ZG_HUF_SLOT_U16]
  uint8_t* huf_maxbits;        // [nhuf_slots]
  uint32_t* status;            // [nblocks] first error per block (ZgStatus), 0 = ok
  uint32_t* tab_status;        // [nblocks] what zg_k_tables left in status (zg_k_huf runs beside zg_k_seq and must not see its errors)
  uint32_t* lit_status;        // [nblocks] zg_k_huf's errors, folded into status by zg_k_merge (literals are decoded before sequences: they outrank)
  uint8_t* lit_arena;          // regenerated Huffman literals
  ZgSeq* seq_arena;            // decoded sequences
  uint2* raw_arena;            // zg_k_seq's raw records, 4 x u16 {OF, ML, LL table entries, bits taken}, same indexing as seq_arena
  ZgBlockSeqOut* seq_out;      // [nblocks]
  ZgBlockPos* pos;             // [nblocks]
  ZgFrameOut* frame_out;       // [nframes]
  uint8_t* dst;                // decompressed output of the submit (frames back to back)
  uint64_t dst_cap;
  const uint8_t* dict;         // dictionary contents (may be null)
  // work lists
  const uint32_t* seq_blocks;  // compressed blocks with nseq > 0
  uint32_t nseq_blocks;
  const uint32_t* huf_items;   // (block << 2) | stream
  const ZgHufGroup* huf_groups;
  uint32_t nhuf_groups;
  uint32_t* totals;            // [4]: [0..1] total output bytes (u64), [2] overflow flag, [3] a match reaches further back than its frame's window (zg_k_seqpost)
  uint32_t sweep_window;       // 0: a frame's window size bounds its matches (checked); else this many bytes instead (tests)
  uint32_t flags;              // bit 0: force the in-order fallback for every frame (tests); bits 2-3: shape of zg_k_flat (0: 1024 threads x 16 KiB tiles, 1: 512 x 8 KiB)
  uint64_t og_words;           // size of the flatten scratch in u32
  uint32_t* og;                // flatten scratch: one u32 "effective offset" per output byte of a unit (0 = literal byte, final already)
  const ZgUnit* units;
  uint32_t nunits;
  ZgUnitInfo* unit_info;       // [nunits]
  ZgSweepDesc* sweep_desc;     // one per entry of step_units, written by zg_k_swprep after zg_k_flat
  const uint32_t* step_units;  // sweep step s fills the units step_units[list_off(s) ...] (unit s of every frame that has one)
  unsigned long long* dbg;     // phase cycle counters (profiling builds), diagnostics only
};
// one launch of zg_k_sweep
// zg_k_sweep: threads per workgroup, groups of 4 output bytes a thread has in flight; a workgroup takes ZG_SW_BATCH bytes of a unit.
// (128 or 256 threads with 2 or 4 groups each measure the same; one group per thread, or several batches per workgroup, are
// slower. What matters more is that a launch has no workgroups that only come and go: see zg_launch_sweep.)
#ifndef ZG_SW_T
#define ZG_SW_T 256
#endif
#ifndef ZG_SW_B
#define ZG_SW_B 2
#endif
#define ZG_SW_BATCH (4u * ZG_SW_T * ZG_SW_B)
struct ZgSweepStep { uint32_t list_off, nunits, slices, pad; };

void zg_launch_tables(const ZgBatchDev& d, hipStream_t s, int part);   // part 0: Huffman trees, part 1: FSE tables
void zg_launch_huf(const ZgBatchDev& d, hipStream_t s);
void zg_launch_seq(const ZgBatchDev& d, hipStream_t s);
void zg_launch_seqpost(const ZgBatchDev& d, hipStream_t s);
void zg_launch_merge(const ZgBatchDev& d, hipStream_t s);
void zg_launch_scan(const ZgBatchDev& d, hipStream_t s);
void zg_launch_lit(const ZgBatchDev& d, hipStream_t s);
void zg_launch_flat(const ZgBatchDev& d, hipStream_t s);
void zg_launch_sparse(const ZgBatchDev& d, hipStream_t s);   // the matches of frames marked sparse, in order (after zg_launch_flat)
bool zg_launch_sweep(const ZgBatchDev& d, hipStream_t s, const ZgSweepStep* steps, uint32_t nsteps, hipStream_t s2, hipEvent_t* evs, uint32_t nev,
                     uint32_t unit_bytes, uint32_t window_max, uint32_t window_min);   // s2 / evs: the side stream of the split sweep (nev == 0: one stream, step by step); returns whether it split
void zg_launch_lz(const ZgBatchDev& d, hipStream_t s);
void zg_launch_calib(const void* src, void* dst, uint64_t bytes, hipStream_t s);
