// zg_kernels.h — launch interface of the gfx950 kernels (implemented in zg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zg_types.h"

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
// measurement switches of the sweep (tools/dev/README.md): read from the environment ONCE, when an engine is created (zg::Tuning) —
// mode: ZGPU_SWEEP_MODE (timing experiments: wrong results), nbatch: batches per workgroup, group: steps whose heads share a launch,
// head_lds: unused LDS a head workgroup asks for (keeps the heads to a few workgroups per CU)
struct ZgSweepTuning { uint32_t mode = 0, nbatch = 0, group = 16, head_lds = 52u * 1024u, head_nbatch = 0; };   // nbatch 0: chosen by zg_launch_sweep

void zg_launch_tables(const ZgBatchDev& d, hipStream_t s, int part);   // part 0: Huffman trees, part 1: FSE tables
void zg_launch_huf(const ZgBatchDev& d, hipStream_t s);
void zg_launch_seq(const ZgBatchDev& d, hipStream_t s, bool packed);   // packed: 16-bit table entries, three workgroups per CU (submits of more blocks than one round holds)
void zg_launch_seqpost(const ZgBatchDev& d, hipStream_t s);
void zg_launch_merge(const ZgBatchDev& d, hipStream_t s);
void zg_launch_litfix(const ZgBatchDev& d, hipStream_t s);   // ZG_FLAG_LIT_DIRECT: literal verdicts found after the scan -> block and frame statuses
void zg_launch_scan(const ZgBatchDev& d, hipStream_t s, uint32_t max_frame_blocks);   // max_frame_blocks: of the submit's frames (picks the workgroup size)
void zg_launch_lit(const ZgBatchDev& d, hipStream_t s);
void zg_launch_flat(const ZgBatchDev& d, hipStream_t s, hipStream_t s2, hipEvent_t* ev, int flat4);   // s2, ev[2]: the direct units beside the pointer-mode ones
void zg_launch_sparse(const ZgBatchDev& d, hipStream_t s);   // the matches of frames marked sparse, in order (after zg_launch_flat)
bool zg_launch_sweep(const ZgBatchDev& d, hipStream_t s, const ZgSweepStep* steps, uint32_t nsteps, hipStream_t s2, hipEvent_t* evs, uint32_t nev,
                     uint32_t unit_bytes, uint32_t window_max, uint32_t window_min, const ZgSweepTuning& tn);   // s2 / evs: the side stream of the split sweep (nev == 0: one stream, step by step); returns whether it split
void zg_launch_lz(const ZgBatchDev& d, hipStream_t s);
void zg_launch_partial(const ZgBatchDev& d, hipStream_t s, uint32_t frame, uint32_t block, uint32_t nexec, bool lits_of_next, uint32_t limit);   // what the reference's buffer holds of a block whose sequence execution failed (Batch::sync, runs of one frame)
void zg_launch_exact(const ZgBatchDev& d, hipStream_t s, uint32_t drain_rule);   // zg_exact.h: the reference's DecodeBuffer bookkeeping, exactly (rare path, Batch::sync)
void zg_launch_calib(const void* src, void* dst, uint64_t bytes, hipStream_t s);
