// zg_types.h — plain structs shared by the host parser, the engine and the HIP kernels.
//
// Vocabulary follows the reference (ruzstd): frames, blocks, literals section, sequences section,
// FSE tables (LL/OF/ML), Huffman table, offset history, decode window.
#pragma once
#include <stdint.h>

// Status codes. Each maps onto a leaf of the reference's error enums (ruzstd/src/decoding/errors.rs);
// the C ABI (include/zgpu.h) re-exports them as ZGPU_E_*.
enum ZgStatus : int32_t {
  ZG_OK = 0,
  // frame layer — FrameDecoderError / ReadFrameHeaderError / FrameHeaderError
  ZG_SKIP_FRAME = 1,
  ZG_BAD_MAGIC = 2,
  ZG_HEADER_READ = 3,
  ZG_WINDOW_TOO_BIG_SPEC = 4,
  ZG_WINDOW_TOO_SMALL = 5,
  ZG_WINDOW_SIZE_TOO_BIG = 6,
  ZG_DICT_NOT_PROVIDED = 7,
  ZG_NOT_INITIALIZED = 8,
  ZG_FAILED_READ_BLOCK_HEADER = 9,
  ZG_FAILED_READ_BLOCK_BODY = 10,
  ZG_FAILED_READ_CHECKSUM = 11,
  ZG_TARGET_TOO_SMALL = 12,
  ZG_FAILED_SKIP_FRAME = 13,
  // block layer — BlockHeaderReadError / DecompressBlockError
  ZG_RESERVED_BLOCK = 20,
  ZG_BLOCK_SIZE_TOO_LARGE = 21,
  ZG_MALFORMED_SECTION_HEADER = 22,
  ZG_LITERALS_HEADER = 23,
  ZG_SEQUENCES_HEADER = 24,
  // literals — DecompressLiteralsError / HuffmanTableError
  ZG_LIT_UNINIT_HUF = 30,
  ZG_LIT_MISSING_JUMP = 31,
  ZG_LIT_MISSING_BYTES = 32,
  ZG_LIT_EXTRA_PADDING = 33,
  ZG_LIT_BITSTREAM_MISMATCH = 34,
  ZG_LIT_COUNT_MISMATCH = 35,
  ZG_HUF_TABLE = 36,
  // sequences — DecodeSequenceError / FSETableError / FSEDecoderError
  ZG_FSE_TABLE = 40,
  ZG_FSE_UNINIT = 41,
  ZG_SEQ_MISSING_MODE = 42,
  ZG_SEQ_RLE_BYTE = 43,
  ZG_SEQ_EXTRA_PADDING = 44,
  ZG_SEQ_UNSUPPORTED_OFFSET = 45,
  ZG_SEQ_NOT_ENOUGH_BYTES = 46,
  ZG_SEQ_EXTRA_BITS = 47,
  // execution — ExecuteSequencesError / DecodeBufferError
  ZG_EXE_NOT_ENOUGH_LITERALS = 50,
  ZG_EXE_ZERO_OFFSET = 51,
  ZG_EXE_OFFSET_TOO_BIG = 52,
  ZG_EXE_DICT_TOO_SMALL = 53,
  ZG_DICT_DECODE = 60,
  // engine limits (inputs the reference would accept or panic on; documented in DESIGN.md)
  ZG_UNSUPPORTED = 80,         // e.g. uneven 4-stream Huffman split, block regenerating > 2^31 bytes
  ZG_INTERNAL = 90,            // where the reference would panic/assert
  ZG_NOMEM = 91,
  ZG_HIP_ERROR = 92,
  ZG_BAD_ARG = 93
};

enum { ZG_BT_RAW = 0, ZG_BT_RLE = 1, ZG_BT_COMPRESSED = 2 };
enum { ZG_LT_RAW = 0, ZG_LT_RLE = 1, ZG_LT_COMPRESSED = 2, ZG_LT_TREELESS = 3 };
enum { ZG_MODE_PREDEFINED = 0, ZG_MODE_RLE = 1, ZG_MODE_FSE = 2, ZG_MODE_REPEAT = 3 };

// FSE table arena: one slot per block of the batch + 1 predefined slot + 1 carry slot per frame.
// Slot layout in u32 entries: [LL 512][ML 512][OF 256].
#define ZG_FSE_SLOT_U32 1280
#define ZG_FSE_LL_OFF 0
#define ZG_FSE_ML_OFF 512
#define ZG_FSE_OF_OFF 1024
// Huffman table arena: one slot of 2048 u16 per table.
#define ZG_HUF_SLOT_U16 2048
#define ZG_HUF_GROUP 2           // streams (one wave each) per zg_k_huf workgroup; two keep its LDS small enough to run beside zg_k_seq
#define ZG_REF_UNINIT (-1)

// Packed FSE decode entry (u32): [0,16) next-state base_line, [16,20) num_bits, [20,26) symbol (code),
// [26,31) number of extra bits that code reads from the stream.
#define ZG_FSE_PACK(bl, nb, sym, xb) ((uint32_t)(bl) | ((uint32_t)(nb) << 16) | ((uint32_t)(sym) << 20) | ((uint32_t)(xb) << 26))
#define ZG_FSE_BL(e) ((e) & 0xFFFFu)
#define ZG_FSE_NB(e) (((e) >> 16) & 15u)
#define ZG_FSE_SYM(e) (((e) >> 20) & 63u)
#define ZG_FSE_XB(e) (((e) >> 26) & 31u)
// Huffman-weight FSE tables hold 8-bit symbols: [0,16) base_line, [16,20) num_bits, [20,28) symbol.
#define ZG_FSEW_PACK(bl, nb, sym) ((uint32_t)(bl) | ((uint32_t)(nb) << 16) | ((uint32_t)(sym) << 20))
#define ZG_FSEW_SYM(e) (((e) >> 20) & 255u)
// Packed Huffman entry (u16): low byte symbol, high byte num_bits.
#define ZG_HUF_PACK(sym, nb) ((uint16_t)((sym) | ((nb) << 8)))

// One block as described by the host parser (block header + both section headers + table lineage).
struct ZgBlock {
  uint64_t src_off;        // offset of the block body in the compressed buffer
  uint32_t src_len;        // body length (Block_Content: 1 for RLE blocks)
  uint32_t regen_size;     // raw/RLE block: decompressed size; compressed block: literals regenerated size
  uint32_t lit_comp_size;  // compressed/treeless literals: compressed size (tree description included)
  uint32_t lit_off;        // offset in the body of the literals payload (after the 1-5 byte header)
  uint32_t seq_off;        // offset in the body of the first byte after the sequences header
  uint32_t nseq;           // number of sequences
  uint8_t btype;           // ZG_BT_*
  uint8_t lit_type;        // ZG_LT_*
  uint8_t nstreams;        // 1 or 4 (Huffman literals)
  uint8_t seq_modes;       // modes byte (LL bits 7-6, OF 5-4, ML 3-2)
  uint32_t frame;          // index of the owning frame in the batch
  int32_t huf_slot;        // Huffman table slot this block decodes with (ZG_REF_UNINIT if none)
  int32_t ll_slot, of_slot, ml_slot;  // FSE arena slots holding the tables this block decodes with
  uint32_t host_status;    // error found by the host parser for this block (it and later blocks are not decoded)
  uint64_t lit_base;       // offset of this block's regenerated literals in the literals arena
  uint64_t seq_base;       // index of this block's first sequence in the sequence arena
  uint32_t seq_idx;        // position of this block in the list of blocks that have sequences (flatten scratch slot)
  uint32_t seq_host_status; // error the host parser found in the block's SEQUENCES section header: the reference meets it after it has decoded the
                            // literals (block_decoder.rs:131-190), so the literal stages still run and their errors come first (zg_k_merge)
};

// One frame of the batch.
struct ZgFrame {
  uint32_t first_block, nblocks;   // range in the batch's block array
  uint32_t first_unit, nunits;     // range in the batch's unit array
  uint32_t carry_slot;             // FSE arena slot holding the tables carried into this frame (dictionary / previous submit)
  int32_t carry_huf_slot;
  uint32_t hist_init[3];           // offset history at the first block (1,4,8 or dictionary / carried)
  uint32_t fixed_base;             // 1: the frame's new bytes start at out_base_fixed; 0: frames are packed back to back
  uint64_t out_base_fixed;
  uint64_t window_size;
  uint64_t prior_out;              // bytes of this frame already decoded by earlier submits (streaming)
  uint64_t dict_len;               // dictionary content length reachable before the frame start
  uint64_t prior_reach;            // of prior_out, the most recent bytes a match may still reach: what the caller has not drained (DecodeBuffer holds nothing older)
  uint32_t seq_first, seq_count;   // the frame's blocks that have sequences: a range of the batch's seq_blocks list
  uint32_t sparse;                 // so few sequences (literal-heavy data) that its matches are copied in order by one wave (zg_k_sparse)
  uint32_t pad2;                   //   instead of going through the sweep: a chain of launches per unit would cost more than the copies
  uint64_t prior_counted;          // DecodeBuffer::total_output_counter (decode_buffer.rs:16,74-77,108) after the earlier submits of this frame
};

// What the table kernel records per block.
struct ZgBlockAux {
  uint32_t seq_bits_off;   // offset in the body where the sequence bitstream starts (after table descriptions)
  uint32_t huf_desc_bytes; // bytes used by the Huffman tree description
  uint8_t log[3];          // accuracy logs of the tables DEFINED by this block (LL, OF, ML); 0 for RLE
  uint8_t pad;
};

// What zg_k_fparse leaves for zg_k_ftab: the three FSE table descriptions of a block's sequences section, read (one LANE per block:
// read_probabilities is a bit-serial chain, fse_decoder.rs:224-307) but not built yet.
struct ZgFtabParsed {
  int16_t probs[3][64];        // LL, OF, ML in section order: the probabilities of an FSE-described table (entries behind np are 0)
  uint8_t np[3], al[3];        // its symbols and accuracy log
  uint8_t rle[3];              // the symbol of an RLE-mode table
  uint8_t nparsed;             // descriptions read without an error: the error of description `nparsed` is `status` (3 and 0 when all are fine)
  uint8_t pad[2];
  uint32_t status;
  uint32_t done;               // bytes the descriptions take (the bitstream starts behind them)
};

// What the sequence kernel records per block.
struct ZgBlockSeqOut {
  uint32_t sum_ll;         // Σ literal lengths
  uint32_t sum_ml;         // Σ match lengths (block output size = regen_size + sum_ml)
  uint32_t hist_end[3];    // offset history after the block, symbolic encoding (see zg_dev.h)
  uint32_t pad;            // zg_k_seq -> zg_k_seqpost: bit position of the first sequence; after zg_k_seqpost: 1 + index of the sequence it rejected (0: none)
};

// One decoded sequence, ready for execution: 12 bytes. Positions are block-relative; on the flatten path a block
// regenerates at most 128 KiB, so they fit 17 bits. For larger (non-conforming) blocks the position fields wrap, but
// ml and ll = (next.lit_start - lit_start) mod 2^17 stay exact: the in-order fallback rebuilds positions from them.
struct ZgSeq {
  uint32_t of;             // resolved offset, or symbolic reference into the block's initial offset history
  uint32_t w1;             // [16:0] mdst: position where the match starts (= sum of earlier ll+ml, + this ll); [31:17] ml bits 14..0
  uint32_t w2;             // [16:0] lit_start: index of this sequence's first literal in the block's literals; [19:17] ml bits 17..15
};
#define ZG_SEQ_MDST(q) ((q).w1 & 0x1FFFFu)
#define ZG_SEQ_LIT(q) ((q).w2 & 0x1FFFFu)
#define ZG_SEQ_ML(q) (((q).w1 >> 17) | ((((q).w2 >> 17) & 7u) << 15))
#define ZG_SEQ_W1(mdst, ml) (((mdst) & 0x1FFFFu) | (((ml) & 0x7FFFu) << 17))
#define ZG_SEQ_W2(lit, ml) (((lit) & 0x1FFFFu) | ((((ml) >> 15) & 7u) << 17))

// Per block, after the scan.
struct ZgBlockPos {
  uint64_t out_base;       // absolute position of the block's first output byte in the batch output
  uint32_t hist_init[3];   // resolved offset history at block start
  uint32_t active;         // 1 if the block is to be executed (no earlier error in its frame)
};

struct ZgFrameOut {
  uint64_t out_base;       // where the frame's new bytes start in the batch output
  uint64_t out_size;       // bytes produced by this submit
  uint32_t status;         // first error (ZgStatus) or 0
  uint32_t bad_block;      // frame-relative index of the failing block
  uint32_t hist_end[3];    // offset history after the last good block
  uint32_t good_blocks;
  uint32_t fast;           // 1: every block regenerates <= 128 KiB -> flatten + sweep path; 0: in-order fallback (zg_k_lz)
  uint32_t err_packed;     // (frame-relative block << 8) | status of the first execution error, 0xFFFFFFFF if none
  uint64_t og_base;        // where the frame's flatten scratch starts (in u32): the scratch is indexed by output position
  uint64_t counted;        // what this submit adds to total_output_counter: the output of its compressed blocks (raw and RLE blocks are not
                           // counted, decode_buffer.rs:62-72), minus matches served from the dictionary alone (:159-172) once zg_k_exact ran
};

// how the caller's surface drains the reference's DecodeBuffer inside one submit (zg_exact.h)
#define ZG_DRAIN_NONE 0u        // not at all (FrameDecoder::decode_blocks, decode_from_to, the thin boundary: the caller drains between submits)
#define ZG_DRAIN_DECODE_ALL 1u  // FrameDecoder::decode_all: rounds of UptoBytes(1 MiB), each followed by a drain down to window_size

// LZ77 execution works on units: runs of consecutive blocks of one frame that zg_k_flatten resolves together.
// noseq: bit 0: none of the unit's blocks has sequences: all of it is literal bytes, final after zg_k_lit; bit 1 (ZG_UNIT_DIRECT):
// the frame's first unit, resolved to bytes by the flatten itself (zg_flat4.h). Either way: no scratch words, no sweep step.
// flatten stage (zg_flat4.h): parent markers of a tile byte, and the largest block output the flatten path handles
#define ZG_PAR_LIT 0xFFFFu   // tile byte is a literal (or dead)
#define ZG_PAR_EXIT 0x8000u  // tile byte is a match byte whose parent lies before the tile
#define ZG_FLAT_MAX 131072u  // largest block output the flatten path handles (Block_Maximum_Size)
#define ZG_UNIT_NOSEQ 1u
#define ZG_UNIT_DIRECT 2u
struct ZgUnit { uint32_t frame, first_block, nblocks, noseq; uint32_t desc, pad; };   // desc: the unit's entry in the sweep descriptors (0xFFFFFFFF: it has no sweep step)
struct ZgUnitInfo { uint32_t size; uint32_t noseq; uint32_t done; uint32_t pad; };   // written by zg_k_flatten: bytes of the unit; done: the run (epoch) whose flatten has finished the unit

// what a sweep workgroup needs to know about its unit
// head: batches (ZG_SW_BATCH bytes) at the front of the unit that nothing later depends on when no match reaches further back than the
// frame's window (the bytes a later unit may copy from are the last `window` bytes in front of it)
struct ZgSweepDesc { uint64_t out; uint64_t og; uint32_t size; uint32_t live; uint32_t head; uint32_t pad; };

// Huffman work: one group = streams that decode with the same table.
struct ZgHufGroup { int32_t slot; uint32_t first_item; uint32_t nitems; uint32_t pad; };

// 8 / 12 / 16 bytes as plain words (what the zx_* buffer primitives of zg_flat4.h move)
struct ZxU2 { uint32_t x, y; };
struct ZxU3 { uint32_t x, y, z; };
struct __attribute__((aligned(16))) ZxU4 { uint32_t x, y, z, w; };

// zg_k_seq's record of one sequence: the three FSE states it was decoded from, packed into 4 bytes (accuracy logs are <= 8 / 9 / 9)
#define ZG_RAW_PACK(of, ml, ll) ((uint32_t)(of) | ((uint32_t)(ml) << 8) | ((uint32_t)(ll) << 17))
#define ZG_RAW_OF(r) ((r) & 255u)
#define ZG_RAW_ML(r) (((r) >> 8) & 511u)
#define ZG_RAW_LL(r) (((r) >> 17) & 511u)

// Measurement switches that change bytes or verdicts (timing modes that drop a cost, the ramped chain beside the flatten) exist only in
// the development build (make dev -> libzgpu_dev.so, -DZG_DEV_SWITCHES): in the product library the expressions that read them are the
// constant 0 and the code behind them is not compiled.
#ifdef ZG_DEV_SWITCHES
#define ZG_DEVSW(x) (x)
#else
#define ZG_DEVSW(x) 0u
#endif

// The Huffman literals are decoded AFTER the position scan (literal-heavy submits, chosen by the host): the literals of a block
// without sequences go straight to the block's place in the output instead of through the arena and zg_k_lit's copy.
#define ZG_FLAG_LIT_DIRECT 0x40u

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
  ZgFtabParsed* ftab_parsed;   // [nblocks] zg_k_fparse -> zg_k_ftab
  uint8_t* slot_log;           // [nslots][4]: accuracy logs LL, OF, ML of the tables held by each FSE slot
  uint32_t* fse_arena;         // [nslots][ZG_FSE_SLOT_U32]
  uint16_t* huf_arena;         // [nhuf_slots][ZG_HUF_SLOT_U16]
  uint8_t* huf_maxbits;        // [nhuf_slots]
  uint32_t* status;            // [nblocks] first error per block (ZgStatus), 0 = ok
  uint32_t* tab_status;        // [nblocks] what zg_k_tables left in status (zg_k_huf runs beside zg_k_seq and must not see its errors)
  uint32_t* lit_status;        // [nblocks] zg_k_huf's errors, folded into status by zg_k_merge (literals are decoded before sequences: they outrank)
  uint32_t* lit_counts;        // [4 * nblocks] symbols each of the four streams of a block's literals holds (zg_k_huf -> zg_k_huf_uneven)
  uint8_t* lit_arena;          // regenerated Huffman literals
  ZgSeq* seq_arena;            // decoded sequences
  uint32_t* raw_arena;         // zg_k_seq's records (ZG_RAW_PACK: the three states of every sequence), same indexing as seq_arena
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
  uint32_t flags;              // bit 0: force the in-order fallback for every frame (tests); bits 2-3: shape of zg_k_flatten (0: 1024 threads x 16 KiB tiles, 1: 512 x 8 KiB);
                               // bits 4-5, 7: timing experiments of zg_k_flatten; bit 6: ZG_FLAG_LIT_DIRECT
  uint64_t og_words;           // size of the flatten scratch in u32
  uint32_t* og;                // flatten scratch: one u32 "effective offset" per output byte of a unit (0 = literal byte, final already)
  const ZgUnit* units;
  uint32_t nunits;
  const uint32_t* unit_list;   // the units in launch order: [0, nunits - ndirect) pointer-mode units and units without sequences (zg_k_flatten),
  uint32_t ndirect;            //   the last ndirect entries direct units (zg_k_flatten4)
  uint32_t pad_ul;
  ZgUnitInfo* unit_info;       // [nunits]
  ZgSweepDesc* sweep_desc;     // one per entry of step_units, written by zg_k_swprep after zg_k_flatten (or by the flatten itself: overlap_epoch)
  uint32_t overlap_epoch;      // != 0: the sweep chain runs beside the flatten; a unit's flatten publishes its descriptor and sets unit_info.done to this
  uint32_t pad_ov;
  const uint32_t* step_units;  // sweep step s fills the units step_units[list_off(s) ...] (unit s of every frame that has one)
  unsigned long long* dbg;     // phase cycle counters (profiling builds), diagnostics only
  uint64_t dst_cap_pre;        // != 0: the output (and the flatten scratch) were sized BEFORE the run from the frames' declared content sizes, and the
                               // LZ77 stages are already enqueued behind the scan: zg_k_scanf raises totals[2] when the frames produce more than this
                               // (a frame lied about its size), which turns every later kernel into a no-op; Batch::sync() then sizes and repeats them
};
