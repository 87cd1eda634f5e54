// zg_host_parse.h — host side of the boundary: frame header, block headers, the two section headers of a
// compressed block, and table lineage. This is the work the north star leaves on the host
// (ruzstd: decoding/frame.rs, block_decoder.rs:201-247, blocks/literals_section.rs:117-223,
// blocks/sequence_section.rs:108-167); everything per-byte runs in the HIP kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "zg_types.h"

namespace zg {

constexpr uint32_t kMagic = 0xFD2FB528u;                       // common/mod.rs:6
constexpr uint64_t kMinWindow = 1024;                          // common/mod.rs:10
constexpr uint64_t kMaxWindow = (1ull << 41) + 7ull * (1ull << 38);  // common/mod.rs:14
constexpr uint32_t kMaxBlockSize = 128 * 1024;                 // common/mod.rs:21
constexpr uint64_t kDefaultMaxWindow = 1024ull * 1024 * 128;   // frame_decoder.rs:25

struct FrameHeader {          // decoding/frame.rs:88-114
  uint8_t descriptor = 0;
  uint8_t window_descriptor = 0;
  bool has_dict_id = false;
  uint32_t dict_id = 0;
  uint64_t frame_content_size = 0;
  uint32_t header_size = 0;
  bool single_segment() const { return (descriptor >> 5) & 1; }
  bool content_checksum() const { return (descriptor >> 2) & 1; }
  bool has_fcs() const { return (descriptor >> 6) != 0 || single_segment(); }
};

// read_frame_header (frame.rs:6-85). On ZG_SKIP_FRAME *skip_magic/*skip_len are filled and *consumed = 8.
int read_frame_header(const uint8_t* src, size_t len, FrameHeader* h, size_t* consumed, uint32_t* skip_magic, uint32_t* skip_len);
// FrameHeader::window_size (frame.rs:116-139)
int frame_window_size(const FrameHeader& h, uint64_t* out);

struct BlockHeader {          // blocks/block.rs:31-44
  bool last = false;
  uint8_t type = 0;           // ZG_BT_*; 3 = reserved
  uint32_t decompressed_size = 0;
  uint32_t content_size = 0;
};
// read_block_header (block_decoder.rs:201-247) on 3 bytes. Returns ZG_RESERVED_BLOCK / ZG_BLOCK_SIZE_TOO_LARGE / 0.
int read_block_header(const uint8_t* p, BlockHeader* h);

// Table lineage of one frame: which arena slot holds the table each entropy stream decodes with right now
// (the state DecoderScratch carries from block to block, scratch.rs:15-27).
struct Lineage {
  int32_t huf = ZG_REF_UNINIT;
  int32_t ll = ZG_REF_UNINIT, of = ZG_REF_UNINIT, ml = ZG_REF_UNINIT;
};

struct ZgStepRange { uint32_t list_off, nunits, max_blocks; };

// Builds the submit: appends blocks (with parsed section headers and lineage) and frames.
class BatchBuilder {
 public:
  std::vector<ZgBlock> blocks;
  std::vector<ZgFrame> frames;
  std::vector<uint32_t> seq_blocks;
  std::vector<uint32_t> huf_items;
  std::vector<ZgHufGroup> huf_groups;
  std::vector<ZgUnit> units;
  std::vector<uint32_t> unit_list;      // the units in launch order: pointer-mode units (and units without sequences) first, then the direct ones
  uint32_t n_direct = 0;                //   (the last n_direct entries): the two bodies of the flatten are two kernels (zg_k_flatten / zg_k_flatten4)
  std::vector<uint32_t> step_units;     // concatenated unit lists of the sweep steps
  std::vector<ZgStepRange> steps;       // sweep step s: units step_units[list_off .. list_off + nunits)
  uint32_t unit_blocks = 0;    // blocks per unit; 0 = choose from the submit size
  uint32_t unit_blocks_used = 0;   // what finish() chose
  uint32_t sparse_max = 2048;      // a frame with at most this many sequences (and <= sparse_per_block per block) skips the sweep: zg_k_sparse (0: never)
  uint32_t sparse_per_block = 4;
  bool direct_units = true;    // first units of frames that start from nothing are resolved to bytes by the flatten itself (no scratch, no sweep step)
  uint32_t ramp_percent = 0;   // (measurement, ZGPU_RAMP) a submit of ONE long frame: unit sizes grow from (100 - ramp)% to (100 + ramp)% of the
                               // mean along the frame, the flatten raises a flag per unit and the sweep chain runs beside it on a third stream.
                               // Measured slower than flatten-then-sweep at every ramp (DESIGN.md "what did not work"), so 0 = off.
  bool ramped = false;         // finish() chose ramped units
  uint32_t direct_max_units = 32;   // ... in frames of at most this many units
  uint32_t direct_share10 = 13;     // tenths of a share of its frame's blocks a direct first unit gets when other units follow (finish())
  uint32_t flat_slots = 256;   // workgroups of zg_k_flatten the device runs at once (engine: CUs x workgroups per CU)
  uint64_t lit_bytes = 0;      // literals arena size
  uint64_t seq_count = 0;      // sequence arena size
  uint32_t nhuf_slots = 0;
  uint64_t out_bound = 0;      // upper bound of the output when every compressed block regenerates <= 128 KiB
  // finish(): the Huffman literals are worth decoding AFTER the position scan, straight into the output where a block has no
  // sequences (ZG_FLAG_LIT_DIRECT): literal-heavy input, where the arena -> output copy costs more than running the literals
  // chain behind the sequences chain instead of beside it
  bool lit_direct = false;
  bool lit_direct_allowed = true;   // (ZGPU_LIT_DIRECT=0: measurement / tests)
  uint32_t chain_slots = 8192;      // sequence chains the device runs at once (engine: CUs x 32)
  int seq_packed_force = -1;        // (ZGPU_SEQ_PACKED, development build) 0 / 1: never / always the packed form of zg_k_seq
  bool seq_packed = false;          // finish(): zg_k_seq runs in its packed-entry form (more blocks with sequences than one round holds)

  // Start a frame. carry: lineage handed in by a dictionary or an earlier submit (slots are resolved in finish()).
  // carry_mask: bit 0 Huffman, bit 1 LL, bit 2 OF, bit 3 ML — tables that already exist when the frame (or this run of
  // its blocks) starts: handed in by a dictionary or by an earlier submit of the same frame.
  uint32_t begin_frame(uint64_t window_size, const uint32_t hist[3], uint32_t carry_mask);
  // lineage after the last added block, slots resolved (valid after finish()): which arena slot holds each table now
  Lineage final_lineage() const { return final_; }
  // Add one block of the current frame. body points at Block_Content (content_size bytes available).
  // src_off is the body's offset in the buffer that will be uploaded. Returns the block's host status.
  int add_block(const BlockHeader& bh, const uint8_t* body, uint64_t src_off);
  // Mark the current frame as failed at its next block (host-side errors: truncated input, reserved block...).
  void fail_frame(int status);
  // Resolve slot numbers and build the work lists. Call once after the last block.
  void finish();

  uint32_t predefined_slot() const { return (uint32_t)blocks.size(); }
  uint32_t carry_slot(uint32_t frame) const { return (uint32_t)blocks.size() + 1 + frame; }
  uint32_t nslots() const { return (uint32_t)blocks.size() + 1 + (uint32_t)frames.size(); }

 private:
  Lineage cur_, final_;
  bool frame_failed_ = false;
  static constexpr int32_t kPredef = -2;   // resolved in finish()
  static constexpr int32_t kCarry = -3;
  static constexpr int32_t kCarryHuf = -3;
};

}  // namespace zg
