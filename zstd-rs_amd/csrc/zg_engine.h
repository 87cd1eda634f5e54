// zg_engine.h — one engine per (GPU, pair of HIP streams): device memory, upload of a parsed submit, the kernel
// pipeline, and result download. Host buffers never enter the kernels; torch is not involved.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "zg_host_parse.h"
#include "zg_kernels.h"

namespace zg {

// Bytes allocated in front of every output buffer: zg_k_sweep fetches the source of output byte i of a 4-byte group as
// byte i of the dword that starts i bytes before it, so up to 3 bytes in front of the first output byte are touched.
constexpr size_t kOutFront = 256;

// growable device buffer; a buffer that is grown again grows by at least half of its size (amortised reuse)
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n, bool keep = false, hipStream_t s = nullptr);
  void release();
  template <typename T> T* as() const { return (T*)p; }
};

// What the host knows about each frame of a submit (header fields + where it sits in the input).
struct FrameInfo {
  FrameHeader header;
  uint64_t window_size = 0;
  uint64_t src_begin = 0, src_end = 0;   // byte range of the frame in the input (header .. checksum)
  bool has_checksum = false;
  uint32_t checksum = 0;                 // Content_Checksum read from the data
  uint32_t nblocks = 0;
  int host_status = 0;                   // error found while walking the frame (truncated input, bad header, ...)
};

// What one frame carries from one submit of its blocks to the next (the reference's DecoderScratch, scratch.rs:15-27):
// the decode window (a dictionary's content sits in front of it), the offset history, and the Huffman / FSE tables a
// later Treeless / Repeat block may decode with. The device holds [front pad][dictionary content][the last `have` bytes of
// the frame]; bytes the caller has drained and that no match may reach any more are dropped when the buffer is rebuilt.
struct FrameState {
  DevBuf d_out;
  uint64_t base = 0;             // dictionary content bytes in front of the frame bytes
  uint64_t have = 0;             // frame bytes on the device (the most recent ones)
  uint64_t produced = 0;         // frame bytes decoded so far
  uint32_t hist[3] = {1, 4, 8};  // scratch.rs:44
  uint64_t counted = 0;          // DecodeBuffer::total_output_counter so far (decode_buffer.rs:16): decides between the two "offset too far" errors
  DevBuf d_fse;                  // carried FSE tables, one arena slot (ZG_FSE_SLOT_U32 packed entries)
  DevBuf d_huf;                  // carried Huffman table (ZG_HUF_SLOT_U16 entries)
  DevBuf d_tmp;                  // make_room: staging for the undrained bytes when they are moved down over themselves
  uint8_t logs[4] = {0, 0, 0, 0};// accuracy logs LL, OF, ML of the carried FSE tables
  uint8_t huf_maxbits = 0;
  uint32_t carry_mask = 0;       // bit 0 Huffman, 1 LL, 2 OF, 3 ML: which tables exist
  uint64_t window_size = 0;
  uint8_t* out_ptr() const { return (uint8_t*)d_out.p + kOutFront; }   // first byte of [dictionary content][frame bytes]
  // Make room for `extra` more frame bytes. `keep` = frame bytes that must stay reachable (what the caller still holds
  // undrained: DecodeBuffer semantics, decode_buffer.rs:79-111,182-219); older bytes are dropped when the buffer is rebuilt.
  int make_room(uint64_t extra, uint64_t keep, hipStream_t s);
  int reserve_window(uint64_t payload, hipStream_t s);
  // would make_room(extra, keep) move or reallocate what the buffer holds? (a download of earlier bytes must not be in flight then)
  bool room_moves(uint64_t extra, uint64_t keep) const {
    if (keep > have) keep = have;
    if (base && keep < have && d_out.p) return true;
    return !d_out.p || kOutFront + base + have + extra + 64 > d_out.cap;
  }
  void reset();
  void release();
};

enum { ZG_T_TABLES = 0, ZG_T_HUF, ZG_T_SEQ, ZG_T_SEQPOST, ZG_T_SCAN, ZG_T_LIT, ZG_T_FLAT, ZG_T_SWEEP, ZG_T_LZ, ZG_T_TOTAL, ZG_T_COUNT };

// Device buffers and events of one submit in flight. Engines keep finished ones for reuse: a stream of submits (block runs
// of a streaming decoder, frames pulled from a work queue) allocates once.
struct Scratch {
  DevBuf d_src, d_blocks, d_frames, d_aux, d_fparsed, d_slot_log, d_fse, d_huf, d_hufmax, d_status, d_lit, d_seq, d_seqout, d_pos, d_frameout,
      d_dst, d_seqblocks, d_hufitems, d_hufgroups, d_totals, d_og, d_units, d_unitinfo, d_stepunits, d_swdesc, d_dbg, d_raw, d_unitlist;
  hipEvent_t ev[ZG_T_COUNT + 1] = {};
  hipEvent_t ev_up = nullptr;                     // a submit prepared beside another one: its uploads are done (Engine::upload side)
  hipEvent_t ev_huf[2] = {}, ev_fork = nullptr, ev_fork3 = nullptr;   // zg_k_huf runs on the engine's second stream beside zg_k_seq
  hipEvent_t ev_sw[80] = {};                      // split sweep: heads on the second stream (zg_launch_sweep)
  hipEvent_t ev_flat[2] = {};                     // the direct units' flatten on the second stream beside the pointer-mode units' (zg_launch_flat)
  bool have_events = false;
  int init_events();
  void release();
  void release_but_output();
};

// Measurement and test switches (the ZGPU_* environment variables of tools/dev/README.md). The PRODUCT library (libzgpu.so) reads none of
// them: an environment variable must not be able to make a drop-in decoder slower, let alone wrong. The development build (make dev ->
// libzgpu_dev.so, -DZG_DEV_SWITCHES; what the tests that force a path and tools/dev load) reads them ONCE, when an engine is created:
// nothing on the submit path (prepare / run / sync, zg_launch_sweep) calls getenv. A test that wants a switch sets it before it creates
// its (development) context; a switch set afterwards is ignored (zgpu_debug_tuning shows what an engine took).
struct Tuning {
  bool dev_build = false;          // libzgpu_dev.so (-DZG_DEV_SWITCHES): the only build in which any of the switches below is read
  uint32_t unit_blocks = 0;        // ZGPU_UNIT_BLOCKS: blocks per flatten unit (0: chosen by BatchBuilder::finish)
  bool direct = true;              // ZGPU_DIRECT=0: no direct units
  uint32_t ramp_percent = 0;       // ZGPU_RAMP
  bool sparse_set = false;         // ZGPU_SPARSE_MAX given
  uint32_t sparse_max = 0;
  int lit_direct = -1;             // ZGPU_LIT_DIRECT: 0 never, else the host's choice
  uint32_t direct_share10 = 0;     // ZGPU_DIRECT_SHARE (tenths, >= 10)
  bool debug_timers = false;       // ZGPU_DEBUG_TIMERS
  bool force_inorder = false;      // ZGPU_FORCE_INORDER=1
  uint32_t flat_mode = 0;          // ZGPU_FLAT_MODE (timing experiments)
  uint32_t sweep_w = 0;            // ZGPU_SWEEP_W
  bool overlap = true;             // ZGPU_OVERLAP=0
  bool no_sweep = false;           // ZGPU_DEBUG_NO_SWEEP
  bool sweep_split = true;         // ZGPU_SWEEP_SPLIT=0
  bool no_exact = false;           // ZGPU_DEBUG_NO_EXACT
  bool no_presize = false;         // ZGPU_PRESIZE=0
  int flat4 = -1;                  // ZGPU_FLAT4: how the direct units of a submit are flattened — -1 (default): chosen per submit from the number of
                                   // direct units (Batch::launch_phase2); 0: by the kernel of the pointer-mode units (one 1024-thread workgroup per
                                   // CU, round 4's form); 1 / 4 / 6: by zg_k_flatten4 with 1024 / 512 / 256 threads and 8 / 4 / 2 KiB tiles (2 / 4 / 8
                                   // workgroups per CU); 2, 3, 5, 7: other shapes (measurement)
  int flat_shape = 0;              // ZGPU_FLAT_T=512 -> 1
  int seq_packed = -1;             // ZGPU_SEQ_PACKED: zg_k_seq's table entries — -1 (default): packed (16 bits, three workgroups per CU) when the submit has more
                                   // blocks with sequences than one round holds, else unpacked; 0 / 1: never / always
  ZgSweepTuning sweep;             // ZGPU_SWEEP_MODE / _NB / _GROUP / _HEAD_LDS
  static Tuning from_env();
};

class Engine;

// One submit: parsed input + its device state. Created by Engine::prepare.
class Batch {
 public:
  ~Batch();
  BatchBuilder bb;
  std::vector<FrameInfo> info;
  std::vector<ZgFrameOut> frame_out;     // valid after run() (sizes) / sync() (final status)
  uint64_t total_out = 0;                // valid after run()
  float ms[ZG_T_COUNT] = {0};
  int parse_status = 0;                  // frame-layer error that stops decode_all (first failing frame's status)
  uint64_t src_len = 0;
  uint64_t keep_bytes = 0;               // streaming submits: frame bytes that must stay reachable on the device (FrameState::make_room)
  uint32_t drain_rule = 0;               // how the caller's surface drains the reference's DecodeBuffer inside this submit (zg_exact.h: ZG_DRAIN_*)
  bool far_seen = false;                 // a sequence set an offset beyond its frame's window (zg_k_seqpost); valid after sync()
  uint64_t declared_total = 0;           // sum of the frames' Frame_Content_Size when every frame of the submit declares one (else 0): the
  bool all_declared = false;             //   output can be sized before the run, and the LZ77 stages enqueued without waiting for the scan
  bool presized = false;                 // the last run() did that
  bool exact_ran = false;                // sync(): zg_k_exact replayed the reference's buffer bookkeeping for this submit (tests)

  // Enqueue the kernel pipeline on the engine's streams. Two phases with one host round trip in between: the entropy
  // stages and the scan run first; the host reads the exact output size of every frame, sizes the output and the flatten
  // scratch to it, and then enqueues the LZ77 stages. Returns when the second phase is enqueued.
  int run();
  // streaming submits: after sync(), fold this run into the frame's carried state (tables, history, produced bytes)
  int commit(FrameState* fs);
  uint32_t carry_mask_in = 0;            // streaming runs: the tables that existed when the run was parsed
  uint32_t carry_mask_after() const;
  bool wait_upload = false;              // prepared with side uploads: run() waits for them on the device
  bool saw_last_block = false;           // the run ended with the frame's last block
  int sync();                            // wait, download per-frame results, compute timings
  int size_output();                     // after the scan: read the frames' sizes, size output + flatten scratch exactly (one host round trip)
  int launch_phase2();                   // the LZ77 stages (and, for literal-heavy submits, the Huffman streams in front of them)
  void launch_sweep(bool split, hipStream_t main = nullptr);   // main: the stream of the chain of steps (default: the engine's first)
  bool split_sweep = false;              // the last run used the split sweep: sync() checks that it was entitled to
  bool synced = false;                   // sync() ran after the last run(): outputs may be read
  uint32_t sweep_mode = 0;               // of the last run: 0 plain chain of steps, 1 split, 2 split and then repeated as a plain chain (tests)
  int read_output(uint64_t off, uint8_t* dst, uint64_t n);   // D2H
  int read_output_async(uint64_t off, uint8_t* dst, uint64_t n, hipStream_t s);
  const uint8_t* device_output() const { return dev.dst; }
  // after sync(): free everything but the plaintext (a finished submit that waits to be read: zgpu_pool_decode_all)
  void release_scratch();
  // intermediates, for parity tests
  int read_block_status(std::vector<uint32_t>* out);
  int read_literals(uint32_t block, std::vector<uint8_t>* out);
  int read_sequences(uint32_t block, std::vector<ZgSeq>* seqs, ZgBlockSeqOut* so, ZgBlockPos* pos);
  int read_fse_slot(uint32_t slot, std::vector<uint32_t>* entries, uint8_t logs[4]);
  int read_huf_slot(uint32_t slot, std::vector<uint16_t>* entries, int* max_bits);
  int read_debug(uint64_t out[1024]);
  int read_scratch(int what, uint64_t off, void* dst, uint64_t n);   // what: 0 flatten scratch (u32 per output byte), 1 ZgUnitInfo[]
  int unit_scratch_base(uint32_t unit, uint64_t* base);              // word index of a unit's first byte in the flatten scratch

 private:
  friend class Engine;
  Engine* eng = nullptr;
  Scratch* sc = nullptr;
  ZgBatchDev dev{};
  std::vector<ZgSweepStep> sweep_steps;
  uint64_t og_words = 0;
  bool ran = false;
  uint32_t epoch_ = 0;                   // runs of this batch so far (the flatten's per-unit flags carry it)
  FrameState* fs = nullptr;              // streaming submit: the frame state this run reads from / writes into
};

class Engine {
 public:
  static int create(int device, Engine** out);
  ~Engine();
  uint64_t max_window = kDefaultMaxWindow;
  // Walk `len` bytes of concatenated frames (decode_all semantics, frame_decoder.rs:541-577), upload everything.
  int prepare(const uint8_t* src, size_t len, Batch** out);
  // Same for a run of blocks of ONE frame that starts at a block header (the FrameDecoder mirror parsed the frame
  // header itself). *consumed = bytes of the run (block headers, bodies, checksum).
  // max_blocks: 0 = up to the last block of the frame. fs carries the frame's state across calls; keep = frame bytes
  // that must stay reachable (FrameState::make_room).
  // side: the run in front (of the same frame) is still on the GPU — this run's bytes and tables travel on the upload stream beside it and its run()
  // waits for them; carry_mask_now: the tables that will exist when it starts (Batch::carry_mask_after of the run in front), fs->carry_mask if null
  int prepare_run(const uint8_t* src, size_t len, FrameState* fs, bool has_checksum, uint32_t max_blocks, uint64_t keep, Batch** out, size_t* consumed,
                  bool side = false, const uint32_t* carry_mask_now = nullptr);
  // Same for blocks whose 3-byte headers the CALLER parsed (the thin boundary: ruzstd keeps read_block_header,
  // block_decoder.rs:201-247): hb[i] describes Block_Content i inside src. The run ends with the first block marked last.
  struct HostBlock { uint64_t src_off; uint32_t src_len; uint32_t raw_rle_size; uint8_t type; uint8_t last; };
  int prepare_blocks(const uint8_t* src, size_t len, const HostBlock* hb, size_t n, FrameState* fs, uint64_t keep, Batch** out);
  hipStream_t stream() const { return stream_; }
  hipStream_t copy_stream() const { return stream2_; }
  hipStream_t download_stream() const { return stream3_; }
  hipStream_t download_stream2() const { return stream5_; }  // the second half of a large download (two copy engines instead of one: zg_stream.cpp)
  hipStream_t upload_stream() const { return stream4_; }     // uploads of a run that is prepared while the run in front is on the GPU (prepare_run side = true)
   // D2H of a finished submit while the next one runs (zgpu_pool_decode_all): nothing else uses it then
  int device() const { return device_; }
  int compute_units() const { return cus_; }
  const Tuning& tuning() const { return tn_; }
  std::string last_error;

 private:
  friend class Batch;
  int device_ = 0;
  int cus_ = 256;                // compute units of the device (MI355X: 256)
  Tuning tn_;                    // the ZGPU_* switches as they were when the engine was created
  bool no_presize_ = false;      // (ZGPU_PRESIZE=0: measurement / tests) never size the output before the run
  int flat_shape_ = 0;           // zg_k_flatten shape: 0 = 1024 threads x 16 KiB tiles (one workgroup per CU), 1 = 512 x 8 KiB (two)
  hipStream_t stream_ = nullptr, stream2_ = nullptr, stream3_ = nullptr, stream4_ = nullptr, stream5_ = nullptr;   // stream3_: the flatten, when the sweep chain runs beside it
  std::vector<Scratch*> free_;   // finished submits' buffers, for reuse
  Scratch* acquire();
  void recycle(Scratch* s);
  int fail(hipError_t e, const char* what);
  int upload(Batch* b, const uint8_t* src, size_t len, Batch** out, bool side = false);
};

// Host-only walk of concatenated frames into a BatchBuilder (no GPU involved; unit-tested on CPU).
int parse_frames(const uint8_t* src, size_t len, uint64_t max_window, BatchBuilder* bb, std::vector<FrameInfo>* info);
int parse_block_run(const uint8_t* src, size_t len, uint64_t window, bool has_checksum, const uint32_t hist[3], uint32_t carry_mask,
                    uint32_t max_blocks, BatchBuilder* bb, std::vector<FrameInfo>* info, size_t* consumed, bool* saw_last);
// Byte ranges of the frames (and skippable frames) of a buffer, by walking frame and block headers only: what a work queue
// needs to hand whole frames to different GPUs. Stops at the first malformed frame (its status is returned; ranges found so
// far stay valid).
struct FrameSpan { uint64_t begin, end; uint64_t content_size; bool has_content_size; bool skippable; };
int split_frames(const uint8_t* src, size_t len, std::vector<FrameSpan>* out);

}  // namespace zg
