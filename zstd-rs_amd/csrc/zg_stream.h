// zg_stream.h — the io::Read surface with read-ahead: host side of StreamingDecoder::read (ruzstd/src/decoding/streaming_decoder.rs:119-155).
//
// The reference decodes lazily: read(buf) calls decode_blocks(UptoBytes(missing)) until buf.len() bytes can be collected, i.e. one
// block per call for an ordinary 8 KiB reader (std::io::copy, cli/src/main.rs:142-144). A GPU submit costs as long as one block's
// sequence chain (~1.5 ms) whatever it holds, so the same schedule on this engine is ~0.08 GB/s. Nothing a caller can observe through
// io::Read forbids decoding AHEAD of the reader, as long as
//   * the bytes are the same,
//   * read() returns what the reference's would (buf.len() bytes unless the frame ends first),
//   * an error surfaces in the read() call in which the reference's decoder would have met it,
//   * nothing beyond the frame's last block (+ checksum) is taken from the source.
// StreamCore does that in three modes:
//   INLINE    runs of blocks decoded on the caller's thread when a read() needs bytes, but more than it needs (8, 32, 128 ... blocks);
//   PIPE      a worker thread owns the engine: it decodes run k + 1 while run k's plaintext travels to a pinned host ring (its own DMA
//             stream) and the reader drains run k - 1 from the ring; a hasher thread keeps XXH64 off the reader's path; large reads
//             are copied out by several threads. Bounded by a read-ahead budget (ring size) whatever the frame's length;
//   LOCKSTEP  the reference's own schedule, block by block (frames the caller asked to decode without read-ahead, and the fallback).
// Speculation rule: a run that was decoded ahead is taken only if it is CLEAN — no block failed and no sequence set an offset beyond
// the frame's window (what such a match may reach depends on what the caller had drained by then: decode_buffer.rs:79-111). Otherwise
// the run is dropped, its good prefix is decoded again on its own, the rest of its source bytes is handed back, and the stream goes on
// in LOCKSTEP from exactly the state the reference would be in: same bytes, same error, same read() call. (Conforming streams are
// always clean.)
//
// The engine is behind StreamBackend, so that this file — threads, ring, hand-offs, fallback — runs on the CPU in tests/emu against a
// table-driven stand-in (tests/test_stream_cpu.py, also under ThreadSanitizer); the product's backend is in zg_stream.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "zg_host_parse.h"
#include "zg_types.h"
#include "zg_xxh64.h"

namespace zg {

struct StreamRun {             // what the host learns about one run of blocks
  uint32_t nblocks = 0;        // blocks the run held
  uint32_t good_blocks = 0;    // blocks in front of the first one that failed (== nblocks when none did)
  int status = 0;              // that block's error leaf (0: none)
  bool far = false;            // a sequence set an offset beyond the frame's window
  uint64_t out_size = 0;       // plaintext bytes of the good blocks
  bool saw_last = false;       // the run ended with the frame's last block (and none failed)
  bool has_checksum = false;   // ... and the 4 bytes behind it were there
  uint32_t checksum = 0;
};

class StreamBackend {
 public:
  virtual ~StreamBackend() {}
  // A run = `nblocks` whole blocks of the frame: src starts at a block header and holds them completely (+ the Content_Checksum behind
  // the last block of a frame that has one, when `len` reaches that far). Three steps, so that the host work of run k + 1 overlaps the
  // device work of run k:
  //   prepare  host walk, plan, uploads. May be called while the run in front is still on the device (before its wait()).
  //   launch   the prepared run goes to the device; the run in front has been committed or dropped. keep: frame bytes in front of the
  //            run that must stay in reach on the device.
  //   wait     its verdicts. Nothing is folded into the frame's state yet (commit / discard).
  virtual int prepare(const uint8_t* src, size_t len, uint32_t nblocks) = 0;
  virtual int launch(uint64_t keep) = 0;
  virtual int wait(StreamRun* out) = 0;
  virtual void drop_prepared() = 0;
  int run(const uint8_t* src, size_t len, uint32_t nblocks, uint64_t keep, StreamRun* out) {
    int e = prepare(src, len, nblocks);
    if (!e) e = launch(keep);
    if (!e) e = wait(out);
    return e;
  }
  virtual int commit() = 0;                                         // fold the last run into the frame's state (DecoderScratch after decode_block_content)
  virtual void discard() = 0;                                       // drop it instead
  virtual int fetch(uint8_t* dst, uint64_t off, uint64_t n) = 0;    // plaintext [off, off + n) of the last COMMITTED run -> host memory, asynchronous
  virtual int fetch_wait() = 0;
  // LOCKSTEP after PIPE: the reference's buffer holds everything the reader has not drained, and a (non-conforming) match may reach all
  // of it: make the frame's most recent n bytes (host copy at `held`) reachable on the device again
  virtual int rebase(const uint8_t* held, uint64_t n) = 0;
  virtual int pipe_begin(uint64_t window_bytes) { return 0; }       // the calling thread hands the engine to a worker (own streams, window reserved)
  virtual void pipe_end() {}
  virtual void thread_init() {}                                     // first call on the worker thread
  virtual void* host_alloc(size_t n) = 0;                           // DMA-able host memory (ring, staging)
  virtual void host_free(void* p, size_t n) = 0;
};

struct StreamOpts {
  uint64_t read_ahead = 0;          // bytes of plaintext that may be decoded ahead of the reader (ring size). 0: default; 1: none (LOCKSTEP only)
  bool hash = true;                 // XXH64 of the drained bytes (ruzstd's `hash` feature, default on)
  uint32_t first_run_blocks = 8;    // INLINE: blocks of the first run; each later one holds four times as many
  uint64_t pipe_after = 32ull << 20;   // INLINE -> PIPE once this many bytes are decoded and the frame goes on (or at once, when the header declares more)
  uint32_t copy_threads = 3;        // PIPE: helper threads for reads of 512 KiB and more
  uint64_t max_run_src = 96ull << 20;  // PIPE: source bytes per run at most (size of a staging buffer)
  bool allow_pipe = true;
  uint32_t test_fail_thread = 0;    // (tests/emu only) 1: starting the first thread of PIPE fails; 2: the worker runs, the next thread fails
};
constexpr uint64_t kStreamDefaultReadAhead = 512ull << 20;
constexpr uint64_t kStreamNoReadAhead = 1;

typedef size_t (*StreamReadFn)(void* user, uint8_t* dst, size_t n);

// A run of whole blocks taken from the source.
struct StreamStage {
  const uint8_t* p = nullptr;
  size_t len = 0;               // block headers + bodies (+ checksum)
  uint32_t nblocks = 0;
  bool last = false;            // holds the frame's last block (and its checksum, if the frame has one)
  int stop = 0;                 // why the pull ended early: 0, or the frame-layer error the NEXT block meets (truncated header / body / checksum, reserved type, size)
  uint8_t* own = nullptr;       // callback sources: the buffer the bytes were read into
  size_t own_cap = 0;
};

// The source: an io::Read callback or a memory slice (zero copy), with the bytes a dropped run hands back in front.
class StreamSrc {
 public:
  StreamReadFn read = nullptr;
  void* user = nullptr;
  const uint8_t* slice = nullptr;
  size_t slice_len = 0, slice_pos = 0;
  bool has_checksum = false;
  bool ended = false;           // the last block (+ checksum) has been taken: nothing behind it belongs to this frame
  std::vector<uint8_t> back;    // callback sources: bytes taken from the callback and handed back (served first)
  size_t back_pos = 0;
  bool is_slice() const { return slice != nullptr || read == nullptr; }

  size_t get(uint8_t* dst, size_t n) {                 // callback sources: read_exact-like, short only at the end of the input
    size_t got = 0;
    if (back_pos < back.size()) {
      const size_t k = back.size() - back_pos < n ? back.size() - back_pos : n;
      memcpy(dst, back.data() + back_pos, k);
      back_pos += k; got = k;
      if (back_pos == back.size()) { back.clear(); back_pos = 0; }
    }
    while (got < n) {
      const size_t r = read(user, dst + got, n - got);
      if (r == 0) break;
      got += r;
    }
    return got;
  }
  void unget(const uint8_t* p, size_t n) {             // hand bytes back: they are served before anything else
    if (!n) return;
    if (is_slice()) { slice_pos -= n; ended = false; return; }
    std::vector<uint8_t> nb(p, p + n);
    nb.insert(nb.end(), back.begin() + back_pos, back.end());
    back.swap(nb); back_pos = 0; ended = false;
  }

  // Take up to max_blocks whole blocks (at most max_bytes of source, when that is not 0 and at least one block fits).
  // spec: the run will be decoded ahead — a last block whose checksum is cut off stays in the source (LOCKSTEP decodes it and then
  // reports the missing checksum, like the reference: frame_decoder.rs:347-359).
  // Callback sources read into st->own (capacity own_cap, grown with `grow` when that is given); slices are viewed in place.
  void pull(uint32_t max_blocks, size_t max_bytes, bool spec, StreamStage* st, std::vector<uint8_t>* grow) {
    st->p = nullptr; st->len = 0; st->nblocks = 0; st->last = false; st->stop = 0;
    if (ended) return;
    if (is_slice()) {
      const uint8_t* const base = slice + slice_pos;
      size_t p = 0;
      const size_t avail = slice_len - slice_pos;
      while (st->nblocks < max_blocks) {
        if (avail - p < 3) { st->stop = ZG_FAILED_READ_BLOCK_HEADER; break; }
        BlockHeader bh;
        const int hs = read_block_header(base + p, &bh);
        if (hs) { st->stop = hs; break; }
        if (avail - p - 3 < bh.content_size) { st->stop = ZG_FAILED_READ_BLOCK_BODY; break; }
        size_t q = p + 3 + bh.content_size;
        if (max_bytes && st->nblocks && q + 4 > max_bytes) break;
        if (bh.last && has_checksum) {
          if (avail - q < 4) {
            st->stop = ZG_FAILED_READ_CHECKSUM;
            if (spec) break;
            p = q; st->nblocks++; st->last = true;
            break;
          }
          q += 4;
        }
        p = q; st->nblocks++;
        if (bh.last) { st->last = true; break; }
      }
      st->p = base; st->len = p;
      slice_pos += p;
      if (st->last && !st->stop) ended = true;
      if (st->last && st->stop) ended = true;         // (LOCKSTEP took the last block without its checksum: nothing more to take)
      return;
    }
    // callback source
    uint8_t* buf = st->own;
    size_t cap = st->own_cap;
    auto room = [&](size_t need) -> bool {
      if (need <= cap) return true;
      if (!grow) return false;
      grow->resize(need + (need >> 1) + 4096);
      buf = grow->data(); cap = grow->size();
      return true;
    };
    size_t p = 0;
    while (st->nblocks < max_blocks) {
      if (!room(p + 3)) break;
      const size_t h = get(buf + p, 3);
      if (h < 3) { unget(buf + p, h); st->stop = ZG_FAILED_READ_BLOCK_HEADER; break; }
      BlockHeader bh;
      const int hs = read_block_header(buf + p, &bh);
      if (hs) { unget(buf + p, 3); st->stop = hs; break; }
      const size_t want = 3 + (size_t)bh.content_size + ((bh.last && has_checksum) ? 4u : 0u);
      if ((max_bytes && st->nblocks && p + want > max_bytes) || !room(p + want)) { unget(buf + p, 3); break; }
      const size_t b = get(buf + p + 3, bh.content_size);
      if (b < bh.content_size) { unget(buf + p, 3 + b); st->stop = ZG_FAILED_READ_BLOCK_BODY; break; }
      size_t q = p + 3 + bh.content_size;
      if (bh.last && has_checksum) {
        const size_t c = get(buf + q, 4);
        if (c < 4) {
          st->stop = ZG_FAILED_READ_CHECKSUM;
          if (spec) { unget(buf + p, q - p + c); break; }
          p = q; st->nblocks++; st->last = true;
          break;
        }
        q += 4;
      }
      p = q; st->nblocks++;
      if (bh.last) { st->last = true; break; }
    }
    st->p = buf; st->len = p;
    if (grow && buf == grow->data()) { st->own = nullptr; st->own_cap = 0; }
    if (st->last) ended = true;
  }
};

// bytes of the first n blocks of a run (block headers are well-formed: the pull checked them)
inline size_t stream_prefix_len(const uint8_t* p, size_t len, uint32_t n) {
  size_t q = 0;
  for (uint32_t i = 0; i < n && q + 3 <= len; i++) {
    BlockHeader bh;
    if (read_block_header(p + q, &bh)) break;
    q += 3 + bh.content_size;
  }
  return q < len ? q : len;
}

// memcpy of a large read over a few helper threads (one reader thread copies ~15-20 GB/s; the ring is filled at ~50)
class StreamCopyPool {
 public:
  ~StreamCopyPool() { stop(); }
  void start(uint32_t n) {
    for (uint32_t i = 0; i < n; i++) th_.emplace_back([this, i]() { run(i); });
    parts_.resize(n);
  }
  void stop() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
    th_.clear();
  }
  void copy(uint8_t* dst, const uint8_t* src, size_t n) {
    const size_t nt = th_.size();
    if (!nt || n < (512u << 10)) { memcpy(dst, src, n); return; }
    const size_t share = (((n + nt) / (nt + 1)) + 4095) & ~(size_t)4095;   // ceil(n / (nt + 1)), page-aligned: (nt + 1) shares cover n
    size_t off = share < n ? share : n;                 // [0, off) is the caller's
    uint32_t used = 0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < nt && off < n; i++) {
        const size_t k = n - off < share ? n - off : share;
        parts_[i] = Part{dst + off, src + off, k};
        off += k; used++;
      }
      for (size_t i = used; i < nt; i++) parts_[i] = Part{nullptr, nullptr, 0};
      pending_.store(used, std::memory_order_relaxed);
      gen_++;
    }
    cv_.notify_all();
    memcpy(dst, src, share < n ? share : n);
    while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }
 private:
  struct Part { uint8_t* d; const uint8_t* s; size_t n; };
  void run(uint32_t i) {
    uint64_t seen = 0;
    for (;;) {
      Part p;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
        p = parts_[i];
      }
      if (p.n) { memcpy(p.d, p.s, p.n); pending_.fetch_sub(1, std::memory_order_release); }
    }
  }
  std::vector<std::thread> th_;
  std::vector<Part> parts_;
  std::mutex mu_;
  std::condition_variable cv_;
  uint64_t gen_ = 0;
  bool quit_ = false;
  std::atomic<uint32_t> pending_{0};
};

class StreamCore {
 public:
  enum Mode { INLINE = 0, PIPE = 1, LOCKSTEP = 2 };
  // where the time goes (microseconds; diagnostics, zgpu_streaming_stats): worker — [0] waiting for a run from the reader, [1] in the
  // backend's run (parse + upload + kernels), [2] waiting for the previous run's download, [3] commit, [4] waiting for room in the ring;
  // reader — [5] waiting for bytes, [6] copying out of the ring, [7] taking runs from the source
  std::atomic<uint64_t> tus[8] = {};
  struct Tick {
    std::atomic<uint64_t>* a; std::chrono::steady_clock::time_point t0;
    explicit Tick(std::atomic<uint64_t>* x) : a(x), t0(std::chrono::steady_clock::now()) {}
    ~Tick() { a->fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed); }
  };
  StreamSrc src;
  // the frame (set by the owner before the first read)
  uint64_t window = 0;
  uint64_t content_size = 0;        // 0: not declared
  uint64_t header_bytes = 0;        // counted into bytes_read_from_source
  StreamCore(StreamBackend* be, const StreamOpts& o) : be_(be), o_(o) {
    if (o_.read_ahead == 0) o_.read_ahead = kStreamDefaultReadAhead;
    mode_ = o_.read_ahead == kStreamNoReadAhead ? LOCKSTEP : INLINE;
    hash_.reset(0);
    ramp_ = o_.first_run_blocks ? o_.first_run_blocks : 1;
  }
  ~StreamCore() { stop_pipe(false); if (stage_vec_own_) {} }

  // ---- FrameDecoder accessors (frame_decoder.rs:246-297), as the streaming decoder's get_ref() shows them. While the stream reads ahead
  // they run ahead of what the reader has been given: blocks_decoded and bytes_read_from_source count what has been DECODED.
  bool is_finished() const { return fin_.load(std::memory_order_acquire); }
  uint64_t blocks_decoded() const { return blocks_.load(std::memory_order_relaxed); }
  uint64_t bytes_read_from_source() const { return header_bytes + src_bytes_.load(std::memory_order_relaxed); }
  bool checksum_from_data(uint32_t* out) const { if (!have_cs_.load(std::memory_order_acquire)) return false; *out = cs_; return true; }
  size_t can_collect() const {
    const uint64_t a = avail();
    if (is_finished()) return (size_t)a;
    return a > window ? (size_t)(a - window) : 0;
  }
  // XXH64 of the bytes handed to the reader so far, low 32 bits (frame_decoder.rs:263-270)
  uint32_t calculated_checksum() {
    if (mode_ == PIPE && hasher_on_) {
      const uint64_t t = tail_.load(std::memory_order_relaxed);
      while (hashed_.load(std::memory_order_acquire) != t) std::this_thread::yield();
    }
    return (uint32_t)hash_.digest();
  }
  Mode mode() const { return mode_; }
  uint64_t runs() const { return runs_; }             // runs decoded ahead and taken (tests)
  uint64_t dropped_runs() const { return dropped_; }  // runs decoded ahead and dropped (tests)
  uint64_t host_bytes() const { return buf_.capacity() + ring_cap_; }

  // impl Read for StreamingDecoder (streaming_decoder.rs:119-155)
  int read(uint8_t* dst, size_t cap, size_t* n_out) {
    *n_out = 0;
    if (is_finished() && avail() == 0) return ZG_OK;                       // :125-130
    size_t done = 0;
    for (;;) {
      if (mode_ == PIPE) {
        bool again = false;
        const int st = read_pipe(dst, cap, &done, &again);
        if (st) return st;
        if (!again) break;
        continue;                                                           // the pipe stopped: go on in LOCKSTEP (or INLINE never again)
      }
      if (can_collect() >= cap - done || is_finished()) break;
      const int st = mode_ == INLINE ? step_inline(cap - done) : step_lockstep(cap - done);
      if (st) return st;
    }
    if (mode_ != PIPE) {
      size_t k = can_collect();
      if (k > cap - done) k = cap - done;
      done += drain_buf(dst ? dst + done : nullptr, k);
    }
    *n_out = done;
    return ZG_OK;
  }

 private:
  StreamBackend* be_;
  StreamOpts o_;
  Mode mode_;
  // INLINE / LOCKSTEP: decoded, not yet drained bytes (DecodeBuffer, decode_buffer.rs:9-17)
  std::vector<uint8_t> buf_;
  size_t head_ = 0;
  Xxh64 hash_;
  std::vector<uint8_t> stage_vec_;
  bool stage_vec_own_ = false;
  uint32_t ramp_ = 8;
  uint64_t decoded_ = 0;            // plaintext bytes decoded so far
  uint64_t runs_ = 0, dropped_ = 0;
  std::atomic<bool> fin_{false};
  std::atomic<bool> have_cs_{false};
  uint32_t cs_ = 0;
  std::atomic<uint64_t> blocks_{0}, src_bytes_{0};

  size_t held_buf() const { return buf_.size() - head_; }
  uint64_t avail() const { return held_buf() + (pub_.load(std::memory_order_acquire) - tail_.load(std::memory_order_relaxed)); }

  size_t drain_buf(uint8_t* dst, size_t n) {                               // DecodeBuffer::drain_to (decode_buffer.rs:256-314)
    if (!n) return 0;
    if (dst) memcpy(dst, buf_.data() + head_, n);
    if (o_.hash) hash_.update(buf_.data() + head_, n);
    head_ += n;
    if (head_ == buf_.size()) { buf_.clear(); head_ = 0; }
    else if (head_ > (1u << 22) && head_ > held_buf()) { buf_.erase(buf_.begin(), buf_.begin() + head_); head_ = 0; }
    return n;
  }

  // defer_fin (PIPE, worker thread): the frame counts as finished only when its last bytes are in the ring (publish_fin)
  void account(const StreamRun& r, size_t src_len, uint32_t nblocks_counted, bool defer_fin = false) {
    blocks_.fetch_add(nblocks_counted, std::memory_order_relaxed);
    src_bytes_.fetch_add(src_len, std::memory_order_relaxed);
    decoded_ += r.out_size;
    if (r.saw_last) {
      if (r.has_checksum) { cs_ = r.checksum; have_cs_.store(true, std::memory_order_release); }
      if (!src.has_checksum || r.has_checksum) {                              // is_finished also needs the checksum (:289-293)
        if (defer_fin) fin_pending_ = true;
        else fin_.store(true, std::memory_order_release);
      }
      last_seen_ = true;
    }
  }
  // the block that failed: its header was read and counted before its body failed (frame_decoder.rs:325-341) — every error but the
  // three of read_block_header and a checksum that is not there
  void count_failed_header(int st) {
    if (st != ZG_FAILED_READ_BLOCK_HEADER && st != ZG_RESERVED_BLOCK && st != ZG_BLOCK_SIZE_TOO_LARGE && st != ZG_FAILED_READ_CHECKSUM)
      src_bytes_.fetch_add(3, std::memory_order_relaxed);
  }
  void publish_fin() { if (fin_pending_) { fin_pending_ = false; fin_.store(true, std::memory_order_release); } }
  bool last_seen_ = false;          // the last block has been decoded (the frame yields no more bytes)
  bool fin_pending_ = false;

  // append the committed run's plaintext to the host buffer (synchronous)
  int fetch_to_buf(uint64_t n) {
    if (!n) return ZG_OK;
    const size_t old = buf_.size();
    buf_.resize(old + n);
    int st = be_->fetch(buf_.data() + old, 0, n);
    if (!st) st = be_->fetch_wait();
    return st;
  }

  // ---- LOCKSTEP: the reference's schedule (streaming_decoder.rs:134-150, decode_blocks(UptoBytes) frame_decoder.rs:309-377)
  int step_lockstep(size_t missing) {
    if (last_seen_) return ZG_FAILED_READ_BLOCK_HEADER;                      // (the last block is decoded, its checksum was cut off: the reference looks for another block header, frame_decoder.rs:325-327)
    uint32_t m = (uint32_t)((missing - can_collect() + kMaxBlockSize - 1) / kMaxBlockSize);   // UptoBytes(need) never stops before ceil(need / 128 KiB) blocks
    if (m == 0) m = 1;
    StreamStage st;
    st.own = nullptr; st.own_cap = 0;
    src.pull(m, 0, false, &st, &stage_vec_);
    if (st.nblocks) {
      StreamRun r;
      int e = be_->run(st.p, st.len, st.nblocks, held_buf(), &r);
      if (e) return e;
      if ((e = be_->commit())) return e;                                     // (a failed run too: what its good blocks produced is there, like the reference's buffer)
      if ((e = fetch_to_buf(r.out_size))) return e;
      const size_t good_len = r.good_blocks == st.nblocks ? st.len : stream_prefix_len(st.p, st.len, r.good_blocks);
      account(r, good_len, r.good_blocks);
      if (r.status) { count_failed_header(r.status); return r.status; }
    }
    if (st.stop) { count_failed_header(st.stop); return st.stop; }
    if (st.nblocks == 0) return ZG_FAILED_READ_BLOCK_HEADER;                 // (only behind an earlier error: the frame's blocks are used up and it is not finished)
    return ZG_OK;
  }

  // ---- INLINE: a run decoded ahead on the caller's thread when a read needs bytes
  int step_inline(size_t missing) {
    if (o_.allow_pipe && !last_seen_ && (decoded_ >= o_.pipe_after || (decoded_ == 0 && content_size >= o_.pipe_after))) {
      const int st = start_pipe();
      if (st == ZG_OK) return ZG_OK;
      o_.allow_pipe = false;                                                  // (no memory for the ring ...: stay inline)
    }
    uint32_t m = (uint32_t)((missing - can_collect() + kMaxBlockSize - 1) / kMaxBlockSize);
    if (m < ramp_) m = ramp_;
    if (ramp_ < 2048) ramp_ *= 4;
    StreamStage st;
    src.pull(m, 0, true, &st, &stage_vec_);
    if (st.nblocks == 0) { mode_ = LOCKSTEP; return ZG_OK; }                  // the next block cannot be read whole: the error belongs to the read() that needs it
    StreamRun r;
    int e = be_->run(st.p, st.len, st.nblocks, held_buf(), &r);
    if (e) return e;
    if (clean(r, st.nblocks)) {
      if ((e = be_->commit()) || (e = fetch_to_buf(r.out_size))) return e;
      account(r, st.len, st.nblocks);
      runs_++;
      if (st.stop) mode_ = LOCKSTEP;
      return ZG_OK;
    }
    be_->discard();
    dropped_++;
    size_t used = 0;
    if ((e = salvage(st.p, st.len, r, &used, false))) return e;
    src.unget(st.p + used, st.len - used);
    mode_ = LOCKSTEP;
    return ZG_OK;
  }

  static bool clean(const StreamRun& r, uint32_t nblocks) { return r.status == 0 && !r.far && r.good_blocks == nblocks && r.nblocks == nblocks; }

  // A dropped run: the blocks in front of the one that failed are decoded again as a run of their own (clean by construction unless a
  // far offset hides among them: then nothing is taken). *used = source bytes taken. to_ring: PIPE (worker thread) -> ring, else -> buf_.
  int salvage(const uint8_t* p, size_t len, const StreamRun& bad, size_t* used, bool to_ring) {
    *used = 0;
    const uint32_t j = bad.far ? 0u : bad.good_blocks;
    if (j == 0) return ZG_OK;
    const size_t plen = stream_prefix_len(p, len, j);
    StreamRun r;
    int e = be_->run(p, plen, j, to_ring ? window : held_buf(), &r);
    if (e) return e;
    if (!clean(r, j)) { be_->discard(); return ZG_OK; }
    if ((e = be_->commit())) return e;
    account(r, plen, j, to_ring);
    if (to_ring) { if ((e = fetch_to_ring(r.out_size, true))) return e; publish_fin(); }
    else if ((e = fetch_to_buf(r.out_size))) return e;
    runs_++;
    *used = plen;
    return ZG_OK;
  }

  // ================================================ PIPE ================================================================
  struct Job { const uint8_t* p = nullptr; size_t len = 0; uint32_t nblocks = 0; bool last = false; int stop = 0; int stage = -1; };
  enum PipeState { P_RUNNING = 0, P_DONE = 1, P_STOPPED = 2, P_FAILED = 3 };
  uint8_t* ring_ = nullptr;
  uint64_t ring_cap_ = 0;
  uint64_t run_bytes_ = 0;                 // plaintext a run may produce at most
  uint32_t run_blocks_ = 0;
  std::atomic<uint64_t> pub_{0};           // bytes written to the ring and visible (stream offset, starts at 0 when the pipe starts)
  std::atomic<uint64_t> tail_{0};          // bytes the reader has taken from the ring
  std::atomic<uint64_t> hashed_{0};        // bytes the hasher is done with (== tail_ when it is off)
  uint64_t reserved_ = 0;                  // worker: bytes it has started to fetch
  std::mutex mu_;
  std::condition_variable cv_worker_, cv_reader_;
  std::deque<Job> jobs_;
  uint32_t taken_ = 0;                     // (under mu_) runs the worker has taken from the queue and not judged yet (at most two)
  bool stopping_ = false;                  // (under mu_) the worker has handed its jobs back: nothing may be queued any more
  std::atomic<int> pstate_{P_RUNNING};
  int pipe_err_ = 0;
  bool stop_req_ = false;
  std::atomic<bool> hash_stop_{false};
  bool hasher_on_ = false;
  std::thread worker_, hasher_;
  bool pipe_up_ = false;
  struct StageBuf { uint8_t* p = nullptr; size_t cap = 0; bool busy = false; };
  std::vector<StageBuf> stages_;
  std::vector<uint8_t> leftover_;          // worker -> reader: source bytes of the dropped run (behind its salvaged prefix), callback sources
  size_t leftover_slice_back_ = 0;         // ... slices: how far the source goes back
  StreamCopyPool pool_;
  uint64_t last_pump_ = 0;

  int start_pipe() {
    // ring: the last `window` bytes stay back while the frame is unfinished, so the ring holds the window + two runs
    uint64_t cap = o_.read_ahead;
    if (cap < window + (2ull << 20)) cap = window + (2ull << 20);
    if (cap < held_buf() + (8ull << 20)) cap = held_buf() + (8ull << 20);
    cap = (cap + 4095) & ~4095ull;
    run_bytes_ = (cap - window) / 2;
    if (run_bytes_ > (8191ull * kMaxBlockSize)) run_bytes_ = 8191ull * kMaxBlockSize;   // (one round of sequence chains on a 256-CU device)
    run_blocks_ = (uint32_t)(run_bytes_ / kMaxBlockSize);
    if (run_blocks_ == 0) return ZG_NOMEM;
    ring_ = (uint8_t*)be_->host_alloc(cap);
    if (!ring_) return ZG_NOMEM;
    ring_cap_ = cap;
    if (!src.is_slice()) {
      stages_.resize(3);
      const size_t scap = (size_t)o_.max_run_src + kMaxBlockSize + 16;
      for (auto& s : stages_) {
        s.p = (uint8_t*)be_->host_alloc(scap);
        if (!s.p) { free_pipe_memory(); return ZG_NOMEM; }
        s.cap = scap;
      }
    }
    int st = be_->pipe_begin(window + 3 * run_bytes_);   // (three runs: the device window is compacted without waiting for the run that is being fetched)
    if (st) { free_pipe_memory(); return st; }
    // what the inline runs left undrained becomes the ring's first content: one source of bytes from here on
    const size_t h = held_buf();
    if (h) memcpy(ring_, buf_.data() + head_, h);
    std::vector<uint8_t>().swap(buf_); head_ = 0;
    pub_.store(h, std::memory_order_relaxed); tail_.store(0, std::memory_order_relaxed); hashed_.store(0, std::memory_order_relaxed);
    reserved_ = h;
    pstate_.store(P_RUNNING, std::memory_order_relaxed);
    stop_req_ = false; hash_stop_.store(false); pipe_err_ = 0; taken_ = 0; stopping_ = false;
    hasher_on_ = o_.hash;
    mode_ = PIPE;
    pipe_up_ = true;
    try {
      if (o_.test_fail_thread == 1) throw std::bad_alloc();
      pool_.start(o_.copy_threads);
      worker_ = std::thread([this]() { worker_main(); });
      if (o_.test_fail_thread == 2) throw std::bad_alloc();
      if (hasher_on_) hasher_ = std::thread([this]() { hasher_main(); });
    } catch (...) {                       // a thread could not be started: the reference's schedule from what the ring holds
      const int st2 = stop_pipe(true);
      return st2 ? st2 : ZG_OK;
    }
    produce_jobs();
    return ZG_OK;
  }
  void free_pipe_memory() {
    if (ring_) { be_->host_free(ring_, ring_cap_); ring_ = nullptr; ring_cap_ = 0; }
    for (auto& s : stages_) if (s.p) be_->host_free(s.p, s.cap);
    stages_.clear();
  }

  // reader thread: keep two runs queued in front of the worker
  void produce_jobs() {
    for (;;) {
      int sb = -1;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (stopping_ || pstate_.load(std::memory_order_relaxed) != P_RUNNING) return;
        if (jobs_.size() + taken_ >= 3u) return;   // the one being decoded, the one that is prepared beside it, one more
        if (src.ended || src_stop_) return;
        if (!src.is_slice()) {
          for (size_t i = 0; i < stages_.size(); i++) if (!stages_[i].busy) { sb = (int)i; break; }
          if (sb < 0) return;
          stages_[sb].busy = true;
        }
      }
      StreamStage st;
      if (sb >= 0) { st.own = stages_[sb].p; st.own_cap = stages_[sb].cap; }
      Tick tk(&tus[7]);
      src.pull(pipe_ramp_ < run_blocks_ ? pipe_ramp_ : run_blocks_, src.is_slice() ? (size_t)0 : (size_t)o_.max_run_src, true, &st, nullptr);   // (a slice is uploaded from where it lies: only the block count bounds a run)
      if (pipe_ramp_ < run_blocks_) pipe_ramp_ *= 8;      // the first runs are short (the first read returns soon), the later ones as long as the budget allows
      if (st.stop) src_stop_ = true;                    // what cannot be read whole stays in the source: LOCKSTEP meets it when its turn comes
      std::lock_guard<std::mutex> lk(mu_);
      if (stopping_ || pstate_.load(std::memory_order_relaxed) != P_RUNNING) {
        // the worker stopped while this run was being taken from the source: it was never queued, so it goes back here — and what the
        // worker hands back (earlier bytes) is put in front of it when the pipe is taken down
        src.unget(st.p, st.len);
        if (st.stop) src_stop_ = false;
        if (sb >= 0) stages_[sb].busy = false;
        return;
      }
      if (st.nblocks == 0) {
        if (sb >= 0) stages_[sb].busy = false;
        if (!no_more_jobs_) { no_more_jobs_ = true; cv_worker_.notify_all(); }
        return;
      }
      Job j; j.p = st.p; j.len = st.len; j.nblocks = st.nblocks; j.last = st.last; j.stop = st.stop; j.stage = sb;
      jobs_.push_back(j);
      if (st.last || st.stop) no_more_jobs_ = true;
      cv_worker_.notify_all();
      if (no_more_jobs_) return;
    }
  }
  uint32_t pipe_ramp_ = 256;
  bool src_stop_ = false;          // the source could not yield the next whole block (its error waits for LOCKSTEP)
  bool no_more_jobs_ = false;      // (under mu_) the reader will queue nothing more

  uint64_t ring_free() const {
    const uint64_t low = hasher_on_ ? hashed_.load(std::memory_order_acquire) : tail_.load(std::memory_order_acquire);
    return ring_cap_ - (reserved_ - low);
  }
  // worker: bring the committed run's n bytes into the ring (waits for room; returns early when the reader wants to stop)
  int fetch_to_ring(uint64_t n, bool wait_done) {
    if (ring_free() < n) {
      Tick tk(&tus[4]);
      uint32_t nap = 40;                                    // (a reader that has gone away for a while must not cost a core: back off to 1 ms)
      while (ring_free() < n) {
        { std::lock_guard<std::mutex> lk(mu_); if (stop_req_) return ZG_OK; }
        std::this_thread::sleep_for(std::chrono::microseconds(nap));
        if (nap < 1000) nap *= 2;
      }
    }
    const uint64_t w = reserved_ % ring_cap_;
    const uint64_t first = n < ring_cap_ - w ? n : ring_cap_ - w;
    int e = ZG_OK;
    if (first) e = be_->fetch(ring_ + w, 0, first);
    if (!e && n > first) e = be_->fetch(ring_, first, n - first);
    if (e) return e;
    reserved_ += n;
    fetch_pending_ += n;
    if (wait_done) return land_fetch();
    return ZG_OK;
  }
  uint64_t fetch_pending_ = 0;
  int land_fetch() {
    if (!fetch_pending_) return ZG_OK;
    const int e = be_->fetch_wait();
    if (e) return e;
    { std::lock_guard<std::mutex> lk(mu_); pub_.store(pub_.load(std::memory_order_relaxed) + fetch_pending_, std::memory_order_release); }
    fetch_pending_ = 0;
    cv_reader_.notify_all();
    return ZG_OK;
  }
  void set_pstate(int s, int err) {
    { std::lock_guard<std::mutex> lk(mu_); pipe_err_ = err; pstate_.store(s, std::memory_order_release); }
    cv_reader_.notify_all();
  }

  // take the next queued run; block: wait for one (false: only if it is there). 0 = got one, 1 = none (now / ever), 2 = asked to stop
  int take_job(Job* job, bool block) {
    Tick tk(&tus[0]);
    std::unique_lock<std::mutex> lk(mu_);
    if (block) cv_worker_.wait(lk, [&]() { return stop_req_ || !jobs_.empty() || no_more_jobs_; });
    if (stop_req_) return 2;
    if (jobs_.empty()) return 1;
    *job = jobs_.front(); jobs_.pop_front();
    taken_++;
    return 0;
  }
  void worker_main() {
    be_->thread_init();
    Job cur, nxt;
    bool have_nxt = false, nxt_ready = false;
    int e = ZG_OK;
    {
      const int g = take_job(&cur, true);
      if (g == 2) return;
      if (g == 1) { set_pstate(P_STOPPED, 0); return; }     // the source ran dry in front of the frame's end before a single run
      { Tick tk(&tus[1]); e = be_->prepare(cur.p, cur.len, cur.nblocks); if (!e) e = be_->launch(window); }
      if (e) { give_back(&cur, nullptr, 0); set_pstate(P_FAILED, e); return; }
    }
    for (;;) {
      // run `cur` is on the device: meanwhile the next one is taken from the queue, walked and uploaded
      if (!have_nxt && take_job(&nxt, false) == 0) have_nxt = true;
      if (have_nxt && !nxt_ready) {
        Tick tk(&tus[1]);
        e = be_->prepare(nxt.p, nxt.len, nxt.nblocks);
        if (e) { StreamRun dummy; (void)be_->wait(&dummy); be_->discard(); give_back(&cur, &nxt, 0); set_pstate(P_FAILED, e); return; }
        nxt_ready = true;
      }
      StreamRun r;
      { Tick tk(&tus[1]); e = be_->wait(&r); }
      int e2;
      { Tick tk(&tus[2]); e2 = land_fetch(); }  // the run in front travelled to the ring meanwhile
      if (!e) e = e2;
      if (e) { be_->discard(); be_->drop_prepared(); give_back(&cur, have_nxt ? &nxt : nullptr, 0); set_pstate(P_FAILED, e); return; }
      if (!clean(r, cur.nblocks)) {
        be_->discard();
        be_->drop_prepared();
        dropped_++;
        size_t used = 0;
        e = salvage(cur.p, cur.len, r, &used, true);
        give_back(&cur, have_nxt ? &nxt : nullptr, used);
        set_pstate(e ? P_FAILED : P_STOPPED, e);
        return;
      }
      { Tick tk(&tus[3]); e = be_->commit(); }
      if (e) { be_->drop_prepared(); set_pstate(P_FAILED, e); return; }
      account(r, cur.len, cur.nblocks, true);
      runs_++;
      e = fetch_to_ring(r.out_size, false);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (cur.stage >= 0) stages_[cur.stage].busy = false;
        taken_--;
      }
      if (e) { be_->drop_prepared(); set_pstate(P_FAILED, e); return; }
      if (r.saw_last || cur.stop) {
        e = land_fetch();
        if (!e) publish_fin();
        set_pstate(e ? P_FAILED : (r.saw_last ? P_DONE : P_STOPPED), e);
        return;
      }
      if (!have_nxt) {
        const int g = take_job(&nxt, true);
        if (g == 2) break;
        if (g == 1) { e = land_fetch(); set_pstate(e ? P_FAILED : P_STOPPED, e); return; }   // nothing more will come: the source ran dry in front of the frame's end
        have_nxt = true;
      }
      if (!nxt_ready) {
        Tick tk(&tus[1]);
        e = be_->prepare(nxt.p, nxt.len, nxt.nblocks);
      }
      if (!e) { Tick tk(&tus[1]); e = be_->launch(window); }
      if (e) { be_->drop_prepared(); give_back(&nxt, nullptr, 0); set_pstate(P_FAILED, e); return; }
      cur = nxt; have_nxt = false; nxt_ready = false;
    }
    (void)land_fetch();
  }
  // the dropped run's bytes behind `used`, the run that was taken behind it (if any) and every run still queued go back to the source, in order
  void give_back(const Job* job, const Job* next, size_t used) {
    std::lock_guard<std::mutex> lk(mu_);
    stopping_ = true;                       // (in the same critical section: a run the reader is taking from the source right now is not queued any more)
    if (src.is_slice()) {
      size_t back = job->len - used;
      if (next) back += next->len;
      for (const Job& j : jobs_) back += j.len;
      leftover_slice_back_ = back;
    } else {
      leftover_.assign(job->p + used, job->p + job->len);
      if (next) leftover_.insert(leftover_.end(), next->p, next->p + next->len);
      for (const Job& j : jobs_) leftover_.insert(leftover_.end(), j.p, j.p + j.len);
    }
    jobs_.clear();
    taken_ = 0;
  }

  void hasher_main() {
    uint64_t h = 0;
    uint32_t nap = 30;
    for (;;) {
      const uint64_t t = tail_.load(std::memory_order_acquire);
      if (h < t) {
        uint64_t n = t - h;
        if (n > (4u << 20)) n = 4u << 20;
        const uint64_t w = h % ring_cap_;
        const uint64_t first = n < ring_cap_ - w ? n : ring_cap_ - w;
        hash_.update(ring_ + w, first);
        if (n > first) hash_.update(ring_, n - first);
        h += n;
        hashed_.store(h, std::memory_order_release);
        nap = 30;
        continue;
      }
      if (hash_stop_.load(std::memory_order_acquire)) {
        if (h == tail_.load(std::memory_order_acquire)) return;
        continue;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(nap));
      if (nap < 500) nap *= 2;
    }
  }

  // reader: take n bytes out of the ring
  void drain_ring(uint8_t* dst, size_t n) {
    const uint64_t t = tail_.load(std::memory_order_relaxed);
    const uint64_t w = t % ring_cap_;
    const uint64_t first = n < ring_cap_ - w ? n : ring_cap_ - w;
    if (dst) {
      if (n >= (1u << 20)) { Tick tk(&tus[6]); pool_.copy(dst, ring_ + w, first); if (n > first) pool_.copy(dst + first, ring_, n - first); }
      else { memcpy(dst, ring_ + w, first); if (n > first) memcpy(dst + first, ring_, n - first); }
    }
    tail_.store(t + n, std::memory_order_release);
  }

  int read_pipe(uint8_t* dst, size_t cap, size_t* done, bool* again) {
    *again = false;
    // a read that can never be held at once (the ring keeps the window back and is bounded) is served piece by piece
    // (while the reader waits for `want` collectable bytes the worker must still find room for a whole run: want <= a run's size)
    const uint64_t holdable = run_bytes_;
    for (;;) {
      const uint64_t a = pub_.load(std::memory_order_acquire) - tail_.load(std::memory_order_relaxed);
      const int ps = pstate_.load(std::memory_order_acquire);
      const bool fin = ps == P_DONE;
      const uint64_t a2 = fin ? pub_.load(std::memory_order_acquire) - tail_.load(std::memory_order_relaxed) : a;   // (DONE is set after the last publish)
      const uint64_t coll = fin ? a2 : (a > window ? a - window : 0);
      const size_t want = cap - *done;
      if (coll >= want || fin) {
        const size_t k = coll < want ? (size_t)coll : want;
        drain_ring(dst ? dst + *done : nullptr, k);
        *done += k;
        pump(k);
        return ZG_OK;
      }
      if (want > holdable && coll > 0) {
        drain_ring(dst ? dst + *done : nullptr, (size_t)coll);
        *done += (size_t)coll;
        pump((size_t)coll);
        continue;
      }
      if (ps == P_RUNNING) {
        produce_jobs();
        Tick tk(&tus[5]);
        std::unique_lock<std::mutex> lk(mu_);
        const uint64_t seen = pub_.load(std::memory_order_relaxed);
#ifdef ZG_STREAM_TSAN   // (ThreadSanitizer of this toolchain does not know pthread_cond_clockwait, which wait_for uses: it then reports locks that are not there)
        cv_reader_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(2),
                              [&]() { return pub_.load(std::memory_order_relaxed) != seen || pstate_.load(std::memory_order_relaxed) != P_RUNNING; });
#else
        cv_reader_.wait_for(lk, std::chrono::milliseconds(2), [&]() { return pub_.load(std::memory_order_relaxed) != seen || pstate_.load(std::memory_order_relaxed) != P_RUNNING; });
#endif
        continue;
      }
      // the worker has stopped in front of the frame's end (a run was dropped, the source ran dry, the engine failed)
      if (pub_.load(std::memory_order_acquire) - tail_.load(std::memory_order_relaxed) != a) continue;   // its last publish came in between
      const int err = pipe_err_;
      const int st = stop_pipe(true);
      if (ps == P_FAILED) return err ? err : ZG_INTERNAL;
      if (st) return st;
      *again = true;
      return ZG_OK;
    }
  }
  // every MiB or so the reader looks whether the worker can be given another run
  void pump(size_t k) {
    last_pump_ += k;
    if (last_pump_ >= (1u << 20)) { last_pump_ = 0; if (pstate_.load(std::memory_order_relaxed) == P_RUNNING) produce_jobs(); }
  }

  // Join the threads. to_lockstep: the stream goes on on the caller's thread — what the ring still holds moves to the host buffer, the
  // source gets back what was not taken, the device gets everything the reader still holds in reach again.
  int stop_pipe(bool to_lockstep) {
    if (!pipe_up_) return ZG_OK;
    { std::lock_guard<std::mutex> lk(mu_); stop_req_ = true; }
    cv_worker_.notify_all();
    if (worker_.joinable()) worker_.join();
    if (hasher_.joinable()) { hash_stop_.store(true, std::memory_order_release); hasher_.join(); }
    pool_.stop();
    hasher_on_ = false;
    pipe_up_ = false;
    int st = ZG_OK;
    if (to_lockstep) {
      // (a run the worker had fetched but not published when it was told to stop cannot exist here: it only stops by itself)
      const uint64_t t = tail_.load(std::memory_order_relaxed), p = pub_.load(std::memory_order_acquire);
      const size_t n = (size_t)(p - t);
      buf_.resize(n); head_ = 0;
      const uint64_t w = t % ring_cap_;
      const uint64_t first = n < ring_cap_ - w ? n : ring_cap_ - w;
      if (first) memcpy(buf_.data(), ring_ + w, first);
      if (n > first) memcpy(buf_.data() + first, ring_, n - first);
      if (src.is_slice()) { if (leftover_slice_back_) { src.slice_pos -= leftover_slice_back_; src.ended = false; } }
      else if (!leftover_.empty()) src.unget(leftover_.data(), leftover_.size());
      leftover_.clear(); leftover_slice_back_ = 0;
      be_->pipe_end();
      st = be_->rebase(buf_.data(), n);
      mode_ = LOCKSTEP;
    } else be_->pipe_end();
    pub_.store(0, std::memory_order_relaxed); tail_.store(0, std::memory_order_relaxed); hashed_.store(0, std::memory_order_relaxed);
    free_pipe_memory();
    return st;
  }
};

}  // namespace zg
