// TEST-ONLY: the host side of the streaming decoder (zstd-rs_amd/csrc/zg_stream.h: source handling, runs decoded ahead, the worker /
// hasher / copy threads, the ring, the fall back to the reference's block-by-block schedule) on the CPU, against a TABLE-DRIVEN
// stand-in for the engine. The stand-in knows the frame's plaintext and, per block, how many bytes it yields, whether it fails and
// whether it "sets an offset beyond the window" (tests choose these freely): a run reports what the engine would report for those
// blocks, and checks that the bytes it is handed ARE those blocks. Not part of the product, nothing here is a decoder.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../zstd-rs_amd/csrc/zg_stream.h"

using namespace zg;

namespace {

struct Table {
  std::vector<uint8_t> src;            // the frame behind its header: block headers + bodies (+ checksum)
  std::vector<uint8_t> plain;
  std::vector<uint32_t> out;           // bytes block i yields
  std::vector<uint32_t> status;        // != 0: block i fails with this
  std::vector<uint8_t> far;            // block i sets an offset beyond the window
  std::vector<size_t> off, end;        // block i's header .. body end in src
  std::vector<uint64_t> out_off;
  bool has_checksum = false;
  uint32_t checksum = 0;
};

class TableBackend : public StreamBackend {
 public:
  Table* t;
  uint32_t next = 0;                   // first block that is not committed yet
  uint64_t produced = 0;
  // the run in flight
  bool have_run = false;
  uint32_t run_blocks = 0, run_good = 0;
  uint64_t run_out = 0;
  uint64_t committed_off = 0;
  struct Pending { uint8_t* dst; uint64_t off, n; };
  std::vector<Pending> pend;
  // what the tests look at
  uint64_t nruns = 0, ncommits = 0, ndiscards = 0, nrebase = 0, rebase_bytes = 0, bad_src = 0, max_keep_gap = 0, pipe_begins = 0;
  uint64_t window = 0;
  bool in_pipe = false;
  uint64_t last_reason = 0;
  void obj(uint64_t why) { bad_src++; last_reason = why; }

  // prepare / launch / wait like the engine's backend: one run may be prepared while another is "on the device"
  const uint8_t* p_src = nullptr; size_t p_len = 0; uint32_t p_nblocks = 0; bool have_prep = false;
  StreamRun cur;
  bool launched = false;
  int prepare(const uint8_t* src, size_t len, uint32_t nblocks) override {
    if (have_prep) obj(7);
    p_src = src; p_len = len; p_nblocks = nblocks; have_prep = true;
    return ZG_OK;
  }
  void drop_prepared() override { have_prep = false; }
  int launch(uint64_t keep) override {
    if (!have_prep) { obj(8); return ZG_INTERNAL; }
    have_prep = false;
    const int e = run_now(p_src, p_len, p_nblocks, keep, &cur);
    launched = e == ZG_OK;
    return e;
  }
  int wait(StreamRun* o) override {
    if (!launched) { obj(9); return ZG_INTERNAL; }
    launched = false;
    *o = cur;
    return ZG_OK;
  }
  int run_now(const uint8_t* src, size_t len, uint32_t nblocks, uint64_t keep, StreamRun* o) {
    nruns++;
    if (have_run) obj(1);                                               // a run must be committed or dropped before the next one
    if (next + nblocks > t->out.size() || nblocks == 0) { obj(2); return ZG_INTERNAL; }
    // the bytes must be exactly those blocks (+ the checksum behind the frame's last block, if it is there)
    const size_t b0 = t->off[next], b1 = t->end[next + nblocks - 1];
    const bool is_last = (t->src[t->off[next + nblocks - 1]] & 1u) != 0;   // the Last_Block bit of the run's last block header
    const size_t want = b1 - b0;
    if (!(len == want || (is_last && t->has_checksum && len == want + 4)) || memcmp(src, t->src.data() + b0, len) != 0) obj(3);
    // what must stay in reach: at least the window (or everything there is)
    const uint64_t need = produced < window ? produced : window;
    if (keep < need) obj(4);
    o->nblocks = nblocks; o->status = 0; o->good_blocks = nblocks; o->far = false; o->out_size = 0;
    for (uint32_t i = 0; i < nblocks; i++) {
      if (t->far[next + i]) o->far = true;                              // (the engine's flag covers the whole run, also behind a failing block)
      if (!o->status && t->status[next + i]) { o->status = (int)t->status[next + i]; o->good_blocks = i; }
      if (!o->status) o->out_size += t->out[next + i];
    }
    o->saw_last = is_last && !o->status;
    o->has_checksum = o->saw_last && t->has_checksum && len == want + 4;
    o->checksum = o->has_checksum ? t->checksum : 0;
    have_run = true; run_blocks = nblocks; run_good = o->good_blocks; run_out = o->out_size;
    return ZG_OK;
  }

  int commit() override {
    if (!have_run) { obj(5); return ZG_INTERNAL; }
    have_run = false; ncommits++;
    committed_off = t->out_off[next];
    next += run_good; produced += run_out;
    return ZG_OK;
  }
  void discard() override { if (have_run) ndiscards++; have_run = false; }
  int fetch(uint8_t* dst, uint64_t off, uint64_t n) override { pend.push_back(Pending{dst, off, n}); return ZG_OK; }   // lands at fetch_wait, like a DMA
  int fetch_wait() override {
    for (const Pending& p : pend) memcpy(p.dst, t->plain.data() + committed_off + p.off, p.n);
    pend.clear();
    return ZG_OK;
  }
  int rebase(const uint8_t* held, uint64_t n) override {
    nrebase++; rebase_bytes = n;
    if (n > produced || memcmp(held, t->plain.data() + produced - n, n) != 0) obj(6);
    return ZG_OK;
  }
  int pipe_begin(uint64_t) override { pipe_begins++; in_pipe = true; return ZG_OK; }
  void pipe_end() override { in_pipe = false; }
  void* host_alloc(size_t n) override { return malloc(n); }
  void host_free(void* p, size_t) override { free(p); }
};

struct Harness {
  Table t;
  TableBackend be;
  StreamCore* core = nullptr;
  // callback source
  size_t cb_pos = 0, cb_chunk = 0;
  uint64_t cb_calls = 0;
};

size_t cb_read(void* user, uint8_t* dst, size_t n) {
  Harness* h = (Harness*)user;
  h->cb_calls++;
  size_t k = h->t.src.size() - h->cb_pos;
  if (k > n) k = n;
  if (h->cb_chunk && k > h->cb_chunk) k = h->cb_chunk;                  // a source that returns short reads
  memcpy(dst, h->t.src.data() + h->cb_pos, k);
  h->cb_pos += k;
  return k;
}

}  // namespace

extern "C" {

// src: the frame behind its header (it may be cut off, or have bytes of something else behind it: src_len says how much the SOURCE
// yields; the table describes the blocks that are whole). callback != 0: an io::Read source that returns at most chunk bytes per call.
void* zgemu_stream_new(const uint8_t* src, size_t src_len, const uint8_t* plain, size_t plain_len, const uint32_t* out, const uint32_t* status,
                       const uint8_t* far, uint32_t nblocks, uint64_t window, int has_checksum, uint32_t checksum, uint64_t content_size,
                       uint64_t read_ahead, uint64_t pipe_after, uint32_t first_run_blocks, uint32_t copy_threads, int hash, uint64_t max_run_src,
                       int callback, size_t chunk) {
  Harness* h = new Harness();
  h->t.src.assign(src, src + src_len);
  h->t.plain.assign(plain, plain + plain_len);
  h->t.out.assign(out, out + nblocks); h->t.status.assign(status, status + nblocks); h->t.far.assign(far, far + nblocks);
  h->t.has_checksum = has_checksum != 0; h->t.checksum = checksum;
  size_t p = 0;
  uint64_t oo = 0;
  for (uint32_t i = 0; i < nblocks; i++) {
    BlockHeader bh;
    if (p + 3 > src_len || read_block_header(src + p, &bh) || p + 3 + bh.content_size > src_len) { delete h; return nullptr; }
    h->t.off.push_back(p); p += 3 + bh.content_size; h->t.end.push_back(p);
    h->t.out_off.push_back(oo); oo += out[i];
  }
  h->be.t = &h->t; h->be.window = window;
  StreamOpts o;
  o.read_ahead = read_ahead; o.hash = hash != 0;
  if (pipe_after) o.pipe_after = pipe_after;
  if (first_run_blocks) o.first_run_blocks = first_run_blocks;
  o.copy_threads = copy_threads;
  if (max_run_src) o.max_run_src = max_run_src;
  if (const char* e = getenv("ZGEMU_FAIL_THREAD")) o.test_fail_thread = (uint32_t)atoi(e);
  h->core = new StreamCore(&h->be, o);
  h->core->window = window; h->core->content_size = content_size; h->core->header_bytes = 0;
  h->core->src.has_checksum = has_checksum != 0;
  if (callback) { h->core->src.read = cb_read; h->core->src.user = h; h->cb_chunk = chunk; }
  else { h->core->src.slice = h->t.src.data(); h->core->src.slice_len = h->t.src.size(); }
  return h;
}
void zgemu_stream_free(void* hp) { Harness* h = (Harness*)hp; if (!h) return; delete h->core; delete h; }
int zgemu_stream_read(void* hp, uint8_t* dst, size_t cap, size_t* n) { return ((Harness*)hp)->core->read(dst, cap, n); }
uint32_t zgemu_stream_checksum(void* hp) { return ((Harness*)hp)->core->calculated_checksum(); }
// out: [0] mode [1] runs taken [2] runs dropped [3] backend runs [4] commits [5] discards [6] rebases [7] bytes the source has given
// [8] things the stand-in objected to (wrong source bytes, a run on top of a run, too little kept) [9] is_finished [10] blocks decoded
// [11] bytes_read_from_source [12] checksum from data present [13] its value [14] pipe begins [15] source callbacks
void zgemu_stream_stats(void* hp, uint64_t* out) {
  Harness* h = (Harness*)hp;
  StreamCore* c = h->core;
  uint32_t cs = 0;
  const bool hc = c->checksum_from_data(&cs);
  const uint64_t v[16] = {(uint64_t)c->mode(), c->runs(), c->dropped_runs(), h->be.nruns, h->be.ncommits, h->be.ndiscards, h->be.nrebase,
                          c->src.is_slice() ? (uint64_t)c->src.slice_pos : (uint64_t)h->cb_pos, h->be.bad_src | (h->be.last_reason << 32), c->is_finished() ? 1u : 0u,
                          c->blocks_decoded(), c->bytes_read_from_source(), hc ? 1u : 0u, cs, h->be.pipe_begins, h->cb_calls};
  memcpy(out, v, sizeof v);
}

}  // extern "C"

// StreamCopyPool on its own: every size around the share boundaries with 1..4 helpers; returns the number of sizes that copied wrong
extern "C" int zgemu_pool_selftest(uint32_t seed) {
  int bad = 0;
  std::vector<uint8_t> src(12u << 20), dst(12u << 20);
  uint32_t x = seed * 2654435761u + 1u;
  for (auto& b : src) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); }
  for (uint32_t nt = 1; nt <= 4; nt++) {
    StreamCopyPool pool;
    pool.start(nt);
    for (uint32_t k = 0; k < 160; k++) {
      x = x * 1664525u + 1013904223u;
      const size_t unit = (size_t)((x >> 8) % 600 + 128) * 4096;            // a page-aligned share ...
      for (int d = -2; d <= 2; d++) {                                       // ... times (nt + 1), and a few bytes around it
        const size_t n = unit * (nt + 1) + (size_t)(d + 2);
        if (n > src.size()) continue;
        memset(dst.data(), 0, n);
        pool.copy(dst.data(), src.data(), n);
        if (memcmp(dst.data(), src.data(), n) != 0) bad++;
      }
    }
    pool.stop();
  }
  return bad;
}
