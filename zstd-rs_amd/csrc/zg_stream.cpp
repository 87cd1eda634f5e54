// zg_stream.cpp — StreamingDecoder mirror of the C ABI (ruzstd/src/decoding/streaming_decoder.rs:40-156) on top of zg_stream.h: the
// engine-side StreamBackend (runs of blocks through Engine::prepare_run / Batch, the frame's device window, DMA-able host memory)
// and the zgpu_streaming_* entry points of include/zgpu.h.
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>
#include "zg_capi_int.h"
#include "zg_stream.h"

using namespace zg;

namespace {

// Pinned host memory is expensive to get (the pages are faulted in and locked: ~0.1 s per few hundred MiB) and a stream needs a ring and
// staging buffers of that size: blocks are kept for the next stream of the process (bounded), not returned to the runtime.
struct PinnedCache {
  struct Blk { void* p; size_t n; };
  std::mutex mu;
  std::vector<Blk> free_;
  size_t held = 0;
  static constexpr size_t kMaxHeld = 3ull << 30;
  void* get(size_t n) {
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); i++)
        if (free_[i].n >= n && free_[i].n <= n + n / 4 + (1u << 20) && (best == free_.size() || free_[i].n < free_[best].n)) best = i;
      if (best != free_.size()) {
        void* p = free_[best].p;
        held -= free_[best].n;
        sizes_.push_back(Blk{p, free_[best].n});
        free_.erase(free_.begin() + best);
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    sizes_.push_back(Blk{p, n});
    return p;
  }
  void put(void* p) {
    size_t n = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < sizes_.size(); i++)
        if (sizes_[i].p == p) { n = sizes_[i].n; sizes_.erase(sizes_.begin() + i); break; }
      if (n && held + n <= kMaxHeld) { free_.push_back(Blk{p, n}); held += n; return; }
    }
    (void)hipHostFree(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (const Blk& b : free_) (void)hipHostFree(b.p);
    free_.clear(); held = 0;
  }
  std::vector<Blk> sizes_;   // blocks handed out (their real sizes)
};
PinnedCache g_pinned;

// Engines of streaming decoders that are not in use. A stream whose worker thread decodes ahead has an engine of its own (streams, scratch
// buffers sized to its runs); the next stream on the same device takes it over instead of allocating all of that again. Process-wide (a
// stream may outlive the context it was made from), at most two per device.
struct IdleEngines {
  std::mutex mu;
  std::vector<Engine*> v;
  Engine* take(int device) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < v.size(); i++) if (v[i]->device() == device) { Engine* e = v[i]; v.erase(v.begin() + i); return e; }
    return nullptr;
  }
  void give(Engine* e) {
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t same = 0;
      for (Engine* x : v) same += x->device() == e->device();
      if (same < 2) { v.push_back(e); return; }
    }
    delete e;
  }
  void trim() {
    std::vector<Engine*> all;
    { std::lock_guard<std::mutex> lk(mu); all.swap(v); }
    for (Engine* e : all) delete e;
  }
};
IdleEngines g_idle_engines;

class GpuStreamBackend : public StreamBackend {
 public:
  explicit GpuStreamBackend(zgpu_decoder* d) : d_(d), eng_(d->ctx->eng) {}
  ~GpuStreamBackend() override {
    delete nb_;
    delete b_;
    if (own_) g_idle_engines.give(own_);
  }

  // host walk + plan + uploads of the NEXT run. While a run is on the GPU everything travels on the upload stream beside it
  // (Engine::prepare_run side = true) and the tables that will exist are the ones that run leaves behind.
  int prepare(const uint8_t* src, size_t len, uint32_t nblocks) override {
    delete nb_; nb_ = nullptr;
    FrameState* fs = &d_->fs;
    size_t used = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const bool side = b_ != nullptr;
    const uint32_t mask = side ? b_->carry_mask_after() : fs->carry_mask;
    int st = eng_->prepare_run(src, len, fs, d_->fh.content_checksum(), nblocks, 0, &nb_, &used, side, &mask);
    if (st) { nb_ = nullptr; return st; }
    if (nb_->bb.blocks.empty()) { const int ps = nb_->parse_status; delete nb_; nb_ = nullptr; return ps ? ps : ZGPU_E_INTERNAL; }
    us_prepare += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    return ZGPU_OK;
  }
  // the prepared run goes to the GPU (the run in front has been committed or dropped)
  int launch(uint64_t keep) override {
    if (!nb_ || b_) return ZGPU_E_INTERNAL;
    FrameState* fs = &d_->fs;
    // the plaintext of the run in front may still be on its way to the host: nothing may move under it. A move is harmless when the
    // buffer is compacted in place and both what moves (the kept tail, to the front) and what this run will write behind it stay in
    // FRONT of the bytes that are being fetched — the window is reserved for three runs, so that is the usual case.
    const uint64_t bound = ((uint64_t)nb_->bb.blocks.size() + 1u) * kMaxBlockSize;   // (+ the block Batch::size_output reserves behind the run, zg_k_partial)
    if (fetching_ && fs->room_moves(bound, keep)) {
      const uint64_t k2 = keep < fs->have ? keep : fs->have;
      const bool inplace = fs->d_out.p && !fs->base && kOutFront + k2 + bound + 64 <= fs->d_out.cap;
      const bool safe = inplace && k2 + bound <= committed_base_ && fs->have - k2 >= k2;
      if (!safe) { const int w = fetch_wait(); if (w) return w; }
    }
    b_ = nb_; nb_ = nullptr;
    b_->keep_bytes = keep;
    b_->drain_rule = d_->drain_rule;
    t_launch_ = std::chrono::steady_clock::now();
    const int st = b_->run();
    if (st) { delete b_; b_ = nullptr; }
    return st;
  }
  int wait(StreamRun* out) override {
    if (!b_) return ZGPU_E_INTERNAL;
    int st = b_->sync();
    if (st) { delete b_; b_ = nullptr; return st; }
    us_kernels += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_launch_).count();
    for (int i = 0; i < ZG_T_COUNT; i++) kus[i] += (uint64_t)(b_->ms[i] * 1000.f);
    if (b_->frame_out.empty()) { delete b_; b_ = nullptr; return ZGPU_E_INTERNAL; }
    const size_t nb = b_->bb.blocks.size();
    const ZgFrameOut fo = b_->frame_out[0];          // (a failed frame: out_size ends with its last good block, Batch::sync)
    out->nblocks = (uint32_t)nb;
    // a block's own verdict (the kernels', or what the host found in its section headers); frame-layer trouble — a truncated block, a
    // missing checksum — is the source's business (StreamSrc::pull), not a failing block
    out->status = fo.status ? (int)fo.status : (b_->bb.blocks.back().host_status ? b_->parse_status : 0);
    out->good_blocks = fo.status ? fo.good_blocks : (uint32_t)nb - (b_->bb.blocks.back().host_status ? 1u : 0u);   // frame_decoder.rs:328,341,343
    out->far = b_->far_seen;
    out->out_size = fo.out_size;
    out->saw_last = b_->saw_last_block && !out->status;
    out->has_checksum = out->saw_last && !b_->info.empty() && b_->info[0].has_checksum;
    out->checksum = out->has_checksum ? b_->info[0].checksum : 0u;
    run_base_ = fo.out_base;
    return ZGPU_OK;
  }
  void drop_prepared() override { delete nb_; nb_ = nullptr; }
  int commit() override {
    if (!b_) return ZGPU_E_INTERNAL;
    const int st = b_->commit(&d_->fs);
    delete b_; b_ = nullptr;
    committed_base_ = run_base_;
    return st;
  }
  void discard() override { delete b_; b_ = nullptr; }
  int fetch(uint8_t* dst, uint64_t off, uint64_t n) override {
    if (!n) return ZGPU_OK;
    if (hipSetDevice(eng_->device()) != hipSuccess) return ZGPU_E_HIP;
    // A large download goes out in pieces on separate streams. One copy engine moves ~28 GB/s over PCIe, and whether the runtime gives a
    // single large copy one engine or more depends on what else the process has done on the device (measured: 45 GB/s in a fresh process,
    // 28 once torch or another engine had run anything — tools/dev/stream_probe2.py); separate copies get an engine each whatever ran before
    // (38 GB/s and more in every case; timing the first download and splitting only when it was slow was tried: too noisy to decide on).
    const uint8_t* src = d_->fs.out_ptr() + committed_base_ + off;
    // (three pieces: 3.5-3.8 ms of waiting per GB against 4.8 with two; a fourth would have to share a stream with kernels)
    hipStream_t ss[3] = {eng_->download_stream(), eng_->download_stream2(), eng_->upload_stream()};
    const int parts = n >= (8u << 20) ? 3 : 1;
    const uint64_t piece = ((n / parts) + 4095) & ~4095ull;
    uint64_t o = 0;
    for (int i = 0; i < parts && o < n; i++) {
      const uint64_t k = (i == parts - 1 || n - o < piece) ? n - o : piece;
      if (hipMemcpyAsync(dst + o, src + o, k, hipMemcpyDeviceToHost, ss[i]) != hipSuccess) return ZGPU_E_HIP;
      o += k;
    }
    fetching_ = true;
    return ZGPU_OK;
  }
  int fetch_wait() override {
    if (!fetching_) return ZGPU_OK;
    fetching_ = false;
    hipError_t e = hipStreamSynchronize(eng_->download_stream());
    for (hipStream_t x : {eng_->download_stream2(), eng_->upload_stream()}) { const hipError_t e2 = hipStreamSynchronize(x); if (e == hipSuccess) e = e2; }
    return e == hipSuccess ? ZGPU_OK : ZGPU_E_HIP;
  }
  int rebase(const uint8_t* held, uint64_t n) override {
    FrameState& fs = d_->fs;
    if (n <= fs.have) return ZGPU_OK;
    // [front pad][dictionary][the n most recent bytes of the frame]: the older part of what the reader still holds comes back from the host
    DevBuf nb;
    int st = nb.reserve(kOutFront + fs.base + n + (4u << 20));
    if (st) return st;
    uint8_t* np = (uint8_t*)nb.p + kOutFront;
    if ((fs.base && hipMemcpy(np, fs.out_ptr(), fs.base, hipMemcpyDeviceToDevice) != hipSuccess) ||
        hipMemcpy(np + fs.base, held, n, hipMemcpyHostToDevice) != hipSuccess) { nb.release(); return ZGPU_E_HIP; }
    fs.d_out.release();
    fs.d_out = nb;
    fs.have = n;
    return ZGPU_OK;
  }
  int pipe_begin(uint64_t window_bytes) override {
    // the worker thread gets an engine of its own (streams, scratch pool): the context's engine stays with the caller's thread
    if (!own_) {
      own_ = g_idle_engines.take(d_->ctx->eng->device());
      if (!own_) {
        const int st = Engine::create(d_->ctx->eng->device(), &own_);
        if (st) { own_ = nullptr; return st; }
      }
      own_->max_window = d_->ctx->eng->max_window;
    }
    const int st = d_->fs.reserve_window(d_->fs.base + window_bytes + (1u << 20), d_->ctx->eng->stream());
    if (st) return st;
    eng_ = own_;
    return ZGPU_OK;
  }
  void pipe_end() override {
    (void)fetch_wait();
    delete nb_; nb_ = nullptr;
    eng_ = d_->ctx->eng;
  }
  void thread_init() override { (void)hipSetDevice(eng_->device()); }
  void* host_alloc(size_t n) override { return g_pinned.get(n); }
  void host_free(void* p, size_t) override { g_pinned.put(p); }
  // diagnostics (zgpu_streaming_stats): host microseconds in prepare (walk + upload) and in run + sync, the kernels' HIP-event times summed over the runs
  uint64_t us_prepare = 0, us_kernels = 0, kus[ZG_T_COUNT] = {};

 private:
  zgpu_decoder* d_;
  Engine* eng_;
  Engine* own_ = nullptr;
  Batch* b_ = nullptr;
  uint64_t run_base_ = 0, committed_base_ = 0;
  bool fetching_ = false;
  Batch* nb_ = nullptr;                 // the run that is prepared (parsed, uploaded) and not launched yet
  std::chrono::steady_clock::time_point t_launch_;
};

}  // namespace

struct zgpu_streaming {
  zgpu_decoder* dec = nullptr;
  GpuStreamBackend* be = nullptr;
  StreamCore* core = nullptr;
  zgpu_read_fn read = nullptr;
  void* user = nullptr;
  const uint8_t* slice = nullptr;
  size_t slice_len = 0, slice_pos = 0;
  size_t take(uint8_t* dst, size_t n) {                 // header bytes: from the slice or the callback
    if (!read) {
      const size_t k = slice_len - slice_pos < n ? slice_len - slice_pos : n;
      memcpy(dst, slice + slice_pos, k);
      slice_pos += k;
      return k;
    }
    size_t got = 0;
    while (got < n) {
      const size_t r = read(user, dst + got, n - got);
      if (r == 0) break;
      got += r;
    }
    return got;
  }
};

// ---- accessors of the decoder behind a stream (zg_capi.cpp asks these while d->stream is set) ------------------------------------
bool zg_stream_is_finished(const StreamCore* c) { return c->is_finished(); }
size_t zg_stream_can_collect(const StreamCore* c) { return c->can_collect(); }
uint64_t zg_stream_blocks_decoded(const StreamCore* c) { return c->blocks_decoded(); }
uint64_t zg_stream_bytes_read(const StreamCore* c) { return c->bytes_read_from_source(); }
bool zg_stream_checksum_from_data(const StreamCore* c, uint32_t* out) { return c->checksum_from_data(out); }
uint32_t zg_stream_calculated_checksum(StreamCore* c) { return c->calculated_checksum(); }
uint64_t zg_stream_host_bytes(const StreamCore* c) { return c->host_bytes(); }
size_t zg_stream_take(StreamCore* c, uint8_t* dst, size_t n) { size_t got = 0; return (n && c->read(dst, n, &got) == ZGPU_OK) ? got : 0; }

extern "C" {

static int streaming_build2(zgpu_ctx* c, zgpu_streaming* s, const zgpu_stream_opts* opts, zgpu_streaming** out);
static int streaming_build(zgpu_ctx* c, zgpu_streaming* s, const zgpu_stream_opts* opts, zgpu_streaming** out) {
  int st;
  try { st = streaming_build2(c, s, opts, out); } catch (...) { st = ZGPU_E_NOMEM; }
  if (st) zgpu_streaming_destroy(s);       // whatever of it exists (the decoder, the backend, the core)
  return st;
}
static int streaming_build2(zgpu_ctx* c, zgpu_streaming* s, const zgpu_stream_opts* opts, zgpu_streaming** out) {
  // StreamingDecoder::new (streaming_decoder.rs:51-58): reads the frame header from the source
  int st = zgpu_decoder_create(c, &s->dec);
  if (st) return st;
  uint8_t head[18];
  size_t have = s->take(head, 5);
  if (have == 5 && head[0] == 0x28 && head[1] == 0xB5 && head[2] == 0x2F && head[3] == 0xFD) {
    const unsigned desc = head[4], single = (desc >> 5) & 1;
    static const unsigned kDid[4] = {0, 1, 2, 4}, kFcs[4] = {0, 2, 4, 8};
    const unsigned extra = (single ? 0 : 1) + kDid[desc & 3] + ((desc >> 6) == 0 ? (single ? 1 : 0) : kFcs[desc >> 6]);
    have += s->take(head + 5, extra);
  } else if (have == 5) have += s->take(head + 5, 3);   // a skippable frame's 8-byte header
  size_t used = 0;
  uint32_t sm = 0, sl = 0;
  st = zgpu_decoder_init(s->dec, head, have, &used, &sm, &sl);
  if (st) return st;
  zgpu_decoder* d = s->dec;
  StreamOpts o;
  if (opts) {
    o.read_ahead = opts->read_ahead_bytes;
    o.hash = opts->no_checksum == 0;
    if (opts->pipe_after_bytes) o.pipe_after = opts->pipe_after_bytes;
    if (opts->first_run_blocks) o.first_run_blocks = opts->first_run_blocks;
    if (opts->copy_threads) o.copy_threads = opts->copy_threads == 0xFFFFFFFFu ? 0u : opts->copy_threads;
  }
  d->hash_on = o.hash;
  s->be = new (std::nothrow) GpuStreamBackend(d);
  s->core = s->be ? new (std::nothrow) StreamCore(s->be, o) : nullptr;
  if (!s->core) return ZGPU_E_NOMEM;
  StreamCore* k = s->core;
  k->window = d->window_size;
  k->content_size = d->fh.has_fcs() ? d->fh.frame_content_size : 0;
  k->header_bytes = used;
  k->src.has_checksum = d->fh.content_checksum();
  if (s->read) { k->src.read = s->read; k->src.user = s->user; }
  else { k->src.slice = s->slice; k->src.slice_len = s->slice_len; k->src.slice_pos = s->slice_pos; }
  d->stream = k;
  *out = s;
  return ZGPU_OK;
}

int zgpu_streaming_create(zgpu_ctx* c, zgpu_read_fn read, void* user, zgpu_streaming** out) {
  return zgpu_streaming_create_ex(c, read, user, nullptr, out);
}
int zgpu_streaming_create_ex(zgpu_ctx* c, zgpu_read_fn read, void* user, const zgpu_stream_opts* opts, zgpu_streaming** out) {
  if (!c || !read || !out) return ZGPU_E_BAD_ARG;
  zgpu_streaming* s = new (std::nothrow) zgpu_streaming();
  if (!s) return ZGPU_E_NOMEM;
  s->read = read; s->user = user;
  return streaming_build(c, s, opts, out);
}
int zgpu_streaming_create_slice(zgpu_ctx* c, const uint8_t* src, size_t len, const zgpu_stream_opts* opts, zgpu_streaming** out) {
  if (!c || (!src && len) || !out) return ZGPU_E_BAD_ARG;
  zgpu_streaming* s = new (std::nothrow) zgpu_streaming();
  if (!s) return ZGPU_E_NOMEM;
  static const uint8_t kNone = 0;
  s->slice = src ? src : &kNone; s->slice_len = len;
  return streaming_build(c, s, opts, out);
}
void zgpu_streaming_destroy(zgpu_streaming* s) {
  if (!s) return;
  if (s->dec) s->dec->stream = nullptr;
  delete s->core;                                       // joins the worker threads
  delete s->be;
  zgpu_decoder_destroy(s->dec);
  delete s;
}
zgpu_decoder* zgpu_streaming_decoder(zgpu_streaming* s) { return s ? s->dec : nullptr; }   // get_ref (:66-85)
size_t zgpu_streaming_source_position(const zgpu_streaming* s) {
  if (!s || s->read) return 0;
  return s->core->src.slice_pos;
}

int zgpu_streaming_read(zgpu_streaming* s, uint8_t* dst, size_t cap, size_t* n_out) {
  // impl Read for StreamingDecoder (streaming_decoder.rs:119-155)
  if (!s || !n_out || (!dst && cap)) return ZGPU_E_BAD_ARG;
  try {
    return s->core->read(dst, cap, n_out);
  } catch (...) {          // (an allocation that failed, a thread that could not be started: nothing unwinds across the boundary)
    return ZGPU_E_NOMEM;
  }
}

// std::io::copy(&mut decoder, &mut writer) with a buffer of buf_size bytes (what the reference's CLI does with 8 KiB,
// cli/src/main.rs:142-144); writer == NULL: io::sink()
int zgpu_streaming_copy(zgpu_streaming* s, size_t buf_size, zgpu_write_fn write, void* user, uint64_t* total) {
  if (!s || !buf_size) return ZGPU_E_BAD_ARG;
  if (total) *total = 0;
  std::vector<uint8_t> own;
  uint8_t* buf = nullptr;
  void* pinned = nullptr;
  if (buf_size >= (1u << 20)) pinned = buf = (uint8_t*)g_pinned.get(buf_size);     // (large copy buffers: DMA-able, like the ring)
  if (!buf) { own.resize(buf_size); buf = own.data(); }
  uint64_t sum = 0;
  int st = ZGPU_OK;
  for (;;) {
    size_t n = 0;
    try { st = s->core->read(buf, buf_size, &n); } catch (...) { st = ZGPU_E_NOMEM; }
    if (st || n == 0) break;
    if (write) {
      size_t off = 0;
      while (off < n) {                                   // write_all
        const size_t w = write(user, buf + off, n - off);
        if (w == 0 || w > n - off) { st = ZGPU_E_BAD_ARG; break; }
        off += w;
      }
      if (st) break;
    }
    sum += n;
  }
  if (pinned) g_pinned.put(pinned);
  if (total) *total = sum;
  return st;
}

// what finished streams left behind for the next one — the worker engines with their device buffers (at most two per device), the pinned
// ring / staging memory (at most 3 GiB) — goes back to the runtime
void zgpu_release_caches(void) {
  g_idle_engines.trim();
  g_pinned.trim();
}

int zgpu_streaming_stats(const zgpu_streaming* s, uint64_t* out, int n) {
  if (!s || !out || n <= 0) return 0;
  uint64_t v[24] = {(uint64_t)s->core->mode(), s->core->runs(), s->core->dropped_runs(), s->core->host_bytes()};
  for (int i = 0; i < 8; i++) v[4 + i] = s->core->tus[i].load();
  v[12] = s->be->us_prepare; v[13] = s->be->us_kernels;
  for (int i = 0; i < ZG_T_COUNT && i < 10; i++) v[14 + i] = s->be->kus[i];
  const int k = n < 24 ? n : 24;
  for (int i = 0; i < k; i++) out[i] = v[i];
  return k;
}

}  // extern "C"
