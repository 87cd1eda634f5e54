// zg_pool.cpp — frames are independent: a host-side work queue shards them over the GPUs of one node, one worker thread +
// one engine (HIP streams, device buffers) per GPU, no data-path collective. The reference's counterpart is
// the frame loop of FrameDecoder::decode_all (ruzstd/src/decoding/frame_decoder.rs:541-577), which decodes the frames of a
// buffer one after the other.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>
#include <thread>
#include <vector>
#include "../../include/zgpu.h"
#include "zg_engine.h"

using namespace zg;

namespace {

// Longest-processing-time-first: jobs in descending cost, each to the worker that is free first. This is what a dynamic
// pull from a queue sorted by size converges to when time is proportional to cost; it is also the static plan.
void lpt_plan(const uint64_t* cost, uint32_t n, uint32_t nw, std::vector<uint32_t>* order, std::vector<uint32_t>* worker, std::vector<uint64_t>* load) {
  order->resize(n);
  for (uint32_t i = 0; i < n; i++) (*order)[i] = i;
  std::stable_sort(order->begin(), order->end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  worker->assign(n, 0);
  load->assign(nw ? nw : 1, 0);
  for (uint32_t k = 0; k < n; k++) {
    uint32_t best = 0;
    for (uint32_t w = 1; w < nw; w++)
      if ((*load)[w] < (*load)[best]) best = w;
    (*worker)[(*order)[k]] = best;
    (*load)[best] += cost[(*order)[k]];
  }
}

struct Staged {            // one worker's resident submit: its frames concatenated
  std::vector<uint8_t> blob;
  std::vector<uint32_t> frames;   // caller's frame indices, in blob order
  Batch* batch = nullptr;
  int status = 0;
  float kernel_ms = 0, wall_ms = 0;
};

}  // namespace

struct zgpu_pool {
  std::vector<Engine*> eng;
  std::vector<Staged> staged;
  std::vector<uint32_t> frame_worker, frame_slot;   // staged frames: which worker, which frame of its batch
  uint32_t nframes = 0;
};

extern "C" {

int zgpu_pool_create(int n_gpus, zgpu_pool** out) {
  if (!out) return ZGPU_E_BAD_ARG;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return ZGPU_E_HIP;   // no GPU: fail loudly, no CPU path
  if (n_gpus <= 0 || n_gpus > have) n_gpus = have;
  zgpu_pool* p = new (std::nothrow) zgpu_pool();
  if (!p) return ZGPU_E_NOMEM;
  for (int i = 0; i < n_gpus; i++) {
    Engine* e = nullptr;
    int st = Engine::create(i, &e);
    if (st) { zgpu_pool_destroy(p); return st; }
    p->eng.push_back(e);
  }
  p->staged.resize(p->eng.size());
  *out = p;
  return ZGPU_OK;
}

// one engine on a given device (a process that owns one GPU of the node, e.g. one rank of a torch.distributed job)
int zgpu_pool_create_on(const int* devices, int n, zgpu_pool** out) {
  if (!out || !devices || n <= 0) return ZGPU_E_BAD_ARG;
  zgpu_pool* p = new (std::nothrow) zgpu_pool();
  if (!p) return ZGPU_E_NOMEM;
  for (int i = 0; i < n; i++) {
    Engine* e = nullptr;
    int st = Engine::create(devices[i], &e);
    if (st) { zgpu_pool_destroy(p); return st; }
    p->eng.push_back(e);
  }
  p->staged.resize(p->eng.size());
  *out = p;
  return ZGPU_OK;
}

static void pool_unstage(zgpu_pool* p) {
  for (Staged& s : p->staged) { delete s.batch; s = Staged(); }
  p->frame_worker.clear(); p->frame_slot.clear(); p->nframes = 0;
}

void zgpu_pool_destroy(zgpu_pool* p) {
  if (!p) return;
  pool_unstage(p);
  for (Engine* e : p->eng) delete e;
  delete p;
}
int zgpu_pool_num_gpus(const zgpu_pool* p) { return p ? (int)p->eng.size() : 0; }

int zgpu_pool_plan(const uint64_t* cost, uint32_t n, uint32_t n_workers, uint32_t* order_out, uint32_t* worker_out, uint64_t* load_out) {
  if ((!cost && n) || n_workers == 0) return ZGPU_E_BAD_ARG;
  std::vector<uint32_t> order, worker;
  std::vector<uint64_t> load;
  lpt_plan(cost, n, n_workers, &order, &worker, &load);
  for (uint32_t i = 0; i < n; i++) { if (order_out) order_out[i] = order[i]; if (worker_out) worker_out[i] = worker[i]; }
  if (load_out) for (uint32_t w = 0; w < n_workers; w++) load_out[w] = load[w];
  return ZGPU_OK;
}

// Stage a set of frames: LPT assignment by compressed size, one resident submit per GPU (host block walk + H2D here,
// outside any timed region). Each entry of `frames` is one zstd frame (or a run of concatenated frames).
int zgpu_pool_stage(zgpu_pool* p, const uint8_t* const* frames, const size_t* lens, uint32_t n) {
  if (!p || (!frames && n) || (!lens && n)) return ZGPU_E_BAD_ARG;
  pool_unstage(p);
  const uint32_t nw = (uint32_t)p->eng.size();
  std::vector<uint64_t> cost(n);
  for (uint32_t i = 0; i < n; i++) cost[i] = lens[i];
  std::vector<uint32_t> order, worker;
  std::vector<uint64_t> load;
  lpt_plan(cost.data(), n, nw, &order, &worker, &load);
  p->frame_worker = worker;
  p->frame_slot.assign(n, 0);
  p->nframes = n;
  for (uint32_t w = 0; w < nw; w++) p->staged[w].blob.reserve(load[w]);
  for (uint32_t i = 0; i < n; i++) {             // blob order = caller's order among a worker's frames
    Staged& s = p->staged[worker[i]];
    s.frames.push_back(i);
    s.blob.insert(s.blob.end(), frames[i], frames[i] + lens[i]);
  }
  std::vector<std::thread> th;
  for (uint32_t w = 0; w < nw; w++)
    th.emplace_back([p, w]() {
      Staged& s = p->staged[w];
      if (s.frames.empty()) return;
      s.status = p->eng[w]->prepare(s.blob.data(), s.blob.size(), &s.batch);
      if (!s.status && s.batch) s.status = s.batch->parse_status;
    });
  for (auto& t : th) t.join();
  // a caller's entry may hold several frames (skippable ones hold none): slot = index of its first frame in the worker's batch
  for (uint32_t w = 0; w < nw; w++) {
    Staged& s = p->staged[w];
    if (s.status) return s.status;
    uint32_t slot = 0;
    for (uint32_t i : s.frames) {
      p->frame_slot[i] = slot;
      std::vector<FrameSpan> sp;
      (void)split_frames(frames[i], lens[i], &sp);
      for (const FrameSpan& f : sp) slot += f.skippable ? 0 : 1;
    }
  }
  return ZGPU_OK;
}

// One pass over everything staged: every GPU decodes its submit, all at once. gpu_ms[g] = kernel pipeline time of GPU g
// (HIP events); returns when all are done. *wall_ms = time from the first enqueue to the last completion.
int zgpu_pool_run(zgpu_pool* p, float* gpu_ms, float* wall_ms) {
  if (!p) return ZGPU_E_BAD_ARG;
  const uint32_t nw = (uint32_t)p->eng.size();
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (uint32_t w = 0; w < nw; w++)
    th.emplace_back([p, w]() {
      Staged& s = p->staged[w];
      if (!s.batch) return;
      int st = s.batch->run();
      if (!st) st = s.batch->sync();
      s.status = st;
      s.kernel_ms = s.batch->ms[ZG_T_TOTAL];
    });
  for (auto& t : th) t.join();
  if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int st = 0;
  for (uint32_t w = 0; w < nw; w++) {
    if (gpu_ms) gpu_ms[w] = p->staged[w].kernel_ms;
    if (!st && p->staged[w].status) st = p->staged[w].status;
  }
  return st;
}

// result of staged entry i after a run: which GPU took it, size and status of its (first) frame
int zgpu_pool_frame(zgpu_pool* p, uint32_t i, int* gpu, uint64_t* out_size, uint32_t* status) {
  if (!p || i >= p->nframes) return ZGPU_E_BAD_ARG;
  const Staged& s = p->staged[p->frame_worker[i]];
  if (!s.batch || p->frame_slot[i] >= s.batch->frame_out.size()) return ZGPU_E_BAD_ARG;
  const ZgFrameOut& fo = s.batch->frame_out[p->frame_slot[i]];
  if (gpu) *gpu = p->eng[p->frame_worker[i]]->device();
  if (out_size) *out_size = fo.out_size;
  if (status) *status = fo.status;
  return ZGPU_OK;
}
int zgpu_pool_read(zgpu_pool* p, uint32_t i, uint8_t* dst, size_t cap, size_t* written) {
  if (!p || i >= p->nframes) return ZGPU_E_BAD_ARG;
  Staged& s = p->staged[p->frame_worker[i]];
  if (!s.batch || p->frame_slot[i] >= s.batch->frame_out.size()) return ZGPU_E_BAD_ARG;
  const ZgFrameOut& fo = s.batch->frame_out[p->frame_slot[i]];
  if (fo.status) return (int)fo.status;
  if (fo.out_size > cap) return ZGPU_E_TARGET_TOO_SMALL;
  int st = s.batch->read_output(fo.out_base, dst, fo.out_size);
  if (!st && written) *written = (size_t)fo.out_size;
  return st;
}

// FrameDecoder::decode_all (frame_decoder.rs:541-577) over all GPUs of the pool: the buffer is cut into frames on the host
// (frame + block headers only), runs of consecutive frames become jobs of >= 64 MiB of input (see below; or a single larger frame), the
// jobs are queued largest first and pulled by one worker per GPU; the plaintext is written back to back in input order.
int zgpu_pool_decode_all(zgpu_pool* p, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* written) {
  if (!p || !written || (!src && len) || (!dst && cap)) return ZGPU_E_BAD_ARG;
  *written = 0;
  std::vector<FrameSpan> spans;
  const int walk = split_frames(src, len, &spans);     // frames in front of a malformed one are still decoded; the error wins below
  struct Job { uint64_t begin, end; Batch* batch = nullptr; int status = 0; uint32_t gpu = 0; uint64_t out_off = 0, out_size = 0; };
  std::vector<Job> jobs;
  // A submit costs ~2 ms whatever its size (the length of one block's sequence chain), so jobs are as large as balance allows:
  // about four per GPU when there are several GPUs, one otherwise; at most 1 GiB of input (~25 GB of device memory while it runs).
  const uint64_t nw_ = p->eng.size();
  uint64_t kJob = nw_ > 1 ? (uint64_t)len / (4 * nw_) : (uint64_t)len;
  if (kJob < (64ull << 20)) kJob = 64ull << 20;
  if (kJob > (1ull << 30)) kJob = 1ull << 30;
  for (const FrameSpan& s : spans) {
    if (!jobs.empty() && jobs.back().end - jobs.back().begin < kJob && s.end - s.begin < kJob) jobs.back().end = s.end;
    else { Job j; j.begin = s.begin; j.end = s.end; jobs.push_back(j); }
  }
  std::vector<uint32_t> order(jobs.size());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].end - jobs[a].begin > jobs[b].end - jobs[b].begin; });
  std::atomic<uint32_t> next(0);
  const uint32_t nw = (uint32_t)p->eng.size();
  std::vector<std::thread> th;
  for (uint32_t w = 0; w < nw; w++)
    th.emplace_back([&, w]() {
      for (;;) {
        const uint32_t k = next.fetch_add(1);
        if (k >= order.size()) break;
        Job& j = jobs[order[k]];
        j.gpu = w;
        j.status = p->eng[w]->prepare(src + j.begin, (size_t)(j.end - j.begin), &j.batch);
        if (!j.status && j.batch->parse_status) j.status = j.batch->parse_status;
        if (!j.status) j.status = j.batch->run();      // outputs stay on the GPU until every job's size is known
        if (!j.status) j.status = j.batch->sync();
        if (!j.status)
          for (const ZgFrameOut& fo : j.batch->frame_out)
            if (fo.status) { j.status = (int)fo.status; break; }
        if (!j.status) j.out_size = j.batch->total_out;
      }
    });
  for (auto& t : th) t.join();
  int st = 0;
  uint64_t total = 0;
  for (Job& j : jobs) {                                  // the first error in input order is the one the reference would return
    if (j.status) { st = j.status; break; }
    j.out_off = total;
    total += j.out_size;
  }
  if (!st && walk) st = walk;
  if (!st && total > cap) st = ZGPU_E_TARGET_TOO_SMALL;
  if (!st) {
    th.clear();
    std::vector<int> cst(nw, 0);
    for (uint32_t w = 0; w < nw; w++)
      th.emplace_back([&, w]() {
        for (Job& j : jobs)
          if (j.gpu == w && j.batch && !cst[w]) cst[w] = j.batch->read_output(0, dst + j.out_off, j.out_size);
      });
    for (auto& t : th) t.join();
    for (int c : cst) if (c && !st) st = c;
  }
  for (Job& j : jobs) delete j.batch;
  if (!st) *written = (size_t)total;
  return st;
}

}  // extern "C"
