// zg_pool.cpp — frames are independent: a host-side work queue shards them over the GPUs of one node, one worker thread +
// one engine (HIP streams, device buffers) per GPU, no data-path collective. The reference's counterpart is
// the frame loop of FrameDecoder::decode_all (ruzstd/src/decoding/frame_decoder.rs:541-577), which decodes the frames of a
// buffer one after the other.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
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

struct Staged {            // one resident submit: a job's frames concatenated
  std::vector<uint8_t> blob;
  std::vector<uint32_t> frames;   // caller's frame indices, in blob order
  uint32_t gpu = 0, lane = 0;     // which GPU of the pool; which of its two engines
  Batch* batch = nullptr;
  int status = 0;
};

}  // namespace

// the calling thread's current HIP device is left as it was found
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

struct zgpu_pool {
  std::vector<Engine*> eng;        // one per GPU: the resident (staged) submits and zgpu_pool_run
  std::vector<Engine*> eng2;       // a second engine per GPU for zgpu_pool_decode_all: the upload of one job, the kernels of another and the
                                   // download of a third overlap on their own streams (same index as eng)
  std::vector<Staged> staged;      // the resident jobs (zgpu_pool_stage): a GPU's frames are cut into up to eight of them, alternating between its engines
  std::vector<float> gpu_wall_ms;  // of the last pass, per GPU
  std::vector<uint32_t> frame_worker, frame_slot, frame_count;   // staged entries: which job, first frame of its batch, number of frames (0: skippable only)
  uint32_t nframes = 0;
  // measurement switches, read from the environment when the pool is created (nothing on the run / decode path calls getenv):
  // ZGPU_POOL_JOBS (resident jobs per GPU), ZGPU_DA_SPLIT (jobs per GPU zgpu_pool_decode_all aims at), ZGPU_DA_FLOOR_MB (its smallest job)
  uint32_t tn_pool_jobs = 0, tn_da_split = 0, tn_da_floor_mb = 0;
  // persistent workers: one thread per engine, woken per pass (no thread is created inside a timed region)
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::function<void(uint32_t)> task;   // task(worker index)
  uint64_t gen = 0;
  uint32_t pending = 0;
  bool quit = false;
  void start_workers(uint32_t n) {
    for (uint32_t w = 0; w < n; w++)
      threads.emplace_back([this, w]() {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(uint32_t)> fn;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&]() { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            fn = task;
          }
          fn(w);
          {
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
          }
        }
      });
  }
  // run fn(w) on every worker thread, return when all are done
  void run_on_workers(const std::function<void(uint32_t)>& fn) {
    std::unique_lock<std::mutex> lk(mu);
    task = fn;
    pending = (uint32_t)threads.size();
    gen++;
    cv_go.notify_all();
    cv_done.wait(lk, [&]() { return pending == 0; });
  }
  void stop_workers() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv_go.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
  }
};

extern "C" {

static int pool_build(const int* devices, int n, zgpu_pool** out) {
  DeviceGuard guard;
  zgpu_pool* p = new (std::nothrow) zgpu_pool();
  if (!p) return ZGPU_E_NOMEM;
  for (int i = 0; i < n; i++) {
    Engine *e = nullptr, *e2 = nullptr;
    int st = Engine::create(devices[i], &e);
    if (!st) st = Engine::create(devices[i], &e2);
    if (st) { delete e; for (Engine* x : p->eng) delete x; for (Engine* x : p->eng2) delete x; delete p; return st; }
    p->eng.push_back(e);
    p->eng2.push_back(e2);
  }
  p->gpu_wall_ms.assign(p->eng.size(), 0.f);
#ifdef ZG_DEV_SWITCHES   // (the development build only, like zg::Tuning)
  { const char* e = getenv("ZGPU_POOL_JOBS"); if (e && atoi(e) > 0) p->tn_pool_jobs = (uint32_t)atoi(e); }
  { const char* e = getenv("ZGPU_DA_SPLIT"); if (e && atoi(e) > 0) p->tn_da_split = (uint32_t)atoi(e); }
  { const char* e = getenv("ZGPU_DA_FLOOR_MB"); if (e && atoi(e) > 0) p->tn_da_floor_mb = (uint32_t)atoi(e); }
#endif
  p->start_workers(2u * (uint32_t)p->eng.size());   // workers [0, n): the GPUs' first engines; [n, 2n): their second ones (decode_all only)
  *out = p;
  return ZGPU_OK;
}

int zgpu_pool_create(int n_gpus, zgpu_pool** out) {
  if (!out) return ZGPU_E_BAD_ARG;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return ZGPU_E_HIP;   // no GPU: fail loudly, no CPU path
  if (n_gpus <= 0 || n_gpus > have) n_gpus = have;
  std::vector<int> dev(n_gpus);
  for (int i = 0; i < n_gpus; i++) dev[i] = i;
  return pool_build(dev.data(), n_gpus, out);
}

// one engine pair on each given device (a process that owns one GPU of the node, e.g. one rank of a torch.distributed job)
int zgpu_pool_create_on(const int* devices, int n, zgpu_pool** out) {
  if (!out || !devices || n <= 0) return ZGPU_E_BAD_ARG;
  return pool_build(devices, n, out);
}

static void pool_unstage(zgpu_pool* p) {
  for (Staged& s : p->staged) delete s.batch;
  p->staged.clear();
  p->frame_worker.clear(); p->frame_slot.clear(); p->frame_count.clear(); p->nframes = 0;
}

void zgpu_pool_destroy(zgpu_pool* p) {
  if (!p) return;
  DeviceGuard guard;
  p->stop_workers();
  pool_unstage(p);
  for (Engine* e : p->eng) delete e;
  for (Engine* e : p->eng2) delete e;
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
  DeviceGuard guard;
  pool_unstage(p);
  const uint32_t nw = (uint32_t)p->eng.size();
  std::vector<uint64_t> cost(n);
  for (uint32_t i = 0; i < n; i++) cost[i] = lens[i];
  std::vector<uint32_t> order, worker;
  std::vector<uint64_t> load;
  lpt_plan(cost.data(), n, nw, &order, &worker, &load);
  p->frame_worker.assign(n, 0);
  p->frame_slot.assign(n, 0);
  p->frame_count.assign(n, 0);
  p->nframes = n;
  // A GPU's frames are ONE resident job. ZGPU_POOL_JOBS=n (measurement) cuts them into n jobs that alternate between the GPU's two
  // engines, so that stages of different jobs run side by side. Measured on one GPU's share of the BASELINE configs (round 3):
  // text frames lose (8 GiB of 64 MiB frames: 141 -> 126 / 121 / 114 GB/s with 2 / 4 / 8 jobs: two flattens cannot share a CU's
  // LDS, and a sweep beside anything else runs at a quarter of its speed), literal-heavy frames gain 8 % with two (the Huffman
  // streams of one job beside the flatten of the other).
  const uint32_t je = p->tn_pool_jobs;
  for (uint32_t w = 0; w < nw; w++) {
    std::vector<uint32_t> mine;
    for (uint32_t i = 0; i < n; i++) if (worker[i] == w) mine.push_back(i);
    if (mine.empty()) continue;
    uint32_t J = 1;
    if (je) J = je < mine.size() ? je : (uint32_t)mine.size();
    const uint32_t first = (uint32_t)p->staged.size();
    p->staged.resize(first + J);
    for (uint32_t j = 0; j < J; j++) { p->staged[first + j].gpu = w; p->staged[first + j].lane = j & 1u; }
    // consecutive runs of frames, balanced by bytes (blob order = caller's order among a job's frames)
    uint64_t acc = 0;
    uint32_t j = 0;
    for (uint32_t i : mine) {
      while (j + 1 < J && acc >= (load[w] * (j + 1)) / J) j++;
      Staged& sj = p->staged[first + j];
      sj.frames.push_back(i);
      sj.blob.insert(sj.blob.end(), frames[i], frames[i] + lens[i]);
      p->frame_worker[i] = first + j;
      acc += lens[i];
    }
  }
  p->run_on_workers([p, nw](uint32_t w) {
    const uint32_t g = w % nw, lane = w / nw;
    for (Staged& s : p->staged) {
      if (s.gpu != g || s.lane != lane || s.frames.empty()) continue;
      Engine* eng = lane ? p->eng2[g] : p->eng[g];
      s.status = eng->prepare(s.blob.data(), s.blob.size(), &s.batch);
      if (!s.status && s.batch) s.status = s.batch->parse_status;
      std::vector<uint8_t>().swap(s.blob);      // the compressed bytes live on the device now
    }
  });
  // a caller's entry may hold several frames (skippable ones hold none): slot = index of its first frame in its job's batch
  int st = 0;
  for (Staged& s : p->staged) {
    if (s.status) { st = s.status; break; }
    uint32_t slot = 0;
    for (uint32_t i : s.frames) {
      p->frame_slot[i] = slot;
      std::vector<FrameSpan> sp;
      (void)split_frames(frames[i], lens[i], &sp);
      uint32_t c = 0;
      for (const FrameSpan& f : sp) c += f.skippable ? 0 : 1;
      p->frame_count[i] = c;
      slot += c;
    }
  }
  if (st) pool_unstage(p);      // nothing half-staged is left behind
  return st;
}

// One pass over everything staged: every GPU decodes its submit, all at once. gpu_ms[g] = kernel pipeline time of GPU g
// (HIP events); returns when all are done. *wall_ms = time from the first enqueue to the last completion.
int zgpu_pool_run(zgpu_pool* p, float* gpu_ms, float* wall_ms) {
  if (!p) return ZGPU_E_BAD_ARG;
  DeviceGuard guard;
  const uint32_t nw = (uint32_t)p->eng.size();
  auto t0 = std::chrono::steady_clock::now();
  std::vector<float> lane_ms(2 * nw, 0.f);
  auto pass = [&](uint32_t w) {
    const uint32_t g = w % nw, lane = w / nw;
    auto a = std::chrono::steady_clock::now();
    bool any = false;
    for (Staged& s : p->staged) {
      if (s.gpu != g || s.lane != lane || !s.batch) continue;
      int st = s.batch->run();
      if (!st) st = s.batch->sync();
      s.status = st;
      any = true;
    }
    if (any) lane_ms[w] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - a).count();
  };
  if (p->staged.size() == 1) pass(p->staged[0].lane * nw + p->staged[0].gpu);      // one job: on the caller's thread
  else p->run_on_workers(pass);
  if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int st = 0;
  for (uint32_t g = 0; g < nw; g++) {
    // a GPU with one job: the kernel pipeline time of that job (HIP events); with several in flight: how long its engines were at it
    uint32_t jobs = 0;
    float km = 0;
    for (const Staged& s : p->staged) if (s.gpu == g && s.batch) { jobs++; km = s.batch->ms[ZG_T_TOTAL]; }
    p->gpu_wall_ms[g] = jobs == 1 ? km : (lane_ms[g] > lane_ms[nw + g] ? lane_ms[g] : lane_ms[nw + g]);
    if (gpu_ms) gpu_ms[g] = p->gpu_wall_ms[g];
  }
  for (const Staged& s : p->staged) if (!st && s.status) st = s.status;
  return st;
}

// per-kernel times (HIP events, ms; the order of zgpu_batch_timings) of GPU g's last pass, summed over its resident jobs (jobs that
// ran side by side stretch each other: the sum then exceeds the pass), what those jobs hold, and how many they are
int zgpu_pool_timings(const zgpu_pool* p, uint32_t g, float* ms, int n, uint64_t* plain_bytes, uint64_t* comp_bytes, uint32_t* nblocks, uint32_t* njobs) {
  if (!p || g >= p->eng.size() || (!ms && n)) return ZGPU_E_BAD_ARG;
  int k = n < ZG_T_COUNT ? n : ZG_T_COUNT;
  for (int i = 0; i < k; i++) ms[i] = 0.f;
  uint64_t pb = 0, cb = 0;
  uint32_t nb = 0, nj = 0;
  for (const Staged& s : p->staged) {
    if (s.gpu != g || !s.batch) continue;
    for (int i = 0; i < k; i++) ms[i] += s.batch->ms[i];
    pb += s.batch->total_out; cb += s.batch->src_len; nb += (uint32_t)s.batch->bb.blocks.size(); nj++;
  }
  if (plain_bytes) *plain_bytes = pb;
  if (comp_bytes) *comp_bytes = cb;
  if (nblocks) *nblocks = nb;
  if (njobs) *njobs = nj;
  return ZGPU_OK;
}

// the LZ77 plan of GPU g's resident jobs (after a run): out[0] units, [1] direct units (resolved to bytes by the flatten), [2] units
// without sequences, [3] pointer-mode units (scratch words + a sweep step), [4] sweep steps, [5] plaintext bytes of the pointer-mode
// units, [6] of the direct units (both from the frames' sizes by block share: what the traffic accounting of the profiles needs)
int zgpu_pool_plan_stats(const zgpu_pool* p, uint32_t g, uint64_t* out, int n) {
  if (!p || g >= p->eng.size() || !out || n < 7) return ZGPU_E_BAD_ARG;
  for (int i = 0; i < n; i++) out[i] = 0;
  for (const Staged& s : p->staged) {
    if (s.gpu != g || !s.batch) continue;
    const zg::BatchBuilder& bb = s.batch->bb;
    out[4] += bb.steps.size();
    for (const ZgUnit& u : bb.units) {
      out[0]++;
      const ZgFrame& fr = bb.frames[u.frame];
      const uint64_t fsz = u.frame < s.batch->frame_out.size() ? s.batch->frame_out[u.frame].out_size : 0;
      const uint64_t share = fr.nblocks ? fsz * u.nblocks / fr.nblocks : 0;
      if (u.noseq & ZG_UNIT_DIRECT) { out[1]++; out[6] += share; }
      else if (u.noseq || fr.sparse) out[2]++;
      else { out[3]++; out[5] += share; }
    }
  }
  return ZGPU_OK;
}

// result of staged entry i after a run: which GPU took it, size and status of its (first) frame
int zgpu_pool_frame(zgpu_pool* p, uint32_t i, int* gpu, uint64_t* out_size, uint32_t* status) {
  if (!p || i >= p->nframes) return ZGPU_E_BAD_ARG;
  const Staged& s = p->staged[p->frame_worker[i]];
  if (gpu) *gpu = p->eng[s.gpu]->device();
  if (p->frame_count[i] == 0) { if (out_size) *out_size = 0; if (status) *status = 0; return s.batch ? ZGPU_OK : ZGPU_E_BAD_ARG; }   // skippable frames only
  if (!s.batch || p->frame_slot[i] >= s.batch->frame_out.size()) return ZGPU_E_BAD_ARG;
  // an entry is one frame or a run of frames (skippable ones hold no slot): their bytes lie back to back in the job's output, the
  // first frame that failed is the entry's verdict (until late in round 6 only the entry's FIRST frame was looked at: tools/dev/soak_pool.py)
  uint64_t size = 0;
  uint32_t st = 0;
  for (uint32_t k = 0; k < p->frame_count[i] && p->frame_slot[i] + k < s.batch->frame_out.size(); k++) {
    const ZgFrameOut& fo = s.batch->frame_out[p->frame_slot[i] + k];
    if (fo.status) { st = fo.status; break; }
    size += fo.out_size;
  }
  if (out_size) *out_size = size;
  if (status) *status = st;
  return ZGPU_OK;
}
int zgpu_pool_read(zgpu_pool* p, uint32_t i, uint8_t* dst, size_t cap, size_t* written) {
  if (!p || i >= p->nframes) return ZGPU_E_BAD_ARG;
  Staged& s = p->staged[p->frame_worker[i]];
  if (p->frame_count[i] == 0) { if (written) *written = 0; return s.batch ? ZGPU_OK : ZGPU_E_BAD_ARG; }
  if (!s.batch || p->frame_slot[i] >= s.batch->frame_out.size()) return ZGPU_E_BAD_ARG;
  DeviceGuard guard;
  uint64_t size = 0;
  for (uint32_t k = 0; k < p->frame_count[i] && p->frame_slot[i] + k < s.batch->frame_out.size(); k++) {
    const ZgFrameOut& fo = s.batch->frame_out[p->frame_slot[i] + k];
    if (fo.status) return (int)fo.status;                       // (decode_all of the entry: the first failing frame's error)
    size += fo.out_size;
  }
  if (size > cap) return ZGPU_E_TARGET_TOO_SMALL;
  int st = s.batch->read_output(s.batch->frame_out[p->frame_slot[i]].out_base, dst, size);
  if (!st && written) *written = (size_t)size;
  return st;
}

// FrameDecoder::decode_all (frame_decoder.rs:541-577) over all GPUs of the pool: the buffer is cut into frames on the host
// (frame + block headers only), runs of consecutive frames become jobs, the jobs are queued largest first and pulled by two
// workers per GPU, each with its own engine: while one job's kernels run, the other engine uploads the next job or downloads
// the previous one (H2D, kernels and D2H overlap when src and dst are pinned host memory; pageable buffers work, but their copies
// are staged by the runtime). The plaintext is written back to back in input order. When every frame declares its
// Frame_Content_Size — what every libzstd frame does — a job's place in dst is known before it is decoded: it is copied out and
// its device memory released as soon as it is done (device memory is proportional to the jobs in flight, not to the input).
// Otherwise the outputs wait on the GPUs — without their scratch — until all sizes are known.
int zgpu_pool_decode_all(zgpu_pool* p, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* written) {
  if (!p || !written || (!src && len) || (!dst && cap)) return ZGPU_E_BAD_ARG;
  *written = 0;
  DeviceGuard guard;
  std::vector<FrameSpan> spans;
  const int walk = split_frames(src, len, &spans);     // frames in front of a malformed one are still decoded; the error wins below
  struct Job { uint64_t begin, end; Batch* batch = nullptr; int status = 0; uint32_t worker = 0; uint64_t out_off = 0, out_size = 0, want_size = 0; bool placed = false; };
  std::vector<Job> jobs;
  const uint64_t nw = p->eng.size();
  // jobs: about sixteen per GPU (eight per engine) so that the copies of one overlap the kernels of another, but not below 32 MiB of
  // input (a submit costs ~2 ms whatever its size: the length of one block's sequence chain) nor above 512 MiB. Measured on 1 GiB of
  // 64 MiB frames (tools/dev/e2e.py): 4 / 8 / 16 jobs per GPU 30.5 / 36.1 / 40.5 GB/s, 16 with a 16 MiB floor 37.4
  uint64_t per_gpu = 16, floor_mb = 32;
  if (p->tn_da_split) per_gpu = p->tn_da_split;         // (measurement) jobs per GPU aimed at
  if (p->tn_da_floor_mb) floor_mb = p->tn_da_floor_mb;  // (measurement) smallest job, MiB of input
  uint64_t kJob = (uint64_t)len / (per_gpu * nw);
  if (kJob < (floor_mb << 20)) kJob = floor_mb << 20;
  if (kJob > (512ull << 20)) kJob = 512ull << 20;
  bool sizes_known = true;
  for (const FrameSpan& s : spans) {
    if (!s.skippable && !s.has_content_size) sizes_known = false;
    if (!jobs.empty() && jobs.back().end - jobs.back().begin < kJob && s.end - s.begin < kJob) jobs.back().end = s.end;
    else { Job j; j.begin = s.begin; j.end = s.end; jobs.push_back(j); }
    if (!s.skippable) jobs.back().want_size += s.content_size;
  }
  if (walk) {
    // the walk stopped inside a frame (or at a header it could not read): what that frame holds in front of the stop is decoded as well —
    // the reference meets an error in one of THOSE blocks first (zgpu_decode_all; found by tools/dev/soak_concat.py: the pool answered
    // with the walk's error). A job of its own behind the whole frames: its prepare() stops where this walk did.
    const uint64_t tail = spans.empty() ? 0 : spans.back().end;
    if (tail < len) { Job j; j.begin = tail; j.end = len; jobs.push_back(j); }
  }
  uint64_t want_total = 0;
  for (Job& j : jobs) { j.out_off = want_total; want_total += j.want_size; }
  const bool direct_out = sizes_known && walk == 0 && want_total <= cap;
  std::vector<uint32_t> order(jobs.size());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].end - jobs[a].begin > jobs[b].end - jobs[b].begin; });
  std::atomic<uint32_t> next(0);
  p->run_on_workers([&](uint32_t w) {
    Engine* eng = w < nw ? p->eng[w] : p->eng2[w - nw];
    // the download of a finished job runs on the engine's third stream while the worker is already at its next job: the wait for
    // it comes one job later (H2D of job k+1 and D2H of job k use different DMA directions; the kernels in between hide both)
    Job* downloading = nullptr;
    auto land = [&]() {
      if (!downloading) return;
      if (hipStreamSynchronize(eng->download_stream()) != hipSuccess && !downloading->status) downloading->status = ZGPU_E_HIP;
      if (hipStreamSynchronize(eng->download_stream2()) != hipSuccess && !downloading->status) downloading->status = ZGPU_E_HIP;
      if (hipStreamSynchronize(eng->upload_stream()) != hipSuccess && !downloading->status) downloading->status = ZGPU_E_HIP;
      delete downloading->batch; downloading->batch = nullptr;
      downloading = nullptr;
    };
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= order.size()) break;
      Job& j = jobs[order[k]];
      j.worker = w;
      j.status = eng->prepare(src + j.begin, (size_t)(j.end - j.begin), &j.batch);
      // (a walk that stopped inside the job: what lies in front of that point is decoded first — an error there is the one the
      //  reference meets first, zgpu_decode_all)
      const int jwalk = (!j.status && j.batch) ? j.batch->parse_status : 0;
      if (jwalk && j.batch->bb.blocks.empty()) j.status = jwalk;
      if (!j.status) { j.batch->drain_rule = ZG_DRAIN_DECODE_ALL; j.status = j.batch->run(); }
      if (!j.status) j.status = j.batch->sync();
      if (!j.status)
        for (const ZgFrameOut& fo : j.batch->frame_out)
          if (fo.status) { j.status = (int)fo.status; break; }
      if (!j.status && jwalk) j.status = jwalk;
      if (!j.status) j.out_size = j.batch->total_out;
      if (!j.status && direct_out && j.out_size == j.want_size) {
        land();                            // (at most one download in flight per engine: its buffers are this job's predecessor's)
        // (three pieces on three streams: a single copy may get one copy engine, ~28 GB/s, or more, depending on what the process has done on the
        //  device before; two copies get one each — see GpuStreamBackend::fetch, zg_stream.cpp)
        hipStream_t ds[3] = {eng->download_stream(), eng->download_stream2(), eng->upload_stream()};
        const int parts = j.out_size >= (8u << 20) ? 3 : 1;
        const uint64_t piece = ((j.out_size / parts) + 4095) & ~4095ull;
        uint64_t o = 0;
        for (int i = 0; i < parts && o < j.out_size && !j.status; i++) {
          const uint64_t k = (i == parts - 1 || j.out_size - o < piece) ? j.out_size - o : piece;
          j.status = j.batch->read_output_async(o, dst + j.out_off + o, k, ds[i]);
          o += k;
        }
        j.placed = true;
        downloading = &j;
      } else if (j.batch) {
        j.batch->release_scratch();        // only the plaintext stays on the device until its place is known
      }
    }
    land();
  });
  int st = 0;
  uint64_t total = 0;
  bool all_placed = true;
  for (Job& j : jobs) {                                  // the first error in input order is the one the reference would return
    if (j.status) { st = j.status; break; }
    if (!j.placed || j.out_off != total) all_placed = false;
    total += j.out_size;
  }
  if (!st && walk) st = walk;
  if (!st && total > cap) st = ZGPU_E_TARGET_TOO_SMALL;
  if (!st && !all_placed) {
    // sizes were not declared, or a frame lied about its size: every job's final place is known only now. Jobs that were copied
    // out already move inside dst FIRST, before anything else is written (a download into the gap in front of a placed job may be
    // longer than the gap): the ones that move towards the end last to first, the ones that move towards the front first to
    // last — their final places are disjoint and in input order, so no move overwrites bytes another move still has to read.
    std::vector<uint64_t> fin(jobs.size());
    uint64_t off = 0;
    for (size_t k = 0; k < jobs.size(); k++) { fin[k] = off; off += jobs[k].out_size; }
    for (size_t k = jobs.size(); k-- > 0;)
      if (jobs[k].placed && fin[k] > jobs[k].out_off) memmove(dst + fin[k], dst + jobs[k].out_off, jobs[k].out_size);
    for (size_t k = 0; k < jobs.size(); k++)
      if (jobs[k].placed && fin[k] < jobs[k].out_off) memmove(dst + fin[k], dst + jobs[k].out_off, jobs[k].out_size);
    for (size_t k = 0; k < jobs.size(); k++) {
      Job& j = jobs[k];
      if (j.placed || !j.batch) continue;
      const int c = j.batch->read_output(0, dst + fin[k], j.out_size);
      if (c && !st) st = c;
    }
  }
  for (Job& j : jobs) delete j.batch;
  if (!st) *written = (size_t)total;
  return st;
}

}  // extern "C"
