// chain_fill_bench.hip — the whole split sweep of one long frame, reduced to its shape: N units of HEAD + TAIL bytes; a unit's bytes are
// gathered from the TAIL of the unit in front of it (tools/dev/chain_bench.hip has the tails alone). Question: does ONE
// persistent launch whose tail workgroups form the chain and whose head workgroups fill the gaps beat today's two streams of launches?
//   L    tails: one launch per step; heads: one launch per GS units on a second stream behind an event (the engine today)
//   P0   tails: one persistent launch (sharded arrival counters, 16-byte sc1 stores, sc1 gathers, scratch prefetch); heads: one launch after it
//   P1   one persistent launch; workgroups take a role ticket: the first NT arrivals are the chain (so every chain workgroup is resident before
//        any head workgroup can wait for one), the rest take head items of unit u as soon as step u-1 has arrived: filler between the steps
// Every output is checked against a host model; every spin is bounded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr uint32_t TAIL = 2u << 20, HEAD = 232u * 8192u, US = HEAD + TAIL, GT = TAIL / 4, GH = HEAD / 4, SH = 16, SHS = 32, OOB = 0xFFFFFFFFu;
struct P { uint8_t* dst; const v4u* ogt; const v4u* ogh; uint32_t* cnt; uint32_t* abort_; uint32_t* ticket; uint32_t n, flags, nt; };

__host__ __device__ inline uint64_t mix(uint64_t i) { uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
// group g (of the head: kind 0, of the tail: kind 1) of unit u: four effective offsets (0: literal byte) and the literal values
__host__ __device__ inline void group_of(uint32_t kind, uint32_t u, uint32_t g, uint32_t e[4], uint8_t lit[4]) {
  const uint64_t r = mix(((uint64_t)u * 2 + kind) * GT + g);
  const uint32_t a = (uint32_t)(r & 0xFFFFFF) % (TAIL - 8), b = (uint32_t)((r >> 24) & 0xFFFFFF) % (TAIL - 8), c = (uint32_t)(r >> 48) & 7u;
  const uint32_t pos = (kind ? HEAD : 0u) + 4 * g;            // unit-relative position of the group
  for (uint32_t i = 0; i < 4; i++) {
    const bool isl = u == 0 || ((r >> (52 + 3 * i)) & 7u) == 0;
    e[i] = isl ? 0u : US + pos + i - (HEAD + (i < c ? a : b) + i);
    lit[i] = (uint8_t)(mix(((uint64_t)(u * 2 + kind) << 40) | ((uint64_t)g << 2) | i) >> 13);
  }
}
__global__ void k_init_og(v4u* ogt, v4u* ogh, uint32_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t e[4]; uint8_t l[4];
  if (i < (uint64_t)n * GT) { group_of(1, (uint32_t)(i / GT) + 1, (uint32_t)(i % GT), e, l); ogt[i] = (v4u){e[0], e[1], e[2], e[3]}; }
  if (i < (uint64_t)n * GH) { group_of(0, (uint32_t)(i / GH) + 1, (uint32_t)(i % GH), e, l); ogh[i] = (v4u){e[0], e[1], e[2], e[3]}; }
}
__global__ void k_init_dst(uint8_t* dst, uint32_t n) {        // literal bytes in place (the flatten wrote them), everything else poisoned
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)(n + 1) * (US / 4)) return;
  const uint32_t u = (uint32_t)(i / (US / 4)), gg = (uint32_t)(i % (US / 4)), kind = gg >= GH, g = kind ? gg - GH : gg;
  uint32_t e[4]; uint8_t l[4];
  group_of(kind, u, g, e, l);
  uint32_t v = 0;
  for (int k = 0; k < 4; k++) v |= (uint32_t)(e[k] ? 0xEEu : l[k]) << (8 * k);
  *(uint32_t*)(dst + (uint64_t)u * US + 4 * gg) = v;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <bool SC1> __device__ __forceinline__ v2u ld64(__amdgpu_buffer_rsrc_t rs, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, SC1 ? 16 : 0); }

// one item: T*B groups from g0 of a region (og: its scratch words, base: byte offset of its group 0 in dst, ng: its groups)
template <int B> __device__ __forceinline__ void load_og(const v4u* og, uint32_t g0, uint32_t ng, v4u (&o)[B]) {
  const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < B; k++) { const uint32_t g = g0 + w * 64 * B + k * 64 + l; o[k] = og[g < ng ? g : 0]; }
}
template <int B, bool SC1, bool ST16> __device__ __forceinline__ void item(__amdgpu_buffer_rsrc_t rs, uint32_t base, uint32_t g0, uint32_t ng, const v4u (&o)[B], uint32_t* lds) {
  const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
  v2u rA[B], rB[B], rC[B], rD[B], rW[B];
#pragma unroll
  for (int k = 0; k < B; k++) {
    const v4u q = o[k];
    const uint32_t g = g0 + w * 64 * B + k * 64 + l, wrel = base + 4 * g;
    const bool ux = q.x != 0, uy = q.y != 0, uz = q.z != 0, uw = q.w != 0, all = ux && uy && uz && uw;
    const bool nD = uw && !(ux && q.w == q.x), nB = uy && !(ux && q.y == q.x) && !(uw && q.y == q.w);
    const bool nC = uz && !(ux && q.z == q.x) && !(uw && q.z == q.w) && !(uy && q.z == q.y);
    rA[k] = ld64<SC1>(rs, ux ? (wrel - q.x) & ~3u : OOB);
    rD[k] = ld64<SC1>(rs, nD ? (wrel - q.w) & ~3u : OOB);
    rB[k] = ld64<SC1>(rs, nB ? (wrel - q.y) & ~3u : OOB);
    rC[k] = ld64<SC1>(rs, nC ? (wrel - q.z) & ~3u : OOB);
    rW[k] = ld64<false>(rs, (!all && g < ng) ? wrel : OOB);
  }
#pragma unroll
  for (int k = 0; k < B; k++) {
    const v4u q = o[k];
    const uint32_t g = g0 + w * 64 * B + k * 64 + l;
    const bool ux = q.x != 0, uy = q.y != 0, uz = q.z != 0, uw = q.w != 0;
    auto fun = [&](const v2u r, uint32_t e) { return __builtin_amdgcn_alignbit(r.y, r.x, ((0u - e) & 3u) * 8u); };
    const uint32_t lA = fun(rA[k], q.x), lD = fun(rD[k], q.w), lB = fun(rB[k], q.y), lC = fun(rC[k], q.z), lW = rW[k].x;
    const uint32_t sw = (ux && q.w == q.x) ? lA : lD;
    const uint32_t sy = (ux && q.y == q.x) ? lA : (uw && q.y == q.w) ? sw : lB;
    const uint32_t sz = (ux && q.z == q.x) ? lA : (uw && q.z == q.w) ? sw : (uy && q.z == q.y) ? sy : lC;
    const uint32_t v = ((ux ? lA : lW) & 0xFFu) | ((uy ? sy : lW) & 0xFF00u) | ((uz ? sz : lW) & 0xFF0000u) | ((uw ? sw : lW) & 0xFF000000u);
    if (ST16) lds[w * 64 * B + k * 64 + l] = v;
    else __builtin_amdgcn_raw_buffer_store_b32(v, rs, g < ng ? base + 4 * g : OOB, 0, 0);
  }
  if (ST16) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (l < 16 * B) {
      const v4u v = *(const v4u*)(lds + w * 64 * B + 4 * l);
      const uint32_t g = g0 + w * 64 * B + 4 * l;
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, g < ng ? base + 4 * g : OOB, 0, SC1 ? 16 : 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
}

// ---- L: plain launches (256 threads x 2 groups, as zg_k_sweep)
__global__ void __launch_bounds__(256) k_tail(P p, uint32_t u) {
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(p.dst, (p.n + 1) * US);
  v4u o[2];
  load_og<2>(p.ogt + (size_t)(u - 1) * GT, blockIdx.x * 512, GT, o);
  item<2, false, false>(rs, u * US + HEAD, blockIdx.x * 512, GT, o, nullptr);
}
__global__ void __launch_bounds__(256) k_heads(P p, uint32_t u0) {      // blockIdx.y: unit u0 + y
  extern __shared__ uint32_t unused_lds[];                               // (the engine's trick: an LDS allocation keeps the heads to a few workgroups per CU)
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(p.dst, (p.n + 1) * US);
  const uint32_t u = u0 + blockIdx.y;
  v4u o[2];
  load_og<2>(p.ogh + (size_t)(u - 1) * GH, blockIdx.x * 512, GH, o);
  item<2, false, false>(rs, u * US, blockIdx.x * 512, GH, o, nullptr);
}

// ---- P: the persistent form. 512 threads x 4 groups = 8 KiB items: NT = 256 chain workgroups cover a tail, 232 head items cover a head
constexpr int PT = 512, PB = 4;
template <int SLEEP> __device__ __forceinline__ bool wait_step(const P& p, uint32_t s, uint32_t m) {   // wave 0: every chain slot of step s stored and drained? false: gave up
  const uint32_t t = threadIdx.x;
  const uint32_t want = (t < SH && m > t) ? (m - t + SH - 1) / SH : 0u;
  const uint32_t* q = t == 63 ? p.abort_ : p.cnt + (s * SH + (t < SH ? t : 0)) * SHS;
  for (uint32_t spins = 0;; spins++) {
    const uint32_t v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__any(t == 63 && v != 0)) return false;
    if (__all(t == 63 || v >= want)) return true;
    if (spins > (1u << 15)) { if (t == 0) __hip_atomic_store(p.abort_, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    __builtin_amdgcn_s_sleep(SLEEP);
  }
}
__global__ void __launch_bounds__(PT) k_persist(P p) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[PT * PB];
  __shared__ uint32_t s_x;
  const uint32_t t = threadIdx.x, NT = p.nt;
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(p.dst, (p.n + 1) * US);
  if (t == 0) s_x = (p.flags & 1u) ? __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : blockIdx.x;
  __syncthreads();
  const uint32_t r = s_x;
  __syncthreads();
  v4u o[PB];
  if (r < NT) {                                                          // ---- a chain workgroup: slot r of every step
    for (uint32_t u = 1; u <= p.n; u++) {
      load_og<PB>(p.ogt + (size_t)(u - 1) * GT, r * PT * PB, GT, o);      // (does not depend on step u-1: requested before the wait)
      if (u > 1) {
        if (t < 64) { const bool ok = wait_step<1>(p, u - 1, NT); if (t == 0) s_x = ok; }
        __syncthreads();
        if (!s_x) return;
      }
      item<PB, true, true>(rs, u * US + HEAD, r * PT * PB, GT, o, lds);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // every storing wave drains
      __syncthreads();
      if (t == 0) __hip_atomic_fetch_add(p.cnt + (u * SH + r % SH) * SHS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {                                                               // ---- a head workgroup: item r - NT of every unit's head, behind the chain
    const uint32_t h = r - NT, NH = gridDim.x - NT;
    for (uint32_t u = 1; u <= p.n; u++) {
      bool any = false;
      for (uint32_t j = h; j * PT * PB < GH; j += NH) {
        load_og<PB>(p.ogh + (size_t)(u - 1) * GH, j * PT * PB, GH, o);
        if (!any && u > 1) {
          if (t < 64) { const bool ok = wait_step<8>(p, u - 1, NT); if (t == 0) s_x = ok; }
          __syncthreads();
          if (!s_x) return;
          __syncthreads();
        }
        any = true;
        item<PB, true, true>(rs, u * US, j * PT * PB, GH, o, lds);
      }
    }
  }
}
__global__ void __launch_bounds__(PT) k_heads_all(P p) {               // P0: every head in one launch, after the chain (blockIdx.y: unit 1 + y)
  __shared__ __attribute__((aligned(16))) uint32_t lds[PT * PB];
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(p.dst, (p.n + 1) * US);
  const uint32_t u = 1 + blockIdx.y;
  v4u o[PB];
  load_og<PB>(p.ogh + (size_t)(u - 1) * GH, blockIdx.x * PT * PB, GH, o);
  item<PB, false, true>(rs, u * US, blockIdx.x * PT * PB, GH, o, lds);
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 255;
  P p; memset(&p, 0, sizeof p);
  p.n = n;
  const size_t dst_bytes = (size_t)(n + 1) * US + 4096, cnt_bytes = (size_t)(n + 2) * SH * SHS * 4;
  v4u *ogt, *ogh;
  CK(hipMalloc(&p.dst, dst_bytes)); CK(hipMalloc(&ogt, (size_t)n * GT * 16)); CK(hipMalloc(&ogh, (size_t)n * GH * 16)); CK(hipMalloc(&p.cnt, cnt_bytes)); CK(hipMalloc(&p.abort_, 256));
  p.ogt = ogt; p.ogh = ogh; p.ticket = p.abort_ + 32;
  hipLaunchKernelGGL(k_init_og, dim3((uint32_t)(((uint64_t)n * GT + 255) / 256)), dim3(256), 0, 0, ogt, ogh, n);
  CK(hipDeviceSynchronize());
  std::vector<uint8_t> ref((size_t)(n + 1) * US), got((size_t)(n + 1) * US);
  for (uint32_t u = 0; u <= n; u++)
    for (uint32_t kind = 0; kind < 2; kind++)
      for (uint32_t g = 0; g < (kind ? GT : GH); g++) {
        uint32_t e[4]; uint8_t l[4];
        group_of(kind, u, g, e, l);
        const size_t at = (size_t)u * US + (kind ? HEAD : 0) + 4 * g;
        for (int i = 0; i < 4; i++) ref[at + i] = e[i] ? ref[at + i - e[i]] : l[i];
      }
  // (heads copy from the previous TAIL, tails too: the model above fills a unit's head before its tail, both from finished bytes)
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ev[300];
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto run = [&](const char* name, int mode, uint32_t gs, uint32_t nh) {
    float best = 1e9f; bool ok = true; uint32_t ab = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k_init_dst, dim3((uint32_t)(((uint64_t)(n + 1) * (US / 4) + 255) / 256)), dim3(256), 0, s1, p.dst, n);
      CK(hipMemsetAsync(p.cnt, 0, cnt_bytes, s1)); CK(hipMemsetAsync(p.abort_, 0, 256, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s1));
      if (mode == 0) {                                       // L
        uint32_t ne = 0, done_heads = 0;                     // heads of units <= done_heads are enqueued
        auto heads_upto = [&](uint32_t upto) {               // needs the tails of units < upto: everything enqueued on s1 so far
          if (upto <= done_heads) return;
          CK(hipEventRecord(ev[ne], s1)); CK(hipStreamWaitEvent(s2, ev[ne], 0)); ne++;
          hipLaunchKernelGGL(k_heads, dim3((GH + 511) / 512, upto - done_heads), dim3(256), 52 * 1024, s2, p, done_heads + 1);
          done_heads = upto;
        };
        for (uint32_t u = 1; u <= n; u++) {
          hipLaunchKernelGGL(k_tail, dim3((GT + 511) / 512), dim3(256), 0, s1, p, u);
          if (u % gs == 0) heads_upto(u + 1 < n ? u + 1 : n);
        }
        heads_upto(n);
        CK(hipEventRecord(ev[ne], s2)); CK(hipStreamWaitEvent(s1, ev[ne], 0));
      } else if (mode == 1) {                                // P0
        p.flags = 0; p.nt = 256;
        hipLaunchKernelGGL(k_persist, dim3(256), dim3(PT), 0, s1, p);
        hipLaunchKernelGGL(k_heads_all, dim3((GH + PT * PB - 1) / (PT * PB), n), dim3(PT), 0, s1, p);
      } else {                                               // P1
        p.flags = 1; p.nt = 256;
        hipLaunchKernelGGL(k_persist, dim3(256 + nh), dim3(PT), 0, s1, p);
      }
      CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      CK(hipMemcpy(&ab, p.abort_, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(got.data(), p.dst, got.size(), hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < got.size(); i++) bad += got[i] != ref[i];
      if (bad) { ok = false; printf("   rep %d: %zu wrong bytes\n", rep, bad); }
    }
    printf("%-66s %8.3f ms  %6.2f us/unit  %s%s\n", name, best, best * 1000.f / n, ok ? "OK" : "BAD", ab ? "  GAVE UP" : "");
    fflush(stdout);
  };
  run("L   tails: launch per step; heads: launch per 16 units, 2nd stream", 0, 16, 0);
  run("L   heads per 8 units", 0, 8, 0);
  run("L   heads per 32 units", 0, 32, 0);
  run("P0  tails: persistent chain; heads: one launch behind it", 1, 0, 0);
  run("P1  one persistent launch: 256 chain + 232 head workgroups", 2, 0, 232);
  run("P1  256 chain + 116 head workgroups (2 items per unit)", 2, 0, 116);
  return 0;
}
