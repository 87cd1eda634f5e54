// chain_bench.hip — prices a persistent, fence-free sweep chain against the chain of launches zg_k_sweep uses.
//
// The sweep's dependency, reduced to its shape: NSTEPS units, each with a TAIL of 2 MiB whose match bytes are gathered from the TAIL
// of the unit in front of it (finished output), 16 bytes of scratch ("effective offsets") per group of four output bytes, ~2 distinct
// sources per group, sources uniformly spread over the previous tail (the conservative case). Every variant's output is compared
// with a host model byte by byte — a stale gather (a hand-off without the visibility it needs) shows as a wrong byte, not as a time.
//
//   L   one launch per step (1024 workgroups x 256 threads, plain loads, dword stores): what the engine does today
//   P   ONE launch, NW resident workgroups, workgroup r owns slot r of every step: scratch words of step s requested BEFORE the wait for
//       step s-1; hand-off without fences — 16-byte sc1 (write-through) stores, every storing wave drains vmcnt, ONE lane arrives on a
//       sharded agent-scope counter; consumers poll relaxed (sc1 loads) and gather with sc1 loads (guide: Guideline 16, R1)
//   flags: 1 two-level arrival (last arriver of a shard bumps ONE top word; pollers read one line)   2 empty steps (sync only)
//          4 dword sc1 stores instead of 16-byte ones   8 scratch words requested AFTER the wait   16 plain stores + release fence /
//          acquire fence + plain loads (the fenced form)   32 launch chain with empty steps
// Every spin is bounded (a give-up flag ends the whole grid): a mistake must show as FAIL, never as a hang.
// build: hipcc --offload-arch=gfx950 -O3 -o chain_bench tools/dev/chain_bench.hip ; run: ./chain_bench [nsteps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr uint32_t TAIL = 2u << 20, G = TAIL / 4, USTRIDE = 3900000u & ~3u, SH = 16, SHS = 32 /* u32 between shards: 128 B */;
constexpr uint32_t OOB = 0xFFFFFFFFu;
struct P { uint8_t* dst; const v4u* og; uint32_t* cnt; uint32_t* top; uint32_t* abort_; uint32_t nsteps, flags; long long* stamps; };

__host__ __device__ inline uint64_t mix(uint64_t i) { uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
// group g of step s (1-based; step 0 is all literal): four effective offsets (0: literal byte) and the literal values
__host__ __device__ inline void group_of(uint32_t s, uint32_t g, uint32_t e[4], uint8_t lit[4]) {
  const uint64_t r = mix((uint64_t)s * G + g);
  uint32_t a = (uint32_t)(r & 0xFFFFFF) % (TAIL - 8);
  const uint32_t b = (uint32_t)((r >> 24) & 0xFFFFFF) % (TAIL - 16), c = (uint32_t)(r >> 48) & 7u;
#ifdef CONT   // runs that go on into the next group (real scratch: a run of equal offsets is 2.8 bytes long): byte 0 continues the previous group's byte 3
  if (g > 0 && (r >> 61) < 5) { const uint64_t rp = mix((uint64_t)s * G + g - 1); if (((uint32_t)(rp >> 48) & 7u) <= 3u) a = (uint32_t)((rp >> 24) & 0xFFFFFF) % (TAIL - 16) + 4; }
#endif
  for (uint32_t i = 0; i < 4; i++) {
    const bool isl = s == 0 || ((r >> (52 + 3 * i)) & 7u) == 0;
    const uint32_t src = (i < c ? a : b) + i;
    e[i] = isl ? 0u : USTRIDE + 4 * g + i - src;
    lit[i] = (uint8_t)(mix(((uint64_t)s << 40) | ((uint64_t)g << 2) | i) >> 13);
  }
}
__global__ void k_init_og(v4u* og, uint32_t nsteps) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)nsteps * G) return;
  uint32_t e[4]; uint8_t l[4];
  group_of((uint32_t)(i / G) + 1, (uint32_t)(i % G), e, l);
  og[i] = (v4u){e[0], e[1], e[2], e[3]};
}
__global__ void k_init_dst(uint8_t* dst, uint32_t nsteps) {   // literal bytes in place (the flatten wrote them), everything else poisoned
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)(nsteps + 1) * G) return;
  const uint32_t s = (uint32_t)(i / G), g = (uint32_t)(i % G);
  uint32_t e[4]; uint8_t l[4];
  group_of(s, g, e, l);
  uint32_t v = 0;
  for (int k = 0; k < 4; k++) v |= (uint32_t)(e[k] ? 0xEEu : l[k]) << (8 * k);
  *(uint32_t*)(dst + (uint64_t)s * USTRIDE + 4 * g) = v;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <bool SC1> __device__ __forceinline__ v2u ld64(__amdgpu_buffer_rsrc_t rs, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, SC1 ? 16 : 0); }

// one item: groups [g0, g0 + T*B) of step s; wave w owns 64*B consecutive groups, lane l group k*64 + l of them (coalesced scratch reads)
template <int B, int T> __device__ __forceinline__ void load_og(const P& p, uint32_t s, uint32_t g0, v4u (&o)[B]) {
  const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
#pragma unroll
  for (int k = 0; k < B; k++) { const uint32_t g = g0 + w * 64 * B + k * 64 + l; o[k] = p.og[(uint64_t)(s - 1) * G + (g < G ? g : 0)]; }
}
template <int B, int T, bool SC1, bool SC1ST, bool ST16, int MASK = 0> __device__ __forceinline__ void item(const P& p, __amdgpu_buffer_rsrc_t rs, uint32_t s, uint32_t g0, const v4u (&o)[B], uint32_t* lds) {
  const uint32_t t = threadIdx.x, w = t >> 6, l = t & 63;
  v2u rA[B], rB[B], rC[B], rD[B], rW[B];
#pragma unroll
  for (int k = 0; k < B; k++) {
    const v4u q = o[k];
    const uint32_t g = g0 + w * 64 * B + k * 64 + l, wrel = s * USTRIDE + 4 * g;
    const bool ux = q.x != 0, uy = q.y != 0, uz = q.z != 0, uw = q.w != 0, all = ux && uy && uz && uw;
    const bool nD = uw && !(ux && q.w == q.x), nB = uy && !(ux && q.y == q.x) && !(uw && q.y == q.w);
    const bool nC = uz && !(ux && q.z == q.x) && !(uw && q.z == q.w) && !(uy && q.z == q.y);
    bool skipA = false; uint32_t sA = 0;
    if (MASK == 4) {     // a run that continues from the lane below: its first bytes are in the upper half of that lane's window for ITS last byte
      const uint32_t pw = __shfl_up(q.w, 1, 64), px = __shfl_up(q.x, 1, 64);
      const uint32_t j = q.w == q.x ? 4u : q.z == q.x ? 3u : q.y == q.x ? 2u : 1u;
      sA = (0u - q.x) & 3u;
      skipA = l != 0 && ux && pw == q.x && pw != px && sA + j <= 4u;
    }
    if (MASK == 1) {                                          // exec-masked: a lane that needs nothing issues nothing
      rA[k] = rD[k] = rB[k] = rC[k] = rW[k] = (v2u){0u, 0u};
      if (ux) rA[k] = ld64<SC1>(rs, (wrel - q.x) & ~3u);
      if (nD) rD[k] = ld64<SC1>(rs, (wrel - q.w) & ~3u);
      if (nB) rB[k] = ld64<SC1>(rs, (wrel - q.y) & ~3u);
      if (nC) rC[k] = ld64<SC1>(rs, (wrel - q.z) & ~3u);
      if (!all && g < G) rW[k] = ld64<SC1>(rs, wrel);
    } else {
    rA[k] = ld64<SC1>(rs, (MASK != 3 && ux && !skipA) ? (wrel - q.x) & ~3u : OOB);
    rD[k] = ld64<SC1>(rs, (MASK != 3 && nD) ? (wrel - q.w) & ~3u : OOB);
    if (MASK != 2) rB[k] = ld64<SC1>(rs, (MASK != 3 && nB) ? (wrel - q.y) & ~3u : OOB); else rB[k] = rA[k];
    if (MASK != 2) rC[k] = ld64<SC1>(rs, (MASK != 3 && nC) ? (wrel - q.z) & ~3u : OOB); else rC[k] = rD[k];
    rW[k] = ld64<SC1>(rs, (MASK != 3 && !all && g < G) ? wrel : OOB);      // the bytes in place (16-byte stores rewrite literal bytes too)
    }
  }
#pragma unroll
  for (int k = 0; k < B; k++) {
    const v4u q = o[k];
    const uint32_t g = g0 + w * 64 * B + k * 64 + l;
    const bool ux = q.x != 0, uy = q.y != 0, uz = q.z != 0, uw = q.w != 0;
    auto fun = [&](const v2u r, uint32_t e) { return __builtin_amdgcn_alignbit(r.y, r.x, ((0u - e) & 3u) * 8u); };
    uint32_t lA = fun(rA[k], q.x);
    if (MASK == 4) {
      const uint32_t pw = __shfl_up(q.w, 1, 64), px = __shfl_up(q.x, 1, 64), phi = __shfl_up(rD[k].y, 1, 64);
      const uint32_t j = q.w == q.x ? 4u : q.z == q.x ? 3u : q.y == q.x ? 2u : 1u, sA = (0u - q.x) & 3u;
      if (l != 0 && ux && pw == q.x && pw != px && sA + j <= 4u) lA = phi >> (8u * sA);
    }
    const uint32_t lD = fun(rD[k], q.w), lB = fun(rB[k], q.y), lC = fun(rC[k], q.z), lW = rW[k].x;
    const uint32_t sw = (ux && q.w == q.x) ? lA : lD;
    const uint32_t sy = (ux && q.y == q.x) ? lA : (uw && q.y == q.w) ? sw : lB;
    const uint32_t sz = (ux && q.z == q.x) ? lA : (uw && q.z == q.w) ? sw : (uy && q.z == q.y) ? sy : lC;
    const uint32_t v = ((ux ? lA : lW) & 0xFFu) | ((uy ? sy : lW) & 0xFF00u) | ((uz ? sz : lW) & 0xFF0000u) | ((uw ? sw : lW) & 0xFF000000u);
    if (ST16) lds[w * 64 * B + k * 64 + l] = v;
    else __builtin_amdgcn_raw_buffer_store_b32(v, rs, g < G ? s * USTRIDE + 4 * g : OOB, 0, SC1ST ? 16 : 0);
  }
  if (ST16) {   // the wave's 64*B dwords leave as 16-byte pieces: lane l < 16*B stores groups 4l .. 4l+3 of the wave's run
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (l < 16 * B) {
      const v4u v = *(const v4u*)(lds + w * 64 * B + 4 * l);
      const uint32_t g = g0 + w * 64 * B + 4 * l;
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, g < G ? s * USTRIDE + 4 * g : OOB, 0, SC1ST ? 16 : 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
}

template <int B, int MASK = 0> __global__ void __launch_bounds__(256) k_step(P p, uint32_t s) {      // L: one launch per step
  constexpr int T = 256;
  if (p.flags & 32u) return;
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(p.dst, (p.nsteps + 1) * USTRIDE);
  v4u o[B];
  load_og<B, T>(p, s, blockIdx.x * T * B, o);
  item<B, T, false, false, false, MASK>(p, rs, s, blockIdx.x * T * B, o, nullptr);
}

template <int B, int T> __global__ void __launch_bounds__(T) k_persist(P p) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[T * B];
  __shared__ uint32_t s_dead;
  const uint32_t r = blockIdx.x, NW = gridDim.x, t = threadIdx.x, IPS = (G + T * B - 1) / (T * B);
  const uint32_t m = NW < IPS ? NW : IPS;                       // workgroups that have an item in every step
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(p.dst, (p.nsteps + 1) * USTRIDE);
  const bool two = p.flags & 1u, empty = p.flags & 2u, st4 = p.flags & 4u, late = p.flags & 8u, fenced = p.flags & 16u;
  const bool pl_ld = p.flags & 64u, pl_st = p.flags & 128u, nodrain = p.flags & 256u, acq = p.flags & 512u;
  bool dead = false;
  for (uint32_t s = 1; s <= p.nsteps && !dead; s++) {
    v4u o[B];
    const bool stamp = (p.flags & 1024u) && t == 0;
    long long* sp = p.stamps + ((size_t)r * (p.nsteps + 1) + s) * 5;
    if (stamp) sp[0] = wall_clock64();
    if (!empty && !late && r < IPS) load_og<B, T>(p, s, r * T * B, o);
    if (s > 1) {                                                // every slot of step s-1 stored and drained?
      if (t < 64) {
        const uint32_t want = two ? (t == 0 ? (m < SH ? m : SH) : 0u) : (t < SH && m > t ? (m - t + SH - 1) / SH : 0u);
        const uint32_t* q = t == 63 ? p.abort_ : two ? p.top + (s - 1) * SHS : p.cnt + ((s - 1) * SH + (t < SH ? t : 0)) * SHS;
        for (uint32_t spins = 0;; spins++) {
          const uint32_t v = (two && t != 0 && t != 63) ? 0u : __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ab = __any(t == 63 && v != 0);
          if (ab) { dead = true; break; }
          if (__all(t == 63 || v >= want)) break;
          if (spins > (1u << 15)) { if (t == 0) __hip_atomic_store(p.abort_, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (fenced || acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      if (t == 0) s_dead = dead;
      __syncthreads();
      dead = s_dead;
      if (dead) break;
    }
    if (stamp) sp[1] = wall_clock64();
    if (!empty) {
      for (uint32_t j = r; j < IPS; j += NW) {
        if (late || j != r) load_og<B, T>(p, s, j * T * B, o);
        if (fenced) item<B, T, false, false, false>(p, rs, s, j * T * B, o, lds);
        else if (st4) item<B, T, true, true, false>(p, rs, s, j * T * B, o, lds);
        else if (pl_ld && pl_st) item<B, T, false, false, true>(p, rs, s, j * T * B, o, lds);
        else if (pl_ld) item<B, T, false, true, true>(p, rs, s, j * T * B, o, lds);
        else if (pl_st) item<B, T, true, false, true>(p, rs, s, j * T * B, o, lds);
        else if ((p.flags >> 11) == 1u) item<B, T, true, true, true, 1>(p, rs, s, j * T * B, o, lds);
        else if ((p.flags >> 11) == 2u) item<B, T, true, true, true, 2>(p, rs, s, j * T * B, o, lds);
        else if ((p.flags >> 11) == 3u) item<B, T, true, true, true, 3>(p, rs, s, j * T * B, o, lds);
        else item<B, T, true, true, true>(p, rs, s, j * T * B, o, lds);
      }
    }
    if (stamp) sp[2] = wall_clock64();
    if (r < m) {                                                 // arrive: every storing wave drains, ONE lane publishes
      if (!nodrain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (stamp) sp[3] = wall_clock64();
      if (t == 0) {
        if (fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const uint32_t h = r % SH, exp_h = (m - h + SH - 1) / SH;
        const uint32_t old = __hip_atomic_fetch_add(p.cnt + (s * SH + h) * SHS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (two && old + 1 == exp_h) __hip_atomic_fetch_add(p.top + s * SHS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamp) sp[4] = wall_clock64();
      }
    }
  }
}

int main(int argc, char** argv) {
  const uint32_t nsteps = argc > 1 ? (uint32_t)atoi(argv[1]) : 255;
  P p; memset(&p, 0, sizeof p);
  p.nsteps = nsteps;
  const size_t dst_bytes = (size_t)(nsteps + 1) * USTRIDE + 4096, og_bytes = (size_t)nsteps * G * 16, cnt_bytes = (size_t)(nsteps + 2) * SH * SHS * 4;
  v4u* og; CK(hipMalloc(&p.dst, dst_bytes)); CK(hipMalloc(&og, og_bytes)); CK(hipMalloc(&p.cnt, cnt_bytes)); CK(hipMalloc(&p.top, cnt_bytes)); CK(hipMalloc(&p.abort_, 256));
  p.og = og;
  const size_t stamp_n = (size_t)1024 * (nsteps + 1) * 5;
  CK(hipMalloc(&p.stamps, stamp_n * 8));
  std::vector<long long> stamps(stamp_n);
  hipLaunchKernelGGL(k_init_og, dim3((uint32_t)(((uint64_t)nsteps * G + 255) / 256)), dim3(256), 0, 0, og, nsteps);
  CK(hipDeviceSynchronize());
  // host model
  std::vector<uint8_t> ref((size_t)(nsteps + 1) * TAIL), got((size_t)(nsteps + 1) * TAIL);
  for (uint32_t s = 0; s <= nsteps; s++)
    for (uint32_t g = 0; g < G; g++) {
      uint32_t e[4]; uint8_t l[4];
      group_of(s, g, e, l);
      for (int i = 0; i < 4; i++) ref[(size_t)s * TAIL + 4 * g + i] = e[i] ? ref[(size_t)(s - 1) * TAIL + (USTRIDE + 4 * g + i - e[i])] : l[i];
    }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, uint32_t flags, int B, uint32_t nw, int T = 256) {
    p.flags = flags;
    float best = 1e9f; bool ok = true; uint32_t ab = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k_init_dst, dim3((uint32_t)(((uint64_t)(nsteps + 1) * G + 255) / 256)), dim3(256), 0, st, p.dst, nsteps);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      if (nw == 0) {
        for (uint32_t s = 1; s <= nsteps; s++) {
          if (B == 2 && (flags >> 11) == 4u) hipLaunchKernelGGL((k_step<2, 4>), dim3((G + 256 * 2 - 1) / (256 * 2)), dim3(256), 0, st, p, s);
          else if (B == 2) hipLaunchKernelGGL(k_step<2>, dim3((G + 256 * 2 - 1) / (256 * 2)), dim3(256), 0, st, p, s);
          else hipLaunchKernelGGL(k_step<4>, dim3((G + 256 * 4 - 1) / (256 * 4)), dim3(256), 0, st, p, s);
        }
      } else {
        CK(hipMemsetAsync(p.cnt, 0, cnt_bytes, st)); CK(hipMemsetAsync(p.top, 0, cnt_bytes, st)); CK(hipMemsetAsync(p.abort_, 0, 256, st));
        if (T == 1024) { if (B == 2) hipLaunchKernelGGL((k_persist<2, 1024>), dim3(nw), dim3(1024), 0, st, p); else hipLaunchKernelGGL((k_persist<1, 1024>), dim3(nw), dim3(1024), 0, st, p); }
        else if (T == 512) { if (B == 2) hipLaunchKernelGGL((k_persist<2, 512>), dim3(nw), dim3(512), 0, st, p); else hipLaunchKernelGGL((k_persist<4, 512>), dim3(nw), dim3(512), 0, st, p); }
        else if (B == 1) hipLaunchKernelGGL((k_persist<1, 256>), dim3(nw), dim3(256), 0, st, p);
        else if (B == 2) hipLaunchKernelGGL((k_persist<2, 256>), dim3(nw), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((k_persist<4, 256>), dim3(nw), dim3(256), 0, st, p);
      }
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
      CK(hipMemcpy(&ab, p.abort_, 4, hipMemcpyDeviceToHost));
      if (!(flags & 34u) && ((flags >> 11) < 2u || (flags >> 11) == 4u)) {
        for (uint32_t s = 0; s <= nsteps; s++) CK(hipMemcpy(got.data() + (size_t)s * TAIL, p.dst + (size_t)s * USTRIDE, TAIL, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < got.size(); i++) bad += got[i] != ref[i];
        if (bad) { ok = false; printf("   rep %d: %zu wrong bytes\n", rep, bad); }
      }
    }
    if ((flags & 1024u) && nw) {
      // where a step's time goes (last repetition): per workgroup and step, stamps at 0 step start, 1 wait over, 2 gathers used / stores issued,
      // 3 stores drained + workgroup barrier, 4 arrival returned. Units of 10 ns (100 MHz wall clock).
      CK(hipMemcpy(stamps.data(), p.stamps, stamp_n * 8, hipMemcpyDeviceToHost));
      double w01 = 0, w12 = 0, w23 = 0, w34 = 0, crit = 0, skew = 0, maxdata = 0; size_t n = 0;
      for (uint32_t s = 2; s <= nsteps; s++) {
        long long last_arr = 0, first_go = 1ll << 62, last_go = 0, prev_last_arr = 0; double md = 0;
        for (uint32_t r = 0; r < nw; r++) {
          const long long* q = &stamps[((size_t)r * (nsteps + 1) + s) * 5];
          const long long* qp = &stamps[((size_t)r * (nsteps + 1) + s - 1) * 5];
          w01 += q[1] - q[0]; w12 += q[2] - q[1]; w23 += q[3] - q[2]; w34 += q[4] - q[3]; n++;
          last_arr = q[4] > last_arr ? q[4] : last_arr; first_go = q[1] < first_go ? q[1] : first_go; last_go = q[1] > last_go ? q[1] : last_go;
          prev_last_arr = qp[4] > prev_last_arr ? qp[4] : prev_last_arr;
          md = (double)(q[3] - q[1]) > md ? (double)(q[3] - q[1]) : md;
        }
        crit += (double)(first_go - prev_last_arr); skew += (double)(last_go - first_go); maxdata += md;
      }
      const double k = 0.01, ns = nsteps - 1;
      printf("   per wg+step (us): wait %.2f  gathers+compute %.2f  drain+barrier %.2f  arrive %.2f | per step: last arrival -> first go %.2f, go skew %.2f, slowest wg's data phase %.2f\n",
             w01 / n * k, w12 / n * k, w23 / n * k, w34 / n * k, crit / ns * k, skew / ns * k, maxdata / ns * k);
    }
    printf("%-58s %8.3f ms  %6.2f us/step  %s%s\n", name, best, best * 1000.f / nsteps, (flags & 34u) ? "(sync only)" : ok ? "OK" : "BAD", ab ? "  GAVE UP" : "");
    fflush(stdout);
  };
#ifdef CONT
  printf("scratch with runs that continue into the next group (64 %% of the groups whose neighbour ends on a second run)\n");
#endif
  run("L  launch per step, B=2 (1024 wg)", 0, 2, 0);
  run("L  launch per step, B=2, first load served by the lane below", 4u << 11, 2, 0);
  return 0;
}
