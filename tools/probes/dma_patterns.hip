// dev probe: what does the one-tap kernels' data movement reach by itself, by ACCESS PATTERN?
//   loads : LDS-DMA (buffer_load_dwordx4 ... lds), a tile = R rows, each K step takes a segment of S bytes of every row
//           (row pitch P bytes), NL loader waves per workgroup, INF instructions (1 KB each) in flight per wave
//   stores: NS store waves writing R x OB-byte row tiles from registers (SO bytes of a row per lane group)
//   one persistent workgroup per CU, tiles b, b + G, ...; no compute, no barrier: pure traffic.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_patterns dma_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  return i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma16s(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void wait_vm(int n) {   // n uniform
  switch (n) {
#define W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20)
    W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40)
    W(41) W(42) W(43) W(44) W(45) W(46) W(47) W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60)
    W(61) W(62)
#undef W
    default: break;
  }
}

struct Cfg {
  // loads
  int R, S, P;        // rows per tile, segment bytes per row and step, row pitch in bytes
  int NL, INF;        // loader waves, instructions in flight per loader wave
  int ntiles;         // row tiles in the tensor
  int order;          // 0: tiles b, b+G, ...   1: contiguous span of tiles per workgroup
  int pf;             // sparse L2 prefetch: lanes 128 B apart touch the tile `pf` tiles ahead (0 = off); done by wave NL+NS
  // stores
  int NS, OB, SO;     // store waves, output row bytes, bytes of a row one lane group writes per instruction
  int nload;          // 0: no loads (stores only)
};

// waves: [0, NL) loaders, [NL, NL + NS) store waves, optionally one prefetch wave
__global__ __launch_bounds__(1024) void k_traffic(const unsigned char* __restrict__ src, unsigned long long src_bytes,
                                                  unsigned char* __restrict__ dst, Cfg c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int G = gridDim.x;
  int t_lo, t_hi, t_step;
  if (c.order == 0) { t_lo = blockIdx.x; t_hi = c.ntiles; t_step = G; }
  else { const int per = (c.ntiles + G - 1) / G; t_lo = per * blockIdx.x; t_hi = min(c.ntiles, t_lo + per); t_step = 1; }
  if (wave < c.NL) {
    if (!c.nload) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const i32x4 rs = make_rsrc(src, (uint32_t)(src_bytes > 0xffffffffull ? 0xffffffffull : src_bytes));
    const int lpr = c.S / 16, rpi = 64 / lpr;          // lanes per row segment, rows per instruction
    const int ipt = c.R / rpi;                          // instructions per (tile, K step)
    const int ipw = ipt / c.NL;                         // ... per loader wave
    const int spt = c.P / c.S;                          // K steps per tile
    const uint32_t lane_off = (uint32_t)(lane / lpr) * (uint32_t)c.P + (uint32_t)(lane % lpr) * 16u;
    const int slots = 2 * c.INF;                        // LDS ring of 1 KB slots per wave (never read)
    int slot = 0, pending = 0;
    for (int t = t_lo; t < t_hi; t += t_step) {
      const uint32_t tbase = (uint32_t)t * (uint32_t)c.R * (uint32_t)c.P;
      for (int ks = 0; ks < spt; ++ks) {
        for (int i = 0; i < ipw; ++i) {
          if (pending >= c.INF) { wait_vm(c.INF - 1); pending = c.INF - 1; }
          const int instr = wave * ipw + i;
          const uint32_t soff = tbase + (uint32_t)(instr * rpi) * (uint32_t)c.P + (uint32_t)ks * (uint32_t)c.S;
          dma16s(rs, lds0 + (uint32_t)(wave * slots + slot) * 1024u, lane_off, soff);
          slot = slot + 1 == slots ? 0 : slot + 1;
          ++pending;
        }
      }
    }
    wait_vm(0);
    return;
  }
  if (wave < c.NL + c.NS) {
    const int sw = wave - c.NL;
    const int lpr = c.SO / 16, rpi = 64 / lpr;          // lanes per row piece, rows per instruction
    const int pieces = c.OB / c.SO;                     // row pieces per row
    const int ipt = (c.R / rpi) * pieces;               // instructions per tile
    const int ipw = ipt / c.NS;
    const f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int t = t_lo; t < t_hi; t += t_step) {
      unsigned char* tb = dst + (size_t)t * c.R * c.OB;
      for (int i = 0; i < ipw; ++i) {
        const int instr = sw * ipw + i;
        const int rg = instr / pieces, pc = instr % pieces;
        unsigned char* p = tb + (size_t)(rg * rpi + lane / lpr) * c.OB + (size_t)pc * c.SO + (size_t)(lane % lpr) * 16;
        *reinterpret_cast<f32x4*>(p) = v;
      }
    }
    return;
  }
  if (c.pf > 0 && wave == c.NL + c.NS) {
    // one dword per 128-byte line, 8 KB per instruction, `pf` tiles ahead of the loaders; at most 8 in flight
    const size_t tile_bytes = (size_t)c.R * c.P;
    const int ipt = (int)(tile_bytes / 8192);
    float acc = 0.f;
    for (int t = t_lo; t < t_hi; t += t_step) {
      const int tp = t + c.pf * t_step;
      if (tp >= t_hi) break;
      const unsigned char* tb = src + (size_t)tp * tile_bytes;
      for (int i = 0; i < ipt; ++i) acc += *reinterpret_cast<const float*>(tb + (size_t)i * 8192 + (size_t)lane * 128);
      // pace: do not run arbitrarily far ahead of the loaders (no sync in this probe: a sleep per tile)
      __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 12345.678f) dst[0] = 1;
    return;
  }
}

static double run(const char* name, const Cfg& c, const unsigned char* src, size_t src_bytes, unsigned char* dst, int grid,
                  double bytes_moved) {
  const int waves = c.NL + c.NS + (c.pf > 0 ? 1 : 0);
  const size_t lds = (size_t)c.NL * 2 * c.INF * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_traffic), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_traffic, dim3(grid), dim3(64 * waves), lds, 0, src, (unsigned long long)src_bytes, dst, c);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(s);
    const int R = 5;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_traffic, dim3(grid), dim3(64 * waves), lds, 0, src, (unsigned long long)src_bytes, dst, c);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    if (ms / R < best) best = ms / R;
  }
  if (hipGetLastError() != hipSuccess) printf("ERROR\n");
  const double tbs = bytes_moved / (best * 1e-3) / 1e12;
  printf("%-72s %8.1f us  %6.2f TB/s\n", name, best * 1e3, tbs);
  fflush(stdout);
  return tbs;
}

int main() {
  const size_t bytes = 256ull << 20;   // 268 MB in, up to 268 MB out
  unsigned char *x, *y;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes);
  hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
  char nm[160];
  const int grid = 256;
  // ---- A: loads only.  pitch x segment x in-flight
  printf("== loads only (268 MB), R=128, NL loader waves x INF KB in flight each, tiles b, b+G\n");
  for (int P : {256, 1024, 4096}) {
    for (int S : {128, 256, 512, 1024}) {
      if (S > P) continue;
      for (int NL : {2, 4, 8}) {
        for (int INF : {4, 8, 16}) {
          Cfg c{};
          c.R = 128; c.S = S; c.P = P; c.NL = NL; c.INF = INF; c.ntiles = (int)(bytes / ((size_t)c.R * P)); c.order = 0; c.nload = 1;
          if ((c.R / (64 / (S / 16))) % NL) continue;
          if ((size_t)NL * 2 * INF * 1024 > 160 * 1024) continue;
          snprintf(nm, sizeof nm, "load P=%4d S=%4d NL=%d INF=%2d (%3d KB in flight/CU)", P, S, NL, INF, NL * INF);
          run(nm, c, x, bytes, y, grid, (double)bytes);
        }
      }
    }
  }
  printf("== loads only, contiguous span per workgroup (order 1)\n");
  for (int P : {1024, 4096}) for (int S : {128, 1024}) {
    Cfg c{}; c.R = 128; c.S = S; c.P = P; c.NL = 4; c.INF = 16; c.ntiles = (int)(bytes / ((size_t)c.R * P)); c.order = 1; c.nload = 1;
    snprintf(nm, sizeof nm, "load span P=%4d S=%4d NL=4 INF=16", P, S);
    run(nm, c, x, bytes, y, grid, (double)bytes);
  }
  printf("== loads only, two workgroups per CU (grid 512)\n");
  for (int P : {1024, 4096}) for (int S : {128, 512}) {
    Cfg c{}; c.R = 128; c.S = S; c.P = P; c.NL = 4; c.INF = 8; c.ntiles = (int)(bytes / ((size_t)c.R * P)); c.order = 0; c.nload = 1;
    snprintf(nm, sizeof nm, "load grid512 P=%4d S=%4d NL=4 INF=8", P, S);
    run(nm, c, x, bytes, y, 512, (double)bytes);
  }
  // ---- B: stores only
  printf("== stores only (268 MB): R=128 rows x OB bytes, NS store waves, SO bytes per lane group\n");
  for (int OB : {256, 1024}) for (int SO : {64, 256, 1024}) for (int NS : {2, 4, 8}) {
    if (SO > OB) continue;
    Cfg c{}; c.R = 128; c.NS = NS; c.OB = OB; c.SO = SO; c.ntiles = (int)(bytes / ((size_t)c.R * OB)); c.order = 0; c.nload = 0; c.NL = 0;
    const int ipt = (c.R / (64 / (SO / 16))) * (OB / SO);
    if (ipt % NS) continue;
    snprintf(nm, sizeof nm, "store OB=%4d SO=%4d NS=%d", OB, SO, NS);
    run(nm, c, x, bytes, y, grid, (double)bytes);
  }
  // ---- C: loads + stores together (the 256 -> 256 layer: 268 MB in, 268 MB out)
  printf("== loads + stores (268 + 268 MB)\n");
  for (int S : {128, 256, 1024}) for (int NL : {2, 4}) for (int INF : {8, 16}) for (int NS : {2, 4}) for (int SO : {256, 1024}) {
    Cfg c{}; c.R = 128; c.S = S; c.P = 1024; c.NL = NL; c.INF = INF; c.ntiles = (int)(bytes / ((size_t)c.R * 1024)); c.order = 0; c.nload = 1;
    c.NS = NS; c.OB = 1024; c.SO = SO;
    if ((c.R / (64 / (S / 16))) % NL) continue;
    const int ipt = (c.R / (64 / (SO / 16))) * (c.OB / SO);
    if (ipt % NS) continue;
    snprintf(nm, sizeof nm, "ld+st S=%4d NL=%d INF=%2d | NS=%d SO=%4d", S, NL, INF, NS, SO);
    run(nm, c, x, bytes, y, grid, 2.0 * bytes);
  }
  // ---- D: sparse L2 prefetch ahead of the loaders
  printf("== loads with a sparse prefetch wave (one dword per 128 B line, pf tiles ahead)\n");
  for (int P : {1024, 4096}) for (int S : {128, 512}) for (int pf : {0, 1, 2, 4}) {
    Cfg c{}; c.R = 128; c.S = S; c.P = P; c.NL = 4; c.INF = 8; c.ntiles = (int)(bytes / ((size_t)c.R * P)); c.order = 0; c.nload = 1; c.pf = pf;
    snprintf(nm, sizeof nm, "load+pf P=%4d S=%4d NL=4 INF=8 pf=%d", P, S, pf);
    run(nm, c, x, bytes, y, grid, (double)bytes);
  }
  return 0;
}
