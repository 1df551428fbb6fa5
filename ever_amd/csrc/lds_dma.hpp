// Global -> LDS by DMA (`buffer_load_dwordx4 ... offen lds`), hand-counted completion, raw barrier: shared by the kernels
// whose operands reach LDS without a VGPR round trip (conv_wgrad_tr.hip, conv1x1_dma.hip).
#pragma once
#include "common.hpp"

namespace evk {

constexpr uint32_t kDmaOOB = 0x80000000u;  // beyond every buffer's num_records: the DMA writes zeros

// LDS-DMA through inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front
// of the next LDS read it cannot prove disjoint — i.e. in front of every fragment read, draining the steps that are meant
// to stay in flight (seen in the ISA of the builtin form).  An asm statement is invisible to that bookkeeping; its
// completion is counted by hand (wait_vmcnt + ring_barrier).  M0 = LDS destination base, saved and restored.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  return i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// 64 lanes x 16 bytes: lane i's bytes [voff, voff + 16) of the buffer land at LDS byte lds_byte + 16 i
__device__ __forceinline__ void dma16(i32x4 rsrc, uint32_t lds_byte, uint32_t voff) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
// makes a value opaque to the optimiser: without it the loop-invariant (lane constant + immediate) sums of all the
// fragment addresses are hoisted out of the step loop into as many VGPRs (spills) instead of one base + offset fields
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// no fence: a __syncthreads would wait for vmcnt(0), i.e. drain the DMA that is meant to stay in flight across the barrier;
// lgkmcnt(0): this wave's fragment reads of the slot the next DMA overwrites have returned; the "memory" clobber keeps the
// compiler from moving DMA issue or LDS reads across it
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace evk
