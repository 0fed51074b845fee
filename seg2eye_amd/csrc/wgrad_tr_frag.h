// Transpose-read MFMA fragments of the patch-resident weight-gradient kernels (conv_wgrad_patch.hip, conv_wgrad_batch.hip).
#pragma once
#include "common.h"

namespace {

// One MFMA operand = rows r and r+4 of a 4x16 transpose block (lane roles: conv_wgrad.hip).  The reads are inline asm with
// hand-counted waits: through the builtin the compiler puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 that
// follows an LDS-DMA (it cannot tell the DMA's destination stage from the stage being read), which serialises the next
// slab's loads with this slab's MFMAs.
struct TrFrag { u32x2_t lo, hi; };
template <int HI_OFF>
__device__ __forceinline__ void tr_issue(TrFrag& f, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(addr), "n"(HI_OFF) : "memory");
}
// all but the newest KEEP LDS reads are back; the fragments are operands so no MFMA on them can move above the wait
template <int KEEP>
__device__ __forceinline__ void tr_ready(TrFrag& b) {
    static_assert(KEEP >= 0 && KEEP <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b.lo), "+v"(b.hi) : "n"(KEEP) : "memory");
}
template <int KEEP>
__device__ __forceinline__ void tr_ready(TrFrag& a, TrFrag& b) {
    static_assert(KEEP >= 0 && KEEP <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(KEEP) : "memory");
}
__device__ __forceinline__ bf16x8_t tr_operand(const TrFrag& f) {
    return __builtin_bit_cast(bf16x8_t, u32x4_t{f.lo.x, f.lo.y, f.hi.x, f.hi.y});
}

}  // namespace
