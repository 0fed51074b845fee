// Shared device/host helpers for the Seg2Eye gfx950 kernels.
// gfx950 (CDNA4) only: 64-wide wavefronts, MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/seg2eye_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// native 16-B vector used for ALL type-punned 16-B global/LDS accesses: HIP's uint4 struct copies lower to
// memcpy (defeats SROA), and without may_alias clang's TBAA miscompiles float memory read through it
typedef __attribute__((ext_vector_type(4), may_alias)) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2), may_alias)) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---------------------------------------------------------------- error reporting (host)
void s2e_set_error(const char* fmt, ...);
#define S2E_FAIL(code, ...) do { s2e_set_error(__VA_ARGS__); return (code); } while (0)
#define S2E_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) S2E_FAIL(S2E_ERR_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------- 16-byte vector <-> float lanes
template <typename T> struct Vec;
template <> struct Vec<float>  { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b16) { return __builtin_bit_cast(float, b16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    bf16_t h = (bf16_t)f;                                   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

// NOTE: take the vector BY VALUE and index with []: hipcc (ROCm 7.2) miscompiles
// __builtin_bit_cast(float, r.x) on a const-reference ext-vector (loads element 0 only).
template <typename T> __device__ __forceinline__ void unpack16(u32x4_t r, float* f);
template <> __device__ __forceinline__ void unpack16<float>(u32x4_t r, float* f) {
    const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    f[0] = __builtin_bit_cast(float, r0); f[1] = __builtin_bit_cast(float, r1);
    f[2] = __builtin_bit_cast(float, r2); f[3] = __builtin_bit_cast(float, r3);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(u32x4_t r, float* f) {
    const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    f[0] = bf16_bits_to_f32(r0 & 0xffffu); f[1] = __builtin_bit_cast(float, r0 & 0xffff0000u);
    f[2] = bf16_bits_to_f32(r1 & 0xffffu); f[3] = __builtin_bit_cast(float, r1 & 0xffff0000u);
    f[4] = bf16_bits_to_f32(r2 & 0xffffu); f[5] = __builtin_bit_cast(float, r2 & 0xffff0000u);
    f[6] = bf16_bits_to_f32(r3 & 0xffffu); f[7] = __builtin_bit_cast(float, r3 & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* f);
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* f) {
    return u32x4_t{__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                   __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3])};
}
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* f) {
    return u32x4_t{f32_to_bf16_bits(f[0]) | (f32_to_bf16_bits(f[1]) << 16),
                   f32_to_bf16_bits(f[2]) | (f32_to_bf16_bits(f[3]) << 16),
                   f32_to_bf16_bits(f[4]) | (f32_to_bf16_bits(f[5]) << 16),
                   f32_to_bf16_bits(f[6]) | (f32_to_bf16_bits(f[7]) << 16)};
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> one dword of two bf16 (RNE) in ONE v_cvt_pk_bf16_f32 (pack16 converts element by element: cvt + shift + or per pair)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <typename T> __device__ __forceinline__ float load1(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void store1(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 64 lanes in six DPP adds (VALU only, unlike the ds_bpermute butterflies above); the total is valid in LANE 63 only
__device__ __forceinline__ float wave_sum_last(float v) {
#define S2E_DPP_ADD(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
    S2E_DPP_ADD(0xB1, 0xF);        // quad_perm [1,0,3,2]
    S2E_DPP_ADD(0x4E, 0xF);        // quad_perm [2,3,0,1]
    S2E_DPP_ADD(0x141, 0xF);       // row_half_mirror
    S2E_DPP_ADD(0x140, 0xF);       // row_mirror: every lane of a 16-lane row holds the row's sum
    S2E_DPP_ADD(0x142, 0xA);       // row_bcast15 into rows 1 and 3
    S2E_DPP_ADD(0x143, 0xC);       // row_bcast31 into rows 2 and 3
#undef S2E_DPP_ADD
    return v;
}

// XCD-aware bijective remap of a 1-D grid: blocks that share (bid % 8) sit on one XCD
// (observed round-robin placement; speed only, never correctness) and get a contiguous
// range of logical tile ids, so neighbouring tiles reuse operand panels in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// compile-time loop: f(integral_constant<int, I>) for I in [0, N) -- indices stay constants, so small
// per-thread arrays are always register-allocated (never scratch / LDS-promoted)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Zero-fill as a KERNEL, never hipMemsetAsync: inside a captured hipGraph (ROCm 7.x) memset nodes were
// observed to race with neighbouring kernel nodes (first replay fine, later replays corrupt), while
// kernel nodes are strictly ordered.  bytes must be a multiple of 4; ptr 4-byte aligned.
__global__ void s2e_zero_kernel(uint32_t* __restrict__ p, size_t n_words);
int s2e_zero_async(void* ptr, size_t bytes, hipStream_t st);

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// S2E_DETERMINISTIC=1 (read once per process): every gradient of the train step is summed in a fixed order -- the generic weight
// gradient's partial tiles for every split launch, one reduction pass instead of several combined with float atomics, the patch
// weight gradient's bias sums through the workspace -- so that two runs of a trainer produce the same bits.
// Costs ~0.5 ms per step; off by default.  (The logged loss VALUES and the bilinear resize's backward still use float atomics.)
int s2e_deterministic(void);
