// Internal interface of the plane-patch convolution (conv_plane.hip): the stride-2 3x3 / 4x4 layers (netE, the PatchGAN
// downsampling layers) forward and data gradient, and the 1x1 learned shortcuts, patch-resident with the weights streamed
// straight into registers.
#pragma once
#include "common.h"

enum { PLANE_NONE = 0, PLANE_K1 = 1, PLANE_K3S1 = 2, PLANE_K3S2F = 3, PLANE_K3S2D = 4, PLANE_K4S2F = 5, PLANE_K4S2D = 6, PLANE_K4S1 = 7 };

// PLANE_NONE = another kernel runs this shape; otherwise the mode conv_plane.hip runs it in.  The weight operand of such a launch
// is in the PLANE layout (s2e_pack_conv_weight with transposed | 4).
int s2e_conv_plane_mode(int dtype, const s2e_conv_desc* d);
int s2e_conv_plane_launch(int mode, const void* x, const void* w, const float* bias, const void* res, const void* aux, void* y,
                          const s2e_conv_desc* d, hipStream_t st);

// ---- the PLANE layout of a packed weight (bf16): for every (64-row group g, 32-element chunk c of the K dimension, tap t) one 4-KB
// block at ((g * nch + c) * taps + t) * 4096 bytes holding the four 16x16x32 MFMA fragments f = 0..3 in register order: lane L's 16
// bytes of fragment f at (f * 64 + L) * 16 = elements k = 32 c + 8 (L >> 4) .. + 7 of row 64 g + 32 (f >> 1) + 8 ((L & 15) >> 2) +
// 4 (f & 1) + (L & 3) -- fragments 2m and 2m + 1 interleave 4-row groups, so that a lane's accumulators of the pair are 8 consecutive
// rows (output channels).  rows = cout (forward) or cin (transposed: the data gradient's matrix, K = cout); rows past the end are
// zeros.  The K dimension must be a multiple of 32.  One thread per 16 bytes; unit = bx * 256 + threadIdx.x.
#define S2E_PACK_PLANE 4          /* s2e_pack_conv_weight / s2e_pack_job.transposed bit 2 */
__host__ __device__ static inline long s2e_plane_pack_units(int cout, int cin, int taps, int transposed) {
    const int rows = (transposed & 1) ? cin : cout, kd = (transposed & 1) ? cout : cin;
    return (long)((rows + 63) / 64) * (kd / 32) * taps * 256;
}
#ifdef __HIPCC__
__device__ __forceinline__ void pack_plane_block(const float* __restrict__ w, bf16_t* __restrict__ out, const float* __restrict__ sigma,
                                                 int cout, int cin, int taps, int transposed, int bx) {
    const bool tr = (transposed & 1) != 0, cl = (transposed & 2) != 0;
    const int rows = tr ? cin : cout, kd = tr ? cout : cin, nch = kd >> 5;
    const long unit = (long)bx * 256 + threadIdx.x;
    if (unit >= s2e_plane_pack_units(cout, cin, taps, transposed)) return;
    const int L = (int)(unit & 63), f = (int)((unit >> 6) & 3);
    long rest = unit >> 8;
    const int t = (int)(rest % taps); rest /= taps;
    const int c = (int)(rest % nch), g = (int)(rest / nch);
    const int row = 64 * g + 32 * (f >> 1) + 8 * ((L & 15) >> 2) + 4 * (f & 1) + (L & 3);
    const int k0 = 32 * c + 8 * (L >> 4);
    const float inv = sigma ? 1.f / *sigma : 1.f;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e, co = tr ? k : row, ci = tr ? row : k;
        float x = 0.f;
        if (row < rows) x = cl ? w[((size_t)co * taps + t) * cin + ci] : w[((size_t)co * cin + ci) * taps + t];
        v[e] = x * inv;
    }
    *(u32x4_t*)(out + unit * 8) = u32x4_t{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
}
#endif
