// The PatchGAN's first layer (discriminator.py:78-85 of the reference): 4x4 stride-2 pad-2 convolution of an 8-channel map (the one-hot
// label map + the image, padded to 8 channels: 16 bytes a pixel) to 64 channels, bias + LeakyReLU -- bf16, NHWC, gfx950.
//
// K = 16 taps x 8 channels = 128 and 64 output channels: 4.4 GFLOP against 51 MB of tensors at 256^2 x 16 images, an HBM-bound layer.  The
// implicit GEMM (conv_igemm.hip) gathers its A operand tap by tap -- sixteen 16-byte pieces per output pixel, every second pixel of a
// row -- and runs it at 1.3 TB/s (40 us).  Here the input rows land in LDS as they lie in memory (coalesced, LDS-DMA, out-of-map pixels as
// zeros by the buffer bounds check) and ONE KERNEL ROW IS ONE MFMA K-step: the four taps (ky, 0..3) of output pixel X are the input pixels
// 2X .. 2X + 3 of row 2Y + ky -- 64 contiguous bytes -- so a lane of v_mfma_f32_16x16x32_bf16 (k = 8 kq .. + 7 of column j) reads its
// operand with one ds_read_b128 at ((2Y + ky) * pitch + 2 (X0 + j) + kq) * 16.
//
//   workgroup : 256 threads, four output rows of one image (one per wave); LDS: the ten input rows they touch, 32 G + 8 pixels each
//               (G = 16-pixel groups of an output row; <= 47 KB: three workgroups a CU).
//   operands  : weights are the A operand (M = output channels), all 16 K-steps x 4 row blocks of them in registers (64 VGPRs), read once
//               from the standard packed matrix [co][tap * 8 + ci]; pixels are B.  A row block m = 2 jj + e holds the channels
//               32 jj + 8 q + 4 e + r in its rows 4 q + r, so the accumulators of lane (q, pixel j) are 8 CONSECUTIVE channels per jj:
//               the epilogue is two 16-byte stores a pixel, bias (the accumulators start from it) / residual / LeakyReLU in registers.
#include "conv_small.h"
#include <stdlib.h>

namespace {

struct TrFragC { u32x2_t lo, hi; };

struct C8Params {
    const void* x; const void* w; const float* bias; const void* res; void* y;
    int N, Hi, Wi, Ho, Wo, Kpad, out_act;
    int G, pitch, yblocks;            // 16-pixel groups per output row; LDS row pitch in pixels; row blocks per image
    unsigned x_bytes;
};

constexpr int C8_RT = 4, C8_ROWS = 2 * C8_RT + 2;       // output rows a workgroup, input rows it stages
constexpr int C8_MAXG = 9, C8_MAXPITCH = 32 * C8_MAXG + 8;
constexpr int C8_LDS = C8_ROWS * C8_MAXPITCH * 16;      // 47,360 B

__global__ __launch_bounds__(256, 2) void conv_c8s2_fwd_kernel(const C8Params p) {
    typedef bf16_t T;
    __shared__ __attribute__((aligned(16))) char smem[C8_LDS + 1024];     // (+ the tail of the last 64-pixel piece)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / p.yblocks, Y0 = (blockIdx.x - n * p.yblocks) * C8_RT;
    const int pitch = p.pitch;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- stage the input rows 2 Y0 - 2 .. 2 Y0 + 7: LDS pixel (r, c) = input pixel (2 Y0 - 2 + r, c - 2)
    const int npx = C8_ROWS * pitch, npieces = (npx + 63) >> 6;
    for (int q = wave; q < npieces; q += 4) {
        const int pp = 64 * q + lane;
        const int r = pp / pitch, c = pp - r * pitch;
        const int iy = 2 * Y0 - 2 + r, ix = c - 2;
        const bool ok = r < C8_ROWS && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const unsigned off = ok ? 16u * (unsigned)((n * p.Hi + iy) * p.Wi + ix) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(smem + q * 1024), 16, (int)off, 0, 0, 0);
    }

    // ---- weights: A fragment (m, ky) of lane (i = lane & 15, kq = lane >> 4) = row co(m, i), k = (ky * 4 + kq) * 8 .. + 7
    const int i16 = lane & 15, kq = lane >> 4;
    const T* __restrict__ wg = (const T*)p.w;
    u32x4_t wa[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int co = 32 * (m >> 1) + 8 * (i16 >> 2) + 4 * (m & 1) + (i16 & 3);
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) wa[m][ky] = *(const u32x4_t*)(wg + (size_t)co * p.Kpad + (ky * 4 + kq) * 8);
    }
    // bias of this lane's channels: block m, register r -> channel 32 (m >> 1) + 8 kq + 4 (m & 1) + r
    f32x4_t binit[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c0 = 32 * (m >> 1) + 8 * kq + 4 * (m & 1);
        binit[m] = p.bias ? *(const f32x4_t*)(p.bias + c0) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int Y = Y0 + wave;
    if (Y >= p.Ho) return;
    const int b_base = ((2 * wave) * pitch + 2 * i16 + kq) * 16;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    const size_t row0 = ((size_t)n * p.Ho + Y) * p.Wo;
    for (int g = 0; g < p.G; ++g) {
        u32x4_t xb[4];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) xb[ky] = *(const u32x4_t*)(smem + b_base + (ky * pitch + 32 * g) * 16);
        f32x4_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = binit[m];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[m][ky]), __builtin_bit_cast(bf16x8_t, xb[ky]), acc[m], 0, 0, 0);
        const int X = 16 * g + i16;
        if (X < p.Wo) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                float v[8] = {acc[2 * jj][0], acc[2 * jj][1], acc[2 * jj][2], acc[2 * jj][3],
                              acc[2 * jj + 1][0], acc[2 * jj + 1][1], acc[2 * jj + 1][2], acc[2 * jj + 1][3]};
                const size_t o = (row0 + X) * 64 + 32 * jj + 8 * kq;
                if (rg) {
                    float rr[8];
                    unpack16<T>(*(const u32x4_t*)(rg + o), rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rr[e];
                }
                if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = lrelu02(v[e]);
                }
                *(u32x4_t*)(yg + o) = u32x4_t{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
            }
        }
    }
}

// ---- the data gradient of the same layer (the G step: d loss / d fake image): dx[n][oy][ox][8] from gy[n][Y][X][64].
// With py = oy + 2 = 2 Y + ky the taps of an output pixel are ky = (py & 1) + 2 dy, Y = (py >> 1) - dy, dy = 0, 1 (and the same in x): the
// four output pixels (2 ly - 2 + pa, 2 lx - 2 + pb) of one plane-local position (ly, lx) read the SAME 2 x 2 neighbourhood of gy, so the
// layer is one GEMM  out[position][(pa, pb, ci) = 32] = gy2x2[position][(dy, dx, co) = 256] x W'[256][32]:  eight K-steps of
// v_mfma_f32_16x16x32_bf16 and two row blocks.  No LDS: W' (16 KB) sits in registers -- read as 16-byte pieces of the transposed pack
// [ci][tap * 64 + co] -- and a lane fetches its 16 bytes of gy per K-step straight from global memory (every gy pixel is read by four
// positions: L1 / L2 hits; out-of-map pixels are zeros by the buffer bounds check).  The weights are the A operand with row 4 q + r of
// block m = (plane q, channel 4 m + r), so lane (q, position j) ends with the 8 channels of ONE output pixel: one 16-byte store.
struct C8DParams {
    const void* gy; const void* w; void* dx;
    int N, Hg, Wg, Ho, Wo, Kpad;      // gy map; dx map; row pitch of the transposed pack
    int LY, LXG;                       // plane-local rows 1 .. LY; 16-position groups per row
    unsigned gy_bytes;
};

__global__ __launch_bounds__(256, 2) void conv_c8s2_dgrad_kernel(const C8DParams p) {
    typedef bf16_t T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / p.LY, ly = blockIdx.x - n * p.LY + 1;
    const int i16 = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.gy, 0, (int)p.gy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const T* __restrict__ wt = (const T*)p.w;
    // A fragment (m, K-step s = (dy, dx, h)) of lane (i16 = 4 q' + r', kq): W_t[ci = 4 m + r'][tap * 64 + 32 h + 8 kq ..], tap = (pa + 2 dy, pb + 2 dx)
    u32x4_t wa[2][8];
    {
        const int qp = i16 >> 2, rp = i16 & 3, pa = qp >> 1, pb = qp & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int dy = s >> 2, dx = (s >> 1) & 1, h = s & 1;
                const int tap = (pa + 2 * dy) * 4 + pb + 2 * dx;
                wa[m][s] = *(const u32x4_t*)(wt + (size_t)(4 * m + rp) * p.Kpad + tap * 64 + 32 * h + 8 * kq);
            }
    }
    T* __restrict__ dxg = (T*)p.dx;
    const int pa = kq >> 1, pb = kq & 1;              // the plane this lane writes
    const int oy = 2 * ly + pa - 2;
    for (int g = wave; g < p.LXG; g += 4) {
        const int lx = 16 * g + i16 + 1;
        u32x4_t gb[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int dy = s >> 2, dx = (s >> 1) & 1, h = s & 1;
            const int Y = ly - dy, X = lx - dx;
            const bool ok = (unsigned)Y < (unsigned)p.Hg && (unsigned)X < (unsigned)p.Wg;
            const unsigned off = ok ? 2u * (unsigned)(((n * p.Hg + Y) * p.Wg + X) * 64 + 32 * h + 8 * kq) : OOB;
            gb[s] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)off, 0, 0));
        }
        f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[m][s]), __builtin_bit_cast(bf16x8_t, gb[s]), acc[m], 0, 0, 0);
        const int ox = 2 * lx + pb - 2;
        if ((unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo)
            *(u32x4_t*)(dxg + ((size_t)(n * p.Ho + oy) * p.Wo + ox) * 8) =
                u32x4_t{pack2_bf16(acc[0][0], acc[0][1]), pack2_bf16(acc[0][2], acc[0][3]), pack2_bf16(acc[1][0], acc[1][1]), pack2_bf16(acc[1][2], acc[1][3])};
    }
}

// ---- the weight gradient of the same layer (the D step): dW[co][(ky, kx, ci)] += sum over output pixels gy[pixel][co] * x[2Y + ky - 2][2X + kx - 2][ci].
// A slab is one output row: its gy pixels (128-B rows, LDS-DMA, swizzled as conv_wgrad_flat.hip's x operand) and the four input rows it
// touches (16 B a pixel, as they lie in memory).  The contraction runs over pixels: gy fragments by ds_read_b64_tr_b16; the B operand's
// column (ky, kx, ci) of pixel X is the 2 bytes at ((ky) * pitch + 2 X + kx) * 16 + 2 ci of the row block -- consecutive pixels 32 B apart,
// so a lane assembles its 8 pixels with eight ds_read_u16(_d16_hi) at immediate offsets of one address (conv_wgrad_patch.hip's 8-channel
// kernel does the same at stride 16).  Four waves = 2 (co blocks of 32) x 2 (kernel-row pairs); v_mfma_f32_32x32x16_bf16; 256 persistent
// workgroups add their [64][128] tiles into dW with fp32 atomics; the bias gradient is one extra MFMA against ones.
#define S2E_C8_U16_PAIR(lo, hi, addr, off_lo, off_hi) \
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(off_lo) : "memory"); \
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(off_hi) : "memory")

struct C8WJob { const void* x; const void* gy; float* dw; float* dbias; float* part; int N, Hi, Wi, Ho, Wo, G, pitch, rows; unsigned x_bytes, gy_bytes; };
constexpr int C8W_TILE = 64 * 128 + 64;             // a workgroup's partial tile: dW [64][128], then the bias sums
constexpr int C8W_MAX_JOBS = 4;
struct C8WMulti { int n; int first[C8W_MAX_JOBS + 1]; C8WJob j[C8W_MAX_JOBS]; };

constexpr int C8W_XB = 4 * C8_MAXPITCH * 16 + 1024, C8W_GB = 16 * C8_MAXG * 128, C8W_STAGE = C8W_XB + C8W_GB;     // 19,968 + 18,432

__global__ __launch_bounds__(256, 2) void conv_c8s2_wgrad_kernel(const C8WMulti b) {
    typedef bf16_t T;
    __shared__ __attribute__((aligned(16))) char smem[2 * C8W_STAGE];
    typedef __attribute__((address_space(3))) void* lptr_t;
    int jb = 0;
    while (jb + 1 < b.n && (int)blockIdx.x >= b.first[jb + 1]) ++jb;
    const C8WJob& p = b.j[jb];
    const int blk = (int)blockIdx.x - b.first[jb], nblk = b.first[jb + 1] - b.first[jb];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cbk = wave >> 1, half = wave & 1;       // 32-co block; kernel rows 2 half, 2 half + 1
    const int hh = lane >> 5, l31 = lane & 31;
    const int pitch = p.pitch, G = p.G;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.gy, 0, (int)p.gy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    const int nxp = (4 * pitch + 63) >> 6, ngp = 2 * G;          // 64-pixel x pieces; 8-pixel gy pieces
    auto stage_slab = [&](int item, int buf) __attribute__((always_inline)) {
        const int n = item / p.Ho, Y = item - n * p.Ho;
        for (int q = wave; q < nxp; q += 4) {
            const int pp = 64 * q + lane;
            const int r = pp / pitch, c = pp - r * pitch;
            const int iy = 2 * Y - 2 + r, ix = c - 2;
            const bool ok = r < 4 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const unsigned off = ok ? 16u * (unsigned)((n * p.Hi + iy) * p.Wi + ix) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(smem + buf * C8W_STAGE + q * 1024), 16, (int)off, 0, 0, 0);
        }
        for (int q = wave; q < ngp; q += 4) {
            const int X = 8 * q + (lane >> 3);
            const int ch = ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8;
            const unsigned off = X < p.Wo ? 2u * (unsigned)(((n * p.Ho + Y) * p.Wo + X) * 64 + ch) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lptr_t)(smem + buf * C8W_STAGE + C8W_XB + q * 1024), 16, (int)off, 0, 0, 0);
        }
    };

    const int i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, g2 = (lane >> 4) & 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // gy fragment of the group at LDS pixel 16 g: pixel r = 16 g + 8 hh + q4, chunk (cbk * 4 + 2 g2 + (pq >> 1)) with bit 2 flipped by bit 1 of r
    const uint32_t a_lane = lds0 + C8W_XB + (((uint32_t)(((8 * hh + q4) << 7) + (((cbk * 4 + 2 * g2 + (pq >> 1)) << 4) + (pq & 1) * 8))) ^ (uint32_t)((q4 & 2) << 5));
    // B column l31 = (kx = l31 >> 3, ci = l31 & 7) of kernel row ky: byte ((ky * pitch + 2 (16 g + 8 hh + i) + kx) * 16 + 2 ci, i = 0 .. 7
    uint32_t b_lane[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) b_lane[c] = lds0 + (uint32_t)((((2 * half + c) * pitch + 16 * hh + (l31 >> 3)) << 4) + (l31 & 7) * 2);
    f32x16_t acc[2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; accb[r] = 0.f; }
    u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    asm volatile("" : "+v"(ones));
    const bool want_bias = p.dbias != nullptr && half == 0;

    int item = blk;
    if (item < p.rows) stage_slab(item, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (; item < p.rows; item += nblk) {
        if (item + nblk < p.rows) stage_slab(item + nblk, buf ^ 1);
        const uint32_t stage = (uint32_t)(buf * C8W_STAGE);
        for (int g = 0; g < G; ++g) {
            TrFragC A;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(A.lo) : "v"(a_lane + stage + (uint32_t)(g << 11)) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(A.hi) : "v"(a_lane + stage + (uint32_t)(g << 11)) : "memory");
            uint32_t B[2][4], Bh[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t ad = b_lane[c] + stage + (uint32_t)(g << 9);       // 32 pixels of the input row per 16 output pixels
                S2E_C8_U16_PAIR(B[c][0], Bh[c][0], ad, 0, 32);
                S2E_C8_U16_PAIR(B[c][1], Bh[c][1], ad, 64, 96);
                S2E_C8_U16_PAIR(B[c][2], Bh[c][2], ad, 128, 160);
                S2E_C8_U16_PAIR(B[c][3], Bh[c][3], ad, 192, 224);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(A.lo), "+v"(A.hi), "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][2]), "+v"(B[0][3]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[1][2]), "+v"(B[1][3]),
                           "+v"(Bh[0][0]), "+v"(Bh[0][1]), "+v"(Bh[0][2]), "+v"(Bh[0][3]), "+v"(Bh[1][0]), "+v"(Bh[1][1]), "+v"(Bh[1][2]), "+v"(Bh[1][3]) :: "memory");
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, u32x4_t{A.lo.x, A.lo.y, A.hi.x, A.hi.y});
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, u32x4_t{B[c][0] | Bh[c][0], B[c][1] | Bh[c][1], B[c][2] | Bh[c][2], B[c][3] | Bh[c][3]}), acc[c], 0, 0, 0);
            if (want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, ones), accb, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
    // ---- combine: register r of lane (hh, l31) = row co = 32 cbk + (r & 3) + 8 (r >> 2) + 4 hh, column 32 ky + l31.  Every workgroup of a job
    // holds a tile of the SAME 32 KB: as atomics straight into dW they serialise (512 workgroups: 115 us, measured), so the tiles go to the
    // workspace and conv_c8s2_wgrad_reduce_kernel folds them (no workspace: atomics, from a quarter of the workgroups)
    float* __restrict__ tile = p.part ? p.part + (size_t)blk * C8W_TILE : nullptr;
    if (blk < p.rows) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = (32 * cbk + (r & 3) + 8 * (r >> 2) + 4 * hh) * 128 + 32 * (2 * half + c) + l31;
                if (tile) tile[e] = acc[c][r]; else atomicAdd(p.dw + e, acc[c][r]);
            }
        if (half == 0 && l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * cbk + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (tile) tile[64 * 128 + co] = accb[r]; else if (p.dbias) atomicAdd(p.dbias + co, accb[r]);
            }
        }
    }
}

// dW / dbias += the workgroups' partial tiles: thread e of (blockIdx.x, job blockIdx.z) sums the tiles t = blockIdx.y, + 32, ... of element e
__global__ __launch_bounds__(256) void conv_c8s2_wgrad_reduce_kernel(const C8WMulti b) {
    const C8WJob& p = b.j[blockIdx.z];
    if (!p.part) return;
    const int e = blockIdx.x * 256 + threadIdx.x, ntile = min(b.first[blockIdx.z + 1] - b.first[blockIdx.z], p.rows);
    if (e >= C8W_TILE) return;
    float a = 0.f;
    for (int t = blockIdx.y; t < ntile; t += 32) a += p.part[(size_t)t * C8W_TILE + e];
    if (e < 64 * 128) atomicAdd(p.dw + e, a);
    else if (p.dbias) atomicAdd(p.dbias + (e - 64 * 128), a);
}

int c8_on() {
    static const int v = [] { const char* e = getenv("S2E_CONV_C8"); return e ? atoi(e) : 1; }();
    return v;
}

}  // namespace

// the shapes of conv_c8s2_fwd_kernel: bf16, 8 -> 64 channels, 4x4 stride 2 pad 2, no input activation, no mask, maps up to 144 output columns
bool s2e_c8s2_fwd_ok(int dtype, const s2e_conv_desc* d) {
    if (!c8_on() || dtype != S2E_BF16 || d->transposed || d->Cin != 8 || d->Cout != 64 || d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 2) return false;
    if (d->in_act != S2E_ACT_NONE || d->aux_mode != S2E_AUX_NONE || (d->out_act != S2E_ACT_NONE && d->out_act != S2E_ACT_LRELU)) return false;
    if (d->Ho != (d->Hi + 4 - 4) / 2 + 1 || d->Wo != (d->Wi + 4 - 4) / 2 + 1 || d->Wo > 16 * C8_MAXG || d->Wo < 1) return false;
    return (long)d->N * d->Hi * d->Wi * 16 < (1L << 31);
}

int s2e_c8s2_fwd_launch(const SmallConvParams& sp, hipStream_t st) {
    C8Params p{};
    p.x = sp.x; p.w = sp.w; p.bias = sp.bias; p.res = sp.res; p.y = sp.y;
    p.N = sp.N; p.Hi = sp.Hi; p.Wi = sp.Wi; p.Ho = sp.Ho; p.Wo = sp.Wo; p.Kpad = sp.Kpad; p.out_act = sp.out_act;
    p.G = ceil_div(sp.Wo, 16);
    p.pitch = 32 * p.G + 8;
    p.yblocks = ceil_div(sp.Ho, C8_RT);
    p.x_bytes = (unsigned)((long)sp.N * sp.Hi * sp.Wi * 16);
    conv_c8s2_fwd_kernel<<<sp.N * p.yblocks, 256, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_c8s2_fwd_kernel");
    return S2E_OK;
}

// the data gradient: transposed, 64 -> 8 channels, no activation, no mask
bool s2e_c8s2_dgrad_ok(int dtype, const s2e_conv_desc* d) {
    if (!c8_on() || dtype != S2E_BF16 || !d->transposed || d->Cin != 64 || d->Cout != 8 || d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 2) return false;
    if (d->in_act != S2E_ACT_NONE || d->out_act != S2E_ACT_NONE || d->aux_mode != S2E_AUX_NONE) return false;
    if (d->Hi != d->Ho / 2 + 1 || d->Wi != d->Wo / 2 + 1) return false;            // (Hi, Wi: the gy map; Ho, Wo: dx = the forward's input)
    return (long)d->N * d->Hi * d->Wi * 128 < (1L << 31);
}

int s2e_c8s2_dgrad_launch(const SmallConvParams& sp, hipStream_t st) {
    if (sp.res) S2E_FAIL(S2E_ERR_UNSUPPORTED, "conv_c8s2_dgrad: no residual");
    C8DParams p{};
    p.gy = sp.x; p.w = sp.w; p.dx = sp.y;
    p.N = sp.N; p.Hg = sp.Hi; p.Wg = sp.Wi; p.Ho = sp.Ho; p.Wo = sp.Wo; p.Kpad = sp.Kpad;
    p.LY = (sp.Ho + 1) >> 1;
    p.LXG = ceil_div((sp.Wo + 1) >> 1, 16);
    p.gy_bytes = (unsigned)((long)sp.N * sp.Hi * sp.Wi * 128);
    conv_c8s2_dgrad_kernel<<<sp.N * p.LY, 256, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_c8s2_dgrad_kernel");
    return S2E_OK;
}

// the weight gradient (forward description: transposed = 0)
bool s2e_c8s2_wgrad_ok(int dtype, const s2e_conv_desc* d) {
    if (!c8_on() || s2e_deterministic() || dtype != S2E_BF16 || d->transposed || d->Cin != 8 || d->Cout != 64 || d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 2) return false;
    if (d->in_act != S2E_ACT_NONE) return false;
    if (d->Ho != d->Hi / 2 + 1 || d->Wo != d->Wi / 2 + 1 || d->Wo > 16 * C8_MAXG || d->Wo < 1) return false;
    return (long)d->N * d->Hi * d->Wi * 16 < (1L << 31) && (long)d->N * d->Ho * d->Wo * 128 < (1L << 31);
}

size_t s2e_c8s2_wgrad_workspace_bytes(int n_jobs) {
    return (size_t)ceil_div(n_jobs, C8W_MAX_JOBS) * 512 * C8W_TILE * sizeof(float);
}

int s2e_c8s2_wgrad_launch(const s2e_wgrad_multi_job* jobs, const int* idx, int n_all, void* workspace, size_t workspace_bytes, hipStream_t st) {
    static const double total_wg = [] { const char* e = getenv("S2E_C8W_WGS"); return e ? atof(e) : 512.0; }();
    char* ws = (char*)workspace;
    for (int base = 0; base < n_all; base += C8W_MAX_JOBS) {
        const int n = n_all - base < C8W_MAX_JOBS ? n_all - base : C8W_MAX_JOBS;
        C8WMulti b{};
        b.n = n;
        const bool tiles = ws && workspace_bytes >= (size_t)(base / C8W_MAX_JOBS + 1) * 512 * C8W_TILE * sizeof(float) && total_wg <= 512.0;
        double work = 0.0;
        for (int i = 0; i < n; ++i) work += (double)jobs[idx[base + i]].d.N * jobs[idx[base + i]].d.Ho * jobs[idx[base + i]].d.Wo;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const s2e_wgrad_multi_job& J = jobs[idx[base + i]];
            const s2e_conv_desc* d = &J.d;
            if (!s2e_c8s2_wgrad_ok(S2E_BF16, d)) S2E_FAIL(S2E_ERR_UNSUPPORTED, "conv_c8s2_wgrad: job %d is not a shape of this kernel", idx[base + i]);
            if (!J.x || !J.gy || !J.dw) S2E_FAIL(S2E_ERR_ARG, "conv_c8s2_wgrad: null pointer in job %d", idx[base + i]);
            C8WJob& p = b.j[i];
            p.x = J.x; p.gy = J.gy; p.dw = J.dw; p.dbias = J.dbias;
            p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo;
            p.G = ceil_div(d->Wo, 16); p.pitch = 32 * p.G + 8; p.rows = d->N * d->Ho;
            p.x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * 16);
            p.gy_bytes = (unsigned)((long)d->N * d->Ho * d->Wo * 128);
            int share = (int)((tiles ? total_wg : total_wg / 4) * ((double)d->N * d->Ho * d->Wo) / work);
            share = share < 1 ? 1 : (share > p.rows ? p.rows : share);
            if (tiles) p.part = (float*)(ws + ((size_t)(base / C8W_MAX_JOBS) * 512 + blocks) * C8W_TILE * sizeof(float));
            b.first[i] = blocks;
            blocks += share;
        }
        b.first[n] = blocks;
        conv_c8s2_wgrad_kernel<<<blocks, 256, 0, st>>>(b);
        S2E_CHECK_LAUNCH("conv_c8s2_wgrad_kernel");
        if (tiles) {
            conv_c8s2_wgrad_reduce_kernel<<<dim3(ceil_div(C8W_TILE, 256), 32, n), 256, 0, st>>>(b);
            S2E_CHECK_LAUNCH("conv_c8s2_wgrad_reduce_kernel");
        }
    }
    return S2E_OK;
}
