// Patch-resident convolution: 3x3 and 4x4 stride 1, and 4x4 stride 2 through a space-to-depth view (forward and data-gradient) for gfx950.
//
// Why a second conv kernel.  The generic implicit GEMM (conv_igemm.hip) re-gathers the im2col panel from L2 for every
// tap: a 128 x 128 x 64 K-step moves 32 KB through the CU's vector-memory path for 2.1 MFLOP.  That path -- not the
// MFMA pipe, not LDS, not HBM -- is what the large layers run against: with the MFMAs removed the same loop takes 80 %
// of the full kernel's time (55 GB/s per CU, 14 TB/s chip-wide; the LDS-DMA gather ceiling measured for this chip is
// 66-73 GB/s per CU), with the loads removed it runs at 1.2-1.5 PFLOP/s.  Deeper pipelines, more resident waves and a
// 256-pixel tile at the same per-tap gather all measured flat, because none of them changes bytes per FLOP enough.
//
// Here a workgroup owns a RECTANGLE of up to 256 output pixels (4 x 64, 8 x 32, 16 x 16, 7 x 34 ...: any width <= 64,
// chosen per shape by the host) and, per 64-channel chunk of Cin, brings the input patch with its halo
// ((TH+KS-1) x (TW+KS-1) pixels x 128 B, <= 400 pixels = 50 KB) into LDS ONCE; the KS x KS taps are shifted views of that
// patch.  Per chunk and 256 pixels (3x3) the vector-memory path carries 50 KB of activations + 9 x 16 KB of weights =
// 194 KB instead of 9 x (32 + 16) = 432 KB at the same tile (576 KB as two 128-pixel tiles).
//
//   LDS        : 2 patch buffers (chunk c is read while c+1 lands, one 1-KiB piece per thread per tap) + 3 weight K-step
//                stages (K-step kt+2 in flight while kt is multiplied) = 2 x 51,200 + 3 x 16,384 B = 151.5 KB -> one
//                512-thread workgroup per CU, PERSISTENT over tiles (tile i+1's first loads are issued before tile i is
//                written out).
//   patch image: pixel pp = py * PW + px at byte pp * 128, 16-B chunk index XORed with (pp >> 1) & 7 (source-side
//                swizzle of the LDS-DMA, as in conv_igemm.hip).  With a width that is a multiple of 32 an A fragment is
//                32 consecutive tx of one tile row = 32 consecutive pp at ANY tap shift, so every ds_read_b128 stays
//                conflict-free; other widths let a fragment span two rows (a 2-way conflict on a few lanes).
//   waves      : 8 = 4 (pixels) x 2 (channels), wave tile 64 x BN/2, v_mfma_f32_32x32x16_bf16 (32x32x2 f32 for fp32);
//                fragment reads hand-pipelined (two register sets, counted lgkmcnt waits, bare s_barrier).
//   K order    : chunk-major, tap-minor; the packed weight matrix (row co, k = tap * Cin + ci) is used as it is.
//   data-gradient (stride 1): the same kernel with the patch origin moved to o + pad - (KS-1) and the taps mirrored.
//   split      : long-K layers with few tiles are split over channel chunks (fp32 slabs + conv_igemm's finish kernel).
//   epilogue   : fp32 accumulators -> LDS one 64-row wave row at a time (staged in the idle patch buffer) -> 16-B row
//                stores with bias / residual / activation / mask fused (same contract as conv_igemm.hip).
#include "conv_patch.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) const uint32_t pz_zero16[4] = {0u, 0u, 0u, 0u};

struct PatchParams {
    const void* x; const void* w; const float* bias; const void* res; const void* aux; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Kpad;
    int org;                      // patch origin = tile origin + org (forward: -pad; data-gradient: pad - (KS-1))
    int flip;                     // data-gradient: patch offset t pairs with weight tap T-1-t
    int out_act, aux_mode;
    int tw, th;                   // rectangle of output pixels: width (<= 64), height; tw * th <= 256 (rows past it idle)
    int tw_shift;                 // log2(tw) when tw is a power of two (every rectangle of the power-of-two layers), else -1
    int tiles_x, tiles_y, tiles_n, tiles;
    int splits, cps, tiles_out, M;    // split-K over channel chunks: split s owns chunks [s * cps, (s+1) * cps); tiles = tiles_out * splits
    float* partial;                   // splits > 1: fp32 slabs [splits][M][Cout], combined by conv_finish_kernel (conv_igemm.hip)
    // FUSE (SPADE+Style modulation in the epilogue of the [gamma | beta] conv; Cout = 2 * mC, bias = [b_gamma | b_beta]):
    const void* mx;                   // the tensor being modulated, (N, Ho, Wo, mC)
    const float* mstats;              // (N, mC, 2) {mean, rstd}
    const float* mstyle; int msld;    // style codes {s0 | s1}: row n at mstyle + n * msld, 2 * mC floats
    void* mgamma;                     // NULL, or (N, Ho, Wo, mC): gamma (with its bias) is stored for the backward pass
    int mC, mlrelu;
    int mup;                          // mx is (N, Ho/2, Wo/2, mC): the nearest 2x upsampling that precedes the block, folded into the read
    // FUSE, label-sparse launches: only the rectangles listed in rect_list[0 .. *rect_count) are computed (the others are
    // label-uniform and take gamma / beta from a per-class table: s2e_spade_modulate_uniform); NULL = every rectangle
    const int* rect_list; const int* rect_count;
    // S2D (stride-2 4x4 pad-2 layers as a 2x2 stride-1 conv over the space-to-depth VIEW of the full-resolution tensor):
    int c_shift;                      // log2 of the full-resolution tensor's channel count C (a power of two >= 64)
};

template <typename T> struct PMfma;
template <> struct PMfma<bf16_t> {
    static __device__ __forceinline__ void run(u32x4_t a, u32x4_t b, f32x16_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct PMfma<float> {           // same k permutation on both operands: exact contraction (see conv_igemm.hip)
    static __device__ __forceinline__ void run(u32x4_t a, u32x4_t b, f32x16_t& acc) {
        const f32x4_t fa = __builtin_bit_cast(f32x4_t, a), fb = __builtin_bit_cast(f32x4_t, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
    }
};

// FUSE: the launch is the [gamma | beta] conv of a SPADE (Cout = 2C, BN = 128).  A Cout tile then pairs 64 gamma rows of
// the packed weight with the 64 beta rows of the SAME channels, and the epilogue -- instead of writing 2C channels of
// gamma, beta for a second kernel to read back -- reads x once and writes
//     out = [lrelu] 0.5 * ((x - mean) * rstd * (1 + gamma) + beta + x * (1 + s0) + s1)         (C channels)
// (normalization.py:91-105,163-169,184-192 of the reference in one pass); gamma itself is stored only when the backward
// pass will need it.  gamma and beta stay fp32 from the accumulators to the result.
//
// S2D: a 4x4 stride-2 pad-2 convolution (the PatchGAN discriminators' downsampling layers, discriminator.py:84-96) is a 2x2
// stride-1 convolution over the space-to-depth view X'[n][Y][X][(py, px, c)] = X[n][2Y + py][2X + px][c] of its input, and that
// view needs no copy: in NHWC memory the (px, c) half of an X' pixel is 2C contiguous elements of row 2Y + py.  A 64-channel
// chunk of X' (C a power of two >= 64) therefore lies inside ONE phase (py, px): the patch DMA only adds a per-chunk offset
// and checks that phase's bounds, the weight K-step of (chunk, 2x2 tap) is the original tap (2 ky' + py, 2 kx' + px) of the
// ordinary packed matrix.  S2D = 1: forward (patch origin -1, 17 x 17 X' pixels for a 16 x 16 rectangle: every input byte once
// per chunk instead of once per tap).  S2D = 2: the data gradient -- a 2x2 conv from gy to the space-to-depth view of dx,
// 4C output "channels" whose tile columns are scattered to their full-resolution pixels by the epilogue; the rows of the
// transposed packed matrix a tile needs are C-row blocks, one phase per 64-row weight piece.
template <typename T, int BN, int KS, bool FUSE = false, int S2D = 0>
__global__ __launch_bounds__(512, 1) void conv_patch_kernel(const PatchParams p) {
    static_assert(!FUSE || (BN == 128 && KS == 3), "fused modulation: 3x3, 64 gamma + 64 beta columns per tile");
    static_assert(S2D == 0 || (KS == 2 && !FUSE && (S2D == 1 || BN == 128)), "space-to-depth forms: 2x2 taps");
    constexpr int BM = 256, NW = 8, NT = 512;
    constexpr int VEC = Vec<T>::N, BK = 8 * VEC;      // one 128-byte row of K per pixel / weight row
    constexpr int TAPS = KS * KS;
    constexpr int WN = 2, WM = NW / WN, WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PPX = 400;                          // patch capacity in pixels: 6 x 66 (3x3, 4 x 64 rectangle), 11 x 35 (4x4, 8 x 32)
    constexpr int NPIECE = PPX / 8;                   // 1-KiB LDS-DMA pieces (8 pixels x 128 B) per patch
    constexpr int NR = (NPIECE + NW - 1) / NW;        // pieces per thread per patch
    constexpr int NBJ = BN / 64;                      // weight pieces per thread per K-step
    constexpr int NBS = 3;                            // weight stages: K-step kt+2 is in flight while kt is multiplied
    constexpr int PD = NBS - 1;
    constexpr int P_BYTES = PPX * 128, B_BYTES = BN * 128;
    constexpr int EP_ROWS = WTM;                      // epilogue staging: one wave row (64 pixels) per pass
    // The next chunk's patch pieces go out during taps 0 .. TAPS-2 of this one: a piece issued in the LAST tap would still be in
    // flight (the tap's closing wait keeps its own loads outstanding) when the next chunk's first fragments are read.
    constexpr int PPT = (NR + TAPS - 2) / (TAPS - 1); // pieces per tap: 1 (3x3, 4x4), 3 for the 2x2 forms
    static_assert(EP_ROWS * BN * 4 <= P_BYTES, "epilogue staging must fit one patch buffer");
    static_assert(2 * P_BYTES + NBS * B_BYTES <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[2 * P_BYTES + NBS * B_BYTES];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;
    const int TW = p.tw, TH = p.th;
    const int PW = TW + KS - 1, PH = TH + KS - 1;
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wgt = (const T*)p.w;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ resg = (const T*)p.res;
    const T* __restrict__ auxg = (const T*)p.aux;
    const int nch = p.cps, nk = nch * TAPS;           // channel chunks (and K-steps) of ONE work item

    // ---- tiles: the grid is persistent (one workgroup per CU); round k works tiles k*G .. k*G+G-1, handed out so that an
    // XCD's workgroups hold a contiguous range (Cout tiles of one rectangle, then x, then y neighbours share L2 lines)
    struct Tile { int tn, n, oy0, ox0, split; };
    auto decode = [&](int id, int rect = -1) __attribute__((always_inline)) -> Tile {
        Tile q;
        q.split = id / p.tiles_out; id -= q.split * p.tiles_out;
        q.tn = id % p.tiles_n; id /= p.tiles_n;
        // (n, y, x) rectangle index from the dense list; `rect` >= 0: the entry was fetched ahead (one tile earlier)
        if constexpr (FUSE) { if (p.rect_list) id = rect >= 0 ? rect : p.rect_list[id]; }
        q.ox0 = (id % p.tiles_x) * TW; id /= p.tiles_x;
        q.oy0 = (id % p.tiles_y) * TH;
        q.n = id / p.tiles_y;
        return q;
    };
    const int G = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, G);
    int tile_id = slot;
    int n_tiles = p.tiles;
    if constexpr (FUSE) { if (p.rect_count) n_tiles = *p.rect_count * p.tiles_n; }   // dense rectangles x Cout tiles
    if (tile_id >= n_tiles) return;

    // ---- loads of one tile.  Patch piece q = r * NW + wave covers patch pixels 8q .. 8q+7; this lane brings the 16 bytes
    // at physical chunk lane & 7 of pixel 8q + (lane >> 3), i.e. logical chunk (lane & 7) ^ swz(pixel).
    long aoff[NR];                                    // element offset of those 16 B in chunk 0; < 0: zero page
    const T* wrow;                                    // this lane's 16 B of weight row 8 * wave + (lane >> 3), k = 0
    const int brow = 8 * wave + (lane >> 3);
    // the patch pixel a lane fetches for piece r (and its swizzled channel offset) does not depend on the tile: the two
    // divisions per piece are done once per kernel, not once per tile (7 pieces: ~1.5k cycles of every tile's turnaround)
    int ppyx[NR], pcol[NR];                           // (py << 16) | px, py >= PH for pixels past the patch; lc * VEC
    static_for<0, NR>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int pp = 8 * (r * NW + wave) + (lane >> 3);
        const int py = pp / PW, px = pp - py * PW;
        ppyx[r] = (py << 16) | px;
        pcol[r] = ((lane & 7) ^ ((pp >> 1) & 7)) * VEC;
    });
    int aval[NR];                                     // S2D = 1: bit (py << 1 | px) = that phase of the X' pixel lies inside the image
    int chunk0 = 0;                                   // S2D = 1: first chunk of this work item's split
    const T* wrow2[NBJ];                              // S2D = 2: weight row of each 64-row piece (row c of the transposed matrix)
    int wq2[NBJ];                                     //          and the phase (py << 1 | px) its output channels belong to
    auto aim = [&](const Tile& q) __attribute__((always_inline)) {
        const int iy0 = q.oy0 + p.org, ix0 = q.ox0 + p.org;
        if constexpr (S2D == 1) {
            chunk0 = q.split * p.cps;
            static_for<0, NR>([&](auto R) {
                constexpr int r = decltype(R)::value;
                const int py = ppyx[r] >> 16, px = ppyx[r] & 0xffff;
                const int iy = 2 * (iy0 + py), ix = 2 * (ix0 + px);
                const bool in = py < PH;
                const int vy0 = in && (unsigned)iy < (unsigned)p.Hi, vy1 = in && (unsigned)(iy + 1) < (unsigned)p.Hi;
                const int vx0 = (unsigned)ix < (unsigned)p.Wi, vx1 = (unsigned)(ix + 1) < (unsigned)p.Wi;
                aval[r] = (vy0 & vx0) | ((vy0 & vx1) << 1) | ((vy1 & vx0) << 2) | ((vy1 & vx1) << 3);
                aoff[r] = ((long)(q.n * p.Hi + iy) * p.Wi + ix) * p.Cin + pcol[r];
            });
            wrow = wgt + (size_t)(q.tn * BN + brow) * p.Kpad + ((lane & 7) ^ ((brow >> 1) & 7)) * VEC;
            return;
        }
        const long cbase = (long)q.split * p.cps * BK;
        static_for<0, NR>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int py = ppyx[r] >> 16, px = ppyx[r] & 0xffff;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = py < PH && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            aoff[r] = ok ? ((long)(q.n * p.Hi + iy) * p.Wi + ix) * p.Cin + cbase + pcol[r] : -1L;
        });
        if constexpr (S2D == 2) {
            static_for<0, NBJ>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int row = q.tn * BN + 64 * j + brow;                    // output channel of the X' view: (phase, c)
                wq2[j] = row >> p.c_shift;
                wrow2[j] = wgt + (size_t)(row & ((1 << p.c_shift) - 1)) * p.Kpad + q.split * p.cps * BK + ((lane & 7) ^ ((brow >> 1) & 7)) * VEC;
            });
            return;
        }
        // FUSE: tile rows 0..63 = gamma rows 64 tn .. 64 tn + 63 of the packed [gamma | beta] matrix, rows 64..127 = the beta
        // rows of the same channels (mC rows further down): piece j = 1 of dma_w is mC rows away instead of 64
        wrow = wgt + (size_t)(q.tn * (FUSE ? 64 : BN) + brow) * p.Kpad + q.split * p.cps * BK + ((lane & 7) ^ ((brow >> 1) & 7)) * VEC;
    };
    auto dma_patch = [&](auto R, int chunk, int buf) __attribute__((always_inline)) -> int {
        constexpr int r = decltype(R)::value;
        if (r * NW + wave >= NPIECE) return 0;        // wave-uniform
        const void* src;
        if constexpr (S2D == 1) {
            const int k0 = (chunk0 + chunk) * BK, qq = k0 >> p.c_shift, c0 = k0 & (p.Cin - 1);
            const long off = (long)((qq >> 1) * p.Wi + (qq & 1)) * p.Cin + c0;
            src = ((aval[r] >> qq) & 1) ? (const void*)(xg + aoff[r] + off) : (const void*)pz_zero16;
        } else {
            src = aoff[r] >= 0 ? (const void*)(xg + aoff[r] + chunk * BK) : (const void*)pz_zero16;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * P_BYTES + (r * NW + wave) * 1024), 16, 0, 0);
        return 1;
    };
    auto dma_w = [&](int kt, int stage) __attribute__((always_inline)) {     // K-step kt = chunk * TAPS + patch offset
        const int chunk = kt / TAPS, tp = kt - chunk * TAPS;
        if constexpr (S2D == 2) {                     // patch offset (dy, dx) pairs with the 2x2 tap (1 - dy, 1 - dx) of each piece's phase
            const int ky = 2 * (1 - (tp >> 1)), kx = 2 * (1 - (tp & 1));
            static_for<0, NBJ>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int tap_o = (ky + (wq2[j] >> 1)) * 4 + kx + (wq2[j] & 1);
                __builtin_amdgcn_global_load_lds((gptr_t)(const void*)(wrow2[j] + (size_t)tap_o * p.Cin + chunk * BK),
                                                 (lptr_t)(smem + 2 * P_BYTES + stage * B_BYTES + (8 * wave + 64 * j) * 128), 16, 0, 0);
            });
            return;
        }
        const T* src;
        if constexpr (S2D == 1) {
            const int k0 = (chunk0 + chunk) * BK, qq = k0 >> p.c_shift, c0 = k0 & (p.Cin - 1);
            src = wrow + (size_t)((2 * (tp >> 1) + (qq >> 1)) * 4 + 2 * (tp & 1) + (qq & 1)) * p.Cin + c0;
        } else {
            src = wrow + (p.flip ? TAPS - 1 - tp : tp) * p.Cin + chunk * BK;
        }
        static_for<0, NBJ>([&](auto J) {
            constexpr int j = decltype(J)::value;
            __builtin_amdgcn_global_load_lds((gptr_t)(const void*)(src + (size_t)((FUSE ? p.mC : 64) * j) * p.Kpad),
                                             (lptr_t)(smem + 2 * P_BYTES + stage * B_BYTES + (8 * wave + 64 * j) * 128), 16, 0, 0);
        });
    };
    auto prologue = [&](int pbuf) __attribute__((always_inline)) {
        static_for<0, NR>([&](auto R) { dma_patch(R, 0, pbuf); });
        dma_w(0, 0);
        if (PD > 1 && nk > 1) dma_w(1, 1);
    };
    auto wait_keep = [&](int n) __attribute__((always_inline)) {             // all but the n youngest loads have landed
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    };
    static_assert(NBJ + PPT <= 5, "wait_keep covers up to 5 loads per tap");

    // patch pixel of this lane's fragment row at offset (0,0): 32 lanes = 32 consecutive tx of one tile row
    int pp0[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int r = wm * WTM + mi * 32 + l31;
        const int ty = r / TW;                        // (rows r >= TW * TH: no pixel; they read patch pixel 0 and are never stored)
        pp0[mi] = r < TW * TH ? ty * PW + (r - ty * TW) : 0;
    }
    int boff[TN], bq[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = wn * WTN + ni * 32 + l31;
        boff[ni] = row * 128; bq[ni] = (row >> 1) & 7;
    }
    f32x16_t acc[TM][TN];
    // ---- the multiply loop is software-pipelined by hand.  Left to the compiler it reads one fragment set, waits for
    // it (lgkmcnt(0)) and only then issues its MFMAs, so every LDS round trip is exposed, and __syncthreads() carries a
    // vmcnt(0) that would drain the K-step-ahead loads at every barrier.  Here: two fragment sets (the reads of step s+1
    // are issued before the MFMAs of step s wait for theirs), counted waits, a bare s_barrier, and the first fragment set
    // of the NEXT K-step is requested right after the barrier, underneath the last four MFMAs of this one.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    u32x4_t fa[2][TM], fb[2][TN];
    uint32_t a_addr[TM], b_addr[TN];                  // fragment addresses of the current K-step at s = 0; s flips bits 5-6
    auto aim_frags = [&](int dy, int dx, int pbuf, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int pp = pp0[mi] + dy * PW + dx;
            a_addr[mi] = lds0 + pbuf * P_BYTES + pp * 128 + ((h ^ ((pp >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) b_addr[ni] = lds0 + 2 * P_BYTES + stage * B_BYTES + boff[ni] + ((h ^ bq[ni]) << 4);
    };
    auto read_frags = [&](int set, int sstep) __attribute__((always_inline)) {     // logical chunk 2s + h = (h ^ q) ^ 2s
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][mi]) : "v"(a_addr[mi] ^ (uint32_t)(sstep << 5)) : "memory");
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][ni]) : "v"(b_addr[ni] ^ (uint32_t)(sstep << 5)) : "memory");
    };
    // wait until only the NEWEST `TM + TN` LDS reads (or none) are outstanding; the fragment registers are operands so
    // that no MFMA consuming them can be scheduled above the wait
    auto frags_ready = [&](int set, bool all) __attribute__((always_inline)) {
        static_assert(TM == 2 && (TN == 2 || TN == 1), "operand lists below");
        if constexpr (TN == 2) {
            if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
        } else {
            if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]) :: "memory");
        }
    };
    auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) PMfma<T>::run(fa[set][mi], fb[set][ni], acc[mi][ni]);
    };

    constexpr int TPR = BN / VEC, RPP = NT / TPR;
    const int cw = (tid % TPR) * VEC;
    // fp32 accumulators -> LDS (one 64-pixel wave row per pass, staged in the patch buffer `sbuf`) -> 16-B row stores
    // the per-channel bias is the same for every row this thread writes: fetched once per tile (left in the row loop it
    // is re-loaded per row -- the stores may alias it -- and every sweep waits out a global-load round trip), and BEFORE
    // the next tile's DMAs are issued: vector-memory data returns in order, so a bias load queued behind 50 KB of patch
    // pieces is not back before they are (measured: the first sweep waited ~2k cycles for it)
    float bv[VEC];
    auto load_bias = [&](const Tile& q) __attribute__((always_inline)) {
        const int co = q.tn * BN + cw;
#pragma unroll
        for (int j = 0; j < VEC; ++j) bv[j] = (p.bias && co < p.Cout) ? p.bias[co + j] : 0.f;
    };
    // Every LDS access of the epilogues is inline asm and their passes meet at bare s_barriers (round 3): through C++ accesses the
    // compiler puts s_waitcnt vmcnt(0) in front of each (the next tile's LDS-DMA pieces are in flight and "may alias"), and
    // __syncthreads() carries one as well -- every pass then waited out the previous pass's stores and the first one the prologue.
    auto stage_acc = [&](int ep, uint32_t cs0) __attribute__((always_inline)) {
        if (wm == ep) {
            const uint32_t wbase = cs0 + (uint32_t)((4 * h) * BN * 4 + (wn * WTN + l31) * 4);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(wbase), "v"(acc[mi][ni][r]),
                                     "n"((mi * 32 + (r & 3) + 8 * (r >> 2)) * BN * 4 + ni * 32 * 4) : "memory");
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto lds_read4 = [&](uint32_t addr) __attribute__((always_inline)) -> f32x4_t {
        f32x4_t f;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(addr) : "memory");
        return f;
    };
    auto epilogue = [&](const Tile& q, int sbuf) __attribute__((always_inline)) {
        const uint32_t cs0 = lds0 + (uint32_t)(sbuf * P_BYTES);
        const int co = q.tn * BN + cw;
        const bool cok = co < p.Cout;
        constexpr int SWEEPS = EP_ROWS / RPP;
        static_assert(EP_ROWS % RPP == 0, "whole sweeps");
#pragma unroll
        for (int ep = 0; ep < WM; ++ep) {
            // residual / mask operands of this pass are requested before the staging round trip, not after it
            size_t o[SWEEPS]; bool live[SWEEPS];
            u32x4_t rr[SWEEPS], aa[SWEEPS];
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                const int tr = ep * EP_ROWS + sw * RPP + tid / TPR;
                // (a 32-bit division is ~40 VALU instructions; the epilogue's index arithmetic, not its stores, was the largest
                // part of its time in the round-3 ablation: DESIGN 3.1f)
                const int ty = p.tw_shift >= 0 ? tr >> p.tw_shift : tr / TW;
                int oy = q.oy0 + ty, ox = q.ox0 + (tr - ty * TW);
                if constexpr (S2D == 2) {             // column co of the X' view = (phase, c): full-resolution pixel (2Y + py, 2X + px)
                    const int qq = co >> p.c_shift;
                    oy = 2 * oy + (qq >> 1); ox = 2 * ox + (qq & 1);
                    o[sw] = (((size_t)(q.n * p.Ho + oy) * p.Wo + ox) << p.c_shift) + (co & ((1 << p.c_shift) - 1));
                } else {
                    o[sw] = ((size_t)(q.n * p.Ho + oy) * p.Wo + ox) * p.Cout + co;
                }
                live[sw] = cok && tr < TW * TH && oy < p.Ho && ox < p.Wo;
                rr[sw] = u32x4_t{0u, 0u, 0u, 0u}; aa[sw] = rr[sw];
                if (live[sw] && resg && p.splits == 1) rr[sw] = *(const u32x4_t*)(resg + o[sw]);
                if (live[sw] && p.aux_mode != S2E_AUX_NONE && p.splits == 1) aa[sw] = *(const u32x4_t*)(auxg + o[sw]);
            }
            if (ep > 0) lds_barrier();
            stage_acc(ep, cs0);
            lds_barrier();
            if (p.splits > 1) {                       // split-K: raw fp32 partial tile -> this split's slab
                float* slab = p.partial + (size_t)q.split * (S2D == 2 ? ((size_t)p.M << p.c_shift) : (size_t)p.M * p.Cout);   // (S2D = 2: M = N Ho Wo of dx)
#pragma unroll
                for (int sw = 0; sw < SWEEPS; ++sw) {
                    if (!live[sw]) continue;
                    const int row = sw * RPP + tid / TPR;
#pragma unroll
                    for (int j = 0; j < VEC; j += 4)
                        *(f32x4_t*)(slab + o[sw] + j) = lds_read4(cs0 + (uint32_t)((row * BN + cw + j) * 4));
                }
                continue;
            }
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                if (!live[sw]) continue;
                const int row = sw * RPP + tid / TPR;
                float v[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const f32x4_t f = lds_read4(cs0 + (uint32_t)((row * BN + cw + j) * 4));
                    v[j] = f[0] + bv[j]; v[j + 1] = f[1] + bv[j + 1]; v[j + 2] = f[2] + bv[j + 2]; v[j + 3] = f[3] + bv[j + 3];
                }
                if (resg) {
                    float t[VEC];
                    unpack16<T>(rr[sw], t);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] += t[j];
                }
                if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = lrelu02(v[j]);
                } else if (p.out_act == S2E_ACT_TANH) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = tanhf(v[j]);
                }
                if (p.aux_mode != S2E_AUX_NONE) {
                    float t[VEC];
                    unpack16<T>(aa[sw], t);
                    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] *= (t[j] > 0.f ? 1.f : neg);
                }
                *(u32x4_t*)(yg + o[sw]) = pack16<T>(v);
            }
        }
    };

    // ---- FUSE epilogue.  A thread owns 16 bytes of x / out = VEC channels of one pixel: the gamma accumulators of those
    // channels sit in columns cw.. of the staged tile, the beta ones 64 columns further.  Per-channel constants (the two
    // biases, mean, rstd, 1 + s0, s1 + b_beta) are fetched once per tile like the bias above.
    constexpr int FTPR = 64 / VEC, FRPP = NT / FTPR;
    const int fcw = (tid % FTPR) * VEC;
    float k_bg[VEC], k_mu[VEC], k_rs[VEC], k_sa[VEC], k_sb[VEC];
    auto load_mod_consts = [&](const Tile& q) __attribute__((always_inline)) {
        // 16-byte loads (c, msld and mC are multiples of 4 floats; the host checks the base pointers): one-dword loads
        // were 48 vector-memory instructions per thread per tile and cost the tile ~7 % (measured against the plain conv)
        const int c = q.tn * 64 + fcw;
        const f32x4_t* stp = (const f32x4_t*)(p.mstats + ((size_t)q.n * p.mC + c) * 2);
        const f32x4_t* s0p = (const f32x4_t*)(p.mstyle + (size_t)q.n * p.msld + c);
        const f32x4_t* s1p = (const f32x4_t*)(p.mstyle + (size_t)q.n * p.msld + p.mC + c);
        const f32x4_t* bgp = (const f32x4_t*)(p.bias + c);
        const f32x4_t* bbp = (const f32x4_t*)(p.bias + p.mC + c);
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const f32x4_t st0 = stp[j / 2], st1 = stp[j / 2 + 1], s0 = s0p[j / 4], s1 = s1p[j / 4];
            f32x4_t bg = {0.f, 0.f, 0.f, 0.f}, bb = bg;
            if (p.bias) { bg = bgp[j / 4]; bb = bbp[j / 4]; }
            k_mu[j] = st0[0]; k_rs[j] = st0[1]; k_mu[j + 1] = st0[2]; k_rs[j + 1] = st0[3];
            k_mu[j + 2] = st1[0]; k_rs[j + 2] = st1[1]; k_mu[j + 3] = st1[2]; k_rs[j + 3] = st1[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) { k_bg[j + i] = bg[i]; k_sa[j + i] = 1.f + s0[i]; k_sb[j + i] = s1[i] + bb[i]; }
        }
    };
    auto epilogue_fused = [&](const Tile& q, int sbuf) __attribute__((always_inline)) {
        const uint32_t cs0 = lds0 + (uint32_t)(sbuf * P_BYTES);
        const T* __restrict__ mx = (const T*)p.mx;
        T* __restrict__ gout = (T*)p.mgamma;
        const int c = q.tn * 64 + fcw;
        constexpr int SWEEPS = EP_ROWS / FRPP > 0 ? EP_ROWS / FRPP : 1;
        static_assert(EP_ROWS % FRPP == 0 || FRPP % EP_ROWS == 0, "whole sweeps");
#pragma unroll
        for (int ep = 0; ep < WM; ++ep) {
            size_t o[SWEEPS]; bool live[SWEEPS];
            u32x4_t xx[SWEEPS];
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                const int lr = sw * FRPP + tid / FTPR;              // row inside this pass
                const int tr = ep * EP_ROWS + lr;
                const int ty = p.tw_shift >= 0 ? tr >> p.tw_shift : tr / TW;
                const int oy = q.oy0 + ty, ox = q.ox0 + (tr - ty * TW);
                live[sw] = lr < EP_ROWS && tr < TW * TH && oy < p.Ho && ox < p.Wo;
                o[sw] = ((size_t)(q.n * p.Ho + oy) * p.Wo + ox) * p.mC + c;
                const size_t oin = p.mup ? ((size_t)(q.n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.mC + c : o[sw];
                xx[sw] = u32x4_t{0u, 0u, 0u, 0u};
                if (live[sw]) xx[sw] = *(const u32x4_t*)(mx + oin);
            }
            if (ep > 0) lds_barrier();
            stage_acc(ep, cs0);
            lds_barrier();
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                if (!live[sw]) continue;
                const int row = sw * FRPP + tid / FTPR;
                float f[VEC], ga[VEC], v[VEC];
                unpack16<T>(xx[sw], f);
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const f32x4_t g4 = lds_read4(cs0 + (uint32_t)((row * BN + fcw + j) * 4));
                    const f32x4_t b4 = lds_read4(cs0 + (uint32_t)((row * BN + 64 + fcw + j) * 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ga[j + i] = g4[i] + k_bg[j + i];
                        const float xh = (f[j + i] - k_mu[j + i]) * k_rs[j + i];
                        v[j + i] = 0.5f * (xh * (1.f + ga[j + i]) + (b4[i] + k_sb[j + i]) + f[j + i] * k_sa[j + i]);
                    }
                }
                if (p.mlrelu) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = lrelu02(v[j]);
                }
                *(u32x4_t*)(yg + o[sw]) = pack16<T>(v);
                if (gout) *(u32x4_t*)(gout + o[sw]) = pack16<T>(ga);
            }
        }
    };

    Tile cur = decode(tile_id);
    aim(cur);
    int pb = 0;                                       // patch buffer of the current tile's chunk 0
    prologue(pb);
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int next_rect = -1;                               // label-sparse launch: the next tile's rectangle, requested a tile ahead
        if constexpr (FUSE) { if (p.rect_list && tile_id + G < n_tiles) next_rect = p.rect_list[(tile_id + G) / p.tiles_n]; }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        int kt = 0, stage = 0;
        aim_frags(0, 0, pb, 0);
        read_frags(0, 0);
        for (int c = 0; c < nch; ++c) {
            const bool more = c + 1 < nch;
            const int pcur = (pb + c) & 1;
            static_for<0, TAPS>([&](auto TAP) {
                constexpr int tap = decltype(TAP)::value;
                constexpr int ntap = (tap + 1) % TAPS;
                int issued = 0;
                static_for<0, PPT>([&](auto PJ) {
                    constexpr int r = tap + decltype(PJ)::value * (TAPS - 1);
                    if constexpr (tap < TAPS - 1 && r < NR) { if (more) issued += dma_patch(std::integral_constant<int, r>{}, c + 1, pcur ^ 1); }
                });
                if (kt + PD < nk) { dma_w(kt + PD, stage == 0 ? NBS - 1 : stage - 1); issued += NBJ; }
                read_frags(1, 1); frags_ready(0, false); mfmas(0);
                read_frags(0, 2); frags_ready(1, false); mfmas(1);
                read_frags(1, 3); frags_ready(0, false); mfmas(0);
                // K-step kt+1 (and every older patch piece) has landed for this wave once all but this K-step's loads are
                // back; every LDS read of K-step kt is back; then all waves meet
                wait_keep(PD > 1 ? issued : 0);       // (one K-step ahead: what was just issued is needed next)
                frags_ready(1, true);
                __builtin_amdgcn_s_barrier();
                stage = stage == NBS - 1 ? 0 : stage + 1;
                ++kt;
                if (kt < nk) {
                    aim_frags(ntap / KS, ntap % KS, ntap == 0 ? pcur ^ 1 : pcur, stage);
                    read_frags(0, 0);
                }
                mfmas(1);
            });
        }
        // every buffer is free now (the last barrier is behind every LDS read): start the next tile's loads, then write
        // this tile out underneath them
        const int pbn = (pb + nch) & 1;               // continues the alternation; the other one stages the epilogue
        const int next_id = tile_id + G;
        const bool has_next = next_id < n_tiles;
        Tile nxt = cur;
        if constexpr (FUSE) load_mod_consts(cur); else load_bias(cur);
        if (has_next) { nxt = decode(next_id, next_rect); aim(nxt); prologue(pbn); }
        if constexpr (FUSE) epilogue_fused(cur, pbn ^ 1); else epilogue(cur, pbn ^ 1);
        if (!has_next) break;
        cur = nxt; tile_id = next_id; pb = pbn;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------ host side
// Shapes this kernel takes: 3x3 or 4x4, stride 1 (forward or data-gradient), no fused input activation, Cin a multiple of
// the 128-byte K row, Cout a multiple of the 16-byte vector and > 32, a rectangle of output pixels (any width <= 64, as
// many rows as fit 256 pixels and the patch buffer) that keeps >= 80 % of the 256 accumulator rows on image pixels, and
// enough work items to fill the chip: >= S2E_CONV_PATCH (default 224) output tiles, or fewer tiles with a long K that is
// split over channel chunks (>= 2 chunks per split) until >= 192 workgroups exist.  Everything else: conv_igemm.hip.
double s2e_patch_rectangle(const s2e_conv_desc* d, int ks, int* tw_out, int* th_out) {
    const int cap = 400;                             // PPX of the kernel
    double best_fill = 0.0;
    for (int tw = 64; tw >= 8; --tw) {
        int th = 256 / tw;
        while (th > 1 && (th + ks - 1) * (tw + ks - 1) > cap) --th;
        if ((th + ks - 1) * (tw + ks - 1) > cap) continue;
        if (th > d->Ho) th = d->Ho;
        const long tiles = (long)ceil_div(d->Ho, th) * ceil_div(d->Wo, tw);
        const double fill = (double)d->Ho * d->Wo / (256.0 * tiles);
        if (fill > best_fill + 1e-9) { best_fill = fill; *tw_out = tw; *th_out = th; }
    }
    return best_fill;
}

int s2e_conv_patch_plan(int dtype, const s2e_conv_desc* d, s2e_patch_plan* plan) {
    static const int min_tiles = [] { const char* e = getenv("S2E_CONV_PATCH"); return e ? atoi(e) : 224; }();
    static const bool allow_split = [] { const char* e = getenv("S2E_CONV_PATCH_SPLIT"); return e ? atoi(e) != 0 : true; }();
    static const bool allow_k4 = [] { const char* e = getenv("S2E_CONV_PATCH_K4"); return e ? atoi(e) != 0 : true; }();
    s2e_patch_plan local;
    if (!plan) plan = &local;
    plan->tw = plan->th = 0; plan->splits = 1; plan->bn = 0;
    if (min_tiles <= 0) return 0;
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    const int ks = d->KH;
    plan->s2d = 0;
    if (dtype == S2E_BF16 && ks == 4 && d->KW == 4 && d->stride == 2 && d->pad == 2 && d->in_act == S2E_ACT_NONE) {
        // the PatchGAN's 4x4 stride-2 pad-2 layers through the space-to-depth view (S2D, see the kernel)
        static const bool allow_s2d = [] { const char* e = getenv("S2E_CONV_PATCH_S2D"); return e ? atoi(e) != 0 : true; }();
        const int C = d->transposed ? d->Cout : d->Cin;          // channels of the full-resolution tensor
        const int Hf = d->transposed ? d->Ho : d->Hi, Wf = d->transposed ? d->Wo : d->Wi;
        const int Hh = d->transposed ? d->Hi : d->Ho, Wh = d->transposed ? d->Wi : d->Wo;
        const int Cg = d->transposed ? d->Cin : d->Cout;          // channels of the half-resolution tensor
        if (!allow_s2d || C < 64 || (C & (C - 1)) || Hh != Hf / 2 + 1 || Wh != Wf / 2 + 1) return 0;
        if (d->transposed ? (Cg % 64 != 0) : (Cg % vec != 0 || Cg <= 32)) return 0;
        s2e_conv_desc grid = *d;                                 // the rectangles tile the half-resolution grid (of y, or of dx's X' view)
        grid.Ho = d->transposed ? (Hf + 1) / 2 : Hh; grid.Wo = d->transposed ? (Wf + 1) / 2 : Wh;
        if (s2e_patch_rectangle(&grid, 2, &plan->tw, &plan->th) < 0.8) return 0;
        const int ncol = d->transposed ? 4 * C : Cg;
        const long tiles = (long)d->N * ceil_div(grid.Ho, plan->th) * ceil_div(grid.Wo, plan->tw) * ceil_div(ncol, ncol > 64 ? 128 : 64);
        if (tiles < 128) return 0;                               // (a half-filled chip still beats the generic gather: K = 16 C)
        plan->s2d = d->transposed ? 2 : 1;
        return 1;
    }
    if (d->KW != ks || (ks != 3 && !(ks == 4 && allow_k4)) || d->stride != 1 || d->in_act != S2E_ACT_NONE) return 0;
    if (d->Cin % (8 * vec) != 0 || d->Cout % vec != 0 || d->Cout <= 32) return 0;
    const int grow = d->transposed ? (ks - 1) - 2 * d->pad : 2 * d->pad - (ks - 1);
    if (d->Ho != d->Hi + grow || d->Wo != d->Wi + grow) return 0;
    const int bn = d->Cout > 64 ? 128 : 64;
    if (s2e_patch_rectangle(d, ks, &plan->tw, &plan->th) < 0.8) return 0;
    const long tiles = (long)d->N * ceil_div(d->Ho, plan->th) * ceil_div(d->Wo, plan->tw) * ceil_div(d->Cout, bn);
    const int nch = d->Cin / (8 * vec);
    if (tiles >= min_tiles) return 1;
    // long tiles (>= 64 K-steps) are worth a partly filled chip: 160 tiles of the 34^2 PatchGAN data-gradient run 0.21 ms
    // here against 0.31 ms in the generic kernel
    if (tiles >= 160 && tiles * 7 >= min_tiles * 5 && nch * ks * ks >= 64) return 1;
    if (!allow_split) return 0;
    // 64-channel tiles first: twice the tiles, so half the splits fill the chip -- half the fp32 slab traffic of the finish pass
    // (round 5: 512 -> 512 at 32^2 58 -> 50 us, 1024 -> 1024 at 16^2 53 -> 51 us, the other split layers 2-4 %)
    if (bn == 128 && d->Cout % 64 == 0) {
        int b2 = 0;
        for (int s = 1; s <= nch / 2; ++s)
            if (nch % s == 0 && 2 * tiles * s <= 256) b2 = s;
        if (b2 && 2 * tiles * b2 >= 192) { plan->splits = b2; plan->bn = 64; return 1; }
    }
    int best = 0;
    for (int s = 2; s <= nch / 2; ++s)               // a divisor of the chunk count, >= 2 chunks per split, <= ~one workgroup per CU
        if (nch % s == 0 && tiles * s <= 256) best = s;
    if (best && tiles * best >= 192) { plan->splits = best; return 1; }
    return 0;
}

size_t s2e_conv_patch_workspace_bytes(int dtype, const s2e_conv_desc* d) {
    s2e_patch_plan plan;
    if (!s2e_conv_patch_plan(dtype, d, &plan) || plan.splits == 1) return 0;
    return (size_t)plan.splits * d->N * d->Ho * d->Wo * d->Cout * sizeof(float);
}

static int cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <typename T, int BN>
static int launch_patch(const PatchParams& p, int ks, hipStream_t st, int s2d = 0) {
    const int grid = p.tiles < cu_count() ? p.tiles : cu_count();      // persistent: one 127-154 KB workgroup per CU
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (s2d == 1) conv_patch_kernel<T, BN, 2, false, 1><<<grid, 512, 0, st>>>(p);
        else if (s2d == 2) { if constexpr (BN == 128) conv_patch_kernel<T, 128, 2, false, 2><<<grid, 512, 0, st>>>(p); }
    }
    if (s2d) { S2E_CHECK_LAUNCH("conv_patch_kernel (space-to-depth)"); return S2E_OK; }
    if (ks == 3) conv_patch_kernel<T, BN, 3><<<grid, 512, 0, st>>>(p);
    else conv_patch_kernel<T, BN, 4><<<grid, 512, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_patch_kernel");
    return S2E_OK;
}

int s2e_conv_patch_launch(int dtype, const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                          const void* aux, void* y, const s2e_conv_desc* d, int kpad, float* partial, hipStream_t st) {
    PatchParams p{};
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.aux = aux; p.y = y;
    p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.Kpad = kpad;
    p.org = d->transposed ? d->pad - (d->KH - 1) : -d->pad;
    p.flip = d->transposed ? 1 : 0;
    p.out_act = d->out_act; p.aux_mode = d->aux_mode;
    p.tw = plan->tw; p.th = plan->th;
    p.tw_shift = (plan->tw & (plan->tw - 1)) == 0 ? __builtin_ctz(plan->tw) : -1;
    const int bn = plan->bn ? plan->bn : (d->Cout > 64 ? 128 : 64);
    p.tiles_x = ceil_div(d->Wo, p.tw); p.tiles_y = ceil_div(d->Ho, p.th); p.tiles_n = ceil_div(d->Cout, bn);
    p.tiles_out = p.N * p.tiles_y * p.tiles_x * p.tiles_n;
    p.splits = plan->splits; p.cps = d->Cin / (dtype == S2E_BF16 ? 64 : 32) / plan->splits;
    p.tiles = p.tiles_out * plan->splits;
    p.M = d->N * d->Ho * d->Wo; p.partial = partial;
    if (plan->s2d) {
        if (dtype != S2E_BF16 || plan->splits != 1) S2E_FAIL(S2E_ERR_UNSUPPORTED, "conv_patch: the space-to-depth forms are bf16, unsplit");
        const int C = plan->s2d == 2 ? d->Cout : d->Cin;
        p.c_shift = __builtin_ctz(C);
        if (plan->s2d == 1) {                         // forward: 2x2 taps over X' (4C channels), patch origin one X' pixel up / left
            p.org = -1; p.flip = 0;
            p.cps = 4 * C / 64;
            return bn == 128 ? launch_patch<bf16_t, 128>(p, 2, st, 1) : launch_patch<bf16_t, 64>(p, 2, st, 1);
        }
        // data gradient: rectangles over the X' grid of dx, 4C columns; the patch (gy) starts at the rectangle's origin
        p.org = 0; p.flip = 1;
        p.Cout = 4 * C;
        p.tiles_x = ceil_div((d->Wo + 1) / 2, p.tw); p.tiles_y = ceil_div((d->Ho + 1) / 2, p.th); p.tiles_n = 4 * C / 128;
        p.tiles_out = p.N * p.tiles_y * p.tiles_x * p.tiles_n; p.tiles = p.tiles_out;
        return launch_patch<bf16_t, 128>(p, 2, st, 2);
    }
    if (dtype == S2E_BF16) return bn == 128 ? launch_patch<bf16_t, 128>(p, d->KH, st) : launch_patch<bf16_t, 64>(p, d->KH, st);
    if (dtype == S2E_F32) return bn == 128 ? launch_patch<float, 128>(p, d->KH, st) : launch_patch<float, 64>(p, d->KH, st);
    S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: bad dtype %d", dtype);
}

// ------------------------------------------------------------------------------------ fused [gamma | beta] conv + modulation
// The shapes the fused launch takes: the 3x3 pad-1 conv nh -> 2C with C a multiple of 64 (a tile pairs 64 gamma with 64 beta
// columns) and nh a multiple of the 128-byte K row; never split over channel chunks (the modulation needs the finished sums).
// It takes smaller maps than s2e_conv_patch_plan does: measured at batch 8 (tools/bench_spade_fused.py), one launch of 128
// workgroups beats conv + modulation as two launches even at 16x16 (33 -> 23 us) and at 8x8, where three quarters of a tile's
// rows are idle (31 -> 22 us).  Below S2E_SPADE_FUSED_TILES (default 96) tiles the two-launch path runs; flags & 1 forces the
// fused kernel at any tile count (tests).
static int fused_plan(int dtype, int N, int H, int W, int C, int nh, int flags, s2e_conv_desc* d, s2e_patch_plan* plan) {
    static const int min_tiles = [] { const char* e = getenv("S2E_SPADE_FUSED_TILES"); return e ? atoi(e) : 96; }();
    if (min_tiles <= 0 && !(flags & 1)) return 0;
    if (C <= 0 || C % 64 != 0 || N <= 0 || H <= 0 || W <= 0) return 0;
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (nh % (8 * vec) != 0) return 0;
    *d = s2e_conv_desc{N, H, W, nh, H, W, 2 * C, 3, 3, 1, 1, 0, S2E_ACT_NONE, S2E_ACT_NONE, S2E_AUX_NONE};
    plan->splits = 1; plan->tw = plan->th = 0;
    const double fill = s2e_patch_rectangle(d, 3, &plan->tw, &plan->th);
    if (fill < ((flags & 1) ? 0.01 : 0.2)) return 0;
    // among the rectangles that fill as well, the SQUAREST one (16 x 16 before 8 x 32 before 4 x 64): the smallest patch with its
    // halo (324 vs 340 vs 396 pixels of DMA per chunk) and -- what matters for the label-sparse launch -- the shape most
    // likely to lie inside one label region (bench maps at 256^2: 72 % / 70 % / 58 % of the rectangles are label-uniform)
    for (int tw = 16; tw < plan->tw; tw *= 2) {
        int th = 256 / tw;
        if (th > H) th = H;
        if ((th + 2) * (tw + 2) > 400) continue;
        const double f = (double)H * W / (256.0 * ceil_div(H, th) * ceil_div(W, tw));
        if (f >= fill - 1e-9) { plan->tw = tw; plan->th = th; break; }
    }
    const long tiles = (long)N * ceil_div(H, plan->th) * ceil_div(W, plan->tw) * (C / 64);
    return (flags & 1) || tiles >= min_tiles;
}

extern "C" int s2e_spade_conv_modulate_supported(int dtype, int N, int H, int W, int C, int nh, int flags) {
    s2e_conv_desc d; s2e_patch_plan plan;
    if (dtype != S2E_BF16 && dtype != S2E_F32) return 0;
    return fused_plan(dtype, N, H, W, C, nh, flags, &d, &plan);
}

extern "C" int s2e_spade_conv_modulate_rect(int dtype, int N, int H, int W, int C, int nh, int flags, int* tw, int* th) {
    s2e_conv_desc d; s2e_patch_plan plan;
    if ((dtype != S2E_BF16 && dtype != S2E_F32) || !fused_plan(dtype, N, H, W, C, nh, flags, &d, &plan)) return 0;
    if (tw) *tw = plan.tw;
    if (th) *th = plan.th;
    return 1;
}

static int spade_conv_modulate_impl(int dtype, const void* actv, const void* w_packed, const float* bias, const void* x,
                                    const float* stats, const float* style, int style_ld, void* out, void* gamma_out,
                                    int N, int H, int W, int C, int nh, int lrelu, int flags, const int* rect_list, const int* rect_count,
                                    void* stream) {
    if (!actv || !w_packed || !x || !stats || !style || !out) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_conv_modulate: null pointer");
    if ((rect_list == nullptr) != (rect_count == nullptr)) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_conv_modulate_sparse: rect_list and rect_count go together");
    if (((uintptr_t)stats | (uintptr_t)style | (uintptr_t)bias) & 15 || (style_ld & 3))
        S2E_FAIL(S2E_ERR_ARG, "s2e_spade_conv_modulate: stats, style and bias must be 16-byte aligned (style_ld a multiple of 4)");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_conv_modulate: bad dtype %d", dtype);
    s2e_conv_desc d; s2e_patch_plan plan;
    if (!fused_plan(dtype, N, H, W, C, nh, flags, &d, &plan))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_spade_conv_modulate: shape N=%d %dx%d C=%d nh=%d is not taken by the fused kernel "
                 "(s2e_spade_conv_modulate_supported); run s2e_conv2d + s2e_modulate_fwd", N, H, W, C, nh);
    if (const int rc = s2e_spade_conv_modulate_duo(dtype, actv, w_packed, bias, x, stats, style, style_ld, out, gamma_out, N, H, W, C, nh,
                                                      lrelu, flags, plan.tw, plan.th, rect_list, rect_count, (hipStream_t)stream))
        return rc < 0 ? rc : S2E_OK;                  // (the duo kernel took it: same rectangles, same lists)
    PatchParams p{};
    p.x = actv; p.w = w_packed; p.bias = bias; p.y = out;
    p.N = N; p.Hi = H; p.Wi = W; p.Cin = nh; p.Ho = H; p.Wo = W; p.Cout = 2 * C;
    p.Kpad = ceil_div(9 * nh, dtype == S2E_BF16 ? 64 : 32) * (dtype == S2E_BF16 ? 64 : 32);
    p.org = -1; p.flip = 0; p.out_act = S2E_ACT_NONE; p.aux_mode = S2E_AUX_NONE;
    p.tw = plan.tw; p.th = plan.th;
    p.tw_shift = (plan.tw & (plan.tw - 1)) == 0 ? __builtin_ctz(plan.tw) : -1;
    p.tiles_x = ceil_div(W, p.tw); p.tiles_y = ceil_div(H, p.th); p.tiles_n = C / 64;
    p.tiles_out = N * p.tiles_y * p.tiles_x * p.tiles_n;
    p.splits = 1; p.cps = nh / (dtype == S2E_BF16 ? 64 : 32); p.tiles = p.tiles_out;
    p.M = N * H * W; p.partial = nullptr;
    p.mx = x; p.mstats = stats; p.mstyle = style; p.msld = style_ld > 0 ? style_ld : 2 * C; p.mgamma = gamma_out;
    p.mup = (flags & 8) != 0;
    if (p.mup && ((H | W) & 1)) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_conv_modulate: flags 8 (x at half resolution) needs even H, W");
    p.mC = C; p.mlrelu = lrelu;
    p.rect_list = rect_list; p.rect_count = rect_count;
    const int grid = p.tiles < cu_count() ? p.tiles : cu_count();
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) conv_patch_kernel<bf16_t, 128, 3, true><<<grid, 512, 0, st>>>(p);
    else conv_patch_kernel<float, 128, 3, true><<<grid, 512, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_patch_kernel (fused modulation)");
    return S2E_OK;
}

extern "C" int s2e_spade_conv_modulate(int dtype, const void* actv, const void* w_packed, const float* bias, const void* x,
                                       const float* stats, const float* style, int style_ld, void* out, void* gamma_out,
                                       int N, int H, int W, int C, int nh, int lrelu, int flags, void* stream) {
    return spade_conv_modulate_impl(dtype, actv, w_packed, bias, x, stats, style, style_ld, out, gamma_out, N, H, W, C, nh, lrelu, flags,
                                    nullptr, nullptr, stream);
}

extern "C" int s2e_spade_conv_modulate_sparse(int dtype, const void* actv, const void* w_packed, const float* bias, const void* x,
                                              const float* stats, const float* style, int style_ld, void* out, void* gamma_out,
                                              int N, int H, int W, int C, int nh, int lrelu, int flags, const int* rect_list,
                                              const int* rect_count, void* stream) {
    if (!rect_list || !rect_count) S2E_FAIL(S2E_ERR_ARG, "s2e_spade_conv_modulate_sparse: rect_list / rect_count missing");
    return spade_conv_modulate_impl(dtype, actv, w_packed, bias, x, stats, style, style_ld, out, gamma_out, N, H, W, C, nh, lrelu, flags,
                                    rect_list, rect_count, stream);
}
