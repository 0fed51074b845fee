// Implicit-GEMM convolution for gfx950: forward conv and data-gradient (transposed gather) in one
// kernel, NHWC activations, MFMA 32x32 tiles, fp32 accumulate.
//
//   GEMM view:  C[m][co] = sum_k A[m][k] * B[co][k]
//     m  = output pixel (n, oy, ox)            M = N*Ho*Wo
//     k  = (tap, ci) = (ky*KW + kx)*Cin + ci   K = KH*KW*Cin   (im2col gathered on the fly, never stored)
//     B  = packed weights, row co, K contiguous (s2e_pack_conv_weight)
//
//   Workgroup = 256 threads = 4 waves (one per SIMD), tile BM=128 pixels x BN in {128,64,32} channels,
//   K-step = one 128-byte row per tile row (64 bf16 / 32 f32).  Register-staged double buffering:
//   global loads of K-tile t+1 are issued before the MFMAs of tile t and written to the other LDS
//   buffer after them; one barrier per K-tile.  LDS rows are 128 B with the 16-B chunk index XORed by
//   (row>>1)&7 so ds_read_b128 of 32 consecutive rows at one logical chunk is conflict-free.
//   Epilogue: accumulators -> LDS (fp32 [BM][BN]) -> coalesced 16-B row stores with bias / residual /
//   activation / mask fused.
//
//   HBM layout: activations NHWC so a tap's Cin run is contiguous (one 16-B load = 8 bf16 channels);
//   the 1-D grid is remapped so tiles sharing an A panel (same pixel rows) run on one XCD's L2.
#include "common.h"
#include "conv_small.h"
#include "conv_patch.h"
#include "conv_stream.h"
#include <stdlib.h>

struct ConvKParams {
    const void* x; const void* w; const float* bias; const void* res; const void* aux; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad, transposed;
    int in_act, out_act, aux_mode;
    int Ktot, Kpad, M, tiles_n;
    int tm_fast;                          // tile order: consecutive tile ids walk the pixel tiles of ONE weight panel (see s2e_conv2d)
    int tiles, splits, kt_per_split;      // split-K: grid = tiles * splits, split s owns K-tiles [s*per, (s+1)*per)
    float* partial;                       // splits > 1: fp32 slabs [splits][M][Cout], combined by conv_finish_kernel
    int cls_tile0[5];                     // S2 kernels: first pixel-tile of each output-parity class (prefix sums)
};

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
    static __device__ __forceinline__ void run(u32x4_t a, u32x4_t b, f32x16_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    // lane half h holds 4 consecutive k; MFMA t pairs element t of half 0 with element t of half 1.
    // A and B use the same (permuted) k order, so the contraction is exact.
    // (vectors by value + whole-vector bit_cast: see the note at unpack16 in common.h)
    static __device__ __forceinline__ void run(u32x4_t a, u32x4_t b, f32x16_t& acc) {
        const f32x4_t fa = __builtin_bit_cast(f32x4_t, a), fb = __builtin_bit_cast(f32x4_t, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
    }
};

template <typename T>
__device__ __forceinline__ u32x4_t apply_lrelu16(u32x4_t r) {
    float f[Vec<T>::N];
    unpack16<T>(r, f);
#pragma unroll
    for (int j = 0; j < Vec<T>::N; ++j) f[j] = lrelu02(f[j]);
    return pack16<T>(f);
}

// VECPATH: Cin is a multiple of the 16-B vector width, so every im2col chunk is one aligned 16-B load.
// The any-Cin element-wise gather (tiny-K layers only: Cin = 1 or 5) is a separate instantiation so
// its index arithmetic never bloats the hot kernel.
// GLDS: stage both operands with LDS-DMA (global_load_lds, 16 B per lane) instead of global_load +
// ds_write_b128.  A ds_write_b128 costs ~13 LDS cycles per wave-instruction; with 8 of them per wave per
// K-tile the register-staged kernel is LDS-write-bound (64 writes + 128 reads ~ 1340 LDS cycles per CU per
// K-tile pair vs 1024 MFMA cycles).  LDS-DMA writes lane-linear (M0 base + lane*16), so one instruction
// fills 8 consecutive 128-B rows; the XOR swizzle moves to the SOURCE side (lane at physical chunk c loads
// logical chunk c ^ swz(row)), and lanes whose tap reads padding fetch from a 16-byte zero page.
// GLDS needs VECPATH and no input activation (nothing passes through registers).
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

// S2: stride-2 DATA-GRADIENT by output-parity class.  In a stride-2 transposed conv an output pixel (oy, ox) only sees
// the taps with ky = oy + pad (mod 2), kx = ox + pad (mod 2): a quarter of a 4x4 kernel.  Walking all taps with a
// validity mask (the generic path) spends 3 of 4 MFMAs on structural zeros (measured 130-165 TFLOP/s).  Here the
// pixel tiles are formed PER CLASS (oy & 1, ox & 1) -- cls_tile0[] holds the first tile of each class -- so a whole
// tile shares its tap subset and the K loop walks only those taps (both operands: the weight matrix is addressed by
// the loader's (ky, kx, ci) state instead of linearly).  Needs VECPATH + GLDS, no split-K.
template <typename T, int BN, bool VECPATH, bool GLDS, bool S2 = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvKParams p) {
    constexpr int BM = 128;
    constexpr int VEC = Vec<T>::N;
    constexpr int BK = 8 * VEC;
    constexpr int WN = (BN >= 64) ? 2 : 1;
    constexpr int WM = 4 / WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NB = BN / 32;                       // B vectors per thread per K-tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    static_assert(2 * STAGE >= BM * BN * 4, "epilogue staging must fit");
    static_assert(!S2 || (VECPATH && GLDS), "the parity-class data-gradient exists for the vector LDS-DMA loader only");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wg / p.tiles, tile = wg - split * p.tiles;
    const int tiles_m = p.tiles / p.tiles_n;
    const int tn = p.tm_fast ? tile / tiles_m : tile % p.tiles_n;
    const int tm = p.tm_fast ? tile - tn * tiles_m : tile / p.tiles_n;
    // S2: this tile's output-parity class and its tap subset
    int s2_tml = 0, s2_qy = 0, s2_qx = 0, s2_Hq = 1, s2_Wq = 1, s2_Mq = 0, ky0 = 0, kx0 = 0, nky = p.KH, nkx = p.KW;
    if constexpr (S2) {
        int cls = 0;
        while (cls < 3 && tm >= p.cls_tile0[cls + 1]) ++cls;
        s2_qy = cls >> 1; s2_qx = cls & 1;
        s2_Hq = (p.Ho - s2_qy + 1) >> 1; s2_Wq = (p.Wo - s2_qx + 1) >> 1;
        s2_Mq = p.N * s2_Hq * s2_Wq;
        s2_tml = tm - p.cls_tile0[cls];
        ky0 = (s2_qy + p.pad) & 1; kx0 = (s2_qx + p.pad) & 1;
        nky = (p.KH - ky0 + 1) >> 1; nkx = (p.KW - kx0 + 1) >> 1;
    }
    const int nk_all = S2 ? (nky * nkx * p.Cin + BK - 1) / BK : p.Kpad / BK;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = S2 ? nk_all : min(nk_all, kt0 + p.kt_per_split);

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wgt = (const T*)p.w;
    // register staging: thread -> rows r0 + 32*i, physical chunk tid&7 (swizzled on the LDS write)
    // LDS-DMA        : wave w, instruction i fills rows 8*(w + 4*i) .. +7; lane -> row offset lane>>3,
    //                  physical chunk lane&7, i.e. LOGICAL chunk (lane&7) ^ swz(row).  swz(row) =
    //                  (row>>1)&7 = (4*(w&1) + (lane>>4)) & 7 for every i, so one (tap, ci) state per thread.
    const int r0 = GLDS ? 8 * wave + (lane >> 3) : tid >> 3;
    const int c = GLDS ? ((lane & 7) ^ ((r0 >> 1) & 7)) : (tid & 7);
    const int swz_st = (c ^ ((r0 >> 1) & 7)) << 4;     // (r0+32i)>>1 & 7 == (r0>>1)&7

    int by[4], bx[4], nb[4];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = 0, oy = 0, ox = 0;
        bool live;
        if constexpr (S2) {
            const int ml = s2_tml * BM + r0 + 32 * i;
            live = ml < s2_Mq;
            if (live) {
                const int hw = s2_Hq * s2_Wq;
                n = ml / hw;
                const int rem = ml - n * hw, y2 = rem / s2_Wq;
                oy = 2 * y2 + s2_qy; ox = 2 * (rem - y2 * s2_Wq) + s2_qx;
            }
        } else {
            const int m = tm * BM + r0 + 32 * i;
            live = m < p.M;
            if (live) {
                n = m / HoWo;
                const int rem = m - n * HoWo;
                oy = rem / p.Wo; ox = rem - oy * p.Wo;
            }
        }
        if (live) {
            by[i] = p.transposed ? oy + p.pad : oy * p.stride - p.pad;
            bx[i] = p.transposed ? ox + p.pad : ox * p.stride - p.pad;
            nb[i] = n * p.Hi * p.Wi;
        } else {
            by[i] = -(1 << 24); bx[i] = -(1 << 24); nb[i] = 0;   // every tap out of range
        }
    }

    // source pixel of tap (ky,kx) for row i; returns false when the tap reads padding
    auto src_pixel = [&](int i, int ky, int kx, int& pix) __attribute__((always_inline)) -> bool {
        int iy, ix; bool v;
        if (!p.transposed) {
            iy = by[i] + ky; ix = bx[i] + kx;
            v = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        } else {
            const int ty = by[i] - ky, tx = bx[i] - kx;
            v = (ty | tx) >= 0;
            if (p.stride == 2) { v = v && (((ty | tx) & 1) == 0); iy = ty >> 1; ix = tx >> 1; }
            else { iy = ty; ix = tx; }
            v = v && iy < p.Hi && ix < p.Wi;
        }
        pix = nb[i] + iy * p.Wi + ix;
        return v;
    };

    // Fast addressing for the vector paths: in every mode the source pixel of tap (ky,kx) for row i is
    //   rowbase[i] + tapoff(ky,kx)   with a ROW-INDEPENDENT tap offset
    //     forward        : rowbase = nb + by*Wi + bx,            tapoff = ky*Wi + kx
    //     dgrad stride 1 : rowbase = nb + by*Wi + bx,            tapoff = -(ky*Wi + kx)
    //     dgrad stride 2 : rowbase = nb + (by>>1)*Wi + (bx>>1),  tapoff = -((ky>>1)*Wi + (kx>>1))
    //       (on a valid tap by-ky is even, and then (by-ky)/2 == (by>>1) - (ky>>1))
    // and validity is one bit per tap, computed once per row (KH*KW <= 32).
    long rowoff[4];                                   // rowbase * Cin  (element offset; may be negative)
    unsigned vmask[4];
    if constexpr (VECPATH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // validity is separable: tap (ky,kx) is valid iff ky is valid for this row's y AND kx for its x
            auto ok1 = [&](int b, int k, int lim) __attribute__((always_inline)) -> bool {
                if (!p.transposed) return (unsigned)(b + k) < (unsigned)lim;
                const int t = b - k;
                if (t < 0) return false;
                if (p.stride == 2) return ((t & 1) == 0) && (t >> 1) < lim;
                return t < lim;
            };
            unsigned xm = 0;
            for (int kx = 0; kx < p.KW; ++kx) xm |= ok1(bx[i], kx, p.Wi) ? (1u << kx) : 0u;
            unsigned m = 0;
            for (int ky = 0; ky < p.KH; ++ky) m |= ok1(by[i], ky, p.Hi) ? (xm << (ky * p.KW)) : 0u;
            vmask[i] = m;
            const int rb = (p.transposed && p.stride == 2) ? nb[i] + (by[i] >> 1) * p.Wi + (bx[i] >> 1)
                                                           : nb[i] + by[i] * p.Wi + bx[i];
            rowoff[i] = (long)rb * p.Cin;
        }
    }
    const int tap_sy = p.transposed ? -p.Wi : p.Wi, tap_sx = p.transposed ? -1 : 1;
    const int tap_sh = (p.transposed && p.stride == 2) ? 1 : 0;
    // element offset of tap (ky,kx), channel ci relative to a row's base
    auto tap_elem_off = [&](int ky, int kx, int ci) __attribute__((always_inline)) -> long {
        return (long)((ky >> tap_sh) * tap_sy + (kx >> tap_sh) * tap_sx) * p.Cin + ci;
    };

    // loader state of the vector path: (ky, kx, ci) of this thread's 16-B chunk in the NEXT K-tile,
    // advanced incrementally (no per-tile integer division)
    int l_ky, l_kx, l_ci;
    {
        const int k0 = kt0 * BK + c * VEC, tap = k0 / p.Cin;
        l_ci = k0 - tap * p.Cin;
        if constexpr (S2) {                              // tap = index into the class's (nky x nkx) tap subset
            const int jy = tap / nkx;
            l_ky = ky0 + 2 * jy; l_kx = kx0 + 2 * (tap - jy * nkx);
        } else {
            l_ky = tap / p.KW; l_kx = tap - l_ky * p.KW;
        }
    }
    auto advance_k = [&]() __attribute__((always_inline)) {
        l_ci += BK;
        if constexpr (S2) {
            while (l_ci >= p.Cin) { l_ci -= p.Cin; l_kx += 2; if (l_kx >= p.KW) { l_kx = kx0; l_ky += 2; } }
        } else {
            while (l_ci >= p.Cin) { l_ci -= p.Cin; if (++l_kx == p.KW) { l_kx = 0; ++l_ky; } }
        }
    };
    struct Stage { u32x4_t a[4]; u32x4_t b[NB]; };
    auto load_tile = [&](int kt, Stage& S) __attribute__((always_inline)) {
        u32x4_t (&ra)[4] = S.a;
        u32x4_t (&rb)[NB] = S.b;
        const int k0 = kt * BK + c * VEC;
        if constexpr (VECPATH) {
            const bool kvalid = l_ky < p.KH;             // <=> k0 < Ktot
            const int tbit = kvalid ? l_ky * p.KW + l_kx : 0;
            const long koff = tap_elem_off(l_ky, l_kx, l_ci);
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const bool v = kvalid && ((vmask[i] >> tbit) & 1u);
                ra[i] = u32x4_t{0, 0, 0, 0};
                if (v) ra[i] = *(const u32x4_t*)(xg + (rowoff[i] + koff));
            });
            advance_k();
        } else {                                         // any-Cin fallback: element-wise gather
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                float f[VEC];
                static_for<0, VEC>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const int k = k0 + j;
                    f[j] = 0.f;
                    if (k < p.Ktot) {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin;
                        const int ky = tap / p.KW, kx = tap - ky * p.KW;
                        int pix;
                        if (src_pixel(i, ky, kx, pix)) f[j] = load1<T>(xg + (size_t)pix * p.Cin + ci);
                    }
                });
                ra[i] = pack16<T>(f);
            });
        }
        if (p.in_act == S2E_ACT_LRELU)
            static_for<0, 4>([&](auto I) { ra[decltype(I)::value] = apply_lrelu16<T>(ra[decltype(I)::value]); });
        static_for<0, NB>([&](auto J) {
            constexpr int j = decltype(J)::value;
            rb[j] = *(const u32x4_t*)(wgt + (size_t)(tn * BN + r0 + 32 * j) * p.Kpad + kt * BK + c * VEC);
        });
    };
    auto store_tile = [&](int buf, const Stage& S) __attribute__((always_inline)) {
        const u32x4_t (&ra)[4] = S.a;
        const u32x4_t (&rb)[NB] = S.b;
        char* base = smem + buf * STAGE;
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            *(u32x4_t*)(base + (r0 + 32 * i) * 128 + swz_st) = ra[i];
        });
        static_for<0, NB>([&](auto J) {
            constexpr int j = decltype(J)::value;
            *(u32x4_t*)(base + A_BYTES + (r0 + 32 * j) * 128 + swz_st) = rb[j];
        });
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int h = lane >> 5, l31 = lane & 31;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* As = smem + buf * STAGE;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int chunk = 2 * s + h;
            u32x4_t a[TM], b[TN];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int row = wm * WTM + mi * 32 + l31;
                a[mi] = *(const u32x4_t*)(As + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int row = wn * WTN + ni * 32 + l31;
                b[ni] = *(const u32x4_t*)(Bs + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) Mfma<T>::run(a[mi], b[ni], acc[mi][ni]);
        }
    };

    const int nk = kt1 - kt0;                       // >= 1 by construction of kt_per_split
    if constexpr (!GLDS) {
        Stage s0;
        load_tile(kt0, s0);
        store_tile(0, s0);
        __syncthreads();
        for (int kt = 0; kt + 1 < nk; ++kt) {       // steady state: prefetch kt+1 around the MFMAs of kt
            const int cur = kt & 1;
            load_tile(kt0 + kt + 1, s0);
            compute(cur);
            store_tile(cur ^ 1, s0);
            __syncthreads();
        }
        compute((nk - 1) & 1);
        __syncthreads();
    } else {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        // rows r0 + 32*i of this thread land at LDS row 8*(wave + 4*i) + (lane>>3) = r0 + 32*i: same rows
        auto dma_tile = [&](int kt, int buf) __attribute__((always_inline)) {
            char* base = smem + buf * STAGE;
            const bool kvalid = l_ky < p.KH;
            const int tbit = kvalid ? l_ky * p.KW + l_kx : 0;
            const long koff = tap_elem_off(l_ky, l_kx, l_ci);
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const bool v = kvalid && ((vmask[i] >> tbit) & 1u);
                const void* src = v ? (const void*)(xg + (rowoff[i] + koff)) : (const void*)g_zero16;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (8 * wave + 32 * i) * 128), 16, 0, 0);
            });
            // weight column of this thread's chunk: linear in k, except in S2 mode where k walks the class's tap subset
            const long kb = S2 ? (kvalid ? (long)(l_ky * p.KW + l_kx) * p.Cin + l_ci : -1L) : (long)kt * BK + c * VEC;
            advance_k();
            static_for<0, NB>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const void* src = (S2 && kb < 0) ? (const void*)g_zero16
                                                 : (const void*)(wgt + (size_t)(tn * BN + r0 + 32 * j) * p.Kpad + kb);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + A_BYTES + (8 * wave + 32 * j) * 128), 16, 0, 0);
            });
        };
        dma_tile(kt0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int cur = kt & 1;
            dma_tile(kt0 + kt + 1, cur ^ 1);        // DMA into the other buffer while this one is read
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        compute((nk - 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: acc -> LDS fp32 [BM][BN] -> coalesced rows
    float* Cs = (float*)smem;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = wn * WTN + ni * 32 + l31;
                Cs[row * BN + col] = acc[mi][ni][r];
            }
    __syncthreads();

    constexpr int TPR = BN / VEC;                      // threads per output row
    constexpr int RPP = 256 / TPR;                     // rows per pass
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ resg = (const T*)p.res;
    const T* __restrict__ auxg = (const T*)p.aux;
    const int cw = (tid % TPR) * VEC;
    const int co = tn * BN + cw;
    const bool cvec = (p.Cout % VEC) == 0;
    if (p.splits > 1) {                              // split-K: raw fp32 partial tile -> this split's slab
        float* slab = p.partial + (size_t)split * p.M * p.Cout;
        constexpr int TPR4 = BN / 4, RPP4 = 256 / TPR4;
        const int cw4 = (tid % TPR4) * 4, co4 = tn * BN + cw4;
        for (int row = tid / TPR4; row < BM; row += RPP4) {
            const int m = tm * BM + row;
            if (m >= p.M || co4 >= p.Cout) continue;
            const f32x4_t t = *(const f32x4_t*)(Cs + row * BN + cw4);
            float* dst = slab + (size_t)m * p.Cout + co4;
            if ((p.Cout & 3) == 0) *(f32x4_t*)dst = t;
            else for (int j = 0; j < 4 && co4 + j < p.Cout; ++j) dst[j] = t[j];
        }
        return;
    }
    // the per-channel bias is the same for every row of this thread: fetch it once (inside the row loop it is re-loaded
    // per row, because the stores may alias it, and every iteration then waits out a global-load round trip)
    float bv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bv[j] = (p.bias && cvec && co < p.Cout) ? p.bias[co + j] : 0.f;
    for (int row = tid / TPR; row < BM; row += RPP) {
        int m = tm * BM + row;
        if constexpr (S2) {                              // class-local row -> output pixel
            const int ml = s2_tml * BM + row;
            if (ml >= s2_Mq) continue;
            const int hw = s2_Hq * s2_Wq, n = ml / hw, rem = ml - n * hw, y2 = rem / s2_Wq;
            m = (n * p.Ho + 2 * y2 + s2_qy) * p.Wo + 2 * (rem - y2 * s2_Wq) + s2_qx;
        }
        if (m >= p.M || co >= p.Cout) continue;
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const f32x4_t t = *(const f32x4_t*)(Cs + row * BN + cw + j);
            v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
        }
        const size_t o = (size_t)m * p.Cout + co;
        if (cvec) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] += bv[j];
            if (resg) {
                float rr[VEC];
                unpack16<T>(*(const u32x4_t*)(resg + o), rr);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[j] += rr[j];
            }
            if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[j] = lrelu02(v[j]);
            } else if (p.out_act == S2E_ACT_TANH) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[j] = tanhf(v[j]);
            }
            if (p.aux_mode != S2E_AUX_NONE) {
                float aa[VEC];
                unpack16<T>(*(const u32x4_t*)(auxg + o), aa);
                const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[j] *= (aa[j] > 0.f ? 1.f : neg);
            }
            *(u32x4_t*)(yg + o) = pack16<T>(v);
        } else {
            for (int j = 0; j < VEC && co + j < p.Cout; ++j) {
                float t = v[j];
                if (p.bias) t += p.bias[co + j];
                if (resg) t += load1<T>(resg + o + j);
                if (p.out_act == S2E_ACT_LRELU) t = lrelu02(t);
                else if (p.out_act == S2E_ACT_TANH) t = tanhf(t);
                if (p.aux_mode != S2E_AUX_NONE) {
                    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
                    t *= (load1<T>(auxg + o + j) > 0.f ? 1.f : neg);
                }
                store1<T>(yg + o + j, t);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ split-K finish
// y = out_act(sum_s slab[s] + bias + residual) * aux-mask, one thread per 16-byte output vector.
template <typename T>
__global__ __launch_bounds__(256) void conv_finish_kernel(const ConvKParams p) {
    constexpr int VEC = Vec<T>::N;
    const long total = (long)p.M * p.Cout;
    const bool cvec = (p.Cout % VEC) == 0;
    const long nvec = cvec ? total / VEC : total;
    const size_t slab = (size_t)p.M * p.Cout;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ resg = (const T*)p.res;
    const T* __restrict__ auxg = (const T*)p.aux;
    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        if (cvec) {
            const size_t o = (size_t)v * VEC;
            const int co = (int)(o % p.Cout);
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = p.bias ? p.bias[co + j] : 0.f;
            for (int s = 0; s < p.splits; ++s) {
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const f32x4_t t = *(const f32x4_t*)(p.partial + s * slab + o + j);
                    acc[j] += t[0]; acc[j + 1] += t[1]; acc[j + 2] += t[2]; acc[j + 3] += t[3];
                }
            }
            if (resg) {
                float rr[VEC];
                unpack16<T>(*(const u32x4_t*)(resg + o), rr);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += rr[j];
            }
            if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = lrelu02(acc[j]);
            } else if (p.out_act == S2E_ACT_TANH) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = tanhf(acc[j]);
            }
            if (p.aux_mode != S2E_AUX_NONE) {
                float aa[VEC];
                unpack16<T>(*(const u32x4_t*)(auxg + o), aa);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] *= (aa[j] > 0.f ? 1.f : neg);
            }
            *(u32x4_t*)(yg + o) = pack16<T>(acc);
        } else {
            const int co = (int)(v % p.Cout);
            float t = p.bias ? p.bias[co] : 0.f;
            for (int s = 0; s < p.splits; ++s) t += p.partial[s * slab + v];
            if (resg) t += load1<T>(resg + v);
            if (p.out_act == S2E_ACT_LRELU) t = lrelu02(t);
            else if (p.out_act == S2E_ACT_TANH) t = tanhf(t);
            if (p.aux_mode != S2E_AUX_NONE) t *= (load1<T>(auxg + v) > 0.f ? 1.f : neg);
            store1<T>(yg + v, t);
        }
    }
}

// ------------------------------------------------------------------------------------ host side
static int bn_for(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

extern "C" int s2e_conv_cout_pad(int cout) { const int bn = bn_for(cout); return ceil_div(cout, bn) * bn; }
extern "C" int s2e_conv_k_pad(int dtype, int k) { const int bk = dtype == S2E_BF16 ? 64 : 32; return ceil_div(k, bk) * bk; }

template <typename T>
static int launch_finish_t(const ConvKParams& p, hipStream_t st) {
    const long nvec = (long)p.M * p.Cout / Vec<T>::N + 1;
    const int fgrid = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
    conv_finish_kernel<T><<<fgrid, 256, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_finish_kernel");
    return S2E_OK;
}
static int launch_finish(int dtype, const ConvKParams& p, hipStream_t st) {
    return dtype == S2E_BF16 ? launch_finish_t<bf16_t>(p, st) : launch_finish_t<float>(p, st);
}

template <typename T, int BN>
static int launch_conv(const ConvKParams& p, hipStream_t st, bool s2 = false) {
    const int grid = p.tiles * p.splits;
    if (s2) {
        conv_igemm_kernel<T, BN, true, true, true><<<grid, 256, 0, st>>>(p);
        S2E_CHECK_LAUNCH("conv_igemm_kernel (stride-2 class mode)");
        return S2E_OK;
    }
    static const bool glds = [] { const char* e = getenv("S2E_IGEMM_GLDS"); return e ? atoi(e) != 0 : true; }();
    if (p.Cin % Vec<T>::N == 0) {
        if (glds && p.in_act == S2E_ACT_NONE) conv_igemm_kernel<T, BN, true, true><<<grid, 256, 0, st>>>(p);
        else conv_igemm_kernel<T, BN, true, false><<<grid, 256, 0, st>>>(p);
    } else conv_igemm_kernel<T, BN, false, false><<<grid, 256, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_igemm_kernel");
    if (p.splits > 1) return launch_finish_t<T>(p, st);
    return S2E_OK;
}

// Stride-2 data-gradients run per output-parity class (conv_igemm_kernel<..., S2 = true>) when the vector LDS-DMA
// loader applies; S2E_IGEMM_S2CLASS=0 falls back to the masked all-taps walk.
static bool s2_class_mode(int dtype, const s2e_conv_desc* d) {
    static const bool on = [] { const char* e = getenv("S2E_IGEMM_S2CLASS"); return e ? atoi(e) != 0 : true; }();
    static const bool glds = [] { const char* e = getenv("S2E_IGEMM_GLDS"); return e ? atoi(e) != 0 : true; }();
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    return on && glds && d->transposed && d->stride == 2 && d->in_act == S2E_ACT_NONE && d->Cin % vec == 0 && d->KH >= 2 && d->KW >= 2;
}
// pixel tiles of the four classes (oy & 1, ox & 1): prefix sums into t0[5]
static void s2_class_tiles(const s2e_conv_desc* d, int* t0) {
    t0[0] = 0;
    for (int c = 0; c < 4; ++c) {
        const int qy = c >> 1, qx = c & 1;
        const long mq = (long)d->N * ((d->Ho - qy + 1) / 2) * ((d->Wo - qx + 1) / 2);
        t0[c + 1] = t0[c] + ceil_div(mq, 128);
    }
}

// Split-K plan: layers whose output tiling cannot fill 256 CUs (small M, huge K: the 1024-channel
// blocks at 8x8 / 16x16, the encoder tail) are split over K-tiles so ~512 workgroups exist.
static void plan_splits(int dtype, const s2e_conv_desc* d, int* tiles, int* tiles_n, int* splits, int* per) {
    const int bn = bn_for(d->Cout);
    const int M = d->N * d->Ho * d->Wo;
    *tiles_n = ceil_div(d->Cout, bn);
    *tiles = ceil_div(M, 128) * (*tiles_n);
    const int nk = s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin) / (dtype == S2E_BF16 ? 64 : 32);
    if (s2_class_mode(dtype, d)) {                   // per-class tiles, never split (K shrinks to a quarter)
        int t0[5];
        s2_class_tiles(d, t0);
        *tiles = t0[4] * (*tiles_n);
        *per = nk; *splits = 1;
        return;
    }
    int s = 1;
    const int target = 512;                          // (2 workgroups per CU; 256 ... 512 measured equal, 768 and 1024 slower)
    if (*tiles < target && nk >= 8) {                // fewer than 2 workgroups per CU
        s = ceil_div(target, *tiles);
        // at least 4 K-tiles per split; a SHORT K (Cin = 128: 18 tiles) over >= 256 tiles is not worth splitting at all
        // (128->2048 @16^2: 259 -> 302 TFLOP/s unsplit; the 64-tile 8x8 layer still is: it would leave 3/4 of the CUs idle).
        // <= 4 tiles (the 8x8 [gamma | beta] data gradient, 2048 -> 128: K = 288 tiles): 16 a split -- at 4 the 72 fp32 slabs were 19 MB
        // beside 4.7 MB of weights, 43 -> 29 us; every other small-map shape measured equal or slower at 8, 16 and 32 (round 5)
        const int min_per = ((nk < 32 && *tiles >= 256) || *tiles <= 4) ? 16 : 4;
        if (s > nk / min_per) s = nk / min_per;
        if (s < 1) s = 1;
    }
    *per = ceil_div(nk, s);
    *splits = ceil_div(nk, *per);                    // no empty split
}

extern "C" size_t s2e_conv2d_workspace_bytes(int dtype, const s2e_conv_desc* d) {
    if (!d) return 0;
    if (s2e_small_conv_kind(dtype, d) != SMALL_NONE) return 0;
    if (s2e_conv_duo_plan(dtype, d, nullptr)) return 0;
    if (s2e_conv_patch_plan(dtype, d, nullptr)) return s2e_conv_patch_workspace_bytes(dtype, d);
    if (s2e_conv_stream_plan(dtype, d)) return s2e_conv_stream_workspace_bytes(dtype, d);
    int tiles, tiles_n, splits, per;
    plan_splits(dtype, d, &tiles, &tiles_n, &splits, &per);
    return splits > 1 ? (size_t)splits * d->N * d->Ho * d->Wo * d->Cout * sizeof(float) : 0;
}

extern "C" int s2e_conv2d_kernel_kind(int dtype, const s2e_conv_desc* d) {
    if (!d) return S2E_KERNEL_GENERIC;
    if (s2e_small_conv_kind(dtype, d) != SMALL_NONE) return S2E_KERNEL_SMALL;
    return (s2e_conv_duo_plan(dtype, d, nullptr) || s2e_conv_patch_plan(dtype, d, nullptr)) ? S2E_KERNEL_PATCH : S2E_KERNEL_GENERIC;
}

// The convolution WITH the InstanceNorm partial sums of its output (the statistics pass over a large map is
// the producer's epilogue): s2e_conv2d_stats_slots = the slots per sample the launch writes (0: this shape's kernel has no such
// epilogue -- run s2e_conv2d + s2e_in_stats), part = (N, slots, Cout, 2) floats {sum y, sum y^2}, to s2e_in_stats_from_partials.
extern "C" int s2e_conv2d_stats_slots(int dtype, const s2e_conv_desc* d) {
    s2e_patch_plan pplan;
    if (!d || d->transposed || d->aux_mode != S2E_AUX_NONE || s2e_small_conv_kind(dtype, d) != SMALL_NONE) return 0;
    if (!s2e_conv_duo_plan(dtype, d, &pplan)) return 0;
    return s2e_conv_duo_stats_slots(d, &pplan);
}

extern "C" int s2e_conv2d_stats(int dtype, const void* x, const void* w, const float* bias, const void* res, void* y,
                                const s2e_conv_desc* d, float* part, void* stream) {
    if (!x || !w || !y || !d || !part) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_stats: null pointer");
    s2e_patch_plan pplan;
    if (!s2e_conv2d_stats_slots(dtype, d) || !s2e_conv_duo_plan(dtype, d, &pplan))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d_stats: this shape's kernel writes no statistics (s2e_conv2d_stats_slots == 0)");
    return s2e_conv_duo_launch(&pplan, x, w, bias, res, nullptr, y, d, s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), part, nullptr, nullptr, (hipStream_t)stream);
}

// s2e_conv2d over a device-side list of 16 x 16 rectangles (label-sparse backward of the SPADE branch): the duo kernel's shapes only.
extern "C" int s2e_conv2d_rects_supported(int dtype, const s2e_conv_desc* d) {
    s2e_patch_plan pplan;
    return d && s2e_small_conv_kind(dtype, d) == SMALL_NONE && s2e_conv_duo_plan(dtype, d, &pplan) ? 1 : 0;
}

extern "C" int s2e_conv2d_rects(int dtype, const void* x, const void* w, const float* bias, const void* res, const void* aux, void* y,
                                const s2e_conv_desc* d, const int* rect_list, const int* rect_count, void* stream) {
    if (!x || !w || !y || !d || !rect_list || !rect_count) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_rects: null pointer");
    if (d->aux_mode != S2E_AUX_NONE && !aux) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_rects: aux_mode without aux");
    s2e_patch_plan pplan;
    if ((res && d->aux_mode != S2E_AUX_NONE) || !s2e_conv2d_rects_supported(dtype, d) || !s2e_conv_duo_plan(dtype, d, &pplan))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d_rects: this shape's kernel takes no rectangle list (s2e_conv2d_rects_supported)");
    return s2e_conv_duo_launch(&pplan, x, w, bias, res, aux, y, d, s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), nullptr, rect_list, rect_count,
                               (hipStream_t)stream);
}

extern "C" int s2e_conv2d(int dtype, const void* x, const void* w, const float* bias, const void* res,
                          const void* aux, void* y, const s2e_conv_desc* d, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (!x || !w || !y || !d) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: null pointer");
    if (d->stride != 1 && d->stride != 2) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d: stride %d", d->stride);
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Hi <= 0 || d->Wi <= 0)
        S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: bad shape");
    if (d->aux_mode != S2E_AUX_NONE && !aux) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: aux_mode without aux");
    if (d->KH * d->KW > 32) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d: kernel %dx%d has more than 32 taps", d->KH, d->KW);
    if ((long)d->N * d->Hi * d->Wi >= (1L << 31) || (long)d->N * d->Ho * d->Wo >= (1L << 31))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d: tensor too large for 32-bit pixel indices");
    if (const int kind = s2e_small_conv_kind(dtype, d)) {           // 1-channel heads: dedicated streaming kernels
        SmallConvParams sp{};
        sp.x = x; sp.w = w; sp.bias = bias; sp.res = res; sp.aux = aux; sp.y = y;
        sp.N = d->N; sp.Hi = d->Hi; sp.Wi = d->Wi; sp.Cin = d->Cin; sp.Ho = d->Ho; sp.Wo = d->Wo; sp.Cout = d->Cout;
        sp.KH = d->KH; sp.KW = d->KW; sp.stride = d->stride; sp.pad = d->pad;
        sp.in_act = d->in_act; sp.out_act = d->out_act; sp.aux_mode = d->aux_mode;
        sp.Kpad = s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin);
        return s2e_small_conv_launch(dtype, kind, sp, (hipStream_t)stream);
    }
    s2e_patch_plan pplan;
    if (!(res && d->aux_mode != S2E_AUX_NONE) && s2e_conv_duo_plan(dtype, d, &pplan))       // the big bf16 3x3 layers with >= 128 output channels: two staggered workgroups per CU (conv_duo.hip)
        return s2e_conv_duo_launch(&pplan, x, w, bias, res, aux, y, d, s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (s2e_conv_patch_plan(dtype, d, &pplan)) {      // big 3x3 / 4x4 stride-1 layers: patch-resident kernel
        const int patch_splits = pplan.splits;
        const size_t need = s2e_conv_patch_workspace_bytes(dtype, d);
        if (need && (!workspace || workspace_bytes < need))
            S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: this shape needs %zu bytes of workspace (s2e_conv2d_workspace_bytes)", need);
        const int rc = s2e_conv_patch_launch(dtype, &pplan, x, w, bias, res, aux, y, d,
                                             s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), (float*)workspace, (hipStream_t)stream);
        if (rc != S2E_OK || patch_splits == 1) return rc;
        ConvKParams f{};                             // split over channel chunks: combine the slabs, then the fused epilogue
        f.bias = bias; f.res = res; f.aux = aux; f.y = y;
        f.out_act = d->out_act; f.aux_mode = d->aux_mode;
        f.M = d->N * d->Ho * d->Wo; f.Cout = d->Cout; f.splits = patch_splits; f.partial = (float*)workspace;
        return launch_finish(dtype, f, (hipStream_t)stream);
    }
    if (s2e_conv_stream_plan(dtype, d))               // every other bf16 vector-channel shape: persistent stream-K kernel
        return s2e_conv_stream_launch(x, w, bias, res, aux, y, d, s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), workspace,
                                      workspace_bytes, (hipStream_t)stream);
    ConvKParams p;
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.aux = aux; p.y = y;
    p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.transposed = d->transposed;
    p.in_act = d->in_act; p.out_act = d->out_act; p.aux_mode = d->aux_mode;
    p.Ktot = d->KH * d->KW * d->Cin;
    p.Kpad = s2e_conv_k_pad(dtype, p.Ktot);
    p.M = d->N * d->Ho * d->Wo;
    const int bn = bn_for(d->Cout);
    plan_splits(dtype, d, &p.tiles, &p.tiles_n, &p.splits, &p.kt_per_split);
    const bool s2 = s2_class_mode(dtype, d);
    if (s2) s2_class_tiles(d, p.cls_tile0);
    // Tile order.  xcd_remap hands each XCD a contiguous range of tile ids, and whatever operand panel those tiles do
    // NOT share is fetched into that XCD's L2 once per XCD.  Default: Cout tiles fastest (neighbours share the
    // im2col panel).  The PMC pass shows the price on the small-M, 1024-channel layers: 80-190 MB fetched per launch
    // against ~25 MB of operands (the weight matrix arrives in all 8 L2s).  Walking the pixel tiles fastest instead
    // (S2E_IGEMM_TMFAST=1: one weight panel per XCD, the small activation replicated) was measured and changes
    // nothing on 1024->1024 @16^2 (577 vs 579 TFLOP/s) and costs 8 % on 128->2048: the re-reads are served by the
    // Infinity Cache and are not what bounds these launches.  Not used (the kernel keeps the code path; p.tm_fast = 0).
    p.tm_fast = 0;                                   // (the pixel-tiles-fastest order: measured, no gain -- see above)
    p.partial = (float*)workspace;
    if (p.splits > 1 && (!workspace || workspace_bytes < s2e_conv2d_workspace_bytes(dtype, d)))
        S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: this shape needs %zu bytes of workspace (s2e_conv2d_workspace_bytes)",
                 s2e_conv2d_workspace_bytes(dtype, d));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) {
        if (bn == 128) return launch_conv<bf16_t, 128>(p, st, s2);
        if (bn == 64) return launch_conv<bf16_t, 64>(p, st, s2);
        return launch_conv<bf16_t, 32>(p, st, s2);
    } else if (dtype == S2E_F32) {
        if (bn == 128) return launch_conv<float, 128>(p, st, s2);
        if (bn == 64) return launch_conv<float, 64>(p, st, s2);
        return launch_conv<float, 32>(p, st, s2);
    }
    S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: bad dtype %d", dtype);
}
