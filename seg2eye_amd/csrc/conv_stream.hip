// Persistent stream-K implicit-GEMM convolution for gfx950 (bf16): the forward convs and data-gradients that the
// patch-resident kernel (conv_patch.hip) does not take -- the PatchGAN's 4x4 stride-2 / stride-1 layers
// (discriminator.py:84-96), the encoder's 3x3 stride-2 layers (encoder.py:23-39), the 1x1 learned shortcuts
// (architecture.py:33-34,53-56), the 8x8 / 16x16 generator layers and every stride-2 data-gradient.
//
// Why a third conv kernel.  Those launches are SMALL (10-60 GFLOP) and odd-sized: with one workgroup per 128 x 128 output
// tile (conv_igemm.hip) a launch has 30-1000 tiles of very different K, so either the chip is half empty, or a second
// round of workgroups runs for a handful of tiles (529 tiles on 512 slots), or K is cut into fp32 slabs that a second
// kernel adds up -- and inside a workgroup every K-step waits out one full memory round trip, because the double buffer
// keeps ONE K-step in flight (1.5-3 us per K-step measured against 0.25 us of MFMA work).
//
//   work unit   : one K-step (64 bf16 of K) of one 128-pixel x BN-channel output tile.  The launch's units, tile-major,
//                 are cut into G equal contiguous ranges, one per workgroup, G = the number of CUs (stream-K): every CU
//                 gets the same number of K-steps whatever the tile count, and no split factor has to be chosen.
//   pipeline    : a ring of S LDS stages filled by LDS-DMA (global_load_lds, 16 B per lane, source-side swizzle as in
//                 conv_igemm.hip); S - 1 K-steps are in flight while one is multiplied; counted s_waitcnt vmcnt, one bare
//                 s_barrier per K-step; fragment reads are inline-asm ds_read_b128 in two register sets with counted
//                 lgkmcnt waits (through the builtins the compiler drains every LDS-DMA before each LDS read).  The
//                 loader runs ahead ACROSS tile boundaries: the next tile's first K-steps land while this tile is
//                 written out.
//   tiles cut by a range boundary: each side stores its fp32 partial accumulators (fragment order, 256-B coalesced) into
//                 its own workspace slot -- at most two per workgroup -- and conv_stream_fixup_kernel adds the slots of a
//                 tile in a FIXED order and runs the same epilogue.  No atomics: results are bit-reproducible.
//   large launches (>= 8 tiles per CU, one K length): ranges are rounded to whole tiles and no fix-up launch is needed.
//   stride-2 data-gradients: per output-parity class as in conv_igemm.hip (a class walks only its quarter of the taps);
//                 classes have different K lengths, which the unit space absorbs.
//   epilogue    : accumulators -> LDS (fp32, 64 rows at a time, in a region of its own beside the ring) -> 16-B row
//                 stores with bias / residual / activation / mask fused (same contract as conv_igemm.hip).
#include "conv_stream.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) const uint32_t sk_zero16[4] = {0u, 0u, 0u, 0u};

struct StreamParams {
    const void* x; const void* w; const float* bias; const void* res; const void* aux; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad, transposed;
    int out_act, aux_mode;
    int Kpad, tiles_n;
    int ncls;                          // 1, or 4: stride-2 data-gradient by output-parity class (oy & 1, ox & 1)
    int cls_nk[4];                     // K-steps of one tile of each class
    int cls_unit0[5];                  // first work unit of each class; cls_unit0[ncls] = all units (< 2^31: the plan checks)
    int units;
    int whole_tiles;                   // ranges are cut at tile boundaries (one K length, no fix-up launch)
    int G;                             // workgroups of the main launch (= ranges)
    int rq, rr, rmul;                  // range i begins at (i * rq + min(i, rr)) * rmul  (rmul = K-steps per tile when whole_tiles)
    float* partial;                    // [2 * G][128 * BN] fp32 slots
    unsigned x_bytes, w_bytes;         // buffer sizes for the loader's range-checked loads (< 2^31: the plan checks)
    // division by launch constants = multiply-high by floor(2^32 / d) + one correction step (sk_div)
    unsigned m_nk[4], m_hw[4], m_w[4], m_tn, m_cin, m_kw, m_nkx[2];
};

struct SkTile { int cls, nk, tml, tn, qy, qx, step, Hq, Wq, Mq; };

// n / d for 0 <= n < 2^31, d >= 1, with m = floor(2^32 / d) (0xffffffff for d = 1): the estimate is q or q - 1
__device__ __forceinline__ int sk_div(int n, int d, unsigned m) {
    int q = (int)__umulhi((unsigned)n, m);
    if (n - q * d >= d) ++q;
    return q;
}

__device__ __forceinline__ int sk_begin(const StreamParams& p, int i) {
    return (i * p.rq + (i < p.rr ? i : p.rr)) * p.rmul;
}

__device__ __forceinline__ SkTile sk_decode(const StreamParams& p, int u, int* kt) {
    int c = 0;
    while (c + 1 < p.ncls && u >= p.cls_unit0[c + 1]) ++c;
    const int local = u - p.cls_unit0[c];
    SkTile t;
    t.nk = p.cls_nk[c];
    t.cls = c;
    const int tl = sk_div(local, t.nk, p.m_nk[c]);
    *kt = local - tl * t.nk;
    t.tml = sk_div(tl, p.tiles_n, p.m_tn); t.tn = tl - t.tml * p.tiles_n;
    if (p.ncls > 1) { t.qy = c >> 1; t.qx = c & 1; t.step = 2; t.Hq = (p.Ho - t.qy + 1) >> 1; t.Wq = (p.Wo - t.qx + 1) >> 1; }
    else { t.qy = 0; t.qx = 0; t.step = 1; t.Hq = p.Ho; t.Wq = p.Wo; }
    t.Mq = p.N * t.Hq * t.Wq;
    return t;
}

// output pixel of row r of a tile (class-local rows in stride-2 class mode); false: past the last pixel
__device__ __forceinline__ bool sk_row_pixel(const StreamParams& p, const SkTile& t, int r, int& n, int& oy, int& ox) {
    const int ml = t.tml * 128 + r;
    if (ml >= t.Mq) return false;
    const int hw = t.Hq * t.Wq;
    n = sk_div(ml, hw, p.m_hw[t.cls]);
    const int rem = ml - n * hw, y2 = sk_div(rem, t.Wq, p.m_w[t.cls]);
    oy = t.step * y2 + t.qy; ox = t.step * (rem - y2 * t.Wq) + t.qx;
    return true;
}

#ifndef SK_LOADERS
#define SK_LOADERS 8
#endif
template <int BN> struct SkCfg {
    static constexpr int TN = BN / 64, NB = BN / 32;
    static constexpr int A_BYTES = 128 * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int S = BN == 128 ? 4 : 5;                       // ring stages; S - 1 K-steps in flight
    static constexpr int EP_BYTES = 64 * BN * 4;                      // epilogue staging: 64 rows fp32
    static constexpr int LDS_BYTES = S * STAGE + EP_BYTES;            // 160 KiB (BN = 128) / 136 KiB (BN = 64)
};

// ---- epilogue shared by the main kernel and the fix-up kernel.  WAVE-LOCAL: a wave (one of a 2 x 2 grid, wave tile
// 64 x BN/2, MFMA accumulator layout; the caller has added the bias) stages 32 of its rows at a time in its OWN slice of Cs
// (fp32 [32][BN/2]) and writes them out as 16-byte row pieces with residual / LeakyReLU / mask fused.  No workgroup barrier:
// in the main kernel the loader waves keep running underneath.
template <int BN>
__device__ __forceinline__ void sk_epilogue(const StreamParams& p, const SkTile& t, f32x16_t (&acc)[2][SkCfg<BN>::TN], char* Cs_c, int wave) {
    constexpr int TN = SkCfg<BN>::TN, WTN = BN / 2;
    constexpr int LPR = WTN / 8, RPS = 64 / LPR, SWEEPS = 32 / RPS;     // lanes per row, rows per sweep
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef bf16_t T;
    const int lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const uint32_t cs0 = (uint32_t)(uintptr_t)(lptr_t)Cs_c + (uint32_t)(wave * 32 * WTN * 4);
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ resg = (const T*)p.res;
    const T* __restrict__ auxg = (const T*)p.aux;
    const int cw = (lane % LPR) * 8;
    const int co = t.tn * BN + wn * WTN + cw;
    const bool cok = co < p.Cout;                                      // Cout is a multiple of 8: whole vectors
    const uint32_t wbase = cs0 + (uint32_t)((4 * h) * WTN * 4 + l31 * 4);
    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        size_t o[SWEEPS]; bool live[SWEEPS];
        u32x4_t rr[SWEEPS], aa[SWEEPS];
#pragma unroll
        for (int sw = 0; sw < SWEEPS; ++sw) {
            int n, oy, ox;
            live[sw] = sk_row_pixel(p, t, wm * 64 + mi * 32 + sw * RPS + lane / LPR, n, oy, ox) && cok;
            o[sw] = live[sw] ? ((size_t)(n * p.Ho + oy) * p.Wo + ox) * p.Cout + co : 0;
            rr[sw] = u32x4_t{0u, 0u, 0u, 0u}; aa[sw] = rr[sw];
            if (live[sw] && resg) rr[sw] = *(const u32x4_t*)(resg + o[sw]);
            if (live[sw] && p.aux_mode != S2E_AUX_NONE) aa[sw] = *(const u32x4_t*)(auxg + o[sw]);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(wbase), "v"(acc[mi][ni][r]),
                             "n"(((r & 3) + 8 * (r >> 2)) * WTN * 4 + ni * 32 * 4) : "memory");
#pragma unroll
        for (int sw = 0; sw < SWEEPS; ++sw) {
            const uint32_t ra = cs0 + (uint32_t)((sw * RPS + lane / LPR) * WTN * 4 + cw * 4);
            f32x4_t f0, f1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(f0), "=&v"(f1) : "v"(ra) : "memory");
            if (!live[sw]) continue;
            float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            if (resg) {
                float q[8];
                unpack16<T>(rr[sw], q);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += q[j];
            }
            if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = lrelu02(v[j]);
            }
            if (p.aux_mode != S2E_AUX_NONE) {
                float q[8];
                unpack16<T>(aa[sw], q);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= (q[j] > 0.f ? 1.f : neg);
            }
            *(u32x4_t*)(yg + o[sw]) = pack16<T>(v);
        }
    }
}

// 512 threads: waves 0-3 multiply (2 x 2 grid of 64 x BN/2 wave tiles), waves 4-7 load.  A wave that issues an LDS-DMA piece
// stalls until the CU's vector-memory path takes it (the path, ~25 B/clk per CU, is what these launches run against: a
// 128 x 128 x 64 K-step is 32 KB = ~1300 cycles of it against 512 cycles of MFMA issue per SIMD), and a stalled wave issues
// no MFMA either: with both jobs in the same four waves (first version of this kernel) the two phases alternated instead of
// overlapping and the kernel ran at HALF the speed of conv_igemm.hip's two workgroups per CU.  One barrier per K-step is the
// only workgroup-wide synchronisation: loaders wait (vmcnt) for their pieces of step j, everybody meets, loaders refill the
// stage step j - 1 was read from while the multiply waves work on step j.
template <int BN>
__global__ __launch_bounds__(256 + 64 * SK_LOADERS, 1) void conv_stream_kernel(const StreamParams p) {
    typedef SkCfg<BN> Cfg;
    typedef bf16_t T;
    constexpr int VEC = 8, BK = 64;
    constexpr int WTN = BN / 2, TM = 2, TN = Cfg::TN, NB = Cfg::NB;
    constexpr int NLW = SK_LOADERS;                                   // loader waves
    constexpr int NA = 16 / NLW, NBW = (NB * 4 + NLW - 1) / NLW;      // activation / weight pieces (8 rows each) per loader wave per K-step
    constexpr int LPT = NA + NBW;                                     // (a wave without a weight piece -- BN = 64, 8 loaders: none -- would break the count)
    static_assert(16 % NLW == 0 && (NB * 4) % NLW == 0, "every loader wave issues the same number of pieces");
    constexpr int A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE, S = Cfg::S, PD = S - 1;
    __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgi = xcd_remap(blockIdx.x, p.G);
    const int u_begin = sk_begin(p, wgi), u_end = sk_begin(p, wgi + 1);
    if (u_begin >= u_end) return;

    if (wave >= 4) {
        // ------------------------------------------------------------ loader waves (run PD units ahead of the multiply)
        // loader wave w, piece i fills rows 8 (w + NLW i) .. + 7 of a stage; lane -> row offset lane >> 3, physical chunk
        // lane & 7 = logical chunk ^ ((row >> 1) & 7), the same for every i
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int lw = wave - 4;
        const int r0 = 8 * lw + (lane >> 3);
        const int c = (lane & 7) ^ ((r0 >> 1) & 7);
        // Range-checked buffer loads: an offset past the buffer returns ZEROS, which is what a padding tap, a row past the
        // last pixel or a K column past the class's taps must read -- one v_cndmask per piece instead of a 64-bit address
        // select against a zero page, and 32-bit offsets throughout (the loader waves run one per SIMD: their VALU work is
        // on the critical path; with 64-bit addresses and compiler-made branches per piece this loop paced the kernel).
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;
        const int KH = p.KH, KW = p.KW, Cin = p.Cin, Hi = p.Hi, Wi = p.Wi, pad = p.pad, stride = p.stride;
        const bool transposed = p.transposed != 0, cmode = p.ncls > 1;
        unsigned rowoff[NA], vmask[NA];                               // byte offset of a row's base pixel (wraps), tap validity bits
        unsigned wrow = 0;                                            // byte offset of weight row tn * BN + r0
        const unsigned jstride = 16u * NLW * (unsigned)p.Kpad;        // 8 * NLW weight rows, in bytes
        int l_ky = 0, l_kx = 0, l_ci = 0, l_kt = 0, l_nk = 0, l_ky0 = 0, l_kx0 = 0;
        const int tap_sy = transposed ? -Wi : Wi, tap_sx = transposed ? -1 : 1;
        const int tap_sh = (transposed && stride == 2) ? 1 : 0;
        auto loader_seek = [&](int u) __attribute__((always_inline)) {
            int kt;
            const SkTile t = sk_decode(p, u, &kt);
            l_kt = kt; l_nk = t.nk;
            int nkx = KW; unsigned m_nkx = p.m_kw;
            if (cmode) { l_ky0 = (t.qy + pad) & 1; l_kx0 = (t.qx + pad) & 1; nkx = (KW - l_kx0 + 1) >> 1; m_nkx = p.m_nkx[l_kx0]; }
            else { l_ky0 = 0; l_kx0 = 0; }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int n, oy, ox;
                if (!sk_row_pixel(p, t, r0 + 8 * NLW * i, n, oy, ox)) { vmask[i] = 0u; rowoff[i] = 0u; continue; }
                const int by = transposed ? oy + pad : oy * stride - pad;
                const int bx = transposed ? ox + pad : ox * stride - pad;
                // the valid taps of one axis are an interval [lo, hi]: forward 0 <= b + k < lim; data-gradient 0 <= b - k and
                // (b - k) >> sh < lim (in class mode only taps of the right parity are walked, so parity is not tested here)
                auto range1 = [&](int b, int lim, int K) __attribute__((always_inline)) -> unsigned {
                    int lo, hi;
                    if (!transposed) { lo = -b; hi = lim - 1 - b; }
                    else { hi = b; lo = b - (lim << tap_sh) + 1; }
                    lo = lo < 0 ? 0 : lo; hi = hi > K - 1 ? K - 1 : hi;
                    return hi < lo ? 0u : ((2u << hi) - 1u) & ~((1u << lo) - 1u);
                };
                const unsigned xm = range1(bx, Wi, KW), ym = range1(by, Hi, KH);
                unsigned m = 0;
                for (int ky = 0; ky < KH; ++ky) m |= ((ym >> ky) & 1u) ? (xm << (ky * KW)) : 0u;
                vmask[i] = m;
                const int rb = (n * Hi + (by >> tap_sh)) * Wi + (bx >> tap_sh);
                rowoff[i] = 2u * (unsigned)(rb * Cin);                // (rb may be negative: the sum with a valid tap's offset is not)
            }
            wrow = 2u * (unsigned)((t.tn * BN + r0) * p.Kpad);
            const int k0 = kt * BK + c * VEC, tap = sk_div(k0, Cin, p.m_cin);
            l_ci = k0 - tap * Cin;
            if (cmode) { const int jy = sk_div(tap, nkx, m_nkx); l_ky = l_ky0 + 2 * jy; l_kx = l_kx0 + 2 * (tap - jy * nkx); }
            else { l_ky = sk_div(tap, KW, p.m_kw); l_kx = tap - l_ky * KW; }
        };
        auto dma_unit = [&](int stage) __attribute__((always_inline)) {
            char* base = smem + stage * STAGE;
            const bool kvalid = l_ky < KH;                           // <=> this chunk's k is inside the real K range
            const unsigned tbit = kvalid ? 1u << (l_ky * KW + l_kx) : 0u;
            const unsigned koff = 2u * (unsigned)(((l_ky >> tap_sh) * tap_sy + (l_kx >> tap_sh) * tap_sx) * Cin + l_ci);
            static_for<0, NA>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const unsigned off = (vmask[i] & tbit) ? rowoff[i] + koff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(base + (8 * lw + 8 * NLW * i) * 128), 16, (int)off, 0, 0, 0);
            });
            // weight column of this chunk: linear in k, except in class mode, where k walks the class's tap subset
            const unsigned wb = cmode ? (kvalid ? wrow + 2u * (unsigned)((l_ky * KW + l_kx) * Cin + l_ci) : OOB)
                                      : wrow + 2u * (unsigned)(l_kt * BK + c * VEC);
            static_for<0, NBW>([&](auto J) {
                constexpr int j = decltype(J)::value;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(base + A_BYTES + (8 * lw + 8 * NLW * j) * 128), 16, (int)(wb + j * jstride), 0, 0, 0);
            });
            l_ci += BK;
            if (cmode) { while (l_ci >= Cin) { l_ci -= Cin; l_kx += 2; if (l_kx >= KW) { l_kx = l_kx0; l_ky += 2; } } }
            else { while (l_ci >= Cin) { l_ci -= Cin; if (++l_kx == KW) { l_kx = 0; ++l_ky; } } }
            ++l_kt;
        };
        // all but the `keep` youngest K-steps' pieces of this wave have landed
        auto wait_units = [&](int keep) __attribute__((always_inline)) {
            static_assert(PD <= 4 && 3 * LPT <= 63, "wait table");
            if (keep <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (keep == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT) : "memory");
            else if (keep == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * LPT) : "memory");
        };
        // one loop for the fill (the first PD turns only load) and the steady state, so that the per-tile set-up (a few
        // hundred instructions of index arithmetic) exists once in the code
        int lu = u_begin, lstage = 0;
        bool seek = true;
        for (int cu = u_begin - PD; cu < u_end; ++cu) {
            if (cu >= u_begin) {
                const int left = u_end - 1 - cu;                      // units after this one
                wait_units(left < PD - 1 ? left : PD - 1);
                __builtin_amdgcn_s_barrier();   // unit cu has landed (every loader waited for its pieces); the multiply is past unit cu - 1
            }
            if (lu < u_end) {                   // refill the stage unit cu - 1 was read from
                if (seek) loader_seek(lu);
                dma_unit(lstage);
                seek = l_kt == l_nk;
                lstage = lstage == S - 1 ? 0 : lstage + 1;
                ++lu;
            }
        }
        return;
    }

    // ---------------------------------------------------------------- multiply waves
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    uint32_t a_off[TM], b_off[TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = wm * 64 + mi * 32 + l31;
        a_off[mi] = lds0 + row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = wn * WTN + ni * 32 + l31;
        b_off[ni] = lds0 + A_BYTES + row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    }
    u32x4_t fa[2][TM], fb[2][TN];
    f32x16_t acc[TM][TN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    };
    auto read_frags = [&](int set, int sstep, uint32_t sbase) __attribute__((always_inline)) {   // logical chunk 2 s + h
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][mi]) : "v"((a_off[mi] + sbase) ^ (uint32_t)(sstep << 5)) : "memory");
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][ni]) : "v"((b_off[ni] + sbase) ^ (uint32_t)(sstep << 5)) : "memory");
    };
    auto frags_ready = [&](int set, bool all) __attribute__((always_inline)) {
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
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[set][mi]),
                                                                      __builtin_bit_cast(bf16x8_t, fb[set][ni]), acc[mi][ni], 0, 0, 0);
    };
    // the bias of this lane's accumulator columns: requested when the multiply moves to a new tile, added before the write-out
    float bfrag[TN];
    auto bias_request = [&](const SkTile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = t.tn * BN + wn * WTN + ni * 32 + l31;
            bfrag[ni] = (p.bias && col < p.Cout) ? p.bias[col] : 0.f;
        }
    };

    int kt_first = 0, c_kt = 0, seg = 0, cstage = 0;
    SkTile cur = sk_decode(p, u_begin, &kt_first);
    c_kt = kt_first;
    zero_acc();
    bias_request(cur);
    for (int cu = u_begin; cu < u_end; ++cu) {
        const uint32_t sbase = (uint32_t)(cstage * STAGE);
        __builtin_amdgcn_s_barrier();                                 // unit cu is in LDS
        read_frags(0, 0, sbase);
        read_frags(1, 1, sbase); frags_ready(0, false); mfmas(0);
        read_frags(0, 2, sbase); frags_ready(1, false); mfmas(1);
        read_frags(1, 3, sbase); frags_ready(0, false); mfmas(0);
        frags_ready(1, true); mfmas(1);
        cstage = cstage == S - 1 ? 0 : cstage + 1;
        ++c_kt;
        if (c_kt == cur.nk || cu + 1 == u_end) {                      // this workgroup's part of the tile is complete
            if (kt_first == 0 && c_kt == cur.nk) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bfrag[ni];
                sk_epilogue<BN>(p, cur, acc, smem + S * STAGE, wave);
            } else {                                                  // cut by a range boundary: partial sums -> slot
                float* slot = p.partial + (size_t)(2 * wgi + (seg > 0 ? 1 : 0)) * (128 * BN);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            slot[(((wave * TM + mi) * TN + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
            }
            ++seg;
            if (cu + 1 < u_end) {
                cur = sk_decode(p, cu + 1, &kt_first);
                c_kt = kt_first;
                zero_acc();
                bias_request(cur);
            }
        }
    }
}

// One workgroup per range boundary: the tile the boundary cuts is summed and written out by the workgroup of the FIRST
// boundary inside it; slots in range order, so the sum has one fixed order.
template <int BN>
__global__ __launch_bounds__(256) void conv_stream_fixup_kernel(const StreamParams p) {
    typedef SkCfg<BN> Cfg;
    constexpr int TM = 2, TN = Cfg::TN;
    __shared__ __attribute__((aligned(16))) char smem[Cfg::EP_BYTES];
    const int b = blockIdx.x;
    if (b + 1 >= p.G) return;
    const int u = sk_begin(p, b + 1);
    if (u >= p.units) return;
    int kt;
    const SkTile t = sk_decode(p, u, &kt);
    if (kt == 0) return;                                              // the boundary sits between two tiles
    const int ut0 = u - kt, ut1 = ut0 + t.nk;
    if (sk_begin(p, b) > ut0) return;                                 // an earlier boundary cuts this tile too: its workgroup sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int i = b; i < p.G; ++i) {
        const int bi = sk_begin(p, i);
        if (bi >= ut1) break;
        if (sk_begin(p, i + 1) <= bi) continue;                       // (an empty range wrote nothing)
        const float* slot = p.partial + (size_t)(2 * i + (bi >= ut0 ? 0 : 1)) * (128 * BN);   // its first segment, or its last
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[mi][ni][r] += slot[(((wave * TM + mi) * TN + ni) * 16 + r) * 64 + lane];
    }
    if (p.bias) {
        const int wn = wave & 1, l31 = lane & 31;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = t.tn * BN + wn * (BN / 2) + ni * 32 + l31;
            const float bv = col < p.Cout ? p.bias[col] : 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] += bv;
        }
    }
    sk_epilogue<BN>(p, t, acc, smem, wave);
}

int sk_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

bool sk_class_mode(const s2e_conv_desc* d) { return d->transposed && d->stride == 2; }
int sk_bn(const s2e_conv_desc* d) { return d->Cout > 64 ? 128 : 64; }

// false: more than 2^31 work units (the kernel indexes them with 32-bit integers)
bool sk_fill_units(const s2e_conv_desc* d, int kpad, StreamParams* p) {
    const int bn = sk_bn(d);
    p->tiles_n = ceil_div(d->Cout, bn);
    long unit0[5] = {0, 0, 0, 0, 0};
    long tiles = 0;
    if (sk_class_mode(d)) {
        p->ncls = 4;
        for (int c = 0; c < 4; ++c) {
            const int qy = c >> 1, qx = c & 1;
            const long mq = (long)d->N * ((d->Ho - qy + 1) / 2) * ((d->Wo - qx + 1) / 2);
            const int ky0 = (qy + d->pad) & 1, kx0 = (qx + d->pad) & 1;
            const int nky = (d->KH - ky0 + 1) >> 1, nkx = (d->KW - kx0 + 1) >> 1;
            p->cls_nk[c] = ceil_div((long)nky * nkx * d->Cin, 64);
            if (p->cls_nk[c] < 1) p->cls_nk[c] = 1;
            const long tc = (long)ceil_div(mq, 128) * p->tiles_n;
            tiles += tc;
            unit0[c + 1] = unit0[c] + tc * p->cls_nk[c];
        }
    } else {
        p->ncls = 1;
        p->cls_nk[0] = kpad / 64;
        p->cls_nk[1] = p->cls_nk[2] = p->cls_nk[3] = 0;
        tiles = (long)ceil_div((long)d->N * d->Ho * d->Wo, 128) * p->tiles_n;
        unit0[1] = tiles * p->cls_nk[0];
        unit0[2] = unit0[3] = unit0[4] = unit0[1];
    }
    if (unit0[p->ncls] >= (1L << 31) - 4096) return false;
    for (int c = 0; c < 5; ++c) p->cls_unit0[c] = (int)unit0[c];
    p->units = (int)unit0[p->ncls];
    const int cus = sk_cu_count();
    p->G = p->units < cus ? p->units : cus;
    if (p->G < 1) p->G = 1;
    bool uniform = true;
    for (int c = 1; c < p->ncls; ++c) uniform = uniform && p->cls_nk[c] == p->cls_nk[0];
    p->whole_tiles = uniform && tiles >= 8L * p->G;
    if (p->whole_tiles) { p->rq = (int)(tiles / p->G); p->rr = (int)(tiles % p->G); p->rmul = p->cls_nk[0]; }
    else { p->rq = p->units / p->G; p->rr = p->units % p->G; p->rmul = 1; }
    auto magic = [](long d) -> unsigned { return d <= 1 ? 0xffffffffu : (unsigned)((1UL << 32) / (unsigned long)d); };
    for (int c = 0; c < 4; ++c) {
        const int qy = c >> 1, qx = c & 1;
        const long hq = p->ncls > 1 ? (d->Ho - qy + 1) / 2 : d->Ho, wq = p->ncls > 1 ? (d->Wo - qx + 1) / 2 : d->Wo;
        p->m_nk[c] = magic(p->cls_nk[c]); p->m_hw[c] = magic(hq * wq); p->m_w[c] = magic(wq);
    }
    p->m_tn = magic(p->tiles_n); p->m_cin = magic(d->Cin); p->m_kw = magic(d->KW);
    p->m_nkx[0] = magic((d->KW + 1) >> 1); p->m_nkx[1] = magic(d->KW >> 1);
    const long xb = (long)d->N * d->Hi * d->Wi * d->Cin * 2, wb = (long)s2e_conv_cout_pad(d->Cout) * kpad * 2;
    if (xb >= (1L << 31) || wb >= (1L << 31)) return false;
    p->x_bytes = (unsigned)xb; p->w_bytes = (unsigned)wb;
    return true;
}

}  // namespace

// S2E_CONV_STREAM: 0 = never; 1 (default) = the shapes it measured faster on (long-K tiles: >= 16 K-steps per tile and >= 64
// tiles, no parity classes -- DESIGN 3.1e has the per-shape table); 2 = every shape the kernel can run (tests, A/B runs).
int s2e_conv_stream_plan(int dtype, const s2e_conv_desc* d) {
    static const int mode = [] { const char* e = getenv("S2E_CONV_STREAM"); return e ? atoi(e) : 1; }();
    if (mode <= 0 || dtype != S2E_BF16) return 0;
    if (d->Cin % 8 != 0 || d->Cout % 8 != 0 || d->Cout <= 32 || d->in_act != S2E_ACT_NONE) return 0;
    if (d->KH * d->KW > 32 || d->out_act == S2E_ACT_TANH) return 0;
    if (d->transposed && d->stride == 2 && (d->KH < 2 || d->KW < 2)) return 0;
    StreamParams p{};
    if (!sk_fill_units(d, s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), &p)) return 0;
    if (mode >= 2) return 1;
    if (p.ncls > 1) return 0;
    const long tiles = p.units / p.cls_nk[0];
    return p.cls_nk[0] >= 16 && tiles >= 64;
}

size_t s2e_conv_stream_workspace_bytes(int dtype, const s2e_conv_desc* d) {
    if (!s2e_conv_stream_plan(dtype, d)) return 0;
    StreamParams p{};
    sk_fill_units(d, s2e_conv_k_pad(dtype, d->KH * d->KW * d->Cin), &p);
    return p.whole_tiles ? 0 : (size_t)2 * p.G * 128 * sk_bn(d) * sizeof(float);
}

int s2e_conv_stream_launch(const void* x, const void* w, const float* bias, const void* res, const void* aux, void* y,
                           const s2e_conv_desc* d, int kpad, void* workspace, size_t workspace_bytes, hipStream_t st) {
    StreamParams p{};
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.aux = aux; p.y = y;
    p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.transposed = d->transposed;
    p.out_act = d->out_act; p.aux_mode = d->aux_mode;
    p.Kpad = kpad;
    if (!sk_fill_units(d, kpad, &p)) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d: too many work units for the stream kernel");
    if (p.units <= 0) return S2E_OK;
    const size_t need = p.whole_tiles ? 0 : (size_t)2 * p.G * 128 * sk_bn(d) * sizeof(float);
    if (need && (!workspace || workspace_bytes < need))
        S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d: this shape needs %zu bytes of workspace (s2e_conv2d_workspace_bytes)", need);
    p.partial = (float*)workspace;
    if (sk_bn(d) == 128) conv_stream_kernel<128><<<p.G, 256 + 64 * SK_LOADERS, 0, st>>>(p);
    else conv_stream_kernel<64><<<p.G, 256 + 64 * SK_LOADERS, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_stream_kernel");
    if (!p.whole_tiles && p.G > 1) {
        if (sk_bn(d) == 128) conv_stream_fixup_kernel<128><<<p.G - 1, 256, 0, st>>>(p);
        else conv_stream_fixup_kernel<64><<<p.G - 1, 256, 0, st>>>(p);
        S2E_CHECK_LAUNCH("conv_stream_fixup_kernel");
    }
    return S2E_OK;
}
