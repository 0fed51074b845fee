// Convolution weight gradient for gfx950 (NHWC, MFMA 32x32, fp32 accumulate + fp32 atomics).
//
//   dW[co][k] += sum_m gy[m][co] * xcol[m][k]      k = (ky*KW+kx)*Cin + ci,  m = (n,oy,ox)
//
// The reduction index m is the SLOW dimension of both operands in HBM (NHWC), while an MFMA lane
// needs consecutive reduction elements.  Tiles are therefore staged in LDS in their natural
// [pixel][channel] order (coalesced 16-B loads) and
//   bf16: fragments are fetched with ds_read_b64_tr_b16, the hardware 4x16 transpose read
//         (rows padded to 320 B so the four rows of one block fall in disjoint bank quarters);
//   f32 : the 32x32x2 MFMA takes ONE f32 per lane (row = lane&31, k = lane>>5), so a plain
//         ds_read_b32 of [pixel][channel] is already the right shape, conflict-free.
// One workgroup owns a 128(co) x 128(k) tile of dW and a contiguous slice of the pixels
// (split-K over workgroups to fill 256 CUs); partial tiles are combined with fp32 atomics in
// 128-B row segments (the shape global float atomics run at full rate for).
#include "common.h"
#include "conv_small.h"
#include "conv_wgrad_patch.h"
#include "conv_wgrad_flat.h"
#include <stdlib.h>

struct WgradParams {
    const void* x; const void* gy; float* dw; float* dbias;   // dbias != NULL: also accumulate sum_m gy[m][co]
    int N, Hi, Wi, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad, in_act;
    int Ktot, M, tiles_k, tiles_co, m_per_split;
    float* partial;        // NULL: fp32 atomics into dw / dbias.  Else [workgroup][128 x 128 fp32 in accumulator order]: plain stores,
    float* bpartial;       // [workgroup][128] bias partial sums; conv_wgrad_reduce_kernel adds both up in a FIXED order
};

template <typename T> struct WgLds;
template <> struct WgLds<bf16_t> { static constexpr int ROW = 320; };   // 128 bf16 = 256 B + 64 B pad
template <> struct WgLds<float>  { static constexpr int ROW = 512; };   // 128 f32

__device__ __forceinline__ u32x2_t lds_tr16_b64(const char* p) {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(u32x2_t, v);
}

// VECPATH: Cin and Cout are multiples of the 16-B vector width (every real layer except the 1- and
// 5-channel heads); the element-wise gather lives in its own instantiation.
// One workgroup's tile: block `blk` of `nblk` of the launch described by p (a launch of its own, or one job of a multi-job launch).
template <typename T, bool VECPATH, int BR>              // BR = pixels per chunk (one barrier per chunk)
__device__ __forceinline__ void wgrad_tile(const WgradParams& p, int blk, int nblk, char* smem) {
    constexpr int VEC = Vec<T>::N;
    constexpr int ROW = WgLds<T>::ROW;
    constexpr int OP_BYTES = BR * ROW;                   // one operand tile
    constexpr int STAGE = 2 * OP_BYTES;
    constexpr int CPR = 128 / VEC;                       // 16-B chunks per row
    constexpr int RPT = 256 / CPR;                       // rows per pass
    constexpr int NI = BR / RPT;                         // vectors per thread per operand

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;             // 2x2 waves, each 64(co) x 64(k)
    // consecutive logical ids (= the (co,k) tiles of ONE pixel slice, which stream the same gy / x rows in
    // lockstep) are placed on one XCD so they share that XCD's L2 instead of each L2 re-fetching the slice
    int bid = xcd_remap(blk, nblk);
    const int tk = bid % p.tiles_k; bid /= p.tiles_k;
    const int tco = bid % p.tiles_co; const int split = bid / p.tiles_co;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    if (m_begin >= m_end) return;

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ gg = (const T*)p.gy;
    const int c = tid % CPR, rb = tid / CPR;
    const int co0 = tco * 128 + c * VEC;                 // gy column of this thread's chunk
    const int k0 = tk * 128 + c * VEC;                   // xcol column of this thread's chunk
    const int HoWo = p.Ho * p.Wo;
    // this thread's fixed tap (vector path)
    int t_ci, t_ky, t_kx;
    { const int tap = k0 / p.Cin; t_ci = k0 - tap * p.Cin; t_ky = tap / p.KW; t_kx = tap - t_ky * p.KW; }
    const bool k_valid = k0 < p.Ktot;

    // per-row pixel coordinates, advanced by BR each chunk
    int rn[NI], roy[NI], rox[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = m_begin + rb + RPT * i;
        rn[i] = m / HoWo; const int rem = m - rn[i] * HoWo;
        roy[i] = rem / p.Wo; rox[i] = rem - roy[i] * p.Wo;
    }

    // bias gradient: the k-tile 0 workgroups also sum the gy vectors they stage (per thread: VEC channels of
    // its rows), reduced across the block's row-threads through LDS at the end -> one atomic per channel
    // the bias sums are spread over the k-tile workgroups of a slice: chunk ch is summed by the
    // workgroup with tk == ch % tiles_k (every gy row exactly once, no straggler tile)
    const bool want_bias = p.dbias != nullptr;
    float bsum[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bsum[j] = 0.f;
    u32x4_t rg[NI], rx[NI];
    auto load_chunk = [&](int ch) __attribute__((always_inline)) {
        static_for<0, NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int m = m_begin + ch * BR + rb + RPT * i;
            const bool mv = m < m_end;
            // ---- gy
            rg[i] = u32x4_t{0, 0, 0, 0};
            if constexpr (VECPATH) {
                if (mv && co0 < p.Cout) rg[i] = *(const u32x4_t*)(gg + (size_t)m * p.Cout + co0);
            } else {
                float f[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    f[j] = (mv && co0 + j < p.Cout) ? load1<T>(gg + (size_t)m * p.Cout + co0 + j) : 0.f;
                rg[i] = pack16<T>(f);
            }
            if (want_bias && (ch % p.tiles_k) == tk) {
                float f[VEC];
                unpack16<T>(rg[i], f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) bsum[j] += f[j];
            }
            // ---- xcol
            rx[i] = u32x4_t{0, 0, 0, 0};
            const int iy0 = roy[i] * p.stride - p.pad, ix0 = rox[i] * p.stride - p.pad;
            if constexpr (VECPATH) {
                const int iy = iy0 + t_ky, ix = ix0 + t_kx;
                if (mv && k_valid && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                    rx[i] = *(const u32x4_t*)(xg + ((size_t)(rn[i] * p.Hi + iy) * p.Wi + ix) * p.Cin + t_ci);
            } else {
                float f[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    f[j] = 0.f;
                    const int k = k0 + j;
                    if (mv && k < p.Ktot) {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin;
                        const int ky = tap / p.KW, kx = tap - ky * p.KW;
                        const int iy = iy0 + ky, ix = ix0 + kx;
                        if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                            f[j] = load1<T>(xg + ((size_t)(rn[i] * p.Hi + iy) * p.Wi + ix) * p.Cin + ci);
                    }
                }
                rx[i] = pack16<T>(f);
            }
            if (p.in_act == S2E_ACT_LRELU) {
                float f[VEC];
                unpack16<T>(rx[i], f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) f[j] = lrelu02(f[j]);
                rx[i] = pack16<T>(f);
            }
            // advance this row by BR pixels for the next chunk
            rox[i] += BR;
            while (rox[i] >= p.Wo) { rox[i] -= p.Wo; ++roy[i]; }
            while (roy[i] >= p.Ho) { roy[i] -= p.Ho; ++rn[i]; }
        });
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
        static_for<0, NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int r = rb + RPT * i;
            *(u32x4_t*)(base + r * ROW + c * 16) = rg[i];
            *(u32x4_t*)(base + OP_BYTES + r * ROW + c * 16) = rx[i];
        });
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int hh = lane >> 5, l31 = lane & 31;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* Gs = smem + buf * STAGE;
        const char* Xs = Gs + OP_BYTES;
        if constexpr (sizeof(T) == 2) {
            const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g2 = (lane >> 4) & 1;
#pragma unroll
            for (int ks = 0; ks < BR / 16; ++ks) {
                const int r = 16 * ks + 8 * hh + q;
                u32x4_t a[2], b[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const char* ptr = Gs + r * ROW + (wm * 64 + mi * 32 + 16 * g2 + 4 * pp) * 2;
                    const u32x2_t lo = lds_tr16_b64(ptr), hi = lds_tr16_b64(ptr + 4 * ROW);
                    a[mi] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const char* ptr = Xs + r * ROW + (wn * 64 + ni * 32 + 16 * g2 + 4 * pp) * 2;
                    const u32x2_t lo = lds_tr16_b64(ptr), hi = lds_tr16_b64(ptr + 4 * ROW);
                    b[ni] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, a[mi]), __builtin_bit_cast(bf16x8_t, b[ni]), acc[mi][ni], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < BR / 2; ++t) {
                const int r = 2 * t + hh;
                float a[2], b[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *(const float*)(Gs + r * ROW + (wm * 64 + mi * 32 + l31) * 4);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[ni] = *(const float*)(Xs + r * ROW + (wn * 64 + ni * 32 + l31) * 4);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    };

    const int nch = (m_end - m_begin + BR - 1) / BR;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch + 1 < nch; ++ch) {
        const int cur = ch & 1;
        load_chunk(ch + 1);
        compute(cur);
        store_chunk(cur ^ 1);
        __syncthreads();
    }
    compute((nch - 1) & 1);

    const int wg_id = (split * p.tiles_co + tco) * p.tiles_k + tk;     // = the logical id decoded above
    if (want_bias) {                                      // block-uniform
        __syncthreads();
        float* red = (float*)smem;                        // [RPT][128]
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[rb * 128 + c * VEC + j] = bsum[j];
        __syncthreads();
        if (tid < 128) {
            float a = 0.f;
            for (int r = 0; r < RPT; ++r) a += red[r * 128 + tid];
            const int co = tco * 128 + tid;
            if (p.partial) p.bpartial[(size_t)wg_id * 128 + tid] = a;
            else if (co < p.Cout) atomicAdd(p.dbias + co, a);
        }
    }
    // ---- combine: lanes 0..31 of a register hold 32 consecutive k of one co row (128 B)
    if (p.partial) {                                      // accumulator order, 256-byte coalesced; summed by conv_wgrad_reduce_kernel
        float* slot = p.partial + (size_t)wg_id * (128 * 128);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) slot[(((wave * 2 + mi) * 2 + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = tco * 128 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int k = tk * 128 + wn * 64 + ni * 32 + l31;
                if (co < p.Cout && k < p.Ktot) atomicAdd(p.dw + (size_t)co * p.Ktot + k, acc[mi][ni][r]);
            }
}

template <typename T, bool VECPATH, int BR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * BR * WgLds<T>::ROW];
    wgrad_tile<T, VECPATH, BR>(p, blockIdx.x, gridDim.x, smem);
}

// Every generic weight gradient of a backward pass in ONE launch (round 6): the 1x1 shortcuts, netE's stride-2 layers, the PatchGAN's
// 4x4 layers, the 8x8 maps -- two dozen launches of 30-160 us that each fill the chip badly and each end with a tail -- run side by side:
// workgroup b belongs to job k with first[k] <= b < first[k + 1] (first[] in steps of 8: a job's workgroups keep the XCD placement
// xcd_remap assumes) and is that job's workgroup b - first[k] of nblk[k].  bf16, vector channel counts.
constexpr int WGM_MAX_JOBS = 26;
struct WgMulti { int n; int first[WGM_MAX_JOBS + 1]; int nblk[WGM_MAX_JOBS]; int splits[WGM_MAX_JOBS]; WgradParams j[WGM_MAX_JOBS]; };
static_assert(sizeof(WgMulti) <= 4000, "kernel arguments");

__global__ __launch_bounds__(256, 2) void conv_wgrad_multi_kernel(const WgMulti b) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 32 * WgLds<bf16_t>::ROW];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first[k + 1]) ++k;
    const int blk = (int)blockIdx.x - b.first[k];
    if (blk >= b.nblk[k]) return;                         // (padding to the next multiple of 8)
    wgrad_tile<bf16_t, true, 32>(b.j[k], blk, b.nblk[k], smem);
}

// dw += sum over the pixel splits of a (co, k) tile's partial tiles, dbias += sum over splits and k-tile workgroups of the bias
// partials -- in a FIXED order: the weight gradient is bit-reproducible run to run, which fp32 atomics were not (two
// runs of a trainer drifted apart through Adam's sign flips of near-zero gradients).
// Work item = one 64-element piece (one accumulator register of one wave: 256 contiguous bytes) of one tile; its splits are walked
// by the four waves of a workgroup (wave g takes splits g, g + 4, ...; eight loads in flight), combined through LDS as
// (g0 + g1) + (g2 + g3).  Workgroups stride over the items: at most 2048 of them however many tiles there are (a 128-tile dW
// would otherwise launch 32768 tiny workgroups), and a ONE-tile dW with 512 splits still spreads over 256 items.
__device__ __forceinline__ void wgrad_reduce_body(const WgradParams& p, int splits, int items, int blk, int nblk, float (*red)[64]) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t stride = (size_t)p.tiles_co * p.tiles_k * 16384;   // one split further
    for (int item = blk; item < items; item += nblk) {
        const int tile = item >> 8, frag = item & 255;              // frag = (wave * 4 + mi * 2 + ni) * 16 + r
        const float* src = p.partial + (size_t)tile * 16384 + frag * 64 + lane;
        float a = 0.f;
        int sidx = g;
        for (; sidx + 28 < splits; sidx += 32) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(sidx + 4 * j) * stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += v[j];
        }
        for (; sidx < splits; sidx += 4) a += src[(size_t)sidx * stride];
        red[g][lane] = a;
        __syncthreads();
        if (g == 0) {
            const float tot = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            const int tco = tile / p.tiles_k, tk = tile - tco * p.tiles_k;
            const int r = frag & 15, ni = (frag >> 4) & 1, mi = (frag >> 5) & 1, wave = frag >> 6;
            const int co = tco * 128 + (wave >> 1) * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int k = tk * 128 + (wave & 1) * 64 + ni * 32 + (lane & 31);
            if (co < p.Cout && k < p.Ktot) p.dw[(size_t)co * p.Ktot + k] += tot;
        }
        __syncthreads();
    }
    // bias: the first 2 * tiles_co workgroups take 64 channels each; the (split, k-tile) partials of a channel are walked by
    // the four waves like the splits above
    if (p.dbias && blk < 2 * p.tiles_co) {
        const int tco = blk >> 1, ch = (blk & 1) * 64 + lane, co = tco * 128 + ch;
        const int P = splits * p.tiles_k;                           // partial j = (split j / tiles_k, k-tile j % tiles_k)
        auto at = [&](int j) __attribute__((always_inline)) -> float {
            const int sidx = j / p.tiles_k, t = j - sidx * p.tiles_k;
            return p.bpartial[((size_t)(sidx * p.tiles_co + tco) * p.tiles_k + t) * 128 + ch];
        };
        float a = 0.f;
        int j = g;
        for (; j + 28 < P; j += 32) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = at(j + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) a += v[q];
        }
        for (; j < P; j += 4) a += at(j);
        red[g][lane] = a;
        __syncthreads();
        if (g == 0 && co < p.Cout) p.dbias[co] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const WgradParams p, int splits, int items) {
    __shared__ float red[4][64];
    wgrad_reduce_body(p, splits, items, blockIdx.x, gridDim.x, red);
}
// the reductions of a multi-job launch's split jobs, side by side
struct WgMultiRed { int n; int first[WGM_MAX_JOBS + 1]; int job[WGM_MAX_JOBS]; };
__global__ __launch_bounds__(256) void conv_wgrad_reduce_multi_kernel(const WgMulti b, const WgMultiRed r) {
    __shared__ float red[4][64];
    int k = 0;
    while (k + 1 < r.n && (int)blockIdx.x >= r.first[k + 1]) ++k;
    const int jb = r.job[k];
    wgrad_reduce_body(b.j[jb], b.splits[jb], b.j[jb].tiles_k * b.j[jb].tiles_co * 256, (int)blockIdx.x - r.first[k], r.first[k + 1] - r.first[k], red);
}

// ------------------------------------------------------------------------------------ bf16 LDS-DMA variant
// Same decomposition, operands staged with global_load_lds (no VGPR round trip, no ds_write_b128 -- the
// register-staged kernel spends more LDS cycles on its stores than the MFMAs take).  Rows are 256 B,
// unpadded; LDS-DMA writes lane-linear, so one instruction fills 4 rows and the XOR swizzle
//   physical 16-B chunk = logical chunk ^ ((row & 3) << 2)
// is applied to the SOURCE address.  With it the four rows a ds_read_b64_tr_b16 block touches fall in four
// disjoint 64-B bank ranges (conflict-free).  Out-of-range rows / channels / padding taps read a zero page.
__device__ __attribute__((aligned(16))) const uint32_t g_wg_zero16[4] = {0u, 0u, 0u, 0u};

__global__ __launch_bounds__(256, 4) void conv_wgrad_glds_kernel(const WgradParams p) {
    typedef bf16_t T;
    constexpr int BR = 32, ROW = 256, OP_BYTES = BR * ROW, STAGE = 2 * OP_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // consecutive logical ids (= the (co,k) tiles of ONE pixel slice, which stream the same gy / x rows in
    // lockstep) are placed on one XCD so they share that XCD's L2 instead of each L2 re-fetching the slice
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tk = bid % p.tiles_k; bid /= p.tiles_k;
    const int tco = bid % p.tiles_co; const int split = bid / p.tiles_co;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    if (m_begin >= m_end) return;

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ gg = (const T*)p.gy;
    // DMA mapping: wave w, instruction j fills rows 4*(w + 4*j) .. +3; lane -> row offset lane>>4, physical
    // chunk lane&15 = logical chunk ^ ((row&3)<<2), row&3 == (lane>>4)&3 for both j
    const int rsub = lane >> 4;
    const int c = (lane & 15) ^ ((rsub & 3) << 2);           // logical chunk this lane fetches
    const int co0 = tco * 128 + c * 8;
    const int k0 = tk * 128 + c * 8;
    const int HoWo = p.Ho * p.Wo;
    int t_ci, t_ky, t_kx;
    { const int tap = k0 / p.Cin; t_ci = k0 - tap * p.Cin; t_ky = tap / p.KW; t_kx = tap - t_ky * p.KW; }
    const bool k_valid = k0 < p.Ktot, co_valid = co0 < p.Cout;
    int rn[2], roy[2], rox[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m_begin + 4 * wave + 16 * j + rsub;
        rn[j] = m / HoWo; const int rem = m - rn[j] * HoWo;
        roy[j] = rem / p.Wo; rox[j] = rem - roy[j] * p.Wo;
    }
    auto dma_chunk = [&](int ch, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
        static_for<0, 2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int m = m_begin + ch * BR + 4 * wave + 16 * j + rsub;
            const bool mv = m < m_end;
            const void* gsrc = (mv && co_valid) ? (const void*)(gg + (size_t)m * p.Cout + co0) : (const void*)g_wg_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(base + (4 * wave + 16 * j) * ROW), 16, 0, 0);
            const int iy = roy[j] * p.stride - p.pad + t_ky, ix = rox[j] * p.stride - p.pad + t_kx;
            const bool xv = mv && k_valid && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const void* xsrc = xv ? (const void*)(xg + ((size_t)(rn[j] * p.Hi + iy) * p.Wi + ix) * p.Cin + t_ci)
                                  : (const void*)g_wg_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)xsrc, (lptr_t)(base + OP_BYTES + (4 * wave + 16 * j) * ROW), 16, 0, 0);
            rox[j] += BR;
            while (rox[j] >= p.Wo) { rox[j] -= p.Wo; ++roy[j]; }
            while (roy[j] >= p.Ho) { roy[j] -= p.Ho; ++rn[j]; }
        });
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int hh = lane >> 5, l31 = lane & 31;
    const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g2 = (lane >> 4) & 1;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* Gs = smem + buf * STAGE;
        const char* Xs = Gs + OP_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = 16 * ks + 8 * hh + q;                      // r & 3 == q; (r + 4) & 3 == q
            u32x4_t a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int chunk = (wm * 64 + mi * 32) / 8 + 2 * g2 + (pp >> 1);
                const char* ptr = Gs + r * ROW + ((chunk ^ (q << 2)) << 4) + (pp & 1) * 8;
                const u32x2_t lo = lds_tr16_b64(ptr), hi = lds_tr16_b64(ptr + 4 * ROW);
                a[mi] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int chunk = (wn * 64 + ni * 32) / 8 + 2 * g2 + (pp >> 1);
                const char* ptr = Xs + r * ROW + ((chunk ^ (q << 2)) << 4) + (pp & 1) * 8;
                const u32x2_t lo = lds_tr16_b64(ptr), hi = lds_tr16_b64(ptr + 4 * ROW);
                b[ni] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_t, a[mi]), __builtin_bit_cast(bf16x8_t, b[ni]), acc[mi][ni], 0, 0, 0);
        }
    };
    // bias gradient from the gy tile in LDS: thread t < 128 owns channel t of this tile
    const bool do_bias = p.dbias != nullptr;
    float bsum = 0.f;
    auto bias_from_lds = [&](int buf, int ch) __attribute__((always_inline)) {
        if (do_bias && (ch % p.tiles_k) == tk && tid < 128) {
            const char* Gs = smem + buf * STAGE;
            const int ch = tid >> 3, sub = (tid & 7) * 2;
#pragma unroll 8
            for (int r = 0; r < BR; ++r)
                bsum += bf16_bits_to_f32(*(const unsigned short*)(Gs + r * ROW + ((ch ^ ((r & 3) << 2)) << 4) + sub));
        }
    };

    const int nch = (m_end - m_begin + BR - 1) / BR;
    dma_chunk(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch + 1 < nch; ++ch) {
        const int cur = ch & 1;
        dma_chunk(ch + 1, cur ^ 1);
        compute(cur);
        bias_from_lds(cur, ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    compute((nch - 1) & 1);
    bias_from_lds((nch - 1) & 1, nch - 1);
    if (do_bias && tid < 128) {
        const int co = tco * 128 + tid;
        if (co < p.Cout) atomicAdd(p.dbias + co, bsum);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = tco * 128 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int k = tk * 128 + wn * 64 + ni * 32 + l31;
                if (co < p.Cout && k < p.Ktot) atomicAdd(p.dw + (size_t)co * p.Ktot + k, acc[mi][ni][r]);
            }
}

// Pixel splits of the generic kernel.  Two costs pull against each other (measured per shape, profiles/r01):
//   * long pixel loops want MANY workgroups: c128->256 @256^2 runs 476 / 497 / 558 TFLOP/s at 512 / 1024 / 2048;
//   * every split ends with a 64-KB partial tile (or, without a workspace, one fp32 atomic per dW element on addresses shared by
//     all splits of the tile): a big dW with few pixels (1024->1024 @16^2: 9.4 M elements, 2048 pixels) loses 30 % going from
//     1 split to 4, and a 1-tile dW (the 8-channel label maps) is best at ~512-deep contention, not 1024 or 2048.
// So: aim for S2E_WGRAD_WG (2048) workgroups, keep >= 32 pixel chunks per split (64 when dW is large), never more
// than 512 splits -- unless that leaves the chip under-filled (< 512 workgroups), then allow 16-chunk splits
// (8-chunk splits for a 1-2 tile dW).
static void generic_wgrad_plan(const s2e_conv_desc* d, WgradParams* p, int* splits_out, int target_override = 0) {
    p->Ktot = d->KH * d->KW * d->Cin;
    p->M = d->N * d->Ho * d->Wo;
    p->tiles_k = ceil_div(p->Ktot, 128);
    p->tiles_co = ceil_div(d->Cout, 128);
    const int tiles = p->tiles_k * p->tiles_co;
    // (c128->256 @256^2: 476 / 497 / 558 TFLOP/s at 512 / 1024 / 2048 workgroups; a job of a multi-job launch gets its share of the
    //  launch's workgroups instead: it does not have to fill the chip alone)
    const int target_wg = target_override > 0 ? target_override : 2048;
    int splits = ceil_div(target_wg, tiles);
    int max_splits = p->M / (tiles >= 64 ? 2048 : 1024);
    if (max_splits > 512) max_splits = 512;
    if (target_override <= 0 && (long)tiles * max_splits < 512) {
        const int fill = ceil_div(512, tiles), cap = p->M / (tiles <= 2 ? 256 : 512);
        max_splits = fill < cap ? fill : cap;
    }
    if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p->m_per_split = ceil_div(ceil_div(p->M, splits), 64) * 64;
    *splits_out = ceil_div(p->M, p->m_per_split);
}

// Partial tiles + fixed-order reduction instead of fp32 atomics when a launch has at least this many pixel splits.  Measured
// (profiles/r03, tools/bench_tail.py): with >= 16 splits on few tiles the atomics contend (E's 64->128 @128^2: 77 -> 70 us,
// 128->256 @64^2: 72 -> 62, the 1x1 shortcuts 59 -> 48 / 51 -> 40); with 2-8 splits over 100+ tiles the atomics drain in the
// shadow of the other workgroups' multiplies while the reduction is a second, serial pass (512->512 @16^2: 56 -> 68 us) -- they
// stay.  One split needs neither (a single workgroup owns the tile: its atomics ARE deterministic).
// S2E_WGRAD_PARTIAL=<n>: threshold (default 16); 2 = every split launch (bit-reproducible weight gradients of this kernel,
// ~ +0.4 ms per step); 0 = never.
static int wgrad_partial_min_splits() {
    static const int n = [] { const char* e = getenv("S2E_WGRAD_PARTIAL"); return e ? atoi(e) : 16; }();
    return s2e_deterministic() ? 2 : n;
}
// Partial tiles + fixed-order reduce for this launch?  S2E_DETERMINISTIC: always -- even an unsplit launch adds its k-tile
// workgroups' bias sums with one float atomic each (9 addends for K = 1152: order-dependent in the last bit).
static bool wgrad_use_partial(int splits) {
    if (s2e_deterministic()) return true;
    const int n = wgrad_partial_min_splits();
    return n > 0 && splits >= 2 && splits >= n;
}

extern "C" size_t s2e_conv2d_wgrad_workspace_bytes(int dtype, const s2e_conv_desc* d) {
    if (!d || d->transposed) return 0;
    if ((long)d->N * d->Hi * d->Wi >= (1L << 31) || (long)d->N * d->Ho * d->Wo >= (1L << 31)) return 0;
    const int kind = s2e_small_wgrad_kind(dtype, d);
    if (kind) return s2e_small_wgrad_workspace_bytes(dtype, kind, d);
    if (const int slab_w = s2e_wgrad_patch_plan(dtype, d)) return s2e_wgrad_patch_workspace_bytes(slab_w, d);
    if (const int slab_w = s2e_wgrad_c8_plan(dtype, d)) return s2e_wgrad_c8_workspace_bytes(slab_w, d);
    WgradParams p{};                                       // generic kernel: one partial tile (+ 128 bias sums) per workgroup
    int splits;
    generic_wgrad_plan(d, &p, &splits);
    if (!wgrad_use_partial(splits)) return 0;
    return (size_t)p.tiles_k * p.tiles_co * splits * (128 * 128 + 128) * sizeof(float);
}

extern "C" int s2e_conv2d_wgrad_kernel_kind(int dtype, const s2e_conv_desc* d) {
    if (!d || d->transposed) return S2E_KERNEL_GENERIC;
    if (s2e_small_wgrad_kind(dtype, d)) return S2E_KERNEL_SMALL;
    return (s2e_wgrad_patch_plan(dtype, d) || s2e_wgrad_c8_plan(dtype, d)) ? S2E_KERNEL_PATCH : S2E_KERNEL_GENERIC;
}

// s2e_conv2d_wgrad over the pixels of a device-side list of 16 x 16 rectangles only (label-sparse backward of the SPADE branch): the
// patch-resident kernel with 8 x 16 slabs.  workspace: s2e_conv2d_wgrad_rects_workspace_bytes (0 = not this shape).
extern "C" size_t s2e_conv2d_wgrad_rects_workspace_bytes(int dtype, const s2e_conv_desc* d) {
    if (!d || d->transposed || s2e_small_wgrad_kind(dtype, d) || !s2e_wgrad_patch_plan(dtype, d) || (d->Hi & 15) || (d->Wi & 15)) return 0;
    const size_t b = s2e_wgrad_patch_workspace_bytes(16, d);
    return b ? b : 16;                                    // (non-zero = supported; few-split shapes need no tiles)
}

extern "C" int s2e_conv2d_wgrad_rects(int dtype, const void* x, const void* gy, float* dw, float* dbias, const s2e_conv_desc* d,
                                      const int* rect_list, const int* rect_count, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !gy || !dw || !d || !rect_list || !rect_count) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad_rects: null pointer");
    if (!s2e_conv2d_wgrad_rects_workspace_bytes(dtype, d))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d_wgrad_rects: this shape's kernel takes no rectangle list");
    return s2e_wgrad_patch_launch(16, x, gy, dw, dbias, d, workspace, workspace_bytes, rect_list, rect_count, (hipStream_t)stream);
}

extern "C" int s2e_conv2d_wgrad(int dtype, const void* x, const void* gy, float* dw, float* dbias, const s2e_conv_desc* d,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !gy || !dw || !d) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad: null pointer");
    if (d->transposed) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad: describe the forward conv (transposed=0)");
    if (d->stride != 1 && d->stride != 2) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d_wgrad: stride %d", d->stride);
    if ((long)d->N * d->Hi * d->Wi >= (1L << 31) || (long)d->N * d->Ho * d->Wo >= (1L << 31))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d_wgrad: tensor too large for 32-bit pixel indices");
    if (const int kind = s2e_small_wgrad_kind(dtype, d)) {          // 1-channel heads: dedicated streaming kernels
        SmallConvParams sp{};
        sp.x = x; sp.gy = gy; sp.dw = dw;
        sp.N = d->N; sp.Hi = d->Hi; sp.Wi = d->Wi; sp.Cin = d->Cin; sp.Ho = d->Ho; sp.Wo = d->Wo; sp.Cout = d->Cout;
        sp.KH = d->KH; sp.KW = d->KW; sp.stride = d->stride; sp.pad = d->pad; sp.in_act = d->in_act;
        if (int rc = s2e_small_wgrad_launch(dtype, kind, d, sp, workspace, workspace_bytes, (hipStream_t)stream)) return rc;
        if (dbias) return s2e_colsum(dtype, gy, (long)d->N * d->Ho * d->Wo, d->Cout, dbias, stream);
        return S2E_OK;
    }
    if (const int slab_w = s2e_wgrad_patch_plan(dtype, d))          // big 3x3 stride-1 layers: patch-resident kernel
        return s2e_wgrad_patch_launch(slab_w, x, gy, dw, dbias, d, workspace, workspace_bytes, nullptr, nullptr, (hipStream_t)stream);
    if (const int slab_w = s2e_wgrad_c8_plan(dtype, d))             // 8-channel (label-map) input: B operand built from a 16-B/pixel patch
        if (workspace && workspace_bytes >= s2e_wgrad_c8_workspace_bytes(slab_w, d))
            return s2e_wgrad_c8_launch(slab_w, x, gy, dw, dbias, d, workspace, (hipStream_t)stream);
    WgradParams p{};
    p.x = x; p.gy = gy; p.dw = dw; p.dbias = dbias;
    p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.in_act = d->in_act;
    int splits;
    generic_wgrad_plan(d, &p, &splits);
    const int tiles = p.tiles_k * p.tiles_co;
    hipStream_t st = (hipStream_t)stream;
    const int br = 32;                               // pixels per chunk (64 measured no faster)
    static const bool glds = [] { const char* e = getenv("S2E_WGRAD_GLDS"); return e ? atoi(e) != 0 : false; }();
    const bool glds_path = dtype == S2E_BF16 && glds && d->in_act == S2E_ACT_NONE && d->Cin % 8 == 0 && d->Cout % 8 == 0;
    const int g = tiles * splits;
    // With a workspace the workgroups store their partial tiles and a second kernel adds them up in a fixed order:
    // deterministic, and cheaper than the atomics once a launch has more than a few splits (64 KB written + read per
    // workgroup at HBM rate against 64 KB of float atomics at ~1.3 TB/s chip-wide).
    const size_t need = (size_t)g * (128 * 128 + 128) * sizeof(float);
    if (wgrad_use_partial(splits) && !glds_path && workspace && workspace_bytes >= need) {
        p.partial = (float*)workspace;
        p.bpartial = p.partial + (size_t)g * (128 * 128);
    }
    if (dtype == S2E_BF16) {
        if (d->Cin % 8 == 0 && d->Cout % 8 == 0) {
            if (glds_path) conv_wgrad_glds_kernel<<<g, 256, 0, st>>>(p);
            else if (br == 64) conv_wgrad_kernel<bf16_t, true, 64><<<g, 256, 0, st>>>(p);
            else conv_wgrad_kernel<bf16_t, true, 32><<<g, 256, 0, st>>>(p);
        } else conv_wgrad_kernel<bf16_t, false, 32><<<g, 256, 0, st>>>(p);
    } else if (dtype == S2E_F32) {
        if (d->Cin % 4 == 0 && d->Cout % 4 == 0) conv_wgrad_kernel<float, true, 32><<<g, 256, 0, st>>>(p);
        else conv_wgrad_kernel<float, false, 32><<<g, 256, 0, st>>>(p);
    } else S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("conv_wgrad_kernel");
    if (p.partial) {
        const int items = tiles * 256;
        int rgrid = items < 2048 ? items : 2048;
        if (rgrid < 2 * p.tiles_co) rgrid = 2 * p.tiles_co;
        conv_wgrad_reduce_kernel<<<rgrid, 256, 0, st>>>(p, splits, items);
        S2E_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
    }
    return S2E_OK;
}

// ---- every generic weight gradient of a backward pass in one launch (+ one for the partial-tile reductions): conv_wgrad_multi_kernel
static bool wgrad_multi_ok(int dtype, const s2e_conv_desc* d) {
    if (!d || dtype != S2E_BF16 || d->transposed || (d->stride != 1 && d->stride != 2)) return false;
    if (d->Cin % 8 != 0 || d->Cout % 8 != 0) return false;
    if ((long)d->N * d->Hi * d->Wi >= (1L << 31) || (long)d->N * d->Ho * d->Wo >= (1L << 31)) return false;
    if (s2e_small_wgrad_kind(dtype, d) || s2e_wgrad_patch_plan(dtype, d) || s2e_wgrad_c8_plan(dtype, d)) return false;
    return true;
}
extern "C" int s2e_conv2d_wgrad_multi_supported(int dtype, const s2e_conv_desc* d) { return wgrad_multi_ok(dtype, d) ? 1 : 0; }

// The plan of jobs [base, base + n) of a multi-job launch: a job's share of the launch's workgroups (S2E_WGRAD_MULTI_WGS in all, default
// 6144: 12 per workgroup slot of the chip) goes by its MFMA work -- planned alone (2048 workgroups each) two dozen jobs made 30-40 k
// workgroups and ~1 GB of partial tiles per step.  Fills ps[i] (shape, tiling, m_per_split) and splits[i].
static void wgrad_multi_plan(const s2e_wgrad_multi_job* jobs, const int* idx, int n, WgradParams* ps, int* splits) {
    static const int total_wg = [] { const char* e = getenv("S2E_WGRAD_MULTI_WGS"); return e ? atoi(e) : 6144; }();
    double work[WGM_MAX_JOBS], work_sum = 0.0;
    for (int i = 0; i < n; ++i) {
        const s2e_conv_desc* d = &jobs[idx[i]].d;
        work[i] = (double)ceil_div(d->Cout, 128) * ceil_div(d->KH * d->KW * d->Cin, 128) * ((double)d->N * d->Ho * d->Wo);
        work_sum += work[i];
    }
    for (int i = 0; i < n; ++i) {
        const s2e_conv_desc* d = &jobs[idx[i]].d;
        WgradParams& p = ps[i];
        p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
        p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.in_act = d->in_act;
        generic_wgrad_plan(d, &p, &splits[i], total_wg > 0 ? (int)(total_wg * work[i] / work_sum) + 1 : 0);
    }
}

// the jobs of a multi-job call that stay in the generic kernel (the others: conv_wgrad_flat.hip, conv_c8.hip); returns their count
static int wgrad_multi_generic_jobs(int dtype, const s2e_wgrad_multi_job* jobs, int n_jobs, int* idx, int* flat_idx, int* n_flat, int* c8_idx, int* n_c8) {
    int n = 0, nf = 0, nc = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (s2e_wgrad_flat_kind(dtype, &jobs[i].d)) { if (flat_idx) flat_idx[nf] = i; ++nf; }
        else if (s2e_c8s2_wgrad_ok(dtype, &jobs[i].d)) { if (c8_idx) c8_idx[nc] = i; ++nc; }
        else idx[n++] = i;
    }
    if (n_flat) *n_flat = nf;
    if (n_c8) *n_c8 = nc;
    return n;
}

extern "C" size_t s2e_conv2d_wgrad_multi_workspace_bytes(int dtype, const s2e_wgrad_multi_job* jobs, int n_jobs) {
    size_t total = 0;
    if (!jobs || n_jobs <= 0 || n_jobs > 4096) return 0;
    for (int i = 0; i < n_jobs; ++i) if (!wgrad_multi_ok(dtype, &jobs[i].d)) return 0;
    int idx_all[4096];
    int n_c8w = 0;
    const int n_gen = wgrad_multi_generic_jobs(dtype, jobs, n_jobs, idx_all, nullptr, nullptr, nullptr, &n_c8w);
    if (n_c8w) total += (s2e_c8s2_wgrad_workspace_bytes(n_c8w) + 255) & ~(size_t)255;
    for (int base = 0; base < n_gen; base += WGM_MAX_JOBS) {
        const int n = n_gen - base < WGM_MAX_JOBS ? n_gen - base : WGM_MAX_JOBS;
        WgradParams ps[WGM_MAX_JOBS];
        int splits[WGM_MAX_JOBS];
        wgrad_multi_plan(jobs, idx_all + base, n, ps, splits);
        for (int i = 0; i < n; ++i)
            if (wgrad_use_partial(splits[i]))
                total += ((size_t)ps[i].tiles_k * ps[i].tiles_co * splits[i] * (128 * 128 + 128) * sizeof(float) + 255) & ~(size_t)255;
    }
    return total;
}

extern "C" int s2e_conv2d_wgrad_multi(int dtype, const s2e_wgrad_multi_job* jobs, int n_jobs, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > 4096) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad_multi: no jobs (or more than 4096)");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    size_t ws_left = workspace ? workspace_bytes : 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (!wgrad_multi_ok(dtype, &jobs[i].d)) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_conv2d_wgrad_multi: job %d is not a generic bf16 shape (s2e_conv2d_wgrad_multi_supported)", i);
        if (!jobs[i].x || !jobs[i].gy || !jobs[i].dw) S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad_multi: null pointer in job %d", i);
    }
    int idx_all[4096], flat_idx[4096], c8_idx[4096], n_flat = 0, n_c8 = 0;
    const int n_gen = wgrad_multi_generic_jobs(dtype, jobs, n_jobs, idx_all, flat_idx, &n_flat, c8_idx, &n_c8);
    // the stride-2 / 4x4 / 1x1 layers: the flat-slab patch-resident kernel (atomics into dW; no workspace)
    if (n_flat) if (int rc = s2e_wgrad_flat_launch(jobs, flat_idx, n_flat, st)) return rc;
    // the PatchGAN's 8-channel first layers (conv_c8.hip)
    if (n_c8) {
        const size_t need = (s2e_c8s2_wgrad_workspace_bytes(n_c8) + 255) & ~(size_t)255;
        const bool have = ws && ws_left >= need;
        if (int rc = s2e_c8s2_wgrad_launch(jobs, c8_idx, n_c8, have ? ws : nullptr, have ? need : 0, st)) return rc;
        if (have) { ws += need; ws_left -= need; }
    }
    for (int base = 0; base < n_gen; base += WGM_MAX_JOBS) {
        const int n = n_gen - base < WGM_MAX_JOBS ? n_gen - base : WGM_MAX_JOBS;
        const int* idx = idx_all + base;
        WgMulti b{};
        WgMultiRed r{};
        b.n = n;
        int blocks = 0, rblocks = 0;
        int splits_of[WGM_MAX_JOBS];
        wgrad_multi_plan(jobs, idx, n, b.j, splits_of);
        for (int i = 0; i < n; ++i) {
            const s2e_wgrad_multi_job& J = jobs[idx[i]];
            WgradParams& p = b.j[i];
            p.x = J.x; p.gy = J.gy; p.dw = J.dw; p.dbias = J.dbias;
            const int splits = splits_of[i];
            const int g = p.tiles_k * p.tiles_co * splits;
            const size_t need = (size_t)g * (128 * 128 + 128) * sizeof(float);
            if (wgrad_use_partial(splits) && ws && ws_left >= need) {
                p.partial = (float*)ws;
                p.bpartial = p.partial + (size_t)g * (128 * 128);
                const size_t adv = (need + 255) & ~(size_t)255;
                ws += adv; ws_left = ws_left > adv ? ws_left - adv : 0;
                const int items = p.tiles_k * p.tiles_co * 256;
                int rgrid = items < 2048 ? items : 2048;
                if (rgrid < 2 * p.tiles_co) rgrid = 2 * p.tiles_co;
                r.first[r.n] = rblocks; r.job[r.n] = i; ++r.n;
                rblocks += rgrid;
            }
            b.first[i] = blocks; b.nblk[i] = g; b.splits[i] = splits;
            blocks += (g + 7) & ~7;                       // (a job's first workgroup on a multiple of 8: xcd_remap's placement)
        }
        b.first[n] = blocks;
        r.first[r.n] = rblocks;
        conv_wgrad_multi_kernel<<<blocks, 256, 0, st>>>(b);
        S2E_CHECK_LAUNCH("conv_wgrad_multi_kernel");
        if (r.n) {
            conv_wgrad_reduce_multi_kernel<<<rblocks, 256, 0, st>>>(b, r);
            S2E_CHECK_LAUNCH("conv_wgrad_reduce_multi_kernel");
        }
    }
    return S2E_OK;
}

// which kernel a job of s2e_conv2d_wgrad_multi runs in: 0 = the generic tile kernel, 1..5 = conv_wgrad_flat.hip's kinds (1x1, 3x3 stride 1,
// 3x3 stride 2, 4x4 stride 1, 4x4 stride 2), 6 = conv_c8.hip's 8-channel 4x4 stride-2 kernel, -1 = not a job of that call
extern "C" int s2e_conv2d_wgrad_multi_kind(int dtype, const s2e_conv_desc* d) {
    if (!wgrad_multi_ok(dtype, d)) return -1;
    if (const int k = s2e_wgrad_flat_kind(dtype, d)) return k;
    return s2e_c8s2_wgrad_ok(dtype, d) ? 6 : 0;
}
