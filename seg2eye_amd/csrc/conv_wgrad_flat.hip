// Patch-resident weight gradient of the stride-2, 4x4 and 1x1 convolutions for gfx950 (bf16, NHWC, fp32 accumulate): "flat" slabs.
//
//   dW[co][t * Cin + ci] += sum over output pixels  gy[pixel][co] * x[stride * pixel + tap t - pad][ci]
//
// conv_wgrad_patch.hip keeps an x patch in LDS and reads the nine taps of a 3x3 stride-1 layer as nine shifted views of it.  Two things
// keep the other layers of the step out of that kernel: a STRIDE (netE's 3x3 and the PatchGAN's 4x4 stride-2 layers: a tap of a stride-2
// conv pairs output pixel X with input pixel 2X + kx, not a shifted view) and RAGGED maps (the PatchGAN's 65^2 / 34^2 / 33^2 / 18^2 / 17^2
// maps fill 16-wide rectangles 45-80 %).  Both go away in a different coordinate system:
//
//   parity planes : the padded input pixel (py, px) = (s Y + ky, s X + kx) lives in plane (py & 1, px & 1) at plane-local position
//                   (py >> 1, px >> 1) = (Y + (ky >> 1), X + (kx >> 1)) (stride 1: one plane, position (Y + ky, X + kx)) -- within a
//                   plane every tap IS a shifted view (conv_plane.hip uses the same planes for the forward / data-gradient kernels);
//   flat index    : planes and the output map share one row pitch LW = Wo + ((K - 1) >> log2 s); the output pixel (Y, X) has the flat
//                   index g = Y * LW + X and tap (ky, kx) reads plane pixel g + o(t), o(t) = (ky >> sh) * LW + (kx >> sh) -- a constant
//                   per tap.  A slab is 64 CONSECUTIVE flat indices of one image, whatever the map's width: the X >= Wo columns
//                   are dummies whose gy rows are loaded as zeros (fill Wo / LW: 92-98 % on the ragged maps).
//
//   workgroup   : 512 threads = 8 waves = 4 (co blocks of 32) x 2 (ci blocks of 32): 128 co x 64 ci x NT taps (3x3: all nine; 4x4: the
//                 eight taps of two kernel rows -- the two halves are separate workgroups; 1x1: one), NT 32x32 accumulators per wave.
//   slab        : 64 gy rows (16 KB) + per plane the 64 + (o_max - o_min) plane pixels its taps touch (3x3 stride 2 at LW = 65:
//                 389 pixels x 128 B = 49 KB; 4x4 stride 2: 4 x 65 pixels; 4x4 stride 1: 64 + LW + 3), double-buffered.
//   operands    : [pixel][channel] in LDS, fragments by ds_read_b64_tr_b16 with conv_wgrad_patch.hip's swizzles (gy: 16-B chunk ^
//                 (row & 3) << 2; x: chunk bit 2 ^ bit 1 of the LDS pixel index); a 16-pixel group starts at a multiple of 16 LDS
//                 pixels, so a fragment address is a per-lane, per-tap constant + a compile-time offset.
//   loads       : LDS-DMA, one 16-byte chunk per lane; which (plane, row, column) an LDS pixel holds is a per-lane constant relative to
//                 the slab origin (divided by LW once per kernel); out-of-map pixels read a zero page.
//   combine     : fp32 atomics in 128-B row segments straight into dW (plain read-modify-writes where a tile has one workgroup); the bias gradient by one extra MFMA per group against ones in
//                 the workgroups of tap group 0 / ci tile 0.
//   launch      : every layer of a backward pass in ONE launch (job table by value in the kernel arguments); a job's share of the
//                 launch's workgroups goes by its MFMA work.
#include "conv_wgrad_flat.h"
#include "wgrad_tr_frag.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) const uint32_t wf_zero16[4] = {0u, 0u, 0u, 0u};

struct WfJob {
    const void* x; const void* gy; float* dw; float* dbias;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Ktot;
    int kind, pad, LW, spi, nslabs;               // spi = slabs per image; nslabs = N * spi
    int tiles_co, tiles_ci, splits, per_split, nblk;
    int flags;                                    // experiments (bit 3: plain read-modify-writes; S2E_WF_NOEPI, timing only: 1 = no combine, 2 = no multiplies, 4 = no loads)
};
constexpr int WF_MAX_JOBS = 24;
struct WfMulti { int n; int first[WF_MAX_JOBS + 1]; WfJob j[WF_MAX_JOBS]; };
static_assert(sizeof(WfMulti) <= 4000, "the job table travels in the kernel arguments");

// patch capacity in pixels and pipeline stages.  Two stages everywhere: the loop runs at (loads) + (fragment reads and multiplies) -- 136 +
// 155 us of the D step's 227-us launch, by ablation -- but it is not load LATENCY that adds: THREE stages for the 4x4 kinds (small
// patches: 3 x 50 KB, the loads of slab s + 2 in flight while slab s is multiplied) measured 224 us.  wf_stages<4>() = 3 selects that form.
template <int K> __host__ __device__ constexpr int wf_xpx() { return K == 4 ? 264 : 392; }
template <int K> __host__ __device__ constexpr int wf_stages() { return 2; }
constexpr int WF_G_BYTES = 64 * 256;
constexpr int wf_max(int a, int b) { return a > b ? a : b; }
constexpr int WF_SMEM = wf_max(wf_stages<4>() * (wf_xpx<4>() * 128 + WF_G_BYTES), wf_stages<3>() * (wf_xpx<3>() * 128 + WF_G_BYTES));

// taps of a workgroup: K = 4 -> kernel rows 2 tg, 2 tg + 1 (8 taps); else all K * K
template <int K> __host__ __device__ constexpr int wf_nt() { return K == 4 ? 8 : K * K; }
template <int K> __host__ __device__ constexpr int wf_ntg() { return K == 4 ? 2 : 1; }
template <int K> __host__ __device__ inline int wf_ky(int tg, int t) { return K == 4 ? 2 * tg + (t >> 2) : t / K; }
template <int K> __host__ __device__ inline int wf_kx(int t) { return K == 4 ? (t & 3) : t % K; }

// the planes of tap group tg at row pitch LW: first LDS pixel, smallest tap offset, pixels held; returns the patch's pixel count
template <int K, int S>
__host__ __device__ inline int wf_planes(int LW, int tg, int* base, int* omin, int* span) {
    constexpr int SH = S - 1, NPL = S * S, NT = wf_nt<K>();
    int omax[NPL];
    for (int P = 0; P < NPL; ++P) { omin[P] = 1 << 20; omax[P] = -1; }
    for (int t = 0; t < NT; ++t) {
        const int ky = wf_ky<K>(tg, t), kx = wf_kx<K>(t);
        const int P = S == 2 ? (ky & 1) * 2 + (kx & 1) : 0;
        const int o = (ky >> SH) * LW + (kx >> SH);
        omin[P] = o < omin[P] ? o : omin[P];
        omax[P] = o > omax[P] ? o : omax[P];
    }
    int acc = 0;
    for (int P = 0; P < NPL; ++P) {
        span[P] = omax[P] < 0 ? 0 : 64 + omax[P] - omin[P];
        base[P] = acc;
        acc += span[P];
    }
    return acc;
}

template <int K, int S>
__device__ __forceinline__ void wf_body(const WfJob& p, const int blk, const int nblk, char* smem) {
    typedef bf16_t T;
    constexpr int SH = S - 1, NPL = S * S, NT = wf_nt<K>(), NTG = wf_ntg<K>();
    constexpr int XPX = wf_xpx<K>(), NS = wf_stages<K>(), PF = NS - 1;
    constexpr int X_BYTES = XPX * 128, STAGE = X_BYTES + WF_G_BYTES;
    static_assert(NS * STAGE <= WF_SMEM && XPX <= 56 * 8, "LDS budget; seven x pieces per wave");
    constexpr int NXP = 7, NPI = 2 + NXP;             // pieces per thread per slab: 2 gy, up to 7 x
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, cib = wave & 1;
    // workgroups of one pixel split (their co / ci tiles and tap groups stream the same rows) sit on one XCD
    int bid = xcd_remap(blk, nblk);
    const int tci = bid % p.tiles_ci; bid /= p.tiles_ci;
    const int tco = bid % p.tiles_co; bid /= p.tiles_co;
    const int tg = bid % NTG;
    const int split = bid / NTG;
    const int s0 = split * p.per_split, s1 = min(p.nslabs, s0 + p.per_split);
    if (s0 >= s1) return;
    const int LW = p.LW;
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ gg = (const T*)p.gy;

    int pbase[NPL], pomin[NPL], pspan[NPL];
    const int npx = wf_planes<K, S>(LW, tg, pbase, pomin, pspan);

    // ---- LDS-DMA pieces of this thread.  i < 2: gy piece 8 i + wave, slab rows 4 q .. 4 q + 3, 16 lanes per 256-B row; i >= 2: x piece
    // 8 (i - 2) + wave, LDS pixels 8 q .. 8 q + 7, 8 lanes per 128-B row.  pk = (row offset) | (column offset << 12) | (plane << 24) of
    // the pixel relative to the slab's first flat index, or NEVER (a channel past Cout, an LDS pixel past the patch).
    constexpr int NEVER = 1 << 30;
    const int g_col = tco * 128 + ((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8;
    const int x_col = tci * 64 + ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8;
    int pk[NPI];
    static_for<0, NPI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < 2) {
            const int j = 4 * (8 * i + wave) + (lane >> 4);
            const int dy = j / LW, dx = j - dy * LW;
            pk[i] = g_col < p.Cout ? (dy | (dx << 12)) : NEVER;
        } else {
            const int pp = 8 * (8 * (i - 2) + wave) + (lane >> 3);
            int P = 0;
#pragma unroll
            for (int q = 1; q < NPL; ++q) if (pp >= pbase[q] && pspan[q] > 0) P = q;
            const int d = pp - pbase[P] + pomin[P];
            const int dy = d / LW, dx = d - dy * LW;
            pk[i] = pp < npx ? (dy | (dx << 12) | (P << 24)) : NEVER;
        }
    });
    struct Slab { int n, ly0, lx0; };
    auto decode = [&](int s) __attribute__((always_inline)) -> Slab {
        Slab q;
        q.n = s / p.spi;
        const int g0 = (s - q.n * p.spi) * 64;
        q.ly0 = g0 / LW;
        q.lx0 = g0 - q.ly0 * LW;
        return q;
    };
    auto dma_piece = [&](auto I, const Slab& q, int buf) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if constexpr (i >= 2) { if (8 * (8 * (i - 2) + wave) >= npx) return; }     // wave-uniform
        const int v = pk[i];
        int lx = q.lx0 + ((v >> 12) & 0xfff), ly = q.ly0 + (v & 0xfff);
        if (lx >= LW) { lx -= LW; ++ly; }
        const void* src;
        char* dst;
        if constexpr (i < 2) {
            const bool ok = v < NEVER && lx < p.Wo && ly < p.Ho;
            src = ok ? (const void*)(gg + ((size_t)(q.n * p.Ho + ly) * p.Wo + lx) * p.Cout + g_col) : (const void*)wf_zero16;
            dst = smem + buf * STAGE + X_BYTES + (8 * i + wave) * 1024;
        } else {
            const int P = (v >> 24) & 3;
            const int iy = (ly << SH) + (S == 2 ? (P >> 1) : 0) - p.pad, ix = (lx << SH) + (S == 2 ? (P & 1) : 0) - p.pad;
            const bool ok = v < NEVER && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            src = ok ? (const void*)(xg + ((size_t)(q.n * p.Hi + iy) * p.Wi + ix) * p.Cin + x_col) : (const void*)wf_zero16;
            dst = smem + buf * STAGE + (8 * (i - 2) + wave) * 1024;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };

    // ---- fragment addressing (lane roles of the transpose read: conv_wgrad.hip)
    const int hh = lane >> 5, l31 = lane & 31;
    const int i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, g2 = (lane >> 4) & 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // gy: row 16 g + 8 hh + q4, chunk (cb * 4 + 2 g2 + (pq >> 1)) ^ (q4 << 2); row + 4 keeps row & 3
    const uint32_t a_base = lds0 + X_BYTES + (8 * hh + q4) * 256 + (((cb * 4 + 2 * g2 + (pq >> 1)) ^ (q4 << 2)) << 4) + (pq & 1) * 8;
    // x: LDS pixel r = 16 g + 8 hh + q4 + (the tap's plane base + offset), chunk (cib * 4 + 2 g2 + (pq >> 1)) with bit 2 flipped by bit 1
    // of r -- 16 g leaves bit 1 alone, so the address is a per-lane, per-tap constant + (16 g << 7); r + 4 keeps bit 1
    const uint32_t x_const = ((cib * 4 + 2 * g2 + (pq >> 1)) << 4) + (pq & 1) * 8;
    uint32_t x_tap[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ky = wf_ky<K>(tg, t), kx = wf_kx<K>(t);
        const int P = S == 2 ? (ky & 1) * 2 + (kx & 1) : 0;
        const uint32_t r = 8 * hh + q4 + pbase[P] + ((ky >> SH) * LW + (kx >> SH)) - pomin[P];
        x_tap[t] = ((r << 7) + x_const) ^ ((r & 2u) << 5);
    }
    f32x16_t acc[NT], accb;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};           // bf16 1.0 x 8
    asm volatile("" : "+v"(ones));
    const bool want_bias = p.dbias != nullptr && tg == 0 && tci == 0;

    // pieces this wave issues per slab (wave-uniform): what may stay in flight behind a slab that has to have landed
    int npw = 2;
#pragma unroll
    for (int i = 0; i < NXP; ++i) npw += 8 * (8 * i + wave) < npx ? 1 : 0;
    auto wait_vm = [&](int n) __attribute__((always_inline)) {       // all but the n youngest loads of this wave are back
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        }
    };
    // slabs s0 .. s0 + PF - 1 ahead of the loop; inside it slab s + PF is requested while slab s is multiplied (its buffer is the one slab
    // s - 1 was read from: free since the barrier at the top of slab s)
    static_for<0, PF>([&](auto D) {
        constexpr int d = decltype(D)::value;
        if (s0 + d < s1) { const Slab q = decode(s0 + d); static_for<0, NPI>([&](auto I) { dma_piece(I, q, d); }); }
    });
    int buf = 0;
    for (int s = s0; s < s1; ++s) {
        // slab s has landed once only the younger slabs' pieces (s + 1 .. s + PF - 1) are outstanding
        wait_vm(PF == 2 && s + 1 < s1 ? npw : 0);
        __syncthreads();
        const bool has_next = s + PF < s1;
        const Slab nxt = decode(has_next ? s + PF : s);
        const int nbuf = buf + PF >= NS ? buf + PF - NS : buf + PF;
        const uint32_t a_stage = a_base + buf * STAGE;
        // 4 NT MFMA steps per slab (u = NT g + t).  The x fragment of step u + FD and, at group boundaries, the gy fragment of the next
        // group are requested at step u.
        constexpr int FD = NT < 3 ? NT : 3, NSTEP = 4 * NT;
        TrFrag Af[2], Bf[FD + 1];
        uint32_t x_stage = lds0 + buf * STAGE;
        auto x_addr = [&](auto U) __attribute__((always_inline)) -> uint32_t {
            constexpr int u = decltype(U)::value, g = u / NT, t = u % NT;
            return x_tap[t] + (x_stage + ((16 * g) << 7));
        };
        const bool no_mm = p.flags & 2;
        if (!no_mm) {
        tr_issue<1024>(Af[0], a_stage);
        static_for<0, FD>([&](auto U) { tr_issue<512>(Bf[decltype(U)::value % (FD + 1)], x_addr(U)); });
        }
        static_for<0, NSTEP>([&](auto U) {
            constexpr int u = decltype(U)::value;
            constexpr int g = u / NT, t = u % NT;
            if constexpr (t == 0) {
                if (has_next && !(p.flags & 4)) {         // the next slab's pieces: all of them in the first half of this one
                    if constexpr (g == 0) static_for<0, 5>([&](auto J) { dma_piece(J, nxt, nbuf); });
                    if constexpr (g == 1) static_for<5, NPI>([&](auto J) { dma_piece(J, nxt, nbuf); });
                }
                asm volatile("" : "+s"(x_stage));     // opaque: else every fragment address of a slab is formed up front (spills)
            }
            if (no_mm) return;
            if constexpr (u + FD < NSTEP) {
                if constexpr ((u + FD) % NT == 0) tr_issue<1024>(Af[((u + FD) / NT) & 1], a_stage + ((u + FD) / NT) * 4096);
                tr_issue<512>(Bf[(u + FD) % (FD + 1)], x_addr(std::integral_constant<int, u + FD>{}));
            }
            // reads issued after the ones this step consumes: steps u-FD+1 .. u, two per x fragment, two per gy fragment
            constexpr int keep = [] {
                int k = 0;
                for (int v = u - FD + 1; v <= u; ++v) {
                    if (v + FD >= NSTEP) continue;            // (v < 0: requested ahead of the loop, in the same order)
                    k += 2 + (((v + FD) % NT == 0) ? 2 : 0);
                }
                return k;
            }();
            TrFrag& A = Af[g & 1];
            TrFrag& B = Bf[u % (FD + 1)];
            if constexpr (t == 0) tr_ready<keep>(A, B);
            else tr_ready<keep>(B);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(A), tr_operand(B), acc[t], 0, 0, 0);
            if constexpr (t == 0) {
                if (want_bias && (g & 1) == cib)
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(A), __builtin_bit_cast(bf16x8_t, ones), accb, 0, 0, 0);
            }
        });
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // (the bias gather below reuses the first stage)

    // ---- combine: lanes 0..31 of a register hold 32 consecutive ci of one (co, tap) row
    if (!(p.flags & 1)) {
        float* __restrict__ dw = p.dw;
        const int tap0 = K == 4 ? 8 * tg : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = tco * 128 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int k = (tap0 + t) * p.Cin + tci * 64 + cib * 32 + l31;
                if (co < p.Cout) {
                    float* q = dw + (size_t)co * p.Ktot + k;
                    if (p.flags & 8) *q += acc[t][r];         // the tile's only workgroup (one split, no second job on this dW)
                    else atomicAdd(q, acc[t][r]);
                }
            }
    }
    if (want_bias) {                                  // block-uniform; the last slab's barrier is behind every LDS read
        float* red = (float*)smem;                    // [2 ci waves][128 co]
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[cib * 128 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] = accb[r];
        }
        __syncthreads();
        if (tid < 128) {
            const int co = tco * 128 + tid;
            if (co < p.Cout) atomicAdd(p.dbias + co, red[tid] + red[128 + tid]);
        }
    }
}

__global__ __launch_bounds__(512, 1) void conv_wgrad_flat_kernel(const WfMulti b) {
    __shared__ __attribute__((aligned(16))) char smem[WF_SMEM];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first[k + 1]) ++k;
    const WfJob& p = b.j[k];
    const int blk = (int)blockIdx.x - b.first[k];
    if (blk >= p.nblk) return;                        // (a job's blocks are padded to a multiple of 8: xcd_remap's placement)
    switch (p.kind) {
        case WF_K1:   wf_body<1, 1>(p, blk, p.nblk, smem); break;
        case WF_K3S1: wf_body<3, 1>(p, blk, p.nblk, smem); break;
        case WF_K3S2: wf_body<3, 2>(p, blk, p.nblk, smem); break;
        case WF_K4S1: wf_body<4, 1>(p, blk, p.nblk, smem); break;
        case WF_K4S2: wf_body<4, 2>(p, blk, p.nblk, smem); break;
        default: break;
    }
}

int wf_mask() {
    static const int m = [] { const char* e = getenv("S2E_WGRAD_FLAT"); return e ? atoi(e) : (1 << WF_K4S1) | (1 << WF_K4S2); }();
    return m;
}

// kind, row pitch and patch size of a layer; WF_NONE when it is not one of this kernel's
int wf_classify(int dtype, const s2e_conv_desc* d, int* LW_out) {
    if (!d || dtype != S2E_BF16 || d->transposed || d->in_act != S2E_ACT_NONE || d->KH != d->KW) return WF_NONE;
    if (d->Cin % 64 != 0 || d->Cout % 8 != 0 || d->N <= 0) return WF_NONE;
    const int K = d->KH, s = d->stride, pad = d->pad;
    int kind = WF_NONE;
    if (K == 1 && s == 1 && pad == 0) kind = WF_K1;
    else if (K == 3 && s == 1 && pad == 1) kind = WF_K3S1;
    else if (K == 3 && s == 2 && pad == 1) kind = WF_K3S2;
    else if (K == 4 && s == 1 && pad == 2) kind = WF_K4S1;
    else if (K == 4 && s == 2 && pad == 2) kind = WF_K4S2;
    if (kind == WF_NONE) return WF_NONE;
    if (d->Ho != (d->Hi + 2 * pad - K) / s + 1 || d->Wo != (d->Wi + 2 * pad - K) / s + 1 || d->Ho <= 0 || d->Wo <= 0) return WF_NONE;
    const int LW = d->Wo + ((K - 1) >> (s - 1));
    if (LW >= 4096 || (long)d->Ho * LW >= (1L << 24)) return WF_NONE;
    if ((long)d->N * d->Hi * d->Wi * d->Cin >= (1L << 31) * 8 || (long)d->N * d->Ho * d->Wo * d->Cout >= (1L << 31) * 8) return WF_NONE;
    int base[4], omin[4], span[4], npx = 0;
    for (int tg = 0; tg < (K == 4 ? 2 : 1); ++tg) {
        int n = 0;
        switch (kind) {
            case WF_K1:   n = wf_planes<1, 1>(LW, tg, base, omin, span); break;
            case WF_K3S1: n = wf_planes<3, 1>(LW, tg, base, omin, span); break;
            case WF_K3S2: n = wf_planes<3, 2>(LW, tg, base, omin, span); break;
            case WF_K4S1: n = wf_planes<4, 1>(LW, tg, base, omin, span); break;
            default:      n = wf_planes<4, 2>(LW, tg, base, omin, span); break;
        }
        npx = n > npx ? n : npx;
    }
    if (npx > (K == 4 ? wf_xpx<4>() : wf_xpx<3>())) return WF_NONE;
    if (LW_out) *LW_out = LW;
    return kind;
}

}  // namespace

int s2e_wgrad_flat_kind(int dtype, const s2e_conv_desc* d) {
    if (s2e_deterministic()) return WF_NONE;          // (float atomics: the generic kernel's fixed-order partial tiles instead)
    const int kind = wf_classify(dtype, d, nullptr);
    return kind != WF_NONE && ((wf_mask() >> kind) & 1) ? kind : WF_NONE;
}

int s2e_wgrad_flat_launch(const s2e_wgrad_multi_job* jobs, const int* idx, int n_all, hipStream_t st) {
    static const int total_wg = [] { const char* e = getenv("S2E_WGRAD_FLAT_WGS"); const int v = e ? atoi(e) : 256; return v > 0 ? v : 256; }();
    static const bool noepi = getenv("S2E_WF_NOEPI") != nullptr;
    for (int base = 0; base < n_all; base += WF_MAX_JOBS) {
        const int n = n_all - base < WF_MAX_JOBS ? n_all - base : WF_MAX_JOBS;
        WfMulti b{};
        b.n = n;
        double cost[WF_MAX_JOBS], cost_sum = 0.0;
        int units[WF_MAX_JOBS];
        for (int i = 0; i < n; ++i) {
            const s2e_wgrad_multi_job& J = jobs[idx[base + i]];
            const s2e_conv_desc* d = &J.d;
            WfJob& p = b.j[i];
            int LW = 0;
            p.kind = wf_classify(S2E_BF16, d, &LW);
            if (p.kind == WF_NONE) S2E_FAIL(S2E_ERR_UNSUPPORTED, "conv_wgrad_flat: job %d is not a shape of this kernel", idx[base + i]);
            if (!J.x || !J.gy || !J.dw) S2E_FAIL(S2E_ERR_ARG, "conv_wgrad_flat: null pointer in job %d", idx[base + i]);
            p.x = J.x; p.gy = J.gy; p.dw = J.dw; p.dbias = J.dbias;
            p.N = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
            p.Ktot = d->KH * d->KW * d->Cin;
            p.pad = d->pad; p.LW = LW;
            p.spi = ceil_div((long)(d->Ho - 1) * LW + d->Wo, 64);
            p.nslabs = d->N * p.spi;
            p.tiles_co = ceil_div(d->Cout, 128); p.tiles_ci = d->Cin / 64;
            const int ntg = d->KH == 4 ? 2 : 1, nt = d->KH == 4 ? 8 : d->KH * d->KH;
            units[i] = p.tiles_co * p.tiles_ci * ntg;
            cost[i] = (double)units[i] * p.nslabs * (nt + 3);
            cost_sum += cost[i];
        }
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            WfJob& p = b.j[i];
            int splits = (int)(total_wg * cost[i] / cost_sum / units[i] + 0.5);
            splits = splits < 1 ? 1 : (splits > p.nslabs ? p.nslabs : splits);
            p.per_split = ceil_div(p.nslabs, splits);
            p.splits = ceil_div(p.nslabs, p.per_split);
            p.nblk = units[i] * p.splits;
            p.flags = noepi ? atoi(getenv("S2E_WF_NOEPI")) & 7 : 0;
            bool shared = false;
            for (int k = 0; k < n_all; ++k) shared = shared || (k != base + i && jobs[idx[k]].dw == jobs[idx[base + i]].dw);
            if (p.splits == 1 && !shared) p.flags |= 8;
            b.first[i] = blocks;
            blocks += (p.nblk + 7) & ~7;
        }
        b.first[n] = blocks;
        conv_wgrad_flat_kernel<<<blocks, 512, 0, st>>>(b);
        S2E_CHECK_LAUNCH("conv_wgrad_flat_kernel");
    }
    return S2E_OK;
}
