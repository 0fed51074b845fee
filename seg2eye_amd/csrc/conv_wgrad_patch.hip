// Patch-resident 3x3 stride-1 weight gradient for gfx950 (bf16, NHWC, fp32 accumulate).
//
//   dW[co][t * Cin + ci] += sum over pixels  gy[pixel][co] * x[pixel + tap t][ci]
//
// The generic kernel (conv_wgrad.hip) gives a workgroup a 128(co) x 128(k) tile of dW and streams, per 32 pixels,
// 8 KB of gy and 8 KB of tap-shifted x for 1 MFLOP.  Like the forward gather this runs against the CU's
// vector-memory path (see conv_patch.hip), at 65 FLOP per byte brought into LDS.  Here a workgroup owns
// 128 co x 64 ci x ALL NINE TAPS (nine 128 x 64 accumulator tiles: 144 registers per lane over 8 waves) and walks
// pixel slabs of 128 pixels (2 x 64, 4 x 32 or 8 x 16): per slab it brings the gy rows (32 KB) and ONE x patch with its halo
// (4 x 66 pixels x 128 B = 33 KB) into LDS; the nine taps are nine shifted views of the patch.  19 MFLOP per 65 KB:
// 290 FLOP per byte, so the loop is paced by the matrix pipe instead of the load path.
//
//   operands   : the contraction runs over pixels, the slow index of both NHWC tensors, so tiles stay [pixel][channel]
//                in LDS and fragments are fetched with ds_read_b64_tr_b16 (as in conv_wgrad.hip).  gy rows are 256 B
//                with the 16-B chunk XORed by (row & 3) << 2; x rows are 128 B with chunk bit 2 XORed by bit 1 of the
//                patch pixel index: any four consecutive rows -- at ANY tap shift -- then sit in four disjoint 64-B
//                bank ranges.  Both swizzles are applied on the source side of the LDS-DMA.
//   pipeline   : two stages of (patch + gy rows) = 2 x 66,560 B; the next slab's 65 one-KiB pieces are issued during
//                the first five of the eight 16-pixel groups of the current slab; one barrier per slab.
//   waves      : 8 = 4 (co blocks of 32) x 2 (ci blocks of 32); per 16-pixel group a wave reads one gy fragment and nine
//                x fragments and issues nine MFMAs, the next fragment requested before each MFMA.
//   bias grad  : sum over pixels of gy = gy^T x ones: one extra MFMA per group against a constant all-ones fragment,
//                shared between the two ci waves (alternate groups) and between the ci-tile workgroups (alternate slabs).
//   combine    : fp32 atomics in 128-B row segments, one per accumulator register, into the caller's dW.
#include "conv_wgrad_patch.h"
#include "wgrad_tr_frag.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) const uint32_t wz_zero16[4] = {0u, 0u, 0u, 0u};

struct WpParams {
    const void* x; const void* gy; float* dw; float* dbias;
    float* ws;                    // per-workgroup partial tiles [wg][9][128][64] (NULL: atomics straight into dw)
    float* bws;                   // S2E_DETERMINISTIC: per-workgroup bias sums [wg][128] (NULL: one atomic per workgroup into dbias)
    int N, H, W, Cin, Cout, Ktot;
    int sx, sy, nslabs;           // slabs per image in x and y; N * sy * sx
    int per_split, tiles_co, tiles_ci;
    // label-sparse launches (TWS = 4 only): the slabs are the two 8 x 16 halves of the 16 x 16 rectangles rect_list[0 .. *rect_count)
    const int* rect_list; const int* rect_count; int splits;
};

template <int TWS>                                // slab width 1 << TWS (64, 32 or 16)
__global__ __launch_bounds__(512, 1) void conv_wgrad_patch_kernel(const WpParams p) {
    typedef bf16_t T;
    constexpr int NW = 8;
    constexpr int XPX = 264;                          // patch capacity: 4 x 66 (6 x 34 = 204 for the narrow slab)
    constexpr int X_BYTES = XPX * 128, G_BYTES = 128 * 256, STAGE = X_BYTES + G_BYTES;
    constexpr int NPI = 9;                            // pieces per thread per slab: 4 gy, 4 x, +1 x (wave 0)
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 1, cib = wave & 1;
    // workgroups of one pixel split (all their co / ci tiles stream the same rows) sit on one XCD
    const int logical_id = xcd_remap(blockIdx.x, gridDim.x);
    int bid = logical_id;
    const int tci = bid % p.tiles_ci; bid /= p.tiles_ci;
    const int tco = bid % p.tiles_co;
    const int split = bid / p.tiles_co;
    int nslabs = p.nslabs, per_split = p.per_split;
    if (p.rect_count) {                               // the count lives on the device (hipGraph replays follow changing label maps)
        nslabs = 2 * *p.rect_count;
        per_split = (nslabs + p.splits - 1) / p.splits;
    }
    const int s0 = split * per_split, s1 = min(nslabs, s0 + per_split);
    // (a workgroup without slabs -- possible with a list -- still writes its zero partial tile: the reduction reads every slot)
    if (s0 >= s1 && !p.rect_count) return;
    constexpr int TW = 1 << TWS, TH = 128 >> TWS, PW = TW + 2, PH = TH + 2;
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ gg = (const T*)p.gy;

    // ---- LDS-DMA pieces of this thread.  i < 4: gy piece q = 8 i + wave, rows (slab pixels) 4q .. 4q+3, 16 lanes per
    // 256-B row; i >= 4: x piece xq = 8 (i - 4) + wave (xq = 32: wave 0 only), patch pixels 8 xq .. +7, 8 lanes per row.
    // The channel a lane fetches is the same for all its gy pieces and for all its x pieces (the swizzle term depends on
    // the lane only); a piece that is never valid (channel tile past Cout, patch pixel past the patch) gets a row
    // offset that fails every bounds check.
    int pdyx[NPI];                                    // pixel offset from the slab origin: (dy << 16) | (dx + 1)
    constexpr int NEVER = 0x4000 << 16;
    const int g_col = tco * 128 + ((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8;
    const int x_col = tci * 64 + ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8;
    static_for<0, NPI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < 4) {
            const int j = 4 * (8 * i + wave) + (lane >> 4);
            pdyx[i] = g_col < p.Cout ? (((j >> TWS) << 16) | ((j & (TW - 1)) + 1)) : NEVER;
        } else {
            const int pp = 8 * (8 * (i - 4) + wave) + (lane >> 3);
            const int py = pp / PW, px = pp - py * PW;
            pdyx[i] = (py < PH && pp < XPX) ? (((py - 1) << 16) | px) : NEVER;
        }
    });
    struct Slab { int n, y0, x0; };
    auto decode = [&](int s) __attribute__((always_inline)) -> Slab {
        Slab q;
        q.x0 = (s % p.sx) << TWS; s /= p.sx;
        q.y0 = (s % p.sy) * TH;
        q.n = s / p.sy;
        return q;
    };
    // with a list: slab s = half (s & 1) of rectangle r = rect_list[s >> 1] of the (H / 16) x (W / 16) rectangle grid
    auto decode_rect = [&](int r, int half) __attribute__((always_inline)) -> Slab {
        Slab q;
        const int tx = p.W >> 4, ty = p.H >> 4;
        q.x0 = (r % tx) << 4; r /= tx;
        q.y0 = ((r % ty) << 4) + 8 * half;
        q.n = r / ty;
        return q;
    };
    auto dma_piece = [&](auto I, const Slab& q, int buf) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if (i == 8 && wave != 0) return;              // wave-uniform
        const int y = q.y0 + (pdyx[i] >> 16), x = q.x0 + (pdyx[i] & 0xffff) - 1;
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const size_t pix = (size_t)(q.n * p.H + y) * p.W + x;
        const void* src;
        char* dst;
        if constexpr (i < 4) {
            src = ok ? (const void*)(gg + pix * p.Cout + g_col) : (const void*)wz_zero16;
            dst = smem + buf * STAGE + X_BYTES + (8 * i + wave) * 1024;
        } else {
            src = ok ? (const void*)(xg + pix * p.Cin + x_col) : (const void*)wz_zero16;
            dst = smem + buf * STAGE + (8 * (i - 4) + wave) * 1024;
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };

    // ---- fragment addressing (see conv_wgrad.hip for the transpose-read lane roles)
    const int hh = lane >> 5, l31 = lane & 31;
    const int i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, g2 = (lane >> 4) & 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    // gy: row 16 g + 8 hh + q4, chunk (cb * 4 + 2 g2 + (pq >> 1)) ^ (q4 << 2); row + 4 keeps row & 3
    const uint32_t a_base = lds0 + X_BYTES + (8 * hh + q4) * 256 + (((cb * 4 + 2 * g2 + (pq >> 1)) ^ (q4 << 2)) << 4) + (pq & 1) * 8;
    // x: patch row r, chunk (cib * 4 + 2 g2 + (pq >> 1)) with bit 2 flipped by bit 1 of r; row + 4 keeps that bit.
    // r = (lane row + tap offset) + R, with R = the 16-pixel group's first patch pixel -- even, so bit 1 of r is the XOR of
    // the two parts' bit 1 and the address splits into a per-lane, per-tap constant and a wave-uniform part:
    //   addr = x_tap[bit 1 of R][t] + (lds0 + stage + (R << 7))                       one v_add_u32 per fragment
    const uint32_t x_const = ((cib * 4 + 2 * g2 + (pq >> 1)) << 4) + (pq & 1) * 8;
    uint32_t x_tap[2][9];                             // [bit 1 of R]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t r = 8 * hh + q4 + (t / 3) * PW + t % 3;
        x_tap[0][t] = ((r << 7) + x_const) ^ ((r & 2u) << 5);
        x_tap[1][t] = x_tap[0][t] ^ 64u;
    }
    f32x16_t acc[9], accb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};           // bf16 1.0 x 8
    asm volatile("" : "+v"(ones));                  // kept in registers (else rebuilt with three moves in front of every use)
    const bool want_bias = p.dbias != nullptr;

    const int* __restrict__ rl = p.rect_list;
    Slab cur = {0, 0, 0};
    int r_ahead = 0;                                  // list launches: the rectangle of slab s + 2, requested a slab ahead of its use
    if (s0 < s1) {
        cur = rl ? decode_rect(rl[s0 >> 1], s0 & 1) : decode(s0);
        if (rl && s0 + 1 < s1) r_ahead = rl[(s0 + 1) >> 1];
        static_for<0, NPI>([&](auto I) { dma_piece(I, cur, 0); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = s0; s < s1; ++s) {
        const int buf = (s - s0) & 1;
        const bool has_next = s + 1 < s1;
        Slab nxt = cur;
        if (has_next) {
            if (rl) { nxt = decode_rect(r_ahead, (s + 1) & 1); if (s + 2 < s1) r_ahead = rl[(s + 2) >> 1]; }
            else nxt = decode(s + 1);
        }
        const bool bias_slab = want_bias && (s % p.tiles_ci) == tci;
        const uint32_t a_stage = a_base + buf * STAGE;
        // 72 MFMA steps per slab (u = 9 g + t).  The x fragment of step u + FD and, at group boundaries, the gy fragment of
        // the next group are requested at step u: one MFMA (32 cycles) of lookahead does not cover an LDS round trip.
        constexpr int FD = 3, NSTEP = 72;
        TrFrag Af[2], Bf[FD + 1];
        uint32_t x_stage = lds0 + buf * STAGE;
        auto x_addr = [&](auto U) __attribute__((always_inline)) -> uint32_t {
            constexpr int u = decltype(U)::value, g = u / 9, t = u % 9;
            constexpr uint32_t R = ((16 * g) >> TWS) * PW + ((16 * g) & (TW - 1));
            return x_tap[(R >> 1) & 1][t] + (x_stage + (R << 7));
        };
        tr_issue<1024>(Af[0], a_stage);
        static_for<0, FD>([&](auto U) { tr_issue<512>(Bf[decltype(U)::value % (FD + 1)], x_addr(U)); });
        static_for<0, NSTEP>([&](auto U) {
            constexpr int u = decltype(U)::value;
            constexpr int g = u / 9, t = u % 9;
            if constexpr (t == 0) {
                if (has_next) {
                    if constexpr (g < 4) { dma_piece(std::integral_constant<int, 2 * g>{}, nxt, buf ^ 1);
                                           dma_piece(std::integral_constant<int, 2 * g + 1>{}, nxt, buf ^ 1); }
                    if constexpr (g == 4) dma_piece(std::integral_constant<int, 8>{}, nxt, buf ^ 1);
                }
                // opaque to the optimiser: otherwise the 72 fragment addresses of a slab are all formed up front and the
                // accumulators spill
                asm volatile("" : "+s"(x_stage));
            }
            if constexpr (u + FD < NSTEP) {
                if constexpr ((u + FD) % 9 == 0) tr_issue<1024>(Af[((u + FD) / 9) & 1], a_stage + ((u + FD) / 9) * 4096);
                tr_issue<512>(Bf[(u + FD) % (FD + 1)], x_addr(std::integral_constant<int, u + FD>{}));
            }
            // reads issued after the ones this step consumes: steps u-FD+1 .. u, two per x fragment, two per gy fragment
            constexpr int keep = [] {
                int k = 0;
                for (int v = u - FD + 1; v <= u; ++v) {
                    if (v + FD >= NSTEP) continue;            // (v < 0: requested ahead of the loop, in the same order)
                    k += 2 + (((v + FD) % 9 == 0) ? 2 : 0);
                }
                return k;
            }();
            TrFrag& A = Af[g & 1];
            TrFrag& B = Bf[u % (FD + 1)];
            if constexpr (t == 0) tr_ready<keep>(A, B);       // the gy fragment was requested before this group's first x
            else tr_ready<keep>(B);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(A), tr_operand(B), acc[t], 0, 0, 0);
            if constexpr (t == 0) {
                if (bias_slab && (g & 1) == cib)
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(A), __builtin_bit_cast(bf16x8_t, ones), accb, 0, 0, 0);
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = nxt;
    }

    // ---- combine.  Lanes 0..31 of a register hold 32 consecutive ci of one (co, tap) row.  With a workspace the tile is
    // written with plain stores ([tap][co][ci] per workgroup) and wgrad_patch_reduce_kernel folds the splits into dw: one
    // workgroup per CU means 256 x 288 KB = 75 MB of partial sums per launch whatever the layer, 58 us as fp32 atomics
    // (1.3 TB/s chip-wide) against ~30 us as a store plus a reduction pass.
    if (p.ws) {
        float* __restrict__ tile = p.ws + (size_t)logical_id * (9 * 128 * 64);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(t * 128 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 64 + cib * 32 + l31] = acc[t][r];
    } else {
        float* __restrict__ dw = p.dw;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = tco * 128 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int k = t * p.Cin + tci * 64 + cib * 32 + l31;
                if (co < p.Cout) atomicAdd(dw + (size_t)co * p.Ktot + k, acc[t][r]);
            }
    }
    // bias gradient: every column of accb holds the same sums, so lanes 0 and 32 carry a wave's 32 channels.  Gathered
    // through LDS into ONE 128-lane atomic per workgroup: two-lane atomics from every wave of every workgroup onto the
    // same 512 bytes cost ~0.1 ms per launch (same-row contention x 128 instructions per workgroup).
    if (want_bias) {                                  // block-uniform; the last slab's barrier is behind every LDS read
        float* red = (float*)smem;                    // [2 ci waves][128 co]
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[cib * 128 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] = accb[r];
        }
        __syncthreads();
        if (tid < 128) {
            const int co = tco * 128 + tid;
            if (p.bws) p.bws[(size_t)logical_id * 128 + tid] = red[tid] + red[128 + tid];      // folded in workgroup order afterwards
            else if (co < p.Cout) atomicAdd(p.dbias + co, red[tid] + red[128 + tid]);
        }
    }
}

// S2E_DETERMINISTIC: dbias[co] += the workgroups' bias sums of co's tile, in logical workgroup order (split, then ci tile)
__global__ __launch_bounds__(128) void wgrad_patch_bias_reduce_kernel(const float* __restrict__ bws, float* __restrict__ dbias, int Cout,
                                                                      int tiles_co, int tiles_ci, int splits) {
    const int tco = blockIdx.x, co = tco * 128 + threadIdx.x;
    if (co >= Cout) return;
    float a = 0.f;
    for (int s = 0; s < splits; ++s)
        for (int tci = 0; tci < tiles_ci; ++tci) a += bws[(size_t)((s * tiles_co + tco) * tiles_ci + tci) * 128 + threadIdx.x];
    dbias[co] += a;
}

// ------------------------------------------------------------------------------------ Cin = 8 (the label-map convs)
// dW[co][t * 8 + ci] += sum over pixels gy[pixel][co] * x[pixel + tap t][ci]  for an 8-channel x (the one-hot label
// map of SPADE's mlp_shared and of the generator's first conv).  K = 72 is less than one 128-wide tile of the generic
// kernel and its x operand is 16 bytes per pixel, so that kernel spends its time staging padding (100 / 43 / 24
// TFLOP/s at 256^2 / 128^2 / 64^2) while the data would stream from HBM in a third of the time.  Here the x operand
// never goes through a staged tile at all: the 3 x 32 columns (tap, ci) of the B fragment are assembled straight from
// a 16-byte-per-pixel patch in LDS with eight ds_read_u16_d16(_hi) per fragment (consecutive pixels are 16 B apart, so
// the eight reads are immediate offsets of one address); column 72 is a constant one and yields the bias gradient.
// gy rows (128 pixels x 128 co, 32 KB per slab) come in by LDS-DMA as in the kernel above; four waves = four 32-co
// blocks; per 16-pixel group a wave issues three MFMAs.  Per-workgroup partial tiles [128][80] go to the workspace and
// are folded by wgrad_c8_reduce_kernel (every workgroup adding into the same 36 KB would serialise the atomics).
struct WcParams {
    const void* x; const void* gy; float* ws;
    int N, H, W, Cout;
    int sx, sy, nslabs, per_split;
    // label-sparse form (8 x 16 slabs only): the slabs are the halves of the 16 x 16 rectangles rect_list[0 .. *rect_count) of a
    // (tiles_y x tiles_x) rectangle grid per sample; `splits` workgroups share them evenly (the count lives on the device)
    const int* rect_list; const int* rect_count; int tiles_x, tiles_y, splits;
};

// two bf16 from two LDS addresses into one register: the low half by ds_read_u16 (zero-extended), the high half by
// ds_read_u16_d16_hi into a second register, OR-ed after the wait.  (A d16_hi read does NOT preserve the other half on
// this target -- with SRAM ECC the destination's unused half is written as zero -- so the pair cannot share a register.)
#define S2E_U16_PAIR(lo, hi, addr, off_lo, off_hi) \
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(off_lo) : "memory"); \
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(off_hi) : "memory")

template <int TWS>
__device__ __forceinline__ void conv_wgrad_c8_body(const WcParams& p, int split, int tco, float* __restrict__ tile, char* smem) {
    typedef bf16_t T;
    constexpr int TW = 1 << TWS, TH = 128 >> TWS, PW = TW + 2, PH = TH + 2;
    constexpr int XPIECES = 5;                        // 64 pixels x 16 B per piece; 320 >= 4 x 66
    constexpr int X_BYTES = XPIECES * 1024, G_BYTES = 128 * 256, STAGE = X_BYTES + G_BYTES;
    static_assert(PH * PW <= XPIECES * 64, "patch capacity");
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave;                              // 32-co block of this wave
    int s0 = split * p.per_split, s1 = min(p.nslabs, s0 + p.per_split);
    if (p.rect_list) {
        const int nsl = 2 * *p.rect_count, per = (nsl + p.splits - 1) / p.splits;
        s0 = split * per; s1 = min(nsl, s0 + per);
    }
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ gg = (const T*)p.gy;
    const int hh = lane >> 5, l31 = lane & 31;

    f32x16_t acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    if (s0 < s1) {
        // ---- pieces of this thread: i < 8: gy piece q = 4 i + wave (rows 4q .. 4q+3 of the slab, 16 lanes per row);
        // i = 8: x piece `wave` (patch pixels 64 wave .. +63, one lane per pixel); i = 9: x piece 4 (wave 0 only)
        constexpr int NEVER = 0x4000 << 16;
        const int g_col = tco * 128 + ((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8;
        int pdyx[10];
        static_for<0, 10>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i < 8) {
                const int j = 4 * (4 * i + wave) + (lane >> 4);
                pdyx[i] = g_col < p.Cout ? (((j >> TWS) << 16) | ((j & (TW - 1)) + 1)) : NEVER;
            } else {
                const int pp = 64 * (i == 8 ? wave : 4) + lane;
                const int py = pp / PW, px = pp - py * PW;
                pdyx[i] = py < PH ? (((py - 1) << 16) | px) : NEVER;
            }
        });
        struct Slab { int n, y0, x0; };
        auto decode = [&](int s) __attribute__((always_inline)) -> Slab {
            Slab q;
            if (p.rect_list) {                         // (TWS = 4: a rectangle is two 8 x 16 slabs)
                int r = p.rect_list[s >> 1];
                q.x0 = (r % p.tiles_x) * 16; r /= p.tiles_x;
                q.y0 = (r % p.tiles_y) * 16 + (s & 1) * 8;
                q.n = r / p.tiles_y;
                return q;
            }
            q.x0 = (s % p.sx) << TWS; s /= p.sx;
            q.y0 = (s % p.sy) * TH;
            q.n = s / p.sy;
            return q;
        };
        auto dma_piece = [&](auto I, const Slab& q, int buf) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if (i == 9 && wave != 0) return;          // wave-uniform
            const int y = q.y0 + (pdyx[i] >> 16), x = q.x0 + (pdyx[i] & 0xffff) - 1;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const size_t pix = (size_t)(q.n * p.H + y) * p.W + x;
            const void* src;
            char* dst;
            if constexpr (i < 8) {
                src = ok ? (const void*)(gg + pix * p.Cout + g_col) : (const void*)wz_zero16;
                dst = smem + buf * STAGE + X_BYTES + (4 * i + wave) * 1024;
            } else {
                src = ok ? (const void*)(xg + pix * 8) : (const void*)wz_zero16;
                dst = smem + buf * STAGE + (i == 8 ? wave : 4) * 1024;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        };

        const int i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, g2 = (lane >> 4) & 1;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
        const uint32_t a_base = lds0 + X_BYTES + (8 * hh + q4) * 256 + (((cb * 4 + 2 * g2 + (pq >> 1)) ^ (q4 << 2)) << 4) + (pq & 1) * 8;
        // column j = 32 c + l31 of the B operand is (tap j >> 3, channel j & 7); this lane holds pixels 8 hh .. 8 hh + 7 of
        // the group: bytes (R + tap offset + 8 hh + i) * 16 + 2 ci of the patch
        uint32_t b_base[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int j = 32 * c + l31, tap = j >> 3;
            b_base[c] = lds0 + (((tap / 3) * PW + tap % 3 + 8 * hh) << 4) + (j & 7) * 2;
        }
        const bool ones_lane = l31 == 8;              // column 72 (third block): the constant one of the bias gradient

        Slab cur = decode(s0);
        static_for<0, 10>([&](auto I) { dma_piece(I, cur, 0); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = s0; s < s1; ++s) {
            const int buf = (s - s0) & 1;
            const bool has_next = s + 1 < s1;
            Slab nxt = cur;
            if (has_next) nxt = decode(s + 1);
            const uint32_t stage = buf * STAGE;
            static_for<0, 8>([&](auto Gq) {
                constexpr int g = decltype(Gq)::value;
                if (has_next) {
                    if constexpr (g < 5) { dma_piece(std::integral_constant<int, 2 * g>{}, nxt, buf ^ 1);
                                           dma_piece(std::integral_constant<int, 2 * g + 1>{}, nxt, buf ^ 1); }
                }
                constexpr uint32_t R16 = (((16 * g) >> TWS) * PW + ((16 * g) & (TW - 1))) << 4;
                TrFrag A;
                tr_issue<1024>(A, a_base + stage + g * 4096);
                uint32_t B[3][4], Bh[3][4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t ad = b_base[c] + stage + R16;
                    S2E_U16_PAIR(B[c][0], Bh[c][0], ad, 0, 16);
                    S2E_U16_PAIR(B[c][1], Bh[c][1], ad, 32, 48);
                    S2E_U16_PAIR(B[c][2], Bh[c][2], ad, 64, 80);
                    S2E_U16_PAIR(B[c][3], Bh[c][3], ad, 96, 112);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(A.lo), "+v"(A.hi), "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][2]), "+v"(B[0][3]),
                               "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[1][2]), "+v"(B[1][3]),
                               "+v"(B[2][0]), "+v"(B[2][1]), "+v"(B[2][2]), "+v"(B[2][3]),
                               "+v"(Bh[0][0]), "+v"(Bh[0][1]), "+v"(Bh[0][2]), "+v"(Bh[0][3]),
                               "+v"(Bh[1][0]), "+v"(Bh[1][1]), "+v"(Bh[1][2]), "+v"(Bh[1][3]),
                               "+v"(Bh[2][0]), "+v"(Bh[2][1]), "+v"(Bh[2][2]), "+v"(Bh[2][3]) :: "memory");
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int v = 0; v < 4; ++v) B[c][v] |= Bh[c][v];
                if (ones_lane) { B[2][0] = 0x3F803F80u; B[2][1] = 0x3F803F80u; B[2][2] = 0x3F803F80u; B[2][3] = 0x3F803F80u; }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(A),
                             __builtin_bit_cast(bf16x8_t, u32x4_t{B[c][0], B[c][1], B[c][2], B[c][3]}), acc[c], 0, 0, 0);
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur = nxt;
        }
    }
    // partial tile of this workgroup: [128 co][80] (72 weight columns, column 72 = bias sum); an idle split writes zeros
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = 32 * c + l31;
            if (col < 80) tile[(cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 80 + col] = acc[c][r];
        }
}

template <int TWS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_c8_kernel(const WcParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * ((5 * 1024) + 128 * 256)];
    conv_wgrad_c8_body<TWS>(p, blockIdx.x, blockIdx.y, p.ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (128 * 80), smem);
}

// The 8-channel weight gradients of MANY layers (a generator's 19 mlp_shared convs: the one-hot label map against d actv) in one
// launch.  The jobs travel BY VALUE in the kernel arguments (no device table to upload, and a hipGraph node keeps them):
// block b belongs to the job whose [first, first + splits) range of blocks holds it.  Cout = 128 for every job.
struct WcJob { const void* x; const void* gy; float* dw; float* dbias; const int* rect_list; const int* rect_count;
               int H, W, sx, sy, nslabs, per_split, first, splits, ws_tile, ncls; };
constexpr int WC_MAX_JOBS = 24;
struct WcBatch { int n, N; WcJob j[WC_MAX_JOBS]; };

template <int TWS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_c8_batch_kernel(const WcBatch b, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) char smem[2 * ((5 * 1024) + 128 * 256)];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.j[k + 1].first) ++k;
    const WcJob& J = b.j[k];
    WcParams p;
    p.x = J.x; p.gy = J.gy; p.ws = ws; p.N = b.N; p.H = J.H; p.W = J.W; p.Cout = 128;
    p.sx = J.sx; p.sy = J.sy; p.nslabs = J.nslabs; p.per_split = J.per_split;
    p.rect_list = J.rect_list; p.rect_count = J.rect_count; p.tiles_x = J.W >> 4; p.tiles_y = J.H >> 4; p.splits = J.splits;
    const int split = blockIdx.x - J.first;
    conv_wgrad_c8_body<TWS>(p, split, 0, ws + (size_t)(J.ws_tile + split) * (128 * 80), smem);
}

// all jobs' partial tiles -> the OIHW gradients: dw[(co * ncls + ci) * 9 + tap] += sum over the job's splits of column tap * 8 + ci
// (ci < ncls), dbias[co] += column 72.  grid.x = jobs x 37 blocks of 256 outputs (128 x 73), grid.y strides the splits.
__global__ __launch_bounds__(256) void wgrad_c8_batch_reduce_kernel(const WcBatch b, const float* __restrict__ ws) {
    const int job = blockIdx.x / 37, idx = (blockIdx.x - job * 37) * 256 + threadIdx.x;
    if (idx >= 128 * 73) return;
    const WcJob& J = b.j[job];
    const int co = idx / 73, j = idx - co * 73;
    const float* src = ws + (size_t)J.ws_tile * (128 * 80) + (size_t)co * 80 + j;
    float a = 0.f;
    for (int s = blockIdx.y; s < J.splits; s += gridDim.y) a += src[(size_t)s * (128 * 80)];
    if (j < 72) {
        const int tap = j >> 3, ci = j & 7;
        if (ci < J.ncls) atomicAdd(J.dw + ((size_t)co * J.ncls + ci) * 9 + tap, a);
    } else if (J.dbias) {
        atomicAdd(J.dbias + co, a);
    }
}

// dw[co][j] += sum over splits of ws[tco][split][co % 128][j], j < 72; dbias[co] += column 72.  blockIdx.y strides the
// splits (a few partial sums per element, combined with atomics) so that the ~9k outputs still fill the chip.
__global__ __launch_bounds__(256) void wgrad_c8_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                             float* __restrict__ dbias, int Cout, int splits) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * 73) return;
    const int co = idx / 73, j = idx - co * 73;
    const float* src = ws + ((size_t)(co >> 7) * splits) * (128 * 80) + (size_t)(co & 127) * 80 + j;
    float a = 0.f;
    for (int s = blockIdx.y; s < splits; s += gridDim.y) a += src[(size_t)s * (128 * 80)];
    if (j < 72) atomicAdd(dw + (size_t)co * 72 + j, a);
    else if (dbias) atomicAdd(dbias + co, a);
}

static int wp_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// dw[co][t * Cin + ci] += sum over the splits of the (co tile, ci tile) of  ws[wg][t][co % 128][ci % 64].  One thread
// per 4 consecutive ci; blockIdx.y takes every gridDim.y-th split (small dW: keeps the chip busy; the few partial sums
// per element that result are combined with atomics).
__global__ __launch_bounds__(256) void wgrad_patch_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout,
                                                                 int Cin, int tiles_co, int tiles_ci, int splits) {
    const int quads = 9 * Cin / 4;                                   // float4 groups per co row
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Cout * quads) return;
    const int co = (int)(idx / quads), kq = (int)(idx - (long)co * quads);
    const int k = kq * 4, t = k / Cin, ci = k - t * Cin;
    const int tco = co >> 7, tci = ci >> 6;
    const size_t in_tile = ((size_t)(t * 128 + (co & 127)) * 64 + (ci & 63));
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
    for (int s = blockIdx.y; s < splits; s += gridDim.y) {
        const int wg = (s * tiles_co + tco) * tiles_ci + tci;        // the logical id the wgrad kernel decoded
        const f32x4_t v = *(const f32x4_t*)(ws + (size_t)wg * (9 * 128 * 64) + in_tile);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
    float* dst = dw + (size_t)co * (9 * Cin) + k;
    if (gridDim.y == 1) {
        f32x4_t o = *(f32x4_t*)dst;
        o[0] += a[0]; o[1] += a[1]; o[2] += a[2]; o[3] += a[3];
        *(f32x4_t*)dst = o;
    } else {
        atomicAdd(dst, a[0]); atomicAdd(dst + 1, a[1]); atomicAdd(dst + 2, a[2]); atomicAdd(dst + 3, a[3]);
    }
}

}  // namespace

// Shapes this kernel takes: bf16, 3x3, stride 1, pad 1, no fused input activation, Cin a multiple of 64, Cout a
// multiple of 8 and >= 64, slabs at least 80 % inside the image, and at least S2E_WGRAD_PATCH (default 256)
// slab x tile work items -- one per CU; swept in round 2: 1024 -> 256 moves the 16^2 [gamma | beta] and 8^2 layers here from the
// generic kernel, generic weight gradient 1.46 -> 1.25 ms per step against +0.05 here; 128 and 64 measure the same.
int s2e_wgrad_patch_plan(int dtype, const s2e_conv_desc* d) {
    static const int min_items = [] { const char* e = getenv("S2E_WGRAD_PATCH"); return e ? atoi(e) : 256; }();
    if (min_items <= 0 || dtype != S2E_BF16) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->in_act != S2E_ACT_NONE || d->transposed) return 0;
    if (d->Ho != d->Hi || d->Wo != d->Wi || d->Cin % 64 != 0 || d->Cout % 8 != 0 || d->Cout < 64) return 0;
    int best = 0; double best_fill = 0.0;
    for (int tw = 64; tw >= 16; tw >>= 1) {
        const int th = 128 / tw;
        const long covered = (long)ceil_div(d->Ho, th) * th * ceil_div(d->Wo, tw) * tw;
        const double fill = (double)d->Ho * d->Wo / (double)covered;
        if (fill > best_fill + 1e-9) { best_fill = fill; best = tw; }
    }
    if (best_fill < 0.8) return 0;
    const long items = (long)d->N * ceil_div(d->Ho, 128 / best) * ceil_div(d->Wo, best) * ceil_div(d->Cout, 128) * (d->Cin / 64);
    return items >= min_items ? best : 0;
}

static void wp_plan(int slab_w, const s2e_conv_desc* d, WpParams& p, int& splits) {
    p.N = d->N; p.H = d->Hi; p.W = d->Wi; p.Cin = d->Cin; p.Cout = d->Cout; p.Ktot = 9 * d->Cin;
    p.sx = ceil_div(d->Wi, slab_w); p.sy = ceil_div(d->Hi, 128 / slab_w); p.nslabs = d->N * p.sy * p.sx;
    p.tiles_co = ceil_div(d->Cout, 128); p.tiles_ci = d->Cin / 64;
    const int tiles = p.tiles_co * p.tiles_ci;
    splits = wp_cu_count() / tiles;                   // one workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > p.nslabs) splits = p.nslabs;
    p.per_split = ceil_div(p.nslabs, splits);
    splits = ceil_div(p.nslabs, p.per_split);
}

// the partial-tile workspace pays off from a few splits on; below that the atomics are few
size_t s2e_wgrad_patch_workspace_bytes(int slab_w, const s2e_conv_desc* d) {
    const bool on = true;                            // (partial tiles through the workspace: 30 us against 58 us as atomics)
    WpParams p{}; int splits;
    wp_plan(slab_w, d, p, splits);
    // (one split: every dW element has a single writer already.  Two or three -- the 1024-channel layers at 16^2: 128 tiles x 2 -- were
    //  combined with atomics until round 4: 75 MB of fp32 atomics at ~1.3 TB/s; through the workspace the family is 2.69 against
    //  2.76-2.83 ms per step, same box, alternating)
    const int min_splits = 2;
    if (!on || splits < min_splits) return s2e_deterministic() ? (size_t)p.tiles_co * p.tiles_ci * splits * 128 * sizeof(float) : 0;
    return (size_t)p.tiles_co * p.tiles_ci * splits * (9 * 128 * 64 + (s2e_deterministic() ? 128 : 0)) * sizeof(float);
}

int s2e_wgrad_patch_launch(int slab_w, const void* x, const void* gy, float* dw, float* dbias, const s2e_conv_desc* d,
                           void* workspace, size_t workspace_bytes, const int* rect_list, const int* rect_count, hipStream_t st) {
    WpParams p{}; int splits;
    wp_plan(slab_w, d, p, splits);
    p.x = x; p.gy = gy; p.dw = dw; p.dbias = dbias;
    p.rect_list = rect_list; p.rect_count = rect_count; p.splits = splits;
    if (rect_list && (slab_w != 16 || (d->Hi & 15) || (d->Wi & 15)))
        S2E_FAIL(S2E_ERR_UNSUPPORTED, "conv_wgrad_patch: a rectangle list needs 16-wide slabs on a map made of 16 x 16 rectangles");
    const size_t need = s2e_wgrad_patch_workspace_bytes(slab_w, d);
    const int nwg = p.tiles_co * p.tiles_ci * splits;
    const bool have = need && workspace && workspace_bytes >= need;
    const bool tiles_ws = have && need >= (size_t)nwg * (9 * 128 * 64) * sizeof(float);
    p.ws = tiles_ws ? (float*)workspace : nullptr;
    // S2E_DETERMINISTIC: the workgroups' bias sums go through the tail of the workspace and are folded in a fixed order
    p.bws = (have && s2e_deterministic() && dbias) ? (float*)workspace + (tiles_ws ? (size_t)nwg * (9 * 128 * 64) : 0) : nullptr;
    if (slab_w == 64) conv_wgrad_patch_kernel<6><<<nwg, 512, 0, st>>>(p);
    else if (slab_w == 32) conv_wgrad_patch_kernel<5><<<nwg, 512, 0, st>>>(p);
    else conv_wgrad_patch_kernel<4><<<nwg, 512, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_wgrad_patch_kernel");
    if (p.ws) {
        const long threads = (long)d->Cout * (9 * d->Cin / 4);
        const int bx = (int)((threads + 255) / 256);
        int by = 1;
        while (!s2e_deterministic() && bx * by < 1024 && by * 2 <= splits) by *= 2;     // (by > 1: partial sums combined with float atomics)
        wgrad_patch_reduce_kernel<<<dim3(bx, by), 256, 0, st>>>(p.ws, dw, d->Cout, d->Cin, p.tiles_co, p.tiles_ci, splits);
        S2E_CHECK_LAUNCH("wgrad_patch_reduce_kernel");
    }
    if (p.bws) {
        wgrad_patch_bias_reduce_kernel<<<p.tiles_co, 128, 0, st>>>(p.bws, dbias, d->Cout, p.tiles_co, p.tiles_ci, splits);
        S2E_CHECK_LAUNCH("wgrad_patch_bias_reduce_kernel");
    }
    return S2E_OK;
}

// ---- Cin = 8: bf16, 3x3, stride 1, pad 1, no fused input activation, Cout a multiple of 128, slabs >= 80 % inside the
// image and at least 256 of them (64^2 and up at batch 8); needs the workspace.
int s2e_wgrad_c8_plan(int dtype, const s2e_conv_desc* d) {
    static const bool on = [] { const char* e = getenv("S2E_WGRAD_C8"); return e ? atoi(e) != 0 : true; }();
    if (!on || dtype != S2E_BF16) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->in_act != S2E_ACT_NONE || d->transposed) return 0;
    if (d->Ho != d->Hi || d->Wo != d->Wi || d->Cin != 8 || d->Cout % 128 != 0) return 0;
    int best = 0; double best_fill = 0.0;
    for (int tw = 64; tw >= 16; tw >>= 1) {
        const int th = 128 / tw;
        const long covered = (long)ceil_div(d->Ho, th) * th * ceil_div(d->Wo, tw) * tw;
        const double fill = (double)d->Ho * d->Wo / (double)covered;
        if (fill > best_fill + 1e-9) { best_fill = fill; best = tw; }
    }
    if (best_fill < 0.8) return 0;
    const long nslabs = (long)d->N * ceil_div(d->Ho, 128 / best) * ceil_div(d->Wo, best);
    return nslabs >= 256 ? best : 0;
}

static void wc_plan(int slab_w, const s2e_conv_desc* d, WcParams& p, int& splits, int& tiles_co) {
    p.N = d->N; p.H = d->Hi; p.W = d->Wi; p.Cout = d->Cout;
    p.sx = ceil_div(d->Wi, slab_w); p.sy = ceil_div(d->Hi, 128 / slab_w); p.nslabs = d->N * p.sy * p.sx;
    tiles_co = d->Cout / 128;
    splits = 2 * wp_cu_count() / tiles_co;            // two 74-KB workgroups per CU
    if (splits < 1) splits = 1;
    if (splits > p.nslabs / 2) splits = p.nslabs / 2 > 0 ? p.nslabs / 2 : 1;
    p.per_split = ceil_div(p.nslabs, splits);
    splits = ceil_div(p.nslabs, p.per_split);
}

size_t s2e_wgrad_c8_workspace_bytes(int slab_w, const s2e_conv_desc* d) {
    WcParams p{}; int splits, tiles_co;
    wc_plan(slab_w, d, p, splits, tiles_co);
    return (size_t)tiles_co * splits * (128 * 80) * sizeof(float);
}

int s2e_wgrad_c8_launch(int slab_w, const void* x, const void* gy, float* dw, float* dbias, const s2e_conv_desc* d,
                        void* workspace, hipStream_t st) {
    WcParams p{}; int splits, tiles_co;
    wc_plan(slab_w, d, p, splits, tiles_co);
    p.x = x; p.gy = gy; p.ws = (float*)workspace;
    const dim3 grid(splits, tiles_co);
    if (slab_w == 64) conv_wgrad_c8_kernel<6><<<grid, 256, 0, st>>>(p);
    else if (slab_w == 32) conv_wgrad_c8_kernel<5><<<grid, 256, 0, st>>>(p);
    else conv_wgrad_c8_kernel<4><<<grid, 256, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_wgrad_c8_kernel");
    const int bx = ceil_div((long)d->Cout * 73, 256);
    int by = 1;
    while (!s2e_deterministic() && bx * by < 512 && by * 2 <= splits) by *= 2;
    wgrad_c8_reduce_kernel<<<dim3(bx, by), 256, 0, st>>>(p.ws, dw, dbias, d->Cout, splits);
    S2E_CHECK_LAUNCH("wgrad_c8_reduce_kernel");
    return S2E_OK;
}

// ---- batched form (see conv_wgrad_c8_batch_kernel).  Host-side job descriptions; every job: bf16, x (N,H,W,8) one-hot map,
// gy (N,H,W,128), 3x3 stride 1 pad 1; dw_oihw fp32 (128, ncls, 3, 3) and dbias fp32 (128) are ACCUMULATED into.
static int wc_best_slab(int H, int W) {
    int best = 0; double best_fill = 0.0;
    for (int tw = 64; tw >= 16; tw >>= 1) {
        const int th = 128 / tw;
        const long covered = (long)ceil_div(H, th) * th * ceil_div(W, tw) * tw;
        const double fill = (double)H * W / (double)covered;
        if (fill > best_fill + 1e-9) { best_fill = fill; best = tw; }
    }
    return best_fill >= 0.8 ? best : 0;
}
extern "C" int s2e_wgrad_c8_batch_supported(int dtype, int H, int W, int cout) {
    return dtype == S2E_BF16 && cout == 128 && wc_best_slab(H, W) != 0;
}
static void wc_job_plan(int N, const s2e_wgrad_c8_job& h, WcJob& J, int& slab_w) {
    slab_w = h.rect_list ? 16 : wc_best_slab(h.H, h.W);         // (a rectangle list: its 16 x 16 rectangles as 8 x 16 slabs)
    J.x = h.x; J.gy = h.gy; J.dw = h.dw_oihw; J.dbias = h.dbias; J.H = h.H; J.W = h.W; J.ncls = h.ncls;
    J.rect_list = h.rect_list; J.rect_count = h.rect_count;
    J.sx = ceil_div(h.W, slab_w); J.sy = ceil_div(h.H, 128 / slab_w); J.nslabs = N * J.sy * J.sx;
    int splits = ceil_div(J.nslabs, 8);               // ~8 slabs (1024 pixels) per workgroup
    if (splits > 512) splits = 512;
    J.per_split = ceil_div(J.nslabs, splits);
    J.splits = ceil_div(J.nslabs, J.per_split);
}
extern "C" size_t s2e_wgrad_c8_batch_workspace_bytes(int N, const s2e_wgrad_c8_job* jobs, int n_jobs) {
    if (!jobs || n_jobs <= 0 || N <= 0) return 0;
    size_t tiles = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (!wc_best_slab(jobs[i].H, jobs[i].W)) return 0;
        WcJob J; int sw;
        wc_job_plan(N, jobs[i], J, sw);
        tiles += J.splits;
    }
    return tiles * (128 * 80) * sizeof(float);
}
extern "C" int s2e_wgrad_c8_batch(int dtype, int N, const s2e_wgrad_c8_job* jobs, int n_jobs, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (!jobs || n_jobs <= 0 || N <= 0 || !workspace) S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_c8_batch: bad argument");
    if (dtype != S2E_BF16) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_wgrad_c8_batch: bf16 only");
    if (workspace_bytes < s2e_wgrad_c8_batch_workspace_bytes(N, jobs, n_jobs)) S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_c8_batch: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    for (int base = 0; base < n_jobs; base += WC_MAX_JOBS) {          // (more jobs than one argument block holds: several rounds)
        const int cnt = n_jobs - base < WC_MAX_JOBS ? n_jobs - base : WC_MAX_JOBS;
        WcBatch all{}; all.n = cnt; all.N = N;
        int slab[WC_MAX_JOBS];
        int ws_tile = 0;
        for (int i = 0; i < cnt; ++i) {
            const s2e_wgrad_c8_job& h = jobs[base + i];
            if (!h.x || !h.gy || !h.dw_oihw || h.ncls <= 0 || h.ncls > 8 || !wc_best_slab(h.H, h.W) || (h.rect_list != nullptr) != (h.rect_count != nullptr)
                || (h.rect_list && ((h.H | h.W) & 15)))
                S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_c8_batch: bad job %d", base + i);
            wc_job_plan(N, h, all.j[i], slab[i]);
            all.j[i].ws_tile = ws_tile;
            ws_tile += all.j[i].splits;
        }
        for (int sw = 64; sw >= 16; sw >>= 1) {                       // one launch per slab shape present
            WcBatch g{}; g.N = N;
            int first = 0;
            for (int i = 0; i < cnt; ++i)
                if (slab[i] == sw) { g.j[g.n] = all.j[i]; g.j[g.n].first = first; first += all.j[i].splits; ++g.n; }
            if (!g.n) continue;
            if (sw == 64) conv_wgrad_c8_batch_kernel<6><<<first, 256, 0, st>>>(g, ws);
            else if (sw == 32) conv_wgrad_c8_batch_kernel<5><<<first, 256, 0, st>>>(g, ws);
            else conv_wgrad_c8_batch_kernel<4><<<first, 256, 0, st>>>(g, ws);
            S2E_CHECK_LAUNCH("conv_wgrad_c8_batch_kernel");
        }
        wgrad_c8_batch_reduce_kernel<<<dim3(cnt * 37, s2e_deterministic() ? 1 : 8), 256, 0, st>>>(all, ws);
        S2E_CHECK_LAUNCH("wgrad_c8_batch_reduce_kernel");
        ws += (size_t)ws_tile * (128 * 80);
    }
    return S2E_OK;
}
