// Patch-resident 3x3 stride-1 weight gradients of MANY layers in ONE persistent launch (gfx950, bf16, NHWC, fp32 accumulate).
//
//   for every job j:   dW_j[co][t * Cin + ci] += sum over pixels  gy_j[pixel][co] * x_j[pixel + tap t][ci]      (+ bias gradient)
//
// conv_wgrad_patch.hip runs one layer per launch: one workgroup per CU owns a (128 co x 64 ci x 9 taps) tile of dW and a slice of
// the pixels, so EVERY launch ends with #CUs x 288 KB = 75 MB of partial tiles written, read back by a reduce launch and added
// into dW -- whatever the layer.  Over the 29 such launches of a G step that fixed cost (~45 us each: partials, reduce launch,
// pipeline ramp, tail) was 1.3 of the family's 2.8 ms.  Here the weight gradients a backward pass has queued
// (ops.GradSink.push_wgrad) run as one launch in stream-K fashion:
//
//   unit      = one 128-pixel slab (8 x 16 pixels) of one dW tile of one job; a job has T tiles x ns slabs units.
//   quota     = every workgroup (G of them, one per CU) should get ~ U / G units.  Two kinds of jobs:
//     LONG jobs (ns > Q*, the 256^2 / 128^2 maps): a workgroup streams 65 KB per slab, so the T tiles of a job must walk the SAME
//               slabs at the SAME time on one XCD -- x patches and gy rows then come out of its L2 once per slab instead of once
//               per tile (tile after tile, as in the first version of this kernel, the loop ran at 6.2 instead of 3.6 us per
//               slab: every workgroup on its own HBM / Infinity-Cache stream).  So a long job's slabs are cut into nb = ceil(ns / Q*)
//               equal blocks and workgroup wbase + b T + t runs block b of tile t: consecutive workgroups (one XCD under
//               xcd_remap) in lockstep on the same slabs.
//     SHORT jobs (everything else: many tiles, few slabs): their units, numbered (job, tile, slab), are dealt stream-K fashion to
//               ALL workgroups in proportion to what each has left of its quota -- Q* minus its block for the long jobs'
//               workgroups (a 1024-slab job at Q* = 460 is three blocks of 342: 118 spare each), the whole Q* for the others --
//               and a workgroup's unit range is cut into SEGMENTS at tile boundaries.  (First version: long jobs' workgroups
//               did nothing else; the quota then had to grow to 512 until the blocks of the 4096- and 1024-slab jobs fit.)
//     Q*      = the smallest of 64 candidate quotas in [U / G, 2 U / G) for which the long jobs' blocks fit into G workgroups and
//               the total capacity G Q* covers the work, found by every workgroup for itself from the device-side slab counts
//               (one candidate per lane of wave 0).
//   results   = a segment that covers a whole tile has a single owner: its accumulators are added straight into dW (plain
//               read-modify-write: no atomics, no workspace; a plain store when the caller vouches that dW holds zeros).  Any other
//               segment -- a long job's block when nb > 1, the first and the last segment of a stream-K range -- writes a partial
//               tile ("fragment") to workspace slot 3w / 3w + 1 / 3w + 2.
//   fix-up    = a second, small launch: for every tile with more than one owner, add its fragments in workgroup order into dW.
//               Which workgroups, and which slots, follows from the plan workgroup 0 leaves in the workspace.
//
// So a launch writes at most 3 G fragments in total (typically ~G: 75 MB per BATCH instead of per layer), the many-tile layers at
// small maps (1024 -> 1024 at 16 x 16: 128 tiles of 16 slabs) get one owner per tile, and ramp / tail are paid once.
// Label-sparse jobs (a device-side list of 16 x 16 rectangles, two slabs each) take part with their device-side counts: every
// workgroup derives the unit ranges itself from the counts, so hipGraph replays follow label maps that change between replays.
//
// The slab loop is conv_wgrad_patch_kernel<4>'s (8 x 16 slabs: the smallest halo, 180 patch pixels per 128 outputs), minus
// the LDS-DMA pieces that the 16-wide slab never needs (23 instead of 33 one-KiB x pieces).
#include "wgrad_tr_frag.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) const uint32_t wb_zero16[4] = {0u, 0u, 0u, 0u};

struct WbJob {
    const bf16_t* x; const bf16_t* gy; float* dw; float* dbias;
    const int* rect_list; const int* rect_count;      // label-sparse: the slabs are the halves of rect_list[0 .. *rect_count)
    int N, H, W, Cin, Cout, tiles_co, tiles_ci, nslabs;   // nslabs: N * (H / 8) * (W / 16) (ignored with a list)
    int flags, pad_;                                  // S2E_WGRAD_BATCH_DW_ZERO: dW holds zeros (a single-owner tile is stored, not added)
};
constexpr int WB_MAX_JOBS = 32;
struct WbBatch { int n, G, spare_min, ov; WbJob j[WB_MAX_JOBS]; };      // spare_min, ov: see wb_make_plan
constexpr int WB_TILE = 9 * 128 * 64;                  // floats per (partial) tile: [tap][co % 128][ci % 64]
constexpr int WB_SLOTS = 3;                           // fragment slots per workgroup: a long job's block, the first and the last stream-K segment
constexpr int WB_FIX_LIST = 512;                       // most workgroups one stream-K tile can be shared by (the fix-up's slot list)
constexpr int WB_FIX_PARTS = 6;                        // fix-up blocks per tile: 3072 float4 each (18 432 per tile)
struct WbFix { int first_block[WB_MAX_JOBS + 1]; };

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// The launch's plan (LDS of every workgroup; workgroup 0 copies it behind the fragments for the fix-up launch).
struct WbPlan {
    int w_long, q, u_short, n;                        // workgroups that start with a long job's block; the quota; stream-K units; jobs
    long s_total, s_long;                             // spare capacity of all workgroups (>= u_short); of the long jobs' alone
    int ns[WB_MAX_JOBS];                              // slabs per tile
    int nb[WB_MAX_JOBS];                              // long job: blocks per tile (>= 1); short job: 0
    int wbase[WB_MAX_JOBS];                           // long job: its first workgroup
    int cap[WB_MAX_JOBS];                             // long job: what its workgroups have left for stream-K units (q - block size)
    long sbase[WB_MAX_JOBS];                          // long job: spare capacity of all workgroups before wbase
    int pre[WB_MAX_JOBS + 1];                         // first stream-K unit of job k (long jobs contribute none)
};

// Spare capacity of the workgroups before w: long job k's workgroups have cap[k] each, the ones behind w_long the whole quota.
__device__ __forceinline__ long wb_spare_before(const WbPlan* P, const WbBatch& b, int w) {
    if (w >= P->w_long) return P->s_long + (long)(w - P->w_long) * P->q;
    int k = 0;                                        // (the long jobs' workgroup ranges are NOT in job order: see wb_make_plan)
    while (!(P->nb[k] > 0 && w >= P->wbase[k] && w < P->wbase[k] + P->nb[k] * b.j[k].tiles_co * b.j[k].tiles_ci)) ++k;
    return P->sbase[k] + (long)(w - P->wbase[k]) * P->cap[k];
}
// first stream-K unit of workgroup w: the units are dealt in proportion to the spare capacities (both kernels must agree on this
// to the bit)
__device__ __forceinline__ int wb_first_unit(const WbPlan* P, const WbBatch& b, int w) {
    if (P->s_total <= 0) return 0;
    return (int)((wb_spare_before(P, b, w) * (long)P->u_short) / P->s_total);
}

// all 512 threads of a workgroup; P in LDS
__device__ __forceinline__ void wb_make_plan(const WbBatch& b, int G, WbPlan* P) {
    const int tid = threadIdx.x;
    if (tid < b.n) P->ns[tid] = b.j[tid].rect_count ? 2 * *b.j[tid].rect_count : b.j[tid].nslabs;
    __syncthreads();
    if (tid < 64) {
        // Cost model: a slab is one unit; every SEGMENT (a visit of a tile: pipeline fill, 288-KB write-out by one workgroup, bias
        // fold) costs `ov` units on top -- measured ~4 (a plain store of the tile) to ~7 (read-modify-write) slabs' time.  Without it
        // equal unit counts left the workgroups that walk many 16-slab tiles 15-25 % behind the ones on one long block.  A stream-K
        // tile therefore has ns + ov VIRTUAL units, the first ov of them standing for no slab (whoever gets only those does nothing).
        const int ov = b.ov;
        long U = 0;
        for (int k = 0; k < b.n; ++k) U += (long)(P->ns[k] > 0 ? P->ns[k] + ov : 0) * (b.j[k].tiles_co * b.j[k].tiles_ci);
        const int q0 = (int)((U + G - 1) / G) > ov + 1 ? (int)((U + G - 1) / G) : ov + 2;
        const int qc = q0 + (int)(((long)q0 * tid) >> 6);                 // this lane's candidate quota
        // long jobs (a tile does not fit a quota: ns + ov > qc) take nb = ceil(ns / (qc - ov)) blocks of ceil(ns / nb) slabs per tile,
        // one workgroup each; what such a workgroup's quota has left is filled with the short jobs' units when it is worth a segment
        // of its own (>= spare_min units) -- the others' whole quota is
        long wgs = 0, room = 0, us = 0;
        for (int k = 0; k < b.n; ++k) {
            const int ns = P->ns[k], Tn = b.j[k].tiles_co * b.j[k].tiles_ci;
            if (ns + ov > qc) {
                const int nb = (ns + (qc - ov) - 1) / (qc - ov), spare = qc - ov - (ns + nb - 1) / nb;
                wgs += (long)nb * Tn;
                if (spare >= b.spare_min) room += (long)nb * Tn * spare;
            } else if (ns > 0) {
                us += (long)(ns + ov) * Tn;
            }
        }
        room += ((long)G - wgs) * qc;
        const unsigned long long ok = __ballot(wgs <= (long)G && us <= room);
        const int pick = ok ? __builtin_ctzll(ok) : -1;                   // the smallest feasible quota (none: everything stream-K)
        const int qstar = pick >= 0 ? __shfl(qc, pick, 64) : 0x7fffffff;
        if (tid == 0) {
            int w = 0, a = 0;
            long sp = 0;
            for (int k = 0; k < b.n; ++k) {
                const int ns = P->ns[k], Tn = b.j[k].tiles_co * b.j[k].tiles_ci;
                P->pre[k] = a;
                P->nb[k] = 0; P->wbase[k] = 0; P->cap[k] = 0; P->sbase[k] = 0;
                if (ns > 0 && ns + ov <= qstar) a += (ns + ov) * Tn;
            }
            P->pre[b.n] = a;
            // The long jobs' workgroups, jobs with MORE tiles first: the T workgroups that walk one block together must share an
            // XCD (one L2), and xcd_remap gives an XCD 32 consecutive workgroups -- with the tile counts in descending powers of
            // two every group starts on a multiple of its own size and never straddles two XCDs.  (In job order the dense bench mix
            // at 9 + 3 blocks put the 8-tile groups of the 128^2 layers at 117, 125, ...: half of them straddled, streamed from the
            // Infinity Cache at the slow rate, and the launch took 3.0 instead of 2.5 ms.)
            for (int tt = 64; tt >= 1; tt >>= 1)               // (64: that many tiles or more)
                for (int k = 0; k < b.n; ++k) {
                    const int ns = P->ns[k], Tn = b.j[k].tiles_co * b.j[k].tiles_ci;
                    if (ns + ov <= qstar || Tn < tt || (tt < 64 && Tn >= 2 * tt)) continue;
                    const int nb = (ns + (qstar - ov) - 1) / (qstar - ov), spare = qstar - ov - (ns + nb - 1) / nb;
                    P->nb[k] = nb; P->wbase[k] = w; P->cap[k] = spare >= b.spare_min ? spare : 0; P->sbase[k] = sp;
                    w += nb * Tn;
                    sp += (long)nb * Tn * P->cap[k];
                }
            P->w_long = w; P->u_short = a; P->n = b.n;
            // (no feasible quota: one block of units, dealt evenly)
            P->q = pick >= 0 ? qstar : (a + G - 1) / G + 1;
            P->s_long = sp;
            P->s_total = sp + (long)(G - w) * P->q;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(512, 1) void conv_wgrad_batch_kernel(const WbBatch b, float* __restrict__ ws) {
    typedef bf16_t T;
    constexpr int TWS = 4, TW = 16, PW = TW + 2, NPIX = 10 * PW;              // 8 x 16 slab, 10 x 18 patch
    constexpr int XPIECES = (NPIX + 7) / 8;                                   // 23 one-KiB pieces of 8 patch pixels x 128 B
    constexpr int X_BYTES = XPIECES * 1024, G_BYTES = 128 * 256, STAGE = X_BYTES + G_BYTES;
    constexpr int NPI = 7;                            // pieces per thread per slab: 4 gy, 3 x
    constexpr int RED_OFF = 2 * STAGE, PLAN_OFF = RED_OFF + 1024;
    __shared__ __attribute__((aligned(1024))) char smem[PLAN_OFF + ((sizeof(WbPlan) + 15) & ~15)];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const int cb = wave >> 1, cib = wave & 1;

    // ---- the plan (the sparse jobs' slab counts live on the device: every workgroup works it out for itself)
    WbPlan* P = (WbPlan*)(smem + PLAN_OFF);
    const int G = (int)gridDim.x;
    wb_make_plan(b, G, P);
    const int wg = xcd_remap(blockIdx.x, G);          // consecutive workgroups share an XCD's L2
    if (wg == 0 && tid < (int)(sizeof(WbPlan) / 4)) ((int*)(ws + (size_t)WB_SLOTS * G * WB_TILE))[tid] = ((const int*)P)[tid];
    const int w_long = rfl(P->w_long);
    const bool is_long = wg < w_long;                 // this workgroup starts with a block of a long job ...
    // ... and then, like the others, takes its share of the stream-K units
    int u = 0, u_last = 0;
    if (rfl(P->u_short) > 0) {
        u = rfl(wb_first_unit(P, b, wg));
        u_last = rfl(wb_first_unit(P, b, wg + 1));
    }
    if (!is_long && u >= u_last) return;

    // ---- LDS-DMA pieces of this thread (geometry only: the same for every slab of every job).  i < 4: gy piece q = 8 i + wave,
    // slab pixels 4q .. 4q+3, 16 lanes per 256-B row; i >= 4: x piece xq = 8 (i - 4) + wave, patch pixels 8 xq .. +7, 8 lanes per row.
    int pdyx[NPI];                                    // pixel offset from the slab origin: (dy << 16) | (dx + 1)
    constexpr int NEVER = 0x4000 << 16;
    const int g_lane = ((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8;          // channel within the 128-wide co tile (swizzled)
    const int x_lane = ((lane & 7) ^ (((lane >> 4) & 1) << 2)) * 8;           // channel within the 64-wide ci tile
    static_for<0, NPI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < 4) {
            const int j = 4 * (8 * i + wave) + (lane >> 4);
            pdyx[i] = ((j >> TWS) << 16) | ((j & (TW - 1)) + 1);
        } else {
            const int pp = 8 * (8 * (i - 4) + wave) + (lane >> 3);
            const int py = pp / PW, px = pp - py * PW;
            pdyx[i] = pp < NPIX ? (((py - 1) << 16) | px) : NEVER;
        }
    });

    // ---- fragment addressing (see conv_wgrad.hip for the transpose-read lane roles, conv_wgrad_patch.hip for the swizzles)
    const int hh = lane >> 5, l31 = lane & 31;
    const int i16 = lane & 15, q4 = i16 >> 2, pq = i16 & 3, g2 = (lane >> 4) & 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t a_base = lds0 + X_BYTES + (8 * hh + q4) * 256 + (((cb * 4 + 2 * g2 + (pq >> 1)) ^ (q4 << 2)) << 4) + (pq & 1) * 8;
    const uint32_t x_const = ((cib * 4 + 2 * g2 + (pq >> 1)) << 4) + (pq & 1) * 8;
    // (conv_wgrad_patch.hip keeps a second table for groups whose first patch pixel has bit 1 set -- the same addresses with bit 6
    //  flipped.  Here the flip is applied to the finished address instead: the wave-uniform part (LDS base, stage, R << 7) has its low
    //  seven bits clear, so bit 6 of the sum is bit 6 of the lane part.  Nine registers instead of eighteen: the kernel carries
    //  the segment loop's state on top of conv_wgrad_patch_kernel's and the allocator had begun to spill INTO the slab loop.)
    uint32_t x_tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const uint32_t r = 8 * hh + q4 + (t / 3) * PW + t % 3;
        x_tap[t] = ((r << 7) + x_const) ^ ((r & 2u) << 5);
    }

    struct Slab { int n, y0, x0; };
    int k = 0;
    bool long_seg = is_long, first_short = true;
    while (long_seg || u < u_last) {
        int tile, s0, s1, seg_end = 0, slot;
        bool whole;
        if (long_seg) {
            // block blk of tile `tile` of the long job whose workgroups hold this one
            k = 0;
            while (!(rfl(P->nb[k]) > 0 && wg >= rfl(P->wbase[k]) && wg < rfl(P->wbase[k]) + rfl(P->nb[k]) * b.j[k].tiles_co * b.j[k].tiles_ci)) ++k;
            const int Tn = b.j[k].tiles_co * b.j[k].tiles_ci, nb = rfl(P->nb[k]), ns = rfl(P->ns[k]);
            const int i = wg - rfl(P->wbase[k]), blk = i / Tn;
            tile = i - blk * Tn;
            s0 = (int)(((long)blk * ns) / nb);
            s1 = (int)(((long)(blk + 1) * ns) / nb);
            whole = nb == 1;
            slot = WB_SLOTS * wg;
        } else {
            if (first_short) k = 0;
            while (u >= rfl(P->pre[k + 1])) ++k;      // (long jobs and jobs without slabs hold no stream-K unit: stepped over)
            const int ns = rfl(P->ns[k]), pre = rfl(P->pre[k]), nsv = ns + b.ov;       // (virtual units per tile: wb_make_plan)
            tile = (u - pre) / nsv;
            const int va = (u - pre) - tile * nsv;
            seg_end = min(u_last, pre + (tile + 1) * nsv);
            s0 = max(va - b.ov, 0);
            s1 = max(va + (seg_end - u) - b.ov, 0);
            if (s0 >= s1) { u = seg_end; continue; }  // only overhead units of this tile: nothing to do for it
            whole = s0 == 0 && s1 == ns;              // the tile has no other owner: straight into dW
            slot = WB_SLOTS * wg + (first_short ? 1 : 2);
        }
        const WbJob& J = b.j[k];
        const T* __restrict__ xg = J.x;
        const T* __restrict__ gg = J.gy;
        const int* __restrict__ rl = J.rect_list;
        const int H = J.H, W = J.W, Cin = J.Cin, Cout = J.Cout, tiles_ci = J.tiles_ci;
        const int tci = tile % tiles_ci, tco = tile / tiles_ci;
        const int sx = W >> TWS, sy = H >> 3;
        const int g_col = tco * 128 + g_lane, x_col = tci * 64 + x_lane;
        const bool g_ok = g_col < Cout;
        const bool want_bias = J.dbias != nullptr;

        auto decode = [&](int s) __attribute__((always_inline)) -> Slab {
            Slab q;
            q.x0 = (s % sx) << TWS; s /= sx;
            q.y0 = (s % sy) * 8;
            q.n = s / sy;
            return q;
        };
        // with a list: slab s = half (s & 1) of rectangle r = rect_list[s >> 1] of the (H / 16) x (W / 16) rectangle grid
        auto decode_rect = [&](int r, int half) __attribute__((always_inline)) -> Slab {
            Slab q;
            const int tx = W >> 4, ty = H >> 4;
            q.x0 = (r % tx) << 4; r /= tx;
            q.y0 = ((r % ty) << 4) + 8 * half;
            q.n = r / ty;
            return q;
        };
        auto dma_piece = [&](auto I, const Slab& q, int buf) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            if (i == 6 && wave == 7) return;          // (x piece 23 does not exist; wave-uniform)
            const int y = q.y0 + (pdyx[i] >> 16), x = q.x0 + (pdyx[i] & 0xffff) - 1;
            bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const size_t pix = (size_t)(q.n * H + y) * W + x;
            const void* src;
            char* dst;
            if constexpr (i < 4) {
                ok = ok && g_ok;
                src = ok ? (const void*)(gg + pix * Cout + g_col) : (const void*)wb_zero16;
                dst = smem + buf * STAGE + X_BYTES + (8 * i + wave) * 1024;
            } else {
                src = ok ? (const void*)(xg + pix * Cin + x_col) : (const void*)wb_zero16;
                dst = smem + buf * STAGE + (8 * (i - 4) + wave) * 1024;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        };

        f32x16_t acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // bias gradient = column sums of gy.  The MFMA's A operand of a lane is 8 consecutive pixels of ONE output channel (row lane & 31,
        // pixel half lane >> 5), so the sum is four v_dot2c_f32_bf16 against (1, 1) into ONE register per lane -- conv_wgrad_patch.hip
        // spends an extra MFMA against an all-ones fragment and 16 + 4 registers on it, which this kernel does not have to spare.
        float bsum = 0.f;

        Slab cur = rl ? decode_rect(rl[s0 >> 1], s0 & 1) : decode(s0);
        int r_ahead = 0;                              // list jobs: the rectangle of slab s + 2, requested a slab ahead of its use
        if (rl && s0 + 1 < s1) r_ahead = rl[(s0 + 1) >> 1];
        static_for<0, NPI>([&](auto I) { dma_piece(I, cur, 0); });
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
            const bool bias_slab = want_bias && (s % tiles_ci) == tci;
            const uint32_t a_stage = a_base + buf * STAGE;
            // 72 MFMA steps per slab (u = 9 g + t), the x fragment of step + FD and the gy fragment of the next group requested
            // FD steps ahead (conv_wgrad_patch.hip)
            constexpr int FD = 3, NSTEP = 72;
            TrFrag Af[2], Bf[FD + 1];
            uint32_t x_stage = (uint32_t)rfl((int)(lds0 + buf * STAGE));
            auto x_addr = [&](auto Uq) __attribute__((always_inline)) -> uint32_t {
                constexpr int uu = decltype(Uq)::value, g = uu / 9, t = uu % 9;
                constexpr uint32_t R = g * PW;          // group g = slab row g: its first patch pixel
                const uint32_t ad = x_tap[t] + (x_stage + (R << 7));
                return ((R >> 1) & 1) ? (ad ^ 64u) : ad;
            };
            tr_issue<1024>(Af[0], a_stage);
            static_for<0, FD>([&](auto Uq) { tr_issue<512>(Bf[decltype(Uq)::value % (FD + 1)], x_addr(Uq)); });
            static_for<0, NSTEP>([&](auto Uq) {
                constexpr int uu = decltype(Uq)::value;
                constexpr int g = uu / 9, t = uu % 9;
                if constexpr (t == 0) {
                    if (has_next) {
                        // the next slab's 7 pieces behind the first MFMA of groups 0 .. 3
                        if constexpr (g < 3) { dma_piece(std::integral_constant<int, 2 * g>{}, nxt, buf ^ 1);
                                               dma_piece(std::integral_constant<int, 2 * g + 1>{}, nxt, buf ^ 1); }
                        if constexpr (g == 3) dma_piece(std::integral_constant<int, 6>{}, nxt, buf ^ 1);
                    }
                    asm volatile("" : "+s"(x_stage));   // (else all 72 fragment addresses are formed up front and the accumulators spill)
                }
                if constexpr (uu + FD < NSTEP) {
                    if constexpr ((uu + FD) % 9 == 0) tr_issue<1024>(Af[((uu + FD) / 9) & 1], a_stage + ((uu + FD) / 9) * 4096);
                    tr_issue<512>(Bf[(uu + FD) % (FD + 1)], x_addr(std::integral_constant<int, uu + FD>{}));
                }
                constexpr int keep = [] {
                    int kk = 0;
                    for (int v = uu - FD + 1; v <= uu; ++v) {
                        if (v + FD >= NSTEP) continue;
                        kk += 2 + (((v + FD) % 9 == 0) ? 2 : 0);
                    }
                    return kk;
                }();
                TrFrag& A = Af[g & 1];
                TrFrag& B = Bf[uu % (FD + 1)];
                if constexpr (t == 0) tr_ready<keep>(A, B);
                else tr_ready<keep>(B);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(A), tr_operand(B), acc[t], 0, 0, 0);
                if constexpr (t == 0) {
                    if (bias_slab && (g & 1) == cib) {
                        const bf16x2_t one2 = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
                        // (by value and indexed with []: __builtin_bit_cast of .x / .y of an ext-vector behind a REFERENCE reads element 0
                        //  for both on this compiler -- DESIGN 3.8; found here as a bias gradient that summed half the pixels twice)
                        const u32x2_t alo = A.lo, ahi = A.hi;
                        const uint32_t a0 = alo[0], a1 = alo[1], a2 = ahi[0], a3 = ahi[1];
                        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a0), one2, bsum, false);
                        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a1), one2, bsum, false);
                        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a2), one2, bsum, false);
                        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a3), one2, bsum, false);
                    }
                }
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur = nxt;
        }

        // ---- the segment's result.  Lanes 0..31 of a register hold 32 consecutive ci of one (co, tap) row.
        // (the lane parts of the addresses are made opaque HERE: as loop invariants of the segment loop the 144 offsets of either form were
        //  formed ahead of it and spilled)
        if (whole) {
            // tap by tap, with a compiler barrier between them: hoisted together the 144 loads of the read-modify-write need 144 more
            // registers than the kernel has
            const int co0 = tco * 128 + cb * 32 + 4 * hh;
            uint32_t lane_off = (uint32_t)((cb * 32 + 4 * hh) * (9 * Cin) + cib * 32 + l31);
            asm volatile("" : "+v"(lane_off));
            float* dwt = J.dw + (size_t)tco * 128 * (9 * Cin) + tci * 64;        // uniform
            const bool dw_zero = (J.flags & S2E_WGRAD_BATCH_DW_ZERO) != 0;      // (uniform) nothing to add to: 4 instead of 8 bytes per element
            if (dw_zero) {
                static_for<0, 9>([&](auto Tq) {
                    constexpr int t = decltype(Tq)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        if (co0 + ro < Cout) dwt[lane_off + (uint32_t)(ro * 9 * Cin + t * Cin)] = acc[t][r];
                    }
                });
            } else {
                // read-modify-write, tap by tap with a compiler barrier between them: hoisted together the 144 loads need 144 more
                // registers than the kernel has.  (Rare inside a trainer step -- the arenas are fresh; ~25 us per tile for one
                // workgroup.  A two-tap pipeline of the loads was built and spilled INTO the slab loop: not kept.)
                static_for<0, 9>([&](auto Tq) {
                    constexpr int t = decltype(Tq)::value;
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        old[r] = (co0 + ro < Cout) ? dwt[lane_off + (uint32_t)(ro * 9 * Cin + t * Cin)] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        if (co0 + ro < Cout) dwt[lane_off + (uint32_t)(ro * 9 * Cin + t * Cin)] = old[r] + acc[t][r];
                    }
                    asm volatile("" ::: "memory");
                });
            }
        } else {
            float* __restrict__ tl = ws + (size_t)slot * WB_TILE;      // uniform
            uint32_t lane_off = (uint32_t)((cb * 32 + 4 * hh) * 64 + cib * 32 + l31);
            asm volatile("" : "+v"(lane_off));
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tl[lane_off + (uint32_t)((t * 128 + (r & 3) + 8 * (r >> 2)) * 64)] = acc[t][r];
        }
        // bias gradient of the segment's slabs: lanes l and l + 32 hold the two pixel halves of channel cb * 32 + (l & 31); folded, then
        // gathered through LDS into one 128-lane atomic per segment
        if (want_bias) {                              // block-uniform
            float* red = (float*)(smem + RED_OFF);    // [2 ci waves][128 co]
            const float tot = bsum + __shfl_xor(bsum, 32, 64);
            if (lane < 32) red[cib * 128 + cb * 32 + lane] = tot;
            __syncthreads();
            if (tid < 128) {
                const int co = tco * 128 + tid;
                if (co < Cout) atomicAdd(J.dbias + co, red[tid] + red[128 + tid]);
            }
            __syncthreads();                          // (red is rewritten by the next segment)
        }
        if (long_seg) long_seg = false;
        else { u = seg_end; first_short = false; }
    }
}

// The tiles with more than one owner: dW += their fragments, in workgroup order.  Block = (job, tile, part); the plan is the one
// workgroup 0 of the main launch left behind the fragment slots.
__global__ __launch_bounds__(256) void wgrad_batch_fixup_kernel(const WbBatch b, const WbFix f, const float* __restrict__ ws) {
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= f.first_block[k + 1]) ++k;
    const WbJob& J = b.j[k];
    const int rel = (int)blockIdx.x - f.first_block[k];
    const int tile = rel / WB_FIX_PARTS, part = rel - tile * WB_FIX_PARTS;
    const int G = b.G;
    // the plan into LDS first (one coalesced read): the owner search below evaluates the unit map a few dozen times, and from
    // global memory every evaluation was a chain of dependent loads -- the fix-up took longer than the fragments it adds
    __shared__ WbPlan Ps;
    __shared__ int s_slots[WB_FIX_LIST], s_nslots;
    {
        const int* src = (const int*)(ws + (size_t)WB_SLOTS * G * WB_TILE);
        for (int i = threadIdx.x; i < (int)(sizeof(WbPlan) / 4); i += blockDim.x) ((int*)&Ps)[i] = src[i];
    }
    __syncthreads();
    const WbPlan* P = &Ps;
    const int ns = P->ns[k], nb = P->nb[k];
    if (ns == 0 || nb == 1) return;                   // nothing ran / a long job whose tiles have one owner each (block-uniform)
    const int Tn = J.tiles_co * J.tiles_ci;
    // the fragment slots of this tile: long job -- 3 (wbase + blk Tn + tile) for every block; stream-K -- the workgroups w whose
    // unit ranges meet the tile's units [ua, ue): slot 3 w + 1 when the tile holds w's first unit (its first segment), 3 w + 2 else
    if (threadIdx.x == 0) {
        int n = 0;
        if (nb == 0) {
            const int ov = b.ov, nsv = ns + ov;
            const int ua = P->pre[k] + tile * nsv, ue = ua + nsv;        // the tile's VIRTUAL units (wb_make_plan)
            auto owner = [&](int u) {                 // the last workgroup whose first unit is <= u (binary search: first units are monotone)
                int lo = 0, hi = G - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (wb_first_unit(P, b, mid) <= u) lo = mid; else hi = mid - 1;
                }
                return lo;
            };
            const int w_lo = owner(ua), w_hi = owner(ue - 1);
            for (int w = w_lo; w <= w_hi && n < WB_FIX_LIST; ++w) {
                const int u0 = wb_first_unit(P, b, w), u1 = wb_first_unit(P, b, w + 1);
                const int va = max(u0, ua) - ua, vb = min(u1, ue) - ua;
                if (max(vb - ov, 0) > max(va - ov, 0))                   // (it holds slabs of the tile, not just overhead units)
                    s_slots[n++] = WB_SLOTS * w + (u0 >= ua ? 1 : 2);
            }
            if (n == 1) n = 0;                        // one owner of all its slabs: it added the tile itself
        } else {
            n = nb;                                   // (a long job's slots follow from the block index: no list)
        }
        s_nslots = n;
    }
    __syncthreads();
    const int nslots = s_nslots;
    if (nslots == 0) return;
    const int tci = tile % J.tiles_ci, tco = tile / J.tiles_ci;
    const int Ktot = 9 * J.Cin;
    const bool dw_zero = (J.flags & S2E_WGRAD_BATCH_DW_ZERO) != 0;        // nothing in dW yet: stored, not added
#pragma unroll 1
    for (int i = 0; i < 12; ++i) {
        const int idx4 = part * 3072 + i * 256 + (int)threadIdx.x;        // float4 index in the tile: [tap][co % 128][ci % 64 / 4]
        const int c4 = idx4 & 15, row = (idx4 >> 4) & 127, t = idx4 >> 11;
        const int co = tco * 128 + row;
        if (co >= J.Cout) continue;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nslots; ++q) {
            const int slot = nb > 0 ? WB_SLOTS * (P->wbase[k] + q * Tn + tile) : s_slots[q];
            const f32x4_t v = *(const f32x4_t*)(ws + (size_t)slot * WB_TILE + (size_t)idx4 * 4);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
        float* dst = J.dw + (size_t)co * Ktot + t * J.Cin + tci * 64 + c4 * 4;
        if (!dw_zero) {
            const f32x4_t o = *(f32x4_t*)dst;
            a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3];
        }
        *(f32x4_t*)dst = a;
    }
}

// experiment switch S2E_WGRAD_BATCH_SPARE: the least spare capacity (units) of a long job's workgroup that is filled with stream-K
// units; a huge value = long jobs' workgroups do nothing else (the round's first version).  Measured, same box, the replayed step /
// the dense 28-layer mix of tools/check_wgrad_batch.py --bench (spare, ov): (never, 0) 17.39 ms / 2.22 ms; (48, 0) -- equal unit
// counts, no overhead term -- 17.44 / 2.78; (48, 4) 17.27 / 2.36; (24, 4) 17.26 / 2.19; (24, 6) 17.27.  Default 24 with ov = 4.
int wb_spare_min() {
    static const int v = [] { const char* e = getenv("S2E_WGRAD_BATCH_SPARE"); return e && atoi(e) > 0 ? atoi(e) : 24; }();
    return v;
}

// experiment switch S2E_WGRAD_BATCH_OV: what a segment costs beside its slabs, in slabs (wb_make_plan's cost model)
int wb_overhead() {
    static const int v = [] { const char* e = getenv("S2E_WGRAD_BATCH_OV"); return e && atoi(e) >= 0 ? atoi(e) : 4; }();
    return v;
}

int wb_workgroups() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        const char* e = getenv("S2E_WGRAD_BATCH_WGS");   // experiment switch: workgroups of the batched launch (default: one per CU)
        if (e && atoi(e) > 0) v = atoi(e);
        if (v > WB_FIX_LIST) v = WB_FIX_LIST;            // (the fix-up's owner list holds WB_FIX_LIST workgroups per tile: ADVICE r5)
        return v;
    }();
    return n;
}

}  // namespace

// bf16; 3x3, stride 1, pad 1, no fused input activation; a map made of 8 x 16 slabs; Cin a multiple of 64; Cout a multiple of 8, >= 64
extern "C" int s2e_wgrad_batch_supported(int dtype, int N, int H, int W, int Cin, int Cout) {
    if (dtype != S2E_BF16 || N <= 0 || H <= 0 || W <= 0) return 0;
    if ((H & 7) || (W & 15) || Cin % 64 != 0 || Cout % 8 != 0 || Cout < 64) return 0;
    if ((long)N * H * W >= (1L << 31) / 2048) return 0;       // 32-bit pixel indices x channels stay in range of the size_t products
    return 1;
}

extern "C" size_t s2e_wgrad_batch_workspace_bytes(void) {
    return (size_t)WB_SLOTS * wb_workgroups() * WB_TILE * sizeof(float) + ((sizeof(WbPlan) + 255) & ~(size_t)255);      // fragments + the plan
}

extern "C" int s2e_wgrad_batch(int dtype, const s2e_wgrad_batch_job* jobs, int n_jobs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!jobs || n_jobs <= 0 || !workspace) S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_batch: bad argument");
    if (workspace_bytes < s2e_wgrad_batch_workspace_bytes()) S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_batch: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n_jobs; base += WB_MAX_JOBS) {            // (more jobs than one argument block holds: several launches)
        const int cnt = n_jobs - base < WB_MAX_JOBS ? n_jobs - base : WB_MAX_JOBS;
        WbBatch B{};
        WbFix F{};
        B.n = cnt; B.G = wb_workgroups(); B.spare_min = wb_spare_min(); B.ov = wb_overhead();
        int blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const s2e_wgrad_batch_job& h = jobs[base + i];
            if (!h.x || !h.gy || !h.dw || !s2e_wgrad_batch_supported(dtype, h.N, h.H, h.W, h.Cin, h.Cout) ||
                ((h.rect_list != nullptr) != (h.rect_count != nullptr)) || (h.rect_list && ((h.H & 15) || (h.W & 15))))
                S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_batch: bad job %d", base + i);
            for (int q = 0; q < i; ++q)
                if (B.j[q].dw == h.dw) S2E_FAIL(S2E_ERR_ARG, "s2e_wgrad_batch: jobs %d and %d accumulate into the same dW (single-owner tiles are "
                                                "added without atomics): queue them in separate calls", base + q, base + i);
            WbJob& J = B.j[i];
            J.x = (const bf16_t*)h.x; J.gy = (const bf16_t*)h.gy; J.dw = h.dw; J.dbias = h.dbias;
            J.rect_list = h.rect_list; J.rect_count = h.rect_count;
            J.N = h.N; J.H = h.H; J.W = h.W; J.Cin = h.Cin; J.Cout = h.Cout; J.flags = h.flags;
            J.tiles_co = ceil_div(h.Cout, 128); J.tiles_ci = h.Cin / 64;
            J.nslabs = h.N * (h.H >> 3) * (h.W >> 4);
            F.first_block[i] = blocks;
            blocks += J.tiles_co * J.tiles_ci * WB_FIX_PARTS;
        }
        F.first_block[cnt] = blocks;
        conv_wgrad_batch_kernel<<<B.G, 512, 0, st>>>(B, (float*)workspace);
        S2E_CHECK_LAUNCH("conv_wgrad_batch_kernel");
        wgrad_batch_fixup_kernel<<<blocks, 256, 0, st>>>(B, F, (const float*)workspace);
        S2E_CHECK_LAUNCH("wgrad_batch_fixup_kernel");
    }
    return S2E_OK;
}
