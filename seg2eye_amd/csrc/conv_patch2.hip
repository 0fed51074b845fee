// Patch-resident 3x3 stride-1 convolution, second generation (bf16): 512 output pixels per workgroup.
//
// What bounds conv_patch.hip's multiply loop (round 3).  A CU takes in L2-resident data at ~16-17 bytes per
// clock whatever the instruction mix (LDS-DMA or register loads, 4 or 8 waves issuing: every generic kernel variant of
// round 3 ran into the same 37-46 GB/s per CU), i.e. one 1-KiB piece per ~60 cycles.  A 256-pixel x 128-channel x 64-deep
// K-step is 1024 cycles of MFMA issue per SIMD and needs 16 KB of weights + 5.6 KB of patch = 21 bytes per clock: the loop
// is paced by the loads at ~76 % of the matrix pipe -- the 1430 cycles per K-step the in-kernel stamps of round 1 showed.
// The weight stream costs 4096 / BM bytes per MFMA clock whatever BN is, so the cure is BM, not deeper pipelines:
//
//   tile        : TWO rectangles of up to 256 output pixels each (the rectangles of conv_patch.hip: 16 x 16, 8 x 32, 4 x 64;
//                 they need not be neighbours -- the label-sparse SPADE launch hands over whatever pairs its dense list
//                 holds) x 128 output channels.  Weights per MFMA clock halve: 8 + 5.5 = 13.5 B/clk, under the CU's intake.
//   K-step      : 32 channels of one tap (64-byte LDS rows) instead of 64, so that two patches (2 x 400 pixels x 64 B),
//                 double-buffered, still fit: 2 x 51,200 + 3 weight stages x 8,192 = 126,976 B of LDS.
//   waves       : 8 = 4 (pixels) x 2 (channels), wave tile 128 x 64: 128 accumulator registers, 6 fragment reads per 8 MFMAs
//                 (conv_patch.hip: 4 per 4), one weight piece per wave per K-step instead of two.
//   LDS image   : pixel pp of a patch at byte pp * 64, 16-byte chunk index XORed with (px >> sh) & 3 (sh = 1 for rectangles up to
//                 16 wide, 2 for wider ones: brute-forced conflict-free for ds_read_b128 at every tap shift), weight row r at
//                 r * 64 with (r >> 2) & 3; both applied on the source side of the LDS-DMA.
//   loads       : range-checked buffer loads (an offset past the tensor returns zeros: padding, rows past the patch, the
//                 missing second rectangle of an odd list), 32-bit offsets.
//   everything else as conv_patch.hip: persistent workgroups, the next tile's first loads issued before this tile is written
//   out, hand-pipelined fragment reads with counted waits, fp32 staging of 64 rows at a time in the idle patch buffer,
//   bias / residual / activation / mask epilogue, and the FUSE variant ([gamma | beta] conv + SPADE+Style modulation).
#include "conv_patch.h"
#include <stdlib.h>

namespace {

struct Patch2Params {
    const void* x; const void* w; const float* bias; const void* res; const void* aux; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Kpad;
    int org, flip, out_act, aux_mode;
    int tw, th, sh;                   // rectangle, swizzle shift
    int tiles_x, tiles_y, tiles_n;
    int rects;                        // rectangles of a dense launch (N * tiles_y * tiles_x)
    unsigned x_bytes, w_bytes;        // (< 2^31: the plan checks)
    // FUSE
    const void* mx; const float* mstats; const float* mstyle; int msld; void* mgamma; int mC, mlrelu, mup;
    const int* rect_list; const int* rect_count;
    int dbg;                          // S2E_P2_DEBUG (timing experiments only, results are wrong): 1 no epilogue, 2 no loads in the
                                      // multiply loop, 4 no MFMAs, 8 no fragment reads
};

constexpr int P2_PPX = 400;                           // pixels per rectangle patch (with halo)
constexpr int P2_RECT_BYTES = P2_PPX * 64;            // 25,600
constexpr int P2_B_BYTES = 128 * 64;                  // one weight K-step
constexpr int P2_NBS = 3, P2_PD = 2;

// RECTS = 2: one 512-pixel workgroup of 8 waves per CU (the design above; the only instantiation).  RECTS = 1 -- ONE rectangle per
// workgroup of 4 waves, 77 KB of LDS, TWO independent workgroups per CU, so that the two waves of a SIMD share no barrier -- was
// timed in round 3 (its epilogue staging, 32 KB, does not fit the 25-KB patch buffer: results were not valid, the instruction
// streams are the same): 393 / 307 / 172 us against 409 / 312 / 183 (RECTS = 2) and 400 / 317 / 181 (conv_patch.hip) on
// c128->256 @256^2, c256->128 @256^2, c128->512 @128^2.  Three tilings, one rate: DESIGN 3.1f.
template <bool FUSE, int RECTS>
__global__ __launch_bounds__(256 * RECTS, 2) void conv_patch2_kernel(const Patch2Params p) {
    typedef bf16_t T;
    constexpr int NW = 4 * RECTS, NT = 64 * NW, TAPS = 9;
    constexpr int TM = 4, TN = 2;
    constexpr int P_BYTES = RECTS * P2_RECT_BYTES, B_BYTES = P2_B_BYTES, NBS = P2_NBS, PD = P2_PD;
    constexpr int NPIECE = P_BYTES / 1024;             // pieces of 16 pixels per patch buffer: 25 per rectangle
    constexpr int NR = (NPIECE + NW - 1) / NW;         // 7 per wave either way
    constexpr int NBJ = 8 / NW;                        // weight pieces (16 rows) per wave per K-step
    constexpr int NPASS = 4 * RECTS;                   // epilogue passes of 64 tile rows
    static_assert(NR <= TAPS, "one patch piece per tap must cover the patch");
    static_assert(64 * 128 * 4 <= P_BYTES, "the epilogue stages 64 rows in one patch buffer (RECTS = 1 would need 32-row passes)");
    __shared__ __attribute__((aligned(16))) char smem[2 * P_BYTES + NBS * B_BYTES];
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int TW = p.tw, TH = p.th, PW = TW + 2, PH = TH + 2, SH = p.sh;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ resg = (const T*)p.res;
    const T* __restrict__ auxg = (const T*)p.aux;
    const int nch = p.Cin >> 5, nk = nch * TAPS;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- work items: (pair of rectangles, Cout tile); persistent grid, XCD-contiguous ranges
    struct Item { int tn, n0, y0, x0, n1, y1, x1; };     // n < 0: no rectangle   (scalars, not arrays: a run-time index would
                                                         //  put the struct into LDS / scratch)
    int n_rects = p.rects;
    if constexpr (FUSE) { if (p.rect_count) n_rects = *p.rect_count; }
    const int n_items = ((n_rects + RECTS - 1) / RECTS) * p.tiles_n;
    const int G = gridDim.x;
    int item_id = xcd_remap(blockIdx.x, G);
    if (item_id >= n_items) return;
    if (p.dbg >> 8) {                                  // experiment: de-phase the workgroups (delay in microseconds x group of four)
        const int grp = (blockIdx.x >> 3) & 3;         // (blockIdx >> 3: neighbours in one XCD's sequence)
        const long t0 = __builtin_amdgcn_s_memrealtime();
        const long ticks = (long)(p.dbg >> 8) * grp * 100;      // 100 MHz counter
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    }
    auto rect_of = [&](int pair, int k, int fetched) __attribute__((always_inline)) -> int {
        const int idx = RECTS * pair + k;
        if (k >= RECTS || idx >= n_rects) return -1;
        if constexpr (FUSE) { if (p.rect_list) return fetched >= 0 ? fetched : p.rect_list[idx]; }
        return idx;
    };
    auto decode = [&](int id, int f0, int f1) __attribute__((always_inline)) -> Item {
        Item q;
        q.tn = id % p.tiles_n;
        const int pair = id / p.tiles_n;
        auto one = [&](int r, int& n, int& y, int& x) __attribute__((always_inline)) {
            if (r < 0) { n = -1; y = 0; x = 0; return; }
            x = (r % p.tiles_x) * TW; r /= p.tiles_x;
            y = (r % p.tiles_y) * TH;
            n = r / p.tiles_y;
        };
        one(rect_of(pair, 0, f0), q.n0, q.y0, q.x0);
        one(rect_of(pair, 1, f1), q.n1, q.y1, q.x1);
        return q;
    };

    // ---- patch loads.  Piece q = r * 8 + wave (q < 50) covers pixels 16 (q % 25) .. + 15 of rectangle q / 25; this lane brings
    // the 16 bytes at physical chunk lane & 3 of pixel 16 (q % 25) + (lane >> 2), i.e. logical chunk (lane & 3) ^ swz(px)
    unsigned aoff[NR];                                 // byte offset in x of those 16 bytes, channel chunk 0; OOB: zeros
    unsigned woff[NBJ];                                // byte offset in w of this lane's 16 bytes of its weight row(s), k = 0
    auto aim = [&](const Item& q) __attribute__((always_inline)) {
        // (the pixel coordinates of a piece are recomputed per tile -- two small divisions per piece against a 70k-cycle tile --
        // instead of living in 14 registers next to 128 accumulators)
        static_for<0, NR>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int qq = r * NW + wave;
            const bool second = qq >= 25;                               // wave-uniform
            const int pp = 16 * (second ? qq - 25 : qq) + (lane >> 2);
            const int py = pp / PW, px = pp - py * PW;
            const int n = second ? q.n1 : q.n0;
            const int iy = (second ? q.y1 : q.y0) + p.org + py, ix = (second ? q.x1 : q.x0) + p.org + px;
            const bool ok = n >= 0 && py < PH && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            aoff[r] = ok ? 2u * (unsigned)(((n * p.Hi + iy) * p.Wi + ix) * p.Cin) + (unsigned)(((lane & 3) ^ ((px >> SH) & 3)) << 4) : OOB;
        });
        // weight piece of this wave: tile rows 16 wave .. + 15.  FUSE: tile rows 0..63 = gamma rows 64 tn .., rows 64..127 = the
        // beta rows of the same channels (mC rows further down the packed [gamma | beta] matrix)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int trow = 16 * (wave + NW * j) + (lane >> 2);
            const int grow = FUSE ? (trow < 64 ? q.tn * 64 + trow : p.mC + q.tn * 64 + (trow - 64)) : q.tn * 128 + trow;
            woff[j] = 2u * (unsigned)(grow * p.Kpad) + (unsigned)(((lane & 3) ^ ((trow >> 2) & 3)) << 4);
        }
    };
    auto dma_patch = [&](auto R, int chunk, int buf) __attribute__((always_inline)) -> int {
        constexpr int r = decltype(R)::value;
        if (r * NW + wave >= NPIECE) return 0;         // wave-uniform
        const unsigned off = aoff[r] == OOB ? OOB : aoff[r] + 64u * (unsigned)chunk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(smem + buf * P_BYTES + (r * NW + wave) * 1024), 16, (int)off, 0, 0, 0);
        return 1;
    };
    auto dma_w = [&](int kt, int stage) __attribute__((always_inline)) {     // K-step kt = chunk * 9 + patch offset
        const int chunk = kt / TAPS, tp = kt - chunk * TAPS;
        const unsigned koff = 2u * (unsigned)((p.flip ? TAPS - 1 - tp : tp) * p.Cin + chunk * 32);
        static_for<0, NBJ>([&](auto J) {
            constexpr int j = decltype(J)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(smem + 2 * P_BYTES + stage * B_BYTES + (wave + NW * j) * 1024), 16,
                                                     (int)(woff[j] + koff), 0, 0, 0);
        });
    };
    auto prologue = [&](int pbuf) __attribute__((always_inline)) {
        static_for<0, NR>([&](auto R) { dma_patch(R, 0, pbuf); });
        dma_w(0, 0);
        if (nk > 1) dma_w(1, 1);
    };
    auto wait_keep = [&](int n) __attribute__((always_inline)) {             // all but the n youngest loads have landed
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };

    // ---- fragments.  A rows: wave row wm covers tile rows 128 wm .. + 127 = rectangle wm >> 1, its rows 128 (wm & 1) ..
    // fragment address of tap (dy, dx) = a_dx[mi][dx] + (dy * PW * 64 + patch buffer): the swizzle depends on dx only, so three
    // per-lane tables cover the nine taps and a tap costs one v_add per fragment.  (Left alone the compiler hoists all 9 x 4 sums
    // out of the K loop: 36 registers that stay live through the epilogue, where they spilled; the wave-uniform part is made opaque
    // below so that it cannot.)
    uint32_t a_dx[TM][3];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int r = (wm & 1) * 128 + mi * 32 + l31;
        const int ty = r / TW, tx = r - ty * TW;
        const bool in = r < TW * TH;                  // (rows past the rectangle read patch pixel 0 and are never stored)
        const int pp = in ? ty * PW + tx : 0, px = in ? tx : 0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_dx[mi][dx] = (uint32_t)((pp + dx) * 64 + ((h ^ (((px + dx) >> SH) & 3)) << 4));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t a_base = lds0 + (uint32_t)((wm >> 1) * P2_RECT_BYTES);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_dx[mi][dx] += a_base;
    uint32_t b_off[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = wn * 64 + ni * 32 + l31;
        b_off[ni] = lds0 + 2 * P_BYTES + row * 64 + ((h ^ ((row >> 2) & 3)) << 4);
    }
    f32x16_t acc[TM][TN];
    u32x4_t fa[2][TM], fb[2][TN];
    uint32_t a_addr[TM], b_addr[TN];                  // fragment addresses of the current K-step at s = 0; s flips bit 5
    auto aim_frags = [&](int dy, int dx, int pbuf, int stage) __attribute__((always_inline)) {
        int uoff = pbuf * P_BYTES + dy * PW * 64;
        asm volatile("" : "+s"(uoff));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) a_addr[mi] = a_dx[mi][dx] + (uint32_t)uoff;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) b_addr[ni] = b_off[ni] + stage * B_BYTES;
    };
    auto read_frags = [&](int set, int sstep) __attribute__((always_inline)) {     // logical chunk 2 s + h
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][mi]) : "v"(a_addr[mi] ^ (uint32_t)(sstep << 5)) : "memory");
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][ni]) : "v"(b_addr[ni] ^ (uint32_t)(sstep << 5)) : "memory");
    };
    auto frags_ready = [&](int set, bool all) __attribute__((always_inline)) {
        if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
    };
    auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[set][mi]),
                                                                      __builtin_bit_cast(bf16x8_t, fb[set][ni]), acc[mi][ni], 0, 0, 0);
    };

    // ---- epilogue: 8 passes of 64 tile rows (pass ep = rows 64 ep .. + 63 = wave row ep >> 1, its accumulators mi = 2 (ep & 1), + 1)
    // staged as fp32 [64][128] in the idle patch buffer.  pixel of tile row tr: rectangle tr >> 8, its row tr & 255.
    auto row_pixel = [&](const Item& q, int tr, int& n, int& oy, int& ox) __attribute__((always_inline)) -> bool {
        const bool second = tr >= 256;
        const int rr = tr & 255;
        const int ty = rr / TW;
        n = second ? q.n1 : q.n0; oy = (second ? q.y1 : q.y0) + ty; ox = (second ? q.x1 : q.x0) + (rr - ty * TW);
        return n >= 0 && rr < TW * TH && oy < p.Ho && ox < p.Wo;
    };
    // Every LDS access of the epilogue is inline asm and the passes meet at bare s_barriers: through C++ accesses the compiler puts
    // s_waitcnt vmcnt(0) in front of each of them (the next tile's LDS-DMA pieces are in flight and "may alias"), and
    // __syncthreads() carries one too -- so every pass waited out the previous pass's STORES (a ~2 us round trip each: 8 passes =
    // 16 us of a 56-us tile, measured by switching the epilogue off) and the first one the whole prologue.
    auto stage_pass = [&](auto EP, uint32_t cs0) __attribute__((always_inline)) {
        constexpr int ep = decltype(EP)::value;        // compile time: a run-time pass index turned the accumulators into a scratch array
        if (wm == (ep >> 1)) {
            int t = tid;                                   // (opaque: recomputed per pass instead of living -- and spilling -- across them)
            asm volatile("" : "+v"(t));
            const uint32_t wbase = cs0 + (uint32_t)((4 * ((t >> 5) & 1)) * 512 + (wn * 64 + (t & 31)) * 4);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[(ep & 1) * 2 + m2][ni][r];
                        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(wbase), "v"(v),
                                     "n"((m2 * 32 + (r & 3) + 8 * (r >> 2)) * 512 + ni * 128) : "memory");
                    }
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto read_row8 = [&](uint32_t addr, f32x4_t& f0, f32x4_t& f1) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f0), "=&v"(f1) : "v"(addr) : "memory");
    };
    constexpr int TPR = 16, RPP = NT / TPR, SWEEPS = 64 / RPP;      // 32 (16) rows per sweep, 2 (4) sweeps per pass
    const int cw = (tid % TPR) * 8;
    float bv[8];
    auto load_bias = [&](const Item& q) __attribute__((always_inline)) {
        const int co = q.tn * 128 + cw;
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = (p.bias && co < p.Cout) ? p.bias[co + j] : 0.f;
    };
    auto epilogue = [&](const Item& q, int sbuf) __attribute__((always_inline)) {
        const uint32_t cs0 = lds0 + (uint32_t)(sbuf * P_BYTES);
        const int co = q.tn * 128 + cw;
        const bool cok = co < p.Cout;
        const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
        static_for<0, NPASS>([&](auto EP) {
            constexpr int ep = decltype(EP)::value;
            unsigned o[SWEEPS]; bool live[SWEEPS];        // element offsets fit 32 bits (tensors under 2 GB: the plan checks)
            u32x4_t rr[SWEEPS], aa[SWEEPS];
            // (opaque per pass: the eight unrolled passes' pixel arithmetic must not be hoisted to the top of the epilogue, where it
            // would sit in registers beside the 128 accumulators and spill -- a spill reload is a scratch load, and its
            // s_waitcnt vmcnt(0) drains the stores and LDS-DMA pieces this epilogue is meant to overlap with)
            int trow = tid / TPR;
            asm volatile("" : "+v"(trow));
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                int n, oy, ox;
                live[sw] = row_pixel(q, ep * 64 + sw * RPP + trow, n, oy, ox) && cok;
                o[sw] = live[sw] ? (unsigned)(((n * p.Ho + oy) * p.Wo + ox) * p.Cout + co) : 0u;
                rr[sw] = u32x4_t{0u, 0u, 0u, 0u}; aa[sw] = rr[sw];
                if (live[sw] && resg) rr[sw] = *(const u32x4_t*)(resg + o[sw]);
                if (live[sw] && p.aux_mode != S2E_AUX_NONE) aa[sw] = *(const u32x4_t*)(auxg + o[sw]);
            }
            if (ep > 0) lds_barrier();
            if (!(p.dbg & 64)) stage_pass(EP, cs0);
            lds_barrier();
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                const int row = sw * RPP + trow;
                f32x4_t f0, f1;
                read_row8(cs0 + (uint32_t)(row * 512 + cw * 4), f0, f1);
                if (!live[sw]) continue;
                float v[8] = {f0[0] + bv[0], f0[1] + bv[1], f0[2] + bv[2], f0[3] + bv[3], f1[0] + bv[4], f1[1] + bv[5], f1[2] + bv[6], f1[3] + bv[7]};
                if (resg) {
                    float t[8];
                    unpack16<T>(rr[sw], t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += t[j];
                }
                if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = lrelu02(v[j]);
                }
                if (p.aux_mode != S2E_AUX_NONE) {
                    float t[8];
                    unpack16<T>(aa[sw], t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= (t[j] > 0.f ? 1.f : neg);
                }
                if (!(p.dbg & 32)) *(u32x4_t*)(yg + o[sw]) = pack16<T>(v);
                else if (v[0] == 123.456f) yg[o[sw]] = (T)1.f;            // (keep the arithmetic alive)
            }
        });
    };

    // ---- FUSE epilogue (see conv_patch.hip): a thread owns 8 channels of one pixel; gamma in columns fcw.., beta 64 further
    // Per-channel constants of the modulation (b_gamma, mean, rstd, 1 + s0, s1 + b_beta) for the 64 channels of this Cout tile
    // and BOTH rectangles' samples go through LDS: fetched by 160 threads once per tile (16-byte loads, before the next tile's DMAs
    // are issued: vector-memory data returns in order), read back per pass.  Held in registers (40 of them, for the whole
    // epilogue, beside 128 accumulators) they spilled.
    __shared__ __attribute__((aligned(16))) float kc[RECTS][5][64];
    auto load_mod_consts = [&](const Item& q) __attribute__((always_inline)) {
        if (tid < 80 * RECTS) {
            const int k = tid / 80, t = tid - k * 80, which = t >> 4, c4 = (t & 15) * 4;    // 5 arrays x 16 float4 per rectangle
            const int nn = k ? q.n1 : q.n0, n = nn < 0 ? 0 : nn;
            const int c = q.tn * 64 + c4;
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (which == 0) { if (p.bias) v = *(const f32x4_t*)(p.bias + c); }
            else if (which == 1 || which == 2) {                        // mean / rstd are interleaved: {m, r, m, r}
                const f32x4_t a = *(const f32x4_t*)(p.mstats + ((size_t)n * p.mC + c) * 2), b = *(const f32x4_t*)(p.mstats + ((size_t)n * p.mC + c) * 2 + 4);
                v = which == 1 ? f32x4_t{a[0], a[2], b[0], b[2]} : f32x4_t{a[1], a[3], b[1], b[3]};
            } else if (which == 3) {
                const f32x4_t s0 = *(const f32x4_t*)(p.mstyle + (size_t)n * p.msld + c);
                v = f32x4_t{1.f + s0[0], 1.f + s0[1], 1.f + s0[2], 1.f + s0[3]};
            } else {
                const f32x4_t s1 = *(const f32x4_t*)(p.mstyle + (size_t)n * p.msld + p.mC + c);
                f32x4_t bb = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bb = *(const f32x4_t*)(p.bias + p.mC + c);
                v = f32x4_t{s1[0] + bb[0], s1[1] + bb[1], s1[2] + bb[2], s1[3] + bb[3]};
            }
            *(f32x4_t*)&kc[k][which][c4] = v;
        }
    };
    // FUSE epilogue (see conv_patch.hip): a thread owns 8 channels of one pixel; gamma in columns fcw.., beta 64 further.  x is a
    // compiler-managed load at the top of its pass (an asm load with counted waits -- requested a pass ahead so that it does not
    // wait out the previous pass's stores -- was tried: with 128 accumulators live its pending destination registers get copied
    // or spilled by the compiler before the data lands).
    auto epilogue_fused = [&](const Item& q, int sbuf) __attribute__((always_inline)) {
        const uint32_t cs0 = lds0 + (uint32_t)(sbuf * P_BYTES);
        const uint32_t kc0 = (uint32_t)(uintptr_t)(lptr_t)&kc[0][0][0];
        const T* __restrict__ mx = (const T*)p.mx;
        T* __restrict__ gout = (T*)p.mgamma;
        constexpr int FROWS = NT / 8, FSW = 64 / FROWS;   // rows per sweep (8 threads per row): 64 (32); sweeps per pass: 1 (2)
        static_for<0, NPASS>([&](auto EP) {
            constexpr int ep = decltype(EP)::value;
            int t = tid;
            asm volatile("" : "+v"(t));                   // (opaque per pass: see the plain epilogue)
            const int fc = (t & 7) * 8;
            const int c = q.tn * 64 + fc;
            unsigned o[FSW]; bool live[FSW];
            u32x4_t xx[FSW];
#pragma unroll
            for (int sw = 0; sw < FSW; ++sw) {
                int n, oy, ox;
                live[sw] = row_pixel(q, ep * 64 + sw * FROWS + (t >> 3), n, oy, ox);
                o[sw] = live[sw] ? (unsigned)(((n * p.Ho + oy) * p.Wo + ox) * p.mC + c) : 0u;
                const unsigned oin = (live[sw] && p.mup) ? (unsigned)(((n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.mC + c) : o[sw];
                xx[sw] = u32x4_t{0u, 0u, 0u, 0u};
                if (live[sw]) xx[sw] = *(const u32x4_t*)(mx + oin);
            }
            if (ep > 0) lds_barrier();
            stage_pass(EP, cs0);
            lds_barrier();
            constexpr int kk = ep >> 2;                            // rows 256.. belong to the second rectangle (another sample, possibly)
#pragma unroll
            for (int sw = 0; sw < FSW; ++sw) {
                const int row = sw * FROWS + (t >> 3);
                // four channels at a time: the five constants of eight channels at once (40 registers beside the accumulators) spilled
                float f[8], ga[8], v[8];
                unpack16<T>(xx[sw], f);
                const uint32_t ka = kc0 + (uint32_t)((kk * 5 * 64 + fc) * 4), ca = cs0 + (uint32_t)(row * 512 + fc * 4);
                static_for<0, 2>([&](auto HF) {
                    constexpr int hf = decltype(HF)::value;
                    f32x4_t g4, b4, kbg, kmu, krs, ksa, ksb;
                    asm volatile("ds_read_b128 %0, %7 offset:%8\n\tds_read_b128 %1, %7 offset:%9\n\t"
                                 "ds_read_b128 %2, %10 offset:%11\n\tds_read_b128 %3, %10 offset:%12\n\tds_read_b128 %4, %10 offset:%13\n\t"
                                 "ds_read_b128 %5, %10 offset:%14\n\tds_read_b128 %6, %10 offset:%15\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(g4), "=&v"(b4), "=&v"(kbg), "=&v"(kmu), "=&v"(krs), "=&v"(ksa), "=&v"(ksb)
                                 : "v"(ca), "n"(hf * 16), "n"(256 + hf * 16), "v"(ka), "n"(hf * 16), "n"(256 + hf * 16), "n"(512 + hf * 16),
                                   "n"(768 + hf * 16), "n"(1024 + hf * 16) : "memory");
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = hf * 4 + i;
                        ga[j] = g4[i] + kbg[i];
                        const float xh = (f[j] - kmu[i]) * krs[i];
                        v[j] = 0.5f * (xh * (1.f + ga[j]) + (b4[i] + ksb[i]) + f[j] * ksa[i]);
                    }
                });
                if (!live[sw]) continue;
                if (p.mlrelu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = lrelu02(v[j]);
                }
                *(u32x4_t*)(yg + o[sw]) = pack16<T>(v);
                if (gout) *(u32x4_t*)(gout + o[sw]) = pack16<T>(ga);
            }
        });
    };

    auto fetch_rects = [&](int id, int& f0, int& f1) __attribute__((always_inline)) {
        f0 = -1; f1 = -1;
        if constexpr (FUSE) {
            if (p.rect_list && id < n_items) {
                const int pair = id / p.tiles_n;
                f0 = p.rect_list[RECTS * pair];
                if (RECTS > 1 && RECTS * pair + 1 < n_rects) f1 = p.rect_list[RECTS * pair + 1];
            }
        }
    };

    int f0, f1;
    fetch_rects(item_id, f0, f1);
    Item cur = decode(item_id, f0, f1);
    aim(cur);
    int pb = 0;
    prologue(pb);
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int nf0, nf1;                                  // label-sparse launch: the next item's rectangles, requested an item ahead
        fetch_rects(item_id + G, nf0, nf1);
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
                if (!(p.dbg & 2)) {
                    if constexpr (tap < NR) { if (more) issued += dma_patch(TAP, c + 1, pcur ^ 1); }
                    if (kt + PD < nk) { dma_w(kt + PD, stage == 0 ? NBS - 1 : stage - 1); issued += NBJ; }
                }
                if (!(p.dbg & 8)) read_frags(1, 1);
                frags_ready(0, false);
                if (!(p.dbg & 4)) mfmas(0);
                // K-step kt+1 (and every older patch piece) has landed for this wave once all but this K-step's loads are back
                wait_keep(issued);
                frags_ready(1, true);
                __builtin_amdgcn_s_barrier();
                stage = stage == NBS - 1 ? 0 : stage + 1;
                ++kt;
                if (kt < nk) {
                    aim_frags(ntap / 3, ntap % 3, ntap == 0 ? pcur ^ 1 : pcur, stage);
                    if (!(p.dbg & 8)) read_frags(0, 0);
                }
                if (!(p.dbg & 4)) mfmas(1);
            });
        }
        // every buffer is free now: start the next item's loads, then write this one out underneath them
        const int pbn = (pb + nch) & 1;
        const int next_id = item_id + G;
        const bool has_next = next_id < n_items;
        Item nxt = cur;
        if constexpr (FUSE) load_mod_consts(cur); else load_bias(cur);
        if (has_next) { nxt = decode(next_id, nf0, nf1); aim(nxt); prologue(pbn); }
        if (!(p.dbg & 1)) { if constexpr (FUSE) epilogue_fused(cur, pbn ^ 1); else epilogue(cur, pbn ^ 1); }
        if (!has_next) break;
        cur = nxt; item_id = next_id; pb = pbn;
    }
}

int p2_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// S2E_CONV_PATCH2 = the 256-pixel work items a launch must have for this kernel to take it; default 0 = OFF: measured in round 3
// it runs the layers it takes at the first generation's rate (-1.5 % on the whole step), see DESIGN 3.1f.  tools/check_patch2.py
// and the GPU test of the same name run it with S2E_CONV_PATCH2=448.
int p2_min_items() {
    static const int n = [] { const char* e = getenv("S2E_CONV_PATCH2"); return e ? atoi(e) : 0; }();
    return n;
}
template <bool FUSE>
int p2_launch(const Patch2Params& p, long rects_upper, hipStream_t st) {
    const long items = ((rects_upper + 1) / 2) * p.tiles_n;
    const int grid = items < p2_cu_count() ? (int)items : p2_cu_count();
    conv_patch2_kernel<FUSE, 2><<<grid, 512, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_patch2_kernel");
    return S2E_OK;
}

}  // namespace

// Shapes the 512-pixel kernel takes: bf16, 3x3, stride 1 (forward or data-gradient), no fused input activation, Cin a
// multiple of 32, Cout a multiple of 8 and > 64 (one 128-channel tile at least; the 64-channel tiles of conv_patch.hip stay
// there), the rectangle plan of conv_patch.hip, no tanh, tensors under 2 GB, and at least S2E_CONV_PATCH2 work items.
int s2e_conv_patch2_plan(int dtype, const s2e_conv_desc* d, s2e_patch_plan* plan) {
    s2e_patch_plan local;
    if (!plan) plan = &local;
    if (p2_min_items() <= 0 || dtype != S2E_BF16) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->in_act != S2E_ACT_NONE || d->out_act == S2E_ACT_TANH) return 0;
    if (d->Cin % 32 != 0 || d->Cout % 8 != 0 || d->Cout <= 64) return 0;
    const int grow = d->transposed ? 2 - 2 * d->pad : 2 * d->pad - 2;
    if (d->Ho != d->Hi + grow || d->Wo != d->Wi + grow) return 0;
    if ((long)d->N * d->Hi * d->Wi * d->Cin * 2 >= (1L << 31) || (long)d->N * d->Ho * d->Wo * d->Cout * 2 >= (1L << 31)) return 0;
    if ((long)s2e_conv_cout_pad(d->Cout) * s2e_conv_k_pad(dtype, 9 * d->Cin) * 2 >= (1L << 31)) return 0;
    plan->splits = 1;
    if (s2e_patch_rectangle(d, 3, &plan->tw, &plan->th) < 0.8) return 0;
    const long rects = (long)d->N * ceil_div(d->Ho, plan->th) * ceil_div(d->Wo, plan->tw);
    return rects * ceil_div(d->Cout, 128) >= p2_min_items();
}

static void p2_fill(Patch2Params* p, const s2e_conv_desc* d, const s2e_patch_plan* plan, int kpad) {
    p->N = d->N; p->Hi = d->Hi; p->Wi = d->Wi; p->Cin = d->Cin; p->Ho = d->Ho; p->Wo = d->Wo; p->Cout = d->Cout; p->Kpad = kpad;
    p->org = d->transposed ? d->pad - 2 : -d->pad;
    p->flip = d->transposed ? 1 : 0;
    p->out_act = d->out_act; p->aux_mode = d->aux_mode;
    p->tw = plan->tw; p->th = plan->th; p->sh = plan->tw <= 16 ? 1 : 2;
    p->tiles_x = ceil_div(d->Wo, p->tw); p->tiles_y = ceil_div(d->Ho, p->th);
    p->rects = d->N * p->tiles_y * p->tiles_x;
    p->x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Cin * 2);
    static const int dbg = [] { const char* e = getenv("S2E_P2_DEBUG"); return e ? atoi(e) : 0; }();
    p->dbg = dbg;
}

int s2e_conv_patch2_launch(const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                           const void* aux, void* y, const s2e_conv_desc* d, int kpad, hipStream_t st) {
    Patch2Params p{};
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.aux = aux; p.y = y;
    p2_fill(&p, d, plan, kpad);
    p.tiles_n = ceil_div(d->Cout, 128);
    p.w_bytes = (unsigned)((long)s2e_conv_cout_pad(d->Cout) * kpad * 2);
    return p2_launch<false>(p, p.rects, st);
}

// The fused [gamma | beta] conv + modulation through the 512-pixel kernel: 1 = launched, 0 = not this shape (the caller runs
// conv_patch.hip's kernel), < 0 = error.  Same rectangles as conv_patch.hip's plan (tw, th given), so the label-sparse lists
// built for one serve the other.
int s2e_spade_conv_modulate_patch2(int dtype, const void* actv, const void* w_packed, const float* bias, const void* x,
                                   const float* stats, const float* style, int style_ld, void* out, void* gamma_out,
                                   int N, int H, int W, int C, int nh, int lrelu, int flags, int tw, int th,
                                   const int* rect_list, const int* rect_count, hipStream_t st) {
    if (p2_min_items() <= 0 || dtype != S2E_BF16 || nh % 32 != 0 || C % 64 != 0 || (flags & 1)) return 0;
    const long rects = (long)N * ceil_div(H, th) * ceil_div(W, tw);
    if (rects * (C / 64) < p2_min_items()) return 0;
    if ((long)N * H * W * nh * 2 >= (1L << 31) || (long)N * H * W * C * 2 >= (1L << 31)) return 0;
    const int kpad = ceil_div(9 * nh, 64) * 64;
    if ((long)s2e_conv_cout_pad(2 * C) * kpad * 2 >= (1L << 31)) return 0;
    const s2e_conv_desc d{N, H, W, nh, H, W, 2 * C, 3, 3, 1, 1, 0, S2E_ACT_NONE, S2E_ACT_NONE, S2E_AUX_NONE};
    const s2e_patch_plan plan{tw, th, 1};
    Patch2Params p{};
    p.x = actv; p.w = w_packed; p.bias = bias; p.y = out;
    p2_fill(&p, &d, &plan, kpad);
    p.tiles_n = C / 64;
    p.w_bytes = (unsigned)((long)s2e_conv_cout_pad(2 * C) * kpad * 2);
    p.mx = x; p.mstats = stats; p.mstyle = style; p.msld = style_ld > 0 ? style_ld : 2 * C; p.mgamma = gamma_out;
    p.mC = C; p.mlrelu = lrelu; p.mup = (flags & 8) != 0;
    p.rect_list = rect_list; p.rect_count = rect_count;
    // (rects: an upper bound -- a sparse launch reads the count on the device)
    const int rc = p2_launch<true>(p, rects, st);
    return rc == S2E_OK ? 1 : rc;
}
