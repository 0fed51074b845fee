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
};

constexpr int P2_PPX = 400;                           // pixels per rectangle patch (with halo)
constexpr int P2_RECT_BYTES = P2_PPX * 64;            // 25,600
constexpr int P2_P_BYTES = 2 * P2_RECT_BYTES;         // one patch buffer: both rectangles
constexpr int P2_B_BYTES = 128 * 64;                  // one weight K-step
constexpr int P2_NBS = 3, P2_PD = 2;
constexpr int P2_NPIECE = P2_P_BYTES / 1024;          // 50 pieces of 16 pixels
constexpr int P2_NR = (P2_NPIECE + 7) / 8;            // 7 per wave

template <bool FUSE>
__global__ __launch_bounds__(512, 1) void conv_patch2_kernel(const Patch2Params p) {
    typedef bf16_t T;
    constexpr int NW = 8, NT = 512, TAPS = 9;
    constexpr int TM = 4, TN = 2;
    constexpr int P_BYTES = P2_P_BYTES, B_BYTES = P2_B_BYTES, NBS = P2_NBS, PD = P2_PD, NR = P2_NR;
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
    struct Item { int tn; int n[2], oy0[2], ox0[2]; };   // n < 0: no rectangle
    int n_rects = p.rects;
    if constexpr (FUSE) { if (p.rect_count) n_rects = *p.rect_count; }
    const int n_items = ((n_rects + 1) >> 1) * p.tiles_n;
    const int G = gridDim.x;
    int item_id = xcd_remap(blockIdx.x, G);
    if (item_id >= n_items) return;
    auto rect_of = [&](int pair, int k, int fetched) __attribute__((always_inline)) -> int {
        const int idx = 2 * pair + k;
        if (idx >= n_rects) return -1;
        if constexpr (FUSE) { if (p.rect_list) return fetched >= 0 ? fetched : p.rect_list[idx]; }
        return idx;
    };
    auto decode = [&](int id, int f0, int f1) __attribute__((always_inline)) -> Item {
        Item q;
        q.tn = id % p.tiles_n;
        const int pair = id / p.tiles_n;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int r = rect_of(pair, k, k ? f1 : f0);
            if (r < 0) { q.n[k] = -1; q.oy0[k] = 0; q.ox0[k] = 0; continue; }
            q.ox0[k] = (r % p.tiles_x) * TW; r /= p.tiles_x;
            q.oy0[k] = (r % p.tiles_y) * TH;
            q.n[k] = r / p.tiles_y;
        }
        return q;
    };

    // ---- patch loads.  Piece q = r * 8 + wave (q < 50) covers pixels 16 (q % 25) .. + 15 of rectangle q / 25; this lane brings
    // the 16 bytes at physical chunk lane & 3 of pixel 16 (q % 25) + (lane >> 2), i.e. logical chunk (lane & 3) ^ swz(px)
    int ppyx[NR];                                      // (py << 16) | px; py >= PH: past the patch
    unsigned pcol[NR];                                 // byte offset of this lane's logical chunk inside a 64-byte row
    static_for<0, NR>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int q = r * NW + wave;
        const int pp = 16 * (q % 25) + (lane >> 2);
        const int py = pp / PW, px = pp - py * PW;
        ppyx[r] = (py << 16) | px;
        pcol[r] = (unsigned)(((lane & 3) ^ ((px >> SH) & 3)) << 4);
    });
    unsigned aoff[NR];                                 // byte offset in x of those 16 bytes, channel chunk 0; OOB: zeros
    unsigned woff;                                     // byte offset in w of this lane's 16 bytes of its weight row, k = 0
    auto aim = [&](const Item& q) __attribute__((always_inline)) {
        static_for<0, NR>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int k = (r * NW + wave) >= 25 ? 1 : 0;                 // wave-uniform
            const int py = ppyx[r] >> 16, px = ppyx[r] & 0xffff;
            const int iy = q.oy0[k] + p.org + py, ix = q.ox0[k] + p.org + px;
            const bool ok = q.n[k] >= 0 && py < PH && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            aoff[r] = ok ? 2u * (unsigned)(((q.n[k] * p.Hi + iy) * p.Wi + ix) * p.Cin) + pcol[r] : OOB;
        });
        // weight piece of this wave: tile rows 16 wave .. + 15.  FUSE: tile rows 0..63 = gamma rows 64 tn .., rows 64..127 = the
        // beta rows of the same channels (mC rows further down the packed [gamma | beta] matrix)
        const int trow = 16 * wave + (lane >> 2);
        const int grow = FUSE ? (trow < 64 ? q.tn * 64 + trow : p.mC + q.tn * 64 + (trow - 64)) : q.tn * 128 + trow;
        woff = 2u * (unsigned)(grow * p.Kpad) + (unsigned)(((lane & 3) ^ ((trow >> 2) & 3)) << 4);
    };
    auto dma_patch = [&](auto R, int chunk, int buf) __attribute__((always_inline)) -> int {
        constexpr int r = decltype(R)::value;
        if (r * NW + wave >= P2_NPIECE) return 0;      // wave-uniform
        const unsigned off = aoff[r] == OOB ? OOB : aoff[r] + 64u * (unsigned)chunk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(smem + buf * P_BYTES + (r * NW + wave) * 1024), 16, (int)off, 0, 0, 0);
        return 1;
    };
    auto dma_w = [&](int kt, int stage) __attribute__((always_inline)) {     // K-step kt = chunk * 9 + patch offset
        const int chunk = kt / TAPS, tp = kt - chunk * TAPS;
        const unsigned off = woff + 2u * (unsigned)((p.flip ? TAPS - 1 - tp : tp) * p.Cin + chunk * 32);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(smem + 2 * P_BYTES + stage * B_BYTES + wave * 1024), 16, (int)off, 0, 0, 0);
    };
    auto prologue = [&](int pbuf) __attribute__((always_inline)) {
        static_for<0, NR>([&](auto R) { dma_patch(R, 0, pbuf); });
        dma_w(0, 0);
        if (nk > 1) dma_w(1, 1);
    };
    auto wait_keep = [&](int n) __attribute__((always_inline)) {             // all but the n youngest loads have landed
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };

    // ---- fragments.  A rows: wave row wm covers tile rows 128 wm .. + 127 = rectangle wm >> 1, its rows 128 (wm & 1) ..
    int pp0[TM], px0[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int r = (wm & 1) * 128 + mi * 32 + l31;
        const int ty = r / TW, tx = r - ty * TW;
        const bool in = r < TW * TH;                  // (rows past the rectangle read patch pixel 0 and are never stored)
        pp0[mi] = in ? ty * PW + tx : 0;
        px0[mi] = in ? tx : 0;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t a_base = lds0 + (uint32_t)((wm >> 1) * P2_RECT_BYTES);
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
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int pp = pp0[mi] + dy * PW + dx;
            a_addr[mi] = a_base + pbuf * P_BYTES + pp * 64 + ((h ^ (((px0[mi] + dx) >> SH) & 3)) << 4);
        }
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
        const int k = tr >> 8, rr = tr & 255;
        const int ty = rr / TW;
        n = q.n[k]; oy = q.oy0[k] + ty; ox = q.ox0[k] + (rr - ty * TW);
        return n >= 0 && rr < TW * TH && oy < p.Ho && ox < p.Wo;
    };
    auto stage_pass = [&](int ep, float* Cs) __attribute__((always_inline)) {
        if (wm == (ep >> 1)) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = (ep & 1) ? acc[2 + m2][ni][r] : acc[m2][ni][r];
                        Cs[(m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 128 + wn * 64 + ni * 32 + l31] = v;
                    }
        }
    };
    constexpr int TPR = 16, RPP = NT / TPR, SWEEPS = 64 / RPP;      // 32 rows per sweep, 2 sweeps per pass
    const int cw = (tid % TPR) * 8;
    float bv[8];
    auto load_bias = [&](const Item& q) __attribute__((always_inline)) {
        const int co = q.tn * 128 + cw;
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = (p.bias && co < p.Cout) ? p.bias[co + j] : 0.f;
    };
    auto epilogue = [&](const Item& q, int sbuf) __attribute__((always_inline)) {
        float* Cs = (float*)(smem + sbuf * P_BYTES);
        const int co = q.tn * 128 + cw;
        const bool cok = co < p.Cout;
        const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
        for (int ep = 0; ep < 8; ++ep) {
            size_t o[SWEEPS]; bool live[SWEEPS];
            u32x4_t rr[SWEEPS], aa[SWEEPS];
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                int n, oy, ox;
                live[sw] = row_pixel(q, ep * 64 + sw * RPP + tid / TPR, n, oy, ox) && cok;
                o[sw] = live[sw] ? ((size_t)(n * p.Ho + oy) * p.Wo + ox) * p.Cout + co : 0;
                rr[sw] = u32x4_t{0u, 0u, 0u, 0u}; aa[sw] = rr[sw];
                if (live[sw] && resg) rr[sw] = *(const u32x4_t*)(resg + o[sw]);
                if (live[sw] && p.aux_mode != S2E_AUX_NONE) aa[sw] = *(const u32x4_t*)(auxg + o[sw]);
            }
            if (ep > 0) __syncthreads();
            if (ep == 0 || ep == 1) stage_pass(ep, Cs);
            else if (ep == 2 || ep == 3) stage_pass(ep, Cs);
            else if (ep == 4 || ep == 5) stage_pass(ep, Cs);
            else stage_pass(ep, Cs);
            __syncthreads();
#pragma unroll
            for (int sw = 0; sw < SWEEPS; ++sw) {
                if (!live[sw]) continue;
                const int row = sw * RPP + tid / TPR;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    const f32x4_t f = *(const f32x4_t*)(Cs + row * 128 + cw + j);
                    v[j] = f[0] + bv[j]; v[j + 1] = f[1] + bv[j + 1]; v[j + 2] = f[2] + bv[j + 2]; v[j + 3] = f[3] + bv[j + 3];
                }
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
                *(u32x4_t*)(yg + o[sw]) = pack16<T>(v);
            }
        }
    };

    // ---- FUSE epilogue (see conv_patch.hip): a thread owns 8 channels of one pixel; gamma in columns fcw.., beta 64 further
    constexpr int FTPR = 8;                            // threads per row (64 channels / 8); 512 / 8 = 64 rows = one pass per sweep
    const int fcw = (tid % FTPR) * 8;
    float k_bg[8], k_mu[8], k_rs[8], k_sa[8], k_sb[8];
    auto load_mod_consts = [&](const Item& q, int k) __attribute__((always_inline)) {
        const int c = q.tn * 64 + fcw;
        const int n = q.n[k] < 0 ? 0 : q.n[k];
        const f32x4_t* stp = (const f32x4_t*)(p.mstats + ((size_t)n * p.mC + c) * 2);
        const f32x4_t* s0p = (const f32x4_t*)(p.mstyle + (size_t)n * p.msld + c);
        const f32x4_t* s1p = (const f32x4_t*)(p.mstyle + (size_t)n * p.msld + p.mC + c);
        const f32x4_t* bgp = (const f32x4_t*)(p.bias + c);
        const f32x4_t* bbp = (const f32x4_t*)(p.bias + p.mC + c);
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const f32x4_t st0 = stp[j / 2], st1 = stp[j / 2 + 1], s0 = s0p[j / 4], s1 = s1p[j / 4];
            f32x4_t bg = {0.f, 0.f, 0.f, 0.f}, bb = bg;
            if (p.bias) { bg = bgp[j / 4]; bb = bbp[j / 4]; }
            k_mu[j] = st0[0]; k_rs[j] = st0[1]; k_mu[j + 1] = st0[2]; k_rs[j + 1] = st0[3];
            k_mu[j + 2] = st1[0]; k_rs[j + 2] = st1[1]; k_mu[j + 3] = st1[2]; k_rs[j + 3] = st1[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) { k_bg[j + i] = bg[i]; k_sa[j + i] = 1.f + s0[i]; k_sb[j + i] = s1[i] + bb[i]; }
        }
    };
    auto epilogue_fused = [&](const Item& q, int sbuf) __attribute__((always_inline)) {
        float* Cs = (float*)(smem + sbuf * P_BYTES);
        const T* __restrict__ mx = (const T*)p.mx;
        T* __restrict__ gout = (T*)p.mgamma;
        const int c = q.tn * 64 + fcw;
        for (int ep = 0; ep < 8; ++ep) {
            if (ep == 4) load_mod_consts(q, 1);       // rows 256.. belong to the second rectangle (another sample, possibly)
            int n, oy, ox;
            const bool live = row_pixel(q, ep * 64 + tid / FTPR, n, oy, ox);
            const size_t o = live ? ((size_t)(n * p.Ho + oy) * p.Wo + ox) * p.mC + c : 0;
            const size_t oin = (live && p.mup) ? ((size_t)(n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.mC + c : o;
            u32x4_t xx = u32x4_t{0u, 0u, 0u, 0u};
            if (live) xx = *(const u32x4_t*)(mx + oin);
            if (ep > 0) __syncthreads();
            if (ep == 0 || ep == 1) stage_pass(ep, Cs);
            else if (ep == 2 || ep == 3) stage_pass(ep, Cs);
            else if (ep == 4 || ep == 5) stage_pass(ep, Cs);
            else stage_pass(ep, Cs);
            __syncthreads();
            if (!live) continue;
            const int row = tid / FTPR;
            float f[8], ga[8], v[8];
            unpack16<T>(xx, f);
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                const f32x4_t g4 = *(const f32x4_t*)(Cs + row * 128 + fcw + j);
                const f32x4_t b4 = *(const f32x4_t*)(Cs + row * 128 + 64 + fcw + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ga[j + i] = g4[i] + k_bg[j + i];
                    const float xh = (f[j + i] - k_mu[j + i]) * k_rs[j + i];
                    v[j + i] = 0.5f * (xh * (1.f + ga[j + i]) + (b4[i] + k_sb[j + i]) + f[j + i] * k_sa[j + i]);
                }
            }
            if (p.mlrelu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = lrelu02(v[j]);
            }
            *(u32x4_t*)(yg + o) = pack16<T>(v);
            if (gout) *(u32x4_t*)(gout + o) = pack16<T>(ga);
        }
    };

    auto fetch_rects = [&](int id, int& f0, int& f1) __attribute__((always_inline)) {
        f0 = -1; f1 = -1;
        if constexpr (FUSE) {
            if (p.rect_list && id < n_items) {
                const int pair = id / p.tiles_n;
                f0 = p.rect_list[2 * pair];
                if (2 * pair + 1 < n_rects) f1 = p.rect_list[2 * pair + 1];
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
                if constexpr (tap < NR) { if (more) issued += dma_patch(TAP, c + 1, pcur ^ 1); }
                if (kt + PD < nk) { dma_w(kt + PD, stage == 0 ? NBS - 1 : stage - 1); issued += 1; }
                read_frags(1, 1); frags_ready(0, false); mfmas(0);
                // K-step kt+1 (and every older patch piece) has landed for this wave once all but this K-step's loads are back
                wait_keep(issued);
                frags_ready(1, true);
                __builtin_amdgcn_s_barrier();
                stage = stage == NBS - 1 ? 0 : stage + 1;
                ++kt;
                if (kt < nk) {
                    aim_frags(ntap / 3, ntap % 3, ntap == 0 ? pcur ^ 1 : pcur, stage);
                    read_frags(0, 0);
                }
                mfmas(1);
            });
        }
        // every buffer is free now: start the next item's loads, then write this one out underneath them
        const int pbn = (pb + nch) & 1;
        const int next_id = item_id + G;
        const bool has_next = next_id < n_items;
        Item nxt = cur;
        if constexpr (FUSE) load_mod_consts(cur, 0); else load_bias(cur);
        if (has_next) { nxt = decode(next_id, nf0, nf1); aim(nxt); prologue(pbn); }
        if constexpr (FUSE) epilogue_fused(cur, pbn ^ 1); else epilogue(cur, pbn ^ 1);
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

// items a launch must have: S2E_CONV_PATCH2 (default 224 = the first generation's tile threshold; 0 = off)
int p2_min_items() {
    static const int n = [] { const char* e = getenv("S2E_CONV_PATCH2"); return e ? atoi(e) : 224; }();
    return n;
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
    if ((long)d->N * d->Hi * d->Wi * d->Cin * 2 >= (1L << 31)) return 0;
    if ((long)s2e_conv_cout_pad(d->Cout) * s2e_conv_k_pad(dtype, 9 * d->Cin) * 2 >= (1L << 31)) return 0;
    plan->splits = 1;
    if (s2e_patch_rectangle(d, 3, &plan->tw, &plan->th) < 0.8) return 0;
    const long rects = (long)d->N * ceil_div(d->Ho, plan->th) * ceil_div(d->Wo, plan->tw);
    const long items = ((rects + 1) / 2) * ceil_div(d->Cout, 128);
    return items >= p2_min_items();
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
}

int s2e_conv_patch2_launch(const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                           const void* aux, void* y, const s2e_conv_desc* d, int kpad, hipStream_t st) {
    Patch2Params p{};
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.aux = aux; p.y = y;
    p2_fill(&p, d, plan, kpad);
    p.tiles_n = ceil_div(d->Cout, 128);
    p.w_bytes = (unsigned)((long)s2e_conv_cout_pad(d->Cout) * kpad * 2);
    const int items = ((p.rects + 1) / 2) * p.tiles_n;
    const int grid = items < p2_cu_count() ? items : p2_cu_count();
    conv_patch2_kernel<false><<<grid, 512, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_patch2_kernel");
    return S2E_OK;
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
    if (((rects + 1) / 2) * (C / 64) < p2_min_items()) return 0;
    if ((long)N * H * W * nh * 2 >= (1L << 31)) return 0;
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
    const long items = ((rects + 1) / 2) * p.tiles_n;              // upper bound (a sparse launch reads the count on the device)
    const int grid = items < p2_cu_count() ? (int)items : p2_cu_count();
    conv_patch2_kernel<true><<<grid, 512, 0, st>>>(p);
    S2E_CHECK_LAUNCH("conv_patch2_kernel (fused modulation)");
    return 1;
}
