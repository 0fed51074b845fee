// Patch-resident 3x3 stride-1 convolution, "duo" form (bf16): TWO independent 4-wave workgroups per CU, half a tile out of phase.
//
// Why.  conv_patch.hip runs one 512-thread workgroup per CU (151 KB of LDS) and its phases ADD: per tile the
// prologue (address tables, the first loads), the multiply loop and the epilogue (accumulators -> LDS -> 16-byte stores) take
// 123 + 190 + 130 us of a c128->256 @256^2 launch, and while a tile is written out the matrix pipe idles -- 38.6 % busy in the
// SQ counters.  Nothing inside ONE workgroup can fill that gap (its waves meet at every barrier) and de-phasing workgroups on
// DIFFERENT CUs cannot either.  Here every CU holds two workgroups that share nothing but the CU: while one writes its tile out
// and fetches the next patch, the other is in its multiply loop and has the SIMDs' matrix pipes to itself.
//
//   workgroup   : 256 threads = 4 waves (one per SIMD), <= 77 KB of LDS -> two per CU; persistent over work items.
//   tile        : one rectangle of up to 256 output pixels (16 x 16, 8 x 32, 4 x 64 ...: conv_patch.hip's plan) x 128 output channels.
//   waves       : 2 (pixels) x 2 (channels); wave tile 128 x 64 = 8 accumulators of 32 x 32 (128 registers); 6 fragment reads
//                 (ds_read_b128) per 8 MFMAs.
//   K-step      : 32 channels of one tap = 2 x v_mfma_f32_32x32x16_bf16 per accumulator; 64-byte LDS rows.
//   LDS         : 2 patch buffers of 400 pixels x 64 B (chunk c is read while c + 1 lands) + 3 weight stages of 128 rows x 64 B
//                 (K-step kt + 2 in flight while kt is multiplied) = 75,776 B (+ 1.3 KB of per-channel constants, FUSE).
//   LDS image   : pixel pp at byte pp * 64, 16-byte chunk index XORed with (px >> sh) & 3 (sh = 1 for rectangles up to 16 wide, 2
//                 for wider ones), weight row r at r * 64 with (r >> 2) & 3: every ds_read_b128 of a fragment is conflict-free at
//                 every tap shift (brute-forced); both swizzles are applied on the SOURCE side of the LDS-DMA.
//   loads       : range-checked buffer loads straight into LDS (an offset past the tensor returns zeros: padding, rows past the
//                 patch), 32-bit offsets.
//   epilogue    : WAVE-PRIVATE.  A wave stages 16 rows x 64 columns of its own accumulators (4 KB, fp32) in its own slice of the
//                 idle patch buffer and reads them back row-major -- no barrier anywhere in the epilogue, so the four waves write
//                 out independently -- then 16-byte stores with bias / residual / activation / mask fused.  The staging image is
//                 XOR-swizzled so that the 4-byte writes and the 16-byte reads are both conflict-free.
//   FUSE        : the [gamma | beta] conv of a SPADE with the SPADE+Style modulation in the epilogue (see conv_patch.hip): a wave
//                 owns 32 gamma columns and the 32 beta columns of the SAME channels, so the modulation stays wave-private too.
//   constants   : what the epilogues need per channel (bias; FUSE: b_gamma, b_beta, s1, s0, mean / rstd) arrives by LDS-DMA with
//                 the tile's first patch pieces (two 1-KiB pieces, double-buffered): no vector-memory LOAD is issued between a
//                 tile's stores and the next tile's multiply loop, so nothing ever waits for a store to be acknowledged.  The
//                 accumulators START from the bias (FUSE: 1 + b_gamma | b_beta + s1) instead of adding it in the epilogue.
//   priority    : the two workgroups of a CU run the same program at the same rate; the one dispatched first wins every issue
//                 arbitration (priority, then age: MI355X_MICROARCH.md) and ran its multiply loops in 23 us against the other's
//                 33 us -- and then finished its share of the tiles 15 % early, leaving the CU half empty.  Here a workgroup
//                 raises its priority for its multiply loop on every OTHER tile, its partner on the tiles in between (blocks b and
//                 b + gridDim / 2 share a CU: measured on 254 of 256 CUs, tools/probe/dispatch_probe.hip), and both drop to 0 for the
//                 write-out, which tolerates latency.  Starting the second workgroup half a tile late changed nothing (measured).
#include "conv_patch.h"
#include <stdlib.h>

namespace {

__device__ __attribute__((aligned(16))) const uint32_t duo_zero16[4] = {0u, 0u, 0u, 0u};

struct DuoParams {
    const void* x; const void* w; const float* bias; const void* res; const void* aux; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Kpad;
    int org, flip, out_act, aux_mode;
    int tw, th, sh, tw_shift;         // rectangle, swizzle shift, log2(tw) or -1
    int tiles_x, tiles_y, tiles_n;
    int rects;                        // rectangles of a dense launch (N * tiles_y * tiles_x)
    unsigned x_bytes, w_bytes;        // (< 2^31: the plan checks)
    const int* rect_list; const int* rect_count;      // NULL, or the rectangles to compute (rect_count on the device)
    // FUSE
    const void* mx; const float* mstats; const float* mstyle; int msld; void* mgamma; int mC, mlrelu, mup;
    // InstanceNorm partial sums of the OUTPUT (plain launches; NULL = none): per wave and tile, {sum y, sum y^2} of its 64 channels
    // over its rows -> spart[((n * sP + slot) * Cout + c) * 2], slot = rectangle-in-sample * waves-rows-per-tile + wave row;
    // s2e_in_stats_from_partials adds the sP slots of a (sample, channel) in a fixed order (normalization.py:94 of the reference)
    float* spart; int sP;
    long* dbg;                        // -DS2E_DUO_STAMPS builds only (tools/duo_stamps.py): per-tile phase time stamps
    int abl;                          // ... and S2E_DUO_ABL, the timing ablation of the multiply loop
};

constexpr int DU_PPX = 400;                           // pixels per patch (with halo)
constexpr int DU_P_BYTES = DU_PPX * 64;               // 25,600
constexpr int DU_B_BYTES = 128 * 64;                  // one weight K-step
constexpr int DU_NBS = 3, DU_PD = 2;

// BN = output channels per tile: 128 (waves 2 x 2, wave tile 128 x 64), or 64 for the 64-channel layers (waves 4 x 1, wave tile
// 64 x 64: half the accumulators, one weight piece per wave and K-step)
// MF16 = the multiply loop on v_mfma_f32_16x16x32_bf16 instead of 32x32x16: the same FLOPs per cycle and the same LDS reads, but
// the chip -- which lowers its clock under matrix load -- holds a higher one on this shape (MI355X_MICROARCH.md: 1.12-1.15 x the
// FLOP/s at equal cycles).  A K-step is then ONE instruction per 16 x 16 accumulator (k = 32); the wave's 128 x 64 tile is 8 x 4 of
// them.  16 x 16 rectangles only: a fragment is one rectangle row, so every fragment address is a lane base + an immediate.
// EPI = which plain epilogue this instantiation carries: 0 none, 1 residual, 2 mask, 3 / 4 = 0 / 1 with the InstanceNorm partial
// sums, -1 all of them behind run-time branches.  One epilogue per 128-channel kernel: with all five in one function the register
// allocator spilled 33 values that live across them (the next tile's load offsets), and every reload in the tile turnaround is a
// scratch load + vmcnt(0): 6-7 us per tile against 1.2 us without spills (in-kernel stamps, tools/duo_stamps.py).
template <bool FUSE, int BN = 128, bool MF16 = false, int EPI = -1>
__global__ __launch_bounds__(256, 2) void conv_duo_kernel(const DuoParams p) {
    static_assert(BN == 128 || (BN == 64 && !FUSE), "tile widths");
    static_assert(!MF16 || BN == 128, "the 16x16x32 loop is written for 128-channel tiles");
    typedef bf16_t T;
    constexpr int NW = 4, TAPS = 9;
    constexpr int TM = BN / 32, TN = 2;               // wave tile TM * 32 pixels x 64 channels
    constexpr int WROWS = TM * 32, NPASS = 2 * TM;    // tile rows per wave; 16-row epilogue passes per wave
    constexpr int P_BYTES = DU_P_BYTES, B_BYTES = DU_B_BYTES, NBS = DU_NBS, PD = DU_PD;
    constexpr int NPIECE = P_BYTES / 1024;             // 25 pieces of 16 pixels
    constexpr int NR = (NPIECE + NW - 1) / NW;         // 7 per wave
    constexpr int NBJ = BN / 64;                       // weight pieces (16 rows each) per wave per K-step
    constexpr int NR16 = 6;                            // MF16 (16 x 16 rectangles: an 18 x 18 patch = 20.25 pieces): six per wave, every wave
                                                       // issues all six (pixels past the patch read zeros into the buffer's unused tail)
    static_assert(NR <= TAPS, "one patch piece per tap must cover the patch");
    static_assert(NW * 4096 <= P_BYTES, "the waves' staging slices must fit one patch buffer");
    __shared__ __attribute__((aligned(16))) char smem[2 * P_BYTES + NBS * B_BYTES];
    // per-tile constants, double-buffered (tile i + 1's arrive while tile i's epilogue reads its own): 512 floats per buffer --
    // plain: [0, 128) bias of the tile's channels; FUSE: [0, 64) b_gamma, [64, 128) b_beta, [128, 192) s1, [192, 256) s0,
    // [256, 384) {mean, rstd} of the 64 channels; the rest is padding the second piece fills with zeros
    __shared__ __attribute__((aligned(16))) float kcst[2][512];
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
    const int h = lane >> 5, l31 = lane & 31;
    const int TW = p.tw, TH = p.th, PW = TW + 2, PH = TH + 2, SH = p.sh;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ resg = (const T*)p.res;
    const T* __restrict__ auxg = (const T*)p.aux;
    const int nch = p.Cin >> 5, nk = nch * TAPS;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- work items: (rectangle, Cout tile); persistent grid, XCD-contiguous ranges
    struct Item { int tn, n, y0, x0; };
    int n_rects = p.rects;
    if (p.rect_count) n_rects = *p.rect_count;              // a rectangle list (label-sparse launches): the count lives on the device
    const int n_items = n_rects * p.tiles_n;
    const int G = gridDim.x;
    int item_id = xcd_remap(blockIdx.x, G);
    if (item_id >= n_items) return;
    auto decode = [&](int id, int fetched) __attribute__((always_inline)) -> Item {
        Item q;
        q.tn = id % p.tiles_n;
        int r = id / p.tiles_n;
        if (p.rect_list) r = fetched >= 0 ? fetched : p.rect_list[r];
        q.x0 = (r % p.tiles_x) * TW; r /= p.tiles_x;
        q.y0 = (r % p.tiles_y) * TH;
        q.n = r / p.tiles_y;
        return q;
    };

    // ---- patch loads.  Piece qq = r * 4 + wave (qq < 25) covers pixels 16 qq .. + 15; this lane brings the 16 bytes at physical
    // chunk lane & 3 of pixel 16 qq + (lane >> 2), i.e. logical chunk (lane & 3) ^ swz(px)
    unsigned aoff[NR];                                 // byte offset in x of those 16 bytes, channel chunk 0; OOB: zeros
    unsigned woff[NBJ];                                // byte offset in w of this lane's 16 bytes of its weight row(s), k = 0
    auto aim = [&](const Item& q) __attribute__((always_inline)) {
        static_for<0, NR>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int qq = r * NW + wave;
            const int pp = 16 * qq + (lane >> 2);
            const int py = pp / PW, px = pp - py * PW;
            const int iy = q.y0 + p.org + py, ix = q.x0 + p.org + px;
            const bool ok = py < PH && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            aoff[r] = ok ? 2u * (unsigned)(((q.n * p.Hi + iy) * p.Wi + ix) * p.Cin) + (unsigned)(((lane & 3) ^ ((px >> SH) & 3)) << 4) : OOB;
        });
        // weight piece j of this wave: tile rows 16 (wave + 4 j) .. + 15.  FUSE: tile rows 0..63 = gamma rows 64 tn .., rows
        // 64..127 = the beta rows of the same channels (mC rows further down the packed [gamma | beta] matrix)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int trow = 16 * (wave + NW * j) + (lane >> 2);
            const int grow = FUSE ? (trow < 64 ? q.tn * 64 + trow : p.mC + q.tn * 64 + (trow - 64)) : q.tn * BN + trow;
            // (swizzle of the weight rows: what makes the B fragment reads conflict-free -- 32 rows x 2 chunks per read in the
            // 32x32x16 loop, 16 rows x 4 chunks in the 16x16x32 one)
            woff[j] = 2u * (unsigned)(grow * p.Kpad) + (unsigned)(((lane & 3) ^ ((MF16 ? trow >> 1 : trow >> 2) & 3)) << 4);
        }
    };
    auto dma_patch = [&](auto R, int chunk, int buf) __attribute__((always_inline)) -> int {
        constexpr int r = decltype(R)::value;
        if constexpr (MF16) { if (r >= NR16) return 0; }
        else if (r * NW + wave >= NPIECE) return 0;    // wave-uniform
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
    // the tile's per-channel constants: two 1-KiB LDS-DMA pieces issued by wave 0 (lane L brings 16 bytes to kcst[cbuf] + 16 L of
    // the first piece, + 1024 of the second); a missing bias and the padding lanes read a 16-byte zero page
    auto dma_consts = [&](const Item& q, int cbuf) __attribute__((always_inline)) {
        if (wave != 0) return;
        const char* s0p = (const char*)duo_zero16;
        const char* s1p = (const char*)duo_zero16;
        if constexpr (FUSE) {
            const int arr = lane >> 4, i4 = (lane & 15) * 4, c0 = q.tn * 64;
            const float* sty = p.mstyle + (size_t)q.n * p.msld;
            if (arr == 0) { if (p.bias) s0p = (const char*)(p.bias + c0 + i4); }
            else if (arr == 1) { if (p.bias) s0p = (const char*)(p.bias + p.mC + c0 + i4); }
            else if (arr == 2) s0p = (const char*)(sty + p.mC + c0 + i4);
            else s0p = (const char*)(sty + c0 + i4);
            if (lane < 32) s1p = (const char*)(p.mstats + ((size_t)q.n * p.mC + c0) * 2 + lane * 4);
        } else {
            const int c = q.tn * BN + lane * 4;
            if (lane * 4 < BN && p.bias && c < p.Cout) s0p = (const char*)(p.bias + c);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(const void*)s0p, (lptr_t)&kcst[cbuf][0], 16, 0, 0);
        if constexpr (FUSE) __builtin_amdgcn_global_load_lds((gptr_t)(const void*)s1p, (lptr_t)&kcst[cbuf][256], 16, 0, 0);
    };
    auto prologue = [&](const Item& q, int pbuf, int cbuf) __attribute__((always_inline)) {
        dma_consts(q, cbuf);
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

    // ---- fragments.  A rows: wave row wm covers tile rows 128 wm .. + 127.  Fragment address of tap (dy, dx) = a_dx[mi][dx] +
    // (dy * PW * 64 + patch buffer): the swizzle depends on dx only, so three per-lane tables cover the nine taps and a tap costs one
    // v_add per fragment.  (The wave-uniform part is made opaque below so that the compiler cannot hoist all 9 x 4 sums out of the
    // K loop: 36 registers beside 128 accumulators.)
    uint32_t a_dx[TM][3];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int r = wm * WROWS + mi * 32 + l31;
        const int ty = r / TW, tx = r - ty * TW;
        const bool in = r < TW * TH;                  // (rows past the rectangle read patch pixel 0 and are never stored)
        const int pp = in ? ty * PW + tx : 0, px = in ? tx : 0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_dx[mi][dx] = (uint32_t)((pp + dx) * 64 + ((h ^ (((px + dx) >> SH) & 3)) << 4));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_dx[mi][dx] += lds0;
    // B rows of this wave.  Plain: 64 consecutive tile rows (output channels 64 wn ..).  FUSE: 32 gamma rows and the 32 beta rows
    // of the same channels (tile rows 32 wn .. and 64 + 32 wn ..), so that a pixel's gamma and beta meet in ONE wave.
    uint32_t b_off[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = (FUSE ? wn * 32 + ni * 64 : wn * 64 + ni * 32) + l31;
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
    // wait until only the NEWEST set of reads (TM + TN of them) is outstanding, or none; the fragments are operands of the wait so
    // that no MFMA consuming them can be scheduled above it
    auto frags_ready = [&](int set, bool all) __attribute__((always_inline)) {
        if constexpr (TM == 4) {
            if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
        } else {
            if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
            else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fb[set][0]), "+v"(fb[set][1]) :: "memory");
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

    // ---- the same for the 16x16x32 loop (MF16): lane (i = lane & 15, kq = lane >> 4) holds k = 8 kq .. + 7 of fragment row i.
    // A fragment mi = rectangle row 8 wm + mi at tap (dy, dx): a16[dx] + patch buffer + ((mi + dy) * 18 * 64 as an immediate);
    // B fragment ni = 16 weight rows: b16 + stage + an immediate.  The accumulators of rows 0-3 are multiplied while the A
    // fragments of rows 4-7 arrive, those of rows 4-7 while the next K-step's rows 0-3 and B arrive (B in two register sets,
    // alternating with the K-step's parity).
    f32x4_t acc16[MF16 ? 8 : 1][4];
    u32x4_t fal[4], fah[4], fbb[2][4];
    uint32_t a16[3], b16 = 0;
    if constexpr (MF16) {
        const int i = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            a16[dx] = lds0 + (uint32_t)(((8 * wm) * 18 + i + dx) * 64 + ((kq ^ (((i + dx) >> 1) & 3)) << 4));
        b16 = lds0 + 2 * P_BYTES + (uint32_t)((FUSE ? wn * 32 : wn * 64) * 64 + i * 64 + ((kq ^ ((i >> 1) & 3)) << 4));
    }
    // (plain lambdas with int arguments that are constants at every call site: an asm operand inside a GENERIC lambda does not
    // capture -- clang -- and the offsets fold to immediates after inlining)
    auto read_al = [&](uint32_t ab, int dy) __attribute__((always_inline)) {          // A rows 0-3 of a K-step
#pragma unroll
        for (int m = 0; m < 4; ++m)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fal[m]) : "v"(ab), "n"((m + dy) * 1152) : "memory");
    };
    auto read_ah = [&](uint32_t ab, int dy) __attribute__((always_inline)) {          // A rows 4-7
#pragma unroll
        for (int m = 0; m < 4; ++m)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fah[m]) : "v"(ab), "n"((4 + m + dy) * 1152) : "memory");
    };
    auto read_b = [&](uint32_t bb, int par) __attribute__((always_inline)) {
        // FUSE: rows 32 wn + {0, 16} are gamma, 64 + 32 wn + {0, 16} the beta rows of the same channels
#pragma unroll
        for (int n = 0; n < 4; ++n)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbb[par][n]) : "v"(bb), "n"(FUSE ? (n & 1) * 1024 + (n >> 1) * 4096 : n * 1024) : "memory");
    };
    auto mfma16 = [&](int half, int par) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc16[MF16 ? half * 4 + m : 0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8_t, half ? fah[m] : fal[m]), __builtin_bit_cast(bf16x8_t, fbb[par][n]), acc16[MF16 ? half * 4 + m : 0][n], 0, 0, 0);
    };

    // ---- epilogue, wave-private.  Pass q = 0..7 handles the wave's accumulator rows 32 (q >> 1) + 16 (q & 1) .. + 15 (registers
    // r = 8 (q & 1) .. + 7 of acc[q >> 1][*]): staged as fp32 [16 rows][64 columns] (256-byte rows) in this wave's 4-KB slice of the
    // idle patch buffer, 16-byte chunk index XORed with g(row) -- plain: row & 1; FUSE: (row & 1) | ((row & 2) << 2) -- which makes
    // the ds_write_b32 of the accumulator layout AND the ds_read_b128 of the row-major read-back conflict-free.
    //   accumulator register rr (of the half) of lane (h, l31) = staged row 8 (rr >> 2) + 4 h + (rr & 3), column 32 ni + l31.
    // Shapes are interior-only (the plan: power-of-two rectangle width >= 16, full 256-pixel rectangles that tile the map), so a
    // pass's 16 tile rows P .. P + 15 (P = 128 wm + 32 mi + 16 half, a multiple of 16) lie in ONE rectangle row: the pixel offset is
    // a wave-uniform part (scalar registers) plus a lane part that never changes -- no per-lane division, no bounds test.
    auto pass_yx = [&](int P, int& ty, int& tx) __attribute__((always_inline)) { ty = P >> p.tw_shift; tx = P & (TW - 1); };
    auto stage_pass = [&](auto Q, uint32_t wslice) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;          // compile time: a run-time pass index would turn the accumulators into scratch
        int l = lane;
        asm volatile("" : "+v"(l));                    // (opaque: recomputed per pass instead of living across the passes)
        if constexpr (MF16) {
            // accumulator (mi = q, ni) register r of lane (j = l & 15, qd = l >> 4) = staged row 4 qd + r, column 16 ni + j
            const uint32_t c4 = (uint32_t)((l & 15) >> 2), w4 = (uint32_t)((l & 3) * 4);
            const uint32_t rowb = wslice + (uint32_t)((4 * (l >> 4)) * 256);
            const uint32_t base0 = rowb + c4 * 16 + w4, base1 = rowb + (c4 ^ 1u) * 16 + w4;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc16[MF16 ? q : 0][ni][r];
                    if (r & 1)
                        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base1), "v"(v),
                                     "n"(r * 256 + (((ni * 4) ^ (FUSE ? ((r & 2) << 2) : 0)) * 16)) : "memory");
                    else
                        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base0), "v"(v),
                                     "n"(r * 256 + (((ni * 4) ^ (FUSE ? ((r & 2) << 2) : 0)) * 16)) : "memory");
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            return;
        }
        const uint32_t c4 = (uint32_t)((l & 31) >> 2), w4 = (uint32_t)((l & 3) * 4);
        const uint32_t rowb = wslice + (uint32_t)((4 * (l >> 5)) * 256);
        const uint32_t base0 = rowb + c4 * 16 + w4, base1 = rowb + (c4 ^ 1u) * 16 + w4;      // g & 1 = 0 / 1
        // (a pass index past this instantiation's NPASS is only ever named by the fused epilogue, which the 64-channel instantiation
        //  compiles but never runs)
        if constexpr (q < NPASS)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                // g(row) of the staged row 8 (rr >> 2) + 4 h + (rr & 3): bits 0 and 1 of the row are those of rr
                const float v = acc[q >> 1][ni][(q & 1) * 8 + rr];
                if (rr & 1)
                    asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base1), "v"(v),
                                 "n"((8 * (rr >> 2) + (rr & 3)) * 256 + (((ni * 8) ^ (FUSE ? ((rr & 2) << 2) : 0)) * 16)) : "memory");
                else
                    asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base0), "v"(v),
                                 "n"((8 * (rr >> 2) + (rr & 3)) * 256 + (((ni * 8) ^ (FUSE ? ((rr & 2) << 2) : 0)) * 16)) : "memory");
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto read8 = [&](uint32_t a0, uint32_t a1, f32x4_t& f0, f32x4_t& f1) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f0), "=&v"(f1) : "v"(a0), "v"(a1) : "memory");
    };
    // Plain epilogue.  The bias is already in the accumulators (they start from it).  `MODE` 0: no residual, no mask -- no vector-memory
    // load in the whole epilogue, so nothing in it ever waits for memory: the 16 stores of a wave are fire-and-forget (through
    // compiler-managed loads every pass waited for `vmcnt(0)`, i.e. for the previous pass's STORES and the next tile's DMA pieces:
    // 6.5 us of epilogue per tile instead of ~3).  MODE 1 (residual) / 2 (mask): the operand rows of ALL eight passes are requested
    // before the first pass (64 registers: the fragment registers are dead and the accumulators free up pass by pass), so that no
    // load is ever queued behind a store.  Both at once (no layer of the network has it) is not taken by the plan.
    int st_issued = 0;                                 // stores this wave issued in the epilogue (wave-uniform; see the loop top)
    auto epilogue = [&](const Item& q, int sbuf, auto MODE, auto STATS) __attribute__((always_inline)) {
        constexpr int mode = decltype(MODE)::value;
        constexpr bool stats = decltype(STATS)::value;
        f32x2_t ssum[stats ? 4 : 1], ssq[stats ? 4 : 1];          // this lane's 8 channels over its 2 * NPASS rows
        if constexpr (stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ssum[j] = f32x2_t{0.f, 0.f}; ssq[j] = f32x2_t{0.f, 0.f}; }
        }
        const uint32_t wslice = lds0 + (uint32_t)(sbuf * P_BYTES + wave * 4096);
        const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
        const int cw0 = q.tn * BN + wn * 64;           // first output channel of this wave (wave-uniform)
        st_issued = 0;
        if (cw0 >= p.Cout) return;                     // (a Cout tile past the end: nothing to write)
        const int tile_base = ((q.n * p.Ho + q.y0) * p.Wo + q.x0) * p.Cout + cw0;
        const T* __restrict__ opg = mode == 1 ? resg : auxg;
        u32x4_t opr[mode ? 2 * NPASS : 1];
        auto request = [&](auto Q) __attribute__((always_inline)) {      // the operand rows of pass qq
            constexpr int qq = decltype(Q)::value;
            int ty, tx;
            pass_yx(wm * WROWS + (qq >> 1) * 32 + (qq & 1) * 16, ty, tx);
            const int so = tile_base + (ty * p.Wo + tx) * p.Cout;
            const int lo = (lane >> 3) * p.Cout + (lane & 7) * 8;
            opr[2 * qq] = *(const u32x4_t*)(opg + so + lo);
            opr[2 * qq + 1] = *(const u32x4_t*)(opg + so + 8 * p.Cout + lo);
        };
        // passes 0-3 before the first pass; passes 4-7 once the accumulators of passes 0-1 are out of their registers
        if constexpr (mode != 0) static_for<0, NPASS / 2>(request);
        static_for<0, NPASS>([&](auto Q) {
            constexpr int qq = decltype(Q)::value;
            int l = lane;
            asm volatile("" : "+v"(l));                   // (opaque per pass: nothing of a pass is hoisted above it)
            const int r0 = l >> 3, cg = l & 7;
            int ty, tx;
            pass_yx(wm * WROWS + (qq >> 1) * 32 + (qq & 1) * 16, ty, tx);
            const int so = tile_base + (ty * p.Wo + tx) * p.Cout;       // scalar
            const int lo = r0 * p.Cout + cg * 8;
            stage_pass(Q, wslice);
            const uint32_t ra = wslice + (uint32_t)(r0 * 256 + cg * 32), sw = (uint32_t)((r0 & 1) * 16);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4_t f0, f1;
                read8(ra + k * 2048 + sw, ra + k * 2048 + (16 - sw), f0, f1);
                f32x2_t v[4] = {{f0[0], f0[1]}, {f0[2], f0[3]}, {f1[0], f1[1]}, {f1[2], f1[3]}};
                if constexpr (mode == 1) {
                    const u32x4_t t = opr[2 * qq + k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += f32x2_t{bf16_bits_to_f32(t[j] & 0xffffu), __builtin_bit_cast(float, t[j] & 0xffff0000u)};
                }
                if (p.out_act == S2E_ACT_LRELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const f32x2_t s2 = v[j] * 0.2f; v[j] = f32x2_t{fmaxf(v[j][0], s2[0]), fmaxf(v[j][1], s2[1])}; }
                }
                if constexpr (mode == 2) {
                    const u32x4_t t = opr[2 * qq + k];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] *= f32x2_t{bf16_bits_to_f32(t[j] & 0xffffu) > 0.f ? 1.f : neg, __builtin_bit_cast(float, t[j] & 0xffff0000u) > 0.f ? 1.f : neg};
                }
                const u32x4_t packed = u32x4_t{pack2_bf16(v[0][0], v[0][1]), pack2_bf16(v[1][0], v[1][1]),
                                               pack2_bf16(v[2][0], v[2][1]), pack2_bf16(v[3][0], v[3][1])};
                *(u32x4_t*)(yg + so + k * 8 * p.Cout + lo) = packed;
                if constexpr (stats) {                    // of the ROUNDED values: what a pass over the stored tensor would see
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2_t f = {bf16_bits_to_f32(packed[j] & 0xffffu), __builtin_bit_cast(float, packed[j] & 0xffff0000u)};
                        ssum[j] += f; ssq[j] += f * f;
                    }
                }
            }
            if constexpr (mode != 0 && qq == NPASS / 4 - 1) { static_for<NPASS / 2, NPASS>(request); }
        });
        st_issued = 2 * NPASS;
        if constexpr (stats) {
            // fold the 8 row-lanes of a channel group through the wave's staging slice (fixed order: bit-reproducible), one
            // 8-byte store per channel: lane L = channel L of the wave's 64
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t la = wslice + (uint32_t)(lane * 64);          // [lane][{sum x 8 | sq x 8}]
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\tds_write_b128 %0, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                         :: "v"(la), "v"(f32x4_t{ssum[0][0], ssum[0][1], ssum[1][0], ssum[1][1]}), "v"(f32x4_t{ssum[2][0], ssum[2][1], ssum[3][0], ssum[3][1]}),
                            "v"(f32x4_t{ssq[0][0], ssq[0][1], ssq[1][0], ssq[1][1]}), "v"(f32x4_t{ssq[2][0], ssq[2][1], ssq[3][0], ssq[3][1]}) : "memory");
            // channel L = group cg = L >> 3, element e = L & 7: rows r0 = 0..7 are lanes r0 * 8 + cg
            const uint32_t ra = wslice + (uint32_t)(((lane >> 3) * 64) + (lane & 7) * 4);
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int r0 = 0; r0 < 8; ++r0) {
                float t0, t1;
                asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1) : "v"(ra), "n"(r0 * 8 * 64), "n"(r0 * 8 * 64 + 32) : "memory");
                a += t0; b += t1;
            }
            const int rect = (q.y0 / TH) * p.tiles_x + q.x0 / TW;
            const int slot = rect * (256 / WROWS) + wm;
            *(f32x2_t*)(p.spart + (((size_t)q.n * p.sP + slot) * p.Cout + cw0 + lane) * 2) = f32x2_t{a, b};
            st_issued += 1;
        }
    };

    // ---- FUSE epilogue (normalization.py:91-105,163-169,184-192 of the reference in one pass, see conv_patch.hip): a lane owns 8
    // channels of one pixel -- gamma in staged columns 8 cg .., beta 32 columns further -- reads x once and writes
    //     out = [lrelu] 0.5 * ((x - mean) * rstd * (1 + gamma) + beta + x * (1 + s0) + s1)
    // (and gamma itself when the backward pass will need it).  The accumulators START from 1 + b_gamma (gamma columns) and
    // b_beta + s1 (beta columns), so per element the modulation is three FMAs and a multiply with three per-channel constants
    //     out = x * sa' + (0.5 * B + ((x * rs' - mrs') * G)),   rs' = rstd / 2, mrs' = mean * rstd / 2, sa' = (1 + s0) / 2
    // derived once per tile by 48 threads from the raw constants the tile's LDS-DMA pieces brought (kcst).  The x rows of all eight
    // passes are requested before the first pass -- the only vector-memory loads of the epilogue, all issued before its first store.
    __shared__ __attribute__((aligned(16))) float kc[3][64];
    auto load_mod_consts = [&](int cbuf) __attribute__((always_inline)) {
        if (tid < 48) {
            const int which = tid >> 4, c4 = (tid & 15) * 4;
            const f32x4_t a = *(const f32x4_t*)&kcst[cbuf][256 + 2 * c4], b = *(const f32x4_t*)&kcst[cbuf][256 + 2 * c4 + 4];   // {m, r, m, r}
            f32x4_t v;
            if (which == 0) v = f32x4_t{0.5f * a[1], 0.5f * a[3], 0.5f * b[1], 0.5f * b[3]};                          // rstd / 2
            else if (which == 1) v = f32x4_t{0.5f * a[0] * a[1], 0.5f * a[2] * a[3], 0.5f * b[0] * b[1], 0.5f * b[2] * b[3]};   // mean * rstd / 2
            else {
                const f32x4_t s0 = *(const f32x4_t*)&kcst[cbuf][192 + c4];
                v = f32x4_t{0.5f + 0.5f * s0[0], 0.5f + 0.5f * s0[1], 0.5f + 0.5f * s0[2], 0.5f + 0.5f * s0[3]};
            }
            *(f32x4_t*)&kc[which][c4] = v;
        }
        __syncthreads();                                  // (the constants are read by every wave; once per tile)
    };
    auto epilogue_fused = [&](const Item& q, int sbuf) __attribute__((always_inline)) {
        const uint32_t wslice = lds0 + (uint32_t)(sbuf * P_BYTES + wave * 4096);
        const uint32_t kc0 = (uint32_t)(uintptr_t)(lptr_t)&kc[0][0];
        const T* __restrict__ mx = (const T*)p.mx;
        T* __restrict__ gout = (T*)p.mgamma;
        const int cw0 = q.tn * 64 + wn * 32;
        const int tile_base = ((q.n * p.Ho + q.y0) * p.Wo + q.x0) * p.mC + cw0;
        const int xbase = p.mup ? ((q.n * (p.Ho >> 1) + (q.y0 >> 1)) * (p.Wo >> 1) + (q.x0 >> 1)) * p.mC + cw0 : tile_base;
        u32x4_t xx[8];
        static_for<0, 8>([&](auto Q) {
            constexpr int qq = decltype(Q)::value;
            int ty, tx;
            pass_yx(wm * WROWS + (qq >> 1) * 32 + (qq & 1) * 16, ty, tx);
            const int row = lane >> 2, cg = lane & 3;
            const int so = p.mup ? xbase + ((ty >> 1) * (p.Wo >> 1) + (tx >> 1)) * p.mC : xbase + (ty * p.Wo + tx) * p.mC;
            const int lo = (p.mup ? (row >> 1) : row) * p.mC + cg * 8;
            xx[qq] = *(const u32x4_t*)(mx + so + lo);
        });
        static_for<0, 8>([&](auto Q) {
            constexpr int qq = decltype(Q)::value;
            int l = lane;
            asm volatile("" : "+v"(l));
            const int row = l >> 2, cg = l & 3;
            int ty, tx;
            pass_yx(wm * WROWS + (qq >> 1) * 32 + (qq & 1) * 16, ty, tx);
            const int so = tile_base + (ty * p.Wo + tx) * p.mC;         // scalar
            const int lo = row * p.mC + cg * 8;
            stage_pass(Q, wslice);
            const uint32_t g = (uint32_t)((row & 1) | ((row & 2) << 2));
            const uint32_t ga = wslice + (uint32_t)(row * 256) + (((uint32_t)(2 * cg) ^ g) << 4);
            const uint32_t ka = kc0 + (uint32_t)((wn * 32 + cg * 8) * 4);
            const u32x4_t xr = xx[qq];
            uint32_t ov[4], gv[4];
            static_for<0, 2>([&](auto HF) {
                constexpr int hf = decltype(HF)::value;
                f32x4_t g4, b4, krs, kmr, ksa;
                asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\t"
                             "ds_read_b128 %2, %7 offset:%8\n\tds_read_b128 %3, %7 offset:%9\n\tds_read_b128 %4, %7 offset:%10\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(g4), "=&v"(b4), "=&v"(krs), "=&v"(kmr), "=&v"(ksa)
                             : "v"(ga ^ (uint32_t)(hf * 16)), "v"(ga ^ (uint32_t)(128 + hf * 16)), "v"(ka), "n"(hf * 16), "n"(256 + hf * 16),
                               "n"(512 + hf * 16) : "memory");
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t xw = xr[hf * 2 + i];
                    const f32x2_t xf = {bf16_bits_to_f32(xw & 0xffffu), __builtin_bit_cast(float, xw & 0xffff0000u)};
                    const f32x2_t G = {g4[2 * i], g4[2 * i + 1]}, B = {b4[2 * i], b4[2 * i + 1]};
                    const f32x2_t rs = {krs[2 * i], krs[2 * i + 1]}, mr = {kmr[2 * i], kmr[2 * i + 1]}, sa = {ksa[2 * i], ksa[2 * i + 1]};
                    const f32x2_t t = xf * rs - mr;
                    f32x2_t v = xf * sa + (B * 0.5f + t * G);
                    if (p.mlrelu) { const f32x2_t s2 = v * 0.2f; v = f32x2_t{fmaxf(v[0], s2[0]), fmaxf(v[1], s2[1])}; }
                    ov[hf * 2 + i] = pack2_bf16(v[0], v[1]);
                    gv[hf * 2 + i] = pack2_bf16(G[0] - 1.f, G[1] - 1.f);
                }
            });
            *(u32x4_t*)(yg + so + lo) = u32x4_t{ov[0], ov[1], ov[2], ov[3]};
            if (gout) *(u32x4_t*)(gout + so + lo) = u32x4_t{gv[0], gv[1], gv[2], gv[3]};
        });
        st_issued = gout ? 16 : 8;
    };

    auto fetch_rect = [&](int id) __attribute__((always_inline)) -> int {
        if (p.rect_list && id < n_items) return p.rect_list[id / p.tiles_n];
        return -1;
    };

    Item cur = decode(item_id, fetch_rect(item_id));
    aim(cur);
    int pb = 0, cb = 0;
    prologue(cur, pb, cb);
    int tile_i = (int)blockIdx.x >= (G >> 1) ? 1 : 0;      // (parity of the tile count + which of the CU's two workgroups this is)
#ifdef S2E_DUO_STAMPS
    int dbg_i = 0;
    const bool dbg_on = p.dbg && tid == 0 && ((int)blockIdx.x == 0 || (int)blockIdx.x == (G >> 1) || (int)blockIdx.x == 8);
    long* dbg_row = p.dbg ? p.dbg + ((int)blockIdx.x == 0 ? 0 : (int)blockIdx.x == 8 ? 2 : 1) * 64 : nullptr;
    auto stamp = [&](int k) __attribute__((always_inline)) { if (dbg_on && dbg_i < 12) dbg_row[dbg_i * 5 + k] = __builtin_amdgcn_s_memrealtime(); };
#else
    auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
    for (;;) {
        stamp(0);
        // the prologue's DMA pieces were issued BEFORE the epilogue's stores: vector-memory operations retire in order, so once at
        // most `st_issued` are outstanding every piece has landed -- the stores themselves stay in flight across the barrier
        if (st_issued >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (st_issued >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(1);
        if (tile_i & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);     // (see "priority" at the top)
        const int nf = fetch_rect(item_id + G);            // label-sparse launch: the next item's rectangle, requested an item ahead
        // the accumulators start from the per-column constants of the epilogue (bias; FUSE: 1 + b_gamma | b_beta + s1)
        // (inline-asm reads: through a C++ access the compiler waits for vmcnt(0) first -- the constants arrived by LDS-DMA and it
        // cannot know they have landed -- which would drain the stores the wait above leaves in flight)
        if constexpr (MF16) {
            // lane (j = lane & 15)'s accumulator columns: plain 64 wn + 16 ni + j; FUSE gamma (ni 0, 1) / beta (ni 2, 3) of channel
            // 32 wn + 16 (ni & 1) + j
            const uint32_t ka = (uint32_t)(uintptr_t)(lptr_t)&kcst[cb][0] + (uint32_t)((FUSE ? wn * 32 + (lane & 15) : wn * 64 + (lane & 15)) * 4);
            float ci[4];
            if constexpr (FUSE) {
                float g0, g1, b0, b1, s0, s1;
                asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:64\n\tds_read_b32 %2, %6 offset:256\n\tds_read_b32 %3, %6 offset:320\n\t"
                             "ds_read_b32 %4, %6 offset:512\n\tds_read_b32 %5, %6 offset:576\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(g0), "=&v"(g1), "=&v"(b0), "=&v"(b1), "=&v"(s0), "=&v"(s1) : "v"(ka) : "memory");
                ci[0] = 1.f + g0; ci[1] = 1.f + g1; ci[2] = b0 + s0; ci[3] = b1 + s1;
            } else {
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:64\n\tds_read_b32 %2, %4 offset:128\n\tds_read_b32 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(ci[0]), "=&v"(ci[1]), "=&v"(ci[2]), "=&v"(ci[3]) : "v"(ka) : "memory");
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc16[MF16 ? mi : 0][ni] = f32x4_t{ci[ni], ci[ni], ci[ni], ci[ni]};
            // ---- the multiply loop, straight-line.  Two chunks = 18 K-steps per trip, fully unrolled: tap, chunk-of-the-pair,
            // weight stage ((t + 2) % 3: 18 is a multiple of 3) and B register set (t & 1) are compile-time facts of a K-step; what is
            // left at run time is the trip's first chunk c, the patch-buffer parity of the tile and `last` (no chunk after this
            // pair).  The round-4 stamps put the K-step's scalar skeleton -- ~60 instructions, a dozen branches: kt / 9, stage wrap,
            // the wait selection -- at 350 cycles of a 1400-cycle K-step; here a K-step has at most two (uniform) branches.
            const int tapb = 2 * p.Cin;                                     // bytes from one tap to the next in a weight row
            const int fbase = p.flip ? 8 * tapb : 0, fstep = p.flip ? -tapb : tapb;
            const uint32_t pbs0 = (uint32_t)(pb * P_BYTES), pbs1 = (uint32_t)((pb ^ 1) * P_BYTES);      // chunk c + cc lives in pbs[cc]
            auto dma_patch16 = [&](int r, int chunk, uint32_t bufb) __attribute__((always_inline)) {
                unsigned co = 64u * (unsigned)chunk;
                asm volatile("" : "+s"(co));                                  // (opaque: see the K-step)
                const unsigned off = aoff[r] == OOB ? OOB : aoff[r] + co;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(smem + bufb + (r * NW + wave) * 1024), 16, (int)off, 0, 0, 0);
            };
            auto dma_w16 = [&](int chunk, int tp, int stg) __attribute__((always_inline)) {
                int koff = fbase + tp * fstep + 64 * chunk;
                asm volatile("" : "+s"(koff));
#pragma unroll
                for (int j = 0; j < NBJ; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(smem + 2 * P_BYTES + stg * B_BYTES + (wave + NW * j) * 1024), 16,
                                                             (int)(woff[j] + (unsigned)koff), 0, 0, 0);
            };
            read_al(a16[0] + pbs0, 0);
            read_b(b16, 0);
            for (int c = 0; c < nch; c += 2) {
                const bool last = c + 2 >= nch;
                static_for<0, 2 * TAPS>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    constexpr int tap = t % TAPS, cc = t / TAPS, par = t & 1;
                    constexpr int ntap = (tap + 1) % TAPS, u = t + PD;
                    // (opaque per K-step: every address below is "loop-invariant + something" for the compiler, which would hoist
                    // all 18 x 10 sums out of the loop -- a hundred registers beside 128 accumulators -- if it could)
                    uint32_t pcb = cc ? pbs1 : pbs0, pnb = cc ? pbs0 : pbs1;                  // this chunk's / the next chunk's patch buffer
                    uint32_t nsb = (uint32_t)(((t + 1) % NBS) * B_BYTES);
                    int cs = c;
                    asm volatile("" : "+s"(pcb), "+s"(pnb), "+s"(nsb), "+s"(cs));
                    if (cs >= nch) return;                                  // (never taken; a block boundary per K-step: as ONE basic block of 576 MFMAs the
                                                                            //  trip is register-allocated with renamed accumulators and 100 spills)
#ifdef S2E_DUO_STAMPS
                    const int abl = p.abl;                                  // timing ablation (results wrong): 1 no loads, 2 no MFMAs, 4 no fragment reads, 8 no barrier
#else
                    constexpr int abl = 0;
#endif
                    // loads: a patch piece of the NEXT chunk during taps 0-5 (24 pieces of 16 pixels cover the 18 x 18 patch),
                    // the weights of K-step t + 2
                    if (!(abl & 1)) {
                        if constexpr (tap < NR16) { if (cc == 0 || !last) dma_patch16(tap, cs + cc + 1, pnb); }
                        if (u < 2 * TAPS || !last) dma_w16(cs + u / TAPS, u % TAPS, u % NBS);
                    }
                    if (!(abl & 4)) read_ah(a16[tap % 3] + pcb, tap / 3);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fal[0]), "+v"(fal[1]), "+v"(fal[2]), "+v"(fal[3]), "+v"(fbb[par][0]), "+v"(fbb[par][1]),
                                 "+v"(fbb[par][2]), "+v"(fbb[par][3]) :: "memory");
                    if (!(abl & 2)) mfma16(0, par);
                    // K-step t + 1 (and every older patch piece) has landed for this wave once all but THIS K-step's loads are back
                    constexpr int now_p = tap < NR16 ? 1 : 0, now_w = NBJ;
                    if constexpr (cc == 1 && (tap < NR16 || u >= 2 * TAPS)) {
                        if (last) wait_keep(u < 2 * TAPS ? now_w : 0); else wait_keep(now_p + now_w);
                    } else wait_keep(now_p + now_w);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fah[0]), "+v"(fah[1]), "+v"(fah[2]), "+v"(fah[3]) :: "memory");
                    if (!(abl & 8)) __builtin_amdgcn_s_barrier();
                    if (!(abl & 4)) {
                        if (t + 1 < 2 * TAPS || !last) {
                            read_al(a16[ntap % 3] + (ntap == 0 ? pnb : pcb), ntap / 3);
                            read_b(b16 + nsb, par ^ 1);
                        }
                    }
                    if (!(abl & 2)) mfma16(1, par);
                });
            }
        } else {
        float cinit[TN];
            {
                const uint32_t ka = (uint32_t)(uintptr_t)(lptr_t)&kcst[cb][0] + (uint32_t)((FUSE ? wn * 32 + l31 : wn * 64 + l31) * 4);
                float k0, k1, k2;
                if constexpr (FUSE) {
                    asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\tds_read_b32 %2, %3 offset:512\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(k0), "=&v"(k1), "=&v"(k2) : "v"(ka) : "memory");
                    cinit[0] = 1.f + k0; cinit[1] = k1 + k2;
                } else {
                    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)" : "=&v"(k0), "=&v"(k1) : "v"(ka) : "memory");
                    cinit[0] = k0; cinit[1] = k1;
                }
            }
    #pragma unroll
            for (int mi = 0; mi < TM; ++mi)
    #pragma unroll
                for (int ni = 0; ni < TN; ++ni)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = cinit[ni];
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
                    if (kt + PD < nk) { dma_w(kt + PD, stage == 0 ? NBS - 1 : stage - 1); issued += NBJ; }
                    read_frags(1, 1);
                    frags_ready(0, false);
                    mfmas(0);
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
        }
        // every buffer is free now (the last barrier is behind every LDS read): start the next item's loads, then write this one
        // out underneath them -- and, on this CU, underneath the partner workgroup's multiply loop
        stamp(2);
        __builtin_amdgcn_s_setprio(0);
        const int pbn = (pb + nch) & 1;
        const int next_id = item_id + G;
        const bool has_next = next_id < n_items;
        Item nxt = cur;
        if constexpr (FUSE) load_mod_consts(cb);
        if (has_next) { nxt = decode(next_id, nf); aim(nxt); prologue(nxt, pbn, cb ^ 1); }
        stamp(3);
        if constexpr (FUSE) epilogue_fused(cur, pbn ^ 1);
        else if constexpr (EPI == 0) epilogue(cur, pbn ^ 1, std::integral_constant<int, 0>{}, std::false_type{});
        else if constexpr (EPI == 1) epilogue(cur, pbn ^ 1, std::integral_constant<int, 1>{}, std::false_type{});
        else if constexpr (EPI == 2) epilogue(cur, pbn ^ 1, std::integral_constant<int, 2>{}, std::false_type{});
        else if constexpr (EPI == 3) epilogue(cur, pbn ^ 1, std::integral_constant<int, 0>{}, std::true_type{});
        else if constexpr (EPI == 4) epilogue(cur, pbn ^ 1, std::integral_constant<int, 1>{}, std::true_type{});
        else if (p.spart) {
            if (resg) epilogue(cur, pbn ^ 1, std::integral_constant<int, 1>{}, std::true_type{});
            else epilogue(cur, pbn ^ 1, std::integral_constant<int, 0>{}, std::true_type{});
        }
        else if (resg) epilogue(cur, pbn ^ 1, std::integral_constant<int, 1>{}, std::false_type{});
        else if (p.aux_mode != S2E_AUX_NONE) epilogue(cur, pbn ^ 1, std::integral_constant<int, 2>{}, std::false_type{});
        else epilogue(cur, pbn ^ 1, std::integral_constant<int, 0>{}, std::false_type{});
        stamp(4);
#ifdef S2E_DUO_STAMPS
        ++dbg_i;
#endif
        if (!has_next) break;
        cur = nxt; item_id = next_id; pb = pbn; cb ^= 1; ++tile_i;
    }
}

int duo_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// S2E_CONV_DUO = the work items (rectangle x 128-channel tile) a launch must have for this kernel to take it (default 512: one per
// workgroup of the 2-per-CU grid); 0 = never (conv_patch.hip runs everything, for A/B runs).
int duo_min_items() {
    static const int n = [] { const char* e = getenv("S2E_CONV_DUO"); return e ? atoi(e) : 256; }();
    return n;
}

template <bool FUSE>
int duo_launch(DuoParams& p, long rects_upper, hipStream_t st) {
    const long items = rects_upper * p.tiles_n;
    const int cap = 2 * duo_cu_count();
    const int grid = items < cap ? (int)items : cap;
#ifdef S2E_DUO_STAMPS
    static long* const dbg_ptr = [] { const char* e = getenv("S2E_DUO_DBG_PTR"); return e ? (long*)strtoull(e, nullptr, 0) : (long*)nullptr; }();
    p.dbg = dbg_ptr;
    static const int abl_env = [] { const char* e = getenv("S2E_DUO_ABL"); return e ? atoi(e) : 0; }();
    p.abl = abl_env;
#endif
    // S2E_DUO_MF16=0: the 32x32x16 loop everywhere (A/B switch)
    static const bool mf16 = [] { const char* e = getenv("S2E_DUO_MF16"); return e ? atoi(e) != 0 : true; }();
    if (!FUSE && p.Cout <= 64) conv_duo_kernel<false, 64><<<grid, 256, 0, st>>>(p);
    else if (!mf16) conv_duo_kernel<FUSE><<<grid, 256, 0, st>>>(p);
    else if (FUSE) conv_duo_kernel<true, 128, true><<<grid, 256, 0, st>>>(p);
    else {
        const int epi = p.aux_mode != S2E_AUX_NONE ? 2 : (p.res ? 1 : 0) + (p.spart ? 3 : 0);
        if (epi == 0) conv_duo_kernel<false, 128, true, 0><<<grid, 256, 0, st>>>(p);
        else if (epi == 1) conv_duo_kernel<false, 128, true, 1><<<grid, 256, 0, st>>>(p);
        else if (epi == 2) conv_duo_kernel<false, 128, true, 2><<<grid, 256, 0, st>>>(p);
        else if (epi == 3) conv_duo_kernel<false, 128, true, 3><<<grid, 256, 0, st>>>(p);
        else conv_duo_kernel<false, 128, true, 4><<<grid, 256, 0, st>>>(p);
    }
    S2E_CHECK_LAUNCH("conv_duo_kernel");
    return S2E_OK;
}

}  // namespace

// interior-only 16 x 16 rectangles tiling the map exactly: the epilogues address a pass's 16 rows as one scalar offset + a lane
// part and test no bounds, the 16x16x32 loop's fragments are one rectangle row each
static bool duo_rect_ok(int tw, int th, int H, int W) {
    return tw == 16 && th == 16 && H % 16 == 0 && W % 16 == 0;
}

// Shapes the duo kernel takes: bf16, 3x3, stride 1 (forward or data-gradient), no fused input activation, Cin a multiple of 32,
// Cout a multiple of 64 (128-channel tiles; a 64-channel layer runs the 64-wide instantiation), the rectangle plan of
// conv_patch.hip, no tanh, tensors under 2 GB, and at least S2E_CONV_DUO work items (long-K layers with few tiles -- split over
// channel chunks -- stay in conv_patch.hip too); Cout a multiple of 64; rectangles as duo_rect_ok; the caller keeps launches with a
// residual AND a mask (no layer of the network has both) in conv_patch.hip.
int s2e_conv_duo_plan(int dtype, const s2e_conv_desc* d, s2e_patch_plan* plan) {
    s2e_patch_plan local;
    if (!plan) plan = &local;
    if (duo_min_items() <= 0 || dtype != S2E_BF16) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->in_act != S2E_ACT_NONE || d->out_act == S2E_ACT_TANH) return 0;
    if (d->Cin % 32 != 0 || d->Cout % 8 != 0 || d->Cout < 64) return 0;
    const int grow = d->transposed ? 2 - 2 * d->pad : 2 * d->pad - 2;
    if (d->Ho != d->Hi + grow || d->Wo != d->Wi + grow) return 0;
    if ((long)d->N * d->Hi * d->Wi * d->Cin * 2 >= (1L << 31) || (long)d->N * d->Ho * d->Wo * d->Cout * 2 >= (1L << 31)) return 0;
    if ((long)s2e_conv_cout_pad(d->Cout) * s2e_conv_k_pad(dtype, 9 * d->Cin) * 2 >= (1L << 31)) return 0;
    // whole tiles only: the loop runs 64 input channels (two 32-channel K-steps x 9 taps) per trip and a tile's 128 (or 64) weight
    // rows are all live -- the network has no other widths (96 -> 192 went wrong here: tools/bench_tail.py's odd layer)
    if (d->Cin % 64 != 0 || (d->Cout != 64 && d->Cout % 128 != 0)) return 0;
    plan->splits = 1; plan->s2d = 0; plan->bn = 0;
    if (s2e_patch_rectangle(d, 3, &plan->tw, &plan->th) < 0.8) return 0;
    if (d->Ho % 16 == 0 && d->Wo % 16 == 0) plan->tw = plan->th = 16;       // the squarest rectangle: the smallest patch, and the 16x16x32 loop's
    if (!duo_rect_ok(plan->tw, plan->th, d->Ho, d->Wo)) return 0;
    const long rects = (long)d->N * (d->Ho / plan->th) * (d->Wo / plan->tw);
    return rects * (d->Cout <= 64 ? 1 : ceil_div(d->Cout, 128)) >= duo_min_items();
}

static void duo_fill(DuoParams* p, const s2e_conv_desc* d, const s2e_patch_plan* plan, int kpad) {
    p->N = d->N; p->Hi = d->Hi; p->Wi = d->Wi; p->Cin = d->Cin; p->Ho = d->Ho; p->Wo = d->Wo; p->Cout = d->Cout; p->Kpad = kpad;
    p->org = d->transposed ? d->pad - 2 : -d->pad;
    p->flip = d->transposed ? 1 : 0;
    p->out_act = d->out_act; p->aux_mode = d->aux_mode;
    p->tw = plan->tw; p->th = plan->th; p->sh = plan->tw <= 16 ? 1 : 2;
    p->tw_shift = (plan->tw & (plan->tw - 1)) == 0 ? __builtin_ctz(plan->tw) : -1;
    p->tiles_x = ceil_div(d->Wo, p->tw); p->tiles_y = ceil_div(d->Ho, p->th);
    p->rects = d->N * p->tiles_y * p->tiles_x;
    p->x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Cin * 2);
}

// InstanceNorm partial-sum slots per sample a launch of this shape writes (s2e_conv_duo_launch with stats_part): rectangles per
// sample x wave rows per tile.
int s2e_conv_duo_stats_slots(const s2e_conv_desc* d, const s2e_patch_plan* plan) {
    return (d->Ho / plan->th) * (d->Wo / plan->tw) * (d->Cout <= 64 ? 4 : 2);
}

int s2e_conv_duo_launch(const s2e_patch_plan* plan, const void* x, const void* w, const float* bias, const void* res,
                        const void* aux, void* y, const s2e_conv_desc* d, int kpad, float* stats_part, const int* rect_list,
                        const int* rect_count, hipStream_t st) {
    DuoParams p{};
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.aux = aux; p.y = y;
    p.rect_list = rect_list; p.rect_count = rect_count;
    p.spart = stats_part; p.sP = stats_part ? s2e_conv_duo_stats_slots(d, plan) : 0;
    duo_fill(&p, d, plan, kpad);
    p.tiles_n = d->Cout <= 64 ? 1 : ceil_div(d->Cout, 128);
    p.w_bytes = (unsigned)((long)s2e_conv_cout_pad(d->Cout) * kpad * 2);
    return duo_launch<false>(p, p.rects, st);
}

// The fused [gamma | beta] conv + modulation through the duo kernel: 1 = launched, 0 = not this shape (the caller runs
// conv_patch.hip's kernel), < 0 = error.  Same rectangles as conv_patch.hip's plan (tw, th given), so the label-sparse lists
// built for one serve the other.
int s2e_spade_conv_modulate_duo(int dtype, const void* actv, const void* w_packed, const float* bias, const void* x,
                                const float* stats, const float* style, int style_ld, void* out, void* gamma_out,
                                int N, int H, int W, int C, int nh, int lrelu, int flags, int tw, int th,
                                const int* rect_list, const int* rect_count, hipStream_t st) {
    if (duo_min_items() <= 0 || dtype != S2E_BF16 || nh % 32 != 0 || C % 64 != 0 || (flags & 1)) return 0;
    if (!duo_rect_ok(tw, th, H, W)) return 0;
    const long rects = (long)N * (H / th) * (W / tw);
    if (rects * (C / 64) < duo_min_items()) return 0;
    if ((long)N * H * W * nh * 2 >= (1L << 31) || (long)N * H * W * C * 2 >= (1L << 31)) return 0;
    const int kpad = ceil_div(9 * nh, 64) * 64;
    if ((long)s2e_conv_cout_pad(2 * C) * kpad * 2 >= (1L << 31)) return 0;
    const s2e_conv_desc d{N, H, W, nh, H, W, 2 * C, 3, 3, 1, 1, 0, S2E_ACT_NONE, S2E_ACT_NONE, S2E_AUX_NONE};
    const s2e_patch_plan plan{tw, th, 1, 0, 0};
    DuoParams p{};
    p.x = actv; p.w = w_packed; p.bias = bias; p.y = out;
    duo_fill(&p, &d, &plan, kpad);
    p.tiles_n = C / 64;
    p.w_bytes = (unsigned)((long)s2e_conv_cout_pad(2 * C) * kpad * 2);
    p.mx = x; p.mstats = stats; p.mstyle = style; p.msld = style_ld > 0 ? style_ld : 2 * C; p.mgamma = gamma_out;
    p.mC = C; p.mlrelu = lrelu; p.mup = (flags & 8) != 0;
    p.rect_list = rect_list; p.rect_count = rect_count;
    // (rects: an upper bound -- a sparse launch reads the count on the device)
    const int rc = duo_launch<true>(p, rects, st);
    return rc == S2E_OK ? 1 : rc;
}
