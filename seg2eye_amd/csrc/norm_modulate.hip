// HBM-bound kernels of the SPADE+Style block and the discriminator's InstanceNorm+LeakyReLU:
// per-(n,c) statistics, the fused modulation, its backward, and column sums (bias gradients).
//
// Layout: x is NHWC = [N][HW][C]; one thread owns a 16-byte channel group (8 bf16 / 4 f32) so every
// load/store is a full-width coalesced access, and walks rows (pixels).  Per-(n,c) reductions are
// fp32 per thread over a short row slab, combined across the block's row-threads in LDS, then added to
// an fp64 workspace with one atomic per (block, channel): no long fp32 chains, and the E[x^2]-E[x]^2
// finalisation happens in fp64 (cancellation-safe).
#include "common.h"
#include <stdlib.h>

static constexpr int kSlabIters = 16;     // row passes per block in the row-walking kernels
// Row passes per block of the row-walking reduction kernels.  A block ends by writing its per-channel partial sums to
// its own slot of the workspace (plain stores; a later tiny kernel adds the slots of a sample up in fp64, in a fixed order:
// the statistics are bit-reproducible, run to run and across data-parallel replicas).  Round 1 ended every block with fp64
// atomics on addresses shared by all blocks of the sample; the grid was then capped by their contention (in_stats of
// (8,256,256,128): 115 us with 2048 blocks, 30 us with 256).  Without atomics the optimum stays where it was -- few, long
// sequential row streams suit HBM better than many short ones: S2E_SLAB_BLOCKS blocks in total, same box, whole step:
// 348.1 / 343.7 / 341.7 / 342.1 img/s at 256 / 1024 / 2048 / 4096 (modulate_bwd 2.41 / 2.61 / 2.73 / 2.71 ms per step; with the
// atomics, at 256: 2.49) -- default 256, never fewer than 16 passes per block.
// x handed over at HALF resolution (the generator's nearest 2x upsampling folded into the consumers' reads, xw = W of the full map,
// 0 = x at full resolution): the pixel row of x that full-resolution pixel `pr` of sample n reads.  (pr + 0.5) * (1 / W) is at
// least 0.5 / W away from an integer: the fp32 product cannot land on the wrong side.
__device__ __forceinline__ size_t mod_x_row(int n, int pr, int HW, int xw, float inv_xw) {
    if (!xw) return (size_t)n * HW + pr;
    const int y = (int)(((float)pr + 0.5f) * inv_xw);
    const int xx = pr - y * xw;
    return (size_t)n * (HW >> 2) + (size_t)(y >> 1) * (xw >> 1) + (xx >> 1);
}

static int slab_iters_for(int HW, int rpp, int N, int zblocks) {
    const int target = 256;                          // (one block per CU: 1024 / 2048 / 4096 blocks measured slower)
    const int per_n = target / (N * zblocks) > 1 ? target / (N * zblocks) : 1;
    int it = ceil_div(HW, (long)rpp * per_n);
    if (it < kSlabIters) it = kSlabIters;
    return it;
}

struct RowGeom {       // how a 256-thread block maps onto [rows][C/VEC channel groups]
    int cg;            // channel groups in total (C / VEC)
    int cgb;           // channel groups per block (<= 256)
    int rpp;           // rows per pass
    int zblocks;       // blocks along the channel-group axis
};
static RowGeom row_geom(int C, int vec) {
    RowGeom g;
    g.cg = C / vec;
    g.cgb = g.cg < 256 ? g.cg : 256;
    g.rpp = 256 / g.cgb;
    g.zblocks = ceil_div(g.cg, g.cgb);
    return g;
}

// ------------------------------------------------------------------------------------ in_stats
// ------------------------------------------------------------------------------------ small maps: one launch
// A dependent kernel inside a replayed graph costs ~4.6 us however little it does, and the statistics of a small map were
// two of them (partial sums, finalize), a plain InstanceNorm + LeakyReLU three, its backward three more -- 100+ of a step's
// 918 launches.  For maps of up to S2E_IN_SMALL_HW pixels (default 1280: 34^2) ONE block owns all rows of (sample, 8 channel
// groups): thread = (channel group, one of 32 row lanes); fp32 sums per thread, the 32 lanes folded in fp64 in a fixed order
// (bit-reproducible), and -- APPLY -- the same block normalises its slice in a second pass that re-reads x from L2.
static int in_small_hw() {
    static const int v = [] { const char* e = getenv("S2E_IN_SMALL_HW"); return e ? atoi(e) : 1280; }();
    return v;
}

template <typename T, int APPLY, int G>          // G channel groups per block (8, or 4 when that is what fills the chip)
__global__ __launch_bounds__(256) void in_small_kernel(const T* __restrict__ x, T* __restrict__ out, double* __restrict__ ws,
                                                       float* __restrict__ stats, int HW, int C, float eps, int lrelu) {
    constexpr int VEC = Vec<T>::N, CH = G * VEC, RL = 256 / G;    // channels per block; row lanes
    __shared__ float red[RL][CH][2];
    __shared__ float mr[CH][2];
    const int tid = threadIdx.x, gx = tid % G, ry = tid / G;
    const int n = blockIdx.y, c0 = (blockIdx.x * G + gx) * VEC;
    const bool active = c0 < C;
    const T* base = x + (size_t)n * HW * C + c0;
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (active) {
        int r = ry;
        for (; r + 3 * RL < HW; r += 4 * RL) {                // four 16-byte loads in flight
            u32x4_t raw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[k] = *(const u32x4_t*)(base + (size_t)(r + RL * k) * C);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f[VEC];
                unpack16<T>(raw[k], f);
#pragma unroll
                for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
            }
        }
        for (; r < HW; r += RL) {
            float f[VEC];
            unpack16<T>(*(const u32x4_t*)(base + (size_t)r * C), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[ry][gx * VEC + j][0] = s[j]; red[ry][gx * VEC + j][1] = q[j]; }
    __syncthreads();
    if (tid < CH) {
        const int c = blockIdx.x * CH + tid;
        double S = 0.0, Q = 0.0;
        for (int k = 0; k < RL; ++k) { S += (double)red[k][tid][0]; Q += (double)red[k][tid][1]; }
        const double mean = S / HW;
        double var = Q / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
        mr[tid][0] = mf; mr[tid][1] = rf;
        if (c < C) {
            const size_t i = (size_t)n * C + c;
            if (ws) { ws[2 * i] = S; ws[2 * i + 1] = Q; }
            stats[2 * i] = mf; stats[2 * i + 1] = rf;
        }
    }
    if (!APPLY) return;
    __syncthreads();
    if (!active) return;
    float mu[VEC], rs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = mr[gx * VEC + j][0]; rs[j] = mr[gx * VEC + j][1]; }
    T* obase = out + (size_t)n * HW * C + c0;
    for (int r = ry; r < HW; r += 4 * RL) {
        u32x4_t raw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) raw[k] = *(const u32x4_t*)(base + (size_t)min(r + RL * k, HW - 1) * C);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (r + RL * k >= HW) continue;
            float f[VEC], o[VEC];
            unpack16<T>(raw[k], f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { o[j] = (f[j] - mu[j]) * rs[j]; if (lrelu) o[j] = lrelu02(o[j]); }
            *(u32x4_t*)(obase + (size_t)(r + RL * k) * C) = pack16<T>(o);
        }
    }
}

// backward of out = [lrelu]((x - mean) * rstd) for a small map, one launch:  go = g * lrelu'(xhat),
//   dx = rstd * (go - mean(go) - xhat * mean(go * xhat))
template <typename T, int G>
__global__ __launch_bounds__(256) void in_small_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x, const float* __restrict__ stats,
                                                           T* __restrict__ dx, int HW, int C, int lrelu) {
    constexpr int VEC = Vec<T>::N, CH = G * VEC, RL = 256 / G;
    // rows in flight per thread and tensor (round 5: four instead of two measured the same 19-20 us per call on the 33 x 33 maps --
    // the trips are few; the fixed parts -- statistics loads, the fp64 fold of 64 row lanes through LDS, two passes -- are the time)
    constexpr int UR = 2;
    __shared__ float red[RL][CH][2];
    __shared__ float mm[CH][2];
    const int tid = threadIdx.x, gx = tid % G, ry = tid / G;
    const int n = blockIdx.y, c0 = (blockIdx.x * G + gx) * VEC;
    const bool active = c0 < C;
    const size_t off = (size_t)n * HW * C + c0;
    float mu[VEC], rs[VEC], s0[VEC], s1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        mu[j] = active ? stats[((size_t)n * C + c0 + j) * 2] : 0.f;
        rs[j] = active ? stats[((size_t)n * C + c0 + j) * 2 + 1] : 0.f;
        s0[j] = 0.f; s1[j] = 0.f;
    }
    if (active)
        for (int r = ry; r < HW; r += UR * RL) {                 // UR rows x two tensors in flight
            u32x4_t rg[UR], rx[UR];
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                const size_t o = off + (size_t)min(r + RL * k, HW - 1) * C;
                rg[k] = *(const u32x4_t*)(g + o); rx[k] = *(const u32x4_t*)(x + o);
            }
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                if (r + RL * k >= HW) continue;
                float fg[VEC], fx[VEC];
                unpack16<T>(rg[k], fg); unpack16<T>(rx[k], fx);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (fx[j] - mu[j]) * rs[j];
                    const float go = (lrelu && xh < 0.f) ? 0.2f * fg[j] : fg[j];
                    s0[j] += go; s1[j] += go * xh;
                }
            }
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[ry][gx * VEC + j][0] = s0[j]; red[ry][gx * VEC + j][1] = s1[j]; }
    __syncthreads();
    if (tid < CH) {
        double A = 0.0, B = 0.0;
        for (int k = 0; k < RL; ++k) { A += (double)red[k][tid][0]; B += (double)red[k][tid][1]; }
        mm[tid][0] = (float)(A / HW); mm[tid][1] = (float)(B / HW);
    }
    __syncthreads();
    if (!active) return;
    float m0[VEC], m1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { m0[j] = mm[gx * VEC + j][0]; m1[j] = mm[gx * VEC + j][1]; }
    for (int r = ry; r < HW; r += UR * RL) {
        u32x4_t rg[UR], rx[UR];
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            const size_t o = off + (size_t)min(r + RL * k, HW - 1) * C;
            rg[k] = *(const u32x4_t*)(g + o); rx[k] = *(const u32x4_t*)(x + o);
        }
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            if (r + RL * k >= HW) continue;
            float fg[VEC], fx[VEC], o[VEC];
            unpack16<T>(rg[k], fg); unpack16<T>(rx[k], fx);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (fx[j] - mu[j]) * rs[j];
                const float go = (lrelu && xh < 0.f) ? 0.2f * fg[j] : fg[j];
                o[j] = rs[j] * (go - m0[j] - xh * m1[j]);
            }
            *(u32x4_t*)(dx + off + (size_t)(r + RL * k) * C) = pack16<T>(o);
        }
    }
}

// channel groups per block of the small-map kernels: 8, or 4 when 8 would leave most of the CUs without a block
static int small_groups(int N, int C, int vec) { return (long)N * ceil_div(C, 8 * vec) < 192 ? 4 : 8; }
#define S2E_SMALL_LAUNCH(KERNEL, TT, ...) do { \
    if (small_groups(N, C, vec) == 4) { const dim3 sg(ceil_div(C, 4 * vec), N); KERNEL<TT, 4><<<sg, 256, 0, st>>>(__VA_ARGS__); } \
    else { const dim3 sg(ceil_div(C, 8 * vec), N); KERNEL<TT, 8><<<sg, 256, 0, st>>>(__VA_ARGS__); } } while (0)
#define S2E_SMALL_LAUNCH2(KERNEL, TT, AP, ...) do { \
    if (small_groups(N, C, vec) == 4) { const dim3 sg(ceil_div(C, 4 * vec), N); KERNEL<TT, AP, 4><<<sg, 256, 0, st>>>(__VA_ARGS__); } \
    else { const dim3 sg(ceil_div(C, 8 * vec), N); KERNEL<TT, AP, 8><<<sg, 256, 0, st>>>(__VA_ARGS__); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void in_stats_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                               int HW, int C, int cg, int cgb, int rpp, int iters,
                                                               unsigned* __restrict__ counters, double* __restrict__ ws,
                                                               float* __restrict__ stats, float eps) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float red[256 * VEC * 2];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int tx = tid % cgb, ty = tid / cgb;
    const int g = blockIdx.z * cgb + tx;
    const int n = blockIdx.y;
    const bool active = ty < rpp && g < cg;
    float s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (active) {
        const int row0 = blockIdx.x * rpp * iters;
        const int rend = min(HW, row0 + rpp * iters);
        const T* base = x + (size_t)n * HW * C + (size_t)g * VEC;
        int row = row0 + ty;
        for (; row + 3 * rpp < rend; row += 4 * rpp) {              // 4 independent 16-B loads in flight
            const u32x4_t r0 = *(const u32x4_t*)(base + (size_t)row * C);
            const u32x4_t r1 = *(const u32x4_t*)(base + (size_t)(row + rpp) * C);
            const u32x4_t r2 = *(const u32x4_t*)(base + (size_t)(row + 2 * rpp) * C);
            const u32x4_t r3 = *(const u32x4_t*)(base + (size_t)(row + 3 * rpp) * C);
            float f0[VEC], f1[VEC], f2[VEC], f3[VEC];
            unpack16<T>(r0, f0); unpack16<T>(r1, f1); unpack16<T>(r2, f2); unpack16<T>(r3, f3);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                s[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
                q[j] += (f0[j] * f0[j] + f1[j] * f1[j]) + (f2[j] * f2[j] + f3[j] * f3[j]);
            }
        }
        for (; row < rend; row += rpp) {
            float f[VEC];
            unpack16<T>(*(const u32x4_t*)(base + (size_t)row * C), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[(tid * VEC + j) * 2] = s[j]; red[(tid * VEC + j) * 2 + 1] = q[j]; }
    __syncthreads();
    if (active && ty == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < rpp; ++r) {
                a += red[((r * cgb + tx) * VEC + j) * 2];
                b += red[((r * cgb + tx) * VEC + j) * 2 + 1];
            }
            float* w = part + (((size_t)n * gridDim.x + blockIdx.x) * C + g * VEC + j) * 2;     // slot [n][block][c]
            w[0] = a; w[1] = b;
        }
    }
    if (!counters) return;                                   // two-launch form: in_stats_finalize_kernel follows
    // One-launch form: the LAST of the gridDim.x blocks of this (sample, channel range) to get here folds the partial sums --
    // in block order, in fp64, exactly as the finalize kernel does: the same bits whichever block happens to be last.
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = counters + n * gridDim.z + blockIdx.z;
        const unsigned done = atomicAdd(cnt, 1u);
        is_last = done == gridDim.x - 1;
        if (is_last) *cnt = 0;                               // left zero for the next launch (hipGraph replays never re-zero it)
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const int c0 = blockIdx.z * cgb * VEC, c1 = min(C, c0 + cgb * VEC), P = gridDim.x;
    for (int c = c0 + tid; c < c1; c += 256) {
        double sm = 0.0, q = 0.0;
        for (int b = 0; b < P; ++b) {
            const float* w = part + (((size_t)n * P + b) * C + c) * 2;
            sm += (double)__builtin_nontemporal_load(w); q += (double)__builtin_nontemporal_load(w + 1);
        }
        const size_t i = (size_t)n * C + c;
        ws[2 * i] = sm; ws[2 * i + 1] = q;
        const double mean = sm / HW;
        double var = q / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ void in_stats_finalize_kernel(const float* __restrict__ part, double* __restrict__ ws, float* __restrict__ stats,
                                         int total, int C, int P, int HW, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i / C, c = i - n * C;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < P; ++b) {                            // fixed order: the statistics are bit-reproducible
        const float* w = part + (((size_t)n * P + b) * C + c) * 2;
        s += (double)w[0]; q += (double)w[1];
    }
    ws[2 * i] = s; ws[2 * i + 1] = q;                        // {sum x, sum x^2}: BatchNorm SPADE combines them over the batch
    const double mean = s / HW;
    double var = q / HW - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// {mean, rstd} from partial sums another kernel wrote (s2e_conv2d_stats: P slots per sample, hundreds): one block per (sample, 16
// channels); its 256 threads split the slots sixteen ways (each a fixed-order fp64 sum over its share, eight independent loads in
// flight), the sixteen shares are added in a fixed order: bit-reproducible.  (A thread per (sample, channel) walking all P slots
// -- in_stats_finalize_kernel's form, made for P ~ 32 -- took ~30 us at P = 512: eight blocks of serial strided loads.)
__global__ __launch_bounds__(256) void in_stats_from_partials_kernel(const float* __restrict__ part, double* __restrict__ ws,
                                                                     float* __restrict__ stats, int C, int P, int HW, float eps) {
    __shared__ double red[16][16][2];
    const int n = blockIdx.y, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, sl = threadIdx.x >> 4;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const int per = (P + 15) / 16, b0 = sl * per, b1 = min(P, b0 + per);
        const f32x2_t* base = (const f32x2_t*)(part + ((size_t)n * P * C + c) * 2);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            f32x2_t w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = base[(size_t)(b + k) * C];
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += (double)w[k][0]; q += (double)w[k][1]; }
        }
        for (; b < b1; ++b) { const f32x2_t w = base[(size_t)b * C]; s += (double)w[0]; q += (double)w[1]; }
    }
    red[sl][cl][0] = s; red[sl][cl][1] = q;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    s = 0.0; q = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s += red[k][cl][0]; q += red[k][cl][1]; }
    const size_t i = (size_t)n * C + c;
    ws[2 * i] = s; ws[2 * i + 1] = q;
    const double mean = s / HW;
    double var = q / HW - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

extern "C" int s2e_in_stats_from_partials(const float* part, int N, int P, int C, int HW, float eps, double* ws, float* stats,
                                          void* stream) {
    if (!part || !ws || !stats || N <= 0 || P <= 0 || C <= 0 || HW <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_in_stats_from_partials: bad argument");
    in_stats_from_partials_kernel<<<dim3(ceil_div(C, 16), N), 256, 0, (hipStream_t)stream>>>(part, ws, stats, C, P, HW, eps);
    S2E_CHECK_LAUNCH("in_stats_from_partials_kernel");
    return S2E_OK;
}

extern "C" size_t s2e_in_stats_workspace_bytes(int dtype, int N, int HW, int C) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (N <= 0 || HW <= 0 || C <= 0 || C % vec) return 0;
    const RowGeom g = row_geom(C, vec);
    const int P = ceil_div(HW, g.rpp * slab_iters_for(HW, g.rpp, N, g.zblocks));
    return (size_t)N * C * 2 * sizeof(double) + (size_t)N * P * C * 2 * sizeof(float);
}

extern "C" int s2e_in_stats_counters(int dtype, int N, int HW, int C) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (N <= 0 || HW <= 0 || C <= 0 || C % vec || HW <= in_small_hw()) return 0;
    return N * row_geom(C, vec).zblocks;
}

extern "C" int s2e_in_stats(int dtype, const void* x, int N, int HW, int C, float eps, double* ws, float* stats, unsigned* counters,
                            void* stream) {
    if (!x || !ws || !stats || N <= 0 || HW <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_in_stats: bad argument");
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_in_stats: bad dtype %d", dtype);
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_in_stats: C=%d not a multiple of %d", C, vec);
    hipStream_t st = (hipStream_t)stream;
    if (HW <= in_small_hw()) {                               // small map: statistics in one launch
        if (dtype == S2E_BF16) S2E_SMALL_LAUNCH2(in_small_kernel, bf16_t, 0, (const bf16_t*)x, nullptr, ws, stats, HW, C, eps, 0);
        else S2E_SMALL_LAUNCH2(in_small_kernel, float, 0, (const float*)x, nullptr, ws, stats, HW, C, eps, 0);
        S2E_CHECK_LAUNCH("in_small_kernel");
        return S2E_OK;
    }
    const RowGeom g = row_geom(C, vec);
    const int iters = slab_iters_for(HW, g.rpp, N, g.zblocks);
    const int P = ceil_div(HW, g.rpp * iters);
    dim3 grid(P, N, g.zblocks);
    float* part = (float*)(ws + (size_t)N * C * 2);          // [N][P][C][2] floats behind the N*C*2 doubles
    if (dtype == S2E_BF16) in_stats_partial_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, part, HW, C, g.cg, g.cgb, g.rpp, iters, counters, ws, stats, eps);
    else in_stats_partial_kernel<float><<<grid, 256, 0, st>>>((const float*)x, part, HW, C, g.cg, g.cgb, g.rpp, iters, counters, ws, stats, eps);
    S2E_CHECK_LAUNCH("in_stats_partial_kernel");
    if (counters) return S2E_OK;
    in_stats_finalize_kernel<<<ceil_div((long)N * C, 256), 256, 0, st>>>(part, ws, stats, N * C, C, P, HW, eps);
    S2E_CHECK_LAUNCH("in_stats_finalize_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ colsum
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ gsrc, float* __restrict__ out, long M, int C,
                                                     int cg, int cgb, int rpp, int rows_per_block) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float red[256 * VEC];
    const int tid = threadIdx.x;
    const int tx = tid % cgb, ty = tid / cgb;
    const int g = blockIdx.z * cgb + tx;
    const bool active = ty < rpp && g < cg;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    if (active) {
        const long row0 = (long)blockIdx.x * rows_per_block;
        const long rend = (row0 + rows_per_block < M) ? row0 + rows_per_block : M;
        const T* base = gsrc + (size_t)g * VEC;
        long row = row0 + ty;
        for (; row + 3L * rpp < rend; row += 4L * rpp) {          // 4 independent 16-B loads in flight
            u32x4_t r0 = *(const u32x4_t*)(base + (size_t)row * C);
            u32x4_t r1 = *(const u32x4_t*)(base + (size_t)(row + rpp) * C);
            u32x4_t r2 = *(const u32x4_t*)(base + (size_t)(row + 2L * rpp) * C);
            u32x4_t r3 = *(const u32x4_t*)(base + (size_t)(row + 3L * rpp) * C);
            float f0[VEC], f1[VEC], f2[VEC], f3[VEC];
            unpack16<T>(r0, f0); unpack16<T>(r1, f1); unpack16<T>(r2, f2); unpack16<T>(r3, f3);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
        }
        for (; row < rend; row += rpp) {
            float f[VEC];
            unpack16<T>(*(const u32x4_t*)(base + (size_t)row * C), f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[tid * VEC + j] = s[j];
    __syncthreads();
    if (active && ty == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float a = 0.f;
            for (int r = 0; r < rpp; ++r) a += red[(r * cgb + tx) * VEC + j];
            atomicAdd(out + g * VEC + j, a);
        }
    }
}
template <typename T>
__global__ void colsum_scalar_kernel(const T* __restrict__ gsrc, float* __restrict__ out, long M, int C) {
    // any-C fallback (C not a multiple of the vector width, e.g. the 1-channel heads)
    __shared__ float red[256];
    const int c = blockIdx.y;
    float s = 0.f;
    for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < M; row += (long)gridDim.x * blockDim.x)
        s += load1<T>(gsrc + (size_t)row * C + c);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + c, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int s2e_colsum(int dtype, const void* g, long M, int C, float* out, void* stream) {
    if (!g || !out || M <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_colsum: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_colsum: bad dtype %d", dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    hipStream_t st = (hipStream_t)stream;
    if (C % vec) {
        dim3 grid((unsigned)(M / 1024 + 1 < 256 ? M / 1024 + 1 : 256), C);
        if (dtype == S2E_BF16) colsum_scalar_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)g, out, M, C);
        else colsum_scalar_kernel<float><<<grid, 256, 0, st>>>((const float*)g, out, M, C);
        S2E_CHECK_LAUNCH("colsum_scalar_kernel");
        return S2E_OK;
    }
    const RowGeom rg = row_geom(C, vec);
    const int rows_per_block = rg.rpp * 64;      // long slabs: same-address atomics serialise, keep them few
    dim3 grid(ceil_div(M, rows_per_block), 1, rg.zblocks);
    if (dtype == S2E_BF16) colsum_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)g, out, M, C, rg.cg, rg.cgb, rg.rpp, rows_per_block);
    else colsum_kernel<float><<<grid, 256, 0, st>>>((const float*)g, out, M, C, rg.cg, rg.cgb, rg.rpp, rows_per_block);
    S2E_CHECK_LAUNCH("colsum_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ modulation forward
// pre = sc*(xhat*G + beta) + ssc*(x*a + b);   SPADE_STYLE: sc = ssc = 0.5, G = 1+gamma, a = 1+s0, b = s1
//                                             PLAIN_IN   : sc = 1, ssc = 0, G = 1, beta = 0
template <typename T, int MODE>
__global__ __launch_bounds__(256) void modulate_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gb,
                                                           const float* __restrict__ stats, const float* __restrict__ style,
                                                           T* __restrict__ out, int HW, int C, int cg, int cg_shift, int lrelu, int sld) {
    constexpr int VEC = Vec<T>::N;
    // One sample per grid row (32-bit indices, no 64-bit divisions).  When the grid stride is a multiple of cg (cg a
    // power of two <= 256: every real layer) a thread keeps its channel group over the whole loop, and its per-channel
    // constants -- mean, rstd, 1 + s0, s1: 8 of the 11 loads of an iteration -- are loaded once.
    const int n = blockIdx.y;
    const int vps = HW * cg;                               // 16-byte vectors per sample
    const bool fixed_g = cg_shift >= 0 && cg <= 256;
    float mu[VEC], rs[VEC], sa[VEC], sb[VEC];
    auto load_consts = [&](int c0) __attribute__((always_inline)) {
        const float* stp = stats + ((size_t)n * C + c0) * 2;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { mu[j] = stp[2 * j]; rs[j] = stp[2 * j + 1]; }
        if (MODE == S2E_NORM_SPADE_STYLE) {
            const float* s0 = style + (size_t)n * sld + c0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) { sa[j] = 1.f + s0[j]; sb[j] = s0[C + j]; }
        }
    };
    if (fixed_g) load_consts((threadIdx.x & (cg - 1)) * VEC);
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vps; v += gridDim.x * blockDim.x) {
        const int prow = cg_shift >= 0 ? v >> cg_shift : v / cg;
        const int g = v - prow * cg;
        const size_t row = (size_t)n * HW + prow;
        const int c0 = g * VEC;
        if (!fixed_g) load_consts(c0);
        float f[VEC], o[VEC];
        unpack16<T>(*(const u32x4_t*)(x + row * C + c0), f);
        if (MODE == S2E_NORM_SPADE_STYLE) {
            float ga[VEC], be[VEC];
            unpack16<T>(*(const u32x4_t*)(gb + row * 2 * C + c0), ga);
            unpack16<T>(*(const u32x4_t*)(gb + row * 2 * C + C + c0), be);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (f[j] - mu[j]) * rs[j];
                o[j] = 0.5f * (xh * (1.f + ga[j]) + be[j] + f[j] * sa[j] + sb[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (f[j] - mu[j]) * rs[j];
        }
        if (lrelu) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = lrelu02(o[j]);
        }
        *(u32x4_t*)(out + row * C + c0) = pack16<T>(o);
    }
}

extern "C" int s2e_modulate_fwd(int dtype, int mode, const void* x, const void* gb, const float* stats, const float* style,
                                void* out, int N, int HW, int C, int lrelu, int style_ld, void* stream) {
    if (!x || !stats || !out || N <= 0 || HW <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_fwd: bad argument");
    if (mode == S2E_NORM_SPADE_STYLE && (!gb || !style)) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_fwd: SPADE_STYLE needs gb and style");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_fwd: bad dtype %d", dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_modulate_fwd: C=%d not a multiple of %d", C, vec);
    const int cg = C / vec;
    const int sld = style_ld > 0 ? style_ld : 2 * C;
    if ((long)HW * cg >= (1L << 31)) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_modulate_fwd: sample too large for 32-bit indices");
    const int vps = HW * cg;
    int gx = (vps + 255) / 256;
    const int gx_cap = 8192 / N > 1 ? 8192 / N : 1;
    if (gx > gx_cap) gx = gx_cap;
    const dim3 grid(gx, N);
    int cg_shift = -1;
    for (int b = 0; b < 31; ++b) if ((1 << b) == cg) cg_shift = b;
    hipStream_t st = (hipStream_t)stream;
#define S2E_LAUNCH_MOD(TT, MM) modulate_fwd_kernel<TT, MM><<<grid, 256, 0, st>>>((const TT*)x, (const TT*)gb, stats, style, (TT*)out, HW, C, cg, cg_shift, lrelu, sld)
    if (dtype == S2E_BF16) { if (mode == S2E_NORM_SPADE_STYLE) S2E_LAUNCH_MOD(bf16_t, S2E_NORM_SPADE_STYLE); else S2E_LAUNCH_MOD(bf16_t, S2E_NORM_PLAIN_IN); }
    else { if (mode == S2E_NORM_SPADE_STYLE) S2E_LAUNCH_MOD(float, S2E_NORM_SPADE_STYLE); else S2E_LAUNCH_MOD(float, S2E_NORM_PLAIN_IN); }
#undef S2E_LAUNCH_MOD
    S2E_CHECK_LAUNCH("modulate_fwd_kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ modulation backward
// pass 1 (row-walking): go = g*lrelu'(pre); SPADE: dgamma = 0.5*go*xhat, dbeta = 0.5*go -> dgb;
//   per-(n,c) sums  S0 = sum gn, S1 = sum gn*xhat (gn = 0.5*go*G | go),  S2 = sum go*x, S3 = sum go
// pass 2 (elementwise): dx = 0.5*go*a + rstd*(gn - S0/HW - xhat*S1/HW)   (PLAIN: no style term);
//   SPADE reads go back as 2*dbeta instead of recomputing the LeakyReLU mask from gamma/beta.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void modulate_bwd_reduce_kernel(const T* __restrict__ gin, const T* __restrict__ x,
        const T* __restrict__ gb, const float* __restrict__ stats, const float* __restrict__ style,
        T* __restrict__ dgb, float* __restrict__ part, int HW, int C, int cg, int cgb, int rpp, int lrelu, int sld, int iters,
        const T* __restrict__ fout, int xw, float inv_xw) {
    // fout != NULL (S2E_NORM_GAMMA_ONLY): gb holds gamma alone, (N,HW,C); the LeakyReLU mask comes from the sign of the
    // forward's OUTPUT fout (LeakyReLU keeps the sign of its argument) instead of recomputing it from gamma and beta
    constexpr int VEC = Vec<T>::N;
    constexpr int NS = (MODE == S2E_NORM_SPADE_STYLE) ? 4 : 2;
    const int gst = fout ? C : 2 * C;                      // pixels of gb are this many elements apart
    __shared__ float red[256 * VEC * NS];
    const int tid = threadIdx.x;
    const int tx = tid % cgb, ty = tid / cgb;
    const int g = blockIdx.z * cgb + tx;
    const int n = blockIdx.y;
    const bool active = ty < rpp && g < cg;
    float S[NS][VEC];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) S[k][j] = 0.f;
    if (active) {
        const int c0 = g * VEC;
        float mu[VEC], rs[VEC], a[VEC], b[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            mu[j] = stats[((size_t)n * C + c0 + j) * 2];
            rs[j] = stats[((size_t)n * C + c0 + j) * 2 + 1];
            if (MODE == S2E_NORM_SPADE_STYLE) {
                a[j] = 1.f + style[(size_t)n * sld + c0 + j];
                b[j] = style[(size_t)n * sld + C + c0 + j];
            }
        }
        const int row0 = blockIdx.x * rpp * iters;
        const int pend = min(HW, row0 + rpp * iters);
        // one row: everything after the loads (the loads of TWO rows are issued before either is consumed)
        auto consume = [&](size_t row, u32x4_t rx, u32x4_t rg, u32x4_t rga, u32x4_t rbe) __attribute__((always_inline)) {
            float f[VEC], gg[VEC];
            unpack16<T>(rx, f);
            unpack16<T>(rg, gg);
            if (MODE == S2E_NORM_SPADE_STYLE) {
                float ga[VEC], be[VEC], dga[VEC], dbe[VEC];
                unpack16<T>(rga, ga);
                unpack16<T>(rbe, be);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (f[j] - mu[j]) * rs[j];
                    const float G = 1.f + ga[j];
                    float go = gg[j];
                    if (lrelu) {
                        const float pre = fout ? be[j] : 0.5f * (xh * G + be[j] + f[j] * a[j] + b[j]);
                        go *= (pre > 0.f ? 1.f : 0.2f);
                    }
                    dga[j] = 0.5f * go * xh;
                    dbe[j] = 0.5f * go;
                    const float gn = 0.5f * go * G;
                    S[0][j] += gn; S[1][j] += gn * xh; S[2][j] += go * f[j]; S[3][j] += go;
                }
                *(u32x4_t*)(dgb + row * 2 * C + c0) = pack16<T>(dga);
                *(u32x4_t*)(dgb + row * 2 * C + C + c0) = pack16<T>(dbe);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (f[j] - mu[j]) * rs[j];
                    float go = gg[j];
                    if (lrelu) go *= (xh > 0.f ? 1.f : 0.2f);
                    S[0][j] += go; S[1][j] += go * xh;
                }
            }
        };
        const u32x4_t zero4 = {0u, 0u, 0u, 0u};
        int pr = row0 + ty;
        // UR rows per trip: the loads of all UR rows (up to 4 * UR 16-byte loads) are issued before any row is consumed.  One
        // block per CU and 2 rows in flight kept 32 KB per CU on the way -- half of what the HBM latency-bandwidth product asks.
        constexpr int UR = 4;
        for (; pr + (UR - 1) * rpp < pend; pr += UR * rpp) {
            u32x4_t xs[UR], gs[UR], as[UR], bs[UR];
            size_t rr[UR];
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                rr[k] = (size_t)n * HW + pr + k * rpp;
                xs[k] = *(const u32x4_t*)(x + mod_x_row(n, pr + k * rpp, HW, xw, inv_xw) * C + c0);
                gs[k] = *(const u32x4_t*)(gin + rr[k] * C + c0);
                as[k] = zero4; bs[k] = zero4;
                if (MODE == S2E_NORM_SPADE_STYLE) {
                    as[k] = *(const u32x4_t*)(gb + rr[k] * gst + c0);
                    if (fout) { if (lrelu) bs[k] = *(const u32x4_t*)(fout + rr[k] * C + c0); }
                    else bs[k] = *(const u32x4_t*)(gb + rr[k] * gst + C + c0);
                }
            }
#pragma unroll
            for (int k = 0; k < UR; ++k) consume(rr[k], xs[k], gs[k], as[k], bs[k]);
        }
        for (; pr + rpp < pend; pr += 2 * rpp) {
            const size_t r0 = (size_t)n * HW + pr, r1 = r0 + rpp;
            const u32x4_t x0 = *(const u32x4_t*)(x + mod_x_row(n, pr, HW, xw, inv_xw) * C + c0), x1 = *(const u32x4_t*)(x + mod_x_row(n, pr + rpp, HW, xw, inv_xw) * C + c0);
            const u32x4_t g0 = *(const u32x4_t*)(gin + r0 * C + c0), g1 = *(const u32x4_t*)(gin + r1 * C + c0);
            u32x4_t a0 = zero4, b0 = zero4, a1 = zero4, b1 = zero4;
            if (MODE == S2E_NORM_SPADE_STYLE) {
                a0 = *(const u32x4_t*)(gb + r0 * gst + c0); a1 = *(const u32x4_t*)(gb + r1 * gst + c0);
                if (fout) { if (lrelu) { b0 = *(const u32x4_t*)(fout + r0 * C + c0); b1 = *(const u32x4_t*)(fout + r1 * C + c0); } }
                else { b0 = *(const u32x4_t*)(gb + r0 * gst + C + c0); b1 = *(const u32x4_t*)(gb + r1 * gst + C + c0); }
            }
            consume(r0, x0, g0, a0, b0);
            consume(r1, x1, g1, a1, b1);
        }
        if (pr < pend) {
            const size_t r0 = (size_t)n * HW + pr;
            u32x4_t a0 = zero4, b0 = zero4;
            if (MODE == S2E_NORM_SPADE_STYLE) {
                a0 = *(const u32x4_t*)(gb + r0 * gst + c0);
                if (fout) { if (lrelu) b0 = *(const u32x4_t*)(fout + r0 * C + c0); }
                else b0 = *(const u32x4_t*)(gb + r0 * gst + C + c0);
            }
            consume(r0, *(const u32x4_t*)(x + mod_x_row(n, pr, HW, xw, inv_xw) * C + c0), *(const u32x4_t*)(gin + r0 * C + c0), a0, b0);
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[(tid * VEC + j) * NS + k] = S[k][j];
    __syncthreads();
    if (active && ty == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float acc = 0.f;
                for (int r = 0; r < rpp; ++r) acc += red[((r * cgb + tx) * VEC + j) * NS + k];
                part[(((size_t)n * gridDim.x + blockIdx.x) * C + g * VEC + j) * 4 + k] = acc;      // slot [n][block][c][k]
            }
    }
}

// Per-(n,c) coefficients of pass 2, computed once by modulate_bwd_coef_kernel and stored as one float4 per channel in
// the tail of ws (behind the N*C*4 fp64 sums):
//   SPADE_STYLE: dx = dbeta*(P + R*gamma) - Q - x*S     P = 1 + s0 + rstd, R = rstd
//   PLAIN_IN   : dx = R*go - Q - x*S                    P = mean (for the LeakyReLU mask: xhat > 0 <=> x > mean)
//   both       : Q = rstd*(S0 - mean*rstd*S1)/HW,  S = rstd^2*S1/HW
// The element-wise pass then needs 16 B of constants per channel instead of 32 B of doubles + stats + style and no
// fp64 conversions: it was VALU-issue-bound (22 loads and ~350 instructions per 16-byte vector), not HBM-bound.
// batch != 0 (BatchNorm SPADE: statistics over the whole batch): S0, S1 are summed over the samples and HW -> N*HW.
// ws: (N,C,4) fp64 sums.  part != NULL: they are first formed here from the blocks' partial slots ([N][P][C][4] floats, added in
// a fixed order: bit-reproducible); part == NULL: ws already holds them (second stage of s2e_modulate_bwd_staged).
template <int MODE>
__global__ void modulate_bwd_sums_kernel(const float* __restrict__ part, double* __restrict__ ws, int N, int C, int P) {
    constexpr int NS = (MODE == S2E_NORM_SPADE_STYLE) ? 4 : 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < P; ++b) {
        const float* w = part + (((size_t)n * P + b) * C + c) * 4;
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] += (double)w[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ws[(size_t)i * 4 + k] = s[k];
}

template <int MODE>
__global__ void modulate_bwd_coef_kernel(double* __restrict__ ws, const float* __restrict__ part, int P, f32x4_t* __restrict__ coef, const float* __restrict__ stats,
                                         const float* __restrict__ style, float* __restrict__ dstyle, int N, int C, int HW, int sld, int batch,
                                         double batch_count) {
    constexpr int NS = (MODE == S2E_NORM_SPADE_STYLE) ? 4 : 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    if (part && !batch) {                                    // per-sample statistics: this thread's own (n, c) sums
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        int b = 0;
        for (; b + 8 <= P; b += 8) {                         // eight 16-byte slots in flight, added in slot order (one by one every
            f32x4_t w[8];                                    //  load waited for the last: 11 us per call for 32 slots)
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = *(const f32x4_t*)(part + (((size_t)n * P + b + j) * C + c) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] += (double)w[j][k];
        }
        for (; b < P; ++b) {
            const float* w = part + (((size_t)n * P + b) * C + c) * 4;
#pragma unroll
            for (int k = 0; k < NS; ++k) s[k] += (double)w[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ws[(size_t)i * 4 + k] = s[k];
    }
    double s0d = ws[(size_t)i * 4], s1d = ws[(size_t)i * 4 + 1];
    float inv_hw = 1.f / (float)HW;
    if (batch) {
        s0d = 0.0; s1d = 0.0;
        for (int m = 0; m < N; ++m) { s0d += ws[((size_t)m * C + c) * 4]; s1d += ws[((size_t)m * C + c) * 4 + 1]; }
        inv_hw = (float)(1.0 / (batch_count > 0.0 ? batch_count : (double)HW * (double)N));
    }
    const float mean = stats[2 * i], rs = stats[2 * i + 1];
    const float m0 = (float)s0d * inv_hw, m1 = (float)s1d * inv_hw;
    f32x4_t k;
    if (MODE == S2E_NORM_SPADE_STYLE) {
        const double s2d = ws[(size_t)i * 4 + 2], s3d = ws[(size_t)i * 4 + 3];
        dstyle[(size_t)n * sld + c] += 0.5f * (float)s2d;
        dstyle[(size_t)n * sld + C + c] += 0.5f * (float)s3d;
        k[0] = 1.f + style[(size_t)n * sld + c] + rs;
    } else {
        k[0] = mean;
    }
    k[1] = rs;
    k[2] = rs * (m0 - mean * rs * m1);
    k[3] = rs * rs * m1;
    coef[i] = k;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void modulate_bwd_apply_kernel(const T* __restrict__ gin, const T* __restrict__ x,
        const T* __restrict__ gb, const T* __restrict__ dgb, const f32x4_t* __restrict__ coef, T* dx, const T* dxa,
        int vps, int HW, int C, int cg, int cg_shift, int lrelu, int acc, int gst, int xw, float inv_xw) {
    constexpr int VEC = Vec<T>::N;
    const int n = blockIdx.y;                              // one sample per grid row: 32-bit indices, no 64-bit division
    // The channel group of a thread does not change over the grid-stride loop when the stride is a multiple of cg
    // (cg a power of two <= 256: every real layer): its 8 x float4 coefficients are then loaded ONCE.  Loaded per vector
    // they are 8 of the 11 loads of an iteration -- 128 B of cache traffic per lane for 48 B of HBM data -- and the kernel
    // ran at 60 % of the HBM rate.
    const bool fixed_g = cg_shift >= 0 && cg <= 256;
    f32x4_t K[VEC];
    if (fixed_g) {
        const f32x4_t* kp = coef + (size_t)n * C + (threadIdx.x & (cg - 1)) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) K[j] = kp[j];
    }
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vps; v += gridDim.x * blockDim.x) {
        const int prow = cg_shift >= 0 ? v >> cg_shift : v / cg;
        const int g = v - prow * cg;
        const size_t row = (size_t)n * HW + prow;
        const int c0 = g * VEC;
        float f[VEC], o[VEC];
        unpack16<T>(*(const u32x4_t*)(x + mod_x_row(n, prow, HW, xw, inv_xw) * C + c0), f);
        if (!fixed_g) {
            const f32x4_t* kp = coef + (size_t)n * C + c0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) K[j] = kp[j];
        }
        if (MODE == S2E_NORM_SPADE_STYLE) {
            float ga[VEC], dbe[VEC];
            unpack16<T>(*(const u32x4_t*)(gb + row * gst + c0), ga);
            unpack16<T>(*(const u32x4_t*)(dgb + row * 2 * C + C + c0), dbe);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = dbe[j] * (K[j][0] + K[j][1] * ga[j]) - K[j][2] - f[j] * K[j][3];
        } else {
            float gg[VEC];
            unpack16<T>(*(const u32x4_t*)(gin + row * C + c0), gg);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float go = gg[j];
                if (lrelu) go *= (f[j] > K[j][0] ? 1.f : 0.2f);
                o[j] = K[j][1] * go - K[j][2] - f[j] * K[j][3];
            }
        }
        if (acc) {                                         // dxa (= dx, or the tensor the relay must leave alone) holds another consumer's gradient of the same x
            float prev[VEC];
            unpack16<T>(*(const u32x4_t*)(dxa + row * C + c0), prev);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] += prev[j];
        }
        *(u32x4_t*)(dx + row * C + c0) = pack16<T>(o);
    }
}

// The element-wise pass when x is the half-resolution tensor (xw != 0) AND the caller wants the gradient w.r.t. THAT tensor:
// dx_low[n][y][x] = sum over the 2 x 2 full-resolution pixels it was replicated to (the nearest upsampling's backward folded in:
// the full-resolution dx never exists).  One thread per low-resolution 16-byte vector; acc adds into dx_low.
template <typename T>
__global__ __launch_bounds__(256) void modulate_bwd_apply_quad_kernel(const T* __restrict__ x, const T* __restrict__ gb, const T* __restrict__ dgb,
        const f32x4_t* __restrict__ coef, T* dx, const T* dxa, int HW, int C, int cg, int acc, int gst, int xw, float inv_wl) {
    constexpr int VEC = Vec<T>::N;
    const int n = blockIdx.y, wl = xw >> 1, hwl = HW >> 2;
    const int vps = hwl * cg;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vps; v += gridDim.x * blockDim.x) {
        const int pl = v / cg, g = v - pl * cg, c0 = g * VEC;
        const int yl = (int)(((float)pl + 0.5f) * inv_wl), xl = pl - yl * wl;
        const size_t lrow = (size_t)n * hwl + pl;
        f32x4_t K[VEC];
        const f32x4_t* kp = coef + (size_t)n * C + c0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) K[j] = kp[j];
        float f[VEC], o[VEC];
        unpack16<T>(*(const u32x4_t*)(x + lrow * C + c0), f);
        u32x4_t rga[4], rdb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t row = (size_t)n * HW + (size_t)(2 * yl + (q >> 1)) * xw + 2 * xl + (q & 1);
            rga[q] = *(const u32x4_t*)(gb + row * gst + c0);
            rdb[q] = *(const u32x4_t*)(dgb + row * 2 * C + C + c0);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = -4.f * (K[j][2] + f[j] * K[j][3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ga[VEC], dbe[VEC];
            unpack16<T>(rga[q], ga); unpack16<T>(rdb[q], dbe);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] += dbe[j] * (K[j][0] + K[j][1] * ga[j]);
        }
        if (acc) {
            float prev[VEC];
            unpack16<T>(*(const u32x4_t*)(dxa + lrow * C + c0), prev);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] += prev[j];
        }
        *(u32x4_t*)(dx + lrow * C + c0) = pack16<T>(o);
    }
}

// SPADE+Style modulation backward of a SMALL map in one launch (gamma-only form: the fused forward's saved gamma, the
// LeakyReLU mask from the sign of its output): same block shape as in_small_kernel.  Pass 1 writes d[gamma | beta] and sums
// S0..S3 per (sample, channel); pass 2 re-reads g, out, x, gamma (L2) and writes dx.  The three-launch path reads d beta back in
// the compute dtype for pass 2; here go is recomputed from g, so dx differs from it by that rounding only.
template <typename T, int G>
__global__ __launch_bounds__(256) void spade_small_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x, const T* __restrict__ gamma,
        const T* __restrict__ fout, const float* __restrict__ stats, const float* __restrict__ style, T* dx, const T* dxa, T* __restrict__ dgb,
        float* __restrict__ dstyle, int HW, int C, int lrelu, int sld, int acc, int xw, float inv_xw, int quad) {
    constexpr int VEC = Vec<T>::N, CH = G * VEC, RL = 256 / G;
    constexpr int UR = 2;                                    // rows in flight per thread and tensor (see in_small_bwd_kernel)
    __shared__ float red[RL][CH][4];
    __shared__ float mm[CH][2];
    const int tid = threadIdx.x, gx = tid % G, ry = tid / G;
    const int n = blockIdx.y, c0 = (blockIdx.x * G + gx) * VEC;
    const bool active = c0 < C;
    const size_t off = (size_t)n * HW * C + c0, off2 = (size_t)n * HW * 2 * C + c0;
    float mu[VEC], rs[VEC], a[VEC], S[4][VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        mu[j] = active ? stats[((size_t)n * C + c0 + j) * 2] : 0.f;
        rs[j] = active ? stats[((size_t)n * C + c0 + j) * 2 + 1] : 0.f;
        a[j] = active ? 1.f + style[(size_t)n * sld + c0 + j] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) S[k][j] = 0.f;
    }
    if (active)
        for (int r = ry; r < HW; r += UR * RL) {
            u32x4_t rg[UR], rx[UR], ra[UR], ro[UR];
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                const size_t o = off + (size_t)min(r + RL * k, HW - 1) * C;
                rg[k] = *(const u32x4_t*)(g + o); ra[k] = *(const u32x4_t*)(gamma + o);
                rx[k] = *(const u32x4_t*)(x + mod_x_row(n, min(r + RL * k, HW - 1), HW, xw, inv_xw) * C + c0);
                ro[k] = lrelu ? *(const u32x4_t*)(fout + o) : u32x4_t{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int k = 0; k < UR; ++k) {
                if (r + RL * k >= HW) continue;
                float fg[VEC], fx[VEC], ga[VEC], fo[VEC], dga[VEC], dbe[VEC];
                unpack16<T>(rg[k], fg); unpack16<T>(rx[k], fx); unpack16<T>(ra[k], ga); unpack16<T>(ro[k], fo);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (fx[j] - mu[j]) * rs[j];
                    const float go = (lrelu && !(fo[j] > 0.f)) ? 0.2f * fg[j] : fg[j];
                    dga[j] = 0.5f * go * xh;
                    dbe[j] = 0.5f * go;
                    const float gn = 0.5f * go * (1.f + ga[j]);
                    S[0][j] += gn; S[1][j] += gn * xh; S[2][j] += go * fx[j]; S[3][j] += go;
                }
                const size_t o2 = off2 + (size_t)(r + RL * k) * 2 * C;
                *(u32x4_t*)(dgb + o2) = pack16<T>(dga);
                *(u32x4_t*)(dgb + o2 + C) = pack16<T>(dbe);
            }
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[ry][gx * VEC + j][k] = S[k][j];
    __syncthreads();
    if (tid < CH) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < RL; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += (double)red[k][tid][q];
        mm[tid][0] = (float)t[0] / (float)HW; mm[tid][1] = (float)t[1] / (float)HW;
        const int c = blockIdx.x * CH + tid;
        if (c < C) {
            dstyle[(size_t)n * sld + c] += 0.5f * (float)t[2];
            dstyle[(size_t)n * sld + C + c] += 0.5f * (float)t[3];
        }
    }
    __syncthreads();
    if (!active) return;
    float m0[VEC], m1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { m0[j] = mm[gx * VEC + j][0]; m1[j] = mm[gx * VEC + j][1]; }
    if (quad) {                                              // dx w.r.t. the half-resolution x: the 2 x 2 sums (see the quad apply kernel)
        const int wl = xw >> 1, hwl = HW >> 2;
        const float inv_wl = 2.f * inv_xw;
        for (int pl = ry; pl < hwl; pl += RL) {
            const int yl = (int)(((float)pl + 0.5f) * inv_wl), xl = pl - yl * wl;
            const size_t lo = ((size_t)n * hwl + pl) * C + c0;
            float fx[VEC], o[VEC];
            unpack16<T>(*(const u32x4_t*)(x + lo), fx);
            u32x4_t rg[4], ra[4], ro[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t fo = off + ((size_t)(2 * yl + (q >> 1)) * xw + 2 * xl + (q & 1)) * C;
                rg[q] = *(const u32x4_t*)(g + fo); ra[q] = *(const u32x4_t*)(gamma + fo);
                ro[q] = lrelu ? *(const u32x4_t*)(fout + fo) : u32x4_t{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (fx[j] - mu[j]) * rs[j];
                o[j] = -4.f * rs[j] * (m0[j] + xh * m1[j]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float fg[VEC], ga[VEC], fo2[VEC];
                unpack16<T>(rg[q], fg); unpack16<T>(ra[q], ga); unpack16<T>(ro[q], fo2);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float go = (lrelu && !(fo2[j] > 0.f)) ? 0.2f * fg[j] : fg[j];
                    o[j] += 0.5f * go * a[j] + rs[j] * 0.5f * go * (1.f + ga[j]);
                }
            }
            if (acc) {
                float prev[VEC];
                unpack16<T>(*(const u32x4_t*)(dxa + lo), prev);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] += prev[j];
            }
            *(u32x4_t*)(dx + lo) = pack16<T>(o);
        }
        return;
    }
    for (int r = ry; r < HW; r += UR * RL) {
        u32x4_t rg[UR], rx[UR], ra[UR], ro[UR], rp[UR];
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            const size_t o = off + (size_t)min(r + RL * k, HW - 1) * C;
            rg[k] = *(const u32x4_t*)(g + o); ra[k] = *(const u32x4_t*)(gamma + o);
            rx[k] = *(const u32x4_t*)(x + mod_x_row(n, min(r + RL * k, HW - 1), HW, xw, inv_xw) * C + c0);
            ro[k] = lrelu ? *(const u32x4_t*)(fout + o) : u32x4_t{0u, 0u, 0u, 0u};
            rp[k] = acc ? *(const u32x4_t*)(dxa + o) : u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            if (r + RL * k >= HW) continue;
            float fg[VEC], fx[VEC], ga[VEC], fo[VEC], prev[VEC], o[VEC];
            unpack16<T>(rg[k], fg); unpack16<T>(rx[k], fx); unpack16<T>(ra[k], ga); unpack16<T>(ro[k], fo); unpack16<T>(rp[k], prev);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (fx[j] - mu[j]) * rs[j];
                const float go = (lrelu && !(fo[j] > 0.f)) ? 0.2f * fg[j] : fg[j];
                const float gn = 0.5f * go * (1.f + ga[j]);
                o[j] = 0.5f * go * a[j] + rs[j] * (gn - m0[j] - xh * m1[j]) + (acc ? prev[j] : 0.f);
            }
            *(u32x4_t*)(dx + off + (size_t)(r + RL * k) * C) = pack16<T>(o);
        }
    }
}

static int modulate_bwd_impl(int dtype, int mode, const void* g, const void* x, const void* gb, const void* fout, const float* stats,
                             const float* style, void* dx, void* dgb, float* dstyle, double* ws,
                             int N, int HW, int C, int lrelu, int style_ld, void* stream, int stage = 0, double batch_count = 0.0,
                             int xw = 0, int quad = 0, const void* dx_add = nullptr) {
    if (!dx_add) dx_add = dx;                              // S2E_NORM_ACCUMULATE_DX: dx = dx_add + this layer's gradient (in place unless told otherwise)
    const int sld = style_ld > 0 ? style_ld : 2 * C;
    const int gst = fout ? C : 2 * C;
    const int acc = (mode & S2E_NORM_ACCUMULATE_DX) != 0;
    mode &= ~S2E_NORM_ACCUMULATE_DX;
    const int batch = mode == S2E_NORM_SPADE_STYLE_BATCH;
    if (batch) mode = S2E_NORM_SPADE_STYLE;                // same passes; only the coefficient kernel sums over the batch
    if (!g || !x || !stats || !dx || !ws || N <= 0 || HW <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd: bad argument");
    if (mode == S2E_NORM_SPADE_STYLE && (!gb || !style || !dgb || !dstyle))
        S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd: SPADE_STYLE needs gb, style, dgb, dstyle");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd: bad dtype %d", dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_modulate_bwd: C=%d not a multiple of %d", C, vec);
    hipStream_t st = (hipStream_t)stream;
    const float inv_xw = xw ? 1.f / (float)xw : 0.f;
    if (xw && (!fout || batch || stage != 0 || HW % xw || ((HW / xw) | xw) & 1))
        S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd: x at half resolution needs the gamma-only form, per-sample statistics and an even H x W map");
    if (quad && (!xw || mode != S2E_NORM_SPADE_STYLE)) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd: dx_quad goes with x_up_w (SPADE_STYLE mode)");
    if (fout && mode == S2E_NORM_SPADE_STYLE && !batch && stage == 0 && HW <= in_small_hw()) {     // small map: one launch
        if (dtype == S2E_BF16) S2E_SMALL_LAUNCH(spade_small_bwd_kernel, bf16_t, (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)gb, (const bf16_t*)fout,
                                                 stats, style, (bf16_t*)dx, (const bf16_t*)dx_add, (bf16_t*)dgb, dstyle, HW, C, lrelu, sld, acc, xw, inv_xw, quad);
        else S2E_SMALL_LAUNCH(spade_small_bwd_kernel, float, (const float*)g, (const float*)x, (const float*)gb, (const float*)fout,
                              stats, style, (float*)dx, (const float*)dx_add, (float*)dgb, dstyle, HW, C, lrelu, sld, acc, xw, inv_xw, quad);
        S2E_CHECK_LAUNCH("spade_small_bwd_kernel");
        return S2E_OK;
    }
    const RowGeom rg = row_geom(C, vec);
    const int iters = slab_iters_for(HW, rg.rpp, N, rg.zblocks);
    dim3 grid1(ceil_div(HW, rg.rpp * iters), N, rg.zblocks);
    if ((long)HW * rg.cg >= (1L << 31)) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_modulate_bwd: sample too large for 32-bit indices");
    const int vps = HW * rg.cg;                            // 16-byte vectors per sample
    int gx = (vps + 255) / 256;
    const int gx_cap = 8192 / N > 1 ? 8192 / N : 1;
    if (gx > gx_cap) gx = gx_cap;
    dim3 grid2(gx, N);
    int cg_shift = -1;
    for (int b = 0; b < 31; ++b) if ((1 << b) == rg.cg) cg_shift = b;
    const int gridc = ceil_div((long)N * C, 256);
    f32x4_t* coef = (f32x4_t*)(ws + (size_t)N * C * 4);
    const int P = (int)grid1.x;                            // partial-sum slots per sample
    float* part = (float*)(ws + (size_t)N * C * 6);        // [N][P][C][4] floats behind the sums and the coefficients
    // the sums of ALL samples must be complete before the batch-statistics coefficients read them, and a staged call hands
    // them to the caller between the stages: a separate (tiny) launch then; otherwise the coefficient kernel adds up its own
    const bool sums_first = batch || stage == 1;
#define S2E_LAUNCH_BWD(TT, MM) do { \
    if (stage != 2) { modulate_bwd_reduce_kernel<TT, MM><<<grid1, 256, 0, st>>>((const TT*)g, (const TT*)x, (const TT*)gb, stats, style, (TT*)dgb, part, HW, C, rg.cg, rg.cgb, rg.rpp, lrelu, sld, iters, (const TT*)fout, xw, inv_xw); \
        if (sums_first) modulate_bwd_sums_kernel<MM><<<gridc, 256, 0, st>>>(part, ws, N, C, P); } \
    if (stage != 1) { modulate_bwd_coef_kernel<MM><<<gridc, 256, 0, st>>>(ws, (stage == 2 || sums_first) ? nullptr : part, P, coef, stats, style, dstyle, N, C, HW, sld, batch, batch_count); \
    if (quad) { const int gq = ((HW >> 2) * rg.cg + 255) / 256; \
        modulate_bwd_apply_quad_kernel<TT><<<dim3(gq < gx_cap ? gq : gx_cap, N), 256, 0, st>>>((const TT*)x, (const TT*)gb, (const TT*)dgb, coef, (TT*)dx, (const TT*)dx_add, HW, C, rg.cg, acc, gst, xw, 2.f * inv_xw); } \
    else modulate_bwd_apply_kernel<TT, MM><<<grid2, 256, 0, st>>>((const TT*)g, (const TT*)x, (const TT*)gb, (const TT*)dgb, coef, (TT*)dx, (const TT*)dx_add, vps, HW, C, rg.cg, cg_shift, lrelu, acc, gst, xw, inv_xw); } } while (0)
    if (dtype == S2E_BF16) { if (mode == S2E_NORM_SPADE_STYLE) S2E_LAUNCH_BWD(bf16_t, S2E_NORM_SPADE_STYLE); else S2E_LAUNCH_BWD(bf16_t, S2E_NORM_PLAIN_IN); }
    else { if (mode == S2E_NORM_SPADE_STYLE) S2E_LAUNCH_BWD(float, S2E_NORM_SPADE_STYLE); else S2E_LAUNCH_BWD(float, S2E_NORM_PLAIN_IN); }
#undef S2E_LAUNCH_BWD
    S2E_CHECK_LAUNCH("modulate_bwd kernels");
    return S2E_OK;
}

extern "C" int s2e_modulate_bwd(int dtype, int mode, const void* g, const void* x, const void* gb, const float* stats,
                                const float* style, void* dx, void* dgb, float* dstyle, double* ws,
                                int N, int HW, int C, int lrelu, int style_ld, void* stream) {
    return modulate_bwd_impl(dtype, mode, g, x, gb, nullptr, stats, style, dx, dgb, dstyle, ws, N, HW, C, lrelu, style_ld, stream);
}

extern "C" int s2e_modulate_bwd_gamma(int dtype, int mode, const void* g, const void* x, const void* gamma, const void* out,
                                      const float* stats, const float* style, void* dx, void* dgb, float* dstyle, double* ws,
                                      int N, int HW, int C, int lrelu, int style_ld, void* stream) {
    const int m = mode & ~S2E_NORM_ACCUMULATE_DX;
    if (m != S2E_NORM_SPADE_STYLE && m != S2E_NORM_SPADE_STYLE_BATCH) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd_gamma: SPADE_STYLE modes only");
    if (!gamma || !out) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd_gamma: gamma and out are required");
    return modulate_bwd_impl(dtype, mode, g, x, gamma, out, stats, style, dx, dgb, dstyle, ws, N, HW, C, lrelu, style_ld, stream);
}

extern "C" int s2e_modulate_bwd_staged(int dtype, int mode, const void* g, const void* x, const void* gb, const void* out,
                                       const float* stats, const float* style, void* dx, void* dgb, float* dstyle, double* ws,
                                       int N, int HW, int C, int lrelu, int style_ld, int stage, double batch_count, int x_up_w,
                                       int dx_quad, void* stream) {
    if (stage < 0 || stage > 2) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd_staged: stage %d", stage);
    return modulate_bwd_impl(dtype, mode, g, x, gb, out, stats, style, dx, dgb, dstyle, ws, N, HW, C, lrelu, style_ld, stream, stage, batch_count,
                             x_up_w, dx_quad);
}

// s2e_modulate_bwd_staged with the accumulated-into tensor and the result apart: dx = dx_add + this layer's gradient (mode must carry
// S2E_NORM_ACCUMULATE_DX; dx_add == dx or NULL is the in-place form)
extern "C" int s2e_modulate_bwd_relay(int dtype, int mode, const void* g, const void* x, const void* gb, const void* out,
                                      const float* stats, const float* style, void* dx, const void* dx_add, void* dgb, float* dstyle, double* ws,
                                      int N, int HW, int C, int lrelu, int style_ld, int stage, double batch_count, int x_up_w,
                                      int dx_quad, void* stream) {
    if (stage < 0 || stage > 2) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd_relay: stage %d", stage);
    if (!(mode & S2E_NORM_ACCUMULATE_DX)) S2E_FAIL(S2E_ERR_ARG, "s2e_modulate_bwd_relay: mode without S2E_NORM_ACCUMULATE_DX");
    return modulate_bwd_impl(dtype, mode, g, x, gb, out, stats, style, dx, dgb, dstyle, ws, N, HW, C, lrelu, style_ld, stream, stage, batch_count,
                             x_up_w, dx_quad, dx_add);
}

extern "C" size_t s2e_modulate_bwd_workspace_bytes(int dtype, int N, int HW, int C) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (N <= 0 || HW <= 0 || C <= 0 || C % vec) return 0;
    const RowGeom rg = row_geom(C, vec);
    const int P = ceil_div(HW, rg.rpp * slab_iters_for(HW, rg.rpp, N, rg.zblocks));
    return (size_t)N * C * 6 * sizeof(double) + (size_t)N * P * C * 4 * sizeof(float);
}

// ------------------------------------------------------------------------------------ plain InstanceNorm (+ LeakyReLU), whole op
// out = [lrelu 0.2]((x - mean) * rstd) per (sample, channel) over HW (InstanceNorm2d(affine=False), discriminator.py:91-94,
// encoder.py layers); stats (N,C,2) {mean, rstd} is written for the backward.  Small maps: one launch; others: the
// statistics kernels, then the element-wise kernel.  ws: s2e_in_stats_workspace_bytes.
extern "C" int s2e_instance_norm_fwd(int dtype, const void* x, void* out, float* stats, double* ws, int N, int HW, int C,
                                     float eps, int lrelu, void* stream) {
    if (!x || !out || !stats || !ws || N <= 0 || HW <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_instance_norm_fwd: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_instance_norm_fwd: bad dtype %d", dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_instance_norm_fwd: C=%d not a multiple of %d", C, vec);
    if (HW <= in_small_hw()) {
        hipStream_t st = (hipStream_t)stream;
        if (dtype == S2E_BF16) S2E_SMALL_LAUNCH2(in_small_kernel, bf16_t, 1, (const bf16_t*)x, (bf16_t*)out, ws, stats, HW, C, eps, lrelu);
        else S2E_SMALL_LAUNCH2(in_small_kernel, float, 1, (const float*)x, (float*)out, ws, stats, HW, C, eps, lrelu);
        S2E_CHECK_LAUNCH("in_small_kernel");
        return S2E_OK;
    }
    if (const int rc = s2e_in_stats(dtype, x, N, HW, C, eps, ws, stats, nullptr, stream)) return rc;
    return s2e_modulate_fwd(dtype, S2E_NORM_PLAIN_IN, x, nullptr, stats, nullptr, out, N, HW, C, lrelu, 0, stream);
}

// its backward: dx from g, x and the forward's stats.  ws: s2e_modulate_bwd_workspace_bytes (unused for small maps).
extern "C" int s2e_instance_norm_bwd(int dtype, const void* g, const void* x, const float* stats, void* dx, double* ws,
                                     int N, int HW, int C, int lrelu, void* stream) {
    if (!g || !x || !stats || !dx || N <= 0 || HW <= 0 || C <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_instance_norm_bwd: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_instance_norm_bwd: bad dtype %d", dtype);
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (C % vec) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_instance_norm_bwd: C=%d not a multiple of %d", C, vec);
    if (HW <= in_small_hw()) {
        hipStream_t st = (hipStream_t)stream;
        if (dtype == S2E_BF16) S2E_SMALL_LAUNCH(in_small_bwd_kernel, bf16_t, (const bf16_t*)g, (const bf16_t*)x, stats, (bf16_t*)dx, HW, C, lrelu);
        else S2E_SMALL_LAUNCH(in_small_bwd_kernel, float, (const float*)g, (const float*)x, stats, (float*)dx, HW, C, lrelu);
        S2E_CHECK_LAUNCH("in_small_bwd_kernel");
        return S2E_OK;
    }
    return s2e_modulate_bwd(dtype, S2E_NORM_PLAIN_IN, g, x, nullptr, stats, nullptr, dx, nullptr, nullptr, ws, N, HW, C, lrelu, 0, stream);
}
