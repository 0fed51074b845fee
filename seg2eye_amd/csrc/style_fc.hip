// The style FCs of a generator as one skinny GEMM (ApplyStyle / FC, reference models/networks/normalization.py:144-169:
// style = LeakyReLU(w @ W^T + b, 0.2) per SPADE+Style layer; networks/stylebank.py stacks the 21 layers' W and b).
//   forward   big[n][s] = lrelu(b[s] + sum_k w[n][k] W[s][k])                     n < N (batch), s < S (sum of 2C_i), k < K (w_dim)
//   backward  dpre = (dbig (+ gbig)) * lrelu'(big);  gW[s][k] += sum_n dpre[n][s] w[n][k];  gb[s] += sum_n dpre[n][s];
//             dw[n][k] = sum_s dpre[n][s] W[s][k]
// N <= 32 and K in {8, 16, 32, 64}: a thread owns one row s of W (K floats in registers) -- HBM-bound on W, gW (S*K*4 bytes each),
// one launch each way instead of torch's addmm / leaky_relu / where / mm / addmm_ / sum chain (11 launches, ~110 us per step).
// dw is a reduction over s: per-block partial sums, added in block order by a second tiny launch (deterministic; no float atomics).
#include "common.h"

namespace {
constexpr int SF_THREADS = 256;
constexpr int SF_MAXN = 32;

template <int K>
__global__ __launch_bounds__(SF_THREADS) void style_fc_fwd_kernel(const float* __restrict__ w, const float* __restrict__ W,
                                                                  const float* __restrict__ b, float* __restrict__ big, int N, int S, float slope) {
    __shared__ float sw[SF_MAXN * K];
    for (int i = threadIdx.x; i < N * K; i += SF_THREADS) sw[i] = w[i];
    __syncthreads();
    const int s = blockIdx.x * SF_THREADS + threadIdx.x;
    if (s >= S) return;
    float row[K];
#pragma unroll
    for (int k = 0; k < K; k += 4) {
        const f32x4_t v = *(const f32x4_t*)(W + (size_t)s * K + k);
        row[k] = v[0]; row[k + 1] = v[1]; row[k + 2] = v[2]; row[k + 3] = v[3];
    }
    const float bias = b[s];
    for (int n = 0; n < N; ++n) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(sw[n * K + k], row[k], acc);
        big[(size_t)n * S + s] = acc > 0.f ? acc : slope * acc;
    }
}

template <int K>
__global__ __launch_bounds__(SF_THREADS) void style_fc_bwd_kernel(const float* __restrict__ dbig, const float* __restrict__ gbig,
                                                                  const float* __restrict__ big, const float* __restrict__ w,
                                                                  const float* __restrict__ W, float* __restrict__ gW, float* __restrict__ gb,
                                                                  float* __restrict__ partial, int N, int S, float slope) {
    extern __shared__ float smem[];
    float* sw = smem;                                  // [N][K]
    float* sd = sw + SF_MAXN * K;                      // [N][SF_THREADS]: dpre of this block's rows
    float* sW = sd + SF_MAXN * SF_THREADS;             // [SF_THREADS][K]: this block's rows of W (only when dw is wanted)
    for (int i = threadIdx.x; i < N * K; i += SF_THREADS) sw[i] = w[i];
    __syncthreads();
    const int s = blockIdx.x * SF_THREADS + threadIdx.x;
    const bool in = s < S;
    float acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    for (int n = 0; n < N; ++n) {
        float d = 0.f;
        if (in) {
            const size_t o = (size_t)n * S + s;
            d = dbig[o];
            if (gbig) d += gbig[o];
            if (!(big[o] > 0.f)) d *= slope;           // LeakyReLU': big > 0 <=> pre-activation > 0
        }
        sd[n * SF_THREADS + threadIdx.x] = d;
        bsum += d;
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fmaf(d, sw[n * K + k], acc[k]);
    }
    if (in) {
        gb[s] += bsum;
#pragma unroll
        for (int k = 0; k < K; k += 4) {
            f32x4_t* p = (f32x4_t*)(gW + (size_t)s * K + k);
            f32x4_t v = *p;
            v[0] += acc[k]; v[1] += acc[k + 1]; v[2] += acc[k + 2]; v[3] += acc[k + 3];
            *p = v;
        }
    }
    if (!partial) return;
    // ---- dw[n][k] = sum_s dpre[n][s] W[s][k]: this block's 256 rows; style_fc_dw_fold_kernel adds the blocks' partials in order
#pragma unroll
    for (int k = 0; k < K; k += 4) {
        f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (in) v = *(const f32x4_t*)(W + (size_t)s * K + k);
        *(f32x4_t*)(sW + threadIdx.x * K + k) = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * K; idx += SF_THREADS) {
        const int n = idx / K, k = idx % K;
        float t = 0.f;
        for (int j = 0; j < SF_THREADS; ++j) t = fmaf(sd[n * SF_THREADS + j], sW[j * K + k], t);
        partial[(size_t)blockIdx.x * N * K + idx] = t;
    }
}

// dw[idx] = sum over the blocks' partial sums, in block order (a second, tiny launch: a last-block fold inside the kernel above
// needs a device-scope release per block, which on this multi-XCD part writes back the XCD's L2 -- measured 37 us for the pair
// instead of ~10)
__global__ __launch_bounds__(SF_THREADS) void style_fc_dw_fold_kernel(const float* __restrict__ partial, float* __restrict__ dw, int NK, int blocks) {
    const int idx = blockIdx.x * SF_THREADS + threadIdx.x;
    if (idx >= NK) return;
    float t = 0.f;
    for (int g = 0; g < blocks; ++g) t += partial[(size_t)g * NK + idx];
    dw[idx] = t;
}

size_t bwd_smem(int K) { return (size_t)(SF_MAXN * K + SF_MAXN * SF_THREADS + SF_THREADS * K) * sizeof(float); }
}  // namespace

extern "C" int s2e_style_fc_supported(int N, int K) { return N >= 1 && N <= SF_MAXN && (K == 8 || K == 16 || K == 32 || K == 64); }

extern "C" size_t s2e_style_fc_bwd_workspace_bytes(int N, int K, int S) {
    return (size_t)ceil_div(S, SF_THREADS) * N * K * sizeof(float);      // [blocks][N*K] fp32 partial sums of dw; no initialisation needed
}

extern "C" int s2e_style_fc_fwd(const float* w, const float* W, const float* b, float* big, int N, int K, int S, float slope, void* stream) {
    if (!w || !W || !b || !big || S <= 0 || !s2e_style_fc_supported(N, K)) S2E_FAIL(S2E_ERR_ARG, "s2e_style_fc_fwd: bad argument (N=%d K=%d S=%d)", N, K, S);
    const int grid = ceil_div(S, SF_THREADS);
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 8: style_fc_fwd_kernel<8><<<grid, SF_THREADS, 0, st>>>(w, W, b, big, N, S, slope); break;
        case 16: style_fc_fwd_kernel<16><<<grid, SF_THREADS, 0, st>>>(w, W, b, big, N, S, slope); break;
        case 32: style_fc_fwd_kernel<32><<<grid, SF_THREADS, 0, st>>>(w, W, b, big, N, S, slope); break;
        default: style_fc_fwd_kernel<64><<<grid, SF_THREADS, 0, st>>>(w, W, b, big, N, S, slope); break;
    }
    S2E_CHECK_LAUNCH("style_fc_fwd_kernel");
    return S2E_OK;
}

extern "C" int s2e_style_fc_bwd(const float* dbig, const float* gbig, const float* big, const float* w, const float* W, float* gW, float* gb,
                                float* dw, void* workspace, size_t workspace_bytes, int N, int K, int S, float slope, void* stream) {
    if (!dbig || !big || !w || !W || !gW || !gb || S <= 0 || !s2e_style_fc_supported(N, K))
        S2E_FAIL(S2E_ERR_ARG, "s2e_style_fc_bwd: bad argument (N=%d K=%d S=%d)", N, K, S);
    const int grid = ceil_div(S, SF_THREADS);
    float* partial = nullptr;
    if (dw) {
        if (!workspace || workspace_bytes < s2e_style_fc_bwd_workspace_bytes(N, K, S))
            S2E_FAIL(S2E_ERR_ARG, "s2e_style_fc_bwd: dw needs %zu bytes of workspace", s2e_style_fc_bwd_workspace_bytes(N, K, S));
        partial = (float*)workspace;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t sm = bwd_smem(K);
#define S2E_SF(KK) do { \
        static bool attr_done = false; \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)style_fc_bwd_kernel<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_done = true; } \
        style_fc_bwd_kernel<KK><<<grid, SF_THREADS, sm, st>>>(dbig, gbig, big, w, W, gW, gb, partial, N, S, slope); } while (0)
    switch (K) {
        case 8: S2E_SF(8); break;
        case 16: S2E_SF(16); break;
        case 32: S2E_SF(32); break;
        default: S2E_SF(64); break;
    }
#undef S2E_SF
    S2E_CHECK_LAUNCH("style_fc_bwd_kernel");
    if (dw) {
        style_fc_dw_fold_kernel<<<ceil_div(N * K, SF_THREADS), SF_THREADS, 0, st>>>(partial, dw, N * K, grid);
        S2E_CHECK_LAUNCH("style_fc_dw_fold_kernel");
    }
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ the encoder's head
// ConvEncoder: mu = fc_mu(LeakyReLU(x).view(M, -1)) (reference models/networks/encoder.py:68-71) on the NHWC feature map x
// (M, P = so*so pixels, C channels): torch flattens (c, p), so
//     y[m][n] = b[n] + sum_{p,c} lrelu(x[m][p][c]) * W[n][c*P + p]                       M <= 64 samples, N <= 32 outputs
// As a 4x4 valid convolution on the implicit-GEMM kernel this was ONE 128-row tile walking K = 8192 alone (37 us for 8 MFLOP),
// plus two packs, a dtype conversion, a generic data / weight gradient and their re-layout.  Here: one block per sample forward;
// one thread per (p, c) backward, which produces dx, dW and db in the same pass.  x in the compute dtype, everything else fp32.
namespace {
constexpr int FH_MAXN = 32, FH_MAXM = 64, FH_FWD_THREADS = 1024;

// Threads walk W's columns kw = c*P + p (coalesced rows of W, the 512-KB operand); x is the small one: the forward stages the
// sample's row in LDS (LeakyReLU applied once), the backward reads / writes it through L1 with a (p, c) stride.
template <typename T, int NMAX>
__global__ __launch_bounds__(FH_FWD_THREADS) void fc_head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                          float* __restrict__ y, int P, int C, int N, float slope) {
    extern __shared__ float sx[];                    // [P][C + 1] LeakyReLU(x[m]) in memory order (p, c)
    __shared__ float red[FH_FWD_THREADS / 64][FH_MAXN];
    const int m = blockIdx.x, K = P * C;
    for (int k = threadIdx.x; k < K; k += FH_FWD_THREADS) {     // rows padded by one float: the gather below walks p at a fixed c
        const float v = load1<T>(x + (size_t)m * K + k);
        const int pp = k / C;
        sx[k + pp] = v > 0.f ? v : slope * v;
    }
    __syncthreads();
    float acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
    // (a trip is one L2 round trip for its N loads of W: 1024 threads keep the trips few -- 256 threads took 32 of them, 70 us)
#pragma unroll 2
    for (int kw = threadIdx.x; kw < K; kw += FH_FWD_THREADS) {
        const int c = kw / P, pp = kw - c * P;
        const float v = sx[pp * (C + 1) + c];
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) acc[n] = fmaf(v, W[(size_t)n * K + kw], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        if (n < N) {                                         // (uniform: every lane takes the shuffles)
            const float s = wave_sum(acc[n]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][n] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < N) {
        float s = b[threadIdx.x];
#pragma unroll
        for (int wv = 0; wv < FH_FWD_THREADS / 64; ++wv) s += red[wv][threadIdx.x];
        y[(size_t)m * N + threadIdx.x] = s;
    }
}

// Forward in two small launches (round 5).  The one-block-per-sample kernel above keeps 32 of 256 CUs busy, each pulling the whole
// 1-MB weight matrix through one L1: 57 us per call for 17 MFLOP.  Here a block owns 256 columns kw of W for EIGHT samples: W is read
// 4 times instead of 32, by 128 blocks; each thread forms its 8 x N products and folds them over the wave with DPP adds at once
// (no per-thread accumulator array), the four waves' sums meet in LDS, and the block writes its partial (8, N) tile to the
// workspace; fc_head_fin_kernel adds the K / 256 partial tiles in a fixed order (bit-reproducible) and the bias.
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void fc_head_part_kernel(const T* __restrict__ x, const float* __restrict__ W, float* __restrict__ part,
                                                           int M, int P, int C, int N, float slope) {
    __shared__ float red[4][8][FH_MAXN];
    const int K = P * C, kw = blockIdx.x * 256 + threadIdx.x, m0 = blockIdx.y * 8;
    const bool in = kw < K;
    const int c = in ? kw / P : 0, pp = in ? kw - c * P : 0;
    float xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = 0.f;
        if (in && m0 + i < M) { v = load1<T>(x + (size_t)(m0 + i) * K + pp * C + c); v = v > 0.f ? v : slope * v; }
        xv[i] = v;
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
        if (n < N) {                                         // (uniform)
            const float w = in ? W[(size_t)n * K + kw] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float s = wave_sum_last(xv[i] * w);    // total in lane 63
                if ((threadIdx.x & 63) == 63) red[wave][i][n] = s;
            }
        }
    }
    __syncthreads();
    const int i = threadIdx.x >> 5, n = threadIdx.x & 31;
    if (n < N && m0 + i < M)
        part[((size_t)blockIdx.x * M + m0 + i) * FH_MAXN + n] = (red[0][i][n] + red[1][i][n]) + (red[2][i][n] + red[3][i][n]);
}

__global__ void fc_head_fin_kernel(const float* __restrict__ part, const float* __restrict__ b, float* __restrict__ y, int M, int N, int chunks) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int m = idx / N, n = idx - m * N;
    float s = b[n];
    int q = 0;
    for (; q + 8 <= chunks; q += 8) {                        // eight loads in flight, added in order (one by one each load waited for the last)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[((size_t)(q + j) * M + m) * FH_MAXN + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; q < chunks; ++q) s += part[((size_t)q * M + m) * FH_MAXN + n];
    y[idx] = s;
}

template <typename T, int NMAX>
__global__ __launch_bounds__(256) void fc_head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ W, const float* __restrict__ dy,
                                                          T* __restrict__ dx, float* __restrict__ dW, float* __restrict__ db,
                                                          int M, int P, int C, int N, float slope) {
    __shared__ float sdy[FH_MAXM * FH_MAXN];
    const int K = P * C;
    for (int i = threadIdx.x; i < M * N; i += 256) sdy[i] = dy[i];
    __syncthreads();
    if (blockIdx.x == 0 && db && threadIdx.x < N) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += sdy[m * N + threadIdx.x];
        db[threadIdx.x] += s;
    }
    const int kw = blockIdx.x * 256 + threadIdx.x;   // W's column
    if (kw >= K) return;
    const int c = kw / P, pp = kw - c * P;
    const size_t k = (size_t)pp * C + c;             // x's element
    float wv[NMAX], gw[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) { wv[n] = n < N ? W[(size_t)n * K + kw] : 0.f; gw[n] = 0.f; }
    for (int m0 = 0; m0 < M; m0 += 4) {                  // four samples' loads in flight
        float xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = load1<T>(x + (size_t)min(m0 + j, M - 1) * K + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + j;
            if (m < M) {
                const float xv = xs[j];
                const float a = xv > 0.f ? xv : slope * xv;
                float g = 0.f;
#pragma unroll
                for (int n = 0; n < NMAX; ++n)
                    if (n < N) { const float d = sdy[m * N + n]; gw[n] = fmaf(d, a, gw[n]); g = fmaf(d, wv[n], g); }
                if (dx) store1<T>(dx + (size_t)m * K + k, xv > 0.f ? g : slope * g);
            }
        }
    }
    if (dW) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) dW[(size_t)n * K + kw] += gw[n];
    }
}
}  // namespace

extern "C" int s2e_fc_head_supported(int M, int N) { return M >= 1 && M <= FH_MAXM && N >= 1 && N <= FH_MAXN; }

extern "C" size_t s2e_fc_head_fwd_workspace_bytes(int M, int P, int C, int N) {
    if (!s2e_fc_head_supported(M, N) || P <= 0 || C <= 0) return 0;
    return (size_t)ceil_div((long)P * C, 256) * M * FH_MAXN * sizeof(float);
}

// workspace (s2e_fc_head_fwd_workspace_bytes, uninitialised) given: the two-launch form above; NULL: one block per sample
extern "C" int s2e_fc_head_fwd(int dtype, const void* x, const float* W, const float* b, float* y, int M, int P, int C, int N, float slope,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace && x && W && b && y && P > 0 && C > 0 && s2e_fc_head_supported(M, N) && workspace_bytes >= s2e_fc_head_fwd_workspace_bytes(M, P, C, N)) {
        hipStream_t st2 = (hipStream_t)stream;
        const int chunks = ceil_div((long)P * C, 256);
        const dim3 grid(chunks, ceil_div(M, 8));
#define S2E_FHP(TT, NM) fc_head_part_kernel<TT, NM><<<grid, 256, 0, st2>>>((const TT*)x, W, (float*)workspace, M, P, C, N, slope)
        if (dtype == S2E_BF16) { if (N <= 16) S2E_FHP(bf16_t, 16); else S2E_FHP(bf16_t, 32); }
        else if (dtype == S2E_F32) { if (N <= 16) S2E_FHP(float, 16); else S2E_FHP(float, 32); }
#undef S2E_FHP
        else S2E_FAIL(S2E_ERR_ARG, "s2e_fc_head_fwd: bad dtype %d", dtype);
        S2E_CHECK_LAUNCH("fc_head_part_kernel");
        fc_head_fin_kernel<<<ceil_div(M * N, 256), 256, 0, st2>>>((const float*)workspace, b, y, M, N, chunks);
        S2E_CHECK_LAUNCH("fc_head_fin_kernel");
        return S2E_OK;
    }
    if (!x || !W || !b || !y || P <= 0 || C <= 0 || !s2e_fc_head_supported(M, N)) S2E_FAIL(S2E_ERR_ARG, "s2e_fc_head_fwd: bad argument (M=%d N=%d)", M, N);
    if ((size_t)P * C * sizeof(float) > 48 * 1024) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_fc_head_fwd: a sample row of %d x %d features does not fit the LDS stage", P, C);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)P * (C + 1) * sizeof(float);
#define S2E_FH(TT, NM) fc_head_fwd_kernel<TT, NM><<<M, FH_FWD_THREADS, lds, st>>>((const TT*)x, W, b, y, P, C, N, slope)
    if (dtype == S2E_BF16) { if (N <= 16) S2E_FH(bf16_t, 16); else S2E_FH(bf16_t, 32); }
    else if (dtype == S2E_F32) { if (N <= 16) S2E_FH(float, 16); else S2E_FH(float, 32); }
#undef S2E_FH
    else S2E_FAIL(S2E_ERR_ARG, "s2e_fc_head_fwd: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("fc_head_fwd_kernel");
    return S2E_OK;
}

extern "C" int s2e_fc_head_bwd(int dtype, const void* x, const float* W, const float* dy, void* dx, float* dW, float* db, int M, int P, int C,
                               int N, float slope, void* stream) {
    if (!x || !W || !dy || P <= 0 || C <= 0 || !s2e_fc_head_supported(M, N)) S2E_FAIL(S2E_ERR_ARG, "s2e_fc_head_bwd: bad argument (M=%d N=%d)", M, N);
    hipStream_t st = (hipStream_t)stream;
    const int grid = ceil_div((long)P * C, 256);
#define S2E_FH(TT, NM) fc_head_bwd_kernel<TT, NM><<<grid, 256, 0, st>>>((const TT*)x, W, dy, (TT*)dx, dW, db, M, P, C, N, slope)
    if (dtype == S2E_BF16) { if (N <= 16) S2E_FH(bf16_t, 16); else S2E_FH(bf16_t, 32); }
    else if (dtype == S2E_F32) { if (N <= 16) S2E_FH(float, 16); else S2E_FH(float, 32); }
#undef S2E_FH
    else S2E_FAIL(S2E_ERR_ARG, "s2e_fc_head_bwd: bad dtype %d", dtype);
    S2E_CHECK_LAUNCH("fc_head_bwd_kernel");
    return S2E_OK;
}
