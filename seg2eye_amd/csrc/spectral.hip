// Spectral normalisation (torch.nn.utils.spectral_norm semantics: 1 power iteration per forward in
// train mode, eps 1e-12, dim 0) for ALL spectral-normed convs of a network in four launches, the
// sigma-aware weight packers, and the fused gradient through W = W_orig / sigma.
//
//   train:  t = W^T u;  v = t / max(|t|, eps);  s = W v;  u = s / max(|s|, eps);  sigma = u . s
//   eval :  s = W v;  sigma = u . s                              (u, v untouched)
//
// W is the (Cout) x (Cin*kh*kw) row-major view of weight_orig (fp32, lives in the flat parameter arena).
// HBM-bound: each power iteration reads every W twice; nothing else is materialised -- in particular
// W / sigma never exists in memory: the packers divide on the fly while converting to the MFMA layout.
#include "common.h"
#include <stdlib.h>

// The partial dot products of the blocks are combined with 64-bit INTEGER atomics on fixed-point values (2^-40 units):
// integer addition is associative, so t, s -- and with them u, v, sigma -- come out bit-identical whatever order the
// blocks finish in: run to run, and on every data-parallel replica (the fp32 atomics used before made sigma differ by
// ~1e-7 between runs, which bf16 weight rounding and the InstanceNorm chain amplified to ~0.05 on the generated image).
// |partial| < 2^22 fits with room to spare (spectral norms here are O(1..100)); the quantisation (9e-13) is far below
// fp32 resolution of the values being summed.
static constexpr float SN_FIX = 1099511627776.0f;          // 2^40
static constexpr float SN_UNFIX = 1.0f / 1099511627776.0f;
__device__ __forceinline__ void sn_fix_add(long long* dst, float v) {
    atomicAdd((unsigned long long*)dst, (unsigned long long)__double2ll_rn((double)v * (double)SN_FIX));
}
__device__ __forceinline__ float sn_unfix(long long q) { return (float)((double)q * (double)SN_UNFIX); }
// Channels-last masters (L.taps > 1: weight_orig stored [co][tap][ci]): W's MEMORY columns run (tap, ci) while the
// module's `weight_v` buffer keeps torch's (ci, tap) order -- state_dict, replica broadcast and the reference's u, v tests see
// no difference.  The accumulator t is internal and stays in memory order; only the three places that touch v translate.
__device__ __forceinline__ int sn_vidx(const s2e_sn_layer& L, int col) {
    return L.taps > 1 ? (col % L.cin) * L.taps + col / L.cin : col;
}

static constexpr int SN_BR = 32;       // rows per block, two batches of 16: a batch's row loads are all in flight together
static constexpr int SN_BC = 1024;     // columns per block: four consecutive ones (one 16-byte load per row) per thread
static constexpr int SN_T_BR = 32, SN_T_V = 1;     // W^T u pass: 32 x 256 tiles, one column per thread (see sn_gemvT_kernel)
// which = 0: the tiles of the W v pass; 1: of the W^T u pass (its own block map: narrower tiles)
extern "C" int s2e_sn_block_shape(int which, int* rows, int* cols) {
    if (rows) *rows = which ? SN_T_BR : SN_BR;
    if (cols) *cols = which ? 256 * SN_T_V : SN_BC;
    return S2E_OK;
}

// A thread's 16-byte loads need cols % 4 == 0 and an aligned matrix: every layer of the networks here but the encoder's first
// conv (9 columns), which takes the scalar loops.  Rows past the matrix are CLAMPED to its last row and weighted with zero, so
// that a batch's 16 loads are unconditional and all in flight together (a guarded load per row compiles to a branch and a
// wait per row: 2.6 TB/s instead of 4.5).
typedef const __attribute__((address_space(1))) float* sn_gptr;
typedef const __attribute__((address_space(1))) f32x4_t* sn_gptr4;

// ---- t += W^T u over a [BR x 256 V] block (V consecutive columns per thread); block_map = {layer, row0, col0}.
// Measured on the generator's bank (267 MB; a call's duration, same box): 16x256 75 us, 32x256 74, 64x256 75, 128x256 74 --
// the atomics (one per BR elements) are not what bounds it -- and 73 us with 16-byte loads (V = 4) once the atomics are
// re-ordered through LDS; on the 22-25 MB banks of D and E the small tiles win (5.4 us against 8.2: more workgroups than CUs).
template <int BR, int V>
__global__ __launch_bounds__(256) void sn_gemvT_kernel(const s2e_sn_layer* __restrict__ layers, const int* __restrict__ block_map) {
    __shared__ float stage[256 * V];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_sn_layer L = layers[bm[0]];
    const int row0 = bm[1], col = bm[2] + V * threadIdx.x;
    const int nr = min(L.rows - row0, BR);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (col < L.cols) {
        sn_gptr wp = (sn_gptr)L.w + (size_t)row0 * L.cols + col;
        sn_gptr up = (sn_gptr)L.u + row0;
        if (V == 1 || ((L.cols & 3) == 0 && ((uintptr_t)L.w & 15) == 0)) {
#pragma unroll
            for (int b = 0; b < BR; b += 16) {
                float w[16][V];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    sn_gptr src = wp + (size_t)min(b + k, nr - 1) * L.cols;
                    if constexpr (V == 4) { const f32x4_t q = *(sn_gptr4)src; w[k][0] = q[0]; w[k][1] = q[1]; w[k][2] = q[2]; w[k][3] = q[3]; }
                    else w[k][0] = *src;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float u = up[min(b + k, nr - 1)] * (b + k < nr ? 1.f : 0.f);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] += w[k][j] * u;
                }
            }
        } else {
            for (int r = 0; r < nr; ++r)
                for (int j = 0; j < V; ++j) if (col + j < L.cols) acc[j] += wp[(size_t)r * L.cols + j] * up[r];
        }
    }
    if constexpr (V == 1) {
        if (col < L.cols) sn_fix_add(L.t + col, acc[0]);
    } else {
        // an atomic instruction costs per cache line it touches: hand the sums over through LDS so that the lanes of one
        // instruction add to 64 CONSECUTIVE columns (8 lines) instead of every fourth (32 lines; measured 118 vs 75 us)
#pragma unroll
        for (int j = 0; j < V; ++j) stage[V * threadIdx.x + j] = acc[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = j * 256 + threadIdx.x;
            if (bm[2] + c < L.cols) sn_fix_add(L.t + bm[2] + c, stage[c]);
        }
    }
}

// ---- one block per layer: v = t / max(|t|, eps)   (train only).  1024 threads and every load of a thread in flight
// at once: a latency chain (cols <= 9216 -> 9 loads per thread), 20 times per train step.
static constexpr int SN_NT = 1024, SN_MAXL = 12;
__device__ __forceinline__ float sn_block_sum(float q, float* red) {
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < SN_NT / 64; ++w) tot += red[w];
    return tot;
}
__global__ __launch_bounds__(SN_NT) void sn_norm_v_kernel(const s2e_sn_layer* __restrict__ layers, float eps) {
    __shared__ float red[SN_NT / 64];
    const s2e_sn_layer L = layers[blockIdx.x];
    float q = 0.f;
    if (L.cols <= SN_NT * SN_MAXL) {
        float t[SN_MAXL];
#pragma unroll
        for (int k = 0; k < SN_MAXL; ++k) { const int j = threadIdx.x + k * SN_NT; t[k] = j < L.cols ? sn_unfix(L.t[j]) : 0.f; }
#pragma unroll
        for (int k = 0; k < SN_MAXL; ++k) q += t[k] * t[k];
        const float inv = 1.f / fmaxf(sqrtf(sn_block_sum(q, red)), eps);
#pragma unroll
        for (int k = 0; k < SN_MAXL; ++k) { const int j = threadIdx.x + k * SN_NT; if (j < L.cols) L.v[sn_vidx(L, j)] = t[k] * inv; }
    } else {
        for (int j = threadIdx.x; j < L.cols; j += SN_NT) { const float t = sn_unfix(L.t[j]); q += t * t; }
        const float inv = 1.f / fmaxf(sqrtf(sn_block_sum(q, red)), eps);
        for (int j = threadIdx.x; j < L.cols; j += SN_NT) L.v[sn_vidx(L, j)] = sn_unfix(L.t[j]) * inv;
    }
    // s is accumulated (atomics) by the next launch: clear it here instead of a separate zero-fill launch per iteration
    for (int i = threadIdx.x; i < L.rows; i += SN_NT) L.s[i] = 0;
}

// ---- s += W v over a [SN_BR x SN_BC] block
__global__ __launch_bounds__(256) void sn_gemv_kernel(const s2e_sn_layer* __restrict__ layers, const int* __restrict__ block_map) {
    __shared__ float red[SN_BR][4];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_sn_layer L = layers[bm[0]];
    const int row0 = bm[1], col = bm[2] + 4 * threadIdx.x;
    const int nr = min(L.rows - row0, SN_BR);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((L.cols & 3) == 0 && ((uintptr_t)L.w & 15) == 0 && ((uintptr_t)L.v & 15) == 0) {
        const int cc = min(col, L.cols - 4);                   // a thread past the matrix re-reads its last columns, weighted with zero
        f32x4_t v4;
        if (L.taps > 1) { const int j0 = sn_vidx(L, cc); v4 = f32x4_t{L.v[j0], L.v[j0 + L.taps], L.v[j0 + 2 * L.taps], L.v[j0 + 3 * L.taps]}; }   // (cin % 4 == 0: one tap)
        else v4 = *(sn_gptr4)((sn_gptr)L.v + cc);
        const float on = col < L.cols ? 1.f : 0.f;
        const f32x4_t vv = {v4[0] * on, v4[1] * on, v4[2] * on, v4[3] * on};
        sn_gptr wp = (sn_gptr)L.w + (size_t)row0 * L.cols + cc;
#pragma unroll
        for (int b = 0; b < SN_BR; b += 16) {
            f32x4_t w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = *(sn_gptr4)(wp + (size_t)min(b + k, nr - 1) * L.cols);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float q = wave_sum_last((w[k][0] * vv[0] + w[k][1] * vv[1]) + (w[k][2] * vv[2] + w[k][3] * vv[3]));
                if (lane == 63) red[b + k][wave] = q;
            }
        }
    } else {
        sn_gptr wp = (sn_gptr)L.w + (size_t)row0 * L.cols + col;
        for (int r = 0; r < SN_BR; ++r) {
            float q = 0.f;
            if (r < nr)
                for (int j = 0; j < 4; ++j) if (col + j < L.cols) q += wp[(size_t)r * L.cols + j] * L.v[sn_vidx(L, col + j)];
            q = wave_sum(q);
            if (lane == 0) red[r][wave] = q;
        }
    }
    __syncthreads();
    if (threadIdx.x < nr) sn_fix_add(L.s + row0 + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// ---- one block per layer: train: u = s / max(|s|, eps); both: sigma = u . s
__global__ __launch_bounds__(SN_NT) void sn_finalize_kernel(const s2e_sn_layer* __restrict__ layers, float* __restrict__ sigma, int train, float eps) {
    __shared__ float red[SN_NT / 64];
    const s2e_sn_layer L = layers[blockIdx.x];
    float q = 0.f;
    if (train) {
        for (int i = threadIdx.x; i < L.rows; i += SN_NT) { const float s = sn_unfix(L.s[i]); q += s * s; }
    } else {
        for (int i = threadIdx.x; i < L.rows; i += SN_NT) q += L.u[i] * sn_unfix(L.s[i]);
    }
    const float tot = sn_block_sum(q, red);
    if (train) {
        const float inv = 1.f / fmaxf(sqrtf(tot), eps);
        for (int i = threadIdx.x; i < L.rows; i += SN_NT) L.u[i] = sn_unfix(L.s[i]) * inv;
        if (threadIdx.x == 0) sigma[blockIdx.x] = tot * inv;            // u . s = |s|^2 / max(|s|, eps)
    } else if (threadIdx.x == 0) {
        sigma[blockIdx.x] = tot;
    }
    // leave both accumulators cleared for the next iteration / forward (each thread clears what it alone read): the
    // scratch is zero at creation and stays zero between calls, so no zero-fill launch is needed per iteration
    for (int i = threadIdx.x; i < L.rows; i += SN_NT) L.s[i] = 0;
    if (train)
        for (int j = threadIdx.x; j < L.cols; j += SN_NT) L.t[j] = 0;
}

// ------------------------------------------------------------------------------------ small banks: two launches per iteration
// For a bank whose layers are all small (cols <= SN_CHAIN_MAX_COLS: the discriminator's and the encoder's -- the encoder runs N
// iterations per encode, 64 of a step's 80 power-iteration launches) the two per-layer launches of an iteration are folded into
// the GEMV passes: every block of the W v pass recomputes |t| from the layer's whole t (<= 64 KB of L2-resident accumulators,
// the same order in every block: the same bits) and normalises its own columns on the fly; every block of the next W^T u pass
// does the same with s.  Accumulators alternate between two buffers (iteration k adds into t[k & 1] / s[k & 1] and clears the
// other one's slice it owns), one finalising launch at the end writes u, sigma and leaves all four buffers zero: 2 I + 1
// launches for I iterations instead of 4 I.
static constexpr int SN_CHAIN_MAX_COLS = 8192;
__device__ __forceinline__ float sn_block_sum256(float q, float* red) {
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ long long* sn_tb(const s2e_sn_layer& L, int i) { return i ? L.t2 : L.t; }
__device__ __forceinline__ long long* sn_sb(const s2e_sn_layer& L, int i) { return i ? L.s2 : L.s; }

__global__ __launch_bounds__(256) void sn_gemvT_chain_kernel(const s2e_sn_layer* __restrict__ layers, const int* __restrict__ block_map,
                                                             int k, float eps) {
    __shared__ float red[4];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_sn_layer L = layers[bm[0]];
    const int row0 = bm[1], col = bm[2] + threadIdx.x;
    const int nr = min(L.rows - row0, SN_T_BR);
    const long long* sp = sn_sb(L, (k - 1) & 1);
    // the whole 32-row column strip is requested BEFORE |s| is summed: the chain is a string of dependent launches of ~10 us
    // each, and the norm (a pass over s plus a block reduction) otherwise sits in front of the weight loads' latency
    sn_gptr wp = (sn_gptr)L.w + (size_t)row0 * L.cols + min(col, L.cols - 1);
    float w[SN_T_BR];
#pragma unroll
    for (int r = 0; r < SN_T_BR; ++r) w[r] = wp[(size_t)min(r, nr - 1) * L.cols];
    float inv = 1.f;
    if (k > 0) {                                             // u = s / max(|s|, eps), s of the previous iteration
        float q = 0.f;
        for (int i = threadIdx.x; i < L.rows; i += 256) { const float v = sn_unfix(sp[i]); q += v * v; }
        inv = 1.f / fmaxf(sqrtf(sn_block_sum256(q, red)), eps);
    }
    if (col >= L.cols) return;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < SN_T_BR; ++r) {
        const int rr = row0 + min(r, nr - 1);
        const float u = (k > 0 ? sn_unfix(sp[rr]) * inv : L.u[rr]) * (r < nr ? 1.f : 0.f);
        acc += w[r] * u;
    }
    sn_fix_add(sn_tb(L, k & 1) + col, acc);
    if (k > 0 && row0 == 0) sn_tb(L, (k - 1) & 1)[col] = 0;       // (consumed by the previous W v pass)
}

__global__ __launch_bounds__(256) void sn_gemv_chain_kernel(const s2e_sn_layer* __restrict__ layers, const int* __restrict__ block_map,
                                                            int k, float eps) {
    __shared__ float red4[4];
    __shared__ float red[SN_BR][4];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_sn_layer L = layers[bm[0]];
    const int row0 = bm[1], col = bm[2] + 4 * threadIdx.x;
    const int nr = min(L.rows - row0, SN_BR);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long* tp = sn_tb(L, k & 1);
    const bool vec = (L.cols & 3) == 0 && ((uintptr_t)L.w & 15) == 0;
    const int cc = vec ? min(col, L.cols - 4) : col;         // a thread past the matrix re-reads its last columns, weighted with zero (vv = 0)
    sn_gptr wp = (sn_gptr)L.w + (size_t)row0 * L.cols + cc;
    f32x4_t w[SN_BR];                                        // all 32 rows requested before |t| is summed (see the W^T u kernel)
    if (vec) {
#pragma unroll
        for (int r = 0; r < SN_BR; ++r) w[r] = *(sn_gptr4)(wp + (size_t)min(r, nr - 1) * L.cols);
    } else {                                                 // (the encoder's first layer: 9 columns; row by row this path was the chain's longest block)
#pragma unroll
        for (int r = 0; r < SN_BR; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[r][j] = wp[(size_t)min(r, nr - 1) * L.cols + min(j, L.cols - 1 - col)];
    }
    float q = 0.f;
    for (int j0 = threadIdx.x; j0 < L.cols; j0 += 256 * 8) {        // eight accumulator loads in flight (up to 4608 columns: 18 round trips one by one)
        long long a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = tp[min(j0 + 256 * e, L.cols - 1)];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = j0 + 256 * e < L.cols ? sn_unfix(a[e]) : 0.f; q += v * v; }
    }
    const float inv = 1.f / fmaxf(sqrtf(sn_block_sum256(q, red4)), eps);
    float vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vv[j] = col + j < L.cols ? sn_unfix(tp[col + j]) * inv : 0.f;
    if (row0 == 0) {                                         // v = t / max(|t|, eps): the module's buffer, written once per column
#pragma unroll
        for (int j = 0; j < 4; ++j) if (col + j < L.cols) L.v[sn_vidx(L, col + j)] = vv[j];
    }
    if (vec) {
#pragma unroll
        for (int r = 0; r < SN_BR; ++r) {
            const float p = wave_sum_last((w[r][0] * vv[0] + w[r][1] * vv[1]) + (w[r][2] * vv[2] + w[r][3] * vv[3]));
            if (lane == 63) red[r][wave] = p;
        }
    } else {
#pragma unroll
        for (int r = 0; r < SN_BR; ++r) {                    // columns past the matrix carry vv = 0 (their loads repeat the last column)
            const float p = wave_sum_last(((w[r][0] * vv[0] + w[r][1] * vv[1]) + w[r][2] * vv[2]) + w[r][3] * vv[3]);
            if (lane == 63) red[r][wave] = p;
        }
    }
    __syncthreads();
    if (threadIdx.x < nr) {
        sn_fix_add(sn_sb(L, k & 1) + row0 + threadIdx.x, (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
        if (k > 0 && bm[2] == 0) sn_sb(L, (k - 1) & 1)[row0 + threadIdx.x] = 0;    // (consumed by this iteration's W^T u pass)
    }
}

// one block per layer, after the last iteration (which = (iterations - 1) & 1): u = s / max(|s|, eps), sigma = |s|^2 / max(|s|, eps);
// the two accumulators still holding values are cleared
__global__ __launch_bounds__(SN_NT) void sn_finalize_chain_kernel(const s2e_sn_layer* __restrict__ layers, float* __restrict__ sigma, int which, float eps) {
    __shared__ float red[SN_NT / 64];
    const s2e_sn_layer L = layers[blockIdx.x];
    long long* sp = sn_sb(L, which);
    long long* tp = sn_tb(L, which);
    float q = 0.f;
    for (int i = threadIdx.x; i < L.rows; i += SN_NT) { const float s = sn_unfix(sp[i]); q += s * s; }
    const float tot = sn_block_sum(q, red);
    const float inv = 1.f / fmaxf(sqrtf(tot), eps);
    for (int i = threadIdx.x; i < L.rows; i += SN_NT) { L.u[i] = sn_unfix(sp[i]) * inv; sp[i] = 0; }
    if (threadIdx.x == 0) sigma[blockIdx.x] = tot * inv;
    for (int j = threadIdx.x; j < L.cols; j += SN_NT) tp[j] = 0;
}

extern "C" int s2e_sn_chain_max_cols(void) { return SN_CHAIN_MAX_COLS; }

extern "C" int s2e_sn_power_iteration(const s2e_sn_layer* layers, int n_layers, const int* block_map_t, int n_blocks_t,
                                      const int* block_map, int n_blocks,
                                      void* scratch, size_t scratch_bytes, float* sigma, int train, int iterations,
                                      float eps, int chain, void* stream) {
    if (!layers || !block_map || !block_map_t || !scratch || !sigma || n_layers <= 0 || n_blocks <= 0 || n_blocks_t <= 0 || iterations < 1)
        S2E_FAIL(S2E_ERR_ARG, "s2e_sn_power_iteration: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int iters = train ? iterations : 1;
    (void)scratch_bytes;                                     // the accumulators in `scratch` are cleared by the kernels themselves
    if (train && chain) {                                    // small bank (caller checked s2e_sn_chain_max_cols, t2 / s2 set): 2 I + 1 launches
        for (int it = 0; it < iters; ++it) {
            sn_gemvT_chain_kernel<<<n_blocks_t, 256, 0, st>>>(layers, block_map_t, it, eps);
            sn_gemv_chain_kernel<<<n_blocks, 256, 0, st>>>(layers, block_map, it, eps);
        }
        sn_finalize_chain_kernel<<<n_layers, SN_NT, 0, st>>>(layers, sigma, (iters - 1) & 1, eps);
        S2E_CHECK_LAUNCH("sn chain kernels");
        return S2E_OK;
    }
    for (int it = 0; it < iters; ++it) {
        if (train) {
            sn_gemvT_kernel<SN_T_BR, SN_T_V><<<n_blocks_t, 256, 0, st>>>(layers, block_map_t);
            sn_norm_v_kernel<<<n_layers, SN_NT, 0, st>>>(layers, eps);
        }
        sn_gemv_kernel<<<n_blocks, 256, 0, st>>>(layers, block_map);
        sn_finalize_kernel<<<n_layers, SN_NT, 0, st>>>(layers, sigma, train, eps);
    }
    S2E_CHECK_LAUNCH("sn power-iteration kernels");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ weight packers
#include "conv_plane.h"
// OIHW fp32 -> MFMA B-operand layout in the compute dtype, divided by *sigma when sigma != NULL.
// Both directions go through LDS so that global reads AND writes are contiguous runs.
// Every element of the padded matrix is written exactly once per call (padding rows / channels / K tail as
// zeros), so no separate zero-fill is needed.
// Thread mapping as in the gradient re-layout kernels below: no per-element integer division.
// forward pack : out[row][(tap)*cin_pad + ci]   one block = 4 rows (one per wave) x 64 (padded) ci
template <typename T>
__device__ __forceinline__ void pack_fwd_block(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                               int cout, int cin, int taps, int cin_pad, int kpad, int bx, int by, float* lds) {
    // lds: [4][64 * taps]
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = bx * 4 + r, ci0 = by * 64;             // padded row count is a multiple of 32
    const bool valid = row < cout;
    const int nci = valid ? max(0, min(64, cin - ci0)) : 0;       // real channels in this chunk
    const int ncp = min(64, cin_pad - ci0);                        // channels incl. structural-zero padding
    const float inv = sigma ? 1.f / *sigma : 1.f;
    float* my = lds + r * 64 * taps;
    const float* src = w + ((size_t)row * cin + ci0) * taps;
    for (int k = lane; k < nci * taps; k += 64) my[k] = src[k] * inv;
    __syncthreads();
    T* dst = out + (size_t)row * kpad + ci0 + lane;
    if (lane < ncp)
        for (int tap = 0; tap < taps; ++tap) dst[(size_t)tap * cin_pad] = (T)(lane < nci ? my[lane * taps + tap] : 0.f);
    if (by == 0)                                             // K tail [taps*cin_pad, kpad)
        for (int k = taps * cin_pad + lane; k < kpad; k += 64) out[(size_t)row * kpad + k] = (T)0.f;
}
// transposed pack: out[ci][(tap)*cout + co]     one block = 64 co x 8 (padded) ci rows
template <typename T>
__device__ __forceinline__ void pack_tr_block(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                              int cout, int cin, int taps, int rows_pad, int kpad, int bx, int by, float* lds) {
    // lds: [64 co][8*taps + 1]
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int co0 = bx * 64, ci0 = by * 8;
    const int nco = min(64, cout - co0);
    const int nci = max(0, min(8, cin - ci0)), ncp = min(8, rows_pad - ci0);
    const int run = nci * taps, ld = 8 * taps + 1;
    const float inv = sigma ? 1.f / *sigma : 1.f;
    for (int m = q; m < nco; m += 4) {                       // wave q stages co rows q, q+4, ...: contiguous runs
        const float* src = w + ((size_t)(co0 + m) * cin + ci0) * taps;
        for (int k = lane; k < run; k += 64) lds[m * ld + k] = src[k] * inv;
    }
    __syncthreads();
    if (lane < nco)
        for (int cil = 0; cil < ncp; ++cil) {
            T* dst = out + (size_t)(ci0 + cil) * kpad + co0 + lane;
            for (int tap = q; tap < taps; tap += 4)          // wave q writes taps q, q+4, ...: 64 consecutive co each
                dst[(size_t)tap * cout] = (T)(cil < nci ? lds[lane * ld + cil * taps + tap] : 0.f);
        }
    if (bx == 0) {                                           // K tail [taps*cout, kpad) of this block's rows
        const int tail = kpad - taps * cout;
        for (int cil = q; cil < ncp; cil += 4)
            for (int k = lane; k < tail; k += 64) out[(size_t)(ci0 + cil) * kpad + taps * cout + k] = (T)0.f;
    }
}

// forward pack from a CHANNELS-LAST source w[co][tap][ci] (cin_pad == cin, cin % 8 == 0): the source row IS the packed row, so the
// pack is a streaming convert of the (rows_pad x kpad) output: a thread owns 8 consecutive columns (two 16-byte loads, one
// 16/32-byte store), a block 2048 consecutive elements of the flattened output; padding rows / the K tail are written as zeros.
static constexpr int PACK_CL_ELEMS = 2048;
template <typename T>
__device__ __forceinline__ void pack_fwd_cl_block(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                  int cout, int K, int rows_pad, int kpad, int bx) {
    const long idx = (long)bx * PACK_CL_ELEMS + 8 * threadIdx.x;
    if (idx >= (long)rows_pad * kpad) return;
    const int row = (int)(idx / kpad), col = (int)(idx - (long)row * kpad);      // (kpad % 8 == 0: the 8 columns share a row)
    const float inv = sigma ? 1.f / *sigma : 1.f;
    float f[8];
    if (row < cout && col < K) {                                                  // (K % 8 == 0: all 8 real or all 8 padding)
        const f32x4_t a = *(const f32x4_t*)(w + (size_t)row * K + col), b = *(const f32x4_t*)(w + (size_t)row * K + col + 4);
        f[0] = a[0] * inv; f[1] = a[1] * inv; f[2] = a[2] * inv; f[3] = a[3] * inv;
        f[4] = b[0] * inv; f[5] = b[1] * inv; f[6] = b[2] * inv; f[7] = b[3] * inv;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.f;
    }
    if constexpr (std::is_same<T, float>::value) {
        *(u32x4_t*)(out + idx) = pack16<float>(f);
        *(u32x4_t*)(out + idx + 4) = pack16<float>(f + 4);
    } else {
        *(u32x4_t*)(out + idx) = pack16<bf16_t>(f);
    }
}

// transposed pack from a CHANNELS-LAST source w[co][tap][ci] (cin_pad == cin): one block = 64 co x 64 ci of ONE tap, a plain
// tile transpose -- rows of 64 consecutive ci in, rows of 64 consecutive co out.  by = tap * ceil(rows_pad / 64) + ci chunk.
// out_fwd != NULL (round 6): the FORWARD pack of the same weight, [co][tap * cin + ci] with row pitch kpad_f, is written from the same
// tile -- the fp32 master is read once for both layouts (a G step packs 93 M parameters both ways: 372 MB of reads saved).  Only the
// weight's own elements are written there: the padding rows / K tail of that matrix are the caller's to zero, once.
template <typename T>
__device__ __forceinline__ void pack_tr_cl_block(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                 int cout, int cin, int taps, int rows_pad, int kpad, int bx, int by, float* lds,
                                                 T* __restrict__ out_fwd = nullptr, int kpad_f = 0) {
    // lds: [64 co][65]
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gyc = (rows_pad + 63) / 64;
    const int tap = by / gyc, ci0 = (by - tap * gyc) * 64, co0 = bx * 64;
    const float inv = sigma ? 1.f / *sigma : 1.f;
    for (int m = q; m < 64; m += 4) {
        const int co = co0 + m, ci = ci0 + lane;
        lds[m * 65 + lane] = (co < cout && ci < cin) ? w[((size_t)co * taps + tap) * cin + ci] * inv : 0.f;
    }
    __syncthreads();
    if (out_fwd && ci0 + lane < cin)
        for (int m = q; m < 64; m += 4)
            if (co0 + m < cout) out_fwd[(size_t)(co0 + m) * kpad_f + (size_t)tap * cin + ci0 + lane] = (T)lds[m * 65 + lane];
    if (co0 + lane < cout)
        for (int cil = q; cil < 64; cil += 4)
            if (ci0 + cil < rows_pad) out[(size_t)(ci0 + cil) * kpad + (size_t)tap * cout + co0 + lane] = (T)lds[lane * 65 + cil];
    if (bx == 0 && tap == 0) {                               // K tail [taps*cout, kpad) of this block's rows
        const int tail = kpad - taps * cout;
        for (int cil = q; cil < 64; cil += 4)
            if (ci0 + cil < rows_pad)
                for (int k = lane; k < tail; k += 64) out[(size_t)(ci0 + cil) * kpad + taps * cout + k] = (T)0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                       int cout, int cin, int taps, int cin_pad, int kpad) {
    extern __shared__ float lds[];
    pack_fwd_block<T>(w, out, sigma, cout, cin, taps, cin_pad, kpad, blockIdx.x, blockIdx.y, lds);
}
template <typename T>
__global__ __launch_bounds__(256) void pack_tr_kernel(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                      int cout, int cin, int taps, int rows_pad, int kpad) {
    extern __shared__ float lds[];
    pack_tr_block<T>(w, out, sigma, cout, cin, taps, rows_pad, kpad, blockIdx.x, blockIdx.y, lds);
}
template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_cl_kernel(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                          int cout, int K, int rows_pad, int kpad) {
    pack_fwd_cl_block<T>(w, out, sigma, cout, K, rows_pad, kpad, blockIdx.x);
}
template <typename T>
__global__ __launch_bounds__(256) void pack_tr_cl_kernel(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                         int cout, int cin, int taps, int rows_pad, int kpad) {
    extern __shared__ float lds[];
    pack_tr_cl_block<T>(w, out, sigma, cout, cin, taps, rows_pad, kpad, blockIdx.x, blockIdx.y, lds);
}
// the PLANE layout (conv_plane.h): bf16 only
__global__ __launch_bounds__(256) void pack_plane_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, const float* __restrict__ sigma,
                                                         int cout, int cin, int taps, int transposed) {
    pack_plane_block(w, out, sigma, cout, cin, taps, transposed, blockIdx.x);
}
// every conv of a network in one launch: block_map = {job, bx, by} per block (s2e_pack_block_map)
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const s2e_pack_job* __restrict__ jobs, const int* __restrict__ block_map,
                                                         const float* __restrict__ sigma_base, int dtype) {
    extern __shared__ float lds[];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_pack_job J = jobs[bm[0]];
    const float* sg = J.sigma_index >= 0 ? sigma_base + J.sigma_index : nullptr;
    if (J.transposed & S2E_PACK_PLANE) {                     // the PLANE layout (conv_plane.hip's weights)
        if constexpr (std::is_same<T, bf16_t>::value) pack_plane_block(J.w, (bf16_t*)J.out, sg, J.cout, J.cin, J.taps, J.transposed, bm[1]);
    } else if (J.transposed == 2) {                                 // channels-last source (s2e_pack_job: transposed bit 1), forward
        const int kpad = (J.taps * J.cin + (dtype == S2E_BF16 ? 63 : 31)) / (dtype == S2E_BF16 ? 64 : 32) * (dtype == S2E_BF16 ? 64 : 32);
        const int rows = J.cout <= 32 ? 32 : (J.cout <= 64 ? 64 : (J.cout + 127) / 128 * 128);
        pack_fwd_cl_block<T>(J.w, (T*)J.out, sg, J.cout, J.taps * J.cin, rows, kpad, bm[1]);
    } else if (J.transposed & 2) {                           // ... transposed
        const int kpad = (J.taps * J.cout + (dtype == S2E_BF16 ? 63 : 31)) / (dtype == S2E_BF16 ? 64 : 32) * (dtype == S2E_BF16 ? 64 : 32);
        const int rows = J.cin_pad <= 32 ? 32 : (J.cin_pad <= 64 ? 64 : (J.cin_pad + 127) / 128 * 128);
        const int kpad_f = (J.taps * J.cin + (dtype == S2E_BF16 ? 63 : 31)) / (dtype == S2E_BF16 ? 64 : 32) * (dtype == S2E_BF16 ? 64 : 32);
        pack_tr_cl_block<T>(J.w, (T*)J.out, sg, J.cout, J.cin, J.taps, rows, kpad, bm[1], bm[2], lds, (T*)J.out_fwd, kpad_f);
    } else if (!J.transposed) {
        const int kpad = (J.taps * J.cin_pad + (dtype == S2E_BF16 ? 63 : 31)) / (dtype == S2E_BF16 ? 64 : 32) * (dtype == S2E_BF16 ? 64 : 32);
        pack_fwd_block<T>(J.w, (T*)J.out, sg, J.cout, J.cin, J.taps, J.cin_pad, kpad, bm[1], bm[2], lds);
    } else {
        const int kpad = (J.taps * J.cout + (dtype == S2E_BF16 ? 63 : 31)) / (dtype == S2E_BF16 ? 64 : 32) * (dtype == S2E_BF16 ? 64 : 32);
        const int rows = J.cin_pad <= 32 ? 32 : (J.cin_pad <= 64 ? 64 : (J.cin_pad + 127) / 128 * 128);
        pack_tr_block<T>(J.w, (T*)J.out, sg, J.cout, J.cin, J.taps, rows, kpad, bm[1], bm[2], lds);
    }
}

extern "C" int s2e_pack_conv_weight(int dtype, const float* w, void* packed, const float* sigma, int cout, int cin, int kh, int kw,
                                    int cin_pad, int transposed, void* stream) {
    if (!w || !packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || cin_pad < cin)
        S2E_FAIL(S2E_ERR_ARG, "s2e_pack_conv_weight: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_pack_conv_weight: bad dtype %d", dtype);
    const int taps = kh * kw;
    if (taps > 64) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_pack_conv_weight: kernel %dx%d too large", kh, kw);
    const bool tr = (transposed & 1) != 0;
    if (transposed & S2E_PACK_PLANE) {                       // the PLANE layout (conv_plane.h): bf16, no channel padding, K a multiple of 32
        if (dtype != S2E_BF16 || cin_pad != cin || (tr ? cout : cin) % 32 != 0)
            S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_pack_conv_weight: the plane layout needs bf16, cin_pad == cin and a K dimension that is a multiple of 32");
        const long units = s2e_plane_pack_units(cout, cin, taps, transposed);
        pack_plane_kernel<<<ceil_div(units, 256), 256, 0, (hipStream_t)stream>>>(w, (bf16_t*)packed, sigma, cout, cin, taps, transposed);
        S2E_CHECK_LAUNCH("pack_plane_kernel");
        return S2E_OK;
    }
    const int rows = s2e_conv_cout_pad(tr ? cin_pad : cout);
    const int kpad = s2e_conv_k_pad(dtype, taps * (tr ? cout : cin_pad));
    hipStream_t st = (hipStream_t)stream;
    if (transposed & 2) {                                    // source in channels-last order w[co][tap][ci]
        if (cin_pad != cin) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_pack_conv_weight: a channels-last source needs cin_pad == cin");
        if (cin % 8) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_pack_conv_weight: a channels-last source needs cin %% 8 == 0");
        if (!tr) {                                           // forward: the source row IS the packed row: a streaming convert
            const int grid = ceil_div((long)rows * kpad, PACK_CL_ELEMS);
            if (dtype == S2E_BF16) pack_fwd_cl_kernel<bf16_t><<<grid, 256, 0, st>>>(w, (bf16_t*)packed, sigma, cout, taps * cin, rows, kpad);
            else pack_fwd_cl_kernel<float><<<grid, 256, 0, st>>>(w, (float*)packed, sigma, cout, taps * cin, rows, kpad);
        } else {
            dim3 grid(ceil_div(cout, 64), taps * ceil_div(rows, 64));
            const size_t lds = (size_t)64 * 65 * sizeof(float);
            if (dtype == S2E_BF16) pack_tr_cl_kernel<bf16_t><<<grid, 256, lds, st>>>(w, (bf16_t*)packed, sigma, cout, cin, taps, rows, kpad);
            else pack_tr_cl_kernel<float><<<grid, 256, lds, st>>>(w, (float*)packed, sigma, cout, cin, taps, rows, kpad);
        }
        S2E_CHECK_LAUNCH("pack kernels (channels-last source)");
        return S2E_OK;
    }
    if (!transposed) {
        dim3 grid(rows / 4, ceil_div(cin_pad, 64));
        const size_t lds = (size_t)4 * 64 * taps * sizeof(float);
        if (dtype == S2E_BF16) pack_fwd_kernel<bf16_t><<<grid, 256, lds, st>>>(w, (bf16_t*)packed, sigma, cout, cin, taps, cin_pad, kpad);
        else pack_fwd_kernel<float><<<grid, 256, lds, st>>>(w, (float*)packed, sigma, cout, cin, taps, cin_pad, kpad);
    } else {
        dim3 grid(ceil_div(cout, 64), ceil_div(rows, 8));
        const size_t lds = (size_t)64 * (8 * taps + 1) * sizeof(float);
        if (dtype == S2E_BF16) pack_tr_kernel<bf16_t><<<grid, 256, lds, st>>>(w, (bf16_t*)packed, sigma, cout, cin, taps, rows, kpad);
        else pack_tr_kernel<float><<<grid, 256, lds, st>>>(w, (float*)packed, sigma, cout, cin, taps, rows, kpad);
    }
    S2E_CHECK_LAUNCH("pack kernels");
    return S2E_OK;
}

extern "C" long s2e_pack_block_map(int dtype, const s2e_pack_job* jobs_host, int n_jobs, int* block_map_host) {
    if (!jobs_host || n_jobs < 0) return S2E_ERR_ARG;
    long nb = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const s2e_pack_job& J = jobs_host[j];
        int gx, gy;
        if (J.transposed & S2E_PACK_PLANE) { gx = ceil_div(s2e_plane_pack_units(J.cout, J.cin, J.taps, J.transposed), 256); gy = 1; }
        else if (J.transposed == 2)  { gx = ceil_div((long)s2e_conv_cout_pad(J.cout) * s2e_conv_k_pad(dtype, J.taps * J.cin), PACK_CL_ELEMS); gy = 1; }
        else if (J.transposed & 2) { gx = ceil_div(J.cout, 64);   gy = J.taps * ceil_div(s2e_conv_cout_pad(J.cin_pad), 64); }
        else if (!J.transposed) { gx = s2e_conv_cout_pad(J.cout) / 4; gy = ceil_div(J.cin_pad, 64); }
        else                    { gx = ceil_div(J.cout, 64);      gy = ceil_div(s2e_conv_cout_pad(J.cin_pad), 8); }
        if (block_map_host)
            for (int y = 0; y < gy; ++y)
                for (int x = 0; x < gx; ++x) {
                    int* e = block_map_host + 3 * (nb + (long)y * gx + x);
                    e[0] = j; e[1] = x; e[2] = y;
                }
        nb += (long)gx * gy;
    }
    return nb;
}

extern "C" int s2e_pack_conv_weights(int dtype, const s2e_pack_job* jobs, const int* block_map, int n_blocks, int max_taps,
                                     const float* sigma_base, void* stream) {
    if (!jobs || !block_map || n_blocks <= 0 || max_taps <= 0) S2E_FAIL(S2E_ERR_ARG, "s2e_pack_conv_weights: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_pack_conv_weights: bad dtype %d", dtype);
    if (max_taps > 16) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_pack_conv_weights: more than 16 taps");
    size_t lds = (size_t)64 * (8 * max_taps + 1) * sizeof(float);
    if (lds < (size_t)64 * 65 * sizeof(float)) lds = (size_t)64 * 65 * sizeof(float);      // (the channels-last tile transpose)
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2E_BF16) pack_batch_kernel<bf16_t><<<n_blocks, 256, lds, st>>>(jobs, block_map, sigma_base, dtype);
    else pack_batch_kernel<float><<<n_blocks, 256, lds, st>>>(jobs, block_map, sigma_base, dtype);
    S2E_CHECK_LAUNCH("batched pack kernel");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ gradient through W = W_orig / sigma
// Given gwp = dL/dW (packed order: [co][tap*cin_pad + ci], fp32, from s2e_conv2d_wgrad):
//   dL/dW_orig = gW / sigma - (<gW, W_orig> / sigma^2) * u v^T          (sigma = u^T W_orig v, u, v constants)
// written in OIHW order.  Kernel 1: dot = <gW, W_orig>;  kernel 2: the element-wise combination.
__global__ __launch_bounds__(256) void sn_grad_dot_kernel(const float* __restrict__ gwp, const float* __restrict__ w, float* __restrict__ dot,
                                                          int cout, int cin, int taps, int cin_pad) {
    __shared__ float red[4];
    const long total = (long)cout * cin * taps;
    float q = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % cin), co = (int)(r / cin);
        q += w[i] * gwp[(size_t)co * taps * cin_pad + (size_t)tap * cin_pad + ci];
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dot, red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sn_grad_apply_kernel(const float* __restrict__ gwp, const float* __restrict__ u, const float* __restrict__ v,
        const float* __restrict__ sigma, const float* __restrict__ dot, float* __restrict__ out, int cout, int cin, int taps, int cin_pad,
        int accumulate) {
    const long total = (long)cout * cin * taps;
    const float inv = 1.f / *sigma;
    const float c = *dot * inv * inv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % cin), co = (int)(r / cin);
        const float g = gwp[(size_t)co * taps * cin_pad + (size_t)tap * cin_pad + ci] * inv - c * u[co] * v[(size_t)ci * taps + tap];
        out[i] = accumulate ? out[i] + g : g;
    }
}

// Tiled variants for taps > 1: the packed gradient is [co][tap][ci] while W_orig / the output are [co][ci][tap].  A tile
// of SNG_ROWS co rows x 64 ci goes through LDS so that BOTH sides move as contiguous runs; thread (wave r, lane) owns
// row r: its lane index is the ci on the packed side and the flat run index (lane + 64 j) on the OIHW side, so no
// per-element integer division is needed (the element-wise kernels above spend ~150 instructions per element on
// 64-bit / and %: they were ALU-bound, not HBM-bound).  Blocks loop over tiles.
static constexpr int SNG_ROWS = 4;     // co rows per tile = waves per block
__global__ __launch_bounds__(256) void sn_grad_dot_tiled_kernel(const float* __restrict__ gwp, const float* __restrict__ w, float* __restrict__ dot,
                                                                int cout, int cin, int taps, int cin_pad) {
    extern __shared__ float lds[];                          // [SNG_ROWS][64 * taps]
    __shared__ float red[4];
    const int chunks = (cin + 63) / 64, rgroups = (cout + SNG_ROWS - 1) / SNG_ROWS, tiles = rgroups * chunks;
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + r * 64 * taps;
    float q = 0.f;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int rg = t / chunks, ci0 = (t - rg * chunks) * 64, co = rg * SNG_ROWS + r;
        const int nci = min(64, cin - ci0), run = nci * taps;
        const bool rv = co < cout;
        const float* wrow = w + ((size_t)co * cin + ci0) * taps;
        const float* grow = gwp + (size_t)co * taps * cin_pad + ci0 + lane;
        __syncthreads();
        if (rv)
            for (int k = lane; k < run; k += 64) my[k] = wrow[k];
        __syncthreads();
        if (rv && lane < nci)
            for (int tap = 0; tap < taps; ++tap) q += grow[(size_t)tap * cin_pad] * my[lane * taps + tap];
    }
    q = wave_sum(q);
    if (lane == 0) red[r] = q;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dot, red[0] + red[1] + red[2] + red[3]);
}
// out[co][ci][tap] (+)= gwp[co][tap][ci] / sigma - (dot / sigma^2) u[co] v[ci*taps + tap];   u == NULL: plain re-layout
__global__ __launch_bounds__(256) void grad_unpack_tiled_kernel(const float* __restrict__ gwp, const float* __restrict__ u, const float* __restrict__ v,
        const float* __restrict__ sigma, const float* __restrict__ dot, float* __restrict__ out, int cout, int cin, int taps, int cin_pad,
        int accumulate) {
    extern __shared__ float lds[];                          // [SNG_ROWS][64 * taps]
    const int chunks = (cin + 63) / 64, rgroups = (cout + SNG_ROWS - 1) / SNG_ROWS, tiles = rgroups * chunks;
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + r * 64 * taps;
    const float inv = u ? 1.f / *sigma : 1.f;
    const float c = u ? *dot * inv * inv : 0.f;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int rg = t / chunks, ci0 = (t - rg * chunks) * 64, co = rg * SNG_ROWS + r;
        const int nci = min(64, cin - ci0), run = nci * taps;
        const bool rv = co < cout;
        const float* grow = gwp + (size_t)co * taps * cin_pad + ci0 + lane;
        __syncthreads();
        if (rv && lane < nci)
            for (int tap = 0; tap < taps; ++tap) my[lane * taps + tap] = grow[(size_t)tap * cin_pad];
        __syncthreads();
        if (rv) {
            float* orow = out + ((size_t)co * cin + ci0) * taps;
            const float cu = u ? c * u[co] : 0.f;
            const float* vrow = v + (size_t)ci0 * taps;
            for (int k = lane; k < run; k += 64) {
                float gg = my[k] * inv;
                if (u) gg -= cu * vrow[k];
                orow[k] = accumulate ? orow[k] + gg : gg;
            }
        }
    }
}

// ---- all layers of a step at once: block_map = {job, first tile, number of tiles}; same tile bodies as above
__global__ __launch_bounds__(256) void grad_dot_batch_kernel(const s2e_grad_job* __restrict__ jobs, const int* __restrict__ block_map, float* __restrict__ dots) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_grad_job J = jobs[bm[0]];
    if (!J.w_orig) return;                                  // plain re-layout job: nothing to reduce
    const int taps = J.taps, cin = J.cin, cout = J.cout, cin_pad = J.cin_pad;
    const int chunks = (cin + 63) / 64;
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + r * 64 * taps;
    float q = 0.f;
    for (int t = bm[1]; t < bm[1] + bm[2]; ++t) {
        const int rg = t / chunks, ci0 = (t - rg * chunks) * 64, co = rg * SNG_ROWS + r;
        const int nci = min(64, cin - ci0), run = nci * taps;
        const bool rv = co < cout;
        const float* wrow = J.w_orig + ((size_t)co * cin + ci0) * taps;
        const float* grow = J.gw_packed + (size_t)co * taps * cin_pad + ci0 + lane;
        __syncthreads();
        if (rv)
            for (int k = lane; k < run; k += 64) my[k] = wrow[k];
        __syncthreads();
        if (rv && lane < nci)
            for (int tap = 0; tap < taps; ++tap) q += grow[(size_t)tap * cin_pad] * my[lane * taps + tap];
    }
    q = wave_sum(q);
    if (lane == 0) red[r] = q;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dots + J.dot_index, red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void grad_unpack_batch_kernel(const s2e_grad_job* __restrict__ jobs, const int* __restrict__ block_map, const float* __restrict__ dots) {
    extern __shared__ float lds[];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_grad_job J = jobs[bm[0]];
    const int taps = J.taps, cin = J.cin, cout = J.cout, cin_pad = J.cin_pad;
    const int chunks = (cin + 63) / 64;
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* my = lds + r * 64 * taps;
    const bool sn = J.w_orig != nullptr;
    const float inv = sn ? 1.f / *J.sigma : 1.f;
    const float c = sn ? dots[J.dot_index] * inv * inv : 0.f;
    for (int t = bm[1]; t < bm[1] + bm[2]; ++t) {
        const int rg = t / chunks, ci0 = (t - rg * chunks) * 64, co = rg * SNG_ROWS + r;
        const int nci = min(64, cin - ci0), run = nci * taps;
        const bool rv = co < cout;
        const float* grow = J.gw_packed + (size_t)co * taps * cin_pad + ci0 + lane;
        __syncthreads();
        if (rv && lane < nci)
            for (int tap = 0; tap < taps; ++tap) my[lane * taps + tap] = grow[(size_t)tap * cin_pad];
        __syncthreads();
        if (rv) {
            float* orow = J.out + ((size_t)co * cin + ci0) * taps;
            const float cu = sn ? c * J.u[co] : 0.f;
            const float* vrow = sn ? J.v + (size_t)ci0 * taps : nullptr;
            for (int k = lane; k < run; k += 64) {
                float gg = my[k] * inv;
                if (sn) gg -= cu * vrow[k];
                orow[k] += gg;
            }
        }
    }
}
static constexpr int GRAD_TILES_PER_BLOCK = 8;
extern "C" long s2e_grad_block_map(const s2e_grad_job* jobs_host, int n_jobs, int* block_map_host) {
    if (!jobs_host || n_jobs < 0) return S2E_ERR_ARG;
    long nb = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const s2e_grad_job& J = jobs_host[j];
        const int tiles = ceil_div(J.cout, SNG_ROWS) * ceil_div(J.cin, 64);
        for (int t0 = 0; t0 < tiles; t0 += GRAD_TILES_PER_BLOCK, ++nb)
            if (block_map_host) {
                int* e = block_map_host + 3 * nb;
                e[0] = j; e[1] = t0; e[2] = tiles - t0 < GRAD_TILES_PER_BLOCK ? tiles - t0 : GRAD_TILES_PER_BLOCK;
            }
    }
    return nb;
}
extern "C" int s2e_weight_grads_batched(const s2e_grad_job* jobs, const int* block_map, int n_blocks, int max_taps, int any_sn,
                                        float* dots, void* stream) {
    if (!jobs || !block_map || n_blocks <= 0 || max_taps <= 0 || (any_sn && !dots)) S2E_FAIL(S2E_ERR_ARG, "s2e_weight_grads_batched: bad argument");
    if (max_taps > 64) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_weight_grads_batched: more than 64 taps");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)SNG_ROWS * 64 * max_taps * sizeof(float);
    if (any_sn) grad_dot_batch_kernel<<<n_blocks, 256, lds, st>>>(jobs, block_map, dots);
    grad_unpack_batch_kernel<<<n_blocks, 256, lds, st>>>(jobs, block_map, dots);
    S2E_CHECK_LAUNCH("batched weight-gradient kernels");
    return S2E_OK;
}

// grad[co][ci][tap] (+)= gwp[co][tap*cin_pad + ci]: packed wgrad output -> OIHW gradient (no spectral norm)
__global__ __launch_bounds__(256) void unpack_grad_kernel(const float* __restrict__ gwp, float* __restrict__ out, int cout, int cin, int taps,
                                                          int cin_pad, int accumulate) {
    const long total = (long)cout * cin * taps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const long r = i / taps;
        const int ci = (int)(r % cin), co = (int)(r / cin);
        const float g = gwp[(size_t)co * taps * cin_pad + (size_t)tap * cin_pad + ci];
        out[i] = accumulate ? out[i] + g : g;
    }
}
extern "C" int s2e_unpack_weight_grad(const float* gw_packed, float* gw_oihw, int cout, int cin, int kh, int kw, int cin_pad,
                                      int accumulate, void* stream) {
    if (!gw_packed || !gw_oihw || cout <= 0 || cin <= 0 || cin_pad < cin) S2E_FAIL(S2E_ERR_ARG, "s2e_unpack_weight_grad: bad argument");
    const long total = (long)cout * cin * kh * kw;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    const int taps = kh * kw;
    if (taps > 1 && taps <= 64) {
        const int tiles = ceil_div(cout, SNG_ROWS) * ceil_div(cin, 64);
        grad_unpack_tiled_kernel<<<tiles < 2048 ? tiles : 2048, 256, (size_t)SNG_ROWS * 64 * taps * sizeof(float), (hipStream_t)stream>>>(
            gw_packed, nullptr, nullptr, nullptr, nullptr, gw_oihw, cout, cin, taps, cin_pad, accumulate);
    } else
        unpack_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gw_packed, gw_oihw, cout, cin, taps, cin_pad, accumulate);
    S2E_CHECK_LAUNCH("unpack_grad_kernel");
    return S2E_OK;
}

extern "C" int s2e_sn_weight_grad(const float* gw_packed, const float* w_orig, const float* u, const float* v, const float* sigma,
                                  float* dot_ws, float* gw_orig, int cout, int cin, int kh, int kw, int cin_pad, int accumulate,
                                  void* stream) {
    if (!gw_packed || !w_orig || !u || !v || !sigma || !dot_ws || !gw_orig || cout <= 0 || cin <= 0 || cin_pad < cin)
        S2E_FAIL(S2E_ERR_ARG, "s2e_sn_weight_grad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)cout * cin * kh * kw;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    const int taps = kh * kw;
    if (taps > 1 && taps <= 64) {
        const int tiles = ceil_div(cout, SNG_ROWS) * ceil_div(cin, 64);
        const size_t lds = (size_t)SNG_ROWS * 64 * taps * sizeof(float);
        sn_grad_dot_tiled_kernel<<<tiles < 1024 ? tiles : 1024, 256, lds, st>>>(gw_packed, w_orig, dot_ws, cout, cin, taps, cin_pad);
        grad_unpack_tiled_kernel<<<tiles < 2048 ? tiles : 2048, 256, lds, st>>>(gw_packed, u, v, sigma, dot_ws, gw_orig, cout, cin, taps, cin_pad, accumulate);
    } else {
        sn_grad_dot_kernel<<<grid, 256, 0, st>>>(gw_packed, w_orig, dot_ws, cout, cin, taps, cin_pad);
        sn_grad_apply_kernel<<<grid, 256, 0, st>>>(gw_packed, u, v, sigma, dot_ws, gw_orig, cout, cin, taps, cin_pad, accumulate);
    }
    S2E_CHECK_LAUNCH("sn_weight_grad kernels");
    return S2E_OK;
}


// ------------------------------------------------------------------------------------ the same gradient IN PLACE (channels-last masters)
// With the fp32 masters stored in the packed order [co][tap][ci] the weight-gradient kernels accumulate straight into the
// parameter's slice of the gradient arena; what is left of the chain rule above is element-wise and in place:
//     g[i] = g[i] / sigma - (<g, W_orig> / sigma^2) u[row] v[col]           (row = i / K, col = i % K, K = taps * cin)
// Launch 1: every block stores the partial dot product of its 16 Ki-element chunk (plain store).  Launch 2: every block of a
// layer first adds that layer's partials in the same fixed order (deterministic: no float atomics), then rewrites its chunk.
// 16 bytes of traffic per element, all of it contiguous (the OIHW re-layout moved 20, half of it in 4-byte strides).
static constexpr int SNI_CHUNK = 16384;     // elements per block: 256 threads x 16 float4
__global__ __launch_bounds__(256) void sn_grad_inplace_dot_kernel(const s2e_sngrad_job* __restrict__ jobs, const int* __restrict__ block_map,
                                                                  float* __restrict__ partials) {
    __shared__ float red[4];
    const int* bm = block_map + 2 * blockIdx.x;
    const s2e_sngrad_job J = jobs[bm[0]];
    const long total = (long)J.rows * J.cin * J.taps;
    const long i0 = (long)bm[1] * SNI_CHUNK, i1 = i0 + SNI_CHUNK < total ? i0 + SNI_CHUNK : total;
    float q = 0.f;
    for (long i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {         // (total % 4 == 0: cin % 8 == 0)
        const f32x4_t a = *(const f32x4_t*)(J.g + i), b = *(const f32x4_t*)(J.w + i);
        q += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) partials[J.part0 + bm[1]] = (red[0] + red[1]) + (red[2] + red[3]);
    // the layer's first block also lays v out in W's MEMORY order for the second launch (one 16-byte load per four columns there
    // instead of four strided ones)
    if (bm[1] == 0) {
        const int K = J.cin * J.taps;
        float* vm = partials + J.vmem0;
        for (int col = threadIdx.x; col < K; col += 256) { const int tap = col / J.cin, ci = col - tap * J.cin; vm[col] = J.v[(size_t)ci * J.taps + tap]; }
    }
}
__global__ __launch_bounds__(256) void sn_grad_inplace_apply_kernel(const s2e_sngrad_job* __restrict__ jobs, const int* __restrict__ block_map,
                                                                    const float* __restrict__ partials) {
    __shared__ float red[4];
    const int* bm = block_map + 2 * blockIdx.x;
    const s2e_sngrad_job J = jobs[bm[0]];
    float q = 0.f;
    for (int k = threadIdx.x; k < J.nparts; k += 256) q += partials[J.part0 + k];
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.f / *J.sigma;
    const float c = dot * inv * inv;
    const int K = J.cin * J.taps;
    const float* vm = partials + J.vmem0;
    const long total = (long)J.rows * K;
    const long i0 = (long)bm[1] * SNI_CHUNK, i1 = i0 + SNI_CHUNK < total ? i0 + SNI_CHUNK : total;
    long i = i0 + 4 * threadIdx.x;
    int row = (int)(i / K), col = (int)(i - (long)row * K);          // (one 64-bit division per thread; then stepped)
    for (; i < i1; i += 1024, col += 1024) {
        while (col >= K) { col -= K; ++row; }
        const float cu = c * J.u[row];
        const f32x4_t vv = *(const f32x4_t*)(vm + col);             // (v in memory order: written by the first launch)
        f32x4_t a = *(const f32x4_t*)(J.g + i);
        a[0] = a[0] * inv - cu * vv[0];
        a[1] = a[1] * inv - cu * vv[1];
        a[2] = a[2] * inv - cu * vv[2];
        a[3] = a[3] * inv - cu * vv[3];
        *(f32x4_t*)(J.g + i) = a;
    }
}
extern "C" long s2e_sngrad_block_map(s2e_sngrad_job* jobs_host, int n_jobs, int* block_map_host) {
    if (!jobs_host || n_jobs < 0) return S2E_ERR_ARG;
    long nb = 0;
    for (int j = 0; j < n_jobs; ++j) {
        s2e_sngrad_job& J = jobs_host[j];
        const long total = (long)J.rows * J.cin * J.taps;
        const int chunks = (int)((total + SNI_CHUNK - 1) / SNI_CHUNK);
        J.part0 = (int)nb; J.nparts = chunks;                // (one partial per block: the block index IS the slot)
        for (int c = 0; c < chunks; ++c, ++nb)
            if (block_map_host) { block_map_host[2 * nb] = j; block_map_host[2 * nb + 1] = c; }
    }
    long off = (nb + 3) / 4 * 4;                             // behind the partials: every layer's v in memory order
    for (int j = 0; j < n_jobs; ++j) { jobs_host[j].vmem0 = (int)off; off += (long)jobs_host[j].cin * jobs_host[j].taps; }
    return nb;
}
extern "C" long s2e_sngrad_scratch_floats(const s2e_sngrad_job* jobs_host, int n_jobs) {
    if (!jobs_host || n_jobs <= 0) return 0;
    const s2e_sngrad_job& J = jobs_host[n_jobs - 1];        // (after s2e_sngrad_block_map: the last layer's v region ends the scratch)
    return (long)J.vmem0 + (long)J.cin * J.taps;
}
extern "C" int s2e_sn_grads_inplace(const s2e_sngrad_job* jobs, const int* block_map, int n_blocks, float* partials, void* stream) {
    if (!jobs || !block_map || n_blocks <= 0 || !partials) S2E_FAIL(S2E_ERR_ARG, "s2e_sn_grads_inplace: bad argument");
    // (cin % 8 == 0 of every job is the caller's to guarantee: the jobs live in device memory)
    hipStream_t st = (hipStream_t)stream;
    sn_grad_inplace_dot_kernel<<<n_blocks, 256, 0, st>>>(jobs, block_map, partials);
    sn_grad_inplace_apply_kernel<<<n_blocks, 256, 0, st>>>(jobs, block_map, partials);
    S2E_CHECK_LAUNCH("in-place spectral-norm gradient kernels");
    return S2E_OK;
}
