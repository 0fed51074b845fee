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

static constexpr int SN_BR = 64;       // rows per block
static constexpr int SN_BC = 256;      // columns per block (one per thread)

// ---- t += W^T u over a [SN_BR x SN_BC] block; block_map = {layer, row0, col0}
__global__ __launch_bounds__(256) void sn_gemvT_kernel(const s2e_sn_layer* __restrict__ layers, const int* __restrict__ block_map) {
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_sn_layer L = layers[bm[0]];
    const int row0 = bm[1], col = bm[2] + threadIdx.x;
    if (col >= L.cols) return;
    const int rend = min(L.rows, row0 + SN_BR);
    float acc = 0.f;
    const float* wp = L.w + (size_t)row0 * L.cols + col;
#pragma unroll 8
    for (int r = row0; r < rend; ++r, wp += L.cols) acc += *wp * L.u[r];
    atomicAdd(L.t + col, acc);
}

// ---- one block per layer: v = t / max(|t|, eps)   (train only)
__global__ __launch_bounds__(256) void sn_norm_v_kernel(const s2e_sn_layer* __restrict__ layers, float eps) {
    __shared__ float red[4];
    const s2e_sn_layer L = layers[blockIdx.x];
    float q = 0.f;
    for (int j = threadIdx.x; j < L.cols; j += 256) { const float t = L.t[j]; q += t * t; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float inv = 1.f / fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), eps);
    for (int j = threadIdx.x; j < L.cols; j += 256) L.v[j] = L.t[j] * inv;
}

// ---- s += W v over a [SN_BR x SN_BC] block
__global__ __launch_bounds__(256) void sn_gemv_kernel(const s2e_sn_layer* __restrict__ layers, const int* __restrict__ block_map) {
    __shared__ float red[SN_BR][4];
    const int* bm = block_map + 3 * blockIdx.x;
    const s2e_sn_layer L = layers[bm[0]];
    const int row0 = bm[1], col = bm[2] + threadIdx.x;
    const int nr = min(L.rows - row0, SN_BR);
    const bool cv = col < L.cols;
    const float vj = cv ? L.v[col] : 0.f;
    const float* wp = L.w + (size_t)row0 * L.cols + col;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = 0; r < nr; ++r, wp += L.cols) {
        float p = cv ? *wp * vj : 0.f;
        p = wave_sum(p);
        if (lane == 0) red[r][wave] = p;
    }
    __syncthreads();
    if (threadIdx.x < nr) atomicAdd(L.s + row0 + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// ---- one block per layer: train: u = s / max(|s|, eps); both: sigma = u . s
__global__ __launch_bounds__(256) void sn_finalize_kernel(const s2e_sn_layer* __restrict__ layers, float* __restrict__ sigma, int train, float eps) {
    __shared__ float red[4];
    const s2e_sn_layer L = layers[blockIdx.x];
    float q = 0.f;
    if (train) {
        for (int i = threadIdx.x; i < L.rows; i += 256) { const float s = L.s[i]; q += s * s; }
    } else {
        for (int i = threadIdx.x; i < L.rows; i += 256) q += L.u[i] * L.s[i];
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    if (train) {
        const float inv = 1.f / fmaxf(sqrtf(tot), eps);
        for (int i = threadIdx.x; i < L.rows; i += 256) L.u[i] = L.s[i] * inv;
        if (threadIdx.x == 0) sigma[blockIdx.x] = tot * inv;            // u . s = |s|^2 / max(|s|, eps)
    } else if (threadIdx.x == 0) {
        sigma[blockIdx.x] = tot;
    }
}

extern "C" int s2e_sn_power_iteration(const s2e_sn_layer* layers, int n_layers, const int* block_map, int n_blocks,
                                      void* scratch, size_t scratch_bytes, float* sigma, int train, int iterations,
                                      float eps, void* stream) {
    if (!layers || !block_map || !scratch || !sigma || n_layers <= 0 || n_blocks <= 0 || iterations < 1)
        S2E_FAIL(S2E_ERR_ARG, "s2e_sn_power_iteration: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int iters = train ? iterations : 1;
    for (int it = 0; it < iters; ++it) {
        if (int zrc = s2e_zero_async(scratch, scratch_bytes, st)) return zrc;
        if (train) {
            sn_gemvT_kernel<<<n_blocks, 256, 0, st>>>(layers, block_map);
            sn_norm_v_kernel<<<n_layers, 256, 0, st>>>(layers, eps);
        }
        sn_gemv_kernel<<<n_blocks, 256, 0, st>>>(layers, block_map);
        sn_finalize_kernel<<<n_layers, 256, 0, st>>>(layers, sigma, train, eps);
    }
    S2E_CHECK_LAUNCH("sn power-iteration kernels");
    return S2E_OK;
}

// ------------------------------------------------------------------------------------ weight packers
// OIHW fp32 -> MFMA B-operand layout in the compute dtype, divided by *sigma when sigma != NULL.
// Both directions go through LDS so that global reads AND writes are contiguous runs.
// Every element of the padded matrix is written exactly once per call (padding rows / channels / K tail as
// zeros), so no separate zero-fill is needed.
// forward pack : out[row][(tap)*cin_pad + ci]   one block = one row x 64 (padded) ci
template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                       int cout, int cin, int taps, int cin_pad, int kpad) {
    extern __shared__ float lds[];                          // [64][taps + 1]
    const int row = blockIdx.x, ci0 = blockIdx.y * 64;
    const bool valid = row < cout;
    const int nci = valid ? max(0, min(64, cin - ci0)) : 0;       // real channels in this chunk
    const int ncp = min(64, cin_pad - ci0);                        // channels incl. structural-zero padding
    const float inv = sigma ? 1.f / *sigma : 1.f;
    if (nci > 0) {
        const float* src = w + ((size_t)row * cin + ci0) * taps;
        for (int i = threadIdx.x; i < nci * taps; i += 256) lds[(i / taps) * (taps + 1) + (i % taps)] = src[i] * inv;
    }
    __syncthreads();
    T* dst = out + (size_t)row * kpad + ci0;
    for (int i = threadIdx.x; i < ncp * taps; i += 256) {
        const int tap = i / ncp, cil = i - tap * ncp;
        dst[(size_t)tap * cin_pad + cil] = (T)(cil < nci ? lds[cil * (taps + 1) + tap] : 0.f);
    }
    if (blockIdx.y == 0)                                     // K tail [taps*cin_pad, kpad)
        for (int k = taps * cin_pad + threadIdx.x; k < kpad; k += 256) out[(size_t)row * kpad + k] = (T)0.f;
}
// transposed pack: out[ci][(tap)*cout + co]     one block = 64 co x 8 (padded) ci rows
template <typename T>
__global__ __launch_bounds__(256) void pack_tr_kernel(const float* __restrict__ w, T* __restrict__ out, const float* __restrict__ sigma,
                                                      int cout, int cin, int taps, int rows_pad, int kpad) {
    extern __shared__ float lds[];                          // [64 co][8*taps + 1]
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 8;
    const int nco = min(64, cout - co0);
    const int nci = max(0, min(8, cin - ci0)), ncp = min(8, rows_pad - ci0);
    const int run = nci * taps, ld = 8 * taps + 1;
    const float inv = sigma ? 1.f / *sigma : 1.f;
    for (int i = threadIdx.x; i < nco * run; i += 256) {
        const int col = i / run, r = i - col * run;
        lds[col * ld + r] = w[((size_t)(co0 + col) * cin + ci0) * taps + r] * inv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncp * taps * nco; i += 256) {
        const int col = i % nco, rt = i / nco;               // rt = cil*taps + tap
        const int cil = rt / taps, tap = rt - cil * taps;
        out[(size_t)(ci0 + cil) * kpad + (size_t)tap * cout + co0 + col] = (T)(cil < nci ? lds[col * ld + rt] : 0.f);
    }
    if (blockIdx.x == 0)                                     // K tail [taps*cout, kpad) of this block's rows
        for (int i = threadIdx.x; i < ncp * (kpad - taps * cout); i += 256) {
            const int cil = i / (kpad - taps * cout), k = taps * cout + i % (kpad - taps * cout);
            out[(size_t)(ci0 + cil) * kpad + k] = (T)0.f;
        }
}

extern "C" int s2e_pack_conv_weight(int dtype, const float* w, void* packed, const float* sigma, int cout, int cin, int kh, int kw,
                                    int cin_pad, int transposed, void* stream) {
    if (!w || !packed || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || cin_pad < cin)
        S2E_FAIL(S2E_ERR_ARG, "s2e_pack_conv_weight: bad argument");
    if (dtype != S2E_BF16 && dtype != S2E_F32) S2E_FAIL(S2E_ERR_ARG, "s2e_pack_conv_weight: bad dtype %d", dtype);
    const int taps = kh * kw;
    if (taps > 64) S2E_FAIL(S2E_ERR_UNSUPPORTED, "s2e_pack_conv_weight: kernel %dx%d too large", kh, kw);
    const int rows = s2e_conv_cout_pad(transposed ? cin_pad : cout);
    const int kpad = s2e_conv_k_pad(dtype, taps * (transposed ? cout : cin_pad));
    hipStream_t st = (hipStream_t)stream;
    if (!transposed) {
        dim3 grid(rows, ceil_div(cin_pad, 64));
        const size_t lds = (size_t)64 * (taps + 1) * sizeof(float);
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
    unpack_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gw_packed, gw_oihw, cout, cin, kh * kw, cin_pad, accumulate);
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
    sn_grad_dot_kernel<<<grid, 256, 0, st>>>(gw_packed, w_orig, dot_ws, cout, cin, kh * kw, cin_pad);
    sn_grad_apply_kernel<<<grid, 256, 0, st>>>(gw_packed, u, v, sigma, dot_ws, gw_orig, cout, cin, kh * kw, cin_pad, accumulate);
    S2E_CHECK_LAUNCH("sn_weight_grad kernels");
    return S2E_OK;
}
