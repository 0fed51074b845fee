// Degenerate-channel convolutions: Cout == 1 (conv_img 64->1, the PatchGAN heads 512->1) and Cin == 1
// (encoder layer0 1->64).  A 128-wide MFMA tile does 1/32..1/128 useful work on these; they are pure
// HBM-bound streams, so each gets a small vector kernel: one thread owns a 16-byte channel group of the
// wide tensor and a handful of scalar loads of the 1-channel tensor.  Dispatched from s2e_conv2d /
// s2e_conv2d_wgrad (conv_igemm.hip / conv_wgrad.hip); no separate ABI.
#include "common.h"
#include "conv_small.h"

static constexpr int MAXT = 16;     // taps (<= 4x4)

// ---------------------------------------------------------------- forward, Cout == 1
// y[o] = out_act( bias + sum_{tap,ci} in_act(x[i(o,tap)][ci]) * w[tap*Cin + ci] )      one wave-slice of
// G = Cin/VEC lanes (<= 64) per output pixel, reduced with shuffles.
template <typename T>
__global__ __launch_bounds__(256) void fwd_cout1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N;
    const int G = p.Cin / VEC;                       // lanes per pixel (power of two, <= 64)
    const int ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;         // packed row 0: [tap*Cin + ci]
    T* __restrict__ y = (T*)p.y;
    const long M = (long)p.N * p.Ho * p.Wo;
    for (long o = (long)blockIdx.x * ppb + ty; o < M; o += (long)gridDim.x * ppb) {
        const int n = (int)(o / (p.Ho * p.Wo)), rem = (int)(o - (long)n * p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        float acc = 0.f;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.stride - p.pad + ky;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.stride - p.pad + kx;
                if ((unsigned)ix >= (unsigned)p.Wi) continue;
                float xv[VEC], wv[VEC];
                unpack16<T>(*(const u32x4_t*)(x + (((size_t)n * p.Hi + iy) * p.Wi + ix) * p.Cin + tx * VEC), xv);
                unpack16<T>(*(const u32x4_t*)(w + (size_t)(ky * p.KW + kx) * p.Cin + tx * VEC), wv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc += (p.in_act == S2E_ACT_LRELU ? lrelu02(xv[j]) : xv[j]) * wv[j];
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (tx == 0) {
            if (p.bias) acc += p.bias[0];
            if (p.res) acc += load1<T>((const T*)p.res + o);
            if (p.out_act == S2E_ACT_LRELU) acc = lrelu02(acc);
            else if (p.out_act == S2E_ACT_TANH) acc = tanhf(acc);
            store1<T>(y + o, acc);
        }
    }
}

// ---------------------------------------------------------------- forward, Cin == 1
// y[o][co] = out_act( bias[co] + sum_tap in_act(x[i(o,tap)]) * w[co][tap] );   weights [tap][co] in LDS
template <typename T>
__global__ __launch_bounds__(256) void fwd_cin1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N;
    extern __shared__ float wl[];                    // [taps][Cout]
    const int taps = p.KH * p.KW;
    const T* __restrict__ w = (const T*)p.w;         // packed [co][Kpad], column = tap
    for (int i = threadIdx.x; i < taps * p.Cout; i += 256) {
        const int t = i / p.Cout, co = i - t * p.Cout;
        wl[i] = load1<T>(w + (size_t)co * p.Kpad + t);
    }
    __syncthreads();
    const int G = p.Cout / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    if (ty >= ppb) return;
    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const long M = (long)p.N * p.Ho * p.Wo;
    for (long o = (long)blockIdx.x * ppb + ty; o < M; o += (long)gridDim.x * ppb) {
        const int n = (int)(o / (p.Ho * p.Wo)), rem = (int)(o - (long)n * p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = p.bias ? p.bias[tx * VEC + j] : 0.f;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.stride - p.pad + ky;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.stride - p.pad + kx;
                if ((unsigned)ix >= (unsigned)p.Wi) continue;
                float xs = load1<T>(x + ((size_t)n * p.Hi + iy) * p.Wi + ix);
                if (p.in_act == S2E_ACT_LRELU) xs = lrelu02(xs);
                const float* wr = wl + (ky * p.KW + kx) * p.Cout + tx * VEC;
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += xs * wr[j];
            }
        }
        const size_t oo = (size_t)o * p.Cout + tx * VEC;
        if (p.res) {
            float rr[VEC];
            unpack16<T>(*(const u32x4_t*)((const T*)p.res + oo), rr);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += rr[j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = p.out_act == S2E_ACT_LRELU ? lrelu02(acc[j]) : (p.out_act == S2E_ACT_TANH ? tanhf(acc[j]) : acc[j]);
        *(u32x4_t*)(y + oo) = pack16<T>(acc);
    }
}

// ---------------------------------------------------------------- data gradient of a Cout == 1 conv (stride 1)
// dx[q][c] = mask(aux[q][c]) * sum_tap gy[o(q,tap)] * w[c][tap],  o = q + pad - k;   weights [tap][C] in LDS
// (transposed pack: row c, column tap*1 + 0)
template <typename T>
__global__ __launch_bounds__(256) void dgrad_cout1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N;
    extern __shared__ float wl[];                    // [taps][C]   C = p.Cout of this launch (= conv Cin)
    const int taps = p.KH * p.KW, C = p.Cout;
    const T* __restrict__ w = (const T*)p.w;
    for (int i = threadIdx.x; i < taps * C; i += 256) {
        const int t = i / C, c = i - t * C;
        wl[i] = load1<T>(w + (size_t)c * p.Kpad + t);
    }
    __syncthreads();
    const int G = C / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    if (ty >= ppb) return;
    const T* __restrict__ gy = (const T*)p.x;        // (N, Hi, Wi, 1): the conv's output gradient
    const T* __restrict__ aux = (const T*)p.aux;
    T* __restrict__ dx = (T*)p.y;                    // (N, Ho, Wo, C)
    const long M = (long)p.N * p.Ho * p.Wo;
    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
    for (long q = (long)blockIdx.x * ppb + ty; q < M; q += (long)gridDim.x * ppb) {
        const int n = (int)(q / (p.Ho * p.Wo)), rem = (int)(q - (long)n * p.Ho * p.Wo);
        const int qy = rem / p.Wo, qx = rem - qy * p.Wo;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int oy = qy + p.pad - ky;
            if ((unsigned)oy >= (unsigned)p.Hi) continue;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ox = qx + p.pad - kx;
                if ((unsigned)ox >= (unsigned)p.Wi) continue;
                const float g = load1<T>(gy + ((size_t)n * p.Hi + oy) * p.Wi + ox);
                const float* wr = wl + (ky * p.KW + kx) * C + tx * VEC;
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += g * wr[j];
            }
        }
        const size_t oo = (size_t)q * C + tx * VEC;
        if (p.aux_mode != S2E_AUX_NONE) {
            float aa[VEC];
            unpack16<T>(*(const u32x4_t*)(aux + oo), aa);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] *= (aa[j] > 0.f ? 1.f : neg);
        }
        *(u32x4_t*)(dx + oo) = pack16<T>(acc);
    }
}

// ---------------------------------------------------------------- weight gradient, Cout == 1 (stride 1)
// dw[tap*Cin + ci] += sum_q gy[o(q,tap)] * in_act(x[q][ci])       each x vector is read ONCE
template <typename T>
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float red[256 * VEC];
    const int G = p.Cin / VEC, ppb = 256 / G;        // G <= 64 -> ppb >= 4
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ gy = (const T*)p.gy;
    float acc[MAXT][VEC];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
    const long Q = (long)p.N * p.Hi * p.Wi;
    const long per = (Q + gridDim.x - 1) / gridDim.x;
    const long q0 = (long)blockIdx.x * per, q1 = (q0 + per < Q) ? q0 + per : Q;
    for (long q = q0 + ty; q < q1; q += ppb) {
        const int n = (int)(q / (p.Hi * p.Wi)), rem = (int)(q - (long)n * p.Hi * p.Wi);
        const int qy = rem / p.Wi, qx = rem - qy * p.Wi;
        float xv[VEC];
        unpack16<T>(*(const u32x4_t*)(x + (size_t)q * p.Cin + tx * VEC), xv);
        if (p.in_act == S2E_ACT_LRELU) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) xv[j] = lrelu02(xv[j]);
        }
        static_for<0, MAXT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            constexpr int ky = t / 4, kx = t % 4;
            const int oy = qy + p.pad - ky, ox = qx + p.pad - kx;
            if (ky < p.KH && kx < p.KW && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo) {
                const float g = load1<T>(gy + ((size_t)n * p.Ho + oy) * p.Wo + ox);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[t][j] += g * xv[j];
            }
        });
    }
    // reduce over the block's pixel lanes, one tap at a time (compile-time tap index: acc stays in registers)
    static_for<0, MAXT>([&](auto TT) {
        constexpr int t = decltype(TT)::value;
        constexpr int ky = t / 4, kx = t % 4;
        if (ky < p.KH && kx < p.KW) {                 // block-uniform
            __syncthreads();
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[t][j];
            __syncthreads();
            if (ty == 0) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float a = 0.f;
                    for (int r = 0; r < ppb; ++r) a += red[(r * G + tx) * VEC + j];
                    atomicAdd(p.dw + (size_t)(ky * p.KW + kx) * p.Cin + tx * VEC + j, a);
                }
            }
        }
    });
}

// ---------------------------------------------------------------- weight gradient, Cin == 1
// dw[co][tap] += sum_o gy[o][co] * in_act(x[i(o,tap)])
template <typename T>
__global__ __launch_bounds__(256) void wgrad_cin1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N;
    __shared__ float red[256 * VEC];
    const int G = p.Cout / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ gy = (const T*)p.gy;
    float acc[MAXT][VEC];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
    const long M = (long)p.N * p.Ho * p.Wo;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long o0 = (long)blockIdx.x * per, o1 = (o0 + per < M) ? o0 + per : M;
    if (ty < ppb)
        for (long o = o0 + ty; o < o1; o += ppb) {
            const int n = (int)(o / (p.Ho * p.Wo)), rem = (int)(o - (long)n * p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            float gv[VEC];
            unpack16<T>(*(const u32x4_t*)(gy + (size_t)o * p.Cout + tx * VEC), gv);
            static_for<0, MAXT>([&](auto TT) {
                constexpr int t = decltype(TT)::value;
                constexpr int ky = t / 4, kx = t % 4;
                const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                if (ky < p.KH && kx < p.KW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
                    float xs = load1<T>(x + ((size_t)n * p.Hi + iy) * p.Wi + ix);
                    if (p.in_act == S2E_ACT_LRELU) xs = lrelu02(xs);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[t][j] += gv[j] * xs;
                }
            });
        }
    static_for<0, MAXT>([&](auto TT) {
        constexpr int t = decltype(TT)::value;
        constexpr int ky = t / 4, kx = t % 4;
        if (ky < p.KH && kx < p.KW) {                 // block-uniform
            __syncthreads();
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = (ty < ppb) ? acc[t][j] : 0.f;
            __syncthreads();
            if (ty == 0) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float a = 0.f;
                    for (int r = 0; r < ppb; ++r) a += red[(r * G + tx) * VEC + j];
                    atomicAdd(p.dw + (size_t)(tx * VEC + j) * (p.KH * p.KW) + ky * p.KW + kx, a);
                }
            }
        }
    });
}

// ---------------------------------------------------------------- host dispatch
static bool pow2_le64(int g) { return g >= 1 && g <= 64 && (g & (g - 1)) == 0; }

template <typename T>
static int small_fwd(const SmallConvParams& p, int kind, hipStream_t st) {
    const long M = (long)p.N * p.Ho * p.Wo;
    const int vec = Vec<T>::N;
    if (kind == SMALL_FWD_COUT1) {
        const int ppb = 256 / (p.Cin / vec);
        const int grid = (int)((M + ppb - 1) / ppb < 8192 ? (M + ppb - 1) / ppb : 8192);
        fwd_cout1_kernel<T><<<grid, 256, 0, st>>>(p);
    } else {
        const int ppb = 256 / (p.Cout / vec);
        const int grid = (int)((M + ppb - 1) / ppb < 8192 ? (M + ppb - 1) / ppb : 8192);
        const size_t lds = (size_t)p.KH * p.KW * p.Cout * sizeof(float);
        if (kind == SMALL_FWD_CIN1) fwd_cin1_kernel<T><<<grid, 256, lds, st>>>(p);
        else dgrad_cout1_kernel<T><<<grid, 256, lds, st>>>(p);
    }
    S2E_CHECK_LAUNCH("small conv kernel");
    return S2E_OK;
}

int s2e_small_conv_kind(int dtype, const s2e_conv_desc* d) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (d->KH > 4 || d->KW > 4) return SMALL_NONE;
    if (!d->transposed && d->Cout == 1 && d->Cin % vec == 0 && pow2_le64(d->Cin / vec) && d->aux_mode == S2E_AUX_NONE)
        return SMALL_FWD_COUT1;
    if (!d->transposed && d->Cin == 1 && d->Cout % vec == 0 && pow2_le64(d->Cout / vec) && d->Cout <= 512 &&
        d->aux_mode == S2E_AUX_NONE)
        return SMALL_FWD_CIN1;
    if (d->transposed && d->stride == 1 && d->Cin == 1 && d->Cout % vec == 0 && pow2_le64(d->Cout / vec) && d->Cout <= 512 &&
        d->in_act == S2E_ACT_NONE && d->out_act == S2E_ACT_NONE)
        return SMALL_DGRAD_COUT1;
    return SMALL_NONE;
}

int s2e_small_conv_launch(int dtype, int kind, const SmallConvParams& p, hipStream_t st) {
    return dtype == S2E_BF16 ? small_fwd<bf16_t>(p, kind, st) : small_fwd<float>(p, kind, st);
}

int s2e_small_wgrad_kind(int dtype, const s2e_conv_desc* d) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (d->KH > 4 || d->KW > 4) return SMALL_NONE;
    if (d->Cout == 1 && d->stride == 1 && d->Cin % vec == 0 && pow2_le64(d->Cin / vec)) return SMALL_WGRAD_COUT1;
    if (d->Cin == 1 && d->Cout % vec == 0 && pow2_le64(d->Cout / vec)) return SMALL_WGRAD_CIN1;
    return SMALL_NONE;
}

int s2e_small_wgrad_launch(int dtype, int kind, const SmallConvParams& p, hipStream_t st) {
    const long work = kind == SMALL_WGRAD_COUT1 ? (long)p.N * p.Hi * p.Wi : (long)p.N * p.Ho * p.Wo;
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    const int ppb = 256 / ((kind == SMALL_WGRAD_COUT1 ? p.Cin : p.Cout) / vec);     // pixels per pass of one block
    long g = (work + (long)ppb * 8 - 1) / ((long)ppb * 8);                          // >= 8 passes per block ...
    if (g > 256) g = 256;       // ... one block per CU at most: every block ends with atomics on the SAME small dW
    const int grid = (int)(g < 1 ? 1 : g);
    if (dtype == S2E_BF16) {
        if (kind == SMALL_WGRAD_COUT1) wgrad_cout1_kernel<bf16_t><<<grid, 256, 0, st>>>(p);
        else wgrad_cin1_kernel<bf16_t><<<grid, 256, 0, st>>>(p);
    } else {
        if (kind == SMALL_WGRAD_COUT1) wgrad_cout1_kernel<float><<<grid, 256, 0, st>>>(p);
        else wgrad_cin1_kernel<float><<<grid, 256, 0, st>>>(p);
    }
    S2E_CHECK_LAUNCH("small wgrad kernel");
    return S2E_OK;
}
