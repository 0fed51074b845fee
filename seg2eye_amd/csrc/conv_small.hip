// Degenerate-channel convolutions: Cout == 1 (conv_img 64->1, the PatchGAN heads 512->1) and Cin == 1
// (encoder layer0 1->64).  A 128-wide MFMA tile does 1/32..1/128 useful work on these; they are pure
// HBM-bound streams, so each gets a small vector kernel: one thread owns a 16-byte channel group of the
// wide tensor and a handful of scalar loads of the 1-channel tensor.  Dispatched from s2e_conv2d /
// s2e_conv2d_wgrad (conv_igemm.hip / conv_wgrad.hip); no separate ABI.
//
// What makes them stream instead of crawl (each was 5-15x off the HBM time when written naively):
//   * the kernel size is a template parameter (3 or 4) and the tap loops are fully unrolled with CLAMPED
//     addresses + a validity select, so all tap loads of a pixel are independent and issued together
//     (a `continue` on the bounds test serialises them: one L2 round trip per tap);
//   * weights live in registers (read once per thread), not in an LDS table rebuilt by every block;
//   * pixel indices are 32-bit (the dispatch guarantees N*H*W < 2^31): no 64-bit divisions;
//   * the weight-gradient kernels keep two pixels in flight per thread, reduce lanes -> waves -> block
//     in registers/LDS and write ONE partial row per block to a workspace; a second tiny kernel sums the
//     rows (same-address float atomics from ~1000 blocks serialise for longer than the whole stream).
#include "common.h"
#include "conv_small.h"

static constexpr size_t SMALL_WS_CAP = 8u << 20;      // bytes of block partials at most

__device__ __forceinline__ void decode_px(int o, int HW, int W, int& n, int& y, int& x) {
    n = o / HW;
    const int rem = o - n * HW;
    y = rem / W;
    x = rem - y * W;
}
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// ---------------------------------------------------------------- forward, Cout == 1
// y[o] = out_act( bias + sum_{tap,ci} in_act(x[i(o,tap)][ci]) * w[tap*Cin + ci] )      one slice of
// G = Cin/VEC lanes (<= 64) per output pixel, reduced with shuffles.
template <typename T, int KS>
__global__ __launch_bounds__(256) void fwd_cout1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    const int G = p.Cin / VEC;                       // lanes per pixel (power of two, <= 64)
    const int ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;         // packed row 0: [tap*Cin + ci]
    T* __restrict__ y = (T*)p.y;
    u32x4_t wr[NT];
    static_for<0, NT>([&](auto TT) {
        constexpr int t = decltype(TT)::value;
        wr[t] = *(const u32x4_t*)(w + (size_t)t * p.Cin + tx * VEC);
    });
    const int M = p.N * p.Ho * p.Wo, HW = p.Ho * p.Wo;
    for (int o = blockIdx.x * ppb + ty; o < M; o += gridDim.x * ppb) {
        int n, oy, ox;
        decode_px(o, HW, p.Wo, n, oy, ox);
        u32x4_t xr[NT];
        bool ok[NT];
        static_for<0, NT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int iy = oy * p.stride - p.pad + t / KS, ix = ox * p.stride - p.pad + t % KS;
            ok[t] = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            xr[t] = *(const u32x4_t*)(x + (((size_t)n * p.Hi + clampi(iy, p.Hi - 1)) * p.Wi + clampi(ix, p.Wi - 1)) * p.Cin + tx * VEC);
        });
        float acc = 0.f;
        static_for<0, NT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            float xv[VEC], wv[VEC], s = 0.f;
            unpack16<T>(xr[t], xv);
            unpack16<T>(wr[t], wv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += (p.in_act == S2E_ACT_LRELU ? lrelu02(xv[j]) : xv[j]) * wv[j];
            acc += ok[t] ? s : 0.f;
        });
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

// weights of a [channel][Kpad] matrix whose first NT columns are the taps -> registers wf[j][t]
template <typename T, int NT, int VEC>
__device__ __forceinline__ void load_rows(const T* __restrict__ w, int Kpad, int c0, float (&wf)[VEC][NT]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[j][t] = load1<T>(w + (size_t)(c0 + j) * Kpad + t);
}

// ---------------------------------------------------------------- forward, Cin == 1
// y[o][co] = out_act( bias[co] + sum_tap in_act(x[i(o,tap)]) * w[co][tap] )
template <typename T, int KS>
__global__ __launch_bounds__(256) void fwd_cin1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    const int G = p.Cout / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    float wf[VEC][NT];
    load_rows<T, NT, VEC>((const T*)p.w, p.Kpad, tx * VEC, wf);       // packed [co][Kpad], column = tap
    float bv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bv[j] = p.bias ? p.bias[tx * VEC + j] : 0.f;
    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const int M = p.N * p.Ho * p.Wo, HW = p.Ho * p.Wo;
    for (int o = blockIdx.x * ppb + ty; o < M; o += gridDim.x * ppb) {
        int n, oy, ox;
        decode_px(o, HW, p.Wo, n, oy, ox);
        float xs[NT];
        static_for<0, NT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int iy = oy * p.stride - p.pad + t / KS, ix = ox * p.stride - p.pad + t % KS;
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            float v = load1<T>(x + ((size_t)n * p.Hi + clampi(iy, p.Hi - 1)) * p.Wi + clampi(ix, p.Wi - 1));
            if (p.in_act == S2E_ACT_LRELU) v = lrelu02(v);
            xs[t] = ok ? v : 0.f;
        });
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[j] = bv[j];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[j] += xs[t] * wf[j][t];
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
// dx[q][c] = mask(aux[q][c]) * sum_tap gy[o(q,tap)] * w[c][tap],  o = q + pad - k
// (transposed pack: row c, column tap*1 + 0;  this launch's "Cout" is the conv's Cin)
template <typename T, int KS>
__global__ __launch_bounds__(256) void dgrad_cout1_kernel(SmallConvParams p) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    const int C = p.Cout, G = C / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    float wf[VEC][NT];
    load_rows<T, NT, VEC>((const T*)p.w, p.Kpad, tx * VEC, wf);
    const T* __restrict__ gy = (const T*)p.x;        // (N, Hi, Wi, 1): the conv's output gradient
    const T* __restrict__ aux = (const T*)p.aux;
    T* __restrict__ dx = (T*)p.y;                    // (N, Ho, Wo, C)
    const int M = p.N * p.Ho * p.Wo, HW = p.Ho * p.Wo;
    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
    for (int q = blockIdx.x * ppb + ty; q < M; q += gridDim.x * ppb) {
        int n, qy, qx;
        decode_px(q, HW, p.Wo, n, qy, qx);
        float gs[NT];
        static_for<0, NT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int oy = qy + p.pad - t / KS, ox = qx + p.pad - t % KS;
            const bool ok = (unsigned)oy < (unsigned)p.Hi && (unsigned)ox < (unsigned)p.Wi;
            const float v = load1<T>(gy + ((size_t)n * p.Hi + clampi(oy, p.Hi - 1)) * p.Wi + clampi(ox, p.Wi - 1));
            gs[t] = ok ? v : 0.f;
        });
        const size_t oo = (size_t)q * C + tx * VEC;
        float aa[VEC];
        if (p.aux_mode != S2E_AUX_NONE) unpack16<T>(*(const u32x4_t*)(aux + oo), aa);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            acc[j] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[j] += gs[t] * wf[j][t];
            if (p.aux_mode != S2E_AUX_NONE) acc[j] *= (aa[j] > 0.f ? 1.f : neg);
        }
        *(u32x4_t*)(dx + oo) = pack16<T>(acc);
    }
}

// ---------------------------------------------------------------- weight gradients: block partial -> workspace
// acc[t][j] holds this thread's partial for (tap t, channel tx*VEC+j).  Lanes that share tx inside a wave are
// summed with shuffles, the four waves through LDS, and the block's row (NT*C floats, element index given by
// idx(t, c)) is stored to ws[blockIdx.x][*].
template <int NT, int VEC, typename IDX>
__device__ __forceinline__ void block_partial_out(float (&acc)[NT][VEC], int G, int tx, float* red, float* __restrict__ ws_row,
                                                  int nout, IDX idx) {
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[t][j] += __shfl_xor(acc[t][j], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv && lane < G) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int i = idx(t, tx * VEC + j);
                    red[i] = (wv == 0 ? 0.f : red[i]) + acc[t][j];
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < nout; i += 256) ws_row[i] = red[i];
}

// ---------------------------------------------------------------- weight gradient, Cout == 1 (stride 1)
// dw[tap*Cin + ci] += sum_q gy[o(q,tap)] * in_act(x[q][ci])       each x vector is read ONCE
template <typename T, int KS>
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(SmallConvParams p, float* __restrict__ ws) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    extern __shared__ float red[];                   // [NT * Cin]
    const int G = p.Cin / VEC, ppb = 256 / G;        // G <= 64 -> ppb >= 4
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ gy = (const T*)p.gy;
    float acc[NT][VEC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
    const int Q = p.N * p.Hi * p.Wi, HW = p.Hi * p.Wi;
    const int per = (Q + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per, q1 = (q0 + per < Q) ? q0 + per : Q;
    auto consume = [&](u32x4_t xr, int q, bool live) __attribute__((always_inline)) {
        int n, qy, qx;
        decode_px(q, HW, p.Wi, n, qy, qx);
        float xv[VEC];
        unpack16<T>(xr, xv);
        if (p.in_act == S2E_ACT_LRELU) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) xv[j] = lrelu02(xv[j]);
        }
        float gs[NT];
        static_for<0, NT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int oy = qy + p.pad - t / KS, ox = qx + p.pad - t % KS;
            const bool ok = live && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
            const float v = load1<T>(gy + ((size_t)n * p.Ho + clampi(oy, p.Ho - 1)) * p.Wo + clampi(ox, p.Wo - 1));
            gs[t] = ok ? v : 0.f;
        });
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[t][j] += gs[t] * xv[j];
    };
    for (int q = q0 + ty; q < q1; q += 2 * ppb) {
        const int qb = q + ppb;
        const bool lb = qb < q1;
        const int qbc = lb ? qb : q;
        const u32x4_t xa = *(const u32x4_t*)(x + (size_t)q * p.Cin + tx * VEC);
        const u32x4_t xb = *(const u32x4_t*)(x + (size_t)qbc * p.Cin + tx * VEC);
        consume(xa, q, true);
        consume(xb, qbc, lb);
    }
    const int Cin = p.Cin;
    block_partial_out<NT, VEC>(acc, G, tx, red, ws + (size_t)blockIdx.x * NT * Cin, NT * Cin,
                               [Cin](int t, int c) { return t * Cin + c; });
}

// ---------------------------------------------------------------- weight gradient, Cin == 1
// dw[co][tap] += sum_o gy[o][co] * in_act(x[i(o,tap)])
template <typename T, int KS>
__global__ __launch_bounds__(256) void wgrad_cin1_kernel(SmallConvParams p, float* __restrict__ ws) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    extern __shared__ float red[];                   // [Cout * NT]
    const int G = p.Cout / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ gy = (const T*)p.gy;
    float acc[NT][VEC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
    const int M = p.N * p.Ho * p.Wo, HW = p.Ho * p.Wo;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int o0 = blockIdx.x * per, o1 = (o0 + per < M) ? o0 + per : M;
    auto consume = [&](u32x4_t gr, int o, bool live) __attribute__((always_inline)) {
        int n, oy, ox;
        decode_px(o, HW, p.Wo, n, oy, ox);
        float gv[VEC];
        unpack16<T>(gr, gv);
        float xs[NT];
        static_for<0, NT>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int iy = oy * p.stride - p.pad + t / KS, ix = ox * p.stride - p.pad + t % KS;
            const bool ok = live && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            float v = load1<T>(x + ((size_t)n * p.Hi + clampi(iy, p.Hi - 1)) * p.Wi + clampi(ix, p.Wi - 1));
            if (p.in_act == S2E_ACT_LRELU) v = lrelu02(v);
            xs[t] = ok ? v : 0.f;
        });
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[t][j] += xs[t] * gv[j];
    };
    for (int o = o0 + ty; o < o1; o += 2 * ppb) {
        const int ob = o + ppb;
        const bool lb = ob < o1;
        const int obc = lb ? ob : o;
        const u32x4_t ga = *(const u32x4_t*)(gy + (size_t)o * p.Cout + tx * VEC);
        const u32x4_t gb = *(const u32x4_t*)(gy + (size_t)obc * p.Cout + tx * VEC);
        consume(ga, o, true);
        consume(gb, obc, lb);
    }
    block_partial_out<NT, VEC>(acc, G, tx, red, ws + (size_t)blockIdx.x * NT * p.Cout, NT * p.Cout,
                               [](int t, int c) { return c * NT + t; });
}

// dw[i] += sum_b ws[b][i]:   64 outputs x 4 row phases per block, gridDim.y row slabs
__global__ __launch_bounds__(256) void small_wgrad_reduce_kernel(const float* __restrict__ ws, int nb, int nout, float* __restrict__ dw) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < nout) {
        const int stride = 4 * gridDim.y;
        int b = blockIdx.y * 4 + ph;
        for (; b + 3 * stride < nb; b += 4 * stride) {
            a0 += ws[(size_t)b * nout + i];
            a1 += ws[(size_t)(b + stride) * nout + i];
            a2 += ws[(size_t)(b + 2 * stride) * nout + i];
            a3 += ws[(size_t)(b + 3 * stride) * nout + i];
        }
        for (; b < nb; b += stride) a0 += ws[(size_t)b * nout + i];
    }
    red[ph][il] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ph == 0 && i < nout) atomicAdd(dw + i, (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]));
}

// ---------------------------------------------------------------- host dispatch
static bool pow2_le64(int g) { return g >= 1 && g <= 64 && (g & (g - 1)) == 0; }
static int small_ks(const s2e_conv_desc* d) { return (d->KH == d->KW && (d->KH == 3 || d->KH == 4)) ? d->KH : 0; }

template <typename T, int KS>
static int small_fwd(const SmallConvParams& p, int kind, hipStream_t st) {
    const long M = (long)p.N * p.Ho * p.Wo;
    const int vec = Vec<T>::N;
    const int ppb = 256 / ((kind == SMALL_FWD_COUT1 ? p.Cin : p.Cout) / vec);
    long g = (M + ppb - 1) / ppb;
    const long cap = kind == SMALL_FWD_COUT1 ? 4096 : 1024;        // register-resident weights: amortise their load
    if (g > cap) g = cap;
    const int grid = (int)(g < 1 ? 1 : g);
    if (kind == SMALL_FWD_COUT1) fwd_cout1_kernel<T, KS><<<grid, 256, 0, st>>>(p);
    else if (kind == SMALL_FWD_CIN1) fwd_cin1_kernel<T, KS><<<grid, 256, 0, st>>>(p);
    else dgrad_cout1_kernel<T, KS><<<grid, 256, 0, st>>>(p);
    S2E_CHECK_LAUNCH("small conv kernel");
    return S2E_OK;
}

int s2e_small_conv_kind(int dtype, const s2e_conv_desc* d) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (!small_ks(d)) return SMALL_NONE;
    if (!d->transposed && d->Cout == 1 && d->Cin % vec == 0 && pow2_le64(d->Cin / vec) && d->aux_mode == S2E_AUX_NONE)
        return SMALL_FWD_COUT1;
    if (!d->transposed && d->Cin == 1 && d->Cout % vec == 0 && pow2_le64(d->Cout / vec) && d->aux_mode == S2E_AUX_NONE)
        return SMALL_FWD_CIN1;
    if (d->transposed && d->stride == 1 && d->Cin == 1 && d->Cout % vec == 0 && pow2_le64(d->Cout / vec) &&
        d->in_act == S2E_ACT_NONE && d->out_act == S2E_ACT_NONE)
        return SMALL_DGRAD_COUT1;
    return SMALL_NONE;
}

int s2e_small_conv_launch(int dtype, int kind, const SmallConvParams& p, hipStream_t st) {
    if (p.KH == 3) return dtype == S2E_BF16 ? small_fwd<bf16_t, 3>(p, kind, st) : small_fwd<float, 3>(p, kind, st);
    return dtype == S2E_BF16 ? small_fwd<bf16_t, 4>(p, kind, st) : small_fwd<float, 4>(p, kind, st);
}

int s2e_small_wgrad_kind(int dtype, const s2e_conv_desc* d) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (!small_ks(d)) return SMALL_NONE;
    if (d->Cout == 1 && d->stride == 1 && d->Cin % vec == 0 && pow2_le64(d->Cin / vec)) return SMALL_WGRAD_COUT1;
    if (d->Cin == 1 && d->Cout % vec == 0 && pow2_le64(d->Cout / vec)) return SMALL_WGRAD_CIN1;
    return SMALL_NONE;
}

// grid of the partial kernel: >= 4 pixel passes per block, <= 1024 blocks, partial rows within SMALL_WS_CAP
static int small_wgrad_grid(int dtype, int kind, const s2e_conv_desc* d) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    const int c = kind == SMALL_WGRAD_COUT1 ? d->Cin : d->Cout;
    const long work = kind == SMALL_WGRAD_COUT1 ? (long)d->N * d->Hi * d->Wi : (long)d->N * d->Ho * d->Wo;
    const int ppb = 256 / (c / vec);
    long g = (work + (long)ppb * 4 - 1) / ((long)ppb * 4);
    if (g > 1024) g = 1024;
    const long by_ws = (long)(SMALL_WS_CAP / ((size_t)d->KH * d->KW * c * sizeof(float)));
    if (g > by_ws) g = by_ws;
    return (int)(g < 1 ? 1 : g);
}

size_t s2e_small_wgrad_workspace_bytes(int dtype, int kind, const s2e_conv_desc* d) {
    const int c = kind == SMALL_WGRAD_COUT1 ? d->Cin : d->Cout;
    return (size_t)small_wgrad_grid(dtype, kind, d) * d->KH * d->KW * c * sizeof(float);
}

template <typename T, int KS>
static void small_wgrad_go(int kind, const SmallConvParams& p, int grid, size_t lds, float* ws, hipStream_t st) {
    if (kind == SMALL_WGRAD_COUT1) wgrad_cout1_kernel<T, KS><<<grid, 256, lds, st>>>(p, ws);
    else wgrad_cin1_kernel<T, KS><<<grid, 256, lds, st>>>(p, ws);
}

int s2e_small_wgrad_launch(int dtype, int kind, const s2e_conv_desc* d, const SmallConvParams& p, void* workspace,
                           size_t workspace_bytes, hipStream_t st) {
    const size_t need = s2e_small_wgrad_workspace_bytes(dtype, kind, d);
    if (!workspace || workspace_bytes < need)
        S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad: this shape needs a %zu-byte workspace (got %zu); see s2e_conv2d_wgrad_workspace_bytes",
                 need, workspace_bytes);
    const int grid = small_wgrad_grid(dtype, kind, d);
    const int c = kind == SMALL_WGRAD_COUT1 ? d->Cin : d->Cout;
    const int nout = d->KH * d->KW * c;
    const size_t lds = (size_t)nout * sizeof(float);
    float* ws = (float*)workspace;
    if (d->KH == 3) { if (dtype == S2E_BF16) small_wgrad_go<bf16_t, 3>(kind, p, grid, lds, ws, st); else small_wgrad_go<float, 3>(kind, p, grid, lds, ws, st); }
    else            { if (dtype == S2E_BF16) small_wgrad_go<bf16_t, 4>(kind, p, grid, lds, ws, st); else small_wgrad_go<float, 4>(kind, p, grid, lds, ws, st); }
    int slabs = grid / 32;
    slabs = slabs < 1 ? 1 : (slabs > 8 ? 8 : slabs);
    if (s2e_deterministic()) slabs = 1;              // (several row slabs are combined with float atomics)
    small_wgrad_reduce_kernel<<<dim3((nout + 63) / 64, slabs), 256, 0, st>>>(ws, grid, nout, p.dw);
    S2E_CHECK_LAUNCH("small wgrad kernels");
    return S2E_OK;
}
