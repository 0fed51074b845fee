// Degenerate-channel convolutions: Cout == 1 (conv_img 64->1, the PatchGAN heads 512->1) and Cin == 1 (encoder layer0 1->64).
// A 128-wide MFMA tile does 1/32..1/128 useful work on these as an implicit GEMM; they are HBM-bound streams of ONE wide tensor
// (16 B a lane) beside a 1-channel one.  Dispatched from s2e_conv2d / s2e_conv2d_wgrad (conv_igemm.hip / conv_wgrad.hip); no separate
// ABI.  Three generations live here (round 5 rewrote the hot ones; tools/bench_small.py times all of them at the step's sizes):
//   * bf16, the sizes of the benchmark -- on the matrix cores after all, with the TAPS where an implicit GEMM has channels:
//       forward Cout == 1 (Cin 64 / 128 / 256 / 512, stride 1)  : fwd_cout1_mfma_kernel, dot-then-stencil        71 -> 21 us
//       forward Cin == 1, data gradient of Cout == 1 (C % 32 == 0): tap_gemm_kernel, the taps as one K-step   44 -> 23, 37 -> 18 us
//   * any dtype / channel count -- "band" vector kernels: a work item's patch of the 1-channel tensor is copied to LDS once and a
//     pixel's taps are LDS broadcasts (fwd_cin1_kernel, dgrad_cout1_kernel, and the two weight gradients, which have no MFMA form yet);
//   * fwd_cout1_kernel, the original vector kernel, for the remaining Cout == 1 forwards (fp32, odd channel counts).
// What the vector kernels keep from round 1: kernel size as a template parameter with fully unrolled tap loops (no `continue` on
// the bounds test: that serialises the loads), weights in registers (16-B row loads), 32-bit pixel indices, weight gradients as
// one partial row per block to a workspace + a tiny fixed-order reduce kernel (same-address float atomics from ~1000 blocks
// serialise for longer than the whole stream).
#include "common.h"
#include "conv_small.h"

static constexpr size_t SMALL_WS_CAP = 8u << 20;      // bytes of block partials at most
static constexpr int SMALL_INFLIGHT = 2;               // pixels a weight-gradient thread has in flight (4 measured 10 % slower)

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

// ---------------------------------------------------------------- forward, Cout == 1, bf16, Cin in {64, 128, 256, 512}: dot, then stencil
// The kernel above reads every input vector once per TAP (9 or 16 times, from L1 / L2): the 64 -> 1 image conv ran at 0.95 TB/s
// of input, the 512 -> 1 PatchGAN heads at 0.4.  Here a workgroup owns a rectangle of outputs and reads each pixel of the
// rectangle's input patch ONCE:
//   dot     : d[q][t] = sum_ci in_act(x[q][ci]) * w[t][ci] for every patch pixel q and tap t -- a [pixels x Cin] x [Cin x taps]
//             product on the matrix cores (v_mfma_f32_32x32x16_bf16: 32 pixels a row block, taps in columns 0..KS*KS-1, the other
//             columns' weights zero).  A lane's operand is 16 B of one pixel, loaded from global in the fragment layout (two lanes a
//             32-B sector, the K-steps of a row block complete the lines); pixels outside the image are zeroed after the load.
//             d goes to LDS as fp32, row stride odd (9 / 17 words).
//   stencil : y[o] = out_act(bias + sum_t d[o + t][t] (+ res[o])), a thread per output, KS*KS conflict-free LDS reads.
// Four waves = four row-block phases (Cin <= 128), or four channel quarters of every row block (Cin >= 256: the heads' patches are
// a handful of row blocks) with one LDS slab per quarter, summed in the stencil in a fixed order.
// in_act: LeakyReLU is applied to the operand and the result rounded to bf16 (the MFMA's input type), where the vector kernel above
// multiplied the fp32 value.
__device__ __forceinline__ u32x4_t lrelu_bf16x8(u32x4_t r) {
    float f[8];
    unpack16<bf16_t>(r, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = lrelu02(f[j]);
    return u32x4_t{pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7])};
}

template <int KS, int NKW>       // NKW: K-steps (16 channels) a wave multiplies per row block = Cin / 16 / (channel quarters)
__global__ __launch_bounds__(256) void fwd_cout1_mfma_kernel(SmallConvParams p, int tw, int th, int tiles_x, int tiles_y, int wk) {
    constexpr int NT = KS * KS, NTP = NT | 1;
    extern __shared__ float dl[];                    // [wk][PP][NTP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int PW = tw + KS - 1, PH = th + KS - 1, PP = PW * PH, NRB = (PP + 31) >> 5;
    int b = blockIdx.x;
    const int txi = b % tiles_x; b /= tiles_x;
    const int tyi = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = tyi * th, ox0 = txi * tw;
    const int rb0 = wk == 1 ? wave : 0, rbs = wk == 1 ? 4 : 1, kq = wk == 1 ? 0 : wave;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x + (size_t)n * p.Hi * p.Wi * p.Cin + kq * NKW * 16 + h * 8;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w + kq * NKW * 16 + h * 8;         // packed row 0: [tap * Cin + ci]
    u32x4_t bw[NKW];
#pragma unroll
    for (int j = 0; j < NKW; ++j) {
        bw[j] = *(const u32x4_t*)(w + (size_t)min(l31, NT - 1) * p.Cin + j * 16);
        if (l31 >= NT) bw[j] = u32x4_t{0u, 0u, 0u, 0u};
    }
    float* __restrict__ dq = dl + (size_t)kq * PP * NTP;
    for (int rb = rb0; rb < NRB; rb += rbs) {
        const int pp = rb * 32 + l31;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = oy0 - p.pad + py, ix = ox0 - p.pad + px;
        const bool ok = pp < PP && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const bf16_t* xp = x + ((size_t)clampi(iy, p.Hi - 1) * p.Wi + clampi(ix, p.Wi - 1)) * p.Cin;
        u32x4_t a[NKW];
#pragma unroll
        for (int j = 0; j < NKW; ++j) a[j] = *(const u32x4_t*)(xp + j * 16);
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < NKW; ++j) {
            u32x4_t v = a[j];
            if (p.in_act == S2E_ACT_LRELU) v = lrelu_bf16x8(v);
            if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, v), __builtin_bit_cast(bf16x8_t, bw[j]), acc, 0, 0, 0);
        }
        if (l31 < NT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (q < PP) dq[q * NTP + l31] = acc[r];
            }
        }
    }
    __syncthreads();
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const float bias = p.bias ? p.bias[0] : 0.f;
    for (int i = tid; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        float acc = 0.f;
        for (int k = 0; k < wk; ++k) {
            const float* dk = dl + ((size_t)k * PP + ty * PW + tx) * NTP;
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) s += dk[((t / KS) * PW + t % KS) * NTP + t];
            acc += s;
        }
        const size_t o = ((size_t)n * p.Ho + oy) * p.Wo + ox;
        acc += bias;
        if (p.res) acc += load1<bf16_t>((const bf16_t*)p.res + o);
        if (p.out_act == S2E_ACT_LRELU) acc = lrelu02(acc);
        else if (p.out_act == S2E_ACT_TANH) acc = tanhf(acc);
        store1<bf16_t>(y + o, acc);
    }
}

// rectangle of the dot-then-stencil kernel: the largest square (<= 32) that still gives one workgroup per CU, d within 48 KB of LDS
static bool cout1_mfma_plan(const SmallConvParams& p, int* t_out, int* wk_out, int* nkw_out) {
    if (p.stride != 1 || p.KH != p.KW || (p.Cin != 64 && p.Cin != 128 && p.Cin != 256 && p.Cin != 512)) return false;
    const int wk = p.Cin >= 256 ? 4 : 1, ntp = (p.KH * p.KW) | 1;
    static const int cand[] = {32, 24, 16, 12, 9, 8, 7, 6};
    int pick = 0;
    for (int t : cand) {
        const int tt = t < (p.Ho > p.Wo ? p.Ho : p.Wo) ? t : (p.Ho > p.Wo ? p.Ho : p.Wo);
        const long pp = (long)(tt + p.KH - 1) * (tt + p.KH - 1);
        if (pp * ntp * 4 * wk > 48 * 1024) continue;
        pick = tt;
        if ((long)p.N * ceil_div(p.Ho, tt) * ceil_div(p.Wo, tt) >= 256) break;
    }
    if (!pick) return false;
    *t_out = pick; *wk_out = wk; *nkw_out = p.Cin / 16 / wk;
    return true;
}

template <int KS>
static int cout1_mfma_go(const SmallConvParams& p, int t, int wk, int nkw, hipStream_t st) {
    const int tx = ceil_div(p.Wo, t), ty = ceil_div(p.Ho, t);
    const size_t lds = (size_t)wk * (t + KS - 1) * (t + KS - 1) * ((KS * KS) | 1) * sizeof(float);
    if (nkw == 4) fwd_cout1_mfma_kernel<KS, 4><<<p.N * tx * ty, 256, lds, st>>>(p, t, t, tx, ty, wk);
    else          fwd_cout1_mfma_kernel<KS, 8><<<p.N * tx * ty, 256, lds, st>>>(p, t, t, tx, ty, wk);
    S2E_CHECK_LAUNCH("small conv kernel (dot, then stencil)");
    return S2E_OK;
}

// weights of a [channel][Kpad] matrix whose first NT columns are the taps -> registers wf[j][t].  16-B loads (Kpad is a multiple
// of 32 elements, rows start 16-B aligned): 16 of them a thread for the bf16 3x3, where one load per element was 72 -- more vector-memory
// instructions than the thread's whole pixel loop issued.
template <typename T, int NT, int VEC>
__device__ __forceinline__ void load_rows(const T* __restrict__ w, int Kpad, int c0, float (&wf)[VEC][NT]) {
    constexpr int NV = (NT + VEC - 1) / VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        float f[NV * VEC];
#pragma unroll
        for (int v = 0; v < NV; ++v) unpack16<T>(*(const u32x4_t*)(w + (size_t)(c0 + j) * Kpad + v * VEC), f + v * VEC);
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[j][t] = f[t];
    }
}

// ---------------------------------------------------------------- band kernels: the 1-channel tensor of a work item lives in LDS
// The four kernels below (forward Cin == 1, data gradient and weight gradient of Cout == 1, weight gradient of Cin == 1) stream
// the WIDE tensor once, 16 B per lane, and need KS*KS values of the 1-channel tensor per pixel.  As 2-byte global gathers those
// were 9-16 load instructions per pixel beside ONE for the wide tensor (the 64-channel layers ran at 1.3-1.8 TB/s of the wide
// tensor).  A work item is (image, band of rows, segment of <= 256 columns) of the wide tensor; the workgroup first copies the
// item's patch of the 1-channel tensor into LDS as fp32 -- zero outside the tensor, in_act already applied -- and a pixel's taps
// become KS*KS ds_read_b32 at fixed offsets from one base (8+ lanes share an address: broadcasts).
struct Band {
    int rows, segw, bands, segs, items;              // rows per band, columns per segment; items = N * bands * segs
    int H, W;                                        // the wide tensor's spatial size
    int h1, w1;                                      // the 1-channel tensor's
    int s, org, flip;                                // patch pixel of (row r, col c, tap ky, kx): (r*s + ky', c*s + kx'), ky' = flip ? KS-1-ky : ky;
                                                     // patch origin in the 1-channel tensor = (y0, x0) * s + org
    int pc;                                          // LDS row stride = (segw - 1) * s + KS
};
struct BandItem { int n, y0, x0, nr, nc; };
__device__ __forceinline__ BandItem band_item(const Band& bd, int item) {
    BandItem it;
    const int sg = item % bd.segs; item /= bd.segs;
    const int b = item % bd.bands;
    it.n = item / bd.bands;
    it.y0 = b * bd.rows; it.x0 = sg * bd.segw;
    it.nr = min(bd.rows, bd.H - it.y0); it.nc = min(bd.segw, bd.W - it.x0);
    return it;
}
// the item's patch of the 1-channel tensor -> LDS, zero outside the tensor.  Element e of the patch (row-major over the item's
// pr x pcu, LDS row stride bd.pc) belongs to thread e % 256; four of a thread's loads are requested before the first is stored
// (row by row with one load per loop trip this copy was a chain of up to 15 global round trips: longer than the item's pixel loop).
template <typename T, int KS, typename F>
__device__ __forceinline__ void band_stage_each(const Band& bd, const BandItem& it, const T* __restrict__ one, F&& put) {
    const int pr = (it.nr - 1) * bd.s + KS, pcu = (it.nc - 1) * bd.s + KS, ne = pr * pcu;
    const int r0 = it.y0 * bd.s + bd.org, c0 = it.x0 * bd.s + bd.org;
    const T* __restrict__ src = one + (size_t)it.n * bd.h1 * bd.w1;
    for (int e0 = threadIdx.x; e0 < ne; e0 += 4 * 256) {
        T v[4];
        int at[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + u * 256, ne - 1);
            const int r = e / pcu, c = e - r * pcu;
            const int y = r0 + r, xx = c0 + c;
            ok[u] = (unsigned)y < (unsigned)bd.h1 && (unsigned)xx < (unsigned)bd.w1;
            at[u] = r * bd.pc + c;
            v[u] = src[(size_t)clampi(y, bd.h1 - 1) * bd.w1 + clampi(xx, bd.w1 - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * 256 < ne) put(at[u], v[u], ok[u]);
    }
}
template <typename T, int KS>
__device__ __forceinline__ void band_stage(const Band& bd, const BandItem& it, const T* __restrict__ one, bool lrelu, float* __restrict__ pl) {
    band_stage_each<T, KS>(bd, it, one, [&](int at, T v, bool ok) {
        const float f = ok ? (float)v : 0.f;
        pl[at] = lrelu ? lrelu02(f) : f;
    });
}
// pixel slots of a workgroup walk the item's pixels in row-major order, `step` apart, without a division per pixel
struct Walk {
    int r, c;
    __device__ __forceinline__ void init(int i, int nc) { r = i / nc; c = i - r * nc; }
    __device__ __forceinline__ void advance(int step, int nc) { c += step; while (c >= nc) { c -= nc; ++r; } }
};
template <int KS> __device__ __forceinline__ int band_tap(const Band& bd, int t) {
    const int ky = t / KS, kx = t % KS;
    return bd.flip ? (KS - 1 - ky) * bd.pc + (KS - 1 - kx) : ky * bd.pc + kx;
}

// ---------------------------------------------------------------- forward, Cin == 1
// y[o][co] = out_act( bias[co] + sum_tap in_act(x[i(o,tap)]) * w[co][tap] )
template <typename T, int KS>
__global__ __launch_bounds__(256) void fwd_cin1_kernel(SmallConvParams p, Band bd) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    extern __shared__ float pl[];
    const int G = p.Cout / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    float wf[VEC][NT];
    load_rows<T, NT, VEC>((const T*)p.w, p.Kpad, tx * VEC, wf);       // packed [co][Kpad], column = tap
    float bv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bv[j] = p.bias ? p.bias[tx * VEC + j] : 0.f;
    T* __restrict__ y = (T*)p.y;
    for (int item = blockIdx.x; item < bd.items; item += gridDim.x) {
        const BandItem it = band_item(bd, item);
        __syncthreads();
        band_stage<T, KS>(bd, it, (const T*)p.x, p.in_act == S2E_ACT_LRELU, pl);
        __syncthreads();
        Walk wk;
        wk.init(ty, it.nc);
        for (; wk.r < it.nr; wk.advance(ppb, it.nc)) {
            const float* __restrict__ base = pl + wk.r * bd.s * bd.pc + wk.c * bd.s;
            float xs[NT];
            static_for<0, NT>([&](auto TT) { constexpr int t = decltype(TT)::value; xs[t] = base[band_tap<KS>(bd, t)]; });
            float acc[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[j] = bv[j];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[j] += xs[t] * wf[j][t];
            }
            const size_t oo = (((size_t)it.n * bd.H + it.y0 + wk.r) * bd.W + it.x0 + wk.c) * p.Cout + tx * VEC;
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
}

// ---------------------------------------------------------------- data gradient of a Cout == 1 conv (stride 1)
// dx[q][c] = mask(aux[q][c]) * sum_tap gy[o(q,tap)] * w[c][tap],  o = q + pad - k
// (transposed pack: row c, column tap*1 + 0;  this launch's "Cout" is the conv's Cin)
template <typename T, int KS>
__global__ __launch_bounds__(256) void dgrad_cout1_kernel(SmallConvParams p, Band bd) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    extern __shared__ float pl[];
    const int C = p.Cout, G = C / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    float wf[VEC][NT];
    load_rows<T, NT, VEC>((const T*)p.w, p.Kpad, tx * VEC, wf);
    const T* __restrict__ aux = (const T*)p.aux;
    T* __restrict__ dx = (T*)p.y;                    // (N, Ho, Wo, C)
    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
    for (int item = blockIdx.x; item < bd.items; item += gridDim.x) {
        const BandItem it = band_item(bd, item);
        __syncthreads();
        band_stage<T, KS>(bd, it, (const T*)p.x, false, pl);          // p.x: (N, Hi, Wi, 1), the conv's output gradient
        __syncthreads();
        Walk wk;
        wk.init(ty, it.nc);
        for (; wk.r < it.nr; wk.advance(ppb, it.nc)) {
            const float* __restrict__ base = pl + wk.r * bd.pc + wk.c;
            float gs[NT];
            static_for<0, NT>([&](auto TT) { constexpr int t = decltype(TT)::value; gs[t] = base[band_tap<KS>(bd, t)]; });
            const size_t oo = (((size_t)it.n * bd.H + it.y0 + wk.r) * bd.W + it.x0 + wk.c) * C + tx * VEC;
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
}

// ---------------------------------------------------------------- bf16: forward Cin == 1 and data gradient of Cout == 1 on the matrix cores
// Both are y[q][c] = sum_t one[q, t] * w[c][t] -- a [C x taps] x [taps x pixels] product with ONE K-step (9 or 16 taps).  On the
// vector ALU that is 72-128 FMAs per 16 B written (about 14 us of issue time per SIMD for the 64-channel layers at 256 x 256,
// beside a 17 us stream); here a wave multiplies 32 pixels x 32 channels per v_mfma_f32_32x32x16_bf16:
//   A (rows = channels)  : lane (i, h) holds taps h*8 .. h*8+7 of channel cb*32 + perm(i) -- 16 B of the packed matrix row, kept in
//                          registers for every channel block.  perm swaps bits 2 and 3 of the row, so that a lane's 16 results are
//                          channels h*8 .. h*8+7 and 16 + h*8 .. +7 of the block: two 16-B stores.
//   B (columns = pixels) : lane (pixel, h) gathers its 8 taps from the item's LDS patch (bf16, staged as in the band kernels).
//   epilogue             : bias / residual / activation (forward) or the activation mask (data gradient) on the accumulators.
__device__ __forceinline__ int tap_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int KS, bool DGRAD>
__global__ __launch_bounds__(256) void tap_gemm_kernel(SmallConvParams p, Band bd) {
    constexpr int NT = KS * KS;
    extern __shared__ uint16_t tp[];                 // [pr][pc] bf16 bits, then (forward) Cout floats of bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int C = p.Cout, NCB = C >> 5;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ aux = (const bf16_t*)p.aux;
    const bf16_t* __restrict__ res = (const bf16_t*)p.res;
    const float neg = (p.aux_mode == S2E_AUX_RELU_MASK) ? 0.f : 0.2f;
    const int pr_max = (bd.rows - 1) * bd.s + KS;
    float* __restrict__ bl = (float*)(tp + ((pr_max * bd.pc + 7) & ~7));
    if (!DGRAD && p.bias) for (int c = tid; c < C; c += 256) bl[c] = p.bias[c];
    // tap offsets of this lane's half of the K-step (patch elements); taps past KS*KS read tap 0 and meet zero weights
    int toff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = h * 8 + k;
        const int tt = t < NT ? t : 0;
        const int ky = tt / KS, kx = tt % KS;
        toff[k] = bd.flip ? (KS - 1 - ky) * bd.pc + (KS - 1 - kx) : ky * bd.pc + kx;
    }
    for (int item = blockIdx.x; item < bd.items; item += gridDim.x) {
        const BandItem it = band_item(bd, item);
        __syncthreads();
        // the item's patch of the 1-channel tensor -> LDS (bf16; zero outside the tensor; in_act applied)
        band_stage_each<uint16_t, KS>(bd, it, (const uint16_t*)p.x, [&](int at, uint16_t v, bool ok) {
            if (!ok) v = 0;
            if (!DGRAD && p.in_act == S2E_ACT_LRELU) v = (uint16_t)(pack2_bf16(lrelu02(bf16_bits_to_f32(v)), 0.f) & 0xffffu);
            tp[at] = v;
        });
        __syncthreads();
        const int npix = it.nr * it.nc;
        for (int i0 = wave * 32; i0 < npix; i0 += 128) {
            const int i = i0 + l31;
            const bool live = i < npix;
            const int ic = live ? i : npix - 1;
            const int r = ic / it.nc, c = ic - r * it.nc;
            const uint16_t* __restrict__ base = tp + (r * bd.pc + c) * bd.s;
            uint32_t bq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) bq[k] = (uint32_t)base[toff[2 * k]] | ((uint32_t)base[toff[2 * k + 1]] << 16);
            const u32x4_t bfrag = u32x4_t{bq[0], bq[1], bq[2], bq[3]};
            const size_t o0 = (((size_t)it.n * bd.H + it.y0 + r) * bd.W + it.x0 + c) * C + h * 8;
            for (int cb = 0; cb < NCB; ++cb) {
                const u32x4_t afrag = *(const u32x4_t*)(w + (size_t)(cb * 32 + tap_perm(l31)) * p.Kpad + h * 8);
                f32x16_t acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, afrag), __builtin_bit_cast(bf16x8_t, bfrag), acc, 0, 0, 0);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int ch = cb * 32 + half * 16 + h * 8;
                    const size_t oo = o0 + cb * 32 + half * 16;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = acc[half * 8 + j];
                    if (!DGRAD) {
                        if (p.bias) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += bl[ch + j];
                        }
                        if (res) {
                            float rr[8];
                            unpack16<bf16_t>(*(const u32x4_t*)(res + oo), rr);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += rr[j];
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = p.out_act == S2E_ACT_LRELU ? lrelu02(v[j]) : (p.out_act == S2E_ACT_TANH ? tanhf(v[j]) : v[j]);
                    } else if (p.aux_mode != S2E_AUX_NONE) {
                        float aa[8];
                        unpack16<bf16_t>(*(const u32x4_t*)(aux + oo), aa);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= (aa[j] > 0.f ? 1.f : neg);
                    }
                    if (live) *(u32x4_t*)(y + oo) = u32x4_t{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
                }
            }
        }
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
// dw[tap*Cin + ci] += sum_q gy[o(q,tap)] * in_act(x[q][ci])       each x vector is read ONCE, two pixels in flight per thread (four measured 10 % slower)
template <typename T, int KS>
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(SmallConvParams p, Band bd, float* __restrict__ ws) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    extern __shared__ float red[];                   // [NT * Cin] block partial, then the gy patch
    const int G = p.Cin / VEC, ppb = 256 / G;        // G <= 64 -> ppb >= 4
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    float* __restrict__ pl = red + NT * p.Cin;
    const T* __restrict__ x = (const T*)p.x;
    float acc[NT][VEC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
    for (int item = blockIdx.x; item < bd.items; item += gridDim.x) {
        const BandItem it = band_item(bd, item);
        __syncthreads();
        band_stage<T, KS>(bd, it, (const T*)p.gy, false, pl);
        __syncthreads();
        const T* __restrict__ xb = x + (((size_t)it.n * bd.H + it.y0) * bd.W + it.x0) * p.Cin + tx * VEC;
        auto consume = [&](u32x4_t xr, const Walk& w, bool live) __attribute__((always_inline)) {
            float xv[VEC];
            unpack16<T>(xr, xv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) xv[j] = live ? (p.in_act == S2E_ACT_LRELU ? lrelu02(xv[j]) : xv[j]) : 0.f;
            const float* __restrict__ base = pl + w.r * bd.pc + w.c;
            float gs[NT];
            static_for<0, NT>([&](auto TT) { constexpr int t = decltype(TT)::value; gs[t] = base[band_tap<KS>(bd, t)]; });
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[t][j] += gs[t] * xv[j];
        };
        Walk wa;
        wa.init(ty, it.nc);
        while (wa.r < it.nr) {                       // SMALL_INFLIGHT pixels' vectors requested before the first is consumed
            Walk wv[SMALL_INFLIGHT];
            bool lv[SMALL_INFLIGHT];
            u32x4_t xr[SMALL_INFLIGHT];
#pragma unroll
            for (int u = 0; u < SMALL_INFLIGHT; ++u) {
                lv[u] = wa.r < it.nr;
                wv[u] = lv[u] ? wa : wv[0];
                xr[u] = *(const u32x4_t*)(xb + ((size_t)wv[u].r * bd.W + wv[u].c) * p.Cin);
                if (lv[u]) wa.advance(ppb, it.nc);
            }
#pragma unroll
            for (int u = 0; u < SMALL_INFLIGHT; ++u) consume(xr[u], wv[u], lv[u]);
        }
    }
    __syncthreads();
    const int Cin = p.Cin;
    block_partial_out<NT, VEC>(acc, G, tx, red, ws + (size_t)blockIdx.x * NT * Cin, NT * Cin,
                               [Cin](int t, int c) { return t * Cin + c; });
}

// ---------------------------------------------------------------- weight gradient, Cin == 1
// dw[co][tap] += sum_o gy[o][co] * in_act(x[i(o,tap)])
template <typename T, int KS>
__global__ __launch_bounds__(256) void wgrad_cin1_kernel(SmallConvParams p, Band bd, float* __restrict__ ws) {
    constexpr int VEC = Vec<T>::N, NT = KS * KS;
    extern __shared__ float red[];                   // [Cout * NT] block partial, then the x patch
    const int G = p.Cout / VEC, ppb = 256 / G;
    const int tx = threadIdx.x % G, ty = threadIdx.x / G;
    float* __restrict__ pl = red + NT * p.Cout;
    const T* __restrict__ gy = (const T*)p.gy;
    float acc[NT][VEC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
    for (int item = blockIdx.x; item < bd.items; item += gridDim.x) {
        const BandItem it = band_item(bd, item);
        __syncthreads();
        band_stage<T, KS>(bd, it, (const T*)p.x, p.in_act == S2E_ACT_LRELU, pl);
        __syncthreads();
        const T* __restrict__ gb = gy + (((size_t)it.n * bd.H + it.y0) * bd.W + it.x0) * p.Cout + tx * VEC;
        auto consume = [&](u32x4_t gr, const Walk& w, bool live) __attribute__((always_inline)) {
            float gv[VEC];
            unpack16<T>(gr, gv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) gv[j] = live ? gv[j] : 0.f;
            const float* __restrict__ base = pl + w.r * bd.s * bd.pc + w.c * bd.s;
            float xs[NT];
            static_for<0, NT>([&](auto TT) { constexpr int t = decltype(TT)::value; xs[t] = base[band_tap<KS>(bd, t)]; });
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[t][j] += xs[t] * gv[j];
        };
        Walk wa;
        wa.init(ty, it.nc);
        while (wa.r < it.nr) {
            Walk wv[SMALL_INFLIGHT];
            bool lv[SMALL_INFLIGHT];
            u32x4_t gr[SMALL_INFLIGHT];
#pragma unroll
            for (int u = 0; u < SMALL_INFLIGHT; ++u) {
                lv[u] = wa.r < it.nr;
                wv[u] = lv[u] ? wa : wv[0];
                gr[u] = *(const u32x4_t*)(gb + ((size_t)wv[u].r * bd.W + wv[u].c) * p.Cout);
                if (lv[u]) wa.advance(ppb, it.nc);
            }
#pragma unroll
            for (int u = 0; u < SMALL_INFLIGHT; ++u) consume(gr[u], wv[u], lv[u]);
        }
    }
    __syncthreads();
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

// work items of a band kernel: rows per band so that there are about `target` items (1..8 rows), segments of <= 256 columns
static Band band_plan(int N, int H, int W, int h1, int w1, int s, int org, int flip, int KS, long target) {
    Band b{};
    b.H = H; b.W = W; b.h1 = h1; b.w1 = w1; b.s = s; b.org = org; b.flip = flip;
    b.segw = W < 256 ? W : 256;
    b.segs = ceil_div(W, b.segw);
    long rows = ((long)N * H * b.segs + target - 1) / target;
    rows = rows < 1 ? 1 : (rows > 8 ? 8 : rows);
    b.rows = (int)(rows < H ? rows : H);
    b.bands = ceil_div(H, b.rows);
    b.items = N * b.bands * b.segs;
    b.pc = (b.segw - 1) * s + KS;
    return b;
}
static size_t band_patch_bytes(const Band& b, int KS) { return (size_t)((b.rows - 1) * b.s + KS) * b.pc * sizeof(float); }

template <typename T, int KS>
static int small_fwd(const SmallConvParams& p, int kind, hipStream_t st) {
    if (kind == SMALL_FWD_COUT1) {
        const long M = (long)p.N * p.Ho * p.Wo;
        const int ppb = 256 / (p.Cin / Vec<T>::N);
        long g = (M + ppb - 1) / ppb;
        if (g > 4096) g = 4096;                      // register-resident weights: amortise their load
        fwd_cout1_kernel<T, KS><<<(int)(g < 1 ? 1 : g), 256, 0, st>>>(p);
    } else {
        const bool fwd = kind == SMALL_FWD_CIN1;     // else: data gradient of a Cout == 1 conv, gy (p.x) is the 1-channel tensor
        const long tgt = 1024, gcap = 1024;          // (512 .. 8192 items and 512 .. 8192 workgroups measured within 5 % of each other)
        const Band bd = fwd ? band_plan(p.N, p.Ho, p.Wo, p.Hi, p.Wi, p.stride, -p.pad, 0, KS, tgt)
                            : band_plan(p.N, p.Ho, p.Wo, p.Hi, p.Wi, 1, p.pad - (KS - 1), 1, KS, tgt);
        const int grid = bd.items < gcap ? bd.items : (int)gcap;
        if constexpr (std::is_same<T, bf16_t>::value) {
            if (p.Cout % 32 == 0) {                  // (bf16, whole 32-channel blocks: the matrix-core form)
                const size_t lds2 = ((band_patch_bytes(bd, KS) / 2 + 15) & ~(size_t)15) + (size_t)p.Cout * sizeof(float);
                if (fwd) tap_gemm_kernel<KS, false><<<grid, 256, lds2, st>>>(p, bd);
                else tap_gemm_kernel<KS, true><<<grid, 256, lds2, st>>>(p, bd);
                S2E_CHECK_LAUNCH("small conv kernel (taps as one K-step)");
                return S2E_OK;
            }
        }
        const size_t lds = band_patch_bytes(bd, KS);
        if (fwd) fwd_cin1_kernel<T, KS><<<grid, 256, lds, st>>>(p, bd);
        else dgrad_cout1_kernel<T, KS><<<grid, 256, lds, st>>>(p, bd);
    }
    S2E_CHECK_LAUNCH("small conv kernel");
    return S2E_OK;
}

int s2e_small_conv_kind(int dtype, const s2e_conv_desc* d) {
    const int vec = dtype == S2E_BF16 ? 8 : 4;
    if (!small_ks(d)) return SMALL_NONE;
    if (s2e_c8s2_fwd_ok(dtype, d)) return SMALL_FWD_C8S2;
    if (s2e_c8s2_dgrad_ok(dtype, d)) return SMALL_DGRAD_C8S2;
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
    int t, wk, nkw;
    if (kind == SMALL_FWD_C8S2) return s2e_c8s2_fwd_launch(p, st);
    if (kind == SMALL_DGRAD_C8S2) return s2e_c8s2_dgrad_launch(p, st);
    if (kind == SMALL_FWD_COUT1 && dtype == S2E_BF16 && cout1_mfma_plan(p, &t, &wk, &nkw))
        return p.KH == 3 ? cout1_mfma_go<3>(p, t, wk, nkw, st) : cout1_mfma_go<4>(p, t, wk, nkw, st);
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

// work items and grid of the partial kernel: <= 1024 blocks, partial rows within SMALL_WS_CAP
static Band small_wgrad_band(int kind, const s2e_conv_desc* d, int* grid) {
    const int c = kind == SMALL_WGRAD_COUT1 ? d->Cin : d->Cout;
    long gmax = (long)(SMALL_WS_CAP / ((size_t)d->KH * d->KW * c * sizeof(float)));
    gmax = gmax > 1024 ? 1024 : (gmax < 1 ? 1 : gmax);
    const Band bd = kind == SMALL_WGRAD_COUT1 ? band_plan(d->N, d->Hi, d->Wi, d->Ho, d->Wo, 1, d->pad - (d->KH - 1), 1, d->KH, gmax)
                                              : band_plan(d->N, d->Ho, d->Wo, d->Hi, d->Wi, d->stride, -d->pad, 0, d->KH, gmax);
    *grid = (int)(bd.items < gmax ? bd.items : gmax);
    return bd;
}
static int small_wgrad_grid(int dtype, int kind, const s2e_conv_desc* d) {
    (void)dtype;
    int g;
    small_wgrad_band(kind, d, &g);
    return g;
}

size_t s2e_small_wgrad_workspace_bytes(int dtype, int kind, const s2e_conv_desc* d) {
    const int c = kind == SMALL_WGRAD_COUT1 ? d->Cin : d->Cout;
    return (size_t)small_wgrad_grid(dtype, kind, d) * d->KH * d->KW * c * sizeof(float);
}

template <typename T, int KS>
static void small_wgrad_go(int kind, const SmallConvParams& p, const Band& bd, int grid, size_t lds, float* ws, hipStream_t st) {
    if (kind == SMALL_WGRAD_COUT1) wgrad_cout1_kernel<T, KS><<<grid, 256, lds, st>>>(p, bd, ws);
    else wgrad_cin1_kernel<T, KS><<<grid, 256, lds, st>>>(p, bd, ws);
}

int s2e_small_wgrad_launch(int dtype, int kind, const s2e_conv_desc* d, const SmallConvParams& p, void* workspace,
                           size_t workspace_bytes, hipStream_t st) {
    const size_t need = s2e_small_wgrad_workspace_bytes(dtype, kind, d);
    if (!workspace || workspace_bytes < need)
        S2E_FAIL(S2E_ERR_ARG, "s2e_conv2d_wgrad: this shape needs a %zu-byte workspace (got %zu); see s2e_conv2d_wgrad_workspace_bytes",
                 need, workspace_bytes);
    int grid;
    const Band bd = small_wgrad_band(kind, d, &grid);
    const int c = kind == SMALL_WGRAD_COUT1 ? d->Cin : d->Cout;
    const int nout = d->KH * d->KW * c;
    const size_t lds = (size_t)nout * sizeof(float) + band_patch_bytes(bd, d->KH);
    float* ws = (float*)workspace;
    if (d->KH == 3) { if (dtype == S2E_BF16) small_wgrad_go<bf16_t, 3>(kind, p, bd, grid, lds, ws, st); else small_wgrad_go<float, 3>(kind, p, bd, grid, lds, ws, st); }
    else            { if (dtype == S2E_BF16) small_wgrad_go<bf16_t, 4>(kind, p, bd, grid, lds, ws, st); else small_wgrad_go<float, 4>(kind, p, bd, grid, lds, ws, st); }
    int slabs = grid / 32;
    slabs = slabs < 1 ? 1 : (slabs > 8 ? 8 : slabs);
    if (s2e_deterministic()) slabs = 1;              // (several row slabs are combined with float atomics)
    small_wgrad_reduce_kernel<<<dim3((nout + 63) / 64, slabs), 256, 0, st>>>(ws, grid, nout, p.dw);
    S2E_CHECK_LAUNCH("small wgrad kernels");
    return S2E_OK;
}
