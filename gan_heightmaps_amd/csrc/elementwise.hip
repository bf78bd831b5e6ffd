// HBM-bound wavefront kernels of the train step: BatchNorm (stats / apply / backward), activations,
// pooling, nearest and Theano-bilinear 2x resampling, strided copies, losses and the optimisers.
// All tensors are fp32 [N, C, HW] views with an explicit sample stride; lanes run along HW (NCHW
// contiguous), 16 B per lane where the geometry allows.
//
// Replaces Theano Elemwise / Pool / GpuDnnBatchNorm-free BN graph / lasagne.updates expressions
// (SURVEY.md section 2.2 "Loss / optimizer sites" and "Backward sites").
#include "common.h"

namespace {

struct View {
    int N, C, HW;
};

// index helper: one thread handles VEC consecutive hw elements of one (n, c) row
template <int VEC>
__device__ __forceinline__ bool decode(const View v, long& n, long& c, long& i) {
    const long per_row = v.HW / VEC;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)v.N * v.C * per_row;
    if (idx >= total) return false;
    const long row = idx / per_row;
    i = (idx - row * per_row) * VEC;
    n = row / v.C;
    c = row - n * v.C;
    return true;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

inline int grid_for(const View v, int vec) { return ceil_div((long)v.N * v.C * (v.HW / vec), 256); }

// ---------------------------------------------------------------------------------------------
// BatchNorm
// ---------------------------------------------------------------------------------------------
constexpr int BN_MAX_SPLIT = 64;

__device__ __forceinline__ double block_sum(double v, double* red) {
    // wave reduce then LDS across the 4 waves
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// grid (S, C): partial sum / sum of squares in fp64
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, int N, int HW, long nstride,
                                                        int S, double* __restrict__ ws) {
    const int c = blockIdx.y, s = blockIdx.x;
    const long total = (long)N * HW;
    const long chunk = (total + S - 1) / S;
    const long lo = s * chunk, hi = min(lo + chunk, total);
    double a = 0.0, b = 0.0;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const long n = e / HW, i = e - n * HW;
        const float v = x[n * nstride + (long)c * HW + i];
        a += v;
        b += (double)v * v;
    }
    __shared__ double red[4];
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        ws[((long)c * BN_MAX_SPLIT + s) * 2 + 0] = a;
        ws[((long)c * BN_MAX_SPLIT + s) * 2 + 1] = b;
    }
}

// The backward kernels need act'(y) of the layer output y = act((x - mean) * gamma * inv + beta).  With y == nullptr they
// RECOMPUTE it from x with the forward pass's own expression (bit-identical: same fmaf, same activation code), so the
// backward pass does not read the output tensor at all -- and the forward pass need not keep it for the backward.
__device__ __forceinline__ float bn_y(const float* y, long off, float xv, float m, float sc, float be, int act, float alpha) {
    return y ? y[off] : ghm_act(fmaf(xv - m, sc, be), act, alpha);
}
__device__ __forceinline__ float4 bn_y4(const float* y, long off, const float4 xx, float m, float sc, float be, int act, float alpha) {
    if (y) return *reinterpret_cast<const float4*>(y + off);
    return make_float4(ghm_act(fmaf(xx.x - m, sc, be), act, alpha), ghm_act(fmaf(xx.y - m, sc, be), act, alpha),
                       ghm_act(fmaf(xx.z - m, sc, be), act, alpha), ghm_act(fmaf(xx.w - m, sc, be), act, alpha));
}

// Row-structured partials, grid (S = N * segs, C): block (n, seg) of channel c sums one contiguous run of the
// (n, c) row with 16-byte loads and no per-element index arithmetic; sums and products in fp64.  BWD: sums of dz and dz * xhat (dz = dout * act'(y)); otherwise sums of x and x^2.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_rows_partial(const float* __restrict__ x, long xs, const float* __restrict__ dout,
                                                       long ds, const float* __restrict__ y, long ys, int HW, int segs,
                                                       int seg_len, const float* __restrict__ mean,
                                                       const float* __restrict__ inv, int act, float alpha,
                                                       double* __restrict__ ws, const float* __restrict__ gamma = nullptr,
                                                       const float* __restrict__ beta = nullptr) {
    const int c = blockIdx.y, s = blockIdx.x;
    const int n = s / segs, seg = s - n * segs;
    const int lo = seg * seg_len, hi = min(lo + seg_len, HW);
    const long row = (long)c * HW;
    const float* xp = x + n * xs + row;
    double a = 0.0, b = 0.0;
    if constexpr (BWD) {
        const float* dp = dout + n * ds + row;
        const float* yp = y ? y + n * ys + row : nullptr;
        const float m = mean[c], iv = inv[c];
        const float sc = y ? 0.f : gamma[c] * iv, be = y ? 0.f : beta[c];
#pragma unroll 2
        for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
            const float4 d = *reinterpret_cast<const float4*>(dp + i);
            const float4 xx = *reinterpret_cast<const float4*>(xp + i);
            const float4 yy = bn_y4(yp, i, xx, m, sc, be, act, alpha);
            const float z0 = d.x * ghm_dact_from_out(yy.x, act, alpha), z1 = d.y * ghm_dact_from_out(yy.y, act, alpha);
            const float z2 = d.z * ghm_dact_from_out(yy.z, act, alpha), z3 = d.w * ghm_dact_from_out(yy.w, act, alpha);
            a += ((double)z0 + (double)z1) + ((double)z2 + (double)z3);
            b += ((double)z0 * ((xx.x - m) * iv) + (double)z1 * ((xx.y - m) * iv)) +
                 ((double)z2 * ((xx.z - m) * iv) + (double)z3 * ((xx.w - m) * iv));
        }
    } else {
#pragma unroll 4
        for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xp + i);
            // every product and sum in fp64: channels whose mean is large against their spread (DCGAN generator
            // at initialisation) lose E[x^2] - mean^2 to fp32 rounding otherwise
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            a += (x0 + x1) + (x2 + x3);
            b += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
        }
    }
    __shared__ double red[4];
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        ws[((long)c * BN_MAX_SPLIT + s) * 2 + 0] = a;
        ws[((long)c * BN_MAX_SPLIT + s) * 2 + 1] = b;
    }
}

__global__ void bn_stats_final(const double* __restrict__ ws, int C, int S, double count, float eps,
                               float* __restrict__ mean, float* __restrict__ inv, float* run_mean, float* run_inv,
                               float ra) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < S; ++s) {
        a += ws[((long)c * BN_MAX_SPLIT + s) * 2 + 0];
        b += ws[((long)c * BN_MAX_SPLIT + s) * 2 + 1];
    }
    const double mu = a / count;
    double var = b / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float m = (float)mu;
    const float iv = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = m;
    inv[c] = iv;
    if (run_mean) {
        run_mean[c] = (1.f - ra) * run_mean[c] + ra * m;
        run_inv[c] = (1.f - ra) * run_inv[c] + ra * iv;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long xs, float* __restrict__ y,
                                                       long ys, View v, const float* __restrict__ mean,
                                                       const float* __restrict__ inv, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int act, float alpha) {
    long n, c, i;
    if (!decode<VEC>(v, n, c, i)) return;
    // (x - mean) * (gamma * inv) + beta, in this order: when the batch variance is ~0 (batch of 1, 1x1 maps) the
    // subtraction cancels exactly like the reference expression does; folding mean into a pre-computed shift
    // leaves a rounding residue that the next layers' 1/sqrt(eps) = 100 gains amplify
    const float sc = gamma[c] * inv[c];
    const float m = mean[c], be = beta[c];
    const float* xp = x + n * xs + c * v.HW + i;
    float* yp = y + n * ys + c * v.HW + i;
    if constexpr (VEC == 4) {
        float4 t = *reinterpret_cast<const float4*>(xp);
        t.x = ghm_act(fmaf(t.x - m, sc, be), act, alpha);
        t.y = ghm_act(fmaf(t.y - m, sc, be), act, alpha);
        t.z = ghm_act(fmaf(t.z - m, sc, be), act, alpha);
        t.w = ghm_act(fmaf(t.w - m, sc, be), act, alpha);
        *reinterpret_cast<float4*>(yp) = t;
    } else {
        *yp = ghm_act(fmaf(*xp - m, sc, be), act, alpha);
    }
}

// grid (S, C): partial sums of dz and dz*xhat, dz = dout * act'(y)
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ dout, long ds, const float* __restrict__ y,
                                                      long ys, const float* __restrict__ x, long xs, int N, int HW, int S,
                                                      const float* __restrict__ mean, const float* __restrict__ inv,
                                                      int act, float alpha, double* __restrict__ ws,
                                                      const float* __restrict__ gamma = nullptr,
                                                      const float* __restrict__ beta = nullptr) {
    const int c = blockIdx.y, s = blockIdx.x;
    const long total = (long)N * HW;
    const long chunk = (total + S - 1) / S;
    const long lo = s * chunk, hi = min(lo + chunk, total);
    const float m = mean[c], iv = inv[c];
    const float sc = y ? 0.f : gamma[c] * iv, be = y ? 0.f : beta[c];
    double a = 0.0, b = 0.0;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const long n = e / HW, i = e - n * HW;
        const long o = (long)c * HW + i;
        const float xv = x[n * xs + o];
        const float dz = dout[n * ds + o] * ghm_dact_from_out(bn_y(y, n * ys + o, xv, m, sc, be, act, alpha), act, alpha);
        const float xh = (xv - m) * iv;
        a += dz;
        b += (double)dz * xh;
    }
    __shared__ double red[4];
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        ws[((long)c * BN_MAX_SPLIT + s) * 2 + 0] = a;
        ws[((long)c * BN_MAX_SPLIT + s) * 2 + 1] = b;
    }
}

__global__ void bn_bwd_final(const double* __restrict__ ws, int C, int S, float* __restrict__ sums, float* dgamma,
                             float* dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < S; ++s) {
        a += ws[((long)c * BN_MAX_SPLIT + s) * 2 + 0];
        b += ws[((long)c * BN_MAX_SPLIT + s) * 2 + 1];
    }
    sums[2 * c + 0] = (float)a;
    sums[2 * c + 1] = (float)b;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)b;
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dout, long ds, const float* __restrict__ y,
                                                    long ys, const float* __restrict__ x, long xs, float* __restrict__ dx,
                                                    long dxs, View v, const float* __restrict__ mean,
                                                    const float* __restrict__ inv, const float* __restrict__ gamma,
                                                    const float* __restrict__ sums, float inv_count, int act,
                                                    float alpha, const float* __restrict__ beta = nullptr) {
    long n, c, i;
    if (!decode<VEC>(v, n, c, i)) return;
    const float m = mean[c], iv = inv[c], g = gamma[c] * iv;
    const float be = y ? 0.f : beta[c];
    const float mb = sums[2 * c] * inv_count, mg = sums[2 * c + 1] * inv_count;
    const long o = c * v.HW + i;
    if constexpr (VEC == 4) {
        const float4 d = *reinterpret_cast<const float4*>(dout + n * ds + o);
        const float4 xx = *reinterpret_cast<const float4*>(x + n * xs + o);
        const float4 yy = bn_y4(y, n * ys + o, xx, m, g, be, act, alpha);
        float4 r;
        r.x = g * (d.x * ghm_dact_from_out(yy.x, act, alpha) - mb - (xx.x - m) * iv * mg);
        r.y = g * (d.y * ghm_dact_from_out(yy.y, act, alpha) - mb - (xx.y - m) * iv * mg);
        r.z = g * (d.z * ghm_dact_from_out(yy.z, act, alpha) - mb - (xx.z - m) * iv * mg);
        r.w = g * (d.w * ghm_dact_from_out(yy.w, act, alpha) - mb - (xx.w - m) * iv * mg);
        *reinterpret_cast<float4*>(dx + n * dxs + o) = r;
    } else {
        const float xv = x[n * xs + o];
        const float dz = dout[n * ds + o] * ghm_dact_from_out(bn_y(y, n * ys + o, xv, m, g, be, act, alpha), act, alpha);
        const float xh = (xv - m) * iv;
        dx[n * dxs + o] = g * (dz - mb - xh * mg);
    }
}

// ---------------------------------------------------------------------------------------------
// activations, copies
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long xs, float* __restrict__ y, long ys,
                                                      View v, int act, float alpha) {
    long n, c, i;
    if (!decode<VEC>(v, n, c, i)) return;
    const float* xp = x + n * xs + c * v.HW + i;
    float* yp = y + n * ys + c * v.HW + i;
    if constexpr (VEC == 4) {
        float4 t = *reinterpret_cast<const float4*>(xp);
        t.x = ghm_act(t.x, act, alpha);
        t.y = ghm_act(t.y, act, alpha);
        t.z = ghm_act(t.z, act, alpha);
        t.w = ghm_act(t.w, act, alpha);
        *reinterpret_cast<float4*>(yp) = t;
    } else {
        *yp = ghm_act(*xp, act, alpha);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, long ds, const float* __restrict__ y,
                                                      long ys, float* __restrict__ dx, long dxs, View v, int act,
                                                      float alpha, int accumulate) {
    long n, c, i;
    if (!decode<VEC>(v, n, c, i)) return;
    const long o = c * v.HW + i;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float g = dout[n * ds + o + k] * ghm_dact_from_out(y[n * ys + o + k], act, alpha);
        float* p = dx + n * dxs + o + k;
        if (accumulate) g += *p;
        *p = g;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void copy_view_kernel(const float* __restrict__ x, long xs, float* __restrict__ y,
                                                        long ys, View v, int accumulate) {
    long n, c, i;
    if (!decode<VEC>(v, n, c, i)) return;
    const long o = c * v.HW + i;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float g = x[n * xs + o + k];
        float* p = y + n * ys + o + k;
        if (accumulate) g += *p;
        *p = g;
    }
}

// x[n, ...] *= num[n] / den[n]: the per-sample factor between two backward passes through a net whose output is ONE scalar per
// sample -- the pass is linear in that scalar's gradient (step.py, "rank-one" generator gradient).  den[n] == 0: the element
// stays 0 where it is exactly 0 (a dead final ReLU: both passes are exactly zero there) and becomes NaN where it is not --
// a head whose output is exactly 0 WITHOUT a dead ReLU in front has a non-zero generator-loss gradient that no factor recovers
// from a zero discriminator-loss seed; that must fail loudly (finite checks of every parity test), not train on zeros
template <int VEC>
__global__ __launch_bounds__(256) void scale_samples_kernel(float* __restrict__ x, long xs, View v, const float* __restrict__ num,
                                                            long nums, const float* __restrict__ den, long dens) {
    long n, c, i;
    if (!decode<VEC>(v, n, c, i)) return;
    const float d = den[n * dens];
    // the ratio and the product in fp64: a denominator near the bottom of the fp32 range (a discriminator output of 1e-40) would
    // push the fp32 ratio to inf and inf * 0 to NaN; the gradient it multiplies is that small too, and the product is ordinary
    const double r = d != 0.f ? (double)num[n * nums] / (double)d : 0.0;
    float* p = x + n * xs + c * v.HW + i;
#pragma unroll
    for (int k = 0; k < VEC; ++k) p[k] = (d != 0.f || p[k] == 0.f) ? (float)((double)p[k] * r) : __builtin_nanf("");
}

__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// pooling / resampling: one thread per OUTPUT-side 2x2 cell (or per coarse pixel)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes,
                                                           int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo) return;
    const long pl = idx / (Ho * Wo);
    const int rem = (int)(idx - pl * Ho * Wo), i = rem / Wo, j = rem - i * Wo;
    const float* p = x + pl * H * W + (long)(2 * i) * W + 2 * j;
    const float2 a = *reinterpret_cast<const float2*>(p);
    const float2 b = *reinterpret_cast<const float2*>(p + W);
    y[idx] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, float* __restrict__ dx,
                                                           long planes, int H, int W, int act, float alpha) {
    const int Ho = H / 2, Wo = W / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo) return;
    const long pl = idx / (Ho * Wo);
    const int rem = (int)(idx - pl * Ho * Wo), i = rem / Wo, j = rem - i * Wo;
    const long o = pl * H * W + (long)(2 * i) * W + 2 * j;
    const float m = y[idx], g = dy[idx];
    const float2 a = *reinterpret_cast<const float2*>(x + o);
    const float2 b = *reinterpret_cast<const float2*>(x + o + W);
    float2 ra, rb;
    ra.x = (a.x == m) ? g * ghm_dact_from_out(a.x, act, alpha) : 0.f;
    ra.y = (a.y == m) ? g * ghm_dact_from_out(a.y, act, alpha) : 0.f;
    rb.x = (b.x == m) ? g * ghm_dact_from_out(b.x, act, alpha) : 0.f;
    rb.y = (b.y == m) ? g * ghm_dact_from_out(b.y, act, alpha) : 0.f;
    *reinterpret_cast<float2*>(dx + o) = ra;
    *reinterpret_cast<float2*>(dx + o + W) = rb;
}

// Backward of the FUSED conv + activation + 2x2 max-pool (ghm_conv2d_fwd_pool): the forward kept the pooled value y
// and a 4-bit arg-max mask per pooled pixel (bit 2*dr + dc: position (dr, dc) of the window equals the maximum; all
// ties set), so the full-resolution activation never existed.  dx[2i+dr, 2j+dc] = bit ? dy[i,j] * act'(y[i,j]) : 0
// (the activation is monotonic: the derivative at the arg-max is the derivative at the pooled value).
// One thread per TWO pooled pixels: two 16-byte stores.
// BIAS: the block also reduces what it wrote (a block lies inside one (n, c) plane: Ho * W/4 is a multiple of 256) into
// part[c][n * blocks_per_plane + block in plane] -- the conv's bias gradient without re-reading the gradient tensor
// out[c] (+)= sum_k part[c][k], fixed order; one wave per channel
__global__ __launch_bounds__(64) void partial_rows_sum_kernel(const float* __restrict__ part, int C, int S,
                                                              float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < S; k += 64) s += part[(long)c * S + k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + s;
}

template <bool BIAS>
__global__ __launch_bounds__(256) void maxpool2_mask_bwd_kernel(const unsigned char* __restrict__ mask,
                                                                const float* __restrict__ y, const float* __restrict__ dy,
                                                                float* __restrict__ dx, long planes, int H, int W, int act,
                                                                float alpha, float* __restrict__ part, int C, int bpp) {
    const int Ho = H / 2, Wo2 = W / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo2) return;
    const long pl = idx / (Ho * Wo2);
    const int rem = (int)(idx - pl * Ho * Wo2), i = rem / Wo2, j2 = rem - i * Wo2;
    const long po = pl * Ho * (W / 2) + (long)i * (W / 2) + 2 * j2;
    const unsigned m2 = *reinterpret_cast<const unsigned short*>(mask + po);
    const float2 gv = *reinterpret_cast<const float2*>(dy + po);
    const unsigned m0 = m2 & 0xffu, m1 = m2 >> 8;
    float g0, g1;               // y == nullptr: the slope from the mask's sign bit (relu / leaky relu / linear)
    if (y) {
        const float2 yv = *reinterpret_cast<const float2*>(y + po);
        g0 = gv.x * ghm_dact_from_out(yv.x, act, alpha);
        g1 = gv.y * ghm_dact_from_out(yv.y, act, alpha);
    } else {
        g0 = gv.x * ghm_dact_from_sign(m0, act, alpha);
        g1 = gv.y * ghm_dact_from_sign(m1, act, alpha);
    }
    const long o = pl * H * W + (long)(2 * i) * W + 4 * j2;
    *reinterpret_cast<float4*>(dx + o) = make_float4((m0 & 1u) ? g0 : 0.f, (m0 & 2u) ? g0 : 0.f, (m1 & 1u) ? g1 : 0.f, (m1 & 2u) ? g1 : 0.f);
    *reinterpret_cast<float4*>(dx + o + W) = make_float4((m0 & 4u) ? g0 : 0.f, (m0 & 8u) ? g0 : 0.f, (m1 & 4u) ? g1 : 0.f, (m1 & 8u) ? g1 : 0.f);
    if constexpr (BIAS) {
        float sum = g0 * (float)__popc(m0 & 15u) + g1 * (float)__popc(m1 & 15u);
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n = (int)(pl / C), c = (int)(pl - (long)n * C);
            const int blk = (int)(blockIdx.x - pl * bpp);
            part[(long)c * ((planes / C) * bpp) + (long)n * bpp + blk] = (red[0] + red[1]) + (red[2] + red[3]);
        }
    }
}

__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes,
                                                          int H, int W, int p) {
    const int Ho = H / p, Wo = W / p;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo) return;
    const long pl = idx / (Ho * Wo);
    const int rem = (int)(idx - pl * Ho * Wo), i = rem / Wo, j = rem - i * Wo;
    float s = 0.f;
    for (int a = 0; a < p; ++a)
        for (int b = 0; b < p; ++b) s += x[pl * H * W + (long)(i * p + a) * W + j * p + b];
    y[idx] = s / (float)(p * p);
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long planes,
                                                          int H, int W, int p) {
    const int Ho = H / p, Wo = W / p;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H * W) return;
    const long pl = idx / (H * W);
    const int rem = (int)(idx - pl * H * W), u = rem / W, v = rem - u * W;
    const int i = u / p, j = v / p;
    dx[idx] = (i < Ho && j < Wo) ? dy[pl * Ho * Wo + i * Wo + j] / (float)(p * p) : 0.f;
}

// nearest 2x: thread per input pixel writes a 2x2 block
__global__ __launch_bounds__(256) void up_nearest_fwd_kernel(const float* __restrict__ x, long xs, float* __restrict__ y,
                                                             int N, int C, int H, int W) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= (long)N * C * hw) return;
    const long pl = idx / hw;
    const int rem = (int)(idx - pl * hw), i = rem / W, j = rem - i * W;
    const long n = pl / C, c = pl - n * C;
    const float v = x[n * xs + c * hw + rem];
    float* o = y + pl * 4 * hw + (long)(2 * i) * (2 * W) + 2 * j;
    const float2 vv = make_float2(v, v);
    *reinterpret_cast<float2*>(o) = vv;
    *reinterpret_cast<float2*>(o + 2 * W) = vv;
}

__global__ __launch_bounds__(256) void up_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long dxs,
                                                             int N, int C, int H, int W, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= (long)N * C * hw) return;
    const long pl = idx / hw;
    const int rem = (int)(idx - pl * hw), i = rem / W, j = rem - i * W;
    const long n = pl / C, c = pl - n * C;
    const float* g = dy + pl * 4 * hw + (long)(2 * i) * (2 * W) + 2 * j;
    const float2 a = *reinterpret_cast<const float2*>(g);
    const float2 b = *reinterpret_cast<const float2*>(g + 2 * W);
    float s = (a.x + a.y) + (b.x + b.y);
    float* p = dx + n * dxs + c * hw + rem;
    if (accumulate) s += *p;
    *p = s;
}

// Theano bilinear 2x (ratio 2): out[2m] = x[m]; out[2m+1] = (x[m] + x[min(m+1,n-1)])/2, separable.
__global__ __launch_bounds__(256) void up_bilinear_fwd_kernel(const float* __restrict__ x, long xs, float* __restrict__ y,
                                                              int N, int C, int H, int W) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= (long)N * C * hw) return;
    const long pl = idx / hw;
    const int rem = (int)(idx - pl * hw), i = rem / W, j = rem - i * W;
    const long n = pl / C, c = pl - n * C;
    const float* xp = x + n * xs + c * hw;
    const int i1 = min(i + 1, H - 1), j1 = min(j + 1, W - 1);
    const float v00 = xp[i * W + j], v01 = xp[i * W + j1], v10 = xp[i1 * W + j], v11 = xp[i1 * W + j1];
    float* o = y + pl * 4 * hw + (long)(2 * i) * (2 * W) + 2 * j;
    const float r0 = 0.5f * (v00 + v01);
    const float c0 = 0.5f * (v00 + v10), c1 = 0.5f * (v01 + v11);
    *reinterpret_cast<float2*>(o) = make_float2(v00, r0);
    *reinterpret_cast<float2*>(o + 2 * W) = make_float2(c0, 0.5f * (c0 + c1));
}

// adjoint: dx[m] = g[2m] + g[2m+1]/2 + (m>=1 ? g[2m-1]/2 : 0) + (m==n-1 ? g[2n-1]/2 : 0), per axis
__device__ __forceinline__ float bil_row(const float* g, int W2, int j, int W) {
    // horizontal adjoint at fine row pointer g (length W2 = 2W), coarse column j
    float s = g[2 * j] + 0.5f * g[2 * j + 1];
    if (j >= 1) s += 0.5f * g[2 * j - 1];
    if (j == W - 1) s += 0.5f * g[2 * W - 1];
    return s;
}
__global__ __launch_bounds__(256) void up_bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long dxs,
                                                              int N, int C, int H, int W, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= (long)N * C * hw) return;
    const long pl = idx / hw;
    const int rem = (int)(idx - pl * hw), i = rem / W, j = rem - i * W;
    const long n = pl / C, c = pl - n * C;
    const int W2 = 2 * W;
    const float* g = dy + pl * 4 * hw;
    float s = bil_row(g + (long)(2 * i) * W2, W2, j, W) + 0.5f * bil_row(g + (long)(2 * i + 1) * W2, W2, j, W);
    if (i >= 1) s += 0.5f * bil_row(g + (long)(2 * i - 1) * W2, W2, j, W);
    if (i == H - 1) s += 0.5f * bil_row(g + (long)(2 * H - 1) * W2, W2, j, W);
    float* p = dx + n * dxs + c * hw + rem;
    if (accumulate) s += *p;
    *p = s;
}


// Same adjoint, two coarse columns per thread and one plane per blockIdx.y: the five fine columns a pair touches are
// one scalar + one 16-byte load per fine row, and no 64-bit index arithmetic.  Needs W even and 16-byte aligned rows.
__global__ __launch_bounds__(256) void up_bilinear_bwd2_kernel(const float* __restrict__ dy, float* __restrict__ dx, long dxs,
                                                               int C, int H, int W, int accumulate) {
    const int pl = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int Wh = W >> 1;
    if (t >= H * Wh) return;
    const int i = t / Wh, j0 = (t - i * Wh) * 2;
    const int n = pl / C, c = pl - n * C;
    const int W2 = 2 * W;
    const float* g = dy + (long)pl * 4 * H * W + 2 * j0;
    const float wl = j0 >= 1 ? 0.5f : 0.f;                 // fine column 2*j0 - 1 exists
    const float wr = (j0 + 1 == W - 1) ? 1.f : 0.5f;       // the clamped last column takes its right neighbour twice
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = -1; r <= 1; ++r) {
        const int fr = 2 * i + r;
        if (fr < 0) continue;
        const float wrow = r == 0 ? 1.f : ((r == 1 && i == H - 1) ? 1.f : 0.5f);
        const float* row = g + (long)fr * W2;
        const float4 f = *reinterpret_cast<const float4*>(row);
        const float fm = j0 >= 1 ? row[-1] : 0.f;
        s0 += wrow * (f.x + 0.5f * f.y + wl * fm);
        s1 += wrow * (f.z + wr * f.w + 0.5f * f.y);
    }
    float* p = dx + n * dxs + (long)c * H * W + (long)i * W + j0;
    if (accumulate) {
        const float2 o = *reinterpret_cast<const float2*>(p);
        s0 += o.x;
        s1 += o.y;
    }
    *reinterpret_cast<float2*>(p) = make_float2(s0, s1);
}

// ---------------------------------------------------------------------------------------------
// input pipeline (/root/reference/util.py:28-40 + Keras ImageDataGenerator): one pass that turns a
// uint8 NHWC batch into the normalised fp32 NCHW tensor the nets consume, resampled through a per-sample affine
// map (rotation about the centre, then column / row flips) with nearest-neighbour lookup and 'reflect' borders
// -- scipy.ndimage.affine_transform(order=0, mode='reflect') semantics, coordinates in fp64 with separate
// roundings (no fma) so that the sampled indices agree with scipy.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double reflect_coord(double in, int len) {
    if (len <= 1) return 0.0;
    const long sz2 = 2L * len;
    if (in < 0) {
        if (in < -(double)sz2) in = (double)sz2 * (double)(long)(-in / (double)sz2) + in;
        in = in < -(double)len ? in + (double)sz2 : (in > -1e-15 ? 1e-15 : -in) - 1.0;
    } else if (in > (double)(len - 1)) {
        in -= (double)sz2 * (double)(long)(in / (double)sz2);
        if (in >= (double)len) in = (double)sz2 - in - 1.0;
    }
    return in;
}

// xf[n] = {m00, m01, off0, m10, m11, off1, hflip, vflip}
__global__ __launch_bounds__(256) void image_batch_kernel(const unsigned char* __restrict__ src, int N, int H, int W, int C,
                                                          const double* __restrict__ xf, int mode, float* __restrict__ dst,
                                                          long dst_nstride) {
    const long hw = (long)H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * hw) return;
    const int n = (int)(idx / hw), rem = (int)(idx - (long)n * hw);
    const int i = rem / W, j = rem - i * W;
    const double* m = xf + 8 * n;
    const int ii = m[7] != 0.0 ? H - 1 - i : i;           // flips act on the transformed image
    const int jj = m[6] != 0.0 ? W - 1 - j : j;
    double cy = m[2], cx = m[5];
    cy = __dadd_rn(cy, __dmul_rn((double)ii, m[0]));
    cy = __dadd_rn(cy, __dmul_rn((double)jj, m[1]));
    cx = __dadd_rn(cx, __dmul_rn((double)ii, m[3]));
    cx = __dadd_rn(cx, __dmul_rn((double)jj, m[4]));
    cy = reflect_coord(cy, H);
    cx = reflect_coord(cx, W);
    int sy = (int)floor(cy + 0.5), sx = (int)floor(cx + 0.5);
    sy = min(max(sy, 0), H - 1);
    sx = min(max(sx, 0), W - 1);
    const unsigned char* sp = src + (((long)n * H + sy) * W + sx) * C;
    float* dp = dst + (long)n * dst_nstride + rem;
    for (int c = 0; c < C; ++c) {
        const float v = (float)sp[c];
        dp[(long)c * hw] = mode == 0 ? v / 255.0f : (v - 127.5f) / 127.5f;
    }
}

// ---------------------------------------------------------------------------------------------
// losses: single-pass grid-stride with fp64 block partials + one atomic per block
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scalar_loss_kernel(const float* __restrict__ d, long n, float target, int kind,
                                                          float* __restrict__ grad, float gscale, double* __restrict__ part,
                                                          const float* __restrict__ ls, float* __restrict__ loss_out,
                                                          int accumulate) {
    double acc = 0.0;
    if (ls) gscale *= ls[0];            // dynamic loss scale (fp16 products): a device scalar, so replays see it change
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = d[i];
        if (kind == 0) {
            const float e = v - target;
            acc += (double)e * e;
            if (grad) grad[i] = gscale * 2.f * e / (float)n;
        } else {
            acc += -(double)(target * logf(v) + (1.f - target) * logf(1.f - v));
            if (grad) grad[i] = gscale * (-(target / v) + (1.f - target) / (1.f - v)) / (float)n;
        }
    }
    __shared__ double red[4];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        // a single block (the discriminators' outputs: 8 .. 2048 values) finishes the loss itself: the same value
        // loss_final_kernel would produce from one partial, one launch less on the stage stream's turn-around
        if (gridDim.x == 1) loss_out[0] = (accumulate ? loss_out[0] : 0.f) + (float)(acc / (double)n);
        else part[blockIdx.x] = acc / (double)n;
    }
}

// fixed-order sum of the block partials: losses are bit-for-bit repeatable (no float atomics)
// (one wave: lane l sums partials l, l + 64, ... in order, then a butterfly -- a single thread walked up to 1024 dependent
// loads, half of the 85 us the reconstruction loss took at 512^2)
__global__ __launch_bounds__(64) void loss_final_kernel(const double* __restrict__ part, int nblocks, float* loss_out, int accumulate) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += part[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) loss_out[0] = (accumulate ? loss_out[0] : 0.f) + (float)s;
}

template <int VEC>
__global__ __launch_bounds__(256) void recon_loss_kernel(const float* __restrict__ a, long as, const float* __restrict__ b,
                                                         long bs, View v, int l2, float* __restrict__ grad, long gs,
                                                         float gscale, int accumulate, double* __restrict__ part,
                                                         const float* __restrict__ ls) {
    const long chw = (long)v.C * v.HW, total = (long)v.N * chw;
    double acc = 0.0;
    if (ls) gscale *= ls[0];
    // VEC consecutive elements of one sample per thread and iteration (16-byte loads / stores where the views allow it)
    const long per = chw / VEC, groups = (long)v.N * per;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < groups; e += (long)gridDim.x * 256) {
        const long n = e / per, o = (e - n * per) * VEC;
        float av[VEC], bv[VEC], pv[VEC];
        if (VEC == 4) {
            *reinterpret_cast<float4*>(av) = *reinterpret_cast<const float4*>(a + n * as + o);
            *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(b + n * bs + o);
            if (grad && accumulate) *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(grad + n * gs + o);
        } else {
            av[0] = a[n * as + o];
            bv[0] = b[n * bs + o];
            if (grad && accumulate) pv[0] = grad[n * gs + o];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float dlt = av[k] - bv[k];
            float g;
            if (l2) {
                acc += (double)dlt * dlt;
                g = 2.f * dlt;
            } else {
                acc += fabsf(dlt);
                g = (dlt > 0.f) ? 1.f : (dlt < 0.f ? -1.f : 0.f);
            }
            const float w = gscale * g / (float)total;
            pv[k] = (grad && accumulate) ? pv[k] + w : w;
        }
        if (grad) {
            if (VEC == 4) *reinterpret_cast<float4*>(grad + n * gs + o) = *reinterpret_cast<const float4*>(pv);
            else grad[n * gs + o] = pv[0];
        }
    }
    __shared__ double red[4];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc / (double)total;
}

// ---------------------------------------------------------------------------------------------
// optimisers (lasagne.updates.rmsprop / adam)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ acc, long n, const float* __restrict__ hyper,
                                                      float rho, float eps, float gscale, const float* __restrict__ ls) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (ls) {                           // {scale, 1/scale, good steps, overflow flag}: skip the update of an overflowed step
        if (ls[3] != 0.f) return;
        gscale *= ls[1];
    }
    const float lr = hyper[0];
    if (i + 3 < n) {
        float4 pv = *reinterpret_cast<float4*>(p + i);
        float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 av = *reinterpret_cast<float4*>(acc + i);
        gv.x *= gscale; gv.y *= gscale; gv.z *= gscale; gv.w *= gscale;
        av.x = rho * av.x + (1.f - rho) * gv.x * gv.x;
        av.y = rho * av.y + (1.f - rho) * gv.y * gv.y;
        av.z = rho * av.z + (1.f - rho) * gv.z * gv.z;
        av.w = rho * av.w + (1.f - rho) * gv.w * gv.w;
        pv.x -= lr * gv.x / sqrtf(av.x + eps);
        pv.y -= lr * gv.y / sqrtf(av.y + eps);
        pv.z -= lr * gv.z / sqrtf(av.z + eps);
        pv.w -= lr * gv.w / sqrtf(av.w + eps);
        *reinterpret_cast<float4*>(p + i) = pv;
        *reinterpret_cast<float4*>(acc + i) = av;
    } else {
        for (long k = i; k < n; ++k) {
            const float gg = g[k] * gscale;
            const float a2 = rho * acc[k] + (1.f - rho) * gg * gg;
            acc[k] = a2;
            p[k] -= lr * gg / sqrtf(a2 + eps);
        }
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ hyper, float b1,
                                                   float b2, float eps, float gscale, const float* __restrict__ ls) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (ls) {
        if (ls[3] != 0.f) return;
        gscale *= ls[1];
    }
    const float lr = hyper[0], t = hyper[1] + 1.f;     // t_prev + 1
    const float a_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    const float gg = g[i] * gscale;
    const float mn = b1 * m[i] + (1.f - b1) * gg;
    const float vn = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mn;
    v[i] = vn;
    p[i] -= a_t * mn / (sqrtf(vn) + eps);
}

__global__ void adam_tick_kernel(float* hyper, const float* ls) {
    if (ls && ls[3] != 0.f) return;
    hyper[1] += 1.f;
}

// any non-finite value in g[0, n) raises the overflow flag of the loss-scale state (ls[3]); benign race: every writer
// stores the same value
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, long n, float* __restrict__ ls) {
    bool bad = false;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        // exponent all ones <=> inf or nan
        bad |= ((__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u) |
               ((__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= (__float_as_uint(g[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
    if (__any(bad) && (threadIdx.x & 63) == 0) ls[3] = 1.f;
}

// end of a step: overflow -> halve the scale (not below ``lo``), restart the count; otherwise count the step and double
// the scale after ``interval`` clean steps (not above ``hi``); the flag is cleared for the next step
__global__ void loss_scale_update_kernel(float* ls, float interval, float lo, float hi) {
    float s = ls[0], good = ls[2];
    if (ls[3] != 0.f) {
        s = fmaxf(s * 0.5f, lo);
        good = 0.f;
        ls[4] += 1.f;                    // skipped steps so far (reporting)
    } else {
        good += 1.f;
        if (good >= interval) {
            s = fminf(s * 2.f, hi);
            good = 0.f;
        }
    }
    ls[0] = s;
    ls[1] = 1.f / s;
    ls[2] = good;
    ls[3] = 0.f;
}

template <typename... Args>
inline bool vec_ok(int HW, Args... strides_or_ptr_ok) {
    return (HW % 4 == 0) && (... && strides_or_ptr_ok);
}

}  // namespace


// ---- Upscale2DLayer(2, 'repeat') followed by a 5x5 'same' convolution (every stage of dcgan.default_generator,
// architectures/dcgan.py:22-31) == four 3x3 convolutions of the LOW-resolution input, one per output parity
// (p, q), whose weights are sums of the 5x5 taps that land on the same low-resolution pixel:
//   correlation tap a (offset a-2) of output row 2i+p reads low-res row i + floor((p+a-2)/2)
//     p=0: a in {0,1} -> -1, {2,3} -> 0, {4} -> +1        p=1: {0} -> -1, {1,2} -> 0, {3,4} -> +1
// 9 instead of 25 MACs per output value, and the 4x up-sampled tensor is never materialised.  The collapsed
// weights are a packed 3x3 conv with 4K filters ordered (pq, k): its output [N, 4K, H, W] is, as memory, the
// "parity-planar" tensor [4N, K, H, W] (sample 4n+pq = parity plane pq of image n) on which BatchNorm and the
// activation run unchanged; pp_to_hi interleaves the planes into [N, K, 2H, 2W].
__device__ __forceinline__ int upconv_group(int p, int a) {       // low-res offset + 1 of tap a for parity p
    return p == 0 ? (a < 2 ? 0 : (a < 4 ? 1 : 2)) : (a < 1 ? 0 : (a < 3 ? 1 : 2));
}

__global__ __launch_bounds__(256) void upconv_collapse_kernel(const float* __restrict__ wp5, float* __restrict__ wpc, int C,
                                                              int K) {
    const long total = (long)C * 9 * 4 * K;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % K);
    long t = i / K;
    const int pq = (int)(t % 4); t /= 4;
    const int rs = (int)(t % 9);
    const int c = (int)(t / 9);
    const int p = pq >> 1, q = pq & 1, r = rs / 3, s_ = rs % 3;
    float v = 0.f;
    for (int a = 0; a < 5; ++a) {
        if (upconv_group(p, a) != r) continue;
        for (int b = 0; b < 5; ++b)
            if (upconv_group(q, b) == s_) v += wp5[((long)c * 25 + a * 5 + b) * K + k];
    }
    wpc[i] = v;
}

__global__ __launch_bounds__(256) void upconv_expand_kernel(const float* __restrict__ dwpc, float* __restrict__ dwp5, int C,
                                                            int K, int accumulate) {
    const long total = (long)C * 25 * K;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % K);
    long t = i / K;
    const int ab = (int)(t % 25);
    const int c = (int)(t / 25);
    const int a = ab / 5, b = ab % 5;
    float v = 0.f;
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
        const int rs = upconv_group(pq >> 1, a) * 3 + upconv_group(pq & 1, b);
        v += dwpc[(((long)c * 9 + rs) * 4 + pq) * K + k];
    }
    dwp5[i] = accumulate ? dwp5[i] + v : v;
}

// all collapsed up-sample convolutions of a net in one launch: item i owns blocks [block_begin, next.block_begin);
// its first 36*C*K threads collapse the weights, the next 4*K tile the bias
struct CollapseItem {
    const float* wp5;
    const float* bias;
    float* wpc;
    float* bias4;
    int C, K, block_begin, mode;      // mode 0: Upscale2D (nearest) -> 5x5; 1: BilinearUpsample2D -> 3x3 (conv_bilinear.hip)
};

// BilinearUpsample2DLayer(2) -> 3x3 'same' convolution on the zero-extended coarse grid (conv_bilinear.hip: the U0 operator):
// coefficient of fine correlation tap a (offset a - 1) in coarse tap r (offset r - 1) for output parity p
//   p = 0 (u[2m-1], u[2m], u[2m+1] = (x[m-1] + x[m]) / 2, x[m], (x[m] + x[m+1]) / 2):  r = 0: (1/2, 0, 0)  1: (1/2, 1, 1/2)  2: (0, 0, 1/2)
//   p = 1 (u[2m], u[2m+1], u[2m+2] = x[m], (x[m] + x[m+1]) / 2, x[m+1]):              r = 0: none         1: (1, 1/2, 0)    2: (0, 1/2, 1)
__device__ __forceinline__ float blconv_coef(int p, int r, int a) {
    if (p == 0) return r == 1 ? (a == 1 ? 1.f : 0.5f) : ((r == 0 && a == 0) || (r == 2 && a == 2) ? 0.5f : 0.f);
    return r == 0 ? 0.f : (a == 1 ? 0.5f : ((r == 1 && a == 0) || (r == 2 && a == 2) ? 1.f : 0.f));
}

__global__ __launch_bounds__(256) void upconv_collapse_batched_kernel(const CollapseItem* __restrict__ items, int n) {
    int li = 0;
    for (int i = 1; i < n; ++i)
        if ((int)blockIdx.x >= items[i].block_begin) li = i;
    const CollapseItem it = items[li];
    const long total = (long)it.C * 36 * it.K;
    const long i = (long)(blockIdx.x - it.block_begin) * 256 + threadIdx.x;
    if (i < total) {
        const int K = it.K;
        const int k = (int)(i % K);
        long t = i / K;
        const int pq = (int)(t % 4); t /= 4;
        const int rs = (int)(t % 9);
        const int c = (int)(t / 9);
        const int p = pq >> 1, q = pq & 1, r = rs / 3, s_ = rs % 3;
        float v = 0.f;
        if (it.mode == 1) {
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    const float cf = blconv_coef(p, r, a) * blconv_coef(q, s_, b);
                    if (cf != 0.f) v += cf * it.wp5[((long)c * 9 + a * 3 + b) * K + k];
                }
        } else {
            for (int a = 0; a < 5; ++a) {
                if (upconv_group(p, a) != r) continue;
                for (int b = 0; b < 5; ++b)
                    if (upconv_group(q, b) == s_) v += it.wp5[((long)c * 25 + a * 5 + b) * K + k];
            }
        }
        it.wpc[i] = v;
    } else if (it.bias && it.bias4 && i < total + 4L * it.K) {
        const int j = (int)(i - total);
        it.bias4[j] = it.bias[j % it.K];
    }
}

struct ExpandItem {
    const float* dwpc;
    float* dwp5;
    int C, K, block_begin, mode;      // as CollapseItem
};

__global__ __launch_bounds__(256) void upconv_expand_batched_kernel(const ExpandItem* __restrict__ items, int n, int accumulate) {
    int li = 0;
    for (int i = 1; i < n; ++i)
        if ((int)blockIdx.x >= items[i].block_begin) li = i;
    const ExpandItem it = items[li];
    const long total = (long)it.C * (it.mode == 1 ? 9 : 25) * it.K;
    const long i = (long)(blockIdx.x - it.block_begin) * 256 + threadIdx.x;
    if (i >= total) return;
    const int K = it.K;
    const int k = (int)(i % K);
    long t = i / K;
    if (it.mode == 1) {                 // the transposed tap map of the bilinear collapse
        const int ab = (int)(t % 9);
        const int c = (int)(t / 9);
        const int a = ab / 3, b = ab % 3;
        float v = 0.f;
        for (int pq = 0; pq < 4; ++pq)
            for (int rs = 0; rs < 9; ++rs) {
                const float cf = blconv_coef(pq >> 1, rs / 3, a) * blconv_coef(pq & 1, rs % 3, b);
                if (cf != 0.f) v += cf * it.dwpc[(((long)c * 9 + rs) * 4 + pq) * K + k];
            }
        it.dwp5[i] = accumulate ? it.dwp5[i] + v : v;
        return;
    }
    const int ab = (int)(t % 25);
    const int c = (int)(t / 25);
    const int a = ab / 5, b = ab % 5;
    float v = 0.f;
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
        const int rs = upconv_group(pq >> 1, a) * 3 + upconv_group(pq & 1, b);
        v += it.dwpc[(((long)c * 9 + rs) * 4 + pq) * K + k];
    }
    it.dwp5[i] = accumulate ? it.dwp5[i] + v : v;
}

__global__ __launch_bounds__(256) void bias_tile4_kernel(const float* __restrict__ b, float* __restrict__ bc, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 4 * K) bc[i] = b[i % K];
}

// pp [4N, K, H, W] <-> hi [N, K, 2H, 2W]; one thread per low-res pixel moves its 2x2 block (16-byte rows)
__global__ __launch_bounds__(256) void pp_to_hi_kernel(const float* __restrict__ pp, float* __restrict__ hi, long hi_nstride,
                                                       int N, int K, int H, int W) {
    const long total = (long)N * K * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    long t = i / W;
    const int y = (int)(t % H); t /= H;
    const int k = (int)(t % K);
    const int n = (int)(t / K);
    const long plane = (long)K * H * W;
    const float* src = pp + ((long)n * 4) * plane + ((long)k * H + y) * W + x;
    float* dst = hi + (long)n * hi_nstride + ((long)k * 2 * H + 2 * y) * (2 * W) + 2 * x;
    *reinterpret_cast<float2*>(dst) = make_float2(src[0], src[plane]);
    *reinterpret_cast<float2*>(dst + 2 * W) = make_float2(src[2 * plane], src[3 * plane]);
}

__global__ __launch_bounds__(256) void hi_to_pp_kernel(const float* __restrict__ hi, long hi_nstride, float* __restrict__ pp,
                                                       int N, int K, int H, int W) {
    const long total = (long)N * K * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    long t = i / W;
    const int y = (int)(t % H); t /= H;
    const int k = (int)(t % K);
    const int n = (int)(t / K);
    const long plane = (long)K * H * W;
    float* dst = pp + ((long)n * 4) * plane + ((long)k * H + y) * W + x;
    const float* src = hi + (long)n * hi_nstride + ((long)k * 2 * H + 2 * y) * (2 * W) + 2 * x;
    const float2 a = *reinterpret_cast<const float2*>(src), b = *reinterpret_cast<const float2*>(src + 2 * W);
    dst[0] = a.x; dst[plane] = a.y; dst[2 * plane] = b.x; dst[3 * plane] = b.y;
}


// ---- DropoutLayer(p, rescale=True) (architectures/p2p.py:200-223, dcgan.py:25-26): y = x * mask / (1 - p).
// The mask is a counter-based hash of (element index, layer key, step counter): nothing is stored, the backward
// recomputes it, and the oracle evaluates the same hash (Theano's MRG stream itself cannot be reproduced).
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(unsigned idx, unsigned key, unsigned step, float p) {
    const unsigned hsh = lowbias32(lowbias32(idx ^ key) + step * 0x9e3779b9U);
    return (float)(hsh >> 8) * (1.0f / 16777216.0f) >= p;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long xs, float* __restrict__ y, long ys,
                                                      int N, int C, int HW, float p, unsigned key,
                                                      const unsigned* __restrict__ counter) {
    const long total = (long)N * C * HW;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long chw = (long)C * HW;
    const int n = (int)(i / chw);
    const long r = i - (long)n * chw;
    const float scale = 1.0f / (1.0f - p);
    y[(long)n * ys + r] = dropout_keep((unsigned)i, key, *counter, p) ? x[(long)n * xs + r] * scale : 0.f;
}
__global__ void counter_tick_kernel(unsigned* counter) { *counter += 1u; }

#define EW_GRID(total) dim3(ceil_div((long)(total), 256)), dim3(256), 0, ctx->stream

// ---- small tensors: one block per channel does the whole layer (N * HW <= BN_SMALL_MAX values per channel) ----
// The partial / final / apply form above is three dependent launches; for the 1x1 .. 64x64 maps of the U-Net bottleneck
// and the first DCGAN stages those launches ARE the cost (chains of ~150 of them per step on the stage streams).
#define BN_SMALL_MAX 16384

template <int VEC>
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(const float* __restrict__ x, long xs, float* __restrict__ y, long ys,
                                                           int N, int HW, float eps, float* __restrict__ mean,
                                                           float* __restrict__ inv, float* run_mean, float* run_inv, float ra,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int act, float alpha, int apply) {
    const int c = blockIdx.x;
    const int units = HW / VEC, total = N * units;
    const float* xc = x + (long)c * HW;
    double a = 0.0, b = 0.0;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int n = e / units, i = (e - n * units) * VEC;
        if constexpr (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(xc + n * xs + i);
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            a += (x0 + x1) + (x2 + x3);
            b += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
        } else {
            const double v = xc[n * xs + i];
            a += v;
            b += v * v;
        }
    }
    __shared__ double red[4];
    a = block_sum(a, red);
    b = block_sum(b, red);
    const double count = (double)N * HW;
    const double mu = a / count;
    double var = b / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float m = (float)mu;
    const float iv = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mean[c] = m;
        inv[c] = iv;
        if (run_mean) {
            run_mean[c] = (1.f - ra) * run_mean[c] + ra * m;
            run_inv[c] = (1.f - ra) * run_inv[c] + ra * iv;
        }
    }
    if (!apply) return;
    const float sc = gamma[c] * iv, be = beta[c];
    float* yc = y + (long)c * HW;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int n = e / units, i = (e - n * units) * VEC;
        if constexpr (VEC == 4) {
            float4 t = *reinterpret_cast<const float4*>(xc + n * xs + i);
            t.x = ghm_act(fmaf(t.x - m, sc, be), act, alpha);
            t.y = ghm_act(fmaf(t.y - m, sc, be), act, alpha);
            t.z = ghm_act(fmaf(t.z - m, sc, be), act, alpha);
            t.w = ghm_act(fmaf(t.w - m, sc, be), act, alpha);
            *reinterpret_cast<float4*>(yc + n * ys + i) = t;
        } else {
            yc[n * ys + i] = ghm_act(fmaf(xc[n * xs + i] - m, sc, be), act, alpha);
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const float* __restrict__ dout, long ds, const float* __restrict__ y,
                                                           long ys, const float* __restrict__ x, long xs,
                                                           float* __restrict__ dx, long dxs, int N, int HW,
                                                           const float* __restrict__ mean, const float* __restrict__ inv,
                                                           const float* __restrict__ gamma, float* dgamma, float* dbeta,
                                                           int act, float alpha, int accumulate,
                                                           const float* __restrict__ beta = nullptr) {
    const int c = blockIdx.x;
    const int units = HW / VEC, total = N * units;
    const long row = (long)c * HW;
    const float m = mean[c], iv = inv[c];
    const float sc = gamma[c] * iv, be = y ? 0.f : beta[c];
    double a = 0.0, b = 0.0;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int n = e / units, i = (e - n * units) * VEC;
        if constexpr (VEC == 4) {
            const float4 d = *reinterpret_cast<const float4*>(dout + n * ds + row + i);
            const float4 xx = *reinterpret_cast<const float4*>(x + n * xs + row + i);
            const float4 yy = bn_y4(y, n * ys + row + i, xx, m, sc, be, act, alpha);
            const float z0 = d.x * ghm_dact_from_out(yy.x, act, alpha), z1 = d.y * ghm_dact_from_out(yy.y, act, alpha);
            const float z2 = d.z * ghm_dact_from_out(yy.z, act, alpha), z3 = d.w * ghm_dact_from_out(yy.w, act, alpha);
            a += ((double)z0 + (double)z1) + ((double)z2 + (double)z3);
            b += ((double)z0 * ((xx.x - m) * iv) + (double)z1 * ((xx.y - m) * iv)) +
                 ((double)z2 * ((xx.z - m) * iv) + (double)z3 * ((xx.w - m) * iv));
        } else {
            const float xv = x[n * xs + row + i];
            const float dz = dout[n * ds + row + i] * ghm_dact_from_out(bn_y(y, n * ys + row + i, xv, m, sc, be, act, alpha), act, alpha);
            const float xh = (xv - m) * iv;
            a += dz;
            b += (double)dz * xh;
        }
    }
    __shared__ double red[4];
    a = block_sum(a, red);
    b = block_sum(b, red);
    const float fa = (float)a, fb = (float)b;
    if (threadIdx.x == 0) {
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + fa;
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + fb;
    }
    const float inv_count = 1.f / (float)((long)N * HW);
    const float g = gamma[c] * iv, mb = fa * inv_count, mg = fb * inv_count;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int n = e / units, i = (e - n * units) * VEC;
        if constexpr (VEC == 4) {
            const float4 d = *reinterpret_cast<const float4*>(dout + n * ds + row + i);
            const float4 xx = *reinterpret_cast<const float4*>(x + n * xs + row + i);
            const float4 yy = bn_y4(y, n * ys + row + i, xx, m, sc, be, act, alpha);
            float4 r;
            r.x = g * (d.x * ghm_dact_from_out(yy.x, act, alpha) - mb - (xx.x - m) * iv * mg);
            r.y = g * (d.y * ghm_dact_from_out(yy.y, act, alpha) - mb - (xx.y - m) * iv * mg);
            r.z = g * (d.z * ghm_dact_from_out(yy.z, act, alpha) - mb - (xx.z - m) * iv * mg);
            r.w = g * (d.w * ghm_dact_from_out(yy.w, act, alpha) - mb - (xx.w - m) * iv * mg);
            *reinterpret_cast<float4*>(dx + n * dxs + row + i) = r;
        } else {
            const float xv = x[n * xs + row + i];
            const float dz = dout[n * ds + row + i] * ghm_dact_from_out(bn_y(y, n * ys + row + i, xv, m, sc, be, act, alpha), act, alpha);
            const float xh = (xv - m) * iv;
            dx[n * dxs + row + i] = g * (dz - mb - xh * mg);
        }
    }
}

static bool bn_small(long count) { return count <= BN_SMALL_MAX && GHM_OPT("GHM_NO_BN_SMALL") == nullptr; }

extern "C" {

size_t ghm_bn_workspace(int32_t C) { return (size_t)C * BN_MAX_SPLIT * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float); }

static int bn_split(int C, long count) {
    long S = (1024 + C - 1) / C;
    long maxs = count / 2048;
    if (maxs < 1) maxs = 1;
    if (S > maxs) S = maxs;
    if (S > BN_MAX_SPLIT) S = BN_MAX_SPLIT;
    if (S < 1) S = 1;
    return (int)S;
}

// split of the row-structured partial kernels: S = N * segs <= BN_MAX_SPLIT blocks per channel, ~8 blocks per CU in
// total, at least 2048 elements per block; 0 when the geometry does not allow 16-byte row loads
static int bn_row_segs(int N, int C, int HW, int* seg_len) {
    if (HW % 4 != 0 || N > BN_MAX_SPLIT) return 0;
    long segs = (2048 + (long)C * N - 1) / ((long)C * N);
    const long by_split = BN_MAX_SPLIT / N, by_len = HW / 2048 > 0 ? HW / 2048 : 1;
    if (segs > by_split) segs = by_split;
    if (segs > by_len) segs = by_len;
    if (segs < 1) segs = 1;
    int len = (int)((HW + segs - 1) / segs);
    len = (len + 3) / 4 * 4;
    *seg_len = len;
    return (int)((HW + len - 1) / len);
}

int ghm_bn_stats(ghm_ctx* ctx, const float* x, int32_t N, int32_t C, int32_t HW, int64_t nstride, float eps, float* mean,
                 float* inv, float* run_mean, float* run_inv, float run_alpha, void* ws) {
    const long count = (long)N * HW;
    if (bn_small(count)) {
        if (HW % 4 == 0 && nstride % 4 == 0 && aligned16(x))
            hipLaunchKernelGGL((bn_fwd_small_kernel<4>), dim3(C), dim3(256), 0, ctx->stream, x, (long)nstride, nullptr, 0L, N, HW,
                               eps, mean, inv, run_mean, run_inv, run_alpha, nullptr, nullptr, 0, 0.f, 0);
        else
            hipLaunchKernelGGL((bn_fwd_small_kernel<1>), dim3(C), dim3(256), 0, ctx->stream, x, (long)nstride, nullptr, 0L, N, HW,
                               eps, mean, inv, run_mean, run_inv, run_alpha, nullptr, nullptr, 0, 0.f, 0);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    int S = bn_split(C, count), seg_len = 0;
    const int segs = (nstride % 4 == 0 && aligned16(x)) ? bn_row_segs(N, C, HW, &seg_len) : 0;
    if (segs > 0) {
        S = N * segs;
        hipLaunchKernelGGL((bn_rows_partial<false>), dim3(S, C), dim3(256), 0, ctx->stream, x, (long)nstride, nullptr, 0L,
                           nullptr, 0L, HW, segs, seg_len, nullptr, nullptr, 0, 0.f, (double*)ws);
    } else {
        hipLaunchKernelGGL(bn_stats_partial, dim3(S, C), dim3(256), 0, ctx->stream, x, N, HW, (long)nstride, S, (double*)ws);
    }
    GHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_final, dim3(ceil_div(C, 256)), dim3(256), 0, ctx->stream, (const double*)ws, C, S,
                       (double)count, eps, mean, inv, run_mean, run_inv, run_alpha);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_bn_apply(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW,
                 const float* mean, const float* inv, const float* gamma, const float* beta, int32_t act, float alpha) {
    const View v{N, C, HW};
    if (HW % 4 == 0 && xs % 4 == 0 && ys % 4 == 0 && aligned16(x) && aligned16(y)) {
        hipLaunchKernelGGL((bn_apply_kernel<4>), EW_GRID((long)N * C * (HW / 4)), x, (long)xs, y, (long)ys, v, mean, inv,
                           gamma, beta, act, alpha);
    } else {
        hipLaunchKernelGGL((bn_apply_kernel<1>), EW_GRID((long)N * C * HW), x, (long)xs, y, (long)ys, v, mean, inv, gamma,
                           beta, act, alpha);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_bn_forward(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW, float eps,
                   float* mean, float* inv, float* run_mean, float* run_inv, float run_alpha, const float* gamma,
                   const float* beta, int32_t act, float alpha, void* ws) {
    if (bn_small((long)N * HW)) {
        if (HW % 4 == 0 && xs % 4 == 0 && ys % 4 == 0 && aligned16(x) && aligned16(y))
            hipLaunchKernelGGL((bn_fwd_small_kernel<4>), dim3(C), dim3(256), 0, ctx->stream, x, (long)xs, y, (long)ys, N, HW, eps,
                               mean, inv, run_mean, run_inv, run_alpha, gamma, beta, act, alpha, 1);
        else
            hipLaunchKernelGGL((bn_fwd_small_kernel<1>), dim3(C), dim3(256), 0, ctx->stream, x, (long)xs, y, (long)ys, N, HW, eps,
                               mean, inv, run_mean, run_inv, run_alpha, gamma, beta, act, alpha, 1);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    if (int e = ghm_bn_stats(ctx, x, N, C, HW, xs, eps, mean, inv, run_mean, run_inv, run_alpha, ws)) return e;
    return ghm_bn_apply(ctx, x, xs, y, ys, N, C, HW, mean, inv, gamma, beta, act, alpha);
}

// ---- InstanceNorm (north_star names it beside BatchNorm; the reference itself only ever builds BatchNormLayer,
// architectures/p2p.py:146-268): the statistics of BatchNorm taken per (sample, channel) over the map instead of per channel
// over the batch -- y[n] = act((x[n] - mean[n]) * gamma * inv[n] + beta), no running statistics (the deterministic pass
// normalises with the sample's own statistics too).  These are the BatchNorm kernels on one-sample views: N launches of the
// one-launch form for maps <= 16384 pixels (a block per (n, c) row), the three-launch form above that; the same fp64 sums.
// ``group`` consecutive samples of the tensor form one instance (1; 4 for the parity-planar output [4N, K, H, W] of a collapsed
// up-sample convolution, whose four parity planes are one image).  mean / inv: [N / group][C]; gamma / beta / dgamma / dbeta:
// [C]; ws: ghm_bn_workspace(C).
int ghm_instance_norm_fwd(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW, float eps,
                          float* mean, float* inv, const float* gamma, const float* beta, int32_t act, float alpha, void* ws,
                          int32_t group) {
    GHM_CHECK(ctx && x && y && mean && inv && gamma && beta && ws, "ghm_instance_norm_fwd: null argument");
    GHM_CHECK(group >= 1 && N % group == 0, "ghm_instance_norm_fwd: %d samples are not whole groups of %d", N, group);
    for (int n = 0; n < N / group; ++n)
        if (int e = ghm_bn_forward(ctx, x + (long)n * group * xs, xs, y + (long)n * group * ys, ys, group, C, HW, eps,
                                   mean + (long)n * C, inv + (long)n * C, nullptr, nullptr, 0.f, gamma, beta, act, alpha, ws))
            return e;
    return 0;
}

// dz = dout * act'(y) with y recomputed from x; per sample: dx = gamma * inv * (dz - mean_hw(dz) - xhat * mean_hw(dz * xhat));
// dgamma / dbeta sum over the samples (written, or added to when ``accumulate``)
int ghm_instance_norm_bwd(ghm_ctx* ctx, const float* dout, int64_t ds, const float* x, int64_t xs, float* dx, int64_t dxs, int32_t N,
                          int32_t C, int32_t HW, const float* mean, const float* inv, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, int32_t act, float alpha, int32_t accumulate, void* ws, int32_t group) {
    GHM_CHECK(ctx && dout && x && dx && mean && inv && gamma && beta && dgamma && dbeta && ws, "ghm_instance_norm_bwd: null argument");
    GHM_CHECK(group >= 1 && N % group == 0, "ghm_instance_norm_bwd: %d samples are not whole groups of %d", N, group);
    for (int n = 0; n < N / group; ++n)
        if (int e = ghm_bn_backward_x(ctx, dout + (long)n * group * ds, ds, x + (long)n * group * xs, xs, dx + (long)n * group * dxs, dxs,
                                      group, C, HW, mean + (long)n * C, inv + (long)n * C, gamma, beta, dgamma, dbeta, act, alpha,
                                      (accumulate || n > 0) ? 1 : 0, ws))
            return e;
    return 0;
}

// the reduction passes of the BatchNorm backward: dgamma / dbeta and, in the workspace tail, the two per-channel sums the
// apply pass needs (shared with elementwise_q.hip)
int ghm_bn_backward_sums(ghm_ctx* ctx, const float* dout, int64_t ds, const float* y, int64_t ys, const float* x, int64_t xs,
                         int32_t N, int32_t C, int32_t HW, const float* mean, const float* inv, float* dgamma, float* dbeta,
                         int32_t act, float alpha, int32_t accumulate, void* ws, const float* gamma, const float* beta) {
    GHM_CHECK(y != nullptr || (gamma != nullptr && beta != nullptr), "BatchNorm backward without y needs gamma and beta");
    const long count = (long)N * HW;
    int S = bn_split(C, count), seg_len = 0;
    double* wsd = (double*)ws;
    float* sums = (float*)((char*)ws + (size_t)C * BN_MAX_SPLIT * 2 * sizeof(double));
    const bool vec = HW % 4 == 0 && ds % 4 == 0 && (!y || ys % 4 == 0) && xs % 4 == 0 && aligned16(dout) && aligned16(y) && aligned16(x);
    const int segs = vec ? bn_row_segs(N, C, HW, &seg_len) : 0;
    if (segs > 0) {
        S = N * segs;
        hipLaunchKernelGGL((bn_rows_partial<true>), dim3(S, C), dim3(256), 0, ctx->stream, x, (long)xs, dout, (long)ds, y,
                           (long)ys, HW, segs, seg_len, mean, inv, act, alpha, wsd, gamma, beta);
    } else {
        hipLaunchKernelGGL(bn_bwd_partial, dim3(S, C), dim3(256), 0, ctx->stream, dout, (long)ds, y, (long)ys, x, (long)xs, N,
                           HW, S, mean, inv, act, alpha, wsd, gamma, beta);
    }
    GHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final, dim3(ceil_div(C, 256)), dim3(256), 0, ctx->stream, (const double*)wsd, C, S, sums,
                       dgamma, dbeta, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

// the finishing pass of the BatchNorm backward reductions (S partials per channel in the workspace -> sums, dgamma, dbeta)
int ghm_bn_backward_finish(ghm_ctx* ctx, const double* wsd, int32_t C, int32_t S, float* sums, float* dgamma, float* dbeta,
                           int32_t accumulate) {
    hipLaunchKernelGGL(bn_bwd_final, dim3(ceil_div(C, 256)), dim3(256), 0, ctx->stream, wsd, C, S, sums, dgamma, dbeta, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

static int bn_backward_impl(ghm_ctx* ctx, const float* dout, int64_t ds, const float* y, int64_t ys, const float* x, int64_t xs,
                            float* dx, int64_t dxs, int32_t N, int32_t C, int32_t HW, const float* mean, const float* inv,
                            const float* gamma, const float* beta, float* dgamma, float* dbeta, int32_t act, float alpha,
                            int32_t accumulate, void* ws) {
    const long count = (long)N * HW;
    GHM_CHECK(y != nullptr || beta != nullptr, "BatchNorm backward without y needs beta");
    if (bn_small(count)) {
        if (HW % 4 == 0 && ds % 4 == 0 && (!y || ys % 4 == 0) && xs % 4 == 0 && dxs % 4 == 0 && aligned16(dout) && aligned16(y) &&
            aligned16(x) && aligned16(dx))
            hipLaunchKernelGGL((bn_bwd_small_kernel<4>), dim3(C), dim3(256), 0, ctx->stream, dout, (long)ds, y, (long)ys, x,
                               (long)xs, dx, (long)dxs, N, HW, mean, inv, gamma, dgamma, dbeta, act, alpha, accumulate, beta);
        else
            hipLaunchKernelGGL((bn_bwd_small_kernel<1>), dim3(C), dim3(256), 0, ctx->stream, dout, (long)ds, y, (long)ys, x,
                               (long)xs, dx, (long)dxs, N, HW, mean, inv, gamma, dgamma, dbeta, act, alpha, accumulate, beta);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    if (int e = ghm_bn_backward_sums(ctx, dout, ds, y, ys, x, xs, N, C, HW, mean, inv, dgamma, dbeta, act, alpha, accumulate, ws,
                                     gamma, beta))
        return e;
    const float* sums = (const float*)((char*)ws + (size_t)C * BN_MAX_SPLIT * 2 * sizeof(double));
    const bool vec = HW % 4 == 0 && ds % 4 == 0 && (!y || ys % 4 == 0) && xs % 4 == 0 && dxs % 4 == 0 && aligned16(dout) &&
                     aligned16(y) && aligned16(x) && aligned16(dx);
    const View v{N, C, HW};
    if (vec) {
        hipLaunchKernelGGL((bn_bwd_apply<4>), EW_GRID((long)N * C * (HW / 4)), dout, (long)ds, y, (long)ys, x, (long)xs, dx,
                           (long)dxs, v, mean, inv, gamma, sums, 1.f / (float)count, act, alpha, beta);
    } else {
        hipLaunchKernelGGL((bn_bwd_apply<1>), EW_GRID((long)N * C * HW), dout, (long)ds, y, (long)ys, x, (long)xs, dx,
                           (long)dxs, v, mean, inv, gamma, sums, 1.f / (float)count, act, alpha, beta);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_bn_backward(ghm_ctx* ctx, const float* dout, int64_t ds, const float* y, int64_t ys, const float* x, int64_t xs,
                    float* dx, int64_t dxs, int32_t N, int32_t C, int32_t HW, const float* mean, const float* inv,
                    const float* gamma, float* dgamma, float* dbeta, int32_t act, float alpha, int32_t accumulate,
                    void* ws) {
    GHM_CHECK(y != nullptr, "ghm_bn_backward: y == NULL (use ghm_bn_backward_x, which recomputes it from x)");
    return bn_backward_impl(ctx, dout, ds, y, ys, x, xs, dx, dxs, N, C, HW, mean, inv, gamma, nullptr, dgamma, dbeta, act, alpha,
                            accumulate, ws);
}

int ghm_bn_backward_x(ghm_ctx* ctx, const float* dout, int64_t ds, const float* x, int64_t xs, float* dx, int64_t dxs,
                      int32_t N, int32_t C, int32_t HW, const float* mean, const float* inv, const float* gamma,
                      const float* beta, float* dgamma, float* dbeta, int32_t act, float alpha, int32_t accumulate, void* ws) {
    return bn_backward_impl(ctx, dout, ds, nullptr, 0, x, xs, dx, dxs, N, C, HW, mean, inv, gamma, beta, dgamma, dbeta, act, alpha,
                            accumulate, ws);
}

int ghm_act_fwd(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW,
                int32_t act, float alpha) {
    const View v{N, C, HW};
    if (HW % 4 == 0 && xs % 4 == 0 && ys % 4 == 0 && aligned16(x) && aligned16(y)) {
        hipLaunchKernelGGL((act_fwd_kernel<4>), EW_GRID((long)N * C * (HW / 4)), x, (long)xs, y, (long)ys, v, act, alpha);
    } else {
        hipLaunchKernelGGL((act_fwd_kernel<1>), EW_GRID((long)N * C * HW), x, (long)xs, y, (long)ys, v, act, alpha);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_act_bwd(ghm_ctx* ctx, const float* dout, int64_t ds, const float* y, int64_t ys, float* dx, int64_t dxs,
                int32_t N, int32_t C, int32_t HW, int32_t act, float alpha, int32_t accumulate) {
    const View v{N, C, HW};
    if (HW % 4 == 0) {
        hipLaunchKernelGGL((act_bwd_kernel<4>), EW_GRID((long)N * C * (HW / 4)), dout, (long)ds, y, (long)ys, dx, (long)dxs,
                           v, act, alpha, accumulate);
    } else {
        hipLaunchKernelGGL((act_bwd_kernel<1>), EW_GRID((long)N * C * HW), dout, (long)ds, y, (long)ys, dx, (long)dxs, v,
                           act, alpha, accumulate);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_copy_view(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW,
                  int32_t accumulate) {
    const View v{N, C, HW};
    if (HW % 4 == 0) {
        hipLaunchKernelGGL((copy_view_kernel<4>), EW_GRID((long)N * C * (HW / 4)), x, (long)xs, y, (long)ys, v, accumulate);
    } else {
        hipLaunchKernelGGL((copy_view_kernel<1>), EW_GRID((long)N * C * HW), x, (long)xs, y, (long)ys, v, accumulate);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_scale_samples(ghm_ctx* ctx, float* x, int64_t xs, int32_t N, int32_t C, int32_t HW, const float* num, int64_t num_nstride,
                      const float* den, int64_t den_nstride) {
    const View v{N, C, HW};
    if (HW % 4 == 0) {
        hipLaunchKernelGGL((scale_samples_kernel<4>), EW_GRID((long)N * C * (HW / 4)), x, (long)xs, v, num, (long)num_nstride, den,
                           (long)den_nstride);
    } else {
        hipLaunchKernelGGL((scale_samples_kernel<1>), EW_GRID((long)N * C * HW), x, (long)xs, v, num, (long)num_nstride, den,
                           (long)den_nstride);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_axpby(ghm_ctx* ctx, float a, const float* x, float b, float* y, int64_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpby_kernel, EW_GRID(n), a, x, b, y, (long)n);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_maxpool2_fwd(ghm_ctx* ctx, const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W) {
    GHM_CHECK(H % 2 == 0 && W % 2 == 0, "maxpool2 needs even H, W (got %dx%d)", H, W);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, EW_GRID((long)N * C * (H / 2) * (W / 2)), x, y, (long)N * C, H, W);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_maxpool2_bwd(ghm_ctx* ctx, const float* x, const float* y, const float* dy, float* dx, int32_t N, int32_t C,
                     int32_t H, int32_t W, int32_t act, float alpha) {
    GHM_CHECK(H % 2 == 0 && W % 2 == 0, "maxpool2 needs even H, W (got %dx%d)", H, W);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, EW_GRID((long)N * C * (H / 2) * (W / 2)), x, y, dy, dx, (long)N * C, H, W, act,
                       alpha);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_maxpool2_mask_bwd(ghm_ctx* ctx, const uint8_t* mask, const float* y, const float* dy, float* dx, int32_t N,
                          int32_t C, int32_t H, int32_t W, int32_t act, float alpha) {
    GHM_CHECK(H % 2 == 0 && W % 4 == 0, "maxpool2_mask_bwd needs even H and W %% 4 == 0 (got %dx%d)", H, W);
    hipLaunchKernelGGL((maxpool2_mask_bwd_kernel<false>), EW_GRID((long)N * C * (H / 2) * (W / 4)), mask, y, dy, dx,
                       (long)N * C, H, W, act, alpha, (float*)nullptr, C, 0);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_maxpool2_mask_bwd_bias(ghm_ctx* ctx, const uint8_t* mask, const float* y, const float* dy, float* dx, int32_t N,
                               int32_t C, int32_t H, int32_t W, int32_t act, float alpha, float* dbias, int32_t accumulate) {
    GHM_CHECK(H % 2 == 0 && W % 4 == 0, "maxpool2_mask_bwd needs even H and W %% 4 == 0 (got %dx%d)", H, W);
    const long per_plane = (long)(H / 2) * (W / 4);
    if (per_plane % 256 != 0) {          // blocks would straddle planes: two passes
        if (int e = ghm_maxpool2_mask_bwd(ctx, mask, y, dy, dx, N, C, H, W, act, alpha)) return e;
        return ghm_channel_sum(ctx, dx, N, C, H * W, (int64_t)C * H * W, dbias, accumulate);
    }
    const int bpp = (int)(per_plane / 256), S = N * bpp;
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, (size_t)C * S * sizeof(float), &ws)) return e;
    hipLaunchKernelGGL((maxpool2_mask_bwd_kernel<true>), EW_GRID((long)N * C * per_plane), mask, y, dy, dx, (long)N * C, H,
                       W, act, alpha, (float*)ws, C, bpp);
    GHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(C), dim3(64), 0, ctx->stream, (const float*)ws, C, S, dbias, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_avgpool_fwd(ghm_ctx* ctx, const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t p) {
    hipLaunchKernelGGL(avgpool_fwd_kernel, EW_GRID((long)N * C * (H / p) * (W / p)), x, y, (long)N * C, H, W, p);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_avgpool_bwd(ghm_ctx* ctx, const float* dy, float* dx, int32_t N, int32_t C, int32_t H, int32_t W, int32_t p) {
    hipLaunchKernelGGL(avgpool_bwd_kernel, EW_GRID((long)N * C * H * W), dy, dx, (long)N * C, H, W, p);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upconv_collapse_weights(ghm_ctx* ctx, const float* wp5, const float* bias, float* wpc, float* bias4, int32_t C,
                                int32_t K) {
    hipLaunchKernelGGL(upconv_collapse_kernel, EW_GRID((long)C * 36 * K), wp5, wpc, C, K);
    GHM_LAUNCH_CHECK();
    if (bias && bias4) {
        hipLaunchKernelGGL(bias_tile4_kernel, EW_GRID(4L * K), bias, bias4, K);
        GHM_LAUNCH_CHECK();
    }
    return 0;
}

int ghm_upconv_collapse_batched(ghm_ctx* ctx, const void* table, int32_t n_items, int32_t total_blocks) {
    if (n_items <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(upconv_collapse_batched_kernel, dim3(total_blocks), dim3(256), 0, ctx->stream,
                       (const CollapseItem*)table, n_items);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upconv_expand_batched(ghm_ctx* ctx, const void* table, int32_t n_items, int32_t total_blocks, int32_t accumulate) {
    if (n_items <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(upconv_expand_batched_kernel, dim3(total_blocks), dim3(256), 0, ctx->stream, (const ExpandItem*)table,
                       n_items, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upconv_expand_wgrad(ghm_ctx* ctx, const float* dwpc, float* dwp5, int32_t C, int32_t K, int32_t accumulate) {
    hipLaunchKernelGGL(upconv_expand_kernel, EW_GRID((long)C * 25 * K), dwpc, dwp5, C, K, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_pp_to_hi(ghm_ctx* ctx, const float* pp, float* hi, int64_t hi_nstride, int32_t N, int32_t K, int32_t H, int32_t W) {
    GHM_CHECK(((uintptr_t)hi % 8 == 0) && hi_nstride % 2 == 0, "pp_to_hi: 8-byte aligned destination required");
    hipLaunchKernelGGL(pp_to_hi_kernel, EW_GRID((long)N * K * H * W), pp, hi, (long)hi_nstride, N, K, H, W);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_hi_to_pp(ghm_ctx* ctx, const float* hi, int64_t hi_nstride, float* pp, int32_t N, int32_t K, int32_t H, int32_t W) {
    GHM_CHECK(((uintptr_t)hi % 8 == 0) && hi_nstride % 2 == 0, "hi_to_pp: 8-byte aligned source required");
    hipLaunchKernelGGL(hi_to_pp_kernel, EW_GRID((long)N * K * H * W), hi, (long)hi_nstride, pp, N, K, H, W);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_dropout(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW, float p,
                uint32_t key, const void* counter) {
    GHM_CHECK(p >= 0.f && p < 1.f && (long)N * C * HW < (1L << 32), "ghm_dropout: p in [0,1), < 2^32 elements");
    hipLaunchKernelGGL(dropout_kernel, EW_GRID((long)N * C * HW), x, (long)xs, y, (long)ys, N, C, HW, p, key,
                       (const unsigned*)counter);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_counter_tick(ghm_ctx* ctx, void* counter) {
    hipLaunchKernelGGL(counter_tick_kernel, dim3(1), dim3(1), 0, ctx->stream, (unsigned*)counter);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upsample_nearest2_fwd(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int32_t N, int32_t C, int32_t H,
                              int32_t W) {
    hipLaunchKernelGGL(up_nearest_fwd_kernel, EW_GRID((long)N * C * H * W), x, (long)xs, y, N, C, H, W);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upsample_nearest2_bwd(ghm_ctx* ctx, const float* dy, float* dx, int64_t dxs, int32_t N, int32_t C, int32_t H,
                              int32_t W, int32_t accumulate) {
    hipLaunchKernelGGL(up_nearest_bwd_kernel, EW_GRID((long)N * C * H * W), dy, dx, (long)dxs, N, C, H, W, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upsample_bilinear2_fwd(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int32_t N, int32_t C, int32_t H,
                               int32_t W) {
    hipLaunchKernelGGL(up_bilinear_fwd_kernel, EW_GRID((long)N * C * H * W), x, (long)xs, y, N, C, H, W);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upsample_bilinear2_bwd(ghm_ctx* ctx, const float* dy, float* dx, int64_t dxs, int32_t N, int32_t C, int32_t H,
                               int32_t W, int32_t accumulate) {
    if (W % 2 == 0 && dxs % 2 == 0 && aligned16(dy) && ((uintptr_t)dx & 7) == 0 && (long)N * C <= 65535) {
        hipLaunchKernelGGL(up_bilinear_bwd2_kernel, dim3(ceil_div((long)H * (W / 2), 256), N * C), dim3(256), 0, ctx->stream, dy,
                           dx, (long)dxs, C, H, W, accumulate);
    } else {
        hipLaunchKernelGGL(up_bilinear_bwd_kernel, EW_GRID((long)N * C * H * W), dy, dx, (long)dxs, N, C, H, W, accumulate);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

static int loss_grid(long n) {
    long g = (n + 256 * 8 - 1) / (256 * 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

static int scalar_loss(ghm_ctx* ctx, const float* d, int64_t n, float target, int kind, float* loss_out, float* grad,
                       float grad_scale, int32_t accumulate_loss) {
    const int g = loss_grid(n);
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, (size_t)g * sizeof(double), &ws)) return e;
    hipLaunchKernelGGL(scalar_loss_kernel, dim3(g), dim3(256), 0, ctx->stream, d, (long)n, target, kind, grad, grad_scale,
                       (double*)ws, (const float*)ctx->ls_state, loss_out, accumulate_loss);
    GHM_LAUNCH_CHECK();
    if (g > 1) {
        hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)ws, g, loss_out,
                           accumulate_loss);
        GHM_LAUNCH_CHECK();
    }
    return 0;
}

int ghm_lsgan_loss(ghm_ctx* ctx, const float* d, int64_t n, float target, float* loss_out, float* grad,
                   float grad_scale, int32_t accumulate_loss) {
    return scalar_loss(ctx, d, n, target, 0, loss_out, grad, grad_scale, accumulate_loss);
}

int ghm_bce_loss(ghm_ctx* ctx, const float* p, int64_t n, float target, float* loss_out, float* grad, float grad_scale,
                 int32_t accumulate_loss) {
    return scalar_loss(ctx, p, n, target, 1, loss_out, grad, grad_scale, accumulate_loss);
}

int ghm_recon_loss(ghm_ctx* ctx, const float* a, int64_t as, const float* b, int64_t bs, int32_t N, int32_t C, int32_t HW,
                   int32_t l2, float* loss_out, float* grad, int64_t gs, float grad_scale, int32_t accumulate_grad) {
    const View v{N, C, HW};
    const int g = loss_grid((long)N * C * HW);
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, (size_t)g * sizeof(double), &ws)) return e;
    const bool v4 = ((long)C * HW) % 4 == 0 && as % 4 == 0 && bs % 4 == 0 && (grad == nullptr || gs % 4 == 0) && aligned16(a) &&
                    aligned16(b) && aligned16(grad);
    if (v4)
        hipLaunchKernelGGL((recon_loss_kernel<4>), dim3(g), dim3(256), 0, ctx->stream, a, (long)as, b, (long)bs, v, l2, grad, (long)gs,
                           grad_scale, accumulate_grad, (double*)ws, (const float*)ctx->ls_state);
    else
        hipLaunchKernelGGL((recon_loss_kernel<1>), dim3(g), dim3(256), 0, ctx->stream, a, (long)as, b, (long)bs, v, l2, grad, (long)gs,
                           grad_scale, accumulate_grad, (double*)ws, (const float*)ctx->ls_state);
    GHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)ws, g, loss_out, 0);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_image_batch(ghm_ctx* ctx, const uint8_t* src_nhwc, int32_t N, int32_t H, int32_t W, int32_t C,
                    const double* xform, int32_t tanh_range, float* dst, int64_t dst_nstride) {
    if (N == 0) return 0;
    hipLaunchKernelGGL(image_batch_kernel, EW_GRID((long)N * H * W), src_nhwc, N, H, W, C, xform, tanh_range ? 1 : 0, dst,
                       (long)dst_nstride);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_rmsprop(ghm_ctx* ctx, float* p, const float* g, float* acc, int64_t n, const float* hyper, float rho, float eps,
                float grad_scale) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(rmsprop_kernel, EW_GRID((n + 3) / 4), p, g, acc, (long)n, hyper, rho, eps, grad_scale,
                       (const float*)ctx->ls_state);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_adam(ghm_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float b1, float b2,
             float eps, float grad_scale) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_kernel, EW_GRID(n), p, g, m, v, (long)n, hyper, b1, b2, eps, grad_scale,
                       (const float*)ctx->ls_state);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_adam_tick(ghm_ctx* ctx, float* hyper) {
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, ctx->stream, hyper, (const float*)ctx->ls_state);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_set_loss_scale_state(ghm_ctx* ctx, float* state) {
    ctx->ls_state = state;
    return 0;
}

int ghm_grad_check(ghm_ctx* ctx, const float* g, int64_t n) {
    GHM_CHECK(ctx->ls_state, "ghm_grad_check: no loss-scale state on this context (ghm_set_loss_scale_state)");
    GHM_CHECK(((uintptr_t)g & 15) == 0, "ghm_grad_check: gradient buffer must be 16-byte aligned");
    if (n == 0) return 0;
    const long n4 = (n + 3) / 4;
    const int grid = (int)(n4 < 256L * 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(grad_check_kernel, dim3(grid), dim3(256), 0, ctx->stream, g, (long)n, ctx->ls_state);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_loss_scale_update(ghm_ctx* ctx, int32_t growth_interval, float min_scale, float max_scale) {
    GHM_CHECK(ctx->ls_state, "ghm_loss_scale_update: no loss-scale state on this context");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->ls_state, (float)growth_interval,
                       min_scale, max_scale);
    GHM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
