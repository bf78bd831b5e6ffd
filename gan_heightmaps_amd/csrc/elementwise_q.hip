// Element-wise producers with a q-tensor epilogue (include/ghm.h "q tensors"): the reduced-precision modes keep a bf16 /
// fp16 copy of every tensor a low-precision product reads, in channel-block-of-8 layout, written by the kernel that
// produces the tensor -- not by a conversion pass of its own.  A q unit is the 8 channels of one pixel, so these kernels
// give one thread a channel BLOCK of 8 for one or two pixels: eight coalesced fp32 plane accesses, one 16-byte unit
// store per pixel.  The fp32 result is optional: where every consumer reads the q copy it is never written.
// (BatchNormLayer / BilinearUpsample2DLayer / MaxPool2DLayer backward / the parity interleave of the collapsed
// up-sample convolutions: architectures/dcgan.py:17-31,42-47, architectures/p2p.py:146-268, architectures/layers.py:13-26.)
#include "common.h"

typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
typedef float f32x2q __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2q __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2q __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned q_pack2(float a, float b, int dt) {
    const f32x2q v = {a, b};
    return dt == GHM_DTYPE_BF16 ? __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2q))
                                : __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2q));
}
__device__ __forceinline__ u32x4q q_pack8(const float* v, int dt) {
    u32x4q w;
    w.x = q_pack2(v[0], v[1], dt);
    w.y = q_pack2(v[2], v[3], dt);
    w.z = q_pack2(v[4], v[5], dt);
    w.w = q_pack2(v[6], v[7], dt);
    return w;
}

// dt == 3 (host side 'bf16x3', the split-fp32 mode of csrc/conv_split.hip): the fp32 value as three bf16 pieces, in three
// planes ``ps`` units apart (a producer writes whole tensors: ps = samples x sample stride of the q tensor)
__device__ __forceinline__ void q_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    f32x2q v = {a, b};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2q));
    f32x2q r = {a - __uint_as_float(p0 << 16), b - __uint_as_float(p0 & 0xffff0000u)};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2q));
    f32x2q r2 = {r[0] - __uint_as_float(p1 << 16), r[1] - __uint_as_float(p1 & 0xffff0000u)};
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2q));
}
// dt == 4 ('bf16x2'): the first two pieces only (x1 = bf16(x - x0), rounded to nearest)
__device__ __forceinline__ void q_store8(u32x4q* o, const float* v, int dt, long ps) {
    if (dt == 3 || dt == 4) {
        unsigned a[4], b[4], c[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) q_split2(v[2 * t], v[2 * t + 1], a[t], b[t], c[t]);
        o[0] = u32x4q{a[0], a[1], a[2], a[3]};
        o[ps] = u32x4q{b[0], b[1], b[2], b[3]};
        if (dt == 3) o[2 * ps] = u32x4q{c[0], c[1], c[2], c[3]};
    } else {
        *o = q_pack8(v, dt);
    }
}

// decode idx -> (n, channel block, item) with ``items`` items per (n, block)
__device__ __forceinline__ bool q_decode(long idx, int N, int C8, long items, int& n, int& cb, long& it) {
    if (idx >= (long)N * C8 * items) return false;
    it = idx % items;
    const long nc = idx / items;
    cb = (int)(nc % C8);
    n = (int)(nc / C8);
    return true;
}

// y = act((x - mean) * (gamma * inv) + beta): thread = 8 channels x PX consecutive pixels.  PX = 1 (the default since round 5):
// the 64 lanes of a wave write 64 consecutive q units, one kilobyte per store instruction; with two pixels per thread a
// store instruction wrote 16 bytes of every 32 -- the q planes are most of what these passes move, and a pass that writes
// partial segments pays for it (the pooling-mask backward: 0.62 -> 0.38 ms for the same bytes)
template <int PX>
__global__ __launch_bounds__(256) void bn_apply_q_kernel(const float* __restrict__ x, long xs, float* __restrict__ y, long ys,
                                                         int N, int C8, int HW, const float* __restrict__ mean,
                                                         const float* __restrict__ inv, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int act, float alpha,
                                                         u32x4q* __restrict__ q, long qns, int dt) {
    int n, cb;
    long pp;
    if (!q_decode((long)blockIdx.x * 256 + threadIdx.x, N, C8, HW / PX, n, cb, pp)) return;
    float v[PX][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const float sc = gamma[c] * inv[c], m = mean[c], be = beta[c];
        const float* xp = x + (long)n * xs + (long)c * HW + PX * pp;
        float t[PX];
        if constexpr (PX == 2) {
            const float2 t2 = *reinterpret_cast<const float2*>(xp);
            t[0] = t2.x; t[1] = t2.y;
        } else {
            t[0] = xp[0];
        }
#pragma unroll
        for (int u = 0; u < PX; ++u) v[u][j] = ghm_act(fmaf(t[u] - m, sc, be), act, alpha);
        if (y) {
            float* yp = y + (long)n * ys + (long)c * HW + PX * pp;
            if constexpr (PX == 2) *reinterpret_cast<float2*>(yp) = make_float2(v[0][j], v[1][j]);
            else yp[0] = v[0][j];
        }
    }
    const long ps = (long)N * qns;
    u32x4q* o = q + (long)n * qns + (long)cb * HW + PX * pp;
#pragma unroll
    for (int u = 0; u < PX; ++u) q_store8(o + u, v[u], dt, ps);
}

// dx = gamma * inv * (dout * act'(y) - mean(dz) - xhat * mean(dz * xhat)): thread = 8 channels x PX pixels
template <int PX>
__global__ __launch_bounds__(256) void bn_bwd_apply_q_kernel(const float* __restrict__ dout, long ds, const float* __restrict__ y,
                                                             long ys, const float* __restrict__ x, long xs, float* __restrict__ dx,
                                                             long dxs, int N, int C8, int HW, const float* __restrict__ mean,
                                                             const float* __restrict__ inv, const float* __restrict__ gamma,
                                                             const float* __restrict__ sums, float inv_count, int act,
                                                             float alpha, u32x4q* __restrict__ q, long qns, int dt,
                                                             const float* __restrict__ beta) {
    int n, cb;
    long pp;
    if (!q_decode((long)blockIdx.x * 256 + threadIdx.x, N, C8, HW / PX, n, cb, pp)) return;
    float v[PX][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const float m = mean[c], iv = inv[c], g = gamma[c] * iv;
        const float mb = sums[2 * c] * inv_count, mg = sums[2 * c + 1] * inv_count;
        const long o = (long)c * HW + PX * pp;
        float d[PX], xx[PX], yy[PX];
        if constexpr (PX == 2) {
            const float2 d2 = *reinterpret_cast<const float2*>(dout + (long)n * ds + o);
            const float2 x2 = *reinterpret_cast<const float2*>(x + (long)n * xs + o);
            d[0] = d2.x; d[1] = d2.y; xx[0] = x2.x; xx[1] = x2.y;
        } else {
            d[0] = dout[(long)n * ds + o];
            xx[0] = x[(long)n * xs + o];
        }
        // y == nullptr: recomputed from x with the forward pass's own expression (elementwise.hip bn_y)
        if (y) {
            if constexpr (PX == 2) {
                const float2 y2 = *reinterpret_cast<const float2*>(y + (long)n * ys + o);
                yy[0] = y2.x; yy[1] = y2.y;
            } else {
                yy[0] = y[(long)n * ys + o];
            }
        } else {
            const float be = beta[c];
#pragma unroll
            for (int u = 0; u < PX; ++u) yy[u] = ghm_act(fmaf(xx[u] - m, g, be), act, alpha);
        }
#pragma unroll
        for (int u = 0; u < PX; ++u) v[u][j] = g * (d[u] * ghm_dact_from_out(yy[u], act, alpha) - mb - (xx[u] - m) * iv * mg);
        if (dx) {
            if constexpr (PX == 2) *reinterpret_cast<float2*>(dx + (long)n * dxs + o) = make_float2(v[0][j], v[1][j]);
            else dx[(long)n * dxs + o] = v[0][j];
        }
    }
    const long ps = (long)N * qns;
    u32x4q* o = q + (long)n * qns + (long)cb * HW + PX * pp;
#pragma unroll
    for (int u = 0; u < PX; ++u) q_store8(o + u, v[u], dt, ps);
}

// Theano bilinear 2x (layers.py:13-26; elementwise.hip up_bilinear_fwd_kernel): thread = 8 channels x one coarse pixel,
// writes its 2x2 fine pixels
__global__ __launch_bounds__(256) void up_bilinear_fwd_q_kernel(const float* __restrict__ x, long xs, float* __restrict__ y,
                                                                int N, int C8, int H, int W, u32x4q* __restrict__ q, long qns,
                                                                int dt) {
    int n, cb;
    long px;
    const long hw = (long)H * W;
    if (!q_decode((long)blockIdx.x * 256 + threadIdx.x, N, C8, hw, n, cb, px)) return;
    const int i = (int)(px / W), j = (int)(px - (long)i * W);
    const int i1 = min(i + 1, H - 1), j1 = min(j + 1, W - 1);
    float a00[8], a01[8], a10[8], a11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long c = cb * 8 + k;
        const float* xp = x + (long)n * xs + c * hw;
        const float v00 = xp[i * W + j], v01 = xp[i * W + j1], v10 = xp[i1 * W + j], v11 = xp[i1 * W + j1];
        const float c0 = 0.5f * (v00 + v10), c1 = 0.5f * (v01 + v11);
        a00[k] = v00;
        a01[k] = 0.5f * (v00 + v01);
        a10[k] = c0;
        a11[k] = 0.5f * (c0 + c1);
        if (y) {
            float* o = y + ((long)n * C8 * 8 + c) * 4 * hw + (long)(2 * i) * (2 * W) + 2 * j;
            *reinterpret_cast<float2*>(o) = make_float2(a00[k], a01[k]);
            *reinterpret_cast<float2*>(o + 2 * W) = make_float2(a10[k], a11[k]);
        }
    }
    const long ps = (long)N * qns;
    u32x4q* o = q + (long)n * qns + (long)cb * 4 * hw + (long)(2 * i) * (2 * W) + 2 * j;
    q_store8(o, a00, dt, ps);
    q_store8(o + 1, a01, dt, ps);
    q_store8(o + 2 * W, a10, dt, ps);
    q_store8(o + 2 * W + 1, a11, dt, ps);
}

// pp [4N, K, H, W] (parity-planar output of a collapsed up-sample convolution) -> hi [N, K, 2H, 2W]
__global__ __launch_bounds__(256) void pp_to_hi_q_kernel(const float* __restrict__ pp, float* __restrict__ hi, long hi_nstride,
                                                         int N, int K8, int H, int W, u32x4q* __restrict__ q, long qns, int dt) {
    int n, kb;
    long px;
    const long hw = (long)H * W;
    if (!q_decode((long)blockIdx.x * 256 + threadIdx.x, N, K8, hw, n, kb, px)) return;
    const int yy = (int)(px / W), xx = (int)(px - (long)yy * W);
    const long plane = (long)K8 * 8 * hw;
    float a[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float* src = pp + ((long)n * 4) * plane + ((long)(kb * 8 + k) * H + yy) * W + xx;
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p][k] = src[p * plane];
        if (hi) {
            float* dst = hi + (long)n * hi_nstride + ((long)(kb * 8 + k) * 2 * H + 2 * yy) * (2 * W) + 2 * xx;
            *reinterpret_cast<float2*>(dst) = make_float2(a[0][k], a[1][k]);
            *reinterpret_cast<float2*>(dst + 2 * W) = make_float2(a[2][k], a[3][k]);
        }
    }
    const long ps = (long)N * qns;
    u32x4q* o = q + (long)n * qns + (long)kb * 4 * hw + (long)(2 * yy) * (2 * W) + 2 * xx;
    q_store8(o, a[0], dt, ps);
    q_store8(o + 1, a[1], dt, ps);
    q_store8(o + 2 * W, a[2], dt, ps);
    q_store8(o + 2 * W + 1, a[3], dt, ps);
}

// ---- BatchNorm of a collapsed up-sample convolution fused with the parity interleave (dcgan.default_generator,
// architectures/dcgan.py:22-31: Upscale2D -> conv -> BatchNorm -> rectify; DESIGN section 4 "collapsed").  The conv's
// output x is parity-planar [4N, K, H, W]; the next layer wants hi [N, K, 2H, 2W].  Forward: normalise + activation and
// the interleave in ONE pass (the parity-planar result is never written); backward: dout is read straight from the hi
// layout through the inverse permutation (no hi_to_pp pass), y is recomputed from x.  Thread = 8 channels x one low-res
// pixel = its 2x2 block.  pp sample 4n + p holds output parity p = 2*dy + dx.
__global__ __launch_bounds__(256) void bn_apply_hi_kernel(const float* __restrict__ x, float* __restrict__ hi, long hi_nstride,
                                                          int N, int K8, int H, int W, const float* __restrict__ mean,
                                                          const float* __restrict__ inv, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int act, float alpha,
                                                          u32x4q* __restrict__ q, long qns, int dt) {
    int n, kb;
    long px;
    const long hw = (long)H * W;
    if (!q_decode((long)blockIdx.x * 256 + threadIdx.x, N, K8, hw, n, kb, px)) return;
    const int yy = (int)(px / W), xx = (int)(px - (long)yy * W);
    const long plane = (long)K8 * 8 * hw;
    float a[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = kb * 8 + k;
        const float sc = gamma[c] * inv[c], m = mean[c], be = beta[c];
        const float* src = x + ((long)n * 4) * plane + ((long)c * H + yy) * W + xx;
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p][k] = ghm_act(fmaf(src[p * plane] - m, sc, be), act, alpha);
        if (hi) {
            float* dst = hi + (long)n * hi_nstride + ((long)c * 2 * H + 2 * yy) * (2 * W) + 2 * xx;
            *reinterpret_cast<float2*>(dst) = make_float2(a[0][k], a[1][k]);
            *reinterpret_cast<float2*>(dst + 2 * W) = make_float2(a[2][k], a[3][k]);
        }
    }
    if (q) {
        const long ps = (long)N * qns;
    u32x4q* o = q + (long)n * qns + (long)kb * 4 * hw + (long)(2 * yy) * (2 * W) + 2 * xx;
        q_store8(o, a[0], dt, ps);
        q_store8(o + 1, a[1], dt, ps);
        q_store8(o + 2 * W, a[2], dt, ps);
        q_store8(o + 2 * W + 1, a[3], dt, ps);
    }
}

// grid (S, K): partial sums of dz and dz * xhat over the channel's N * H * W low-res pixels x 4 parities (fp64)
__global__ __launch_bounds__(256) void bn_bwd_hi_partial(const float* __restrict__ dhi, long dhi_nstride, const float* __restrict__ x,
                                                         int N, int K, int H, int W, int S, const float* __restrict__ mean,
                                                         const float* __restrict__ inv, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int act, float alpha,
                                                         double* __restrict__ ws, int max_split) {
    const int c = blockIdx.y, s = blockIdx.x;
    const long hw = (long)H * W, total = (long)N * hw, plane = (long)K * hw;
    const long chunk = (total + S - 1) / S;
    const long lo = s * chunk, hi_e = min(lo + chunk, total);
    const float m = mean[c], iv = inv[c], sc = gamma[c] * iv, be = beta[c];
    double sa = 0.0, sb = 0.0;
    for (long e = lo + threadIdx.x; e < hi_e; e += 256) {
        const long n = e / hw, i = e - n * hw;
        const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
        const float* dp = dhi + n * dhi_nstride + ((long)c * 2 * H + 2 * yy) * (2 * W) + 2 * xx;
        const float2 d0 = *reinterpret_cast<const float2*>(dp), d1 = *reinterpret_cast<const float2*>(dp + 2 * W);
        const float d[4] = {d0.x, d0.y, d1.x, d1.y};
        const float* xp = x + (n * 4) * plane + (long)c * hw + i;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float xv = xp[p * plane];
            const float dz = d[p] * ghm_dact_from_out(ghm_act(fmaf(xv - m, sc, be), act, alpha), act, alpha);
            sa += dz;
            sb += (double)dz * ((xv - m) * iv);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_down(sa, off, 64);
        sb += __shfl_down(sb, off, 64);
    }
    __shared__ double ra[4], rb[4];
    if ((threadIdx.x & 63) == 0) {
        ra[threadIdx.x >> 6] = sa;
        rb[threadIdx.x >> 6] = sb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[((long)c * max_split + s) * 2 + 0] = (ra[0] + ra[1]) + (ra[2] + ra[3]);
        ws[((long)c * max_split + s) * 2 + 1] = (rb[0] + rb[1]) + (rb[2] + rb[3]);
    }
}

// dx (parity-planar, fp32 and / or q) from dout in hi layout: thread = 8 channels x one low-res pixel
__global__ __launch_bounds__(256) void bn_bwd_hi_apply(const float* __restrict__ dhi, long dhi_nstride, const float* __restrict__ x,
                                                       float* __restrict__ dx, int N, int K8, int H, int W,
                                                       const float* __restrict__ mean, const float* __restrict__ inv,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ sums, float inv_count, int act, float alpha,
                                                       u32x4q* __restrict__ q, int dt) {
    int n, kb;
    long px;
    const long hw = (long)H * W;
    if (!q_decode((long)blockIdx.x * 256 + threadIdx.x, N, K8, hw, n, kb, px)) return;
    const int yy = (int)(px / W), xx = (int)(px - (long)yy * W);
    const long plane = (long)K8 * 8 * hw;
    float a[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = kb * 8 + k;
        const float m = mean[c], iv = inv[c], g = gamma[c] * iv, be = beta[c];
        const float mb = sums[2 * c] * inv_count, mg = sums[2 * c + 1] * inv_count;
        const float* dp = dhi + (long)n * dhi_nstride + ((long)c * 2 * H + 2 * yy) * (2 * W) + 2 * xx;
        const float2 d0 = *reinterpret_cast<const float2*>(dp), d1 = *reinterpret_cast<const float2*>(dp + 2 * W);
        const float d[4] = {d0.x, d0.y, d1.x, d1.y};
        const long off = ((long)n * 4) * plane + (long)c * hw + px;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float xv = x[off + p * plane];
            const float dz = d[p] * ghm_dact_from_out(ghm_act(fmaf(xv - m, g, be), act, alpha), act, alpha);
            a[p][k] = g * (dz - mb - (xv - m) * iv * mg);
            if (dx) dx[off + p * plane] = a[p][k];
        }
    }
    if (q) {        // the parity-planar tensor seen as [N, 4K, H, W]: sample n, channel block p * K8 + kb
#pragma unroll
        for (int p = 0; p < 4; ++p) q_store8(q + ((long)n * 4 + p) * K8 * hw + (long)kb * hw + px, a[p], dt, (long)N * 4 * K8 * hw);
    }
}

// gradient of the (never materialised) full-resolution conv output behind a fused conv + act + 2x2 max-pool
// (elementwise.hip maxpool2_mask_bwd_kernel): thread = 8 channels x TWO neighbouring pooled pixels -> a 2 x 4 window per
// channel (16-byte fp32 rows, 64-byte q rows); with ``part`` the per-channel sums of what it writes (the conv's bias
// gradient) as per-block partials part[c][n * bpp + block]
template <bool BIAS>
__global__ __launch_bounds__(256) void maxpool2_mask_bwd_q_kernel(const unsigned char* __restrict__ mask,
                                                                  const float* __restrict__ y, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, int N, int C8, int H, int W, int act,
                                                                  float alpha, float* __restrict__ part, int bpp,
                                                                  u32x4q* __restrict__ q, long qns, int dt) {
    const int Ho = H / 2, Wo = W / 2, Wo2 = W / 4;
    const long hwp = (long)Ho * Wo, hw = (long)H * W, items = (long)Ho * Wo2;
    // blocks never straddle a (sample, channel block): bpp blocks of 256 pooled-pixel pairs each
    const int blk = blockIdx.x % bpp;
    const long ncb = blockIdx.x / bpp;
    const int cb = (int)(ncb % C8), n = (int)(ncb / C8);
    const long it = (long)blk * 256 + threadIdx.x;
    const bool live = it < items;
    const int i = live ? (int)(it / Wo2) : 0, j2 = live ? (int)(it - (long)i * Wo2) : 0;
    float r0[4][8], r1[4][8], csum[8];          // [fine column][channel] of the window's two rows
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long pl = (long)n * C8 * 8 + cb * 8 + k;
        const long po = pl * hwp + (long)i * Wo + 2 * j2;
        const unsigned m2 = live ? *reinterpret_cast<const unsigned short*>(mask + po) : 0u;
        const float2 gv = live ? *reinterpret_cast<const float2*>(dy + po) : make_float2(0.f, 0.f);
        const unsigned m0 = m2 & 0xffu, m1 = m2 >> 8;
        float g0, g1;           // y == nullptr: the slope from the mask's sign bit
        if (y) {
            const float2 yv = live ? *reinterpret_cast<const float2*>(y + po) : make_float2(0.f, 0.f);
            g0 = gv.x * ghm_dact_from_out(yv.x, act, alpha);
            g1 = gv.y * ghm_dact_from_out(yv.y, act, alpha);
        } else {
            g0 = gv.x * ghm_dact_from_sign(m0, act, alpha);
            g1 = gv.y * ghm_dact_from_sign(m1, act, alpha);
        }
        r0[0][k] = (m0 & 1u) ? g0 : 0.f; r0[1][k] = (m0 & 2u) ? g0 : 0.f; r0[2][k] = (m1 & 1u) ? g1 : 0.f; r0[3][k] = (m1 & 2u) ? g1 : 0.f;
        r1[0][k] = (m0 & 4u) ? g0 : 0.f; r1[1][k] = (m0 & 8u) ? g0 : 0.f; r1[2][k] = (m1 & 4u) ? g1 : 0.f; r1[3][k] = (m1 & 8u) ? g1 : 0.f;
        csum[k] = g0 * (float)__popc(m0 & 15u) + g1 * (float)__popc(m1 & 15u);
        if (dx && live) {
            float* o = dx + pl * hw + (long)(2 * i) * W + 4 * j2;
            *reinterpret_cast<float4*>(o) = make_float4(r0[0][k], r0[1][k], r0[2][k], r0[3][k]);
            *reinterpret_cast<float4*>(o + W) = make_float4(r1[0][k], r1[1][k], r1[2][k], r1[3][k]);
        }
    }
    if (live) {
        const long ps = (long)N * qns;
    u32x4q* o = q + (long)n * qns + (long)cb * hw + (long)(2 * i) * W + 4 * j2;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            q_store8(o + f, r0[f], dt, ps);
            q_store8(o + W + f, r1[f], dt, ps);
        }
    }
    if constexpr (BIAS) {
        __shared__ float red[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s = csum[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            part[(long)(cb * 8 + k) * ((long)N * bpp) + (long)n * bpp + blk] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        }
    }
}

// The same pass with a thread per FINE COLUMN (8 channels x the two rows of one window column): lanes along x write consecutive
// 16-byte q units -- one kilobyte per store instruction.  The form above has a lane write four consecutive units (64 bytes), i.e.
// 16 bytes per 64-byte segment and instruction over three planes x two rows: with the q tensor 6 bytes per element against
// 1.25 read per element, the store pattern is what the pass costs.  Two neighbouring lanes read the same pooled element.
template <bool BIAS>
__global__ __launch_bounds__(256) void maxpool2_mask_bwd_qc_kernel(const unsigned char* __restrict__ mask,
                                                                   const float* __restrict__ y, const float* __restrict__ dy,
                                                                   float* __restrict__ dx, int N, int C8, int H, int W, int act,
                                                                   float alpha, float* __restrict__ part, int bpp,
                                                                   u32x4q* __restrict__ q, long qns, int dt) {
    const int Ho = H / 2, Wo = W / 2;
    const long hwp = (long)Ho * Wo, hw = (long)H * W, items = (long)Ho * W;
    const int blk = blockIdx.x % bpp;
    const long ncb = blockIdx.x / bpp;
    const int cb = (int)(ncb % C8), n = (int)(ncb / C8);
    const long it = (long)blk * 256 + threadIdx.x;
    const bool live = it < items;
    const int i = live ? (int)(it / W) : 0, x = live ? (int)(it - (long)i * W) : 0;
    const int c = x & 1;
    float r0[8], r1[8], csum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long pl = (long)n * C8 * 8 + cb * 8 + k;
        const long po = pl * hwp + (long)i * Wo + (x >> 1);
        const unsigned m = live ? mask[po] : 0u;
        float g = live ? dy[po] : 0.f;
        g *= y ? ghm_dact_from_out(live ? y[po] : 0.f, act, alpha) : ghm_dact_from_sign(m, act, alpha);
        const bool b0 = (m >> c) & 1u, b1 = (m >> (2 + c)) & 1u;
        r0[k] = b0 ? g : 0.f;
        r1[k] = b1 ? g : 0.f;
        csum[k] = g * (float)((int)b0 + (int)b1);
        if (dx && live) {
            float* o = dx + pl * hw + (long)(2 * i) * W + x;
            o[0] = r0[k];
            o[W] = r1[k];
        }
    }
    if (live) {
        const long ps = (long)N * qns;
        u32x4q* o = q + (long)n * qns + (long)cb * hw + (long)(2 * i) * W + x;
        q_store8(o, r0, dt, ps);
        q_store8(o + W, r1, dt, ps);
    }
    if constexpr (BIAS) {
        __shared__ float red[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s = csum[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            part[(long)(cb * 8 + k) * ((long)N * bpp) + (long)n * bpp + blk] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        }
    }
}

// The same gradient in the operand form of the SPARSE matrix instruction (round 6; sp_wgrad_pooled_kernel, conv_split.hip): the
// max-pool backward leaves at most one non-zero per 2x2 window and row unless two columns of a window row tie, so fine row
// 2 i + r is its HALF-WIDTH row cq[.., 2 i + r, j] = (arg-max of window (i, j) in row r ? g : 0) plus one column bit per
// half-pixel -- two values in every four consecutive pixels, v_smfmac_f32_32x32x32_bf16's 2:4 pattern with the pixels as
// contraction index.  idx[n][cb][y][j / 16] holds the column bits of 16 half-pixels of 8 channels (8 x u16 = one 16-byte unit).
// A tie inside a window row (Theano sends the gradient to both columns) cannot be written this way: flags[n * H + y] is set
// and the weight gradient takes that row from the dense q tensor.  Thread = one window, 8 channels, both rows.
__global__ __launch_bounds__(256) void maxpool2_mask_bwd_compress_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ y,
                                                                         const float* __restrict__ dy, int N, int C8, int H, int W,
                                                                         int act, float alpha, u32x4q* __restrict__ cq, long cqns,
                                                                         u32x4q* __restrict__ idx, int* __restrict__ flags, int dt) {
    const int Ho = H / 2, Wo = W / 2;
    const long hwp = (long)Ho * Wo;
    const int bpp = (int)((hwp + 255) / 256);
    const int blk = blockIdx.x % bpp;
    const long ncb = blockIdx.x / bpp;
    const int cb = (int)(ncb % C8), n = (int)(ncb / C8);
    const long it = (long)blk * 256 + threadIdx.x;
    const bool live = it < hwp;
    const int i = live ? (int)(it / Wo) : 0, j = live ? (int)(it - (long)i * Wo) : 0;
    float v[2][8];
    unsigned ax[2] = {0u, 0u}, tie[2] = {0u, 0u};        // bit k = channel k
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long po = ((long)n * C8 * 8 + cb * 8 + k) * hwp + it;
        const unsigned m = live ? mask[po] : 0u;
        float g = live ? dy[po] : 0.f;
        g *= y ? ghm_dact_from_out(live ? y[po] : 0.f, act, alpha) : ghm_dact_from_sign(m, act, alpha);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned b0 = (m >> (2 * r)) & 1u, b1 = (m >> (2 * r + 1)) & 1u;
            v[r][k] = (b0 | b1) ? g : 0.f;
            ax[r] |= (b1 & ~b0 & 1u) << k;
            tie[r] |= (b0 & b1) << k;
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (live) q_store8(cq + (long)n * cqns + ((long)cb * H + 2 * i + r) * Wo + j, v[r], dt, (long)N * cqns);
        unsigned w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = (unsigned)((__ballot((ax[r] >> k) & 1u) >> (lane & 48)) & 0xffffull);
        if (live && (lane & 15) == 0)
            idx[(((long)n * C8 + cb) * H + 2 * i + r) * (Wo / 16) + j / 16] =
                u32x4q{w[0] | (w[1] << 16), w[2] | (w[3] << 16), w[4] | (w[5] << 16), w[6] | (w[7] << 16)};
        if (live && tie[r]) flags[(long)n * H + 2 * i + r] = 1;      // (every writer stores the same value: no atomic needed)
    }
}

// out[c] (+)= fixed-order sum of part[c][0 .. S)
__global__ __launch_bounds__(64) void q_rows_sum_kernel(const float* __restrict__ part, int C, int S, float* __restrict__ out,
                                                        int accumulate) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < S; i += 64) s += part[(long)c * S + i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + s;
}

inline bool q_dtype_ok(int dt) { return dt == GHM_DTYPE_BF16 || dt == GHM_DTYPE_F16 || dt == 3 || dt == 4; }

}  // namespace

#define EWQ_GRID(total) dim3(ceil_div((long)(total), 256)), dim3(256), 0, ctx->stream

extern "C" {

int ghm_bn_apply_q(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int64_t ys, int32_t N, int32_t C, int32_t HW,
                   const float* mean, const float* inv, const float* gamma, const float* beta, int32_t act, float alpha,
                   void* yq, int64_t yq_nstride, int32_t dtype) {
    GHM_CHECK(q_dtype_ok(dtype) && yq && C % 8 == 0 && HW % 2 == 0 && xs % 2 == 0 && ys % 2 == 0 &&
              (((uintptr_t)x | (uintptr_t)y) & 7) == 0 && ((uintptr_t)yq & 15) == 0,
              "ghm_bn_apply_q: bf16 / f16, C %% 8 == 0, even HW and strides, aligned tensors");
    if (GHM_OPT("GHM_Q_TWO_PIXELS"))         // two pixels per thread (the form before round 5)
        hipLaunchKernelGGL(bn_apply_q_kernel<2>, EWQ_GRID((long)N * (C / 8) * (HW / 2)), x, (long)xs, y, (long)ys, N, C / 8, HW, mean,
                           inv, gamma, beta, act, alpha, (u32x4q*)yq, (long)yq_nstride, dtype);
    else
        hipLaunchKernelGGL(bn_apply_q_kernel<1>, EWQ_GRID((long)N * (C / 8) * HW), x, (long)xs, y, (long)ys, N, C / 8, HW, mean,
                           inv, gamma, beta, act, alpha, (u32x4q*)yq, (long)yq_nstride, dtype);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_bn_backward_q(ghm_ctx* ctx, const float* dout, int64_t ds, const float* y, int64_t ys, const float* x, int64_t xs,
                      float* dx, int64_t dxs, int32_t N, int32_t C, int32_t HW, const float* mean, const float* inv,
                      const float* gamma, const float* beta, float* dgamma, float* dbeta, int32_t act, float alpha,
                      int32_t accumulate, void* ws, void* dxq, int64_t dxq_nstride, int32_t dtype) {
    GHM_CHECK(y != nullptr || beta != nullptr, "ghm_bn_backward_q: y == NULL needs beta (y is then recomputed from x)");
    GHM_CHECK(q_dtype_ok(dtype) && dxq && C % 8 == 0 && HW % 2 == 0 && ds % 2 == 0 && ys % 2 == 0 && xs % 2 == 0 &&
              dxs % 2 == 0 && (((uintptr_t)dout | (uintptr_t)y | (uintptr_t)x | (uintptr_t)dx) & 7) == 0 &&
              ((uintptr_t)dxq & 15) == 0,
              "ghm_bn_backward_q: bf16 / f16, C %% 8 == 0, even HW and strides, aligned tensors");
    // the parameter gradients and the two per-channel means: the reduction passes of ghm_bn_backward (they need no dx)
    if (int e = ghm_bn_backward_sums(ctx, dout, ds, y, ys, x, xs, N, C, HW, mean, inv, dgamma, dbeta, act, alpha, accumulate, ws,
                                     gamma, beta))
        return e;
    const float* sums = (const float*)((const char*)ws + ghm_bn_workspace(C) - (size_t)C * 2 * sizeof(float));
    if (GHM_OPT("GHM_Q_TWO_PIXELS"))
        hipLaunchKernelGGL(bn_bwd_apply_q_kernel<2>, EWQ_GRID((long)N * (C / 8) * (HW / 2)), dout, (long)ds, y, (long)ys, x, (long)xs,
                           dx, (long)dxs, N, C / 8, HW, mean, inv, gamma, sums, 1.f / (float)((long)N * HW), act, alpha,
                           (u32x4q*)dxq, (long)dxq_nstride, dtype, beta);
    else
        hipLaunchKernelGGL(bn_bwd_apply_q_kernel<1>, EWQ_GRID((long)N * (C / 8) * HW), dout, (long)ds, y, (long)ys, x, (long)xs,
                           dx, (long)dxs, N, C / 8, HW, mean, inv, gamma, sums, 1.f / (float)((long)N * HW), act, alpha,
                           (u32x4q*)dxq, (long)dxq_nstride, dtype, beta);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_bn_apply_hi(ghm_ctx* ctx, const float* x_pp, float* hi, int64_t hi_nstride, int32_t N, int32_t K, int32_t H, int32_t W,
                    const float* mean, const float* inv, const float* gamma, const float* beta, int32_t act, float alpha,
                    void* hiq, int64_t hiq_nstride, int32_t dtype) {
    GHM_CHECK((hi || hiq) && K % 8 == 0 && hi_nstride % 2 == 0 && ((uintptr_t)hi & 7) == 0 && ((uintptr_t)hiq & 15) == 0 &&
              (!hiq || q_dtype_ok(dtype)), "ghm_bn_apply_hi: K %% 8 == 0, an output, aligned tensors, bf16 / f16 for the q output");
    hipLaunchKernelGGL(bn_apply_hi_kernel, EWQ_GRID((long)N * (K / 8) * H * W), x_pp, hi, (long)hi_nstride, N, K / 8, H, W, mean, inv,
                       gamma, beta, act, alpha, (u32x4q*)hiq, (long)hiq_nstride, dtype);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_bn_backward_hi(ghm_ctx* ctx, const float* dhi, int64_t dhi_nstride, const float* x_pp, float* dx_pp, int32_t N, int32_t K,
                       int32_t H, int32_t W, const float* mean, const float* inv, const float* gamma, const float* beta,
                       float* dgamma, float* dbeta, int32_t act, float alpha, int32_t accumulate, void* ws, void* dxq,
                       int32_t dtype) {
    GHM_CHECK((dx_pp || dxq) && K % 8 == 0 && dhi_nstride % 2 == 0 && ((uintptr_t)dhi & 7) == 0 && ((uintptr_t)dxq & 15) == 0 &&
              (!dxq || q_dtype_ok(dtype)), "ghm_bn_backward_hi: K %% 8 == 0, an output, aligned tensors, bf16 / f16 for the q output");
    const long count = (long)N * H * W;                     // low-res pixels per channel (x 4 parities)
    const int max_split = (int)((ghm_bn_workspace(K) - (size_t)K * 2 * sizeof(float)) / ((size_t)K * 2 * sizeof(double)));
    long S = (2048 + K - 1) / K;                            // ~8 blocks per CU over the K channels
    if (S > count / 512) S = count / 512;
    if (S > max_split) S = max_split;
    if (S < 1) S = 1;
    double* wsd = (double*)ws;
    float* sums = (float*)((char*)ws + ghm_bn_workspace(K) - (size_t)K * 2 * sizeof(float));
    hipLaunchKernelGGL(bn_bwd_hi_partial, dim3((unsigned)S, K), dim3(256), 0, ctx->stream, dhi, (long)dhi_nstride, x_pp, N, K, H, W,
                       (int)S, mean, inv, gamma, beta, act, alpha, wsd, max_split);
    GHM_LAUNCH_CHECK();
    if (int e = ghm_bn_backward_finish(ctx, wsd, K, (int)S, sums, dgamma, dbeta, accumulate)) return e;
    hipLaunchKernelGGL(bn_bwd_hi_apply, EWQ_GRID((long)N * (K / 8) * H * W), dhi, (long)dhi_nstride, x_pp, dx_pp, N, K / 8, H, W, mean,
                       inv, gamma, beta, sums, 1.f / (float)(4 * count), act, alpha, (u32x4q*)dxq, dtype);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_upsample_bilinear2_fwd_q(ghm_ctx* ctx, const float* x, int64_t xs, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                                 void* yq, int64_t yq_nstride, int32_t dtype) {
    GHM_CHECK(q_dtype_ok(dtype) && yq && C % 8 == 0 && ((uintptr_t)y & 7) == 0 && ((uintptr_t)yq & 15) == 0,
              "ghm_upsample_bilinear2_fwd_q: bf16 / f16, C %% 8 == 0, aligned tensors");
    hipLaunchKernelGGL(up_bilinear_fwd_q_kernel, EWQ_GRID((long)N * (C / 8) * H * W), x, (long)xs, y, N, C / 8, H, W,
                       (u32x4q*)yq, (long)yq_nstride, dtype);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_pp_to_hi_q(ghm_ctx* ctx, const float* pp, float* hi, int64_t hi_nstride, int32_t N, int32_t K, int32_t H, int32_t W,
                   void* hiq, int64_t hiq_nstride, int32_t dtype) {
    GHM_CHECK(q_dtype_ok(dtype) && hiq && K % 8 == 0 && hi_nstride % 2 == 0 && ((uintptr_t)hi & 7) == 0 &&
              ((uintptr_t)hiq & 15) == 0, "ghm_pp_to_hi_q: bf16 / f16, K %% 8 == 0, aligned tensors");
    hipLaunchKernelGGL(pp_to_hi_q_kernel, EWQ_GRID((long)N * (K / 8) * H * W), pp, hi, (long)hi_nstride, N, K / 8, H, W,
                       (u32x4q*)hiq, (long)hiq_nstride, dtype);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_maxpool2_mask_bwd_q(ghm_ctx* ctx, const uint8_t* mask, const float* y, const float* dy, float* dx, int32_t N, int32_t C,
                            int32_t H, int32_t W, int32_t act, float alpha, float* dbias, int32_t accumulate, void* dxq,
                            int64_t dxq_nstride, int32_t dtype) {
    GHM_CHECK(q_dtype_ok(dtype) && dxq && C % 8 == 0 && H % 2 == 0 && W % 4 == 0 && ((uintptr_t)dx & 15) == 0 &&
              ((uintptr_t)dxq & 15) == 0, "ghm_maxpool2_mask_bwd_q: bf16 / f16, C %% 8 == 0, even H, W %% 4 == 0, aligned tensors");
    if (!GHM_OPT("GHM_POOLBWD_WINDOWS")) {        // a thread per fine column (see the kernel); the switch restores a thread per window pair
        const long items = (long)(H / 2) * W;
        const int bpp = (int)ceil_div(items, 256);
        const long blocks = (long)N * (C / 8) * bpp;
        if (dbias) {
            const int S = N * bpp;
            void* ws = nullptr;
            if (int e = ghm_scratch(ctx, (size_t)C * S * sizeof(float), &ws)) return e;
            hipLaunchKernelGGL((maxpool2_mask_bwd_qc_kernel<true>), dim3(blocks), dim3(256), 0, ctx->stream, mask, y, dy, dx, N, C / 8,
                               H, W, act, alpha, (float*)ws, bpp, (u32x4q*)dxq, (long)dxq_nstride, dtype);
            GHM_LAUNCH_CHECK();
            hipLaunchKernelGGL(q_rows_sum_kernel, dim3(C), dim3(64), 0, ctx->stream, (const float*)ws, C, S, dbias, accumulate);
        } else {
            hipLaunchKernelGGL((maxpool2_mask_bwd_qc_kernel<false>), dim3(blocks), dim3(256), 0, ctx->stream, mask, y, dy, dx, N,
                               C / 8, H, W, act, alpha, (float*)nullptr, bpp, (u32x4q*)dxq, (long)dxq_nstride, dtype);
        }
        GHM_LAUNCH_CHECK();
        return 0;
    }
    const long items = (long)(H / 2) * (W / 4);
    const int bpp = (int)ceil_div(items, 256);
    const long blocks = (long)N * (C / 8) * bpp;
    if (dbias) {
        const int S = N * bpp;
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, (size_t)C * S * sizeof(float), &ws)) return e;
        hipLaunchKernelGGL((maxpool2_mask_bwd_q_kernel<true>), dim3(blocks), dim3(256), 0, ctx->stream, mask, y, dy, dx, N, C / 8,
                           H, W, act, alpha, (float*)ws, bpp, (u32x4q*)dxq, (long)dxq_nstride, dtype);
        GHM_LAUNCH_CHECK();
        hipLaunchKernelGGL(q_rows_sum_kernel, dim3(C), dim3(64), 0, ctx->stream, (const float*)ws, C, S, dbias, accumulate);
    } else {
        hipLaunchKernelGGL((maxpool2_mask_bwd_q_kernel<false>), dim3(blocks), dim3(256), 0, ctx->stream, mask, y, dy, dx, N,
                           C / 8, H, W, act, alpha, (float*)nullptr, bpp, (u32x4q*)dxq, (long)dxq_nstride, dtype);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

// the pooled gradient in the sparse-instruction operand form (see maxpool2_mask_bwd_compress_kernel): cq [pieces][N][C/8][H][W/2]
// units, idx [N][C/8][H][W/32] units of 8 x u16 column bits, flags [N * H] ints (zeroed here, set where a window row ties)
int ghm_maxpool2_mask_bwd_compress_q(ghm_ctx* ctx, const uint8_t* mask, const float* y, const float* dy, int32_t N, int32_t C, int32_t H,
                                     int32_t W, int32_t act, float alpha, void* cq, int64_t cq_nstride, void* idx, int32_t* flags,
                                     int32_t dtype) {
    GHM_CHECK((dtype == 3 || dtype == 4) && cq && idx && flags && C % 8 == 0 && H % 2 == 0 && W % 32 == 0 && ((uintptr_t)cq & 15) == 0 &&
              ((uintptr_t)idx & 15) == 0, "ghm_maxpool2_mask_bwd_compress_q: split dtypes, C %% 8 == 0, even H, W %% 32 == 0, aligned tensors");
    if (int e = ghm_memset_zero(ctx, flags, (size_t)N * H * sizeof(int32_t))) return e;
    const long hwp = (long)(H / 2) * (W / 2);
    const long blocks = (long)N * (C / 8) * ceil_div(hwp, 256);
    hipLaunchKernelGGL(maxpool2_mask_bwd_compress_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mask, y, dy, N, C / 8, H, W, act, alpha,
                       (u32x4q*)cq, (long)cq_nstride, (u32x4q*)idx, flags, dtype);
    GHM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
