// Backward of the discriminator's first block, Conv2D(1 -> K, 5x5, 'same') -> LeakyRectify -> MaxPool2D(2)
// (architectures/dcgan.py:42-47, d_conv1), WITHOUT the full-resolution gradient tensor.
//
// The forward pass is one kernel (conv_thin.hip fanout_kernel<POOL>): it keeps the pooled activation yp and a 4-bit
// arg-max mask per pooled element.  The gradient of the conv's full-resolution output is then v = gp * act'(yp) at the
// arg-max position(s) of every 2x2 window and ZERO elsewhere.  Rounds 1-3 materialised that tensor (537 MB at 512^2,
// batch 8: written once by maxpool2_mask_bwd, read by the weight gradient, read again by the data gradient) and fed it to
// dense kernels that multiply three zeros for every value.  Here both gradients are computed from the pooled operands:
//
//   weight gradient  dW[k][ta][tb] = sum over pooled elements e of filter k, arg-max pixel (ay, ax):
//                                    v(e) * x[ay + ta - 2][ax + tb - 2]                       (a gather from an LDS tile of x)
//   data gradient    dx[y][x]      = sum over k and the 3x3 pooling cells around (y, x), arg-max (ay, ax) within reach:
//                                    v * W[k][y - ay + 2][x - ax + 2]
//   bias gradient    db[k]         = sum of v (x the number of arg-max positions: ties send the gradient to all of them,
//                                    as Theano's max_pool_2d gradient does and as ghm_maxpool2_mask_bwd does)
//
// VALU kernels: 25 multiply-adds per pooled value instead of 100 per window; fixed summation order (bit-repeatable).
// Algorithmic HBM bytes: gp + yp (4 B each) + the mask (1 B) per pooled element: 302 MB for the weight gradient
// at 512^2, batch 8 (the materialised form moved 1.4 GB for the same result), half of that for the batch-4 data gradient.
#include <cstring>

#include "common.h"

namespace {

constexpr int KS = 5, T = KS * KS, PADW = 2;

struct PoolThinArgs {
    const float* x;                 // [N, 1, H, W]
    long x_nstride;
    const unsigned char* mask;      // [N, K, H/2, W/2] bit b = arg-max at (row b >> 1, column b & 1) of the window
    const float* yp;                // pooled activation
    const float* gp;                // gradient of the pooled activation
    const float* wp;                // packed weights wp[t * K + k] (data gradient)
    float* out;                     // weight gradient: partials [N * bands][K][T + 1]; data gradient: dx [N, 1, H, W]
    long out_nstride;
    int N, K, H, W;
    int act;
    float alpha;
    int accumulate;
    int rb;                         // pooled rows per band (weight gradient)
    int ldp;                        // floats per (row, column parity) plane of the LDS tile, a multiple of 64
    int debug;                      // tuning only (GHM_ABLATE): 1 = no x tile fill, 2 = no operand loads, 4 = no multiply-adds
};

// ---- weight + bias gradient: block = one image x one band of RB pooled rows x NW filters (one per wave) ----
// The x tile is kept DE-INTERLEAVED in LDS: xs[row][column parity][column / 2] with a row stride of a multiple of 64
// floats.  A lane's 5x5 window starts at column 2 * px + bx (bx = the arg-max column bit): its taps of equal column parity
// sit at consecutive indices of one plane, so lane px reads index px + const of a plane and the 64 lanes of a wave hit 64
// different banks whatever their arg-max positions are (interleaved, lanes px and px + 32 share a bank on every read).
// Operand loads run U elements ahead of the multiply-adds (the tie loop keeps the compiler from pipelining them itself).
template <int NW, int U, int KPW>
__global__ __launch_bounds__(NW * 64) void pool_thin_wgrad_kernel(const PoolThinArgs a) {
    extern __shared__ float xs[];                       // [(2 * rb + 4)][2][ldp], the zero padding materialised
    const int Hp = a.H / 2, Wp = a.W / 2;
    const int ldp = a.ldp;
    const int rows = 2 * a.rb + 2 * PADW;
    const int band = blockIdx.x, n = blockIdx.y;
    const int py0 = band * a.rb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xn = a.x + (long)n * a.x_nstride;
    const int half = a.W / 2 + PADW;                     // indices per plane that hold data or padding
    const int fill = rows * 2 * half;
    if (!(a.debug & 1))
    for (int e0 = tid; e0 < fill; e0 += NW * 64 * 8) {   // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * NW * 64;
            const int r = e / (2 * half), c = e - r * (2 * half);        // c = image column + 2
            const int y = 2 * py0 - PADW + r, xx = c - PADW;
            v[u] = (e < fill && (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? xn[(long)y * a.W + xx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * NW * 64;
            const int r = e / (2 * half), c = e - r * (2 * half);
            if (e < fill) xs[(r * 2 + (c & 1)) * ldp + (c >> 1)] = v[u];
        }
    }
    __syncthreads();
    const int total = a.rb * Wp;
#pragma unroll 1
    for (int kq = 0; kq < KPW; ++kq) {
        const int k = (blockIdx.z * KPW + kq) * NW + wave;
        if (k >= a.K) break;
        float acc[T + 1];
#pragma unroll
        for (int t = 0; t <= T; ++t) acc[t] = 0.f;
        const long base = (((long)n * a.K + k) * Hp + py0) * Wp;
        // work items = (pooled row, 64-pixel segment) pairs, walked with wave-uniform counters (no per-lane division)
        const int nseg = (Wp + 63) / 64, items = a.rb * nseg;
        float g[U], gn[U], yn[U];
        unsigned m[U], mn[U];
        auto load = [&](int j0, int row, int seg, float* gg, float* yy, unsigned* mm) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int px = seg * 64 + lane;
                const bool ok = j0 + u < items && px < Wp && !(a.debug & 2);
                const long i = base + (long)row * Wp + px;
                gg[u] = ok ? a.gp[i] : 0.f;
                yy[u] = (ok && a.yp) ? a.yp[i] : 1.f;
                mm[u] = (ok && !(a.debug & 8)) ? (unsigned)a.mask[i] : ((a.debug & 10) ? 1u : 0u);
                if (++seg == nseg) { seg = 0; ++row; }
            }
        };
        int nrow = 0, nsg = 0;                          // position of the batch being prefetched
        auto advance = [&](int& row, int& seg) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (++seg == nseg) { seg = 0; ++row; }
        };
        load(0, nrow, nsg, gn, yn, mn);
        for (int j0 = 0; j0 < items; j0 += U) {
            int row = nrow, seg = nsg;
#pragma unroll
            for (int u = 0; u < U; ++u) {           // yp == nullptr: the slope from the mask's sign bit
                g[u] = gn[u] * (a.yp ? ghm_dact_from_out(yn[u], a.act, a.alpha) : ghm_dact_from_sign(mn[u], a.act, a.alpha));
                m[u] = mn[u] & 15u;
            }
            advance(nrow, nsg);
            load(j0 + U, nrow, nsg, gn, yn, mn);        // the next batch is in flight during this one's multiply-adds
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float v = g[u];
                const int pyl = row, px = seg * 64 + lane;
                if (++seg == nseg) { seg = 0; ++row; }
                unsigned mm = m[u];
                acc[T] += v * (float)__popc(mm);
                if (a.debug & 4) mm = 0;
                while (mm) {                            // one iteration unless the window's maximum is tied
                    const int b = __ffs(mm) - 1, by = b >> 1, bx = b & 1;
                    mm &= mm - 1;
                    const float* pa = xs + ((2 * pyl + by) * 2 + bx) * ldp + px;             // taps tb = 0, 2, 4
                    const float* pb = xs + ((2 * pyl + by) * 2 + (1 - bx)) * ldp + px + bx;  // taps tb = 1, 3
#pragma unroll
                    for (int ta = 0; ta < KS; ++ta) {
                        acc[ta * KS + 0] = fmaf(v, pa[ta * 2 * ldp + 0], acc[ta * KS + 0]);
                        acc[ta * KS + 1] = fmaf(v, pb[ta * 2 * ldp + 0], acc[ta * KS + 1]);
                        acc[ta * KS + 2] = fmaf(v, pa[ta * 2 * ldp + 1], acc[ta * KS + 2]);
                        acc[ta * KS + 3] = fmaf(v, pb[ta * 2 * ldp + 1], acc[ta * KS + 3]);
                        acc[ta * KS + 4] = fmaf(v, pa[ta * 2 * ldp + 2], acc[ta * KS + 4]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t <= T; ++t) {
            float s = acc[t];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            acc[t] = s;
        }
        if (lane == 0) {
            float* o = a.out + (((long)n * gridDim.x + band) * a.K + k) * (T + 1);
#pragma unroll
            for (int t = 0; t <= T; ++t) o[t] = acc[t];
        }
    }
}

// dwp[t * K + k] (+)= sum_s part[s][k][t], db[k] (+)= sum_s part[s][k][T]; one wave per output, fixed order (lane l sums
// slices l, l + 64, ..., then a butterfly) -- one THREAD per output walked S = 256 strided loads in a row: 60 us of latency
__global__ __launch_bounds__(64) void pool_thin_wgrad_final(const float* __restrict__ part, int S, int K, float* __restrict__ dwp,
                                                            float* __restrict__ dbias, int accumulate) {
    const int idx = blockIdx.x, lane = threadIdx.x;
    const int k = idx / (T + 1), t = idx - k * (T + 1);
    const float* p = part + (long)k * (T + 1) + t;
    const long stride = (long)K * (T + 1);
    float v = 0.f;
    for (int s = lane; s < S; s += 64) v += p[(long)s * stride];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) {
        if (t < T) {
            float* o = dwp + (long)t * K + k;
            *o = (accumulate ? *o : 0.f) + v;
        } else if (dbias) {
            dbias[k] = (accumulate ? dbias[k] : 0.f) + v;
        }
    }
}

// ---- data gradient: block = 8 x 32 pooling cells (16 x 64 pixels) of one image, thread = one cell = its 2 x 2 pixels ----
// Per filter a thread visits the 3 x 3 cells around its own; a cell's arg-max at (ry, rx) relative to the thread's top-left
// pixel feeds the thread's four pixels through the 2 x 2 filter block at taps (2 - ry, 2 - rx): the filters are kept in
// LDS with a zero border (7 rows x 8), so the block is always readable and no tap test is needed.
constexpr int DCH = 8, DCW = 32, KC = 8;                 // cell tile, filters staged per round
constexpr int WLR = 8, WLK = 7 * WLR;                    // padded filter: row stride, floats per filter
__global__ __launch_bounds__(DCH * DCW) void pool_thin_dgrad_kernel(const PoolThinArgs a) {
    __shared__ float Wl[64 * WLK];                       // (K <= 64) Wl[k][ta + 1][tb + 1], zero border
    __shared__ float vs[KC][DCH + 2][DCW + 2];
    __shared__ unsigned char ms[KC][DCH + 2][DCW + 2];
    const int Hp = a.H / 2, Wp = a.W / 2;
    const int tid = threadIdx.x, lx = tid % DCW, ly = tid / DCW;
    const int cx0 = blockIdx.x * DCW, cy0 = blockIdx.y * DCH, n = blockIdx.z;
    for (int e = tid; e < a.K * WLK; e += DCH * DCW) {
        const int k = e / WLK, rem = e - k * WLK, r = rem / WLR - 1, c = rem % WLR - 1;
        Wl[e] = ((unsigned)r < (unsigned)KS && (unsigned)c < (unsigned)KS) ? a.wp[(long)(r * KS + c) * a.K + k] : 0.f;
    }
    float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
    constexpr int CELLS = (DCH + 2) * (DCW + 2);
    constexpr int NST = (KC * CELLS + DCH * DCW - 1) / (DCH * DCW);     // staged elements per thread and round
    float sv[NST];
    unsigned char sm[NST];
    // a round's operands go global -> registers (all loads of a thread in flight at once) -> LDS; the next round's loads
    // are issued before this round's arithmetic
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int e = tid + q * DCH * DCW;
            const int kk = e / CELLS, rem = e - kk * CELLS, r = rem / (DCW + 2), c = rem - r * (DCW + 2);
            const int cy = cy0 - 1 + r, cx = cx0 - 1 + c, k = k0 + kk;
            float g = 0.f;
            unsigned char m = 0;
            if (e < KC * CELLS && (unsigned)cy < (unsigned)Hp && (unsigned)cx < (unsigned)Wp && k < a.K) {
                const long i = (((long)n * a.K + k) * Hp + cy) * Wp + cx;
                const unsigned mb = a.mask[i];
                g = a.gp[i] * (a.yp ? ghm_dact_from_out(a.yp[i], a.act, a.alpha) : ghm_dact_from_sign(mb, a.act, a.alpha));
                m = mb & 15u;
            }
            sv[q] = g;
            sm[q] = m;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int e = tid + q * DCH * DCW;
            if (e < KC * CELLS) {
                (&vs[0][0][0])[e] = sv[q];
                (&ms[0][0][0])[e] = sm[q];
            }
        }
        __syncthreads();
        if (k0 + KC < a.K) fetch(k0 + KC);
        const int kn = min(KC, a.K - k0);
#pragma unroll 2
        for (int kk = 0; kk < kn; ++kk) {
            const float* wk = Wl + (k0 + kk) * WLK;
#pragma unroll
            for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    const float v = vs[kk][ly + 1 + dr][lx + 1 + dc];
                    unsigned m = ms[kk][ly + 1 + dr][lx + 1 + dc];
                    // arg-max bit b at (ry, rx) = (2 dr + by, 2 dc + bx); pixel (0, 0) takes tap (2 - ry, 2 - rx), stored at
                    // padded position (3 - ry, 3 - rx): a compile-time corner minus (by, bx)
                    const float* wc = wk + (3 - 2 * dr) * WLR + (3 - 2 * dc);
                    while (m) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        const float* w = wc - ((b >> 1) * WLR + (b & 1));
                        o00 = fmaf(v, w[0], o00);
                        o01 = fmaf(v, w[1], o01);
                        o10 = fmaf(v, w[WLR], o10);
                        o11 = fmaf(v, w[WLR + 1], o11);
                    }
                }
        }
    }
    const int cy = cy0 + ly, cx = cx0 + lx;
    if (cy < Hp && cx < Wp) {
        float* o = a.out + (long)n * a.out_nstride + (long)(2 * cy) * a.W + 2 * cx;
        float2 r0 = make_float2(o00, o01), r1 = make_float2(o10, o11);
        if (a.accumulate) {
            const float2 p0 = *reinterpret_cast<const float2*>(o), p1 = *reinterpret_cast<const float2*>(o + a.W);
            r0.x += p0.x; r0.y += p0.y; r1.x += p1.x; r1.y += p1.y;
        }
        *reinterpret_cast<float2*>(o) = r0;
        *reinterpret_cast<float2*>(o + a.W) = r1;
    }
}

bool geometry_ok(const ghm_conv_desc* d) {
    return d->C == 1 && d->kh == KS && d->kw == KS && d->stride == 1 && d->pad == PADW && d->Ho == d->H && d->Wo == d->W &&
           d->H % 2 == 0 && d->W % 2 == 0 && d->K >= 1;
}

int wgrad_ldp(const ghm_conv_desc* d) { return (d->W / 2 + PADW + 63) / 64 * 64; }

int wgrad_rb(const ghm_conv_desc* d) {          // pooled rows per band: the x tile of a block stays near 50 KB (3 blocks per CU)
    const int Hp = d->H / 2;
    for (int rb = 8; rb >= 1; rb >>= 1)
        if (Hp % rb == 0 && (size_t)(2 * rb + 2 * PADW) * 2 * wgrad_ldp(d) * sizeof(float) <= 52 * 1024) return rb;
    return 0;
}

}  // namespace

extern "C" {

// which of the two gradients of a fused conv + activation + 2x2 max-pool layer can be computed from the pooled operands
// (bit 0: weight + bias gradient, bit 1: data gradient); 0 = neither (materialise with ghm_maxpool2_mask_bwd)
int ghm_conv2d_pool_bwd_sparse_supported(const ghm_conv_desc* d, int32_t act) {
    if (GHM_OPT("GHM_NO_POOL_SPARSE_BWD") || !geometry_ok(d)) return 0;
    if (!(act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU)) return 0;     // what the fused forward serves
    int r = 0;
    if (wgrad_rb(d) > 0) r |= 1;
    if (d->K <= 64 && ((d->x_nstride & 1) == 0)) r |= 2;
    return r;
}

int ghm_conv2d_pool_wgrad_sparse_workspace(const ghm_conv_desc* d, size_t* bytes) {
    const int rb = wgrad_rb(d);
    *bytes = rb ? (size_t)d->N * (d->H / 2 / rb) * d->K * (T + 1) * sizeof(float) : 0;
    return 0;
}

int ghm_conv2d_pool_wgrad_sparse(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const uint8_t* mask, const float* yp,
                                 const float* gp, float* dwp, float* dbias, int32_t act, float alpha, int32_t accumulate,
                                 void* workspace) {
    GHM_CHECK(ghm_conv2d_pool_bwd_sparse_supported(d, act) & 1, "ghm_conv2d_pool_wgrad_sparse: geometry / activation not served");
    GHM_CHECK(workspace != nullptr, "ghm_conv2d_pool_wgrad_sparse needs its workspace (ghm_conv2d_pool_wgrad_sparse_workspace)");
    if (d->N == 0) return 0;
    PoolThinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.x_nstride = d->x_nstride; a.mask = mask; a.yp = yp; a.gp = gp;
    a.out = (float*)workspace;
    a.N = d->N; a.K = d->K; a.H = d->H; a.W = d->W; a.act = act; a.alpha = alpha;
    a.rb = wgrad_rb(d);
    a.ldp = wgrad_ldp(d);
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    const int bands = d->H / 2 / a.rb;
    constexpr int NW = 8;
    int U = 4, KPW = 2;                                  // operand batches in flight per wave; NW * KPW filters share one x tile
    if (const char* f = GHM_OPT("GHM_POOLW_U")) U = atoi(f);
    if (const char* f = GHM_OPT("GHM_POOLW_KPW")) KPW = atoi(f);
    const size_t lds = (size_t)(2 * a.rb + 2 * PADW) * 2 * a.ldp * sizeof(float);
    const dim3 grid(bands, d->N, (d->K + NW * KPW - 1) / (NW * KPW));
#define GHM_POOLW_CASE(U_, KPW_)                                                                                              \
    if (U == U_ && KPW == KPW_)                                                                                               \
        hipLaunchKernelGGL((pool_thin_wgrad_kernel<NW, U_, KPW_>), grid, dim3(NW * 64), lds, ctx->stream, a);                 \
    else
    GHM_POOLW_CASE(4, 1) GHM_POOLW_CASE(4, 2) GHM_POOLW_CASE(4, 4) GHM_POOLW_CASE(8, 1) GHM_POOLW_CASE(8, 2) GHM_POOLW_CASE(8, 4)
    GHM_POOLW_CASE(2, 2) {
        ghm_set_error("ghm_conv2d_pool_wgrad_sparse: no variant U=%d KPW=%d", U, KPW);
        return -3;
    }
#undef GHM_POOLW_CASE
    GHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(pool_thin_wgrad_final, dim3(d->K * (T + 1)), dim3(64), 0, ctx->stream,
                       (const float*)workspace, d->N * bands, d->K, dwp, dbias, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_conv2d_pool_dgrad_sparse(ghm_ctx* ctx, const ghm_conv_desc* d, const uint8_t* mask, const float* yp, const float* gp,
                                 const float* wp, float* dx, int32_t act, float alpha, int32_t accumulate) {
    GHM_CHECK(ghm_conv2d_pool_bwd_sparse_supported(d, act) & 2, "ghm_conv2d_pool_dgrad_sparse: geometry / activation not served");
    GHM_CHECK(((uintptr_t)dx & 7) == 0, "ghm_conv2d_pool_dgrad_sparse: dx must be 8-byte aligned");
    if (d->N == 0) return 0;
    PoolThinArgs a;
    memset(&a, 0, sizeof(a));
    a.mask = mask; a.yp = yp; a.gp = gp; a.wp = wp; a.out = dx; a.out_nstride = d->x_nstride;
    a.N = d->N; a.K = d->K; a.H = d->H; a.W = d->W; a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    const int Hp = d->H / 2, Wp = d->W / 2;
    hipLaunchKernelGGL(pool_thin_dgrad_kernel, dim3(ceil_div(Wp, DCW), ceil_div(Hp, DCH), d->N), dim3(DCH * DCW), 0, ctx->stream, a);
    GHM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
