// BilinearUpsample2DLayer(2) -> 3x3 'same' convolution (every decoder stage of p2p.g_unet(bilinear_upsample=True),
// architectures/p2p.py:204-267 with architectures/layers.py:13-26) WITHOUT the up-sampled tensor.
//
// Theano's ratio-2 up-sampling is separable; per axis (n coarse samples, fine index i):
//     u[2m] = x[m],   u[2m+1] = (x[m] + x[min(m+1, n-1)]) / 2                               
// and the convolution pads u with one ring of zeros.  Write U for that operator including the ring (fine index -1 .. 2n) and
// U0 for the "natural" operator that treats x as zero outside [0, n):  u0[2m] = x[m], u0[2m+1] = (x[m] + x[m+1]) / 2 for every
// m in [-1, n].  Then  U = U0 + D,  where D x is non-zero in exactly two places:  (D x)[-1] = -x[0] / 2,  (D x)[2n-1] = +x[n-1] / 2,
// and in two dimensions
//     U x U^T  =  U0 x U0^T  +  F,        F = (D x) U^T + U0 (x D^T)
// -- F lives on two fine rows (-1 and 2 n1 - 1) and two fine columns (-1 and 2 n2 - 1): the "frame".
//   * conv3x3(U0 x U0^T) is FOUR convolutions of the coarse, zero-padded x, one per output parity (p, q), whose taps are sums of
//     the fine taps (ghm_upconv_collapse_batched, mode 1): even rows see coarse offsets (-1, 0, +1) with (w-/2, w-/2 + w0 + w+/2,
//     w+/2), odd rows offsets (0, +1) with (w- + w0/2, w0/2 + w+): 9 / 6 / 6 / 4 non-zero taps = 25 products per 2x2 output
//     block instead of 36.  They run as ONE packed 3x3 convolution with 4K filters ordered (parity, k) on the ordinary kernels,
//     exactly like the nearest-neighbour form of the DCGAN generator (elementwise.hip), output parity-planar [4N, K, n1, n2].
//   * conv3x3(F) touches the fine output rows 0, 2 n1 - 2, 2 n1 - 1 and columns 0, 2 n2 - 2, 2 n2 - 1 only: six "line" products
//     (one filter row or column against one frame line), each a small GEMM.  This file holds them, forward and both gradients.
// Everything here is exact fp32 arithmetic on v_mfma_f32_32x32x2_f32 (the frame is ~1-5 % of the layer's products).
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// frame lines FL[n][line][c][LP], array index = fine position + 1 (position -1 .. 2n):
//   line 0 = fine row -1        : U  (-x[0, :] / 2)      along columns
//   line 1 = fine row 2 n1 - 1  : U  (+x[n1-1, :] / 2)
//   line 2 = fine column -1     : U0 (-x[:, 0] / 2)      along rows
//   line 3 = fine column 2 n2-1 : U0 (+x[:, n2-1] / 2)
// jobs (an output line <- one filter row / column against one frame line); packed correlation taps (a, b), offsets a-1, b-1:
//   job 0: out row 0         <- taps (0, t) on line 0      job 3: out column 0        <- taps (t, 0) on line 2
//   job 1: out row 2 n1 - 2  <- taps (2, t) on line 1      job 4: out column 2 n2 - 2 <- taps (t, 2) on line 3
//   job 2: out row 2 n1 - 1  <- taps (1, t) on line 1      job 5: out column 2 n2 - 1 <- taps (t, 1) on line 3
// out[k, pos] = sum over c, t of w[k, c, tap(job, t)] * line[c, pos + t - 1]   (array index pos + t)
__device__ __forceinline__ int bl_job_line(int job) { return job == 0 ? 0 : (job < 3 ? 1 : (job == 3 ? 2 : 3)); }
__device__ __forceinline__ int bl_job_tap(int job, int t) {
    const int fixed = (job == 0 || job == 3) ? 0 : ((job == 1 || job == 4) ? 2 : 1);
    return job < 3 ? fixed * 3 + t : t * 3 + fixed;
}

struct BlGeo {
    int N, C, K, n1, n2, LP;      // LP: padded line length (multiple of 32, >= 2 max(n1, n2) + 4)
};

// ---- frame lines from the coarse tensor -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bl_lines_kernel(const float* __restrict__ x, long xs, BlGeo g, float* __restrict__ FL,
                                                       float* __restrict__ FLt) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)g.N * 4 * g.C * g.LP;
    if (idx >= total) return;
    const int ai = (int)(idx % g.LP);
    long t = idx / g.LP;
    const int c = (int)(t % g.C);
    t /= g.C;
    const int line = (int)(t % 4), n = (int)(t / 4);
    const int pos = ai - 1;
    const bool rowline = line < 2;
    const int len = rowline ? g.n2 : g.n1;                 // coarse samples along the line
    const float* base = x + (long)n * xs + (long)c * g.n1 * g.n2;
    auto at = [&](int m) -> float {                        // the coarse vector the line up-samples, zero outside
        if (m < 0 || m >= len) return 0.f;
        if (line == 0) return -0.5f * base[m];
        if (line == 1) return 0.5f * base[(long)(g.n1 - 1) * g.n2 + m];
        if (line == 2) return -0.5f * base[(long)m * g.n2];
        return 0.5f * base[(long)m * g.n2 + g.n2 - 1];
    };
    float v = 0.f;
    if (pos >= -1 && pos <= 2 * len) {
        if (rowline) {                                     // U: Theano's operator, zero ring
            if (pos >= 0 && pos < 2 * len) {
                const int m = pos >> 1;
                v = (pos & 1) ? 0.5f * (at(m) + at(min(m + 1, len - 1))) : at(m);
            }
        } else {                                           // U0: natural operator, x zero outside
            const int m = (pos + 2) / 2 - 1;               // floor(pos / 2) for pos >= -1
            v = (pos & 1) ? 0.5f * (at(m) + at(m + 1)) : at(m);
        }
    }
    FL[idx] = v;
    FLt[(((long)n * 4 + line) * g.LP + ai) * g.C + c] = v;         // [n][line][ai][c]: the weight gradient's operand (lanes along c)
}

// ---- the three line GEMMs: D[i][j] = sum_kk A[i][kk] B[kk][j] on v_mfma_f32_32x32x2_f32 -----------------------------------
// block = 4 waves, one 32 x 32 output tile; the contraction range of the block (blockIdx.z picks the split) is dealt to the
// waves in pairs, the four partial tiles meet in LDS in fixed order
__device__ __forceinline__ f32x16 bl_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void bl_reduce_store(f32x16 acc, float* dst, long row_stride, long col_stride, float* smem, int rows_ok,
                                                int cols_ok, bool accumulate = false) {
    // dst[i * row_stride + j * col_stride] = sum over the four waves; lane: j = lane & 31, rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) smem[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * 33 + j] = acc[e];
    __syncthreads();
    for (int o = tid; o < 1024; o += 256) {
        const int i = o >> 5, jj = o & 31;
        const float v = ((smem[i * 33 + jj] + smem[(32 + i) * 33 + jj]) + smem[(64 + i) * 33 + jj]) + smem[(96 + i) * 33 + jj];
        if (i < rows_ok && jj < cols_ok) {
            float* const q = dst + (long)i * row_stride + (long)jj * col_stride;
            *q = accumulate ? *q + v : v;
        }
    }
}

// forward: DL[split][job][n][k][pos] = sum over (c, t) of wp[c][tap(job, t)][k] * FL[n][line(job)][c][pos + t]
// grid (K / 32, LP / 32, splits * 6 * N)
__global__ __launch_bounds__(256) void bl_gemm_fwd_kernel(const float* __restrict__ wp, const float* __restrict__ FL, BlGeo g,
                                                          int splits, float* __restrict__ DL) {
    __shared__ float smem[4 * 32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kg = lane >> 5;
    int z = blockIdx.z;
    const int n = z % g.N;
    z /= g.N;
    const int job = z % 6, sp = z / 6;
    const int k = blockIdx.x * 32 + li, pos = blockIdx.y * 32 + li;
    // channel pairs of this wave: the block's share [c_lo, c_hi) of C, dealt pairwise to the 4 waves
    const int per = ((g.C / 2 + splits - 1) / splits) * 2;
    const int c_lo = sp * per, c_hi = min(g.C, c_lo + per);
    const float* fl = FL + (((long)n * 4 + bl_job_line(job)) * g.C) * g.LP + pos;
    const int t0 = bl_job_tap(job, 0), t1 = bl_job_tap(job, 1), t2 = bl_job_tap(job, 2);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 4
    for (int c0 = c_lo + 2 * wave; c0 < c_hi; c0 += 8) {
        const int c = c0 + kg;
        const float* w = wp + (long)c * 9 * g.K + k;
        const float* f = fl + (long)c * g.LP;
        const float a0 = w[(long)t0 * g.K], a1 = w[(long)t1 * g.K], a2 = w[(long)t2 * g.K];
        const float b0 = f[0], b1 = f[1], b2 = f[2];
        acc = bl_mfma(a0, b0, acc);
        acc = bl_mfma(a1, b1, acc);
        acc = bl_mfma(a2, b2, acc);
    }
    float* dst = DL + ((((long)sp * 6 + job) * g.N + n) * g.K + blockIdx.x * 32) * g.LP + blockIdx.y * 32;
    bl_reduce_store(acc, dst, g.LP, 1, smem, 32, 32);
}

// data gradient: dFL[split][n][line][c][ai] = sum over the jobs of the line, t, k of wp[c][tap(job, t)][k] * DYL[job][n][k][ai - t + 2]
// (DYL array index = pos + 2: two zero entries in front).  grid (C / 32, LP / 32, splits * 4 * N)
__global__ __launch_bounds__(256) void bl_wt_kernel(const float* __restrict__ wp, int C, int K, float* __restrict__ wT) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // wT[tap][k][c] = wp[c][tap][k]
    if (idx >= (long)9 * K * C) return;
    const int c = (int)(idx % C);
    const long r = idx / C;
    const int k = (int)(r % K), tap = (int)(r / K);
    wT[idx] = wp[((long)c * 9 + tap) * K + k];
}

// (``wp_direct``: wT is the packed layout wp[c][tap][k] itself -- no transposed copy; each lane then walks its own cache lines)
__global__ __launch_bounds__(256) void bl_gemm_dgrad_kernel(const float* __restrict__ wT, const float* __restrict__ DYL, BlGeo g,
                                                            int splits, float* __restrict__ dFL, int wp_direct) {
    __shared__ float smem[4 * 32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kg = lane >> 5;
    int z = blockIdx.z;
    const int n = z % g.N;
    z /= g.N;
    const int line = z % 4, sp = z / 4;
    const int c = blockIdx.x * 32 + li, ai = blockIdx.y * 32 + li;
    const int per = ((g.K / 2 + splits - 1) / splits) * 2;
    const int k_lo = sp * per, k_hi = min(g.K, k_lo + per);
    const int LD = g.LP + 32;                              // DYL rows carry 2 + LP + 30 entries
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int job_lo = line == 0 ? 0 : (line == 1 ? 1 : (line == 2 ? 3 : 4));
    const int njobs = (line == 0 || line == 2) ? 1 : 2;
    for (int jj = 0; jj < njobs; ++jj) {
        const int job = job_lo + jj;
        const float* dy = DYL + (((long)job * g.N + n) * g.K) * LD + ai + 2;
        const long KC = (long)g.K * g.C;
        const long ks = wp_direct ? 1 : g.C;                 // stride of k in the weight operand
        const float* w0 = wp_direct ? wT + ((long)c * 9 + bl_job_tap(job, 0)) * g.K : wT + bl_job_tap(job, 0) * KC + c;
        const float* w1 = wp_direct ? wT + ((long)c * 9 + bl_job_tap(job, 1)) * g.K : wT + bl_job_tap(job, 1) * KC + c;
        const float* w2 = wp_direct ? wT + ((long)c * 9 + bl_job_tap(job, 2)) * g.K : wT + bl_job_tap(job, 2) * KC + c;
#pragma unroll 4
        for (int k0 = k_lo + 2 * wave; k0 < k_hi; k0 += 8) {
            const int k = k0 + kg;
            const float a0 = w0[k * ks], a1 = w1[k * ks], a2 = w2[k * ks];
            const float* d = dy + (long)k * LD;
            const float b0 = d[0], b1 = d[-1], b2 = d[-2];
            acc = bl_mfma(a0, b0, acc);
            acc = bl_mfma(a1, b1, acc);
            acc = bl_mfma(a2, b2, acc);
        }
    }
    float* dst = dFL + ((((long)sp * g.N + n) * 4 + line) * g.C + blockIdx.x * 32) * g.LP + blockIdx.y * 32;
    bl_reduce_store(acc, dst, g.LP, 1, smem, 32, 32);
}

// weight gradient: dWp[split][tap (A, B)][c][k] = sum over n of
//     sum_pos DYL[rowjob(A)][n][k][pos] * FL[n][rowline(A)][c][pos + B]  +  sum_pos DYL[coljob(B)][n][k][pos] * FL[n][colline(B)][c][pos + A]
// grid (K / 32, C / 32, splits * 9); the split deals the samples x position pairs
// (``direct`` != null, one split: the tile is added straight into the fine packed weight gradient dwp[c][tap][k])
__global__ __launch_bounds__(256) void bl_gemm_wgrad_kernel(const float* __restrict__ DYLt, const float* __restrict__ FLt, BlGeo g,
                                                            int splits, float* __restrict__ dWp, float* __restrict__ direct) {
    __shared__ float smem[4 * 32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, kg = lane >> 5;
    const int tap = blockIdx.z % 9, sp = blockIdx.z / 9;
    const int A = tap / 3, B = tap % 3;
    const int k = blockIdx.x * 32 + li, c = blockIdx.y * 32 + li;
    const int LD = g.LP + 32;
    const int rowjob = A == 0 ? 0 : (A == 2 ? 1 : 2), coljob = B == 0 ? 3 : (B == 2 ? 4 : 5);
    const int rowline = A == 0 ? 0 : 1, colline = B == 0 ? 2 : 3;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // contraction: (part, n, position pair); the block's waves take the pairs sp * 4 + wave, + splits * 4, ... of every (part, n)
    for (int part = 0; part < 2; ++part) {
        const int job = part ? coljob : rowjob, line = part ? colline : rowline, sh = part ? A : B;
        const int np2 = part ? g.n1 : g.n2;                 // fine positions 0 .. 2 n - 1 of the output line, in pairs
        for (int n = 0; n < g.N; ++n) {
            const float* ap = DYLt + ((((long)job * g.N + n) * LD + 2 + kg) * g.K + k);
            const float* bp = FLt + ((((long)n * 4 + line) * g.LP + sh + kg) * g.C + c);
#pragma unroll 8
            for (int pp = sp * 4 + wave; pp < np2; pp += splits * 4)
                acc = bl_mfma(ap[(long)(2 * pp) * g.K], bp[(long)(2 * pp) * g.C], acc);
        }
    }
    // D[i = k][j = c] -> dWp[split][tap][c][k], or += into dwp[c][tap][k]
    if (direct) {
        bl_reduce_store(acc, direct + ((long)(blockIdx.y * 32) * 9 + tap) * g.K + blockIdx.x * 32, 1, (long)9 * g.K, smem, 32, 32, true);
        return;
    }
    float* dst = dWp + (((long)sp * 9 + tap) * g.C + blockIdx.y * 32) * g.K + blockIdx.x * 32;
    bl_reduce_store(acc, dst, 1, g.K, smem, 32, 32);
}

// ---- glue: scatter / gather / fold / add --------------------------------------------------------------------------------
// y_pp[n][(p * 2 + q) * K + k][i / 2][j / 2] += the line products that touch fine pixel (i, j); one thread per (n, k, border pixel)
// border pixels of an image: rows {0, 2 n1 - 2, 2 n1 - 1} x all columns (3 * 2 n2), then columns {0, 2 n2 - 2, 2 n2 - 1} x the
// other rows (3 * (2 n1 - 3))
__global__ __launch_bounds__(256) void bl_scatter_fwd_kernel(const float* __restrict__ DL, BlGeo g, int splits, float* __restrict__ y,
                                                             long ys) {
    const int H = 2 * g.n1, W = 2 * g.n2;
    const int nb = 3 * W + 3 * (H - 3);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)g.N * g.K * nb) return;
    const int bp = (int)(idx % nb);
    const long r = idx / nb;
    const int k = (int)(r % g.K), n = (int)(r / g.K);
    int i, j;
    if (bp < 3 * W) {
        const int ri = bp / W;
        j = bp - ri * W;
        i = ri == 0 ? 0 : (ri == 1 ? H - 2 : H - 1);
    } else {
        const int q = bp - 3 * W, ci = q / (H - 3);
        i = 1 + (q - ci * (H - 3));                        // rows 1 .. H - 3
        j = ci == 0 ? 0 : (ci == 1 ? W - 2 : W - 1);
    }
    const int rowjob = i == 0 ? 0 : (i == H - 2 ? 1 : (i == H - 1 ? 2 : -1));
    const int coljob = j == 0 ? 3 : (j == W - 2 ? 4 : (j == W - 1 ? 5 : -1));
    float v = 0.f;
    for (int s = 0; s < splits; ++s) {
        if (rowjob >= 0) v += DL[((((long)s * 6 + rowjob) * g.N + n) * g.K + k) * g.LP + j];
        if (coljob >= 0) v += DL[((((long)s * 6 + coljob) * g.N + n) * g.K + k) * g.LP + i];
    }
    float* p = y + (long)n * ys + ((long)((i & 1) * 2 + (j & 1)) * g.K + k) * g.n1 * g.n2 + (long)(i >> 1) * g.n2 + (j >> 1);
    *p += v;
}

// DYL[job][n][k][2 + pos] = dy (fine, from the parity-planar tensor) on the job's output line; zeros elsewhere in the row
__global__ __launch_bounds__(256) void bl_gather_dy_kernel(const float* __restrict__ dy, long dys, BlGeo g, float* __restrict__ DYL,
                                                           float* __restrict__ DYLt) {
    const int LD = g.LP + 32;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)6 * g.N * g.K * LD) return;
    const int ai = (int)(idx % LD);
    long r = idx / LD;
    const int k = (int)(r % g.K);
    r /= g.K;
    const int n = (int)(r % g.N), job = (int)(r / g.N);
    const int H = 2 * g.n1, W = 2 * g.n2;
    const int pos = ai - 2;
    float v = 0.f;
    if (pos >= 0 && pos < (job < 3 ? W : H)) {
        int i, j;
        if (job < 3) {
            i = job == 0 ? 0 : (job == 1 ? H - 2 : H - 1);
            j = pos;
        } else {
            j = job == 3 ? 0 : (job == 4 ? W - 2 : W - 1);
            i = pos;
        }
        v = dy[(long)n * dys + ((long)((i & 1) * 2 + (j & 1)) * g.K + k) * g.n1 * g.n2 + (long)(i >> 1) * g.n2 + (j >> 1)];
    }
    DYL[idx] = v;
    DYLt[(((long)job * g.N + n) * LD + ai) * g.K + k] = v;          // [job][n][ai][k]: the weight gradient's operand (lanes along k)
}

// dx[n][c][border] += the adjoint of bl_lines_kernel applied to sum over splits of dFL; one thread per (n, c, coarse border pixel)
__global__ __launch_bounds__(256) void bl_fold_kernel(const float* __restrict__ dFL, BlGeo g, int splits, float* __restrict__ dx, long dxs) {
    const int n1 = g.n1, n2 = g.n2;
    const int nb = (n1 >= 2 ? 2 : 1) * n2 + (n2 >= 2 ? 2 : 1) * max(n1 - 2, 0);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)g.N * g.C * nb) return;
    const int bp = (int)(idx % nb);
    const long r = idx / nb;
    const int c = (int)(r % g.C), n = (int)(r / g.C);
    int m1, m2;
    const int nrow = (n1 >= 2 ? 2 : 1) * n2;
    if (bp < nrow) {
        m1 = bp < n2 ? 0 : n1 - 1;
        m2 = bp < n2 ? bp : bp - n2;
    } else {
        const int q = bp - nrow;
        m2 = q < n1 - 2 ? 0 : n2 - 1;
        m1 = 1 + (q < n1 - 2 ? q : q - (n1 - 2));
    }
    const long split_stride = (long)g.N * 4 * g.C * g.LP;
    auto G = [&](int line, int pos) -> float {             // summed over the splits; array index pos + 1
        const float* p = dFL + (((long)n * 4 + line) * g.C + c) * g.LP + pos + 1;
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += p[(long)s * split_stride];
        return v;
    };
    // U^T g [m] over the fine positions 0 .. 2 len - 1 (the ring of U is zero): g[2m] + g[2m+1] / 2 + g[2m-1] / 2 (m >= 1), and the
    // clamped last sample once more: + g[2 len - 1] / 2
    auto UT = [&](int line, int m, int len) -> float {
        float v = G(line, 2 * m) + 0.5f * G(line, 2 * m + 1);
        if (m >= 1) v += 0.5f * G(line, 2 * m - 1);
        if (m == len - 1) v += 0.5f * G(line, 2 * len - 1);
        return v;
    };
    // U0^T g [m] over the fine positions -1 .. 2 len: g[2m] + g[2m+1] / 2 + g[2m-1] / 2 for every m
    auto U0T = [&](int line, int m) -> float { return G(line, 2 * m) + 0.5f * G(line, 2 * m + 1) + 0.5f * G(line, 2 * m - 1); };
    float v = 0.f;
    if (m1 == 0) v += -0.5f * UT(0, m2, n2);
    if (m1 == n1 - 1) v += 0.5f * UT(1, m2, n2);
    if (m2 == 0) v += -0.5f * U0T(2, m1);
    if (m2 == n2 - 1) v += 0.5f * U0T(3, m1);
    float* p = dx + (long)n * dxs + (long)c * n1 * n2 + (long)m1 * n2 + m2;
    *p += v;
}

// dwp[c][tap][k] += sum over splits of dWp[split][tap][c][k]
__global__ __launch_bounds__(256) void bl_wgrad_add_kernel(const float* __restrict__ dWp, int C, int K, int splits, float* __restrict__ dwp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)C * 9 * K) return;
    const int k = (int)(idx % K);
    const long r = idx / K;
    const int tap = (int)(r % 9), c = (int)(r / 9);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += dWp[(((long)s * 9 + tap) * C + c) * K + k];
    dwp[idx] += v;
}

static int bl_lp(int n1, int n2) { return ((2 * (n1 > n2 ? n1 : n2) + 4) + 31) / 32 * 32; }

static bool bl_ok(int N, int C, int K, int n1, int n2) { return N >= 1 && C % 32 == 0 && K % 32 == 0 && n1 >= 2 && n2 >= 2; }

}  // namespace

extern "C" {

// floats of the two per-layer buffers the caller owns: the frame lines (forward -> backward) and the gathered dy lines
int ghm_blconv_frame_sizes(int32_t N, int32_t C, int32_t K, int32_t n1, int32_t n2, int64_t* fl_floats, int64_t* dyl_floats) {
    GHM_CHECK(bl_ok(N, C, K, n1, n2), "ghm_blconv_frame_sizes: C and K must be multiples of 32, maps at least 2 x 2");
    const int LP = bl_lp(n1, n2);
    *fl_floats = 2 * ((int64_t)N * 4 * C * LP + 64);    // [n][line][c][LP] and [n][line][LP][c] (the forward GEMM's last tile reads
                                                        // two entries past a row)
    *dyl_floats = 2 * ((int64_t)6 * N * K * (LP + 32)); // [job][n][k][LD] and [job][n][LD][k]
    return 0;
}

int ghm_blconv_supported(int32_t N, int32_t C, int32_t K, int32_t n1, int32_t n2) { return bl_ok(N, C, K, n1, n2) ? 1 : 0; }

static int bl_splits(long contraction, long tiles, int num_cu) {
    // enough blocks to fill the chip about four times over, at least 64 contraction steps per wave
    int s = (int)((4L * num_cu + tiles - 1) / tiles);
    const int smax = (int)(contraction / 256);
    if (s > smax) s = smax;
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

// y_pp [N, 4K, n1, n2] (parity-planar output of the collapsed convolution, sample stride ``ys``) += conv3x3(frame of x);
// x: the coarse fp32 input [N, C, n1, n2]; wp: the FINE packed 3x3 weights [C][9][K]; FL: ghm_blconv_frame_sizes floats, kept
// for the backward pass
int ghm_blconv_frame_fwd(ghm_ctx* ctx, const float* x, int64_t x_nstride, const float* wp, float* y, int64_t y_nstride, int32_t N,
                         int32_t C, int32_t K, int32_t n1, int32_t n2, float* FL) {
    GHM_CHECK(bl_ok(N, C, K, n1, n2), "ghm_blconv_frame_fwd: geometry not served");
    const BlGeo g{N, C, K, n1, n2, bl_lp(n1, n2)};
    hipLaunchKernelGGL(bl_lines_kernel, dim3(ceil_div((long)N * 4 * C * g.LP, 256)), dim3(256), 0, ctx->stream, x, (long)x_nstride, g, FL,
                       FL + ((long)N * 4 * C * g.LP + 64));
    GHM_LAUNCH_CHECK();
    const int used = (2 * (n1 > n2 ? n1 : n2) + 31) / 32;                    // position tiles that hold output pixels
    const int splits = bl_splits(3L * C, (long)(K / 32) * used * 6 * N, ctx->num_cu);
    void* ws;
    if (ghm_scratch(ctx, (size_t)splits * 6 * N * K * g.LP * 4, &ws)) return -1;
    hipLaunchKernelGGL(bl_gemm_fwd_kernel, dim3(K / 32, used, splits * 6 * N), dim3(256), 0, ctx->stream, wp, (const float*)FL, g, splits,
                       (float*)ws);
    GHM_LAUNCH_CHECK();
    const long nb = 3L * 2 * n2 + 3L * (2 * n1 - 3);
    hipLaunchKernelGGL(bl_scatter_fwd_kernel, dim3(ceil_div((long)N * K * nb, 256)), dim3(256), 0, ctx->stream, (const float*)ws, g, splits,
                       y, (long)y_nstride);
    GHM_LAUNCH_CHECK();
    return 0;
}

// DYL <- the six border lines of dy_pp (the parity-planar gradient of the convolution's output)
int ghm_blconv_frame_gather(ghm_ctx* ctx, const float* dy, int64_t dy_nstride, int32_t N, int32_t C, int32_t K, int32_t n1, int32_t n2,
                            float* DYL) {
    GHM_CHECK(bl_ok(N, C, K, n1, n2), "ghm_blconv_frame_gather: geometry not served");
    const BlGeo g{N, C, K, n1, n2, bl_lp(n1, n2)};
    hipLaunchKernelGGL(bl_gather_dy_kernel, dim3(ceil_div((long)6 * N * K * (g.LP + 32), 256)), dim3(256), 0, ctx->stream, dy,
                       (long)dy_nstride, g, DYL, DYL + (long)6 * N * K * (g.LP + 32));
    GHM_LAUNCH_CHECK();
    return 0;
}

// dx [N, C, n1, n2] += the frame's share of the data gradient (after the collapsed convolution's own data gradient wrote dx)
int ghm_blconv_frame_dgrad(ghm_ctx* ctx, const float* DYL, const float* wp, float* dx, int64_t dx_nstride, int32_t N, int32_t C,
                           int32_t K, int32_t n1, int32_t n2) {
    GHM_CHECK(bl_ok(N, C, K, n1, n2), "ghm_blconv_frame_dgrad: geometry not served");
    const BlGeo g{N, C, K, n1, n2, bl_lp(n1, n2)};
    const int used = (2 * (n1 > n2 ? n1 : n2) + 2 + 31) / 32;                // array indices 0 .. 2 n + 1
    const int splits = bl_splits(6L * K, (long)(C / 32) * used * 4 * N, ctx->num_cu);
    void* ws;
    const size_t part_bytes = (size_t)splits * N * 4 * C * g.LP * 4;
    if (ghm_scratch(ctx, part_bytes + (size_t)9 * K * C * 4, &ws)) return -1;
    float* const wT = (float*)((char*)ws + part_bytes);      // wT[tap][k][c]: the GEMM's lanes run along c
    const int direct = GHM_OPT("GHM_BLCONV_NO_WT") ? 1 : 0;  // (tuning: read the packed weights as they lie, no transposed copy)
    if (!direct) {
        hipLaunchKernelGGL(bl_wt_kernel, dim3(ceil_div((long)9 * K * C, 256)), dim3(256), 0, ctx->stream, wp, C, K, wT);
        GHM_LAUNCH_CHECK();
    }
    // (the fold reads array indices <= 2 n + 1 only: position tiles beyond ``used`` are neither written nor read)
    hipLaunchKernelGGL(bl_gemm_dgrad_kernel, dim3(C / 32, used, splits * 4 * N), dim3(256), 0, ctx->stream,
                       direct ? wp : (const float*)wT, DYL, g, splits, (float*)ws, direct);
    GHM_LAUNCH_CHECK();
    const long nb = (long)(n1 >= 2 ? 2 : 1) * n2 + (long)(n2 >= 2 ? 2 : 1) * (n1 > 2 ? n1 - 2 : 0);
    hipLaunchKernelGGL(bl_fold_kernel, dim3(ceil_div((long)N * C * nb, 256)), dim3(256), 0, ctx->stream, (const float*)ws, g, splits, dx,
                       (long)dx_nstride);
    GHM_LAUNCH_CHECK();
    return 0;
}

// dwp [C][9][K] (the FINE packed weight gradient) += the frame's share (after the expansion of the collapsed gradient wrote dwp)
int ghm_blconv_frame_wgrad(ghm_ctx* ctx, const float* DYL, const float* FL, float* dwp, int32_t N, int32_t C, int32_t K, int32_t n1,
                           int32_t n2) {
    GHM_CHECK(bl_ok(N, C, K, n1, n2), "ghm_blconv_frame_wgrad: geometry not served");
    const BlGeo g{N, C, K, n1, n2, bl_lp(n1, n2)};
    const int npmin = n1 < n2 ? n1 : n2;                    // position pairs per (part, n): every split must own some of them
    int splits = bl_splits((long)2 * N * (n1 + n2), (long)(K / 32) * (C / 32) * 9, ctx->num_cu);
    if (splits > npmin / 4) splits = npmin / 4 > 0 ? npmin / 4 : 1;
    if (splits == 1) {          // enough tiles to fill the chip: one launch, the tiles added straight into dwp
        hipLaunchKernelGGL(bl_gemm_wgrad_kernel, dim3(K / 32, C / 32, 9), dim3(256), 0, ctx->stream,
                           DYL + (long)6 * N * K * (g.LP + 32), FL + ((long)N * 4 * C * g.LP + 64), g, 1, (float*)nullptr, dwp);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    void* ws;
    if (ghm_scratch(ctx, (size_t)splits * 9 * C * K * 4, &ws)) return -1;
    hipLaunchKernelGGL(bl_gemm_wgrad_kernel, dim3(K / 32, C / 32, splits * 9), dim3(256), 0, ctx->stream,
                       DYL + (long)6 * N * K * (g.LP + 32), FL + ((long)N * 4 * C * g.LP + 64), g, splits, (float*)ws, (float*)nullptr);
    GHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bl_wgrad_add_kernel, dim3(ceil_div((long)C * 9 * K, 256)), dim3(256), 0, ctx->stream, (const float*)ws, C, K, splits,
                       dwp);
    GHM_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
