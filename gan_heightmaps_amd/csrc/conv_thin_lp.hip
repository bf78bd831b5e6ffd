// First-layer forward convolutions (<= 4 input channels, 64 filters) on the bf16 / fp16 matrix cores, for the
// reduced-precision modes (BASELINE configs 4 / 5).  architectures/dcgan.py:42-47 (d_conv1: Conv2DLayer(1 -> 64, 5x5) ->
// LeakyRectify(0.2) -> MaxPool2DLayer(2)) is the case built here.
//
// The fp32 kernel that serves these layers in every mode (conv_thin.hip, fanout_kernel) spends 44 us of a 115 us launch in 13
// fp32 k-steps of v_mfma_f32_32x32x2 per pixel tile and 37 us in its epilogue.  Here:
//   * the 25 taps are TWO k-steps of v_mfma_f32_32x32x16: a lane gathers the 8 taps of its pixel from the fp32 rows staged
//     in LDS and packs them (v_cvt_pk): the input is rounded at its (only) consumer -- the "rounded once" rule of
//     oracle/lp.py for a tensor that no kernel produces -- and the weights in the prologue; the bias is added in fp32;
//   * a lane owns one POOLING WINDOW per channel row (four accumulator tiles = the four window positions, lanes along pooled
//     columns): the maximum and the arg-max mask (ties of the ACTIVATED values, as the unfused sequence) are register work
//     without a cross-lane exchange;
//   * the q copy leaves as 8-byte half units (512 contiguous bytes per wave instruction), the mask bytes through an LDS
//     transpose as 16-byte stores; the pooled fp32 tensor is optional (the engine drops it when every consumer reads q).
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int DT>
struct Tl;
template <>
struct Tl<GHM_DTYPE_BF16> {
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <>
struct Tl<GHM_DTYPE_F16> {
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int TP_PRB = 4;           // pooled rows per block (8 convolution rows)
constexpr int TP_MAXQ = 32;        // two k-steps of 16

struct ThinPoolArgs {
    const float* in;
    const float* wp;        // packed fp32 weights wp[(c * T + tap) * K + k]
    const float* bias;      // or null
    float* pool_out;        // [N, 64, H/2, W/2] fp32 or null
    unsigned char* mask;    // [N, 64, H/2, W/2] bytes
    uint2* out_q;           // q tensor of the pooled result (half units) or null
    long out_q_nstride;     // units between samples
    long in_nstride;
    int N, C, H, W;         // input = convolution output grid ('same', stride 1)
    int k, pad, Q;          // Q = C * k * k reduction entries
    int LW, IR;             // staged row length (W + k - 1, padded to a multiple of 4) and rows (2 * TP_PRB + k - 1)
    int act;
    float alpha;
    int off[TP_MAXQ];       // LDS offset of reduction entry q relative to the pixel's window origin (0 for the dead entries >= Q)
};

template <int DT, int NK>
__global__ __launch_bounds__(256, 2) void thin_pool_lp_kernel(const ThinPoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tp_lds[];
    const int rowsz = a.C * a.IR * a.LW;
    float* const rows = tp_lds;                       // [C][IR][LW], zero outside the image
    unsigned char* const scr = reinterpret_cast<unsigned char*>(tp_lds + rowsz + 64);       // [4 waves][32 filters][32 px] mask bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, li = lane & 31;
    const int n = blockIdx.y, pr0 = blockIdx.x * TP_PRB;
    const int Hp = a.H / 2, Wp = a.W / 2;
    const long HWp = (long)Hp * Wp;

    // ---- stage the input rows of this band (2 * TP_PRB + k - 1 rows per channel) with the zero border materialised ----
    {
        const float* img = a.in + (long)n * a.in_nstride;
        const int y0 = 2 * pr0 - a.pad;
        for (int e = tid; e < rowsz; e += 256) {
            const int c = e / (a.IR * a.LW), r2 = e - c * (a.IR * a.LW);
            const int r = r2 / a.LW, col = r2 - r * a.LW;
            const int y = y0 + r, x = col - a.pad;
            rows[e] = ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) ? img[((long)c * a.H + y) * a.W + x] : 0.f;
        }
    }
    // ---- weights: the A fragments of this wave's 32-filter row block (waves 0/2: filters 0..31, waves 1/3: 32..63), rounded
    //      here (dead reduction entries: zero).  A wave per row block keeps the four window accumulators at 64 registers; the
    //      pixel gather is then done by both waves of a pair, which costs LDS reads the kernel has to spare ----
    const int rb = wave & 1;
    u32x4 A[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int q = 16 * j + 8 * kg + t;
            w[t] = q < a.Q ? a.wp[(long)q * 64 + rb * 32 + li] : 0.f;
        }
        A[j].x = Tl<DT>::pack2(w[0], w[1]);
        A[j].y = Tl<DT>::pack2(w[2], w[3]);
        A[j].z = Tl<DT>::pack2(w[4], w[5]);
        A[j].w = Tl<DT>::pack2(w[6], w[7]);
    }
    int off[NK][8];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) off[j][t] = a.off[16 * j + 8 * kg + t];
    float bz[16];                   // bias in the accumulator layout: element e = filter rb*32 + (e&3) + 8*(e>>2) + 4*kg
#pragma unroll
    for (int e = 0; e < 16; ++e) bz[e] = a.bias ? a.bias[rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg] : 0.f;
    __syncthreads();

    const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
    const int tiles = Wp / 32;
    unsigned char* const myscr = scr + wave * (32 * 32);
    for (int item = wave >> 1; item < TP_PRB * tiles; item += 2) {
        const int prl = item / tiles, t0 = (item - prl * tiles) * 32;
        const int pr = pr0 + prl;
        // this lane's pooling window: pooled pixel (pr, t0 + li) = convolution pixels (2 pr + dy, 2 (t0 + li) + dx)
        const int base = (2 * prl) * a.LW + 2 * (t0 + li);
        f32x16 acc[4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[w][e] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int wb = base + (w >> 1) * a.LW + (w & 1);
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = rows[wb + off[j][t]];
                u32x4 B;
                B.x = Tl<DT>::pack2(v[0], v[1]);
                B.y = Tl<DT>::pack2(v[2], v[3]);
                B.z = Tl<DT>::pack2(v[4], v[5]);
                B.w = Tl<DT>::pack2(v[6], v[7]);
                acc[w] = Tl<DT>::mfma(A[j], B, acc[w]);
            }
        }
        // ---- bias, activation, pool, mask, outputs ----
        const long pix = (long)pr * Wp + t0 + li;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float pv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = 4 * g + t;
                float v0 = acc[0][e] + bz[e], v1 = acc[1][e] + bz[e], v2 = acc[2][e] + bz[e], v3 = acc[3][e] + bz[e];
                v0 = v0 > 0.f ? v0 : slope * v0;
                v1 = v1 > 0.f ? v1 : slope * v1;
                v2 = v2 > 0.f ? v2 : slope * v2;
                v3 = v3 > 0.f ? v3 : slope * v3;
                const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                const unsigned mk = (v0 == m ? 1u : 0u) | (v1 == m ? 2u : 0u) | (v2 == m ? 4u : 0u) | (v3 == m ? 8u : 0u) |
                                    (m > 0.f ? GHM_POOL_SIGN : 0u);
                pv[t] = m;
                const int chl = t + 8 * g + 4 * kg;       // filter within the row block
                myscr[chl * 32 + li] = (unsigned char)mk;
                if (a.pool_out) a.pool_out[((long)n * 64 + rb * 32 + chl) * HWp + pix] = m;
            }
            if (a.out_q)
                a.out_q[2 * ((long)n * a.out_q_nstride + (long)(rb * 4 + g) * HWp + pix) + kg] =
                    make_uint2(Tl<DT>::pack2(pv[0], pv[1]), Tl<DT>::pack2(pv[2], pv[3]));
        }
        // mask bytes: lane = (filter, half row): its 16 pooled pixels of this item as one 16-byte store (the same wave wrote
        // them: only the wave's own LDS traffic has to land)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const uint4 m0 = *reinterpret_cast<const uint4*>(myscr + li * 32 + kg * 16);
            *reinterpret_cast<uint4*>(a.mask + ((long)n * 64 + rb * 32 + li) * HWp + (long)pr * Wp + t0 + kg * 16) = m0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

}  // namespace

// ---- library-internal interface (common.h) ----
bool thin_pool_lp_ok(const ghm_conv_desc* d, int act, float alpha, int dtype) {
    if (GHM_OPT("GHM_NO_THIN_LP") || (dtype != GHM_DTYPE_BF16 && dtype != GHM_DTYPE_F16)) return false;
    if (!(act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU)) return false;
    if (d->K != 64 || d->stride != 1 || d->kh != d->kw || d->Ho != d->H || d->Wo != d->W || 2 * d->pad != d->kh - 1) return false;
    const int Q = d->C * d->kh * d->kw;
    if (d->C > 4 || Q > TP_MAXQ || (d->W % 64) || (d->H % 2) || ((d->H / 2) % TP_PRB)) return false;
    const int LW = (d->W + d->kh - 1 + 3) & ~3, IR = 2 * TP_PRB + d->kh - 1;
    return (size_t)(d->C * IR * LW + 64) * 4 + 4 * 32 * 32 <= 64 * 1024;
}

int thin_pool_lp(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias, float* pooled,
                 unsigned char* mask, int act, float alpha, void* yq, long yq_nstride, int dtype) {
    GHM_CHECK(thin_pool_lp_ok(d, act, alpha, dtype), "thin_pool_lp: not served");
    GHM_CHECK((((uintptr_t)mask | (uintptr_t)yq) & 15) == 0, "thin_pool_lp: 16-byte aligned mask / q tensor");
    ThinPoolArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x; a.wp = wp; a.bias = bias; a.pool_out = pooled; a.mask = mask; a.out_q = (uint2*)yq; a.out_q_nstride = yq_nstride;
    a.in_nstride = d->x_nstride;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.k = d->kh; a.pad = d->pad; a.Q = d->C * d->kh * d->kw;
    a.LW = (d->W + d->kh - 1 + 3) & ~3;
    a.IR = 2 * TP_PRB + d->kh - 1;
    a.act = act; a.alpha = alpha;
    const int T = d->kh * d->kw, rowsz = a.C * a.IR * a.LW;
    for (int q = 0; q < a.Q; ++q) {
        const int c = q / T, t = q - c * T, ta = t / d->kw, tb = t - ta * d->kw;
        a.off[q] = (c * a.IR + ta) * a.LW + tb;
    }
    const int nk = (a.Q + 15) / 16;
    const size_t lds = (size_t)(rowsz + 64) * 4 + 4 * 32 * 32;
    const dim3 grid((d->H / 2) / TP_PRB, d->N);
#define GHM_TP_CASE(DT_, NK_)                                                                          \
    if (dtype == DT_ && nk == NK_) {                                                                   \
        hipLaunchKernelGGL((thin_pool_lp_kernel<DT_, NK_>), grid, dim3(256), lds, ctx->stream, a);     \
        GHM_LAUNCH_CHECK();                                                                            \
        return 0;                                                                                      \
    }
    GHM_TP_CASE(GHM_DTYPE_BF16, 1) GHM_TP_CASE(GHM_DTYPE_BF16, 2)
    GHM_TP_CASE(GHM_DTYPE_F16, 1) GHM_TP_CASE(GHM_DTYPE_F16, 2)
#undef GHM_TP_CASE
    ghm_set_error("thin_pool_lp: %d reduction entries not served", a.Q);
    return -3;
}

// ---- C ABI (include/ghm.h) ----
extern "C" int ghm_thin_pool_lp_served(const ghm_conv_desc* d, int32_t act, float alpha, int32_t dtype) {
    return d && d->K % 8 == 0 && !GHM_OPT("GHM_NO_THIN_Q") && thin_pool_lp_ok(d, act, alpha, dtype) ? 1 : 0;
}
