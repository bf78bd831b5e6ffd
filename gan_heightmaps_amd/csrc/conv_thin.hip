// "Thin" convolutions of the train step: one side of the layer has <= 4 channels and the maps are large
// (the first / last layers of all four nets at 512x512 and 256x256).  They carry ~1 % of the FLOPs but every
// one of them streams a 64..128-channel full-resolution tensor, so they are HBM-bound and get their own
// kernels instead of the general implicit-GEMM tiles (which spend their time in prologues and epilogues here):
//
//   fanout_kernel     few -> many channels (d_conv1 / pd_conv1 / conv1 forward, g_out data gradient,
//                     data gradient of the final Deconv2DLayer).  MFMA with the whole weight matrix held in
//                     registers; a loader wave DMAs the thin input rows into LDS (padding materialised) and
//                     four MFMA waves stream 128-pixel groups out with 16-byte stores -- see the kernel.
//   fanin_s2_kernel   many -> few channels through a transposed stride-2 filter (forward of the final
//                     Deconv2DLayer k2 s2, data gradient of pd_conv1 k3 s2): one thread per input-grid
//                     pixel produces the 2x2 output block of every thin channel, so the wide tensor is read
//                     exactly once and all four output parities come out of one launch.  VALU; weights are
//                     broadcast from LDS.
//
// Algorithmic HBM bytes: the wide tensor once (read or written) + the thin tensor once.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define THIN_MAX_Q 36

// ------------------------------------------------------------------------------------------------
// few -> many
//   out[n, row, u, v] = act(bias[row] + sum_q A[row][q] * in[n, ch(q), u*ss + di(q), v*ss + dj(q)])
//   A[row][q] = wp[row*w_rs + (q / T)*w_cs + (q % T)*w_ts]
// ------------------------------------------------------------------------------------------------
struct FanoutArgs {
    const float* in;
    const float* wp;
    const float* bias;
    const float* zeros;       // >= 4 bytes of zeros (ctx->zeros): where padding elements are fetched from
    float* out;
    int N, CS, Hin, Win;      // CS = thin channels
    long in_nstride;
    int R, Hout, Wout;
    long out_nstride;
    int ss, T, Q;             // Q = CS * taps
    long w_rs, w_cs, w_ts;
    int act;
    float alpha;
    int accumulate;
    int ch[THIN_MAX_Q], di[THIN_MAX_Q], dj[THIN_MAX_Q];      // filled by the caller
    // derived by launch_fanout: tap extent, staged-row geometry, per-q LDS and weight offsets
    int dimin, djmin, RPI, KRT, LW;
    int loff[THIN_MAX_Q], widx[THIN_MAX_Q];
};

// LDS hand-over between the loader wave and the MFMA waves: wait for this wave's LDS traffic only.  A
// __syncthreads() would also drain vmcnt, i.e. make every MFMA wave wait for its output stores to be
// acknowledged once per iteration -- the stall this kernel is organised to avoid.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NS> struct StoreVec;
template <> struct StoreVec<4> { typedef float4 type; };
template <> struct StoreVec<2> { typedef float2 type; };

// Block = 5 waves; one iteration = RPI output rows of one image.
//   Wave 4 (the loader) stages the input rows those output rows touch, for all CS thin channels, into LDS with
//   the zero padding materialised: every element is one lane of a global->LDS DMA (global_load_lds_dword:
//   per-lane global address, LDS destination = uniform base + 4*lane; padding lanes read a.zeros), so nothing
//   passes through VGPRs and a whole iteration's input is in flight at once.  Double-buffered, one ahead.
//   Waves 0-3 hold the whole weight matrix as MFMA A fragments and work on groups of NS*32 consecutive
//   pixels: NS accumulator sets with the pixels interleaved (set k, column c <-> pixel NS*c + k), so that a
//   lane owns NS consecutive pixels of every channel row and the epilogue is one NS*4-byte store per lane and
//   row: 32 lanes write NS*128 contiguous bytes of a channel plane.  They never issue a global load, so
//   nothing ever waits on their stores (loads and stores share vmcnt on gfx9); with at most 64 memory
//   instructions outstanding per wave, the wide stores are what keeps enough bytes in flight to reach HBM rate.
template <int KSTEPS, int RB, int NS, bool ACC>
__global__ __launch_bounds__(320, 2) void fanout_kernel(const FanoutArgs a) {
    typedef typename StoreVec<NS>::type vec_t;
    extern __shared__ float lds[];
    float* sbias = lds;                                           // [RB*32]
    int* stab = reinterpret_cast<int*>(lds + RB * 32);            // [2][2*KSTEPS]: LDS offset, weight offset
    float* rows = lds + RB * 32 + 4 * KSTEPS;                     // [2][CS*KRT*LW]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    const int rowsz = a.CS * a.KRT * a.LW;
    if (threadIdx.x < RB * 32) sbias[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;
    if (threadIdx.x < 2 * KSTEPS) {
        const int q = threadIdx.x;
        const bool live = q < a.Q;
        stab[q] = live ? a.loff[q] : 0;                  // dead row: any finite LDS word (its A column is zero)
        stab[2 * KSTEPS + q] = live ? a.widx[q] : -1;
    }
    const int IPI = a.Hout / a.RPI;       // iterations per image
    const int NI = a.N * IPI;
    const int HWin = a.Hin * a.Win;

    auto produce = [&](int r, float* dst) {
        const int n = r / IPI, u0 = (r - n * IPI) * a.RPI;
        const float* img = a.in + (long)n * a.in_nstride;
        for (int ck = 0; ck < a.CS * a.KRT; ++ck) {
            const int c = ck / a.KRT, kr = ck - c * a.KRT;
            const int y = u0 * a.ss + a.dimin + kr;
            const bool rowok = (unsigned)y < (unsigned)a.Hin;
            const float* src = img + (long)c * HWin + (rowok ? y : 0) * a.Win + a.djmin;
            float* d = dst + ck * a.LW;
            for (int x0 = 0; x0 < a.LW; x0 += 64) {          // LW is a multiple of 64
                const int xin = x0 + lane + a.djmin;
                const bool ok = rowok && (unsigned)xin < (unsigned)a.Win;
                const float* g = ok ? src + (x0 + lane) : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(d + x0), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    int r = blockIdx.x;
    if (wv == 4 && r < NI) produce(r, rows);
    __syncthreads();          // tables + first rows (nothing else is in flight yet)

    // MFMA waves: the whole weight matrix as A fragments, lane (l, h) holds A[rb*32 + l][2j + h]
    float A[RB][KSTEPS];
    int off[KSTEPS];
    if (wv < 4) {
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            off[j] = stab[2 * j + h] + NS * l * a.ss;      // this lane's first pixel of a group
            const int wi = stab[2 * KSTEPS + 2 * j + h];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float w = a.wp[wi >= 0 ? (rb * 32 + l) * a.w_rs + wi : 0];      // branch-free: clamp, mask
                A[rb][j] = wi >= 0 ? w : 0.f;
            }
        }
    }
    const int GPR = a.Wout / (NS * 32);       // pixel groups per output row
    const int G = a.RPI * GPR;
    const int HWout = a.Hout * a.Wout;
    // store addressing: wave-uniform 64-bit base (scalar) + 32-bit per-lane byte offset
    const unsigned plane = (unsigned)HWout * 4u;
    const unsigned lane_out = (4u * h * (unsigned)HWout + NS * l) * 4u;
    // linear / relu / leaky relu as one select: v > 0 ? v : slope * v
    const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);

    for (int it = 0; r < NI; ++it, r += gridDim.x) {
        const float* cur = rows + (it & 1) * rowsz;
        if (wv == 4) {
            if (r + (int)gridDim.x < NI) produce(r + gridDim.x, rows + ((it + 1) & 1) * rowsz);
        } else {
            const int n = r / IPI, u0 = (r - n * IPI) * a.RPI;
            for (int g = wv; g < G; g += 4) {
                const int ri = g / GPR, x0 = (g - ri * GPR) * (NS * 32);
                const float* bsrc = cur + (ri * a.LW + x0) * a.ss;
                f32x16 acc[NS][RB];
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[k][rb][e] = 0.f;
#pragma unroll
                for (int j = 0; j < KSTEPS; ++j) {
                    float B[NS];
#pragma unroll
                    for (int k = 0; k < NS; ++k) B[k] = bsrc[off[j] + k * a.ss];
#pragma unroll
                    for (int k = 0; k < NS; ++k)
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
                            acc[k][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][j], B[k], acc[k][rb], 0, 0, 0);
                }
                char* ob = reinterpret_cast<char*>(a.out + (long)n * a.out_nstride + (u0 + ri) * a.Wout + x0);
                unsigned lo = lane_out;
                asm volatile("" : "+v"(lo));      // keep the 32 row offsets out of registers: recompute them per group
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    // bias in the accumulator layout: row = rb*32 + (e&3) + 8*(e>>2) + 4h -> four 16-byte LDS reads
                    float bv[16];
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 t = *reinterpret_cast<const float4*>(sbias + rb * 32 + 8 * e4 + 4 * h);
                        bv[4 * e4] = t.x; bv[4 * e4 + 1] = t.y; bv[4 * e4 + 2] = t.z; bv[4 * e4 + 3] = t.w;
                    }
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        float vals[16];
#pragma unroll
                        for (int e = 0; e < 16; ++e) vals[e] = acc[k][rb][e] + bv[e];
                        if (ACC) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const unsigned row = rb * 32 + (e & 3) + 8 * (e >> 2);
                                vals[e] += reinterpret_cast<const float*>(ob + (lo + row * plane))[k];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[k][rb][e] = vals[e] > 0.f ? vals[e] : slope * vals[e];
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned row = rb * 32 + (e & 3) + 8 * (e >> 2);
                        vec_t v;
                        float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
                        for (int k = 0; k < NS; ++k) vp[k] = acc[k][rb][e];
                        *reinterpret_cast<vec_t*>(ob + (lo + row * plane)) = v;
                    }
                }
            }
        }
        lds_barrier();
    }
}

template <int KSTEPS, int RB, int NS>
static int launch_fanout_t(ghm_ctx* ctx, const FanoutArgs& a) {
    const int NI = a.N * (a.Hout / a.RPI);
    int blocks = ctx->num_cu;              // one 5-wave block per CU (256 VGPRs per wave), NI / blocks iterations each
    if (blocks > NI) blocks = NI;
    const size_t lds = (size_t)(RB * 32 + 4 * KSTEPS + 2 * a.CS * a.KRT * a.LW) * sizeof(float);
    static bool opted_in[2] = {false, false};
    if (!opted_in[a.accumulate ? 1 : 0]) {
        const void* fn = a.accumulate ? (const void*)fanout_kernel<KSTEPS, RB, NS, true>
                                      : (const void*)fanout_kernel<KSTEPS, RB, NS, false>;
        GHM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        opted_in[a.accumulate ? 1 : 0] = true;
    }
    if (a.accumulate)
        hipLaunchKernelGGL((fanout_kernel<KSTEPS, RB, NS, true>), dim3(blocks), dim3(320), lds, ctx->stream, a);
    else
        hipLaunchKernelGGL((fanout_kernel<KSTEPS, RB, NS, false>), dim3(blocks), dim3(320), lds, ctx->stream, a);
    GHM_LAUNCH_CHECK();
    return 0;
}

// geometry shared by the eligibility checks and the launcher: pixels per group, output rows per iteration
static void fanout_plan(int rows_out, int thin_ch, int kh, int kw, int stride, int Hout, int Wout, int* NS, int* RPI,
                        int* KRT, int* LW, size_t* lds) {
    *NS = rows_out <= 64 ? 4 : 2;
    const int gpr = Wout / (*NS * 32);
    int rpi = gpr >= 4 ? 1 : (4 + gpr - 1) / (gpr > 0 ? gpr : 1);      // >= 4 groups per iteration: one per MFMA wave
    while (rpi > 1 && Hout % rpi != 0) --rpi;
    *RPI = rpi;
    *KRT = kh + (rpi - 1) * stride;
    *LW = (((Wout - 1) * stride + kw) + 63) & ~63;      // whole 64-lane DMA units
    *lds = (size_t)2 * thin_ch * *KRT * *LW * sizeof(float) + 2048;
}

static bool fanout_geometry_ok(int rows_out, int thin_ch, int kh, int kw, int stride, int Hout, int Wout) {
    const int q = thin_ch * kh * kw;
    if (!(q <= THIN_MAX_Q && (rows_out == 64 || (rows_out == 128 && q <= 12)))) return false;
    int NS, RPI, KRT, LW;
    size_t lds;
    fanout_plan(rows_out, thin_ch, kh, kw, stride, Hout, Wout, &NS, &RPI, &KRT, &LW, &lds);
    return Wout % (NS * 32) == 0 && lds <= 150 * 1024;
}

static int launch_fanout(ghm_ctx* ctx, FanoutArgs& a, int kh, int kw) {
    int dimax = -(1 << 20), djmax = -(1 << 20);
    a.dimin = a.djmin = 1 << 20;
    for (int q = 0; q < a.Q; ++q) {
        if (a.di[q] < a.dimin) a.dimin = a.di[q];
        if (a.di[q] > dimax) dimax = a.di[q];
        if (a.dj[q] < a.djmin) a.djmin = a.dj[q];
        if (a.dj[q] > djmax) djmax = a.dj[q];
    }
    int NS;
    size_t lds;
    fanout_plan(a.R, a.CS, kh, kw, a.ss, a.Hout, a.Wout, &NS, &a.RPI, &a.KRT, &a.LW, &lds);
    GHM_CHECK(dimax - a.dimin + 1 == kh && djmax - a.djmin + 1 == kw, "fanout: tap extent does not match the filter");
    for (int q = 0; q < a.Q; ++q) {
        const int cs = q / a.T, t = q - cs * a.T;
        a.loff[q] = (a.ch[q] * a.KRT + (a.di[q] - a.dimin)) * a.LW + (a.dj[q] - a.djmin);
        a.widx[q] = (int)(cs * a.w_cs + t * a.w_ts);
    }
    const int ks = (a.Q + 1) / 2;
    if (a.R == 128) {
        if (ks <= 6) return launch_fanout_t<6, 4, 2>(ctx, a);
    } else if (a.R == 64) {
        if (ks <= 5) return launch_fanout_t<5, 2, 4>(ctx, a);
        if (ks <= 13) return launch_fanout_t<13, 2, 4>(ctx, a);
        if (ks <= 18) return launch_fanout_t<18, 2, 4>(ctx, a);
    }
    ghm_set_error("fanout: %d reduction rows x %d output rows unsupported", a.Q, a.R);
    return -3;
}

static bool thin_enabled() { return getenv("GHM_NO_THIN") == nullptr; }
static bool fanout_act_ok(int act) { return act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU; }

// forward conv with <= 4 input channels
bool thin_fanout_fwd_ok(const ghm_conv_desc* d, int act) {
    return thin_enabled() && fanout_act_ok(act) && (long)d->N * d->Ho * d->Wo >= 32768 && (long)d->C * d->H * d->W < (1L << 30) &&
           fanout_geometry_ok(d->K, d->C, d->kh, d->kw, d->stride, d->Ho, d->Wo);
}

int thin_fanout_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                    float* y, int act, float alpha, int accumulate) {
    FanoutArgs a;
    memset(&a, 0, sizeof(a));
    const int T = d->kh * d->kw;
    a.in = x; a.wp = wp; a.bias = bias; a.out = y; a.zeros = ctx->zeros;
    a.N = d->N; a.CS = d->C; a.Hin = d->H; a.Win = d->W; a.in_nstride = d->x_nstride;
    a.R = d->K; a.Hout = d->Ho; a.Wout = d->Wo; a.out_nstride = d->y_nstride;
    a.ss = d->stride; a.T = T; a.Q = d->C * T;
    a.w_rs = 1; a.w_cs = (long)T * d->K; a.w_ts = d->K;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    for (int c = 0; c < d->C; ++c)
        for (int ta = 0; ta < d->kh; ++ta)
            for (int tb = 0; tb < d->kw; ++tb) {
                const int q = c * T + ta * d->kw + tb;
                a.ch[q] = c; a.di[q] = ta - d->pad; a.dj[q] = tb - d->pad;
            }
    return launch_fanout(ctx, a, d->kh, d->kw);
}

// data gradient (stride 1) of a conv with <= 4 filters: dx[C big] <- dy[K small]
bool thin_fanout_dgrad_ok(const ghm_conv_desc* d, int act) {
    return thin_enabled() && fanout_act_ok(act) && d->stride == 1 && d->Ho == d->H && d->Wo == d->W &&
           (long)d->N * d->H * d->W >= 32768 && (long)d->K * d->Ho * d->Wo < (1L << 30) &&
           fanout_geometry_ok(d->C, d->K, d->kh, d->kw, 1, d->H, d->W);
}

int thin_fanout_dgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                      float* dx, int act, float alpha, int accumulate) {
    FanoutArgs a;
    memset(&a, 0, sizeof(a));
    const int T = d->kh * d->kw;
    a.in = dy; a.wp = wp; a.bias = bias; a.out = dx; a.zeros = ctx->zeros;
    a.N = d->N; a.CS = d->K; a.Hin = d->Ho; a.Win = d->Wo; a.in_nstride = d->y_nstride;
    a.R = d->C; a.Hout = d->H; a.Wout = d->W; a.out_nstride = d->x_nstride;
    a.ss = 1; a.T = T; a.Q = d->K * T;
    a.w_rs = (long)T * d->K; a.w_cs = 1; a.w_ts = d->K;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    for (int k = 0; k < d->K; ++k)
        for (int ta = 0; ta < d->kh; ++ta)
            for (int tb = 0; tb < d->kw; ++tb) {
                const int q = k * T + ta * d->kw + tb;
                a.ch[q] = k; a.di[q] = d->pad - ta; a.dj[q] = d->pad - tb;
            }
    return launch_fanout(ctx, a, d->kh, d->kw);
}

// ------------------------------------------------------------------------------------------------
// many -> few through a transposed stride-2 filter (H = 2*Ho, W = 2*Wo):
//   dx[n, c, 2i+pu, 2j+pv] = act(bias[c] + sum_k sum_{(a,b): (pu+PAD-a), (pv+PAD-b) even}
//                                 wp[c][a*KS+b][k] * dy[n, k, i + (pu+PAD-a)/2, j + (pv+PAD-b)/2])
// ------------------------------------------------------------------------------------------------
struct FaninS2Args {
    const float* dy;
    const float* wp;
    const float* bias;
    float* dx;
    int N, K, Ho, Wo;
    long y_nstride, x_nstride;
    int act;
    float alpha;
    int accumulate;
};

template <int KS, int PAD, int CS>
__global__ __launch_bounds__(256) void fanin_s2_kernel(const FaninS2Args a) {
    constexpr int T = KS * KS, RW = CS * T, RWP = (RW + 3) & ~3;
    constexpr int DM = (1 + PAD) / 2;        // neighbours i .. i+DM
    static_assert((KS == 3 && PAD == 1) || (KS == 2 && PAD == 0), "parity tables derived for these filters");
    extern __shared__ float sw[];            // [K][RWP]: every thread reads the same row -> LDS broadcast
    for (int i = threadIdx.x; i < a.K * RWP; i += 256) {
        const int k = i / RWP, r = i - k * RWP;
        sw[i] = r < RW ? a.wp[(long)r * a.K + k] : 0.f;
    }
    __syncthreads();
    const int HWo = a.Ho * a.Wo;
    const long P = (long)a.N * HWo;
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = p < P;
    if (!live) p = P - 1;
    const int n = (int)(p / HWo), rem = (int)(p - (long)n * HWo);
    const int i = rem / a.Wo, j = rem - i * a.Wo;
    const float* src = a.dy + (long)n * a.y_nstride + rem;
    bool ok[DM + 1][DM + 1];
    int noff[DM + 1][DM + 1];
#pragma unroll
    for (int di = 0; di <= DM; ++di)
#pragma unroll
        for (int dj = 0; dj <= DM; ++dj) {
            ok[di][dj] = (i + di < a.Ho) && (j + dj < a.Wo);
            noff[di][dj] = ok[di][dj] ? di * a.Wo + dj : 0;
        }
    float acc[CS][2][2];
#pragma unroll
    for (int c = 0; c < CS; ++c)
        acc[c][0][0] = acc[c][0][1] = acc[c][1][0] = acc[c][1][1] = 0.f;

#pragma unroll 4
    for (int k = 0; k < a.K; ++k) {
        float v[DM + 1][DM + 1];
#pragma unroll
        for (int di = 0; di <= DM; ++di)
#pragma unroll
            for (int dj = 0; dj <= DM; ++dj) {
                const float t = src[(long)k * HWo + noff[di][dj]];
                v[di][dj] = ok[di][dj] ? t : 0.f;
            }
        float w[RWP];
        const float4* wr = reinterpret_cast<const float4*>(sw + k * RWP);
#pragma unroll
        for (int q = 0; q < RWP / 4; ++q) {
            const float4 t = wr[q];
            w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
            for (int pu = 0; pu < 2; ++pu)
#pragma unroll
                for (int ta = 0; ta < KS; ++ta) {
                    if ((pu + PAD - ta) & 1) continue;
                    if (pu + PAD - ta < 0) continue;
#pragma unroll
                    for (int pv = 0; pv < 2; ++pv)
#pragma unroll
                        for (int tb = 0; tb < KS; ++tb) {
                            if ((pv + PAD - tb) & 1) continue;
                            if (pv + PAD - tb < 0) continue;
                            acc[c][pu][pv] = fmaf(w[c * T + ta * KS + tb], v[(pu + PAD - ta) / 2][(pv + PAD - tb) / 2],
                                                  acc[c][pu][pv]);
                        }
                }
    }
    if (!live) return;
    const int W = 2 * a.Wo;
    const long HW = 4L * HWo;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        const float b = a.bias ? a.bias[c] : 0.f;
#pragma unroll
        for (int pu = 0; pu < 2; ++pu) {
            float2* o = reinterpret_cast<float2*>(a.dx + (long)n * a.x_nstride + c * HW + (long)(2 * i + pu) * W + 2 * j);
            float2 r = make_float2(acc[c][pu][0] + b, acc[c][pu][1] + b);
            if (a.accumulate) {
                const float2 old = *o;
                r.x += old.x; r.y += old.y;
            }
            r.x = ghm_act(r.x, a.act, a.alpha);
            r.y = ghm_act(r.y, a.act, a.alpha);
            *o = r;
        }
    }
}

bool thin_fanin_s2_ok(const ghm_conv_desc* d, const float* dx) {
    if (!thin_enabled() || d->stride != 2 || d->kh != d->kw || d->H != 2 * d->Ho || d->W != 2 * d->Wo) return false;
    if (!((d->kh == 3 && d->pad == 1) || (d->kh == 2 && d->pad == 0))) return false;
    if (d->C > 4 || d->C == 2 || d->K > 256 || (long)d->N * d->Ho * d->Wo < 16384) return false;
    return ((uintptr_t)dx % 8 == 0) && (d->x_nstride % 2 == 0);
}

int thin_fanin_s2(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                  float* dx, int act, float alpha, int accumulate) {
    FaninS2Args a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.wp = wp; a.bias = bias; a.dx = dx;
    a.N = d->N; a.K = d->K; a.Ho = d->Ho; a.Wo = d->Wo;
    a.y_nstride = d->y_nstride; a.x_nstride = d->x_nstride;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    const long P = (long)d->N * d->Ho * d->Wo;
    const dim3 grid((unsigned)((P + 255) / 256));
    const int T = d->kh * d->kw;
    const size_t lds = (size_t)d->K * ((d->C * T + 3) & ~3) * sizeof(float);
#define GHM_FANIN_CASE(KS_, PAD_, CS_)                                                                       \
    if (d->kh == KS_ && d->C == CS_) {                                                                       \
        hipLaunchKernelGGL((fanin_s2_kernel<KS_, PAD_, CS_>), grid, dim3(256), lds, ctx->stream, a);         \
        GHM_LAUNCH_CHECK();                                                                                  \
        return 0;                                                                                            \
    }
    GHM_FANIN_CASE(3, 1, 4)
    GHM_FANIN_CASE(3, 1, 3)
    GHM_FANIN_CASE(3, 1, 1)
    GHM_FANIN_CASE(2, 0, 4)
    GHM_FANIN_CASE(2, 0, 3)
    GHM_FANIN_CASE(2, 0, 1)
#undef GHM_FANIN_CASE
    ghm_set_error("fanin_s2: no variant for k=%d C=%d", d->kh, d->C);
    return -3;
}
