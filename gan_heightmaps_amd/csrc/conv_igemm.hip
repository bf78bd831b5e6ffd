// Implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s peak -- MI355X_MICROARCH.md).
//
// One kernel template serves every "gather" convolution of the train step:
//     out[n, r, (u*os+ou), (v*os+ov)] = act( bias[r] + sum_{t in taps} sum_{ch}
//                                            A(t, ch, r) * in[n, ch, u*ss + di[t], v*ss + dj[t]] )
//   forward conv (stride s):       os=1, ss=s, taps = all (a,b), di=a-pad,  A = wp[ch][tap][r]
//   data gradient, stride 1:       os=1, ss=1, taps = all (a,b), di=pad-a,  A = wp[r][tap][ch]  (WT)
//   data gradient, stride 2:       4 launches, one per output parity class (os=2), each with the taps
//                                  whose parity matches; di = (ou+pad-a)/2                        (WT)
//   Deconv2DLayer forward  = data-gradient form (+bias, +act); its data gradient = forward form.
// GEMM view: rows = output channels r (MFMA "A" operand = weights), columns = output pixels (MFMA "B"
// operand = gathered activations), so the accumulator fragment has lanes along pixels and the NCHW
// stores / gathers are pixel-contiguous (coalesced 128 B per 32 lanes).  K index = tap-major,
// channel-minor; a 16-deep K slab is staged through LDS (double-buffered, register-prefetched).
//
// The weight gradient is a second template: rows = (ch, tap) in packed order, columns = filters,
// K = output pixels, split-K over pixel ranges with an fp32 partial reduce.
//
// Replaces Theano's CorrMM / GpuDnnConv{,GradW,GradI}.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS fragment reads for k-step ks+1 are written ahead of the MFMAs of k-step ks; left alone, the scheduler
// sinks each read to just before its use and the wave then waits out the LDS latency on every step.  The fence
// stops LDS reads and MFMAs from crossing (VALU / SALU / VMEM / LDS writes still may: mask bits per the
// amdgcn sched_barrier builtin).  Measured: wgrad_patch 3x3 100 -> 113 TFLOP/s, 5x5 117 -> 122; the forward
// patch kernel (5 reads per 4 MFMAs, 4 waves/SIMD) is 2-4 % slower with it and keeps the compiler's order.
// Round 2: wgrad_patch_kernel moved to GHM_INTERLEAVE below (+1-3 %); wgrad_kernel keeps the fence.
#ifndef GHM_FENCE_MASK
#define GHM_FENCE_MASK 0x0616
#endif
#define GHM_FRAG_FENCE() __builtin_amdgcn_sched_barrier(GHM_FENCE_MASK)
// Alternative used by the forward patch kernel: ask the scheduler for one LDS read after every MFMA of the k-step
// (sched_group_barrier masks: 0x008 MFMA, 0x100 LDS read), so the next step's fragment reads are spread over this
// step's MFMAs instead of bunched before their use.  Measured: 5x5 forward 132 -> 136 TFLOP/s, 3x3 118.7 -> 120.7.
#define GHM_INTERLEAVE(n)                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < (n); ++m_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
    }

#define MAX_TAPS 25

struct IgemmArgs {
    const float* in;
    const float* wp;
    const float* bias;
    float* out;
    int N, CH, Hin, Win;
    long in_nstride;
    int R, Hout, Wout;
    long out_nstride;
    int Hs, Ws;        // output sub-grid
    int os, ou, ov;    // output position = u*os+ou
    int ss;            // source stride
    int T;             // taps in the weight layout (kh*kw)
    int ntaps;         // taps used by this launch
    int act;
    float alpha;
    int accumulate;
    float* partial;        // split-K: raw accumulators go to partial[split][r][p] (no bias / act)
    int slabs_per_split;   // K slabs per blockIdx.y slice
    int* tickets;          // split-K folded into this launch: one arrival counter per output tile (null: separate epilogue)
    int nsplit;            // ... and the number of slices whose blocks arrive at it
    int debug;             // tuning only (GHM_ABLATE): 1 = skip global loads, 2 = also skip LDS stores
    int di[MAX_TAPS], dj[MAX_TAPS], wi[MAX_TAPS];
};

// XCD-aware block remap: consecutive logical tiles (which share weights / halo pixels) land on the
// same XCD's L2 (blocks are dispatched round-robin over the 8 XCDs). Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void splitk_element(const IgemmArgs& a, int S, int r, long p);

template <int BM, int BN, int WM, int WN, bool WT, bool FASTK>
__device__ __forceinline__ void igemm_body(const IgemmArgs& a) {
    constexpr int BK = 16;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int BROWS = 256 / BN, BLOADS = BK / BROWS;
    constexpr int AV = BM / 4;                       // float4 per weight row (forward form)
    constexpr int AROWS = 256 / AV;
    constexpr int APASS = (BK + AROWS - 1) / AROWS;
    constexpr int AT = BM / 16;                      // scalar loads per thread (WT form)
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntr = (a.R + BM - 1) / BM;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = (L % ntr) * BM;
    const int p0 = (L / ntr) * BN;
    const int hw_s = a.Hs * a.Ws;
    const int P = a.N * hw_s;
    const int HinWin = a.Hin * a.Win;
    const int Ktot = a.ntaps * a.CH;
    const int nslabs = (Ktot + BK - 1) / BK;

    // ---- per-thread gather pixel ----
    const int b_pix = tid % BN, b_row0 = tid / BN;
    int sy0 = 0, sx0 = 0;
    bool pvalid;
    const float* inb = a.in;
    {
        const int p = p0 + b_pix;
        pvalid = p < P;
        const int pp = pvalid ? p : 0;
        const int n = pp / hw_s, rem = pp - n * hw_s;
        const int uu = rem / a.Ws, vv = rem - uu * a.Ws;
        sy0 = uu * a.ss;
        sx0 = vv * a.ss;
        inb += (long)n * a.in_nstride;
    }
    const int a_c4 = tid % AV, a_row0 = tid / AV;    // forward-form weight loader
    const int a_k = tid & 15, a_r0 = tid >> 4;       // WT-form weight loader

    float breg[BLOADS];
    float4 areg4[WT ? 1 : APASS];
    float aregs[WT ? AT : 1];
    unsigned bmask = 0, amask = 0;      // validity bits of the prefetched slab, applied when it is stored to LDS

    // Loads are branch-free: an out-of-range element reads a clamped (valid) address and is zeroed by a
    // select, so every load of a slab is issued back to back and waited for once, after the MFMAs.
    auto load_slab = [&](int s) {
        const int kk0 = s * BK;
        if constexpr (FASTK) {                       // CH % 16 == 0 (one tap per slab) and, forward form, R % 4 == 0
            const int ti = kk0 / a.CH;               // uniform
            const int ch0 = kk0 - ti * a.CH;
            const int y = sy0 + a.di[ti], x = sx0 + a.dj[ti];
            const bool ok = pvalid && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
            const float* src = inb + (ok ? (long)(ch0 + b_row0) * HinWin + (y * a.Win + x) : 0L);
            const long cstep = ok ? (long)BROWS * HinWin : 0L;
#pragma unroll
            for (int j = 0; j < BLOADS; ++j) breg[j] = src[j * cstep];
            bmask = ok ? 0xffffffffu : 0u;
            amask = 0;
            const int tapw = a.wi[ti];
            if constexpr (!WT) {
#pragma unroll
                for (int j = 0; j < APASS; ++j) {
                    const int krow = a_row0 + j * AROWS;
                    const int r = r0 + a_c4 * 4;
                    const bool av = krow < BK && r < a.R;
                    const float* wrow = a.wp + ((long)(ch0 + (av ? krow : 0)) * a.T + tapw) * a.R + (av ? r : 0);
                    areg4[j] = *reinterpret_cast<const float4*>(wrow);
                    amask |= (av ? 1u : 0u) << j;
                }
            } else {
#pragma unroll
                for (int j = 0; j < AT; ++j) {
                    const int r = r0 + a_r0 + j * 16;
                    const bool av = r < a.R;
                    aregs[j] = a.wp[((long)(av ? r : 0) * a.T + tapw) * a.CH + ch0 + a_k];
                    amask |= (av ? 1u : 0u) << j;
                }
            }
        } else {
            bmask = 0;
            amask = 0;
#pragma unroll
            for (int j = 0; j < BLOADS; ++j) {
                const int kk = kk0 + b_row0 + j * BROWS;
                const bool kv = kk < Ktot;
                const int kc = kv ? kk : 0;
                const int ti = kc / a.CH, ch = kc - ti * a.CH;
                const int y = sy0 + a.di[ti], x = sx0 + a.dj[ti];
                const bool ok = kv && pvalid && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
                breg[j] = inb[ok ? (long)ch * HinWin + (y * a.Win + x) : 0L];
                bmask |= (ok ? 1u : 0u) << j;
            }
            if constexpr (!WT) {
#pragma unroll
                for (int j = 0; j < APASS; ++j) {
                    const int krow = a_row0 + j * AROWS;
                    const int kk = kk0 + krow;
                    const int r = r0 + a_c4 * 4;
                    const bool kv = krow < BK && kk < Ktot;
                    const int kc = kv ? kk : 0;
                    const int ti = kc / a.CH, ch = kc - ti * a.CH;
                    const float* wrow = a.wp + ((long)ch * a.T + a.wi[ti]) * a.R;
                    float4 v;
                    v.x = wrow[(kv && r + 0 < a.R) ? r + 0 : 0];
                    v.y = wrow[(kv && r + 1 < a.R) ? r + 1 : 0];
                    v.z = wrow[(kv && r + 2 < a.R) ? r + 2 : 0];
                    v.w = wrow[(kv && r + 3 < a.R) ? r + 3 : 0];
                    areg4[j] = v;
                    amask |= ((kv && r + 0 < a.R) ? 1u : 0u) << (4 * j + 0);
                    amask |= ((kv && r + 1 < a.R) ? 1u : 0u) << (4 * j + 1);
                    amask |= ((kv && r + 2 < a.R) ? 1u : 0u) << (4 * j + 2);
                    amask |= ((kv && r + 3 < a.R) ? 1u : 0u) << (4 * j + 3);
                }
            } else {
                const int kk = kk0 + a_k;
                const bool kv = kk < Ktot;
                const int kc = kv ? kk : 0;
                const int ti = kc / a.CH, ch = kc - ti * a.CH;
                const int tapw = a.wi[ti];
#pragma unroll
                for (int j = 0; j < AT; ++j) {
                    const int r = r0 + a_r0 + j * 16;
                    const bool av = kv && r < a.R;
                    aregs[j] = a.wp[((long)(av ? r : 0) * a.T + tapw) * a.CH + ch];
                    amask |= (av ? 1u : 0u) << j;
                }
            }
        }
    };

    auto store_slab = [&](int buf) {
        float* Ab = As + buf * BK * LDA;
        float* Bb = Bs + buf * BK * LDB;
#pragma unroll
        for (int j = 0; j < BLOADS; ++j)
            Bb[(b_row0 + j * BROWS) * LDB + b_pix] = ((bmask >> j) & 1u) ? breg[j] : 0.f;
        if constexpr (!WT) {
#pragma unroll
            for (int j = 0; j < APASS; ++j) {
                const int krow = a_row0 + j * AROWS;
                float4 v = areg4[j];
                if constexpr (FASTK) {
                    if (!((amask >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    if (!((amask >> (4 * j + 0)) & 1u)) v.x = 0.f;
                    if (!((amask >> (4 * j + 1)) & 1u)) v.y = 0.f;
                    if (!((amask >> (4 * j + 2)) & 1u)) v.z = 0.f;
                    if (!((amask >> (4 * j + 3)) & 1u)) v.w = 0.f;
                }
                if (krow < BK) *reinterpret_cast<float4*>(Ab + krow * LDA + a_c4 * 4) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < AT; ++j) Ab[a_k * LDA + a_r0 + j * 16] = ((amask >> j) & 1u) ? aregs[j] : 0.f;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int s_begin = blockIdx.y * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);
    load_slab(s_begin);
    store_slab(0);
    __syncthreads();

    const int frag_k = lane >> 5, frag_i = lane & 31;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more && a.debug < 1) load_slab(s + 1);   // global loads of the next slab fly under the MFMAs
        const float* Ab = As + buf * BK * LDA + wm * (BM / WM) + frag_i;
        const float* Bb = Bs + buf * BK * LDB + wn * (BN / WN) + frag_i;
        float af[2][TM], bf[2][TN];                  // fragment double buffer: LDS latency hides under MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[frag_k * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bb[frag_k * LDB + j * 32];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            if (ks + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(ks + 1) & 1][i] = Ab[((ks + 1) * 2 + frag_k) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(ks + 1) & 1][j] = Bb[((ks + 1) * 2 + frag_k) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            // the next slab goes to the OTHER LDS buffer half-way through, so its ds_writes (and the wait for
            // the global loads) sit between MFMAs instead of in front of the barrier
            if (ks == BK / 4 - 1 && more && a.debug < 2) store_slab(buf ^ 1);
        }
        __syncthreads();
    }

    // ---- epilogue: lanes along pixels (D column = lane&31), rows = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (a.partial) {
        float* pb = a.partial + (long)blockIdx.y * a.R * P;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int p = p0 + wn * (BN / WN) + j * 32 + frag_i;
            if (p >= P) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = r0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * frag_k;
                    if (r < a.R) pb[(long)r * P + p] = acc[i][j][e];
                }
            }
        }
        if (a.tickets) {
            // split-K reduction folded into the producer: the LAST block to arrive at this output tile sums the slices
            // (fixed order 0 .. S-1, whoever arrives last: bit-repeatable) and applies bias / accumulate / activation
            if (!ghm_last_arrival(a.tickets + blockIdx.z * gridDim.x + blockIdx.x, a.nsplit)) return;
            for (int idx = threadIdx.x; idx < BM * BN; idx += 256) {
                const int r = r0 + idx / BN;
                const long p = (long)p0 + idx % BN;
                if (r < a.R && p < P) splitk_element(a, a.nsplit, r, p);
            }
        }
        return;
    }
    const int HWout = a.Hout * a.Wout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = p0 + wn * (BN / WN) + j * 32 + frag_i;
        if (p >= P) continue;
        const int n = p / hw_s, rem = p - n * hw_s;
        const int uu = rem / a.Ws, vv = rem - uu * a.Ws;
        float* ob = a.out + (long)n * a.out_nstride + (long)(uu * a.os + a.ou) * a.Wout + (vv * a.os + a.ov);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = r0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * frag_k;
                if (r < a.R) {
                    float v = acc[i][j][e];
                    if (a.bias) v += a.bias[r];
                    float* o = ob + (long)r * HWout;
                    if (a.accumulate) v += *o;
                    *o = ghm_act(v, a.act, a.alpha);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool WT, bool FASTK>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? (WT ? 3 : 4) : 2) void igemm_kernel(const IgemmArgs a) {
    igemm_body<BM, BN, WM, WN, WT, FASTK>(a);
}

// The four output parity classes of a stride-2 data gradient in ONE launch (blockIdx.z = class; each class has its own
// taps, output offset, split count and partial buffer).  On the 2x2 .. 32x32 maps of the U-Net bottleneck the four
// launches + four split-K epilogues of the per-class form were eight dependent launches of a few microseconds of work
// each on the critical path of the stage stream.
struct IgemmArgs4 {
    IgemmArgs c[4];
    int nsplit[4];
};

template <int BM, int BN, int WM, int WN, bool FASTK>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 3 : 2) void igemm_kernel_x4(const IgemmArgs4 a4) {
    if ((int)blockIdx.y >= a4.nsplit[blockIdx.z]) return;         // uniform: this class has fewer splits
    igemm_body<BM, BN, WM, WN, true, FASTK>(a4.c[blockIdx.z]);
}

// Direct (VALU) form of the same gather convolution for R <= 4 output channels (g_out, d_out, pd_out,
// dconv9, and the data gradients of the 1- and 4-channel first layers): one thread per output pixel.
template <bool WT>
__global__ __launch_bounds__(256) void direct_smallr_kernel(const IgemmArgs a) {
    const int hw_s = a.Hs * a.Ws;
    const long P = (long)a.N * hw_s;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int n = (int)(p / hw_s), rem = (int)(p - (long)n * hw_s);
    const int uu = rem / a.Ws, vv = rem - uu * a.Ws;
    const float* inb = a.in + (long)n * a.in_nstride;
    const int HinWin = a.Hin * a.Win;
    // a long channel reduction with few output pixels is split over blockIdx.y (slabs_per_split = channels per slice):
    // 2048 threads each walking 512 channels x 9 taps is latency-bound, 32 slices of 16 channels fill the chip
    const int ch_begin = blockIdx.y * a.slabs_per_split;
    const int ch_end = min(a.CH, ch_begin + a.slabs_per_split);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.ntaps; ++t) {
        const int y = uu * a.ss + a.di[t], x = vv * a.ss + a.dj[t];
        const bool ok = (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        const float* src = inb + (y * a.Win + x);
        const int tapw = a.wi[t];
#pragma unroll 4
        for (int ch = ch_begin; ch < ch_end; ++ch) {
            const float v = ok ? src[(long)ch * HinWin] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < a.R) {
                    const float w = WT ? a.wp[((long)r * a.T + tapw) * a.CH + ch]
                                       : a.wp[((long)ch * a.T + tapw) * a.R + r];
                    acc[r] = fmaf(v, w, acc[r]);
                }
            }
        }
    }
    if (a.partial) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < a.R) a.partial[((long)blockIdx.y * a.R + r) * P + p] = acc[r];
        return;
    }
    float* ob = a.out + (long)n * a.out_nstride + (long)(uu * a.os + a.ou) * a.Wout + (vv * a.os + a.ov);
    const int HWout = a.Hout * a.Wout;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r < a.R) {
            float v = acc[r];
            if (a.bias) v += a.bias[r];
            float* o = ob + (long)r * HWout;
            if (a.accumulate) v += *o;
            *o = ghm_act(v, a.act, a.alpha);
        }
    }
}

// Data gradient of a convolution with <= 4 filters (d_out, pd_out: architectures/dcgan.py:50, p2p.py:289): the
// reduction runs over those few filters and the taps, so it is element-wise work -- one thread per dx element, any
// stride.  (Through the implicit-GEMM tiles this took 4 parity launches with a 16-deep K slab holding one real row.)
struct SmallKDgradArgs {
    const float* dy;
    const float* wp;       // [C][T][K]
    const float* bias;
    float* dx;
    int N, C, H, W, K, Ho, Wo, kh, kw, stride, pad;
    long x_nstride, y_nstride;
    int act;
    float alpha;
    int accumulate;
    const float* dact_y;       // see PatchArgs
    long dact_nstride;
    int dact;
    float dact_alpha;
};

__global__ __launch_bounds__(256) void smallk_dgrad_kernel(const SmallKDgradArgs a) {
    const long HW = (long)a.H * a.W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.N * a.C * HW) return;
    const long nc = idx / HW;
    const int rem = (int)(idx - nc * HW), u = rem / a.W, v = rem - u * a.W;
    const int n = (int)(nc / a.C), c = (int)(nc - (long)n * a.C);
    const float* dyb = a.dy + (long)n * a.y_nstride;
    const float* wc = a.wp + (long)c * a.kh * a.kw * a.K;
    const int HoWo = a.Ho * a.Wo;
    float s = a.bias ? a.bias[c] : 0.f;
    for (int ta = 0; ta < a.kh; ++ta) {
        const int yy = u + a.pad - ta;
        if (yy < 0 || yy % a.stride) continue;
        const int i = yy / a.stride;
        if (i >= a.Ho) continue;
        for (int tb = 0; tb < a.kw; ++tb) {
            const int xx = v + a.pad - tb;
            if (xx < 0 || xx % a.stride) continue;
            const int j = xx / a.stride;
            if (j >= a.Wo) continue;
            const float* w = wc + (ta * a.kw + tb) * a.K;
            for (int k = 0; k < a.K; ++k) s = fmaf(dyb[(long)k * HoWo + i * a.Wo + j], w[k], s);
        }
    }
    float* o = a.dx + (long)n * a.x_nstride + (long)c * HW + rem;
    if (a.accumulate) s += *o;
    s = ghm_act(s, a.act, a.alpha);
    if (a.dact_y) s *= a.dact_y[(long)n * a.dact_nstride + (long)c * HW + rem] > 0.f ? 1.f : a.dact_alpha;
    *o = s;
}

// K == 1 (d_out, pd_out): the taps of dy a dx pixel sees are the same for every input channel, so a thread owns one pixel
// and CG consecutive channels (blockIdx.y = channel group: the weights are wave-uniform scalar loads): the tap decode and
// the dy loads are paid once per CG outputs instead of once per output.  Same products in the same order as
// smallk_dgrad_kernel (taps row-major, fmaf chain from the bias): bit-identical.  pd_out at 512^2, batch 8 (C512 32x32 k3
// s2): 55 -> 18 us; it sits at the turn-around of the pix2pix stage stream, where nothing of that stage runs beside it.
template <int KS, int CG>
__global__ __launch_bounds__(256) void smallk1_dgrad_kernel(const SmallKDgradArgs a) {
    const int HW = a.H * a.W;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)a.N * HW) return;
    const int n = (int)(p / HW), rem = (int)(p - (long)n * HW), u = rem / a.W, v = rem - u * a.W;
    const int c0 = blockIdx.y * CG;
    const float* dyb = a.dy + (long)n * a.y_nstride;
    float dv[KS * KS];
#pragma unroll
    for (int ta = 0; ta < KS; ++ta) {
        const int yy = u + a.pad - ta;
        const int i = yy / a.stride;
        const bool rok = yy >= 0 && yy % a.stride == 0 && i < a.Ho;
#pragma unroll
        for (int tb = 0; tb < KS; ++tb) {
            const int xx = v + a.pad - tb;
            const int j = xx / a.stride;
            const bool ok = rok && xx >= 0 && xx % a.stride == 0 && j < a.Wo;
            dv[ta * KS + tb] = ok ? dyb[i * a.Wo + j] : 0.f;
        }
    }
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
        const int c = c0 + cc;
        if (c >= a.C) break;
        const float* wc = a.wp + (long)c * KS * KS;
        float s = a.bias ? a.bias[c] : 0.f;
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) s = fmaf(dv[t], wc[t], s);      // (a tap outside the image adds +0: dv = 0)
        float* o = a.dx + (long)n * a.x_nstride + (long)c * HW + rem;
        if (a.accumulate) s += *o;
        s = ghm_act(s, a.act, a.alpha);
        if (a.dact_y) s *= a.dact_y[(long)n * a.dact_nstride + (long)c * HW + rem] > 0.f ? 1.f : a.dact_alpha;
        *o = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Stride-1 convolution with an LDS-staged input PATCH (forward form; the data gradient runs the same
// kernel on the transposed weights, see ghm_conv2d_dgrad_t).  A block owns BM output channels x a 2-D tile
// of RT rows x 32 columns of one image.  Per slab of CB = 2*CP input channels it stages
//   - the weight rows wp[c][tap][r0..r0+BM) of those channels (a contiguous [CB*T][BM] copy), and
//   - the input patch those pixels touch, (RT+KS-1) x (32+KS-1) per channel, ONCE (not once per tap),
// and every MFMA B fragment (32 consecutive pixels of one row) is read from the patch at a compile-time
// offset: k-step (cp, tap) pairs channel 2cp (lanes 0-31) with channel 2cp+1 (lanes 32-63) on the same tap.
// The pixel tile is fixed for the block, so bounds masks and source offsets are loop invariants and a slab
// costs pointer bumps only.  Both LDS images are filled by global->LDS DMA issued one slab ahead (no staging
// registers, no LDS store phase); the wave waits for its own DMA (vmcnt) just before the slab barrier.
// ------------------------------------------------------------------------------------------------
struct PatchArgs {
    const float* in;
    const float* wp;       // [CH][T][R]
    const float* bias;
    float* out;
    float* partial;
    int N, CH, H, W;       // OUTPUT grid (tiles of RT x 32 output pixels)
    int Hin, Win;          // input grid (== H, W for stride 1)
    long in_nstride;
    int R;
    long out_nstride;
    int pad;               // rows/cols of halo before the tile
    int act;
    float alpha;
    int accumulate;
    int slabs_per_split;
    int debug;             // tuning only (GHM_ABLATE): 1 = skip the staging of all slabs but the first
    const float* zeros;    // ctx->zeros: source of padding elements for the LDS-DMA staging
    float* pool_out;       // POOL: dense [N, R, H/2, W/2] maximum of act(conv + bias) over 2x2 windows ...
    unsigned char* pool_mask;   // ... and the 4-bit arg-max mask of every window (bit 2*dr + dc; all ties set)
    // data gradients only: out *= act'(dact_y) with dact_y = the (post-activation) tensor whose gradient this is --
    // the backward of the PRODUCER's nonlinearity folded into this epilogue instead of a separate act_bwd pass
    const float* dact_y;
    long dact_nstride;
    int dact;
    float dact_alpha;
};

// The 2x2 max-pool of a patch-kernel accumulator tile, in the epilogue (POOL): a wave owns TN consecutive pixel rows
// of 32 columns, lanes along columns -- the row pair (2j, 2j+1) of a window is in one lane, the column pair in lanes
// (2t, 2t+1): one cross-lane exchange, even lanes store.  v0 / v1: the two rows' values of this lane's column.
__device__ __forceinline__ void pool2_store(float v0, float v1, bool col_even, bool live, float* po, unsigned char* pm) {
    const float w0 = __shfl_xor(v0, 1, 64), w1 = __shfl_xor(v1, 1, 64);      // the neighbour column's two rows
    const float m = fmaxf(fmaxf(v0, v1), fmaxf(w0, w1));
    if (col_even && live) {
        *po = m;
        *pm = (unsigned char)((v0 == m ? 1u : 0u) | (w0 == m ? 2u : 0u) | (v1 == m ? 4u : 0u) | (w1 == m ? 8u : 0u) |
                              (m > 0.f ? GHM_POOL_SIGN : 0u));
    }
}

template <int KS, int BM, int RT, int WM, int WN, int CP, int ST>
__global__ __launch_bounds__(256, ST == 2 ? 3 : 4) void conv_patch_kernel(const PatchArgs a) {
    constexpr int T = KS * KS, CB = 2 * CP;
    constexpr int BN = RT * 32;
    constexpr int LDA = BM;                         // unpadded: a weight row is one contiguous run of the DMA image
    constexpr int PH = (RT - 1) * ST + KS, PWN = 31 * ST + KS, PW = PWN, PS = PH * PW;
    constexpr int KR = CB * T;                      // weight rows per slab
    constexpr int ASZ = KR * LDA, PSZ = ((CB * PS + 3) / 4) * 4;
    constexpr int TM = BM / (WM * 32), TN = RT / WN;
    constexpr int AV = BM / 4;                      // float4 per weight row
    constexpr int NA4 = KR * AV;                    // float4 of the weight slab
    constexpr int AL = (NA4 + 255) / 256;
    constexpr int NEL = CB * PH * PWN;
    constexpr int BL = (NEL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // 2 x ASZ
    float* Ps = smem + 2 * ASZ;                     // 2 x PSZ

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frag_k = lane >> 5, frag_i = lane & 31;
    // ---- block -> (r tile, image, tile row, tile column); r fastest so neighbours share the patch in L2
    const int ntr = (a.R + BM - 1) / BM;
    const int tiles_x = a.W / 32, tiles_y = a.H / RT;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = (L % ntr) * BM;
    L /= ntr;
    const int tx = L % tiles_x;
    L /= tiles_x;
    const int ty = L % tiles_y;
    const int n = L / tiles_y;
    const int y0 = ty * RT, x0 = tx * 32;
    const int HW = a.H * a.W, HWin = a.Hin * a.Win;
    const int nslabs = a.CH / CB;
    const int s_begin = blockIdx.y * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);

    // ---- staging: both LDS images are LINEAR in the fetch index (weight slab: float4 f = row * AV + c4 at float
    // 4f; patch: element e = (c, py, px) at float e), so every wave-instruction of a global->LDS DMA fills 64
    // consecutive slots (uniform base + lane * size) and nothing passes through VGPRs.  Padding elements and filter
    // columns past R read a.zeros.  Per-lane state: element offsets relative to the per-slab base pointers.
    int a_off[AL];
    unsigned amask = 0;
#pragma unroll
    for (int q = 0; q < AL; ++q) {
        const int f = tid + q * 256;
        const int row = f / AV, c4 = f - row * AV;
        const bool v = f < NA4 && (r0 + c4 * 4) < a.R;
        a_off[q] = v ? row * a.R + r0 + c4 * 4 : 0;
        amask |= (v ? 1u : 0u) << q;
    }
    int p_off[BL];
    unsigned pmask = 0;
#pragma unroll
    for (int q = 0; q < BL; ++q) {
        const int e = tid + q * 256;
        const int c = e / (PH * PWN), r = e - c * (PH * PWN);
        const int py = r / PWN, px = r - py * PWN;
        const int y = y0 * ST + py - a.pad, x = x0 * ST + px - a.pad;
        const bool ok = e < NEL && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        p_off[q] = ok ? c * HWin + y * a.Win + x : 0;
        pmask |= (ok ? 1u : 0u) << q;
    }
    const float* wbase = a.wp + (long)s_begin * KR * a.R;                         // uniform
    const float* ibase = a.in + (long)n * a.in_nstride + (long)s_begin * CB * HWin; // uniform
    const long a_step = (long)KR * a.R, p_step = (long)CB * HWin;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto stage_slab = [&](int buf) {
        float* Ab = As + buf * ASZ + wave * 256;        // this wave's 64 float4 slots of pass q start at q*1024 floats
        float* Pb = Ps + buf * PSZ + wave * 64;
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            if (q * 256 + wave * 64 < NA4) {            // uniform; NA4 % 16 == 0 and the tail lanes fall on whole rows
                const float* g = ((amask >> q) & 1u) ? wbase + a_off[q] : a.zeros;
                if (tid + q * 256 < NA4)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Ab + q * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            if (q * 256 + wave * 64 < NEL) {
                const float* g = ((pmask >> q) & 1u) ? ibase + p_off[q] : a.zeros;
                if (tid + q * 256 < NEL)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Pb + q * 256), 4, 0, 0);
            }
        }
        wbase += a_step;
        ibase += p_step;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (s_begin < s_end) stage_slab(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // lane bases: A rows (frag_k selects the odd channel of the pair), patch (pixel column, row segment)
    const int abase = frag_k * T * LDA + wm * (BM / WM) + frag_i;
    const int pbase = frag_k * PS + (wn * TN * ST) * PW + frag_i * ST;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more && a.debug < 1) stage_slab(buf ^ 1);   // lands during this slab's MFMAs; waited for before the barrier
        const float* Ab = As + buf * ASZ + abase;
        const float* Pb = Ps + buf * PSZ + pbase;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Pb[j * ST * PW];
#pragma unroll
        for (int ks = 0; ks < CP * T; ++ks) {
            if (ks + 1 < CP * T) {
                const int cp = (ks + 1) / T, tap = (ks + 1) % T;
                const int aoff = (2 * cp * T + tap) * LDA;
                const int poff = 2 * cp * PS + (tap / KS) * PW + (tap % KS);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(ks + 1) & 1][i] = Ab[aoff + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(ks + 1) & 1][j] = Pb[poff + j * ST * PW];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            if (ks + 1 < CP * T) GHM_INTERLEAVE(TM * TN);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue ----
    // Loads and stores share vmcnt: a bias (or accumulate) load issued after a store makes the wave wait for that
    // store's acknowledgement, once per element if they interleave.  So the bias goes to registers first (one batch,
    // one wait), the activation class is chosen once, and old values are read 16 at a time.
    // Addresses: wave-uniform row pointer (SGPR pair) + one unsigned 32-bit lane offset, so no per-element 64-bit
    // address registers are held.  Element e of row tile i sits k = i*32 + (e&3) + 8*(e>>2) rows below the wave's
    // first row; the lane adds 4*frag_k rows and its pixel column.
    const long P = (long)a.N * HW;
    const int ru = r0 + wm * (BM / WM);                          // uniform first row of this wave
    const int rl = ru + 4 * frag_k;                              // this lane's first row
    if (a.partial) {
        float* const pb = a.partial + ((long)blockIdx.y * a.R + ru) * P + (long)n * HW + (long)(y0 + wn * TN) * a.W + x0;
        const unsigned lo = 4u * frag_k * (unsigned)P + frag_i;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    float* rowp = pb + (long)k * P + j * a.W;
                    if (rl + k < a.R) rowp[lo] = acc[i][j][e];
                }
        return;
    }
    // bias through LDS (free after the last slab barrier): LDS reads count on lgkmcnt, not on the stores' vmcnt
    float* const sb = smem;
    if (tid < BM) sb[tid] = (a.bias && r0 + tid < a.R) ? a.bias[r0 + tid] : 0.f;
    __syncthreads();
    const float* const lb = sb + wm * (BM / WM) + 4 * frag_k;
    // pooled epilogue (a.pool_out set; row pairs inside a wave): a run-time option of the same kernel -- the main loop is
    // identical and this path needs fewer registers than the plain one
    if constexpr (TN % 2 == 0 && ST == 1) if (a.pool_out) {
        const int Wp = a.W / 2;
        const long HWp = (long)(a.H / 2) * Wp;
        const long base = ((long)n * a.R + rl) * HWp + (long)((y0 + wn * TN) / 2) * Wp + (x0 + frag_i) / 2;
        const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
#pragma unroll
        for (int j2 = 0; j2 < TN / 2; ++j2)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    float v0 = acc[i][2 * j2][e] + lb[k], v1 = acc[i][2 * j2 + 1][e] + lb[k];
                    v0 = v0 > 0.f ? v0 : slope * v0;
                    v1 = v1 > 0.f ? v1 : slope * v1;
                    const long o = base + (long)k * HWp + j2 * Wp;
                    pool2_store(v0, v1, (frag_i & 1) == 0, rl + k < a.R, a.pool_out + o, a.pool_mask + o);
                }
        return;
    }
    float* const ub = a.out + (long)n * a.out_nstride + (long)ru * HW + (long)(y0 + wn * TN) * a.W + x0;
    const unsigned lo = 4u * frag_k * (unsigned)HW + frag_i;
    const bool pwl = a.act == GHM_ACT_LINEAR || a.act == GHM_ACT_RELU || a.act == GHM_ACT_LRELU;
    if (r0 + BM <= a.R && pwl) {
        // every tile of this workload: full row tile, piecewise-linear activation (linear = slope 1)
        const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
        if (!a.accumulate) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                        float* rowp = ub + (long)k * HW + j * a.W;
                        const float v = acc[i][j][e] + lb[k];
                        rowp[lo] = v > 0.f ? v : slope * v;
                    }
        } else {
            // read-modify-write (a data gradient summed into an existing one): 16 old values per batch of stores
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e0 = 0; e0 < 16; e0 += 4) {          // 4 old values per batch of stores (more spill)
                        float old[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = i * 32 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
                            const float* rowp = ub + (long)k * HW + j * a.W;
                            old[e] = rowp[lo];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = i * 32 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
                            float* rowp = ub + (long)k * HW + j * a.W;
                            const float v = acc[i][j][e0 + e] + lb[k] + old[e];
                            rowp[lo] = v > 0.f ? v : slope * v;
                        }
                    }
        }
        return;
    }
    // ragged row tiles, tanh / sigmoid: element by element
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                if (rl + k < a.R) {
                    float* rowp = ub + (long)k * HW + j * a.W;
                    float v = acc[i][j][e] + lb[k];
                    if (a.accumulate) v += rowp[lo];
                    rowp[lo] = ghm_act(v, a.act, a.alpha);
                }
            }
}

// ------------------------------------------------------------------------------------------------
// Data gradient of a 3x3 / stride-2 / pad-1 convolution (U-Net encoder, PatchGAN) with an LDS patch of dy.
// dx[c, 2i+pu, 2j+pv] only receives the taps whose parity matches (1, 2, 2 or 4 of the 9), so the four
// output parity classes are four small stride-1 gathers over the SAME dy patch.  One block computes all
// four classes of BM channels x (2 x 32) class pixels (= 4 x 64 dx pixels): every k-step (channel pair,
// tap) feeds exactly one class, so the 9 taps cost 9 MFMA k-steps -- no zero-insertion waste -- and the
// dy patch (3 x 33 per channel) and the transposed weight rows are staged once for all classes.
// a.in = dy [N, CH=K, Hc, Wc] (class grid = conv output grid), a.out = dx [N, R=C, 2Hc, 2Wc],
// a.wp = wpT[k][8 - tap][c].
// ------------------------------------------------------------------------------------------------
template <int BM, int WM, int CP, bool DACT = false>
__global__ __launch_bounds__(256, 3) void dgrad_s2_patch_kernel(const PatchArgs a) {
    constexpr int T = 9, CB = 2 * CP, WN = 4 / WM, RT = WN;      // a wave: BM/WM channels x one class row x 4 classes
    constexpr int LDA = BM;                         // unpadded: both LDS images are linear in the DMA fetch index
    constexpr int PH = RT + 1, PWN = 33, PW = 33, PS = PH * PW;
    constexpr int KR = CB * T;
    constexpr int ASZ = KR * LDA, PSZ = ((CB * PS + 3) / 4) * 4;
    constexpr int TM = BM / (WM * 32);
    static_assert(WM * WN == 4, "4 waves");
    constexpr int AV = BM / 4;
    constexpr int NA4 = KR * AV;
    constexpr int AL = (NA4 + 255) / 256;
    constexpr int NEL = CB * PH * PWN;
    constexpr int BL = (NEL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ps = smem + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int frag_k = lane >> 5, frag_i = lane & 31;
    const int Hc = a.Hin, Wc = a.Win;                 // class grid == dy grid
    const int ntr = (a.R + BM - 1) / BM;
    const int tiles_x = Wc / 32, tiles_y = (Hc + RT - 1) / RT;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = (L % ntr) * BM;
    L /= ntr;
    const int tx = L % tiles_x;
    L /= tiles_x;
    const int ty = L % tiles_y;
    const int n = L / tiles_y;
    const int i0 = ty * RT, j0 = tx * 32;
    const int HWc = Hc * Wc, HWx = a.H * a.W;
    const int nslabs = a.CH / CB;
    const int s_begin = blockIdx.y * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);

    // staging exactly as in conv_patch_kernel: weight slab (float4 f = row * AV + c4) and dy patch (element e =
    // (c, py, px)) by global->LDS DMA one slab ahead; the patch only needs padding on its far edges (rows i, i+1)
    int a_off[AL];
    unsigned amask = 0;
#pragma unroll
    for (int q = 0; q < AL; ++q) {
        const int f = tid + q * 256;
        const int row = f / AV, c4 = f - row * AV;
        const bool v = f < NA4 && (r0 + c4 * 4) < a.R;
        a_off[q] = v ? row * a.R + r0 + c4 * 4 : 0;
        amask |= (v ? 1u : 0u) << q;
    }
    int p_off[BL];
    unsigned pmask = 0;
#pragma unroll
    for (int q = 0; q < BL; ++q) {
        const int e = tid + q * 256;
        const int c = e / (PH * PWN), r = e - c * (PH * PWN);
        const int py = r / PWN, px = r - py * PWN;
        const int y = i0 + py, x = j0 + px;
        const bool ok = e < NEL && y < Hc && x < Wc;
        p_off[q] = ok ? c * HWc + y * Wc + x : 0;
        pmask |= (ok ? 1u : 0u) << q;
    }
    const float* wbase = a.wp + (long)s_begin * KR * a.R;
    const float* ibase = a.in + (long)n * a.in_nstride + (long)s_begin * CB * HWc;
    const long a_step = (long)KR * a.R, p_step = (long)CB * HWc;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto stage_slab = [&](int buf) {
        float* Ab = As + buf * ASZ + wave * 256;
        float* Pb = Ps + buf * PSZ + wave * 64;
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            if (q * 256 + wave * 64 < NA4) {
                const float* g = ((amask >> q) & 1u) ? wbase + a_off[q] : a.zeros;
                if (tid + q * 256 < NA4)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Ab + q * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            if (q * 256 + wave * 64 < NEL) {
                const float* g = ((pmask >> q) & 1u) ? ibase + p_off[q] : a.zeros;
                if (tid + q * 256 < NEL)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Pb + q * 256), 4, 0, 0);
            }
        }
        wbase += a_step;
        ibase += p_step;
    };

    f32x16 acc[4][TM];                              // [parity class pu*2+pv][row tile]
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cl][i][e] = 0.f;

    if (s_begin < s_end) stage_slab(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int abase = frag_k * T * LDA + wm * (BM / WM) + frag_i;
    const int pbase = frag_k * PS + wn * PW + frag_i;
    // k-step ks = (cp, tw): tw = tap index in wpT order = 8 - original tap (ta, tb); the tap feeds class
    // (pu, pv) = (ta != 1, tb != 1) from dy[i + (ta == 0)][j + (tb == 0)]
    auto a_off_of = [](int ks) { return (2 * (ks / T) * T + ks % T) * LDA; };
    auto p_off_of = [](int ks) {
        const int tw = ks % T, ta = (8 - tw) / 3, tb = (8 - tw) % 3;
        return 2 * (ks / T) * PS + (ta == 0 ? 1 : 0) * PW + (tb == 0 ? 1 : 0);
    };
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more) stage_slab(buf ^ 1);
        const float* Ab = As + buf * ASZ + abase;
        const float* Pb = Ps + buf * PSZ + pbase;
        float af[2][TM], bf[2];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[a_off_of(0) + i * 32];
        bf[0] = Pb[p_off_of(0)];
#pragma unroll
        for (int ks = 0; ks < CP * T; ++ks) {
            if (ks + 1 < CP * T) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(ks + 1) & 1][i] = Ab[a_off_of(ks + 1) + i * 32];
                bf[(ks + 1) & 1] = Pb[p_off_of(ks + 1)];
            }
            const int tw = ks % T, ta = (8 - tw) / 3, tb = (8 - tw) % 3;
            const int cl = (ta == 1 ? 0 : 2) + (tb == 1 ? 0 : 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[cl][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1], acc[cl][i], 0, 0, 0);
            if (ks + 1 < CP * T) GHM_INTERLEAVE(TM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: the two column parities of class pixel (ic, jc) are dx columns 2jc, 2jc+1 -- both in this lane: one
    // 8-byte store per (row parity, channel).  Wave-uniform row pointers + one 32-bit lane offset (as conv_patch_kernel).
    const long P = (long)a.N * HWx;
    const int ic = i0 + wn;                                       // uniform
    const int ru = r0 + wm * (BM / WM), rl = ru + 4 * frag_k;
    float* const sb = smem;                                       // bias through LDS (free after the last slab barrier)
    if (tid < BM) sb[tid] = (a.bias && !a.partial && r0 + tid < a.R) ? a.bias[r0 + tid] : 0.f;
    __syncthreads();
    if (ic >= Hc) return;
    const float* const lb = sb + wm * (BM / WM) + 4 * frag_k;
    const long rowpix = (long)(2 * ic) * a.W + 2 * j0;
    float* const ub = a.partial ? a.partial + ((long)blockIdx.y * a.R + ru) * P + (long)n * HWx + rowpix
                                : a.out + (long)n * a.out_nstride + (long)ru * HWx + rowpix;
    const long rstride = a.partial ? P : (long)HWx;
    const unsigned lo = 4u * frag_k * (unsigned)rstride + 2u * frag_i;
    const float* const yb = DACT ? a.dact_y + (long)n * a.dact_nstride + (long)ru * HWx + rowpix : nullptr;
    const bool plain = a.partial != nullptr;
#pragma unroll
    for (int pu = 0; pu < 2; ++pu)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                if (rl + k < a.R) {
                    float2* o = reinterpret_cast<float2*>(ub + (long)k * rstride + pu * a.W + lo);
                    float2 v = make_float2(acc[2 * pu][i][e], acc[2 * pu + 1][i][e]);
                    if (!plain) {
                        v.x += lb[k]; v.y += lb[k];
                        if (a.accumulate) { const float2 old = *o; v.x += old.x; v.y += old.y; }
                        v.x = ghm_act(v.x, a.act, a.alpha);
                        v.y = ghm_act(v.y, a.act, a.alpha);
                        if constexpr (DACT) {    // its own instantiation (register budget).  relu / lrelu: slope
                            const float2 yy = *reinterpret_cast<const float2*>(yb + (long)k * HWx + pu * a.W + lo);
                            v.x *= yy.x > 0.f ? 1.f : a.dact_alpha;
                            v.y *= yy.y > 0.f ? 1.f : a.dact_alpha;
                        }
                    }
                    *o = v;
                }
            }
}

// wpT[k][T-1-tap][c] = wp[c][tap][k]: packed weights of the adjoint convolution (data gradient as a forward conv)
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ wp, float* __restrict__ wpT,
                                                                int C, int T, int K) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        tile[i][tx] = (c < C && k < K) ? wp[((long)c * T + tap) * K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        if (k < K && c < C) wpT[((long)k * T + (T - 1 - tap)) * C + c] = tile[tx][i];
    }
}

// the same for every layer of a net in ONE launch (29 launches of ~9 us each otherwise): a device table of
// {wp, wpT, C, T, K, first block}; a block finds its layer by a linear scan of the (<= 64-entry) table
struct TransposeItem {
    const float* wp;
    float* wpT;
    int C, T, K, block_begin;
};

__global__ __launch_bounds__(256) void transpose_weights_batched_kernel(const TransposeItem* __restrict__ items, int n) {
    __shared__ float tile[32][33];
    int li = 0;
    while (li + 1 < n && (int)blockIdx.x >= items[li + 1].block_begin) ++li;
    const TransposeItem it = items[li];
    const int local = blockIdx.x - it.block_begin;
    const int kt = (it.K + 31) / 32, ct = (it.C + 31) / 32;
    const int tap = local / (kt * ct), rem = local - tap * (kt * ct);
    const int c0 = (rem / kt) * 32, k0 = (rem % kt) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        tile[i][tx] = (c < it.C && k < it.K) ? it.wp[((long)c * it.T + tap) * it.K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        if (k < it.K && c < it.C) it.wpT[((long)k * it.T + (it.T - 1 - tap)) * it.C + c] = tile[tx][i];
    }
}

// split-K epilogue: sum the partial slices, then bias / accumulate / activation and the NCHW scatter
__device__ __forceinline__ void splitk_epilogue_body(const IgemmArgs& a, int S);
__global__ __launch_bounds__(256) void igemm_splitk_epilogue(const IgemmArgs a, int S) { splitk_epilogue_body(a, S); }
__global__ __launch_bounds__(256) void igemm_splitk_epilogue_x4(const IgemmArgs4 a4) {
    splitk_epilogue_body(a4.c[blockIdx.y], a4.nsplit[blockIdx.y]);
}
__device__ __forceinline__ void splitk_element(const IgemmArgs& a, int S, int r, long p) {
    const int hw_s = a.Hs * a.Ws;
    const long P = (long)a.N * hw_s;
    const float* pp = a.partial + (long)r * P + p;
    const long sstride = (long)a.R * P;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int k = 0;
    for (; k + 3 < S; k += 4) {
        v0 += pp[(long)(k + 0) * sstride];
        v1 += pp[(long)(k + 1) * sstride];
        v2 += pp[(long)(k + 2) * sstride];
        v3 += pp[(long)(k + 3) * sstride];
    }
    for (; k < S; ++k) v0 += pp[(long)k * sstride];
    float v = (v0 + v1) + (v2 + v3);
    const int n = (int)(p / hw_s), rem = (int)(p - (long)n * hw_s);
    const int uu = rem / a.Ws, vv = rem - uu * a.Ws;
    float* o = a.out + (long)n * a.out_nstride + (long)r * a.Hout * a.Wout +
               (long)(uu * a.os + a.ou) * a.Wout + (vv * a.os + a.ov);
    if (a.bias) v += a.bias[r];
    if (a.accumulate) v += *o;
    *o = ghm_act(v, a.act, a.alpha);
}
__device__ __forceinline__ void splitk_epilogue_body(const IgemmArgs& a, int S) {
    const long P = (long)a.N * a.Hs * a.Ws;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * a.R) return;
    const int r = (int)(idx / P);
    splitk_element(a, S, r, idx - (long)r * P);
}

// R <= 4 with few output pixels and a long reduction (d_out, pd_out): one WAVE per output pixel, the 64
// lanes stride over the (tap, channel) reduction and combine with a wave reduction.
template <bool WT>
__global__ __launch_bounds__(256) void direct_smallr_wave_kernel(const IgemmArgs a) {
    const int hw_s = a.Hs * a.Ws;
    const long P = (long)a.N * hw_s;
    const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const int n = (int)(p / hw_s), rem = (int)(p - (long)n * hw_s);
    const int uu = rem / a.Ws, vv = rem - uu * a.Ws;
    const float* inb = a.in + (long)n * a.in_nstride;
    const int HinWin = a.Hin * a.Win;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.ntaps; ++t) {
        const int y = uu * a.ss + a.di[t], x = vv * a.ss + a.dj[t];
        if ((unsigned)y >= (unsigned)a.Hin || (unsigned)x >= (unsigned)a.Win) continue;   // wave-uniform
        const float* src = inb + (y * a.Win + x);
        const int tapw = a.wi[t];
        for (int ch = lane; ch < a.CH; ch += 64) {
            const float v = src[(long)ch * HinWin];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < a.R) {
                    const float w = WT ? a.wp[((long)r * a.T + tapw) * a.CH + ch]
                                       : a.wp[((long)ch * a.T + tapw) * a.R + r];
                    acc[r] = fmaf(v, w, acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_down(acc[r], o, 64);
    if (lane == 0) {
        float* ob = a.out + (long)n * a.out_nstride + (long)(uu * a.os + a.ou) * a.Wout + (vv * a.os + a.ov);
        const int HWout = a.Hout * a.Wout;
        for (int r = 0; r < a.R && r < 4; ++r) {
            float v = acc[r];
            if (a.bias) v += a.bias[r];
            float* o = ob + (long)r * HWout;
            if (a.accumulate) v += *o;
            *o = ghm_act(v, a.act, a.alpha);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dy;
    float* out;
    int N, C, H, W;
    long x_nstride;
    int K, Ho, Wo;
    long y_nstride;
    int kh, kw, stride, pad;
    int CT;               // C * kh * kw rows
    int P;                // N * Ho * Wo
    int slabs_per_split;  // 16-pixel slabs per z-slice
    long split_stride;    // elements between partial slices (0 when writing dwp directly)
    int accumulate;
    int debug;            // tuning only (GHM_ABLATE)
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 3 : 2) void wgrad_kernel(const WgradArgs a) {
    // rows = (channel, tap) of the packed weight layout, columns = filters, K = output pixels in 16-pixel
    // slabs; same pipeline as igemm_kernel: double-buffered LDS, register prefetch of the next slab, its LDS
    // store half-way through the MFMAs, fragment double buffer, one barrier per slab.
    constexpr int BKP = 16;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AL = BM / 16, BL = BN / 16;
    __shared__ __attribute__((aligned(16))) float smem[2 * BKP * (LDA + LDB) + 2 * BM];
    float* As = smem;
    float* Bs = smem + 2 * BKP * LDA;
    int* rowoff = reinterpret_cast<int*>(smem + 2 * BKP * (LDA + LDB));
    int* rowdij = rowoff + BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int T = a.kh * a.kw;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;

    for (int row = tid; row < BM; row += 256) {
        const int g = m0 + row;
        if (g < a.CT) {
            const int c = g / T, tap = g - c * T;
            const int ta = tap / a.kw, tb = tap - ta * a.kw;
            rowoff[row] = c * HW;
            rowdij[row] = ((ta - a.pad) << 16) | ((tb - a.pad) & 0xffff);
        } else {
            rowoff[row] = -1;
            rowdij[row] = 0;
        }
    }
    __syncthreads();

    const int lp = tid & 15, lg = tid >> 4;
    const int total_slabs = (a.P + BKP - 1) / BKP;
    const int s_begin = blockIdx.z * a.slabs_per_split;
    const int s_end = min(s_begin + a.slabs_per_split, total_slabs);

    // pixel cursor of the NEXT slab this thread loads: advanced by 16 pixels per slab without divisions
    int pn, pi, pj;
    long pcur = (long)s_begin * BKP + lp;
    {
        const long pp = pcur < a.P ? pcur : 0;
        pn = (int)(pp / HoWo);
        const int rem = (int)(pp - (long)pn * HoWo);
        pi = rem / a.Wo;
        pj = rem - pi * a.Wo;
    }

    float areg[AL], breg[BL];
    unsigned amask = 0, bmask = 0;
    auto load_slab = [&]() {
        amask = 0;
        bmask = 0;
        const bool pv = pcur < a.P;
        const int sy = pi * a.stride, sx = pj * a.stride;
        const float* xb = a.x + (long)pn * a.x_nstride;
        const float* yb = a.dy + (long)pn * a.y_nstride + (pi * a.Wo + pj);
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            const int row = lg + q * 16;
            const int off = rowoff[row], dij = rowdij[row];
            const int y = sy + (dij >> 16), x = sx + (int)(short)(dij & 0xffff);
            const bool ok = pv && off >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            areg[q] = xb[ok ? off + y * a.W + x : 0];
            amask |= (ok ? 1u : 0u) << q;
        }
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            const int co = n0 + lg + q * 16;
            const bool ok = pv && co < a.K;
            breg[q] = yb[ok ? (long)co * HoWo : 0L];
            bmask |= (ok ? 1u : 0u) << q;
        }
        // advance the cursor by one slab
        pcur += BKP;
        pj += BKP;
        while (pj >= a.Wo) {
            pj -= a.Wo;
            if (++pi >= a.Ho) {
                pi = 0;
                ++pn;
            }
        }
        if (pcur >= a.P) { pn = 0; pi = 0; pj = 0; }
    };
    auto store_slab = [&](int buf) {
        float* Ab = As + buf * BKP * LDA + lp * LDA;
        float* Bb = Bs + buf * BKP * LDB + lp * LDB;
#pragma unroll
        for (int q = 0; q < AL; ++q) Ab[lg + q * 16] = ((amask >> q) & 1u) ? areg[q] : 0.f;
#pragma unroll
        for (int q = 0; q < BL; ++q) Bb[lg + q * 16] = ((bmask >> q) & 1u) ? breg[q] : 0.f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_k = lane >> 5, frag_i = lane & 31;
    if (s_begin < s_end) {
        load_slab();
        store_slab(0);
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more && a.debug < 1) load_slab();
        const float* Ab = As + buf * BKP * LDA + wm * (BM / WM) + frag_i;
        const float* Bb = Bs + buf * BKP * LDB + wn * (BN / WN) + frag_i;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[frag_k * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bb[frag_k * LDB + j * 32];
#pragma unroll
        for (int ks = 0; ks < BKP / 2; ++ks) {
            if (ks + 1 < BKP / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(ks + 1) & 1][i] = Ab[((ks + 1) * 2 + frag_k) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(ks + 1) & 1][j] = Bb[((ks + 1) * 2 + frag_k) * LDB + j * 32];
            }
            GHM_FRAG_FENCE();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            if (ks == BKP / 2 - 3 && more && a.debug < 2) store_slab(buf ^ 1);
        }
        __syncthreads();
    }

    float* ob = a.out + (long)blockIdx.z * a.split_stride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / WN) + j * 32 + frag_i;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * frag_k;
                if (row < a.CT) {
                    float* o = ob + (long)row * a.K + col;
                    float v = acc[i][j][e];
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient with an LDS-staged input PATCH (the im2col tile never exists in HBM or in registers):
// for one 16-pixel slab of an output row, the block stages the CB-channel input patch it touches
// (kh rows x (15*stride + kw) columns per channel) once, and every MFMA "A" fragment (rows = (channel, tap))
// is read from the patch at base(lane) + pixel*stride, base = c*PS + a*PW + b.  PW % 32 == kw and
// PS % 32 == kh*kw make the 32 lanes of a fragment read hit 32 different banks.  Compared with
// wgrad_kernel this loads each input element once per slab instead of kh*kw times and needs no per-element
// tap decode.  Requires Wo % 16 == 0 (every layer with Wo >= 16 in this workload).
// ------------------------------------------------------------------------------------------------
template <int KS, int ST, int BN, int WM, int WN, int BKP>
__global__ __launch_bounds__(256, (BN >= 128 || ST == 2) ? 3 : 4) void wgrad_patch_kernel(const WgradArgs a) {
    constexpr int T = KS * KS;
    constexpr int CB = 128 / T;                 // channels per row tile (5 for 5x5, 14 for 3x3)
    constexpr int ROWS = CB * T;                // valid rows of the 128-row tile
    constexpr int BM = 128;
    constexpr int PWN = (BKP - 1) * ST + KS;    // patch columns actually needed
    constexpr int PW = ((PWN - KS + 31) / 32) * 32 + KS;   // >= PWN and == KS (mod 32)
    constexpr int PS = KS * PW;                 // == T (mod 32)
    static_assert(PW >= PWN && PW % 32 == KS && PS % 32 == T % 32, "patch strides");
    constexpr int PATCH = ((CB * PS + 3) / 4) * 4;
    constexpr int NEL = CB * KS * PWN;          // patch elements to fetch per slab
    constexpr int AL = (NEL + 255) / 256;
    constexpr int LDB = BN + 2;               // 4*LDB = 8 (mod 32): the transposed dy-tile stores spread over all banks
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int PV = BKP / 4;                 // float4 per filter row of the dy tile
    constexpr int CPP = 256 / PV;               // filters covered per pass
    constexpr int BV = (BN + CPP - 1) / CPP;    // float4 loads of the dy tile per thread
    __shared__ __attribute__((aligned(16))) float smem[2 * (PATCH + BKP * LDB)];
    float* Ps = smem;
    float* Bs = smem + 2 * PATCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // The row tiles (x) of one (filter tile, pixel split) read the SAME dy tile.  Workgroups go to the 8 XCDs round
    // robin, so in dispatch order those tiles would fetch it into 8 different L2s; remapped, the blocks that share a
    // dy tile are neighbours on one XCD (profiles/r01_pmc_traffic.json: 4-7x the operand bytes fetched before).
    const int Lg = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int bx = Lg % gridDim.x, by = (Lg / gridDim.x) % gridDim.y, bz = Lg / (gridDim.x * gridDim.y);
    const int c0 = bx * CB, n0 = by * BN;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int slabs_per_row = a.Wo / BKP;
    const int total_slabs = a.N * a.Ho * slabs_per_row;
    const int s_begin = bz * a.slabs_per_split;
    const int s_end = min(s_begin + a.slabs_per_split, total_slabs);

    // ---- slab order: COLUMN STRIPS.  Slab s = (image sn, strip sjb, output row si) with si fastest, so consecutive
    // slabs of a block move straight down the image: every per-lane source offset is an invariant of the whole kernel,
    // the column bounds of an element are invariants of the strip, and a slab costs one uniform pointer bump (+ one
    // uniform "are all kh rows inside the image" test) instead of per-element address and bounds arithmetic
    // (that arithmetic was most of the 12 % the global loads cost this kernel, GHM_ABLATE).
    // Patch element e = (c, ta, col): LDS offset c*PS + ta*PW + col, source offset (c0+c)*HW + ta*W + col relative to
    // rowbase = x[sn] + (si*ST - pad)*W + (sjb*BKP*ST - pad).
    unsigned el_off[AL];                             // source offset; SAFE (the slab's first pixel) when the column is outside
    unsigned el_desc[AL];                            // LDS offset | ta << 16
    unsigned xmask = 0;                              // bit q: the column of element q is inside the image (per strip)
    const unsigned SAFE = (unsigned)(a.pad * a.W + a.pad);
    auto strip_setup = [&](int sjb) {
        xmask = 0;
        const int x0 = sjb * BKP * ST;
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            const int e = min(tid + q * 256, NEL - 1);       // idle lanes repeat the last element (same value, same slot)
            const int c = e / (KS * PWN), r = e - c * (KS * PWN);
            const int ta = r / PWN, col = r - ta * PWN;
            const bool xok = (c0 + c) < a.C && (unsigned)(x0 + col - a.pad) < (unsigned)a.W;
            el_off[q] = xok ? (unsigned)((c0 + c) * HW + ta * a.W + col) : SAFE;
            el_desc[q] = (unsigned)(c * PS + ta * PW + col) | ((unsigned)ta << 16);
            xmask |= (xok ? 1u : 0u) << q;
        }
    };
    // ---- per-thread constants of the dy tile fetch (float4 along pixels) ----
    const int b_p4 = tid % PV, b_co = tid / PV;          // PV x float4 cover the slab's pixels
    unsigned dy_off[BV];
    unsigned dymask = 0;
#pragma unroll
    for (int q = 0; q < BV; ++q) {
        const int col = b_co + q * CPP, co = n0 + col;
        const bool ok = col < BN && co < a.K;
        dy_off[q] = (ok ? (unsigned)co * (unsigned)HoWo : 0u) + b_p4 * 4;
        dymask |= (ok ? 1u : 0u) << q;
    }
    // ---- fragment bases ----
    const int frag_k = lane >> 5, frag_i = lane & 31;
    int abase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + frag_i;
        const int rr = row < ROWS ? row : 0;
        const int c = rr / T, tap = rr - c * T;
        abase[i] = c * PS + (tap / KS) * PW + (tap % KS) + frag_k * ST;
    }

    // slab cursor (uniform): image sn, strip sjb, output row si
    int sn, sjb, si;
    {
        const int per_img = slabs_per_row * a.Ho;
        sn = s_begin / per_img;
        const int rem = s_begin - sn * per_img;
        sjb = rem / a.Ho;
        si = rem - sjb * a.Ho;
    }
    strip_setup(sjb);

    float areg[AL];
    float4 breg[BV];
    unsigned amask = 0;
    auto load_slab = [&]() {
        const int y0 = si * ST;
        const float* rowbase = a.x + (long)sn * a.x_nstride + ((long)(y0 - a.pad) * a.W + (sjb * BKP * ST - a.pad));
        if (y0 >= a.pad && y0 + KS - 1 - a.pad < a.H) {          // every filter row inside the image (uniform)
            amask = xmask;
#pragma unroll
            for (int q = 0; q < AL; ++q) areg[q] = rowbase[el_off[q]];
        } else {
            amask = 0;
#pragma unroll
            for (int q = 0; q < AL; ++q) {
                const int ta = (int)(el_desc[q] >> 16);
                const bool ok = ((xmask >> q) & 1u) && (unsigned)(y0 + ta - a.pad) < (unsigned)a.H;
                areg[q] = rowbase[ok ? el_off[q] : SAFE];
                amask |= (ok ? 1u : 0u) << q;
            }
        }
        const float* yb = a.dy + (long)sn * a.y_nstride + (si * a.Wo + sjb * BKP);
#pragma unroll
        for (int q = 0; q < BV; ++q) breg[q] = *reinterpret_cast<const float4*>(yb + dy_off[q]);
        if (++si >= a.Ho) {
            si = 0;
            if (++sjb >= slabs_per_row) {
                sjb = 0;
                ++sn;
            }
            if (sn >= a.N) sn = 0;                               // past the end: keep addresses valid
            strip_setup(sjb);
        }
    };
    auto store_slab = [&](int buf) {
        float* Pb = Ps + buf * PATCH;
        float* Bb = Bs + buf * BKP * LDB;
#pragma unroll
        for (int q = 0; q < AL; ++q) Pb[el_desc[q] & 0xffffu] = ((amask >> q) & 1u) ? areg[q] : 0.f;
#pragma unroll
        for (int q = 0; q < BV; ++q) {
            const int col = b_co + q * CPP;
            if (col >= BN) continue;
            const bool ok = (dymask >> q) & 1u;
            float* d = Bb + (b_p4 * 4) * LDB + col;
            d[0 * LDB] = ok ? breg[q].x : 0.f;
            d[1 * LDB] = ok ? breg[q].y : 0.f;
            d[2 * LDB] = ok ? breg[q].z : 0.f;
            d[3 * LDB] = ok ? breg[q].w : 0.f;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (s_begin < s_end) {
        load_slab();
        store_slab(0);
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more && a.debug < 1) load_slab();
        const float* Pb = Ps + buf * PATCH;
        const float* Bb = Bs + buf * BKP * LDB + wn * (BN / WN) + frag_i + frag_k * LDB;
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Pb[abase[i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bb[j * 32];
#pragma unroll
        for (int ks = 0; ks < BKP / 2; ++ks) {
            if (ks + 1 < BKP / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(ks + 1) & 1][i] = Pb[abase[i] + (ks + 1) * 2 * ST];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(ks + 1) & 1][j] = Bb[(ks + 1) * 2 * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            if (ks + 1 < BKP / 2) GHM_INTERLEAVE(TM * TN);
            if (ks == BKP / 2 - 3 && more && a.debug < 2) store_slab(buf ^ 1);
        }
        __syncthreads();
    }

    float* ob = a.out + (long)bz * a.split_stride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / WN) + j * 32 + frag_i;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rt = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * frag_k;
                const int row = c0 * T + rt;
                if (rt < ROWS && row < a.CT) {
                    float* o = ob + (long)row * a.K + col;
                    float v = acc[i][j][e];
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
        }
    }
}

// sum of S partial slices: 4 consecutive elements per thread (16-B loads), 4 slices in flight per thread
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ part, int S, long n, long split_stride,
                                                            float* __restrict__ out, int accumulate) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n && (split_stride & 3) == 0) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        int k = 0;
        for (; k + 3 < S; k += 4) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)(k + 0) * split_stride + i);
            const float4 b = *reinterpret_cast<const float4*>(part + (long)(k + 1) * split_stride + i);
            const float4 c = *reinterpret_cast<const float4*>(part + (long)(k + 2) * split_stride + i);
            const float4 d = *reinterpret_cast<const float4*>(part + (long)(k + 3) * split_stride + i);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; k < S; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)k * split_stride + i);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
        float4 r = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                               (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
        float4* o = reinterpret_cast<float4*>(out + i);
        if (accumulate) { const float4 t = *o; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        *o = r;
    } else {
        for (long e = i; e < n && e < i + 4; ++e) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += part[(long)k * split_stride + e];
            if (accumulate) s += out[e];
            out[e] = s;
        }
    }
}

// the same sum for MANY slices (the low-precision weight gradient splits a layer into up to 256 of them): one thread per
// four elements walked S strided loads in a row, four in flight -- 64 dependent round trips to L2 / HBM, 30-60 us of pure
// latency on a gradient of 0.6 MB.  Here G threads share the four elements: thread (tx, ty) sums slices ty, ty + G, ...
// (four in flight), the G partial sums meet in LDS and are added in fixed order (ty = 0 .. G-1): bit-repeatable.
template <int G>
__global__ __launch_bounds__(256) void reduce_splits_wide_kernel(const float* __restrict__ part, int S, long n, long split_stride,
                                                                 float* __restrict__ out, int accumulate) {
    constexpr int EPB = 256 / G;
    __shared__ float4 red[G][EPB];
    const int tx = threadIdx.x % EPB, ty = threadIdx.x / EPB;
    const long i = ((long)blockIdx.x * EPB + tx) * 4;
    const bool vec = i + 3 < n && (split_stride & 3) == 0;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (vec) {
        int k = ty;
        for (; k + 3 * G < S; k += 4 * G) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)(k + 0 * G) * split_stride + i);
            const float4 b = *reinterpret_cast<const float4*>(part + (long)(k + 1 * G) * split_stride + i);
            const float4 c = *reinterpret_cast<const float4*>(part + (long)(k + 2 * G) * split_stride + i);
            const float4 d = *reinterpret_cast<const float4*>(part + (long)(k + 3 * G) * split_stride + i);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; k < S; k += G) {
            const float4 a = *reinterpret_cast<const float4*>(part + (long)k * split_stride + i);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    } else if (i < n) {
        float* acc = &s0.x;
        for (int e = 0; e < 4 && i + e < n; ++e)
            for (int k = ty; k < S; k += G) acc[e] += part[(long)k * split_stride + i + e];
    }
    red[ty][tx] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                              (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (ty == 0 && i < n) {
        float4 r = red[0][tx];
#pragma unroll
        for (int g = 1; g < G; ++g) { const float4 t = red[g][tx]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        if (vec) {
            float4* o = reinterpret_cast<float4*>(out + i);
            if (accumulate) { const float4 t = *o; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
            *o = r;
        } else {
            const float* rr = &r.x;
            for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = (accumulate ? out[i + e] : 0.f) + rr[e];
        }
    }
}

static int launch_reduce_splits(ghm_ctx* ctx, const float* part, int S, long n, long split_stride, float* out, int accumulate) {
    const long groups = (n + 3) / 4;
    if (S >= 64 && !GHM_OPT("GHM_NO_WIDE_REDUCE"))
        hipLaunchKernelGGL((reduce_splits_wide_kernel<8>), dim3(ceil_div(groups, 32)), dim3(256), 0, ctx->stream, part, S, n,
                           split_stride, out, accumulate);
    else if (S >= 16 && !GHM_OPT("GHM_NO_WIDE_REDUCE"))
        hipLaunchKernelGGL((reduce_splits_wide_kernel<4>), dim3(ceil_div(groups, 64)), dim3(256), 0, ctx->stream, part, S, n,
                           split_stride, out, accumulate);
    else
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(ceil_div(groups, 256)), dim3(256), 0, ctx->stream, part, S, n, split_stride,
                           out, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

// per-channel sum over (n, hw): bias gradients. grid (S, C) partials, then one thread per channel.
__global__ __launch_bounds__(256) void channel_sum_partial(const float* __restrict__ x, int N, int HW, long nstride,
                                                           int S, float* __restrict__ part, float* out, int accumulate) {
    const int c = blockIdx.y, sidx = blockIdx.x;
    const long total = (long)N * HW;
    const long chunk = ((total + S - 1) / S + 3) & ~3L;
    const long lo = sidx * chunk, hi = min(lo + chunk, total);
    float s = 0.f;
    if ((HW & 3) == 0 && (nstride & 3) == 0) {
        // four 16-byte loads in flight per thread (one per iteration left a wave a single load deep: 2.4 TB/s on the 67 MB
        // gradients of the 256^2 layers); the four partial sums are added in a fixed order
        const float* xc = x + (long)c * HW;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (long e = lo + threadIdx.x * 4; e < hi; e += 4096) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long ek = e + k * 1024;
                const long n = ek / HW, i = ek - n * HW;
                v[k] = ek < hi ? *reinterpret_cast<const float4*>(xc + n * nstride + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    } else {
        for (long e = lo + threadIdx.x; e < hi; e += 256) {
            const long n = e / HW, i = e - n * HW;
            s += x[n * nstride + (long)c * HW + i];
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (out) out[c] = (accumulate ? out[c] : 0.f) + t;      // S == 1: the block's sum IS the channel's (no second launch)
        else part[(long)c * S + sidx] = t;
    }
}

__global__ void channel_sum_final(const float* __restrict__ part, int C, int S, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(long)c * S + k];
    out[c] = (accumulate ? out[c] : 0.f) + s;
}


// ------------------------------------------------------------------------------------------------
// "taps as GEMM rows" path for stride-1 convolutions with <= 4 channels on one side and many pixels
// (g_out forward / weight gradient, data gradient of the 1-channel first layer).  A k x k convolution with
// R output channels is a 1x1 convolution with R*k*k outputs (MFMA, rows = (tap, r)) followed by a
// shift-and-add of the k*k planes; the weight gradient is the transposed construction (shift-expand the
// output gradient into k*k planes, then a 1x1 weight-gradient GEMM).  The packed layout wp[c][tap][r]
// already is the [C][taps*R] matrix these GEMMs need.
// ------------------------------------------------------------------------------------------------
struct ShiftArgs {
    const float* src;      // planes [N, M*T, H, W]   (plane index = outer*T*inner ... see kernels)
    float* dst;
    const float* bias;
    int N, R, T, H, W;     // R channels, T taps, plane size H x W
    long dst_nstride;
    int act;
    float alpha;
    int accumulate;
    int tap_major;         // 1: plane = tap*R + r (forward)   0: plane = r*T + tap (data gradient)
    int sign;              // +1: read at p + d (forward)      -1: read at p - d (data gradient)
    int di[MAX_TAPS], dj[MAX_TAPS];
};

__global__ __launch_bounds__(256) void shift_add_kernel(const ShiftArgs a) {
    const long hw = (long)a.H * a.W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.N * a.R * hw) return;
    const long nr = idx / hw;
    const int rem = (int)(idx - nr * hw), y = rem / a.W, x = rem - y * a.W;
    const int n = (int)(nr / a.R), r = (int)(nr - (long)n * a.R);
    const float* sb = a.src + (long)n * a.R * a.T * hw;
    float v = a.bias ? a.bias[r] : 0.f;
    for (int t = 0; t < a.T; ++t) {
        const int yy = y + a.sign * a.di[t], xx = x + a.sign * a.dj[t];
        if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
            const int plane = a.tap_major ? t * a.R + r : r * a.T + t;
            v += sb[(long)plane * hw + (long)yy * a.W + xx];
        }
    }
    float* o = a.dst + (long)n * a.dst_nstride + (long)r * hw + rem;
    if (a.accumulate) v += *o;
    *o = ghm_act(v, a.act, a.alpha);
}

// E[n, tap*K + k, p'] = dy[n, k, p' - d_tap]  (zero outside)
__global__ __launch_bounds__(256) void shift_expand_kernel(const float* __restrict__ dy, long dy_nstride, float* __restrict__ E,
                                                           int N, int K, int T, int H, int W, ShiftArgs taps) {
    const long hw = (long)H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * T * K * hw) return;
    const long plane = idx / hw;
    const int rem = (int)(idx - plane * hw), y = rem / W, x = rem - y * W;
    const int n = (int)(plane / (T * K)), tk = (int)(plane - (long)n * T * K);
    const int t = tk / K, k = tk - t * K;
    const int yy = y - taps.di[t], xx = x - taps.dj[t];
    float v = 0.f;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
        v = dy[(long)n * dy_nstride + (long)k * hw + (long)yy * W + xx];
    E[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// host side: variant choice and launch
// ------------------------------------------------------------------------------------------------
namespace {

// Fully-connected layer on a handful of samples (the generator's first layer: z [N, 1000] -> [N, 8192], dcgan.py:17-20):
// P = N <= 8 output "pixels", one tap, forward weight layout wp[ch][r].  The MFMA tiles have nothing to hold on to here
// (4 pixels of a 64-pixel tile, 64 blocks walking 1000 channels each: 0.12 ms, latency-bound, at the head of the step);
// this is a weight-streaming GEMV: thread = output row r (lanes along r: 256-byte rows of W per wave and channel), the
// sample values are wave-uniform scalar loads, the channel range is sliced over blockIdx.y and the slices are summed in
// fixed order by the split-K epilogue (bias, activation, scatter).  Algorithmic bytes: the weight matrix once.
template <int PMAX>
__global__ __launch_bounds__(256) void dense_smallp_kernel(const IgemmArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int P = a.N * a.Hs * a.Ws, hw_s = a.Hs * a.Ws;
    const int ch_begin = blockIdx.y * a.slabs_per_split;
    const int ch_end = min(a.CH, ch_begin + a.slabs_per_split);
    const long HinWin = (long)a.Hin * a.Win;
    long src[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        const int n = p / hw_s, rem = p - n * hw_s, uu = rem / a.Ws, vv = rem - uu * a.Ws;
        src[p] = p < P ? (long)n * a.in_nstride + (long)(uu * a.ss + a.di[0]) * a.Win + (vv * a.ss + a.dj[0]) : 0;
    }
    float acc[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) acc[p] = 0.f;
    if (r < a.R) {
        const float* w = a.wp + ((long)ch_begin * a.T + a.wi[0]) * a.R + r;
        const long wstep = (long)a.T * a.R;
#pragma unroll 8
        for (int ch = ch_begin; ch < ch_end; ++ch, w += wstep) {
            const float wv = *w;
#pragma unroll
            for (int p = 0; p < PMAX; ++p) acc[p] = fmaf(wv, a.in[src[p] + ch * HinWin], acc[p]);
        }
        float* o = a.partial + ((long)blockIdx.y * a.R + r) * P;
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            if (p < P) o[p] = acc[p];
    }
}

struct Variant {
    int bm, bn;
};

Variant pick_variant(int R, long P, int num_cu) {
    int bm = R >= 96 ? 128 : (R >= 48 ? 64 : 32);
    int big, small;
    if (bm == 128) { big = 128; small = 64; }
    else if (bm == 64) { big = 256; small = 64; }
    else { big = 256; small = 128; }
    // the wide pixel tile is the efficient one (measured: 128x128 + split-K beats 128x64 whenever the layer has
    // at least one full wide tile); the narrow tile only serves the 1x1 .. 8x8 maps
    int bn = (P >= big) ? big : small;
    if (const char* f = GHM_OPT("GHM_FORCE_TILE")) {     // test knob: exercise both pixel-tile widths
        if (f[0] == 'b') bn = big;
        if (f[0] == 's') bn = small;
    }
    return {bm, bn};
}

// may the split-K form of this launch end in the shared finishing kernel of conv_small.hip with the BatchNorm inside it?
// (forward convolution on its own output grid, whole map of a channel <= 1024 pixels, rows in whole q-unit groups)
static bool igemm_bn_geometry(const IgemmArgs& a) {
    const long P = (long)a.N * a.Hs * a.Ws;
    return a.os == 1 && a.Hs == a.Hout && a.Ws == a.Wout && a.R % 8 == 0 && a.R > 4 && P <= 1024 && P > 0;
}

template <bool WT>
int launch_igemm(ghm_ctx* ctx, const IgemmArgs& a_in, const SmBn* bn = nullptr) {
    IgemmArgs a = a_in;
    const long P = (long)a.N * a.Hs * a.Ws;
    if (P == 0 || a.R == 0) return 0;
    a.partial = nullptr;
    a.tickets = nullptr;
    a.nsplit = 0;
    a.slabs_per_split = 1 << 30;
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    if (a.R <= 4) {
        a.slabs_per_split = a.CH;
        if (P <= 65536 && a.CH >= 64 && GHM_OPT("GHM_NO_SMALLR_SPLIT") == nullptr) {
            // few output pixels, long channel reduction: slices of >= 8 channels until ~4 blocks per CU are in flight
            int S = (4 * ctx->num_cu + ceil_div(P, 256) - 1) / ceil_div(P, 256);
            if (S > a.CH / 8) S = a.CH / 8;
            if (S > 1) {
                a.slabs_per_split = ceil_div(a.CH, S);
                S = ceil_div(a.CH, a.slabs_per_split);
                void* ws = nullptr;
                if (int e = ghm_scratch(ctx, (size_t)S * a.R * P * sizeof(float), &ws)) return e;
                a.partial = (float*)ws;
                hipLaunchKernelGGL((direct_smallr_kernel<WT>), dim3(ceil_div(P, 256), S), dim3(256), 0, ctx->stream, a);
                GHM_LAUNCH_CHECK();
                hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ceil_div(P * a.R, 256)), dim3(256), 0, ctx->stream, a, S);
                GHM_LAUNCH_CHECK();
                return 0;
            }
        }
        if (P <= 16384 && (long)a.ntaps * a.CH >= 256) {
            hipLaunchKernelGGL((direct_smallr_wave_kernel<WT>), dim3(ceil_div(P, 4)), dim3(256), 0, ctx->stream, a);
        } else {
            hipLaunchKernelGGL((direct_smallr_kernel<WT>), dim3(ceil_div(P, 256)), dim3(256), 0, ctx->stream, a);
        }
        GHM_LAUNCH_CHECK();
        return 0;
    }
    if (!WT && a.ntaps == 1 && P <= 8 && a.R >= 1024 && a.CH >= 64 && a.os == 1 && GHM_OPT("GHM_NO_DENSE_SMALLP") == nullptr) {
        // in range by construction (a 1x1 'valid' tap): no bounds test in the kernel
        GHM_CHECK(a.di[0] >= 0 && a.dj[0] >= 0 && (a.Hs - 1) * a.ss + a.di[0] < a.Hin && (a.Ws - 1) * a.ss + a.dj[0] < a.Win,
                  "dense_smallp: tap outside the input");
        int S = (3 * ctx->num_cu) / ceil_div(a.R, 256);                 // ~3 blocks per CU
        if (S > a.CH / 16) S = a.CH / 16;
        if (S < 1) S = 1;
        a.slabs_per_split = ceil_div(a.CH, S);
        S = ceil_div(a.CH, a.slabs_per_split);
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, (size_t)S * a.R * P * sizeof(float), &ws)) return e;
        a.partial = (float*)ws;
        const dim3 g(ceil_div(a.R, 256), S);
        if (P <= 4)
            hipLaunchKernelGGL((dense_smallp_kernel<4>), g, dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((dense_smallp_kernel<8>), g, dim3(256), 0, ctx->stream, a);
        GHM_LAUNCH_CHECK();
        hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ceil_div(P * a.R, 256)), dim3(256), 0, ctx->stream, a, S);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    const Variant v = pick_variant(a.R, P, ctx->num_cu);
    const bool fast = (a.CH % 16) == 0 && (WT || (a.R % 4) == 0);
    const int grid = ceil_div(a.R, v.bm) * ceil_div(P, v.bn);
    // split-K over the (tap, channel) slabs when the output tiles alone cannot fill the chip
    const int nslabs = ceil_div((long)a.ntaps * a.CH, 16);
    int splits = 1;
    if (grid < ctx->num_cu + ctx->num_cu / 2) {
        splits = (4 * ctx->num_cu + grid - 1) / grid;                // aim at ~4 resident blocks per CU
        const int max_by_work = nslabs / 8 > 0 ? nslabs / 8 : 1;     // keep >= 8 slabs (128 k) per slice
        if (splits > max_by_work) splits = max_by_work;
    }
    if (const char* f = GHM_OPT("GHM_FORCE_SPLITK")) splits = atoi(f) < nslabs ? atoi(f) : nslabs;
    if (bn) {           // the BatchNorm epilogue lives in the finishing kernel: at least two slices
        GHM_CHECK(igemm_bn_geometry(a) && nslabs >= 2, "igemm: BatchNorm epilogue not served for this geometry");
        if (splits < 2) splits = 2;
    }
    if (splits > 1) {
        a.slabs_per_split = ceil_div(nslabs, splits);
        splits = ceil_div(nslabs, a.slabs_per_split);
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, (size_t)splits * a.R * P * sizeof(float), &ws)) return e;
        a.partial = (float*)ws;
        a.nsplit = splits;
        a.tickets = bn ? nullptr : ghm_tickets(ctx, grid);
    }
    const bool folded = a.tickets != nullptr;
    const dim3 g(grid, splits);
#define GHM_IGEMM_CASE(BM_, BN_, WM_, WN_)                                                             \
    if (v.bm == BM_ && v.bn == BN_) {                                                                  \
        if (fast)                                                                                      \
            hipLaunchKernelGGL((igemm_kernel<BM_, BN_, WM_, WN_, WT, true>), g, dim3(256), 0, ctx->stream, a);  \
        else                                                                                           \
            hipLaunchKernelGGL((igemm_kernel<BM_, BN_, WM_, WN_, WT, false>), g, dim3(256), 0, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                            \
        if (bn)         /* sum of the slices + bias, batch statistics, normalise, activation: ONE launch */ \
            return sm_finish_launch(ctx, a.partial, splits, a.R, a.N, a.Hs * a.Ws, a.bias, a.out, a.out_nstride, 0, a.act,  \
                                    a.alpha, nullptr, 0, GHM_DTYPE_BF16, bn);                          \
        if (splits > 1 && !folded) {                                                                   \
            hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ceil_div(P * a.R, 256)), dim3(256), 0, ctx->stream, a, \
                               splits);                                                                \
            GHM_LAUNCH_CHECK();                                                                        \
        }                                                                                              \
        return 0;                                                                                      \
    }
    GHM_IGEMM_CASE(128, 128, 2, 2)
    GHM_IGEMM_CASE(128, 64, 2, 2)
    GHM_IGEMM_CASE(64, 256, 1, 4)
    GHM_IGEMM_CASE(64, 64, 2, 2)
    GHM_IGEMM_CASE(32, 256, 1, 4)
    GHM_IGEMM_CASE(32, 128, 1, 4)
#undef GHM_IGEMM_CASE
    ghm_set_error("no igemm variant for bm=%d bn=%d", v.bm, v.bn);
    return -3;
}


// the four parity classes of a stride-2 data gradient (same R, same sub-grid) through igemm_kernel_x4
int launch_igemm_x4(ghm_ctx* ctx, IgemmArgs4& a4) {
    const IgemmArgs& a0 = a4.c[0];
    const long P = (long)a0.N * a0.Hs * a0.Ws;
    if (P == 0 || a0.R == 0) return 0;
    const Variant v = pick_variant(a0.R, P, ctx->num_cu);
    const bool fast = (a0.CH % 16) == 0;
    const int grid = ceil_div(a0.R, v.bm) * ceil_div(P, v.bn);
    int smax = 1;
    size_t total = 0;
    for (int c = 0; c < 4; ++c) {
        IgemmArgs& a = a4.c[c];
        const int nslabs = ceil_div((long)a.ntaps * a.CH, 16);
        int splits = 1;
        if (4 * grid < ctx->num_cu + ctx->num_cu / 2) {                   // the four classes fill the chip together
            splits = (4 * ctx->num_cu + 4 * grid - 1) / (4 * grid);
            const int max_by_work = nslabs / 8 > 0 ? nslabs / 8 : 1;
            if (splits > max_by_work) splits = max_by_work;
        }
        if (const char* f = GHM_OPT("GHM_FORCE_SPLITK")) splits = atoi(f) < nslabs ? atoi(f) : nslabs;
        if (splits < 1) splits = 1;
        a.slabs_per_split = ceil_div(nslabs, splits);
        splits = ceil_div(nslabs, a.slabs_per_split);
        a4.nsplit[c] = splits;
        if (splits > smax) smax = splits;
        total += (size_t)splits * a.R * P;
        if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    }
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, total * sizeof(float), &ws)) return e;
    float* part = (float*)ws;
    int* const tk = ghm_tickets(ctx, 4 * grid);
    for (int c = 0; c < 4; ++c) {                   // every class goes through its partial slices (reduced by the last block
        a4.c[c].partial = part;                     // to arrive at each output tile, or by one epilogue launch for all)
        a4.c[c].tickets = tk;
        a4.c[c].nsplit = a4.nsplit[c];
        part += (size_t)a4.nsplit[c] * a4.c[c].R * P;
    }
    const dim3 g(grid, smax, 4);
#define GHM_IGEMM4_CASE(BM_, BN_, WM_, WN_)                                                                      \
    if (v.bm == BM_ && v.bn == BN_) {                                                                            \
        if (fast)                                                                                                \
            hipLaunchKernelGGL((igemm_kernel_x4<BM_, BN_, WM_, WN_, true>), g, dim3(256), 0, ctx->stream, a4);    \
        else                                                                                                     \
            hipLaunchKernelGGL((igemm_kernel_x4<BM_, BN_, WM_, WN_, false>), g, dim3(256), 0, ctx->stream, a4);   \
        GHM_LAUNCH_CHECK();                                                                                      \
        if (!tk) {                                                                                               \
            hipLaunchKernelGGL(igemm_splitk_epilogue_x4, dim3(ceil_div(P * a0.R, 256), 4), dim3(256), 0, ctx->stream, a4); \
            GHM_LAUNCH_CHECK();                                                                                  \
        }                                                                                                        \
        return 0;                                                                                                \
    }
    GHM_IGEMM4_CASE(128, 128, 2, 2)
    GHM_IGEMM4_CASE(128, 64, 2, 2)
    GHM_IGEMM4_CASE(64, 256, 1, 4)
    GHM_IGEMM4_CASE(64, 64, 2, 2)
    GHM_IGEMM4_CASE(32, 256, 1, 4)
    GHM_IGEMM4_CASE(32, 128, 1, 4)
#undef GHM_IGEMM4_CASE
    ghm_set_error("no igemm variant for bm=%d bn=%d", v.bm, v.bn);
    return -3;
}

// does this geometry take the taps-as-rows path?  (stride 1, 'same'-style geometry, many pixels)
bool taps_as_rows(const ghm_conv_desc* d, int small_side) {
    return small_side <= 4 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W && d->kh * d->kw > 1 &&
           (long)d->N * d->H * d->W >= 32768 && GHM_OPT("GHM_NO_TAPROWS") == nullptr;
}

void fill_taps(const ghm_conv_desc* d, ShiftArgs& sa) {
    for (int ta = 0; ta < d->kh; ++ta)
        for (int tb = 0; tb < d->kw; ++tb) {
            sa.di[ta * d->kw + tb] = ta - d->pad;
            sa.dj[ta * d->kw + tb] = tb - d->pad;
        }
}

IgemmArgs pointwise_args(const float* in, long in_nstride, int N, int CH, int H, int W, const float* wp, int R,
                         float* out) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.wp = wp; a.bias = nullptr; a.out = out;
    a.N = N; a.CH = CH; a.Hin = H; a.Win = W; a.in_nstride = in_nstride;
    a.R = R; a.Hout = H; a.Wout = W; a.out_nstride = (long)R * H * W;
    a.Hs = H; a.Ws = W; a.os = 1; a.ss = 1; a.T = 1; a.ntaps = 1;
    a.act = GHM_ACT_LINEAR;
    return a;
}


// ---- patch-kernel dispatch (stride-1 forward form) ----
struct PatchPlan {
    bool ok;
    int bm, rt, splits, slabs_per_split, grid;
    size_t lds;
};

PatchPlan plan_patch(int N, int CH, int H, int W, int R, int ks, int num_cu, int st = 1) {
    // H, W: OUTPUT grid
    PatchPlan p;
    p.ok = false;
    if (!(ks == 3 || ks == 5) || GHM_OPT("GHM_NO_PATCH")) return p;
    if (st == 2 && (ks != 3 || GHM_OPT("GHM_NO_PATCH_S2"))) return p;
    // 64 rows x (8 x 32) pixels: 34 KB of LDS -> 4 blocks/CU; measured 3-5 % faster than 128 x (4 x 32) at 2 blocks/CU
    p.bm = 64;
    if (const char* f = GHM_OPT("GHM_PATCH_BM")) p.bm = atoi(f);
    p.rt = p.bm == 128 ? 4 : 8;
    const int cb = ks == 5 ? 2 : 4;
    if (R < 32 || (R & 3) || (W % 32) || (H % p.rt) || (CH % cb) || CH < 2 * cb) return p;
    const int T = ks * ks;
    const int lda = p.bm, ph = (p.rt - 1) * st + ks, pw = 31 * st + ks;
    const int asz = cb * T * lda, psz = ((cb * ph * pw + 3) / 4) * 4;
    p.lds = (size_t)2 * (asz + psz) * sizeof(float);
    // TWO blocks per CU, not the four that 30-34 KB of LDS and 121 VGPRs would allow: the block reserves a third of the
    // CU's LDS + 1 byte.  Alone the kernel loses 0.5-1.2 % (5x5 142.4 -> 140.7 TFLOP/s, it is MFMA-bound at two waves per
    // SIMD), but half of every CU's registers and 50 KB of its LDS stay free for the kernels of the other three streams
    // of the step, which then run BESIDE it instead of queueing for its slots: joint step 163.4 -> 167.0 img/s, HIP-graph
    // form 160.4 -> 164.1, 1024^2 28.8 -> 29.4, batch 8 174.1 -> 176.5 (sweep: 36-52 KB +-0, 54-60 KB +2 %, 66 KB +0.7 %,
    // one block per CU -0.4 %).  The same reservation on wgrad_patch_kernel, dgrad_s2_patch_kernel and the low-precision
    // kernels measured -0.5 to -8 %: only this kernel has the slack.  GHM_PATCH_LDS_MIN overrides (tuning).
    size_t lds_min = 54 * 1024;
    if (const char* f = GHM_OPT("GHM_PATCH_LDS_MIN")) lds_min = (size_t)atol(f);
    if (p.lds < lds_min) p.lds = lds_min;
    const int ntr = (R + p.bm - 1) / p.bm;
    p.grid = ntr * (W / 32) * (H / p.rt) * N;
    const int nslabs = CH / cb;
    p.splits = 1;
    if (p.grid < num_cu + num_cu / 2) {
        p.splits = (4 * num_cu + p.grid - 1) / p.grid;
        const int maxs = nslabs / 4 > 0 ? nslabs / 4 : 1;
        if (p.splits > maxs) p.splits = maxs;
    }
    p.slabs_per_split = (nslabs + p.splits - 1) / p.splits;
    p.splits = (nslabs + p.slabs_per_split - 1) / p.slabs_per_split;
    p.ok = true;
    return p;
}

int launch_patch(ghm_ctx* ctx, const PatchPlan& pl, PatchArgs a, int ks, int st = 1) {
    a.slabs_per_split = pl.slabs_per_split;
    a.partial = nullptr;
    a.zeros = ctx->zeros;
    if (pl.splits > 1) {
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, (size_t)pl.splits * a.R * a.N * a.H * a.W * sizeof(float), &ws)) return e;
        a.partial = (float*)ws;
    }
    const dim3 g(pl.grid, pl.splits);
#define GHM_PATCH_CASE(KS_, BM_, RT_, WM_, WN_, CP_, ST_)                                                     \
    if (ks == KS_ && pl.bm == BM_ && st == ST_) {                                                             \
        hipLaunchKernelGGL((conv_patch_kernel<KS_, BM_, RT_, WM_, WN_, CP_, ST_>), g, dim3(256), pl.lds, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                                   \
    } else
    GHM_PATCH_CASE(5, 128, 4, 2, 2, 1, 1)
    GHM_PATCH_CASE(5, 64, 8, 1, 4, 1, 1)
    GHM_PATCH_CASE(3, 128, 4, 2, 2, 2, 1)
    GHM_PATCH_CASE(3, 64, 8, 1, 4, 2, 1)
    GHM_PATCH_CASE(3, 128, 4, 2, 2, 2, 2)
    GHM_PATCH_CASE(3, 64, 8, 1, 4, 2, 2) {
        ghm_set_error("no conv_patch variant for k=%d bm=%d", ks, pl.bm);
        return -3;
    }
#undef GHM_PATCH_CASE
    if (pl.splits > 1) {
        IgemmArgs e;
        memset(&e, 0, sizeof(e));
        e.partial = a.partial; e.out = a.out; e.bias = a.bias; e.N = a.N; e.R = a.R;
        e.Hout = a.H; e.Wout = a.W; e.out_nstride = a.out_nstride; e.Hs = a.H; e.Ws = a.W; e.os = 1;
        e.act = a.act; e.alpha = a.alpha; e.accumulate = a.accumulate;
        hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ceil_div((long)a.N * a.H * a.W * a.R, 256)), dim3(256), 0,
                           ctx->stream, e, pl.splits);
        GHM_LAUNCH_CHECK();
    }
    return 0;
}

int check_desc(const ghm_conv_desc* d) {
    GHM_CHECK(d->kh * d->kw <= MAX_TAPS, "filter %dx%d exceeds %d taps", d->kh, d->kw, MAX_TAPS);
    GHM_CHECK(d->stride == 1 || d->stride == 2, "stride %d unsupported", d->stride);
    GHM_CHECK(d->Ho == (d->H + 2 * d->pad - d->kh) / d->stride + 1 &&
                  d->Wo == (d->W + 2 * d->pad - d->kw) / d->stride + 1,
              "inconsistent conv geometry H=%d W=%d k=%dx%d s=%d p=%d -> %dx%d", d->H, d->W, d->kh, d->kw,
              d->stride, d->pad, d->Ho, d->Wo);
    GHM_CHECK(d->x_nstride >= (int64_t)d->C * d->H * d->W && d->y_nstride >= (int64_t)d->K * d->Ho * d->Wo,
              "sample strides smaller than the tensors");
    return 0;
}

struct WVariant {
    int bm, bn, splits, slabs_per_split;
    int patch;   // 1: wgrad_patch_kernel (row tile = CB channels x taps)
    int row_tiles;
    int bkp;     // pixels per slab
};

bool wgrad_patch_ok(const ghm_conv_desc* d) {
    const bool k_ok = (d->kh == 3 && d->kw == 3) || (d->kh == 5 && d->kw == 5 && d->stride == 1);
    return k_ok && d->Wo % 16 == 0 && d->C * d->kh * d->kw >= 96 && d->K > 4 && d->pad <= d->kh &&
           d->y_nstride % 4 == 0 && (d->Ho * d->Wo) % 4 == 0 && GHM_OPT("GHM_NO_PATCH") == nullptr;
}

WVariant pick_wgrad(const ghm_conv_desc* d, int num_cu) {
    const int CT = d->C * d->kh * d->kw;
    WVariant v;
    v.patch = wgrad_patch_ok(d) ? 1 : 0;
    v.bm = (CT >= 96 || v.patch) ? 128 : 32;
    v.bn = d->K >= 96 ? 128 : (d->K >= 48 ? 64 : 32);
    if (v.bm == 32) v.bn = 128;
    const int T = d->kh * d->kw;
    v.row_tiles = v.patch ? ceil_div(d->C, 128 / T) : ceil_div(CT, v.bm);
    const long tiles = (long)v.row_tiles * ceil_div(d->K, v.bn);
    const long P = (long)d->N * d->Ho * d->Wo;
    // 32-pixel slabs only for 5x5 (alone: 121 vs 117 TFLOP/s).  3x3 is as fast with 16-pixel slabs (116 vs 115) and then
    // needs 28 KB of LDS and 111-121 VGPRs instead of 45-56 KB and 136-156: more room for the co-running streams (+0.4 % step)
    v.bkp = (v.patch && d->Wo % 32 == 0 && d->kh == 5 && GHM_OPT("GHM_WGRAD_BKP16") == nullptr) ? 32 : 16;
    if (v.patch && d->Wo % 32 == 0 && GHM_OPT("GHM_WGRAD_BKP32")) v.bkp = 32;
    const long slabs = (P + v.bkp - 1) / v.bkp;
    // whole rounds of resident blocks (a partly filled last round costs up to 2x): blocks/CU is 3 for the 128-filter tile,
    // 4 otherwise.  ALONE one round is the optimum (every further split writes and re-reads another partial dW); in the
    // step, where the stage streams' kernels want CUs too, TWO rounds of half-length blocks interleave better -- joint
    // fp32 step, same box (tools/instep_sweep.sh, GHM_WGRAD_ROUNDS): 0.5 rounds 25.3 ms, 1 23.69, 1.5 23.49, 2 23.39,
    // 3 23.41, 4 23.50, 6 23.56.
    long minp = 4096;                                                        // the small maps keep one round
    if (const char* f = GHM_OPT("GHM_WGRAD_ROUNDS_MINP")) minp = atol(f);    // tuning
    double rounds = P >= minp ? 2.0 : 1.0;
    if (const char* f = GHM_OPT("GHM_WGRAD_ROUNDS")) rounds = atof(f);       // tuning
    const long slots = (long)(num_cu * (v.bn >= 128 ? 3 : 4) * rounds);
    long want = slots / tiles;
    if (const char* f = GHM_OPT("GHM_WGRAD_SPLITS")) want = atol(f);
    long max_by_work = slabs / (256 / v.bkp) > 0 ? slabs / (256 / v.bkp) : 1;   // at least 256 pixels per split
    long S = want < max_by_work ? want : max_by_work;
    if (S < 1) S = 1;
    if (S > 1024) S = 1024;
    v.slabs_per_split = (int)((slabs + S - 1) / S);
    v.splits = (int)((slabs + v.slabs_per_split - 1) / v.slabs_per_split);
    return v;
}

}  // namespace

int ghm_splitk_finish(ghm_ctx* ctx, const float* partial, int S, float* out, const float* bias, int N, int R, int H,
                      int W, long out_nstride, int act, float alpha, int accumulate) {
    IgemmArgs e;
    memset(&e, 0, sizeof(e));
    e.partial = const_cast<float*>(partial); e.out = out; e.bias = bias; e.N = N; e.R = R;
    e.Hout = H; e.Wout = W; e.out_nstride = out_nstride; e.Hs = H; e.Ws = W; e.os = 1;
    e.act = act; e.alpha = alpha; e.accumulate = accumulate;
    hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ceil_div((long)N * H * W * R, 256)), dim3(256), 0, ctx->stream, e, S);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_reduce_splits(ghm_ctx* ctx, const float* part, int S, long n, long split_stride, float* out, int accumulate) {
    return launch_reduce_splits(ctx, part, S, n, split_stride, out, accumulate);
}

extern "C" {

static IgemmArgs igemm_fwd_args(const ghm_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act,
                                float alpha, int accumulate) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x; a.wp = wp; a.bias = bias; a.out = y;
    a.N = d->N; a.CH = d->C; a.Hin = d->H; a.Win = d->W; a.in_nstride = d->x_nstride;
    a.R = d->K; a.Hout = d->Ho; a.Wout = d->Wo; a.out_nstride = d->y_nstride;
    a.Hs = d->Ho; a.Ws = d->Wo; a.os = 1; a.ou = 0; a.ov = 0; a.ss = d->stride;
    a.T = d->kh * d->kw; a.ntaps = a.T;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    for (int ta = 0; ta < d->kh; ++ta)
        for (int tb = 0; tb < d->kw; ++tb) {
            const int t = ta * d->kw + tb;
            a.di[t] = ta - d->pad; a.dj[t] = tb - d->pad; a.wi[t] = t;
        }
    return a;
}

int ghm_conv2d_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                   float* y, int32_t act, float alpha, int32_t accumulate) {
    if (int e = check_desc(d)) return e;
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    if (d->C <= 4 && thin_fanout_fwd_ok(d, act)) return thin_fanout_fwd(ctx, d, x, wp, bias, y, act, alpha, accumulate);
    if (d->K <= 4 && thin_fanin_s1_fwd_ok(d)) return thin_fanin_s1_fwd(ctx, d, x, wp, bias, y, act, alpha, accumulate);
    if (taps_as_rows(d, d->K)) {
        const int T = d->kh * d->kw;
        void* ws = nullptr;
        const size_t tb = (size_t)d->N * T * d->K * d->H * d->W * sizeof(float);
        if (int e = ghm_scratch(ctx, tb + (64u << 20), &ws)) return e;     // + room for split-K partials
        float* Tbuf = (float*)((char*)ws + (64u << 20));
        IgemmArgs g = pointwise_args(x, d->x_nstride, d->N, d->C, d->H, d->W, wp, T * d->K, Tbuf);
        if (int e = launch_igemm<false>(ctx, g)) return e;
        ShiftArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.src = Tbuf; sa.dst = y; sa.bias = bias; sa.N = d->N; sa.R = d->K; sa.T = T; sa.H = d->H; sa.W = d->W;
        sa.dst_nstride = d->y_nstride; sa.act = act; sa.alpha = alpha; sa.accumulate = accumulate;
        sa.tap_major = 1; sa.sign = 1;
        fill_taps(d, sa);
        hipLaunchKernelGGL(shift_add_kernel, dim3(ceil_div((long)d->N * d->K * d->H * d->W, 256)), dim3(256), 0,
                           ctx->stream, sa);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    if (d->kh == d->kw && ((d->stride == 1 && d->Ho == d->H && d->Wo == d->W) ||
                           (d->stride == 2 && d->Ho * 2 == d->H && d->Wo * 2 == d->W))) {
        const PatchPlan pl = plan_patch(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, ctx->num_cu, d->stride);
        if (pl.ok) {
            PatchArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.in = x; pa.wp = wp; pa.bias = bias; pa.out = y;
            pa.N = d->N; pa.CH = d->C; pa.H = d->Ho; pa.W = d->Wo; pa.Hin = d->H; pa.Win = d->W;
            pa.in_nstride = d->x_nstride;
            pa.R = d->K; pa.out_nstride = d->y_nstride; pa.pad = d->pad;
            pa.act = act; pa.alpha = alpha; pa.accumulate = accumulate;
            if (const char* f = GHM_OPT("GHM_ABLATE")) pa.debug = atoi(f);
            return launch_patch(ctx, pl, pa, d->kh, d->stride);
        }
    }
    const IgemmArgs a = igemm_fwd_args(d, x, wp, bias, y, act, alpha, accumulate);
    return launch_igemm<false>(ctx, a);
}

// which forward convolutions run on the generic gather kernel (none of the thin / taps-as-rows / LDS-patch forms above)
static bool igemm_fwd_generic(const ghm_conv_desc* d, int act) {
    if (d->C <= 4 && thin_fanout_fwd_ok(d, act)) return false;
    if (d->K <= 4 && thin_fanin_s1_fwd_ok(d)) return false;
    if (taps_as_rows(d, d->K)) return false;
    if (d->kh == d->kw && ((d->stride == 1 && d->Ho == d->H && d->Wo == d->W) ||
                           (d->stride == 2 && d->Ho * 2 == d->H && d->Wo * 2 == d->W)) &&
        plan_patch(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, ghm_plan_cus(), d->stride).ok)
        return false;
    return d->kh * d->kw <= MAX_TAPS;
}

// fp32 Conv2DLayer -> BatchNormLayer -> nonlinearity as ONE product on small maps (the fp32 counterpart of
// ghm_conv2d_bn_fwd_lp_q): the generic gather kernel in split-K form, ending in the finishing kernel of conv_small.hip
int ghm_conv_bn_fused_supported_f32(const ghm_conv_desc* d) {
    if (GHM_OPT("GHM_NO_CONV_BN_F32") || d->kh * d->kw > MAX_TAPS || (d->stride != 1 && d->stride != 2)) return 0;
    if (!igemm_fwd_generic(d, GHM_ACT_LINEAR)) return 0;
    const IgemmArgs a = igemm_fwd_args(d, nullptr, nullptr, nullptr, nullptr, GHM_ACT_LINEAR, 0.f, 0);
    return igemm_bn_geometry(a) && ceil_div((long)a.ntaps * a.CH, 16) >= 2 ? 1 : 0;
}

int ghm_conv2d_bn_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias, float* conv_out,
                      float* y, int64_t y_nstride, const float* gamma, const float* beta, float* mean, float* inv,
                      float* run_mean, float* run_inv, float eps, float run_alpha, int32_t act, float alpha) {
    if (int e = check_desc(d)) return e;
    GHM_CHECK(ghm_conv_bn_fused_supported_f32(d), "ghm_conv2d_bn_fwd: geometry not served (ask ghm_conv_bn_fused_supported_f32)");
    GHM_CHECK(conv_out != nullptr && y != nullptr, "ghm_conv2d_bn_fwd: conv_out and y are required");
    const IgemmArgs a = igemm_fwd_args(d, x, wp, bias, conv_out, act, alpha, 0);
    const SmBn bn{gamma, beta, mean, inv, run_mean, run_inv, eps, run_alpha, y, (long)y_nstride};
    return launch_igemm<false>(ctx, a, &bn);
}

struct DactArg {
    const float* y;
    long nstride;
    int kind;
    float alpha;
};

static bool smallk_dgrad_ok(const ghm_conv_desc* d) {
    return d->K <= 4 && d->C > 4 && GHM_OPT("GHM_NO_SMALLK_DGRAD") == nullptr &&
           !(d->K <= 4 && thin_fanout_dgrad_ok(d, GHM_ACT_LINEAR)) && !thin_fanin_s2_ok(d, nullptr);
}

static int launch_smallk_dgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                               float* dx, int act, float alpha, int accumulate, const float* dact_y, long dact_nstride,
                               int dact, float dact_alpha) {
    SmallKDgradArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.dy = dy; sa.wp = wp; sa.bias = bias; sa.dx = dx;
    sa.N = d->N; sa.C = d->C; sa.H = d->H; sa.W = d->W; sa.K = d->K; sa.Ho = d->Ho; sa.Wo = d->Wo;
    sa.kh = d->kh; sa.kw = d->kw; sa.stride = d->stride; sa.pad = d->pad;
    sa.x_nstride = d->x_nstride; sa.y_nstride = d->y_nstride;
    sa.act = act; sa.alpha = alpha; sa.accumulate = accumulate;
    sa.dact_y = dact_y; sa.dact_nstride = dact_nstride; sa.dact = dact; sa.dact_alpha = dact_alpha;
    if (d->K == 1 && d->kh == d->kw && (d->kh == 3 || d->kh == 5) && GHM_OPT("GHM_NO_SMALLK1") == nullptr) {
        constexpr int CG = 8;
        const dim3 grid(ceil_div((long)d->N * d->H * d->W, 256), ceil_div(d->C, CG));
        if (d->kh == 3) hipLaunchKernelGGL((smallk1_dgrad_kernel<3, CG>), grid, dim3(256), 0, ctx->stream, sa);
        else hipLaunchKernelGGL((smallk1_dgrad_kernel<5, CG>), grid, dim3(256), 0, ctx->stream, sa);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(smallk_dgrad_kernel, dim3(ceil_div((long)d->N * d->C * d->H * d->W, 256)), dim3(256), 0, ctx->stream,
                       sa);
    GHM_LAUNCH_CHECK();
    return 0;
}

static bool pool_geom_ok(const ghm_conv_desc* d, int act) {
    return (act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU) && d->stride == 1 && d->kh == d->kw &&
           d->Ho == d->H && d->Wo == d->W && d->H % 2 == 0 && d->W % 4 == 0 && GHM_OPT("GHM_NO_POOL_FUSE") == nullptr;
}

static bool patch_pool_ok(const ghm_conv_desc* d, int num_cu) {
    const PatchPlan pl = plan_patch(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, num_cu, 1);
    return pl.ok && pl.splits == 1 && pl.bm == 64;
}

int ghm_conv2d_pool_supported(const ghm_conv_desc* d, int32_t act, int32_t dtype) {
    if (!pool_geom_ok(d, act)) return 0;
    if (d->C <= 4) return thin_fanout_fwd_pool_ok(d, act) ? 1 : 0;
    if (dtype != GHM_DTYPE_F32 && lp_conv_pool_supported(d, act, dtype)) return 2;
    return patch_pool_ok(d, ghm_plan_cus()) ? 1 : 0;
}

int ghm_conv2d_fwd_pool(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const void* w, const float* bias,
                        float* pooled, uint8_t* mask, int32_t act, float alpha, int32_t dtype) {
    if (int e = check_desc(d)) return e;
    const int form = ghm_conv2d_pool_supported(d, act, dtype);
    GHM_CHECK(form != 0, "ghm_conv2d_fwd_pool: geometry / activation not served (ask ghm_conv2d_pool_supported)");
    if (d->C <= 4) return thin_fanout_fwd_pool(ctx, d, x, (const float*)w, bias, pooled, mask, act, alpha);
    if (form == 2) return lp_conv_fwd_pool(ctx, d, x, w, bias, pooled, mask, act, alpha, dtype);
    const PatchPlan pl = plan_patch(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, ctx->num_cu, 1);
    GHM_CHECK(pl.ok && pl.splits == 1 && pl.bm == 64, "ghm_conv2d_fwd_pool: no single-pass patch plan on this device");
    PatchArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.in = x; pa.wp = (const float*)w; pa.bias = bias; pa.out = nullptr;
    pa.N = d->N; pa.CH = d->C; pa.H = d->Ho; pa.W = d->Wo; pa.Hin = d->H; pa.Win = d->W; pa.in_nstride = d->x_nstride;
    pa.R = d->K; pa.out_nstride = 0; pa.pad = d->pad;
    pa.act = act; pa.alpha = alpha; pa.accumulate = 0;
    pa.slabs_per_split = pl.slabs_per_split; pa.partial = nullptr; pa.zeros = ctx->zeros;
    pa.pool_out = pooled; pa.pool_mask = mask;
    const dim3 g(pl.grid, 1);
    if (d->kh == 5)
        hipLaunchKernelGGL((conv_patch_kernel<5, 64, 8, 1, 4, 1, 1>), g, dim3(256), pl.lds, ctx->stream, pa);
    else
        hipLaunchKernelGGL((conv_patch_kernel<3, 64, 8, 1, 4, 2, 1>), g, dim3(256), pl.lds, ctx->stream, pa);
    GHM_LAUNCH_CHECK();
    return 0;
}

// the thin (<= 4 input channels) forward kernels with a q copy of their result written by their own epilogue
int ghm_thin_fwd_q_supported(const ghm_conv_desc* d, int32_t act, int32_t pooled, int32_t dtype) {
    // (dtype 3 = host-side 'bf16x3': the q copy as three exact bf16 pieces for the split-fp32 kernels)
    if (!(dtype == GHM_DTYPE_BF16 || dtype == GHM_DTYPE_F16 || dtype == 3 || dtype == 4) || d->C > 4 || d->K % 8 || GHM_OPT("GHM_NO_THIN_Q")) return 0;
    if (pooled) return (thin_fanout_fwd_pool_ok(d, act) && thin_fanout_pool_q_ok(d)) || thin_pool_lp_ok(d, act, 0.f, dtype) ? 1 : 0;
    return thin_fanout_fwd_ok(d, act) ? 1 : 0;
}

int ghm_conv2d_fwd_thin_q(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias, float* y,
                          int32_t act, float alpha, void* yq, int64_t yq_nstride, int32_t dtype) {
    if (int e = check_desc(d)) return e;
    GHM_CHECK(yq && ghm_thin_fwd_q_supported(d, act, 0, dtype), "ghm_conv2d_fwd_thin_q: not served (ask ghm_thin_fwd_q_supported)");
    // (y == NULL: only the q copy is written)
    return thin_fanout_fwd(ctx, d, x, wp, bias, y, act, alpha, 0, yq, (long)yq_nstride, dtype);
}

int ghm_conv2d_fwd_pool_thin_q(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                               float* pooled, uint8_t* mask, int32_t act, float alpha, void* yq, int64_t yq_nstride,
                               int32_t dtype) {
    if (int e = check_desc(d)) return e;
    GHM_CHECK(yq && ghm_thin_fwd_q_supported(d, act, 1, dtype), "ghm_conv2d_fwd_pool_thin_q: not served (ask ghm_thin_fwd_q_supported)");
    if (thin_pool_lp_ok(d, act, alpha, dtype))        // reduced-precision modes: the taps on the bf16 / fp16 matrix cores
        return thin_pool_lp(ctx, d, x, wp, bias, pooled, mask, act, alpha, yq, (long)yq_nstride, dtype);
    return thin_fanout_fwd_pool(ctx, d, x, wp, bias, pooled, mask, act, alpha, yq, (long)yq_nstride, dtype);
}

int ghm_conv2d_dgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                     float* dx, int32_t act, float alpha, int32_t accumulate) {
    if (int e = check_desc(d)) return e;
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    if (d->K <= 4 && thin_fanout_dgrad_ok(d, act))
        return thin_fanout_dgrad(ctx, d, dy, wp, bias, dx, act, alpha, accumulate);
    if (thin_fanin_s2_ok(d, dx)) return thin_fanin_s2(ctx, d, dy, wp, bias, dx, act, alpha, accumulate);
    if (d->C <= 4 && thin_fanin_s1_dgrad_ok(d)) return thin_fanin_s1_dgrad(ctx, d, dy, wp, bias, dx, act, alpha, accumulate);
    if (smallk_dgrad_ok(d)) return launch_smallk_dgrad(ctx, d, dy, wp, bias, dx, act, alpha, accumulate, nullptr, 0, 0, 0.f);
    if (taps_as_rows(d, d->C)) {
        const int T = d->kh * d->kw;
        void* ws = nullptr;
        const size_t tb = (size_t)d->N * T * d->C * d->H * d->W * sizeof(float);
        if (int e = ghm_scratch(ctx, tb + (64u << 20), &ws)) return e;
        float* Tbuf = (float*)((char*)ws + (64u << 20));
        // rows m = c*T + tap, reduce over the conv's filters: wp viewed as [m][K] is the WT-form layout with T=1
        IgemmArgs g = pointwise_args(dy, d->y_nstride, d->N, d->K, d->H, d->W, wp, T * d->C, Tbuf);
        if (int e = launch_igemm<true>(ctx, g)) return e;
        ShiftArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.src = Tbuf; sa.dst = dx; sa.bias = bias; sa.N = d->N; sa.R = d->C; sa.T = T; sa.H = d->H; sa.W = d->W;
        sa.dst_nstride = d->x_nstride; sa.act = act; sa.alpha = alpha; sa.accumulate = accumulate;
        sa.tap_major = 0; sa.sign = -1;
        fill_taps(d, sa);
        hipLaunchKernelGGL(shift_add_kernel, dim3(ceil_div((long)d->N * d->C * d->H * d->W, 256)), dim3(256), 0,
                           ctx->stream, sa);
        GHM_LAUNCH_CHECK();
        return 0;
    }
    const int s = d->stride;
    IgemmArgs4 a4;
    int ncls = 0;
    for (int pu = 0; pu < s; ++pu) {
        for (int pv = 0; pv < s; ++pv) {
            IgemmArgs a;
            memset(&a, 0, sizeof(a));
            a.in = dy; a.wp = wp; a.bias = bias; a.out = dx;
            a.N = d->N; a.CH = d->K; a.Hin = d->Ho; a.Win = d->Wo; a.in_nstride = d->y_nstride;
            a.R = d->C; a.Hout = d->H; a.Wout = d->W; a.out_nstride = d->x_nstride;
            a.Hs = (d->H - pu + s - 1) / s; a.Ws = (d->W - pv + s - 1) / s;
            a.os = s; a.ou = pu; a.ov = pv; a.ss = 1;
            a.T = d->kh * d->kw;
            a.act = act; a.alpha = alpha; a.accumulate = accumulate;
            int nt = 0;
            for (int ta = 0; ta < d->kh; ++ta) {
                if (((pu + d->pad - ta) % s + s) % s != 0) continue;
                for (int tb = 0; tb < d->kw; ++tb) {
                    if (((pv + d->pad - tb) % s + s) % s != 0) continue;
                    // exact division: numerator is a multiple of s (may be negative)
                    a.di[nt] = (pu + d->pad - ta) / s; a.dj[nt] = (pv + d->pad - tb) / s;
                    a.wi[nt] = ta * d->kw + tb;
                    ++nt;
                }
            }
            a.ntaps = nt;
            if (a.Hs <= 0 || a.Ws <= 0) continue;
            GHM_CHECK(ncls < 4, "data gradient: stride %d has more than four parity classes", s);
            a4.c[ncls++] = a;
        }
    }
    // stride 2 on an even grid: the four classes share R and the sub-grid -> ONE launch + ONE epilogue for all of them
    if (ncls == 4 && d->H % 2 == 0 && d->W % 2 == 0 && d->C > 4 && GHM_OPT("GHM_NO_DGRAD_X4") == nullptr)
        return launch_igemm_x4(ctx, a4);
    for (int c = 0; c < ncls; ++c)
        if (int e = launch_igemm<true>(ctx, a4.c[c])) return e;
    return 0;
}

// stride-1 data gradient as a forward conv K -> C on the transposed weights: padding k-1-pad
static ghm_conv_desc swapped_desc(const ghm_conv_desc* d) {
    ghm_conv_desc s = *d;
    s.C = d->K; s.K = d->C; s.H = d->Ho; s.W = d->Wo; s.Ho = d->H; s.Wo = d->W;
    s.pad = d->kh - 1 - d->pad;
    s.x_nstride = d->y_nstride; s.y_nstride = d->x_nstride;
    return s;
}

int ghm_dgrad_t_supported(const ghm_conv_desc* d) {
    if (d->stride == 1) return 1;
    return d->stride == 2 && d->kh == 3 && d->kw == 3 && d->pad == 1 && d->H == 2 * d->Ho && d->W == 2 * d->Wo &&
           d->Wo % 32 == 0 && d->Ho % 2 == 0 && d->K % 4 == 0 && d->K >= 8 && d->C % 4 == 0 && d->C >= 32 &&
           GHM_OPT("GHM_NO_PATCH_S2") == nullptr;
}

int ghm_conv2d_transpose_weights(ghm_ctx* ctx, const ghm_conv_desc* d, const float* wp, float* wpT) {
    const int T = d->kh * d->kw;
    hipLaunchKernelGGL(transpose_weights_kernel, dim3(ceil_div(d->K, 32), ceil_div(d->C, 32), T), dim3(256), 0,
                       ctx->stream, wp, wpT, d->C, T, d->K);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_transpose_weights_batched(ghm_ctx* ctx, const void* table, int32_t n_items, int32_t total_blocks) {
    static_assert(sizeof(TransposeItem) == 32, "table layout is part of the ABI (see ghm.h)");
    if (n_items <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(transpose_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, ctx->stream,
                       (const TransposeItem*)table, n_items);
    GHM_LAUNCH_CHECK();
    return 0;
}

// 3x3 stride-2 pad-1 data gradient on the transposed weights (dgrad_s2_patch_kernel); ``da``: fold the producer's
// activation derivative into the epilogue (single-pass plans only)
struct DgradS2Plan {
    int bm, wm, rt;             // channels per block, waves along channels (4 / wm class rows per block)
    int grid, splits, slabs_per_split;
};

// Tile choice: the split-K form writes `splits` partial dx tensors and re-reads them -- for these layers that is more
// HBM traffic than the kernel's own operands, so a grid that is too small first shrinks the tile (128 ch x 2 rows ->
// 64 x 4 -> 64 x 2 class rows) and only splits the contraction when even the smallest tile leaves CUs idle.
static DgradS2Plan dgrad_s2_plan(const ghm_conv_desc* d, int num_cu) {
    static const int tiles[3][2] = {{128, 2}, {64, 1}, {64, 2}};
    DgradS2Plan p;
    int first = d->C >= 96 ? 0 : 1, forced = -1;
    if (const char* f = GHM_OPT("GHM_DGRAD_S2_TILE")) forced = atoi(f);
    // round 4 (tools/s2_f32_quick.sh, isolated, TFLOP/s for tiles 0 / 1 / 2): N8 C64 256^2 K128 57 / 107 / 113, N8 C128 128^2 K256
    // 111 / 114 / 117, N8 C256 64^2 K512 102 / 103 / 113, N4 variants 54-92 / 94-103 / 93-109: the 64 x 2 tile (three blocks
    // per CU, twice the blocks) is the best or equal everywhere -> first choice; the others stay for GHM_DGRAD_S2_TILE
    (void)first;
    for (int t = 2; t >= 0; --t) {
        if (forced >= 0) t = forced > 2 ? 2 : forced;
        p.bm = tiles[t][0]; p.wm = tiles[t][1]; p.rt = 4 / p.wm;
        p.grid = ((d->C + p.bm - 1) / p.bm) * (d->Wo / 32) * ((d->Ho + p.rt - 1) / p.rt) * d->N;
        break;
    }
    const int nslabs = d->K / 4;
    p.splits = 1;
    if (p.grid < num_cu) {
        p.splits = (2 * num_cu + p.grid - 1) / p.grid;
        const int maxs = nslabs / 4 > 0 ? nslabs / 4 : 1;
        if (p.splits > maxs) p.splits = maxs;
    }
    if (const char* f = GHM_OPT("GHM_DGRAD_S2_SPLITS")) p.splits = atoi(f) < 1 ? 1 : atoi(f);
    p.slabs_per_split = (nslabs + p.splits - 1) / p.splits;
    p.splits = (nslabs + p.slabs_per_split - 1) / p.slabs_per_split;
    return p;
}

static int dgrad_s2_splits(const ghm_conv_desc* d, int num_cu) { return dgrad_s2_plan(d, num_cu).splits; }

static int dgrad_s2_patch_launch(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wpT, const float* bias,
                                 float* dx, int act, float alpha, int accumulate, const DactArg* da) {
    PatchArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.in = dy; pa.wp = wpT; pa.bias = bias; pa.out = dx;
    pa.N = d->N; pa.CH = d->K; pa.H = d->H; pa.W = d->W; pa.Hin = d->Ho; pa.Win = d->Wo;
    pa.in_nstride = d->y_nstride; pa.R = d->C; pa.out_nstride = d->x_nstride; pa.pad = d->pad;
    pa.act = act; pa.alpha = alpha; pa.accumulate = accumulate;
    if (da) { pa.dact_y = da->y; pa.dact_nstride = da->nstride; pa.dact = da->kind; pa.dact_alpha = da->alpha; }
    const DgradS2Plan pl = dgrad_s2_plan(d, ctx->num_cu);
    constexpr int cb = 4;
    const int splits = pl.splits;
    pa.slabs_per_split = pl.slabs_per_split;
    GHM_CHECK(!(da && splits > 1), "dgrad + activation derivative needs a single-pass plan (ask ghm_dgrad_dact_supported)");
    if (splits > 1) {
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, (size_t)splits * pa.R * pa.N * pa.H * pa.W * sizeof(float), &ws)) return e;
        pa.partial = (float*)ws;
    }
    pa.zeros = (const float*)ctx->zeros;
    const size_t lds = (size_t)2 * (cb * 9 * pl.bm + ((cb * (pl.rt + 1) * 33 + 3) / 4) * 4) * sizeof(float);
    const dim3 g(pl.grid, splits);
#define GHM_DS2_CASE(BM_, WM_)                                                                                   \
    if (pl.bm == BM_ && pl.wm == WM_) {                                                                          \
        if (da)                                                                                                  \
            hipLaunchKernelGGL((dgrad_s2_patch_kernel<BM_, WM_, 2, true>), g, dim3(256), lds, ctx->stream, pa);  \
        else                                                                                                     \
            hipLaunchKernelGGL((dgrad_s2_patch_kernel<BM_, WM_, 2>), g, dim3(256), lds, ctx->stream, pa);        \
    }
    GHM_DS2_CASE(128, 2)
    GHM_DS2_CASE(64, 1)
    GHM_DS2_CASE(64, 2)
#undef GHM_DS2_CASE
    GHM_LAUNCH_CHECK();
    if (splits > 1) {
        IgemmArgs e;
        memset(&e, 0, sizeof(e));
        e.partial = pa.partial; e.out = dx; e.bias = bias; e.N = d->N; e.R = d->C;
        e.Hout = d->H; e.Wout = d->W; e.out_nstride = d->x_nstride; e.Hs = d->H; e.Ws = d->W; e.os = 1;
        e.act = act; e.alpha = alpha; e.accumulate = accumulate;
        hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ceil_div((long)d->N * d->H * d->W * d->C, 256)), dim3(256), 0,
                           ctx->stream, e, splits);
        GHM_LAUNCH_CHECK();
    }
    return 0;
}

int ghm_conv2d_dgrad_t(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wpT, const float* bias,
                       float* dx, int32_t act, float alpha, int32_t accumulate) {
    if (int e = check_desc(d)) return e;
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    if (d->stride == 2) {
        GHM_CHECK(ghm_dgrad_t_supported(d), "ghm_conv2d_dgrad_t: this stride-2 geometry has no transposed-weight kernel");
        return dgrad_s2_patch_launch(ctx, d, dy, wpT, bias, dx, act, alpha, accumulate, nullptr);
    }
    // the data gradient of a stride-1 conv is the forward conv K -> C with flipped taps (already folded into
    // wpT by ghm_conv2d_transpose_weights) and padding k-1-pad
    const int padT = d->kh - 1 - d->pad;
    if (d->K <= 4) {
        const ghm_conv_desc sw = swapped_desc(d);
        if (thin_fanout_fwd_ok(&sw, act)) return thin_fanout_fwd(ctx, &sw, dy, wpT, bias, dx, act, alpha, accumulate);
    }
    if (d->kh == d->kw && d->Ho == d->H && d->Wo == d->W) {
        const PatchPlan pl = plan_patch(d->N, d->K, d->H, d->W, d->C, d->kh, ctx->num_cu);
        if (pl.ok) {
            PatchArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.in = dy; pa.wp = wpT; pa.bias = bias; pa.out = dx;
            pa.N = d->N; pa.CH = d->K; pa.H = d->H; pa.W = d->W; pa.Hin = d->H; pa.Win = d->W;
            pa.in_nstride = d->y_nstride;
            pa.R = d->C; pa.out_nstride = d->x_nstride; pa.pad = padT;
            pa.act = act; pa.alpha = alpha; pa.accumulate = accumulate;
            return launch_patch(ctx, pl, pa, d->kh);
        }
    }
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = dy; a.wp = wpT; a.bias = bias; a.out = dx;
    a.N = d->N; a.CH = d->K; a.Hin = d->Ho; a.Win = d->Wo; a.in_nstride = d->y_nstride;
    a.R = d->C; a.Hout = d->H; a.Wout = d->W; a.out_nstride = d->x_nstride;
    a.Hs = d->H; a.Ws = d->W; a.os = 1; a.ou = 0; a.ov = 0; a.ss = 1;
    a.T = d->kh * d->kw; a.ntaps = a.T;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    for (int ta = 0; ta < d->kh; ++ta)
        for (int tb = 0; tb < d->kw; ++tb) {
            const int t = ta * d->kw + tb;
            a.di[t] = ta - padT; a.dj[t] = tb - padT; a.wi[t] = t;
        }
    return launch_igemm<false>(ctx, a);
}

int ghm_dgrad_dact_supported(const ghm_conv_desc* d, int32_t dtype) {
    if (GHM_OPT("GHM_NO_DACT_FUSE")) return 0;
    if (smallk_dgrad_ok(d)) return 1;                                           // fp32 packed wp
    if (dtype != GHM_DTYPE_F32 && lp_dgrad_s2_single_pass(d, dtype)) return 3;   // low-precision transposed pack
    if (d->stride == 2 && ghm_dgrad_t_supported(d) && dgrad_s2_splits(d, ghm_plan_cus()) == 1) return 2;     // fp32 wpT
    return 0;
}

int ghm_conv2d_dgrad_dact(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const void* w, float* dx,
                          const float* dact_y, int64_t dact_nstride, int32_t dact, float dact_alpha, int32_t dtype) {
    if (int e = check_desc(d)) return e;
    const int form = ghm_dgrad_dact_supported(d, dtype);
    GHM_CHECK(form != 0 && dact_y != nullptr, "ghm_conv2d_dgrad_dact: not served (ask ghm_dgrad_dact_supported)");
    GHM_CHECK(dact == GHM_ACT_RELU || dact == GHM_ACT_LRELU, "ghm_conv2d_dgrad_dact: relu / leaky relu only");
    if (dact == GHM_ACT_RELU) dact_alpha = 0.f;          // the kernels multiply by (y > 0 ? 1 : slope)
    if (form == 1)
        return launch_smallk_dgrad(ctx, d, dy, (const float*)w, nullptr, dx, GHM_ACT_LINEAR, 0.f, 0, dact_y, (long)dact_nstride,
                                   dact, dact_alpha);
    if (form == 3) return lp_dgrad_s2_dact(ctx, d, dy, w, dx, dact_y, (long)dact_nstride, dact, dact_alpha, dtype);
    const DactArg da = {dact_y, (long)dact_nstride, dact, dact_alpha};
    return dgrad_s2_patch_launch(ctx, d, dy, (const float*)w, nullptr, dx, GHM_ACT_LINEAR, 0.f, 0, &da);
}

int ghm_conv2d_wgrad_workspace(const ghm_conv_desc* d, size_t* bytes) {
    // upper bound independent of the CU count (splits <= 1024)
    const WVariant v = pick_wgrad(d, ghm_plan_cus());
    const size_t n = (size_t)d->C * d->kh * d->kw * d->K;
    *bytes = (v.splits > 1 ? (size_t)v.splits * n * sizeof(float) : 16);
    return 0;
}

static int wgrad_impl(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* dy, float* dwp,
                      void* workspace, int32_t accumulate) {
    const WVariant v = pick_wgrad(d, ghm_plan_cus());
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.x_nstride = d->x_nstride;
    a.K = d->K; a.Ho = d->Ho; a.Wo = d->Wo; a.y_nstride = d->y_nstride;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.CT = d->C * d->kh * d->kw;
    a.P = d->N * d->Ho * d->Wo;
    a.slabs_per_split = v.slabs_per_split;
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    const long n = (long)a.CT * a.K;
    if (v.splits > 1) {
        GHM_CHECK(workspace != nullptr, "wgrad needs a workspace for %d splits", v.splits);
        a.out = (float*)workspace; a.split_stride = n; a.accumulate = 0;
    } else {
        a.out = dwp; a.split_stride = 0; a.accumulate = accumulate;
    }
    dim3 grid(v.row_tiles, ceil_div(a.K, v.bn), v.splits);
    if (v.patch) {
#define GHM_WPATCH_CASE(KS_, ST_, BN_, WM_, WN_)                                                              \
    if (d->kh == KS_ && d->stride == ST_ && v.bn == BN_) {                                                    \
        if (v.bkp == 32)                                                                                      \
            hipLaunchKernelGGL((wgrad_patch_kernel<KS_, ST_, BN_, WM_, WN_, 32>), grid, dim3(256), 0, ctx->stream, a); \
        else                                                                                                  \
            hipLaunchKernelGGL((wgrad_patch_kernel<KS_, ST_, BN_, WM_, WN_, 16>), grid, dim3(256), 0, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                                   \
    } else
        GHM_WPATCH_CASE(5, 1, 128, 2, 2)
        GHM_WPATCH_CASE(5, 1, 64, 4, 1)
        GHM_WPATCH_CASE(5, 1, 32, 4, 1)
        GHM_WPATCH_CASE(3, 1, 128, 2, 2)
        GHM_WPATCH_CASE(3, 1, 64, 4, 1)
        GHM_WPATCH_CASE(3, 1, 32, 4, 1)
        GHM_WPATCH_CASE(3, 2, 128, 2, 2)
        GHM_WPATCH_CASE(3, 2, 64, 4, 1)
        GHM_WPATCH_CASE(3, 2, 32, 4, 1) {
            ghm_set_error("no wgrad_patch variant for k=%d s=%d bn=%d", d->kh, d->stride, v.bn);
            return -3;
        }
#undef GHM_WPATCH_CASE
    } else
#define GHM_WGRAD_CASE(BM_, BN_, WM_, WN_)                                                          \
    if (v.bm == BM_ && v.bn == BN_) {                                                               \
        hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, WM_, WN_>), grid, dim3(256), 0, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                         \
    } else
    GHM_WGRAD_CASE(128, 128, 2, 2)
    GHM_WGRAD_CASE(128, 64, 4, 1)
    GHM_WGRAD_CASE(128, 32, 4, 1)
    GHM_WGRAD_CASE(32, 128, 1, 4) {
        ghm_set_error("no wgrad variant for bm=%d bn=%d", v.bm, v.bn);
        return -3;
    }
#undef GHM_WGRAD_CASE
    if (v.splits > 1)
        return launch_reduce_splits(ctx, (const float*)workspace, v.splits, n, n, dwp, accumulate);
    return 0;
}

int ghm_conv2d_wgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* dy, float* dwp,
                     void* workspace, int32_t accumulate) {
    if (int e = check_desc(d)) return e;
    if (thin_wgrad_ok(d, x, dy)) return thin_wgrad(ctx, d, x, dy, dwp, accumulate);
    if (taps_as_rows(d, d->K)) {
        // shift-expand dy into T*K planes over the input grid, then a 1x1 weight-gradient GEMM whose "filters"
        // are (tap, k): its packed output [c][1][tap*K + k] IS dwp[c][tap][k]
        const int T = d->kh * d->kw;
        void* ws = nullptr;
        const size_t eb = (size_t)d->N * T * d->K * d->H * d->W * sizeof(float);
        if (int e = ghm_scratch(ctx, eb + (64u << 20), &ws)) return e;
        float* E = (float*)((char*)ws + (64u << 20));
        ShiftArgs sa;
        memset(&sa, 0, sizeof(sa));
        fill_taps(d, sa);
        hipLaunchKernelGGL(shift_expand_kernel, dim3(ceil_div((long)d->N * T * d->K * d->H * d->W, 256)), dim3(256), 0,
                           ctx->stream, dy, (long)d->y_nstride, E, d->N, d->K, T, d->H, d->W, sa);
        GHM_LAUNCH_CHECK();
        ghm_conv_desc d2 = *d;
        d2.K = T * d->K; d2.Ho = d->H; d2.Wo = d->W; d2.kh = d2.kw = 1; d2.stride = 1; d2.pad = 0;
        d2.y_nstride = (int64_t)d2.K * d->H * d->W;
        size_t need = 0;
        ghm_conv2d_wgrad_workspace(&d2, &need);
        GHM_CHECK(need <= (64u << 20), "taps-as-rows weight gradient needs %zu bytes of partials", need);
        return wgrad_impl(ctx, &d2, x, E, dwp, ws, accumulate);
    }
    return wgrad_impl(ctx, d, x, dy, dwp, workspace, accumulate);
}

int ghm_channel_sum(ghm_ctx* ctx, const float* x, int32_t N, int32_t C, int32_t HW, int64_t nstride, float* out,
                    int32_t accumulate) {
    if (C == 0) return 0;
    const long total = (long)N * HW;
    long S = (1024 + C - 1) / C;
    long maxs = total / 4096;
    if (maxs < 1) maxs = 1;
    if (S > maxs) S = maxs;
    if (S > 256) S = 256;
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, (size_t)C * S * sizeof(float), &ws)) return e;
    hipLaunchKernelGGL(channel_sum_partial, dim3((int)S, C), dim3(256), 0, ctx->stream, x, N, HW, (long)nstride, (int)S,
                       (float*)ws, S == 1 ? out : nullptr, accumulate);
    GHM_LAUNCH_CHECK();
    if (S == 1) return 0;
    hipLaunchKernelGGL(channel_sum_final, dim3(ceil_div(C, 256)), dim3(256), 0, ctx->stream, (const float*)ws, C, (int)S,
                       out, accumulate);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_conv2d_variant(const ghm_conv_desc* d, int32_t kind, char* out, int32_t out_len) {
    if (kind == 0 && d->C <= 4 && thin_fanout_fwd_ok(d, GHM_ACT_LINEAR)) {
        snprintf(out, out_len, "fanout_kernel<fwd>");
        return 0;
    }
    if (kind == 1 && d->K <= 4 && thin_fanout_dgrad_ok(d, GHM_ACT_LINEAR)) {
        snprintf(out, out_len, "fanout_kernel<dgrad>");
        return 0;
    }
    if (kind == 0 && d->K <= 4 && thin_fanin_s1_fwd_ok(d)) {
        snprintf(out, out_len, "fanin_s1_kernel<fwd>");
        return 0;
    }
    if (kind == 1 && d->C <= 4 && !thin_fanin_s2_ok(d, nullptr) && thin_fanin_s1_dgrad_ok(d)) {
        snprintf(out, out_len, "fanin_s1_kernel<dgrad>");
        return 0;
    }
    if (kind == 1 && thin_fanin_s2_ok(d, nullptr)) {
        snprintf(out, out_len, "fanin_s2_kernel<%d>", d->kh);
        return 0;
    }
    if (kind == 1 && d->K <= 4 && d->C > 4 && GHM_OPT("GHM_NO_SMALLK_DGRAD") == nullptr) {
        snprintf(out, out_len, "smallk_dgrad_kernel");
        return 0;
    }
    if (kind == 2 && thin_wgrad_ok(d, nullptr, nullptr)) {
        snprintf(out, out_len, "thin_wgrad_kernel<%d>", d->C <= 4 ? d->K : d->C);
        return 0;
    }
    if (kind == 3 && d->stride == 1 && d->K <= 4) {
        const ghm_conv_desc sw = swapped_desc(d);
        if (thin_fanout_fwd_ok(&sw, GHM_ACT_LINEAR)) {
            snprintf(out, out_len, "fanout_kernel<dgrad_t>");
            return 0;
        }
    }
    if (taps_as_rows(d, kind == 1 ? d->C : d->K)) {
        snprintf(out, out_len, "taps_as_rows<%s>", kind == 0 ? "fwd" : (kind == 1 ? "dgrad" : "wgrad"));
        return 0;
    }
    if (kind == 0 && d->kh == d->kw && ((d->stride == 1 && d->Ho == d->H && d->Wo == d->W) ||
                                        (d->stride == 2 && d->Ho * 2 == d->H && d->Wo * 2 == d->W))) {
        const PatchPlan pl = plan_patch(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, 256, d->stride);
        if (pl.ok) {
            snprintf(out, out_len, "conv_patch_kernel<%d, %d, %d, %d, %d, %d, %d> splits=%d", d->kh, pl.bm, pl.rt,
                     pl.bm == 128 ? 2 : 1, pl.bm == 128 ? 2 : 4, d->kh == 5 ? 1 : 2, d->stride, pl.splits);
            return 0;
        }
    }
    if (kind == 3) {      // data gradient through transposed weights
        const PatchPlan pl = plan_patch(d->N, d->K, d->H, d->W, d->C, d->kh, ghm_plan_cus());
        if (d->stride == 1 && d->kh == d->kw && d->Ho == d->H && d->Wo == d->W && pl.ok)
            snprintf(out, out_len, "conv_patch_kernel<%d, %d, %d, %d, %d, %d, 1> splits=%d", d->kh, pl.bm, pl.rt,
                     pl.bm == 128 ? 2 : 1, pl.bm == 128 ? 2 : 4, d->kh == 5 ? 1 : 2, pl.splits);
        else if (d->stride == 2) {
            const DgradS2Plan pl = dgrad_s2_plan(d, ghm_plan_cus());
            snprintf(out, out_len, "dgrad_s2_patch_kernel<%d, %d, 2> splits=%d", pl.bm, pl.wm, pl.splits);
        } else
            snprintf(out, out_len, "igemm_kernel<fwd on wT>");
        return 0;
    }
    if (kind == 2) {
        const WVariant v = pick_wgrad(d, ghm_plan_cus());
        if (v.patch)
            snprintf(out, out_len, "wgrad_patch_kernel<%d, %d, %d, %d, %d, %d> splits=%d", d->kh, d->stride, v.bn,
                     v.bn == 128 ? 2 : 4, v.bn == 128 ? 2 : 1, v.bkp, v.splits);
        else
            snprintf(out, out_len, "wgrad_kernel<%d, %d, %d, %d> splits=%d", v.bm, v.bn, v.bm == 32 ? 1 : (v.bn == 128 ? 2 : 4),
                     v.bm == 32 ? 4 : (v.bn == 128 ? 2 : 1), v.splits);
    } else {
        const int R = kind == 0 ? d->K : d->C;
        const long P = kind == 0 ? (long)d->N * d->Ho * d->Wo : (long)d->N * d->H * d->W / (d->stride * d->stride);
        if (R <= 4) {
            snprintf(out, out_len, "direct_smallr_kernel");
        } else if (kind == 0 && d->kh == 1 && d->kw == 1 && P <= 8 && R >= 1024 && d->C >= 64 && !GHM_OPT("GHM_NO_DENSE_SMALLP")) {
            snprintf(out, out_len, "dense_smallp_kernel");
        } else {
            const Variant v = pick_variant(R, P, ghm_plan_cus());
            snprintf(out, out_len, "igemm_kernel<%d,%d,%s>", v.bm, v.bn, kind == 0 ? "fwd" : "wt");
        }
    }
    return 0;
}

}  // extern "C"
