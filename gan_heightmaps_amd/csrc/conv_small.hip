// Small-map layers on the bf16 / fp16 matrix cores (gfx950): the inner half of the U-Net (architectures/p2p.py:169-240),
// the first stages of the DCGAN generator (dcgan.py:16-24) and the last of its discriminator (dcgan.py:42-50) have maps
// of 1x1 .. 16x16 pixels: at batch 4 a layer is <= 1024 pixels per channel x 4.7 .. 18.9 MB of weights.  Such a layer is
// WEIGHT-STREAMING bound (microseconds at HBM rate) and what it costs in the step is the number and latency of its
// launches, so a layer is TWO launches here, whatever its geometry:
//
//   sm_gemm_kernel    out[r, m] partial sums of a gather GEMM: rows = 32 filters, columns = 128 pixels, K = (16-channel
//                     slab, tap).  A block owns one (row tile, pixel group, split of the slabs) and streams its weights
//                     and its input slabs global -> LDS by DMA through a ring several slabs deep (the weights of a layer
//                     are read once chip-wide; every wave issues the same number of DMA instructions per slab, so one
//                     compile-time s_waitcnt vmcnt keeps the ring full).  Filter taps are a TABLE of (weight tap, LDS
//                     offset): stride-1 / stride-2 forward convolutions, stride-1 data gradients (transposed pack) and
//                     the four output-parity classes of a stride-2 data gradient (blockIdx.z = class, 1 / 2 / 2 / 4 taps,
//                     no zero insertion) are the same kernel.
//   sm_finish_kernel  one block per 8 output channels sees the WHOLE map of those channels: fixed-order sum of the split
//                     partials + bias (+ accumulate), then -- because the block holds every value of its channels --
//                     the BatchNorm that follows the convolution in the nets (batch statistics in fp64, running-statistic
//                     update, normalise, activation) and the q copy for the next product, in the same launch.
//
// Arithmetic: the operands are the q tensors / weight packs of conv_lp.hip (rounded once at their producer), products
// exact, fp32 accumulation: the results equal oracle/lp.py up to fp32 summation order (tests/test_gpu_lp.py).
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
struct Sm;
template <>
struct Sm<GHM_DTYPE_BF16> {
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <>
struct Sm<GHM_DTYPE_F16> {
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int SM_MT = 128;          // pixels per block (4 waves x one 32-pixel MFMA column tile)
constexpr int SM_MAXTAPS = 36;

struct SmArgs {
    const u32x4* in_q;      // the operand q tensor: conv input (forward) / output gradient (data gradient)
    long in_q_nstride;      // units between samples
    const u32x4* zeros;     // >= 16 bytes of zeros in HBM (DMA source of padding units)
    const u32x4* wq;        // weight pack [CH / 8][T][Rpad] units (plain: forward, transposed: data gradient)
    float* partial;         // [splits][R][Mtot] fp32
    int Rpad, T;
    int N, CH, Hin, Win;    // operand geometry
    int H, W;               // pixel grid a block walks (the output grid; stride-2 data gradient: one parity class of it)
    int OH, OW;             // the full output grid (== H, W unless ncls == 4: then 2H x 2W)
    int R;                  // output rows (filters forward, input channels in a data gradient)
    int st;                 // operand step per pixel of the walked grid (the forward stride; 1 in every data gradient)
    int nimg;               // images whose operand maps a block stages (1: a block is a row range of one image)
    int rows;               // walked rows of an image per block
    int LH, LW;             // staged operand window per image (units), zero outside the operand map
    int dy0, dx0;           // operand coordinate of window element (0, 0) = (first walked row * st + dy0, dx0)
    int splits, slabs_per_split;
    int pfs;                // ring slots (3 .. SmRing::MAXPFS)
    int ncls;               // 1, or 4 output parity classes (stride-2 data gradient)
    int cls_begin[5];       // taps of class c: [cls_begin[c], cls_begin[c + 1])
    int tap_w[SM_MAXTAPS];          // tap index inside the pack (dwords: a uniform index then reads them with s_load)
    int tap_off[SM_MAXTAPS];        // window offset of the tap relative to the pixel's base
};

// ring geometry for (NTAPS = most taps a block walks, NQ = operand-window DMA instructions per wave and slab).  The ring's
// DEPTH is a run-time choice (SmArgs::pfs, 3 .. 8 slots): in the train step three streams share every CU, and a block that
// asks for most of the LDS waits for the other streams' blocks to retire before it can start at all -- what a deep ring buys
// in the K loop it loses at dispatch (measured in the step, tools/instep_sweep.sh).
template <int NTAPS, int NQ>
struct SmRing {
    static constexpr int WI = (NTAPS + 3) / 4;                 // weight DMA instructions per wave and slab
    static constexpr int WSLOT = 4 * WI * 64;                  // units
    static constexpr int ISLOT = NQ * 256;                     // units
    static constexpr int SLOT = WSLOT + ISLOT;
    static constexpr int G = WI + NQ;                          // DMA instructions per wave and slab (vmcnt bookkeeping)
    static constexpr int MAXPFS_ = 60 / G + 2;                 // (pfs - 2) * G <= 60 < 64 (6-bit vmcnt)
    static constexpr int MAXPFS = MAXPFS_ > 8 ? 8 : MAXPFS_;
    static_assert(MAXPFS >= 3, "ring needs three slots");
};

template <int N>
__device__ __forceinline__ void sm_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DT, int NTAPS, int NQ>
__global__ __launch_bounds__(256, 1) void sm_gemm_kernel(const SmArgs a) {
    typedef SmRing<NTAPS, NQ> RG;
    const int PFS = a.pfs;
    extern __shared__ __attribute__((aligned(16))) u32x4 sm_lds[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, li = lane & 31;
    const int r0 = blockIdx.x * 32;
    const int cls = blockIdx.z / a.splits, split = blockIdx.z - cls * a.splits;
    const int tb = a.cls_begin[cls], nt = a.cls_begin[cls + 1] - tb;
    const int HW = a.H * a.W, HWin = a.Hin * a.Win;
    const int Mcls = a.N * HW;
    const int m0 = blockIdx.y * SM_MT;
    const int n0 = m0 / HW;
    const int y_lo = a.nimg == 1 && HW >= SM_MT ? (m0 - n0 * HW) / a.W : 0;
    const int IMGU = a.nimg * a.LH * a.LW;                // units per channel block of the staged window(s)
    const int nslabs = a.CH / 16;
    const int s_begin = split * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);
    const int nsl = s_end - s_begin;

    // this lane's pixel (one 32-pixel tile per wave)
    const int m = m0 + wave * 32 + li;
    const bool mvalid = m < Mcls;
    int pn = 0, py_ = 0, px_ = 0, base = 0;
    if (mvalid) {
        pn = m / HW;
        const int p = m - pn * HW;
        py_ = p / a.W;
        px_ = p - py_ * a.W;
        base = ((pn - n0) * a.LH + (py_ - y_lo) * a.st) * a.LW + px_ * a.st;
    }

    // ---- operand-window DMA plan: unit e of the slot = (channel block, image, window row, window column) ----
    int p_off[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * 256;
        const int cb = e / IMGU, rem = e - cb * IMGU;
        const int img = rem / (a.LH * a.LW), r2 = rem - img * (a.LH * a.LW);
        const int ly = r2 / a.LW, lx = r2 - ly * a.LW;
        const int iy = y_lo * a.st + a.dy0 + ly, ix = a.dx0 + lx;
        const bool ok = e < 2 * IMGU && (n0 + img) < a.N && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        p_off[q] = ok ? (int)((long)img * a.in_q_nstride) + cb * HWin + iy * a.Win + ix : -1;
    }
    const u32x4* const ibase = a.in_q + (long)n0 * a.in_q_nstride;
    // ---- weight DMA plan: instruction j = wave + 4 i carries tap j of both channel blocks of the slab (32 rows each) ----
    int w_off[RG::WI];
#pragma unroll
    for (int i = 0; i < RG::WI; ++i) {
        const int t = wave + 4 * i;
        w_off[i] = t < nt ? (kg * a.T + a.tap_w[tb + t]) * a.Rpad + r0 + li : -1;
    }
    const long wslab = (long)2 * a.T * a.Rpad;             // units per 16-channel slab of the pack

    // (the tap tables are indexed dynamically: vector loads.  Everything that reads them sits BEFORE the first DMA, so
    // the waits for them never drain the ring)
    int off[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) off[t] = t < nt ? a.tap_off[tb + t] : 0;

    auto issue = [&](int i) {           // slab s_begin + i -> ring slot i % PFS (a dummy group past the block's range)
        u32x4* const slot = sm_lds + (i % PFS) * RG::SLOT;
        const bool real = i < nsl;
        const long s = s_begin + i;
#pragma unroll
        for (int k = 0; k < RG::WI; ++k) {
            const u32x4* g = (real && w_off[k] >= 0) ? a.wq + s * wslab + w_off[k] : a.zeros;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(slot + (wave + 4 * k) * 64), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u32x4* g = (real && p_off[q] >= 0) ? ibase + s * 2 * HWin + p_off[q] : a.zeros;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(slot + RG::WSLOT + q * 256 + wave * 64), 16, 0, 0);
        }
    };

    f32x16 acc0, acc1;              // two chains (even / odd taps): back-to-back MFMAs never wait for their own accumulator
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;

    for (int i = 0; i < PFS - 1; ++i) issue(i);
    for (int i = 0; i < nsl; ++i) {
        switch (PFS) {                             // this wave's part of slab i has landed ...
            case 3: sm_wait_vmcnt<1 * RG::G>(); break;
            case 4: sm_wait_vmcnt<(RG::MAXPFS >= 4 ? 2 : 0) * RG::G>(); break;
            case 5: sm_wait_vmcnt<(RG::MAXPFS >= 5 ? 3 : 0) * RG::G>(); break;
            case 6: sm_wait_vmcnt<(RG::MAXPFS >= 6 ? 4 : 0) * RG::G>(); break;
            case 7: sm_wait_vmcnt<(RG::MAXPFS >= 7 ? 5 : 0) * RG::G>(); break;
            default: sm_wait_vmcnt<(RG::MAXPFS >= 8 ? 6 : 0) * RG::G>(); break;
        }
        __syncthreads();                           // ... and everybody's; slot (i - 1) % PFS is free again
        issue(i + PFS - 1);
        const u32x4* const Ws = sm_lds + (i % PFS) * RG::SLOT + lane;
        const u32x4* const Is = sm_lds + (i % PFS) * RG::SLOT + RG::WSLOT + kg * IMGU + base;
        // every fragment of the slab is read before its first MFMA (taps beyond the class's count: a zero weight row of
        // the slot's padding -- the dummy DMA filled it with zeros -- times a valid window address)
        u32x4 af[NTAPS], bf[NTAPS];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            af[t] = Ws[t * 64];
            bf[t] = Is[off[t]];
        }
        __builtin_amdgcn_sched_barrier(0);         // (left alone, the compiler sinks every read to just before its MFMA)
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            if (t & 1)
                acc1 = Sm<DT>::mfma(af[t], bf[t], acc1);
            else
                acc0 = Sm<DT>::mfma(af[t], bf[t], acc0);
        }
    }
    sm_wait_vmcnt<0>();                            // the dummy groups: nothing may land in LDS after the block retires

    if (!mvalid) return;
    // ---- partial slice: rows e -> (e & 3) + 8 (e >> 2) + 4 kg, column = this lane's pixel ----
    const long Mtot = (long)a.N * a.OH * a.OW;
    long mo;
    if (a.ncls == 4)
        mo = ((long)pn * a.OH + 2 * py_ + (cls >> 1)) * a.OW + 2 * px_ + (cls & 1);
    else
        mo = m;
    float* const pb = a.partial + ((long)split * a.R + r0 + 4 * kg) * Mtot + mo;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = (e & 3) + 8 * (e >> 2);
        if (r0 + 4 * kg + k < a.R) pb[(long)k * Mtot] = acc0[e] + acc1[e];
    }
}

// ---- finish: 8 channels x the whole map per block ----
struct SmFinishArgs {
    const float* partial;       // [splits][R][Mtot]
    int splits, R, N, HW;       // Mtot = N * HW
    const float* bias;          // or null
    float* out;                 // fp32 NCHW result of the product (convolution output / data gradient) or null
    long out_nstride;
    int accumulate;             // out += (linear epilogue, no BatchNorm)
    int act;
    float alpha;
    // q copy of the FINAL result (after BatchNorm + activation when bn is set) or null
    u32x4* outq;
    long outq_nstride;
    // BatchNorm behind the convolution (training statistics over N * HW per channel)
    int bn;
    const float* gamma;
    const float* beta;
    float* mean;
    float* inv;
    float* run_mean;            // or null
    float* run_inv;
    float eps, run_alpha;
    float* y;                   // fp32 NCHW act(bn(out)) or null
    long y_nstride;
};

__device__ __forceinline__ double sm_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NT threads = 8 channels x NT / 8 pixel lanes, 8 pixels per thread: a block covers NT pixels of its 8 channels -- 256 per
// block (blockIdx.y walks the map) without a BatchNorm, the WHOLE map (<= 1024 pixels) in one 1024-thread block with one.
// In the train step a kernel of this size costs its DEPENDENT MEMORY ROUND TRIPS (microseconds each under the other streams'
// load), so there is exactly one: every split partial of the thread's 8 pixels is loaded at once (64 independent loads,
// eight splits per round), added in split order, and the BatchNorm works on the values in registers -- statistics, running
// update, normalise, activation -- before anything is written.  The 8 channels of a pixel meet in LDS for the q unit.
template <int DT, int NT>
__global__ __launch_bounds__(NT) void sm_finish_kernel(const SmFinishArgs a) {
    constexpr int LPC = NT / 8, FI = 8, FP = LPC * FI;
    const int r0 = blockIdx.x * 8;
    const int tid = threadIdx.x, c = tid / LPC, pl = tid % LPC;
    const long Mtot = (long)a.N * a.HW;
    const long sstride = (long)a.R * Mtot;
    const int r = r0 + c;
    const long mbase = (long)blockIdx.y * FP;
    const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
    const bool pwl = a.act == GHM_ACT_LINEAR || a.act == GHM_ACT_RELU || a.act == GHM_ACT_LRELU;
    __shared__ float stage[8][FP + 4];
    __shared__ double red[NT / 64][2];
    __shared__ float stat[16];

    const float* const prow = a.partial + (long)r * Mtot + mbase + pl;
    float t[FI];
    {
        const float bv = a.bias ? a.bias[r] : 0.f;
#pragma unroll
        for (int j = 0; j < FI; ++j) t[j] = bv;
    }
    for (int s0 = 0; s0 < a.splits; s0 += 8) {
        float v[8][FI];
#pragma unroll
        for (int ss = 0; ss < 8; ++ss)
#pragma unroll
            for (int j = 0; j < FI; ++j)
                v[ss][j] = (s0 + ss < a.splits && mbase + j * LPC + pl < Mtot) ? prow[(long)(s0 + ss) * sstride + j * LPC] : 0.f;
#pragma unroll
        for (int ss = 0; ss < 8; ++ss)
#pragma unroll
            for (int j = 0; j < FI; ++j) t[j] += v[ss][j];
    }
    int pn[FI], pp[FI];
#pragma unroll
    for (int j = 0; j < FI; ++j) {
        const long m = mbase + j * LPC + pl;
        pn[j] = (int)(m / a.HW);
        pp[j] = (int)(m - (long)pn[j] * a.HW);
    }
    if (a.bn) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < FI; ++j)
            if (mbase + j * LPC + pl < Mtot) {
                const double d = t[j];
                s1 += d;
                s2 += d * d;
                a.out[(long)pn[j] * a.out_nstride + (long)r * a.HW + pp[j]] = t[j];      // x of the BatchNorm (its backward reads it)
            }
        double sa = 0.0, sb = 0.0;
        if constexpr (LPC >= 64) {      // a channel = LPC / 64 whole waves: wave sums, then a fixed order over the channel's waves
            s1 = sm_wave_sum(s1);
            s2 = sm_wave_sum(s2);
            if ((tid & 63) == 0) {
                red[tid >> 6][0] = s1;
                red[tid >> 6][1] = s2;
            }
            __syncthreads();
            constexpr int WPC = LPC / 64;
#pragma unroll
            for (int w = 0; w < WPC; ++w) {
                sa += red[c * WPC + w][0];
                sb += red[c * WPC + w][1];
            }
        } else {                        // a channel = one half-wave (xor offsets below 32 stay inside it)
#pragma unroll
            for (int o = LPC / 2; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            sa = s1;
            sb = s2;
        }
        if (pl == 0) {
            const double count = (double)Mtot;
            const double mu = sa / count;
            double var = sb / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float mf = (float)mu;
            const float iv = (float)(1.0 / sqrt(var + (double)a.eps));
            a.mean[r] = mf;
            a.inv[r] = iv;
            if (a.run_mean) {
                a.run_mean[r] = (1.f - a.run_alpha) * a.run_mean[r] + a.run_alpha * mf;
                a.run_inv[r] = (1.f - a.run_alpha) * a.run_inv[r] + a.run_alpha * iv;
            }
            stat[c] = mf;
            stat[8 + c] = iv;
        }
        __syncthreads();
        const float mu = stat[c], sc = a.gamma[r] * stat[8 + c], be = a.beta[r];
#pragma unroll
        for (int j = 0; j < FI; ++j)
            if (mbase + j * LPC + pl < Mtot) {
                const float tt = fmaf(t[j] - mu, sc, be);
                const float v = pwl ? (tt > 0.f ? tt : slope * tt) : ghm_act(tt, a.act, a.alpha);
                if (a.y) a.y[(long)pn[j] * a.y_nstride + (long)r * a.HW + pp[j]] = v;
                stage[c][j * LPC + pl] = v;
            }
    } else {
#pragma unroll
        for (int j = 0; j < FI; ++j)
            if (mbase + j * LPC + pl < Mtot) {
                float v = t[j];
                float* o = a.out ? a.out + (long)pn[j] * a.out_nstride + (long)r * a.HW + pp[j] : nullptr;
                if (a.accumulate) v += *o;
                v = pwl ? (v > 0.f ? v : slope * v) : ghm_act(v, a.act, a.alpha);
                if (o) *o = v;
                stage[c][j * LPC + pl] = v;
            }
    }
    if (!a.outq) return;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < FP / NT; ++h) {        // the block's q units: thread = pixel, its 8 channels from LDS
        const int i = tid + h * NT;
        const long m = mbase + i;
        if (m < Mtot) {
            const int n = (int)(m / a.HW), p = (int)(m - (long)n * a.HW);
            u32x4 u;
            u.x = Sm<DT>::pack2(stage[0][i], stage[1][i]);
            u.y = Sm<DT>::pack2(stage[2][i], stage[3][i]);
            u.z = Sm<DT>::pack2(stage[4][i], stage[5][i]);
            u.w = Sm<DT>::pack2(stage[6][i], stage[7][i]);
            a.outq[(long)n * a.outq_nstride + (long)(r0 / 8) * a.HW + p] = u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct SmPlan {
    bool ok;
    int ntaps;          // template class of the tap count: 4, 9 or 25
    int nq;             // 2 or 5
    dim3 grid;
    size_t lds;
    size_t partial_bytes;
    SmArgs a;           // geometry part filled in
};

static int pick(int v, const int* cls, int n) {
    for (int i = 0; i < n; ++i)
        if (v <= cls[i]) return cls[i];
    return -1;
}

template <int NTAPS, int NQ>
static void sm_ring_choice(SmPlan& p, long budget, int nslabs_blk) {
    typedef SmRing<NTAPS, NQ> RG;
    int pfs = (int)(budget / ((long)RG::SLOT * 16));
    if (pfs > RG::MAXPFS) pfs = RG::MAXPFS;
    if (pfs > nslabs_blk + 1) pfs = nslabs_blk + 1;          // (no use prefetching past the block's own slabs)
    if (pfs < 3) pfs = 3;
    p.a.pfs = pfs;
    p.lds = (size_t)pfs * RG::SLOT * 16;
}

// kind 0: forward convolution; kind 1: data gradient (stride 1: forward form on the transposed pack; stride 2: parity classes)
SmPlan sm_plan(const ghm_conv_desc* d, int kind, int num_cu) {
    SmPlan p;
    memset(&p, 0, sizeof(p));
    p.ok = false;
    if (GHM_OPT("GHM_NO_SM") || GHM_OPT("GHM_NO_LP")) return p;
    if (d->kh != d->kw || d->kh > 5 || d->stride < 1 || d->stride > 2) return p;
    const int KS = d->kh, T = KS * KS;
    SmArgs& a = p.a;
    a.T = T;
    a.N = d->N;
    int taps_max = 0;
    if (kind == 0) {
        if (d->Ho != (d->H + 2 * d->pad - KS) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - KS) / d->stride + 1) return p;
        a.CH = d->C; a.Hin = d->H; a.Win = d->W; a.H = a.OH = d->Ho; a.W = a.OW = d->Wo; a.R = d->K;
        a.st = d->stride; a.ncls = 1;
        a.dy0 = -d->pad; a.dx0 = -d->pad;
        taps_max = T;
    } else if (d->stride == 1) {
        if (d->Ho != d->H + 2 * d->pad - KS + 1 || d->Wo != d->W + 2 * d->pad - KS + 1) return p;
        a.CH = d->K; a.Hin = d->Ho; a.Win = d->Wo; a.H = a.OH = d->H; a.W = a.OW = d->W; a.R = d->C;
        a.st = 1; a.ncls = 1;
        a.dy0 = a.dx0 = -(KS - 1 - d->pad);
        taps_max = T;
    } else {
        // dx[U] = sum over tapsT a' with (U + pad - KS + 1 + a') even of wqT[a'] dy[(U + pad - KS + 1 + a') / 2]
        if ((d->H & 1) || (d->W & 1) || d->Ho != (d->H + 2 * d->pad - KS) / 2 + 1 || d->Wo != (d->W + 2 * d->pad - KS) / 2 + 1) return p;
        a.CH = d->K; a.Hin = d->Ho; a.Win = d->Wo; a.H = d->H / 2; a.W = d->W / 2; a.OH = d->H; a.OW = d->W; a.R = d->C;
        a.st = 1; a.ncls = 4;
    }
    if (a.CH % 16 || a.CH < 16 || a.R % 8 || a.R < 8) return p;
    const int HW = a.H * a.W;
    if (HW > 256 || a.W > 16) return p;                         // maps of at most 16 x 16 pixels walked per image
    if (!((HW >= SM_MT && HW % SM_MT == 0 && SM_MT % a.W == 0) || (HW < SM_MT && SM_MT % HW == 0))) return p;
    const long Mcls = (long)a.N * HW, Mtot = (long)a.N * a.OH * a.OW;
    // which layers run here rather than on conv_lp.hip's tiles (measured isolated, warm, bf16, both paths incl. their
    // epilogue launches): up to 8 x 8 maps this path is at the ~7 us floor of a launch pair, where the fp32 igemm + epilogue
    // (maps below 8 columns) took 38 us in the step; at 16 x 16 (1024 pixels, 9.7 GFLOP) its one-tile waves are LDS-bound
    // (22 us against 12 for lp_conv).  The parity-class form of a stride-2 data gradient has no lp_conv counterpart below 32
    // class columns (fp32 igemm_x4: 0.10 ms at 16 x 16 classes), so it is taken up to 1024 class pixels.
    long maxm = a.ncls == 4 ? 1024 : 512;
    if (const char* f = GHM_OPT("GHM_SM_MAXM")) maxm = atol(f);
    if (Mcls > maxm || Mtot > 8192) return p;
    if (HW >= SM_MT) {
        a.nimg = 1;
        a.rows = SM_MT / a.W;
    } else {
        a.nimg = SM_MT / HW < a.N ? SM_MT / HW : a.N;
        a.rows = a.H;
    }
    // ---- tap table(s) ----
    int dymin = 0, dymax = 0, dxmin = 0, dxmax = 0;
    int ty[SM_MAXTAPS], tx[SM_MAXTAPS], n = 0;
    if (a.ncls == 1) {
        a.cls_begin[0] = 0;
        for (int ta = 0; ta < KS; ++ta)
            for (int tb = 0; tb < KS; ++tb) {
                a.tap_w[n] = ta * KS + tb;
                ty[n] = ta + a.dy0;
                tx[n] = tb + a.dx0;
                ++n;
            }
        for (int c = 1; c <= 4; ++c) a.cls_begin[c] = n;
        dymin = a.dy0; dymax = a.dy0 + KS - 1; dxmin = a.dx0; dxmax = a.dx0 + KS - 1;
    } else {
        bool first = true;
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
            a.cls_begin[c] = n;
            for (int ta = 0; ta < KS; ++ta) {
                const int ey = py + d->pad - KS + 1 + ta;
                if (ey & 1) continue;
                for (int tb = 0; tb < KS; ++tb) {
                    const int ex = px + d->pad - KS + 1 + tb;
                    if (ex & 1) continue;
                    if (n >= SM_MAXTAPS) return p;
                    a.tap_w[n] = ta * KS + tb;
                    ty[n] = ey >= 0 ? ey / 2 : -((-ey) / 2);
                    tx[n] = ex >= 0 ? ex / 2 : -((-ex) / 2);
                    if (first || ty[n] < dymin) dymin = ty[n];
                    if (first || ty[n] > dymax) dymax = ty[n];
                    if (first || tx[n] < dxmin) dxmin = tx[n];
                    if (first || tx[n] > dxmax) dxmax = tx[n];
                    first = false;
                    ++n;
                }
            }
            const int cnt = n - a.cls_begin[c];
            if (cnt > taps_max) taps_max = cnt;
        }
        a.cls_begin[4] = n;
        a.dy0 = dymin; a.dx0 = dxmin;
    }
    a.LH = (a.rows - 1) * a.st + (dymax - dymin) + 1;
    a.LW = (a.W - 1) * a.st + (dxmax - dxmin) + 1;
    for (int i = 0; i < n; ++i) a.tap_off[i] = (ty[i] - dymin) * a.LW + (tx[i] - dxmin);
    static const int tcls[3] = {4, 9, 25};
    p.ntaps = pick(taps_max, tcls, 3);
    const int units = 2 * a.nimg * a.LH * a.LW;
    static const int qcls[3] = {1, 2, 5};
    p.nq = pick((units + 255) / 256, qcls, 3);
    if (p.ntaps < 0 || p.nq < 0) return p;
    // ---- grid: (row tiles, pixel groups, classes x splits); the splits fill the chip once ----
    const int ftiles = (a.R + 31) / 32, pgroups = (int)((Mcls + SM_MT - 1) / SM_MT);
    const int nslabs = a.CH / 16;
    const long blocks0 = (long)ftiles * pgroups * a.ncls;
    int splits = (int)((num_cu + blocks0 - 1) / blocks0);
    if (splits > 8) splits = 8;         // (each split is one more round of dependent loads in the finishing kernel)
    if (const char* f = GHM_OPT("GHM_SM_SPLITS")) splits = atoi(f);
    if (splits > nslabs) splits = nslabs;
    if (splits < 1) splits = 1;
    a.slabs_per_split = (nslabs + splits - 1) / splits;
    a.splits = (nslabs + a.slabs_per_split - 1) / a.slabs_per_split;
    p.grid = dim3(ftiles, pgroups, a.ncls * a.splits);
    p.partial_bytes = (size_t)a.splits * a.R * Mtot * sizeof(float);
    long budget = 32 * 1024;       // (in-step, bf16, img/s: 32 KB 649.7, 64 KB 645.4, 144 KB 642.7 -- within the run-to-run spread; the small footprint is the principled choice)
    if (const char* f = GHM_OPT("GHM_SM_LDS_KB")) budget = atol(f) * 1024;
#define GHM_SM_RING(T_, Q_) \
    if (p.ntaps == T_ && p.nq == Q_) sm_ring_choice<T_, Q_>(p, budget, a.slabs_per_split);
    GHM_SM_RING(4, 1) GHM_SM_RING(4, 2) GHM_SM_RING(4, 5) GHM_SM_RING(9, 1) GHM_SM_RING(9, 2) GHM_SM_RING(9, 5)
    GHM_SM_RING(25, 1) GHM_SM_RING(25, 2) GHM_SM_RING(25, 5)
#undef GHM_SM_RING
    p.ok = true;
    return p;
}

template <int DT, int NTAPS, int NQ>
int sm_launch_gemm(ghm_ctx* ctx, const SmPlan& p) {
    static bool attr_done = false;
    if (!attr_done) {
        GHM_HIP(hipFuncSetAttribute((const void*)sm_gemm_kernel<DT, NTAPS, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((sm_gemm_kernel<DT, NTAPS, NQ>), p.grid, dim3(256), p.lds, ctx->stream, p.a);
    GHM_LAUNCH_CHECK();
    return 0;
}

template <int DT>
int sm_launch(ghm_ctx* ctx, const SmPlan& p) {
#define GHM_SM_CASE(T_, Q_) \
    if (p.ntaps == T_ && p.nq == Q_) return sm_launch_gemm<DT, T_, Q_>(ctx, p);
    GHM_SM_CASE(4, 1) GHM_SM_CASE(4, 2) GHM_SM_CASE(4, 5) GHM_SM_CASE(9, 1) GHM_SM_CASE(9, 2) GHM_SM_CASE(9, 5)
    GHM_SM_CASE(25, 1) GHM_SM_CASE(25, 2) GHM_SM_CASE(25, 5)
#undef GHM_SM_CASE
    ghm_set_error("sm_launch: no variant for %d taps, %d window groups", p.ntaps, p.nq);
    return -3;
}

static inline size_t sm_align256(size_t n) { return (n + 255) / 256 * 256; }

template <int DT>
__global__ __launch_bounds__(256) void sm_q_pack_kernel(const float* __restrict__ x, long x_nstride, int N, int C8, int HW,
                                                        u32x4* __restrict__ q, long q_nstride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * C8 * HW) return;
    const int p = (int)(idx % HW);
    const long nc = idx / HW;
    const int cb = (int)(nc % C8), n = (int)(nc / C8);
    const float* g = x + (long)n * x_nstride + (long)cb * 8 * HW + p;
    u32x4 u;
    u.x = Sm<DT>::pack2(g[0], g[(long)HW]);
    u.y = Sm<DT>::pack2(g[2L * HW], g[3L * HW]);
    u.z = Sm<DT>::pack2(g[4L * HW], g[5L * HW]);
    u.w = Sm<DT>::pack2(g[6L * HW], g[7L * HW]);
    q[(long)n * q_nstride + (long)cb * HW + p] = u;
}

}  // namespace

// ---- library-internal interface (common.h): conv_lp.hip routes the geometries served here ----
bool sm_use(const ghm_conv_desc* d, int kind, int dtype) {
    if (dtype != GHM_DTYPE_BF16 && dtype != GHM_DTYPE_F16) return false;
    if (kind != 0 && kind != 1) return false;
    return sm_plan(d, kind, ghm_plan_cus()).ok;
}

int sm_conv(ghm_ctx* ctx, const ghm_conv_desc* d, int kind, const void* inq, long inq_ns, const float* in32, const void* wq,
            const float* bias, float* out32, long out_nstride, void* outq, long outq_ns, int act, float alpha, int accumulate,
            int dtype, const SmBn* bn) {
    SmPlan p = sm_plan(d, kind, ctx->num_cu);
    GHM_CHECK(p.ok, "sm_conv: geometry not served");
    GHM_CHECK(!(accumulate && (act != GHM_ACT_LINEAR || bn)), "sm_conv: accumulate needs a linear epilogue");
    GHM_CHECK(!(accumulate && !out32), "sm_conv: accumulate needs the fp32 output");
    GHM_CHECK(out32 || (outq && !bn), "sm_conv: no output");
    SmArgs& a = p.a;
    // workspace: [q copy of an fp32 operand][partials]
    const size_t qbytes = in32 ? sm_align256((size_t)a.N * (a.CH / 8) * a.Hin * a.Win * 16) : 0;
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, qbytes + p.partial_bytes, &ws)) return e;
    if (in32) {
        const long total = (long)a.N * (a.CH / 8) * a.Hin * a.Win;
        const long ns32 = kind == 0 ? d->x_nstride : d->y_nstride;
        inq = ws;
        inq_ns = (long)(a.CH / 8) * a.Hin * a.Win;
        if (dtype == GHM_DTYPE_BF16)
            hipLaunchKernelGGL((sm_q_pack_kernel<GHM_DTYPE_BF16>), dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, in32, ns32,
                               a.N, a.CH / 8, a.Hin * a.Win, (u32x4*)ws, inq_ns);
        else
            hipLaunchKernelGGL((sm_q_pack_kernel<GHM_DTYPE_F16>), dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, in32, ns32,
                               a.N, a.CH / 8, a.Hin * a.Win, (u32x4*)ws, inq_ns);
        GHM_LAUNCH_CHECK();
    }
    a.in_q = (const u32x4*)inq;
    a.in_q_nstride = inq_ns;
    a.zeros = (const u32x4*)ctx->zeros;
    a.wq = (const u32x4*)wq;
    a.Rpad = (a.R + 127) / 128 * 128;
    a.partial = (float*)((char*)ws + qbytes);
    if (int e = dtype == GHM_DTYPE_BF16 ? sm_launch<GHM_DTYPE_BF16>(ctx, p) : sm_launch<GHM_DTYPE_F16>(ctx, p)) return e;
    return sm_finish_launch(ctx, a.partial, a.splits, a.R, a.N, a.OH * a.OW, bias, out32, out_nstride, accumulate, act, alpha, outq,
                            outq_ns, dtype, bn);
}

// the finishing kernel on its own: conv_lp.hip's split-K launches (16 x 16 maps: partial slices in the same [split][R][N * HW]
// layout) end in it too -- one launch for sum + bias + activation + fp32 and q outputs, and the BatchNorm when asked for
int sm_finish_launch(ghm_ctx* ctx, const float* partial, int splits, int R, int N, int HW, const float* bias, float* out32,
                     long out_nstride, int accumulate, int act, float alpha, void* outq, long outq_ns, int dtype, const SmBn* bn) {
    GHM_CHECK(R % 8 == 0 && (long)N * HW <= 8192, "sm_finish_launch: rows %% 8 == 0, at most 8192 pixels per channel");
    GHM_CHECK(!bn || out32, "sm_finish_launch: the BatchNorm form needs the fp32 convolution output");
    SmFinishArgs f;
    memset(&f, 0, sizeof(f));
    f.partial = partial; f.splits = splits; f.R = R; f.N = N; f.HW = HW;
    f.bias = bias; f.out = out32; f.out_nstride = out_nstride; f.accumulate = accumulate; f.act = act; f.alpha = alpha;
    f.outq = (u32x4*)outq; f.outq_nstride = outq_ns;
    if (bn) {
        f.bn = 1; f.gamma = bn->gamma; f.beta = bn->beta; f.mean = bn->mean; f.inv = bn->inv; f.run_mean = bn->run_mean;
        f.run_inv = bn->run_inv; f.eps = bn->eps; f.run_alpha = bn->run_alpha; f.y = bn->y; f.y_nstride = bn->y_nstride;
    }
    if (bn && (long)N * HW > 256) {
        GHM_CHECK((long)N * HW <= 1024, "sm_finish_launch: the BatchNorm form holds the whole map of its channels (<= 1024 pixels)");
        if (dtype == GHM_DTYPE_BF16)
            hipLaunchKernelGGL((sm_finish_kernel<GHM_DTYPE_BF16, 1024>), dim3(R / 8), dim3(1024), 0, ctx->stream, f);
        else
            hipLaunchKernelGGL((sm_finish_kernel<GHM_DTYPE_F16, 1024>), dim3(R / 8), dim3(1024), 0, ctx->stream, f);
    } else {
        const dim3 fg(R / 8, (unsigned)(((long)N * HW + 255) / 256));
        if (dtype == GHM_DTYPE_BF16)
            hipLaunchKernelGGL((sm_finish_kernel<GHM_DTYPE_BF16, 256>), fg, dim3(256), 0, ctx->stream, f);
        else
            hipLaunchKernelGGL((sm_finish_kernel<GHM_DTYPE_F16, 256>), fg, dim3(256), 0, ctx->stream, f);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

extern "C" {

// name of the kernel family that serves low-precision product ``kind`` (0 forward, 1 data gradient, 2 weight gradient) of
// this geometry -- what bench.py / the profile tables label a launch with (the fp32 counterpart is ghm_conv2d_variant)
int ghm_lp_variant(const ghm_conv_desc* d, int32_t kind, int32_t dtype, char* out, int32_t out_len) {
    const char* dt = dtype == GHM_DTYPE_BF16 ? "bf16" : (dtype == GHM_DTYPE_F16 ? "f16" : "f32");
    if ((kind == 0 || kind == 1) && sm_use(d, kind, dtype)) {
        snprintf(out, out_len, "sm_gemm_kernel<%s, %d, %d, %s>", dt, d->kh, d->stride, kind == 0 ? "fwd" : "dgrad");
        return 0;
    }
    const char* fam = kind == 2 ? "wgrad" : ((kind == 1 && d->stride == 2) ? "dgrad_s2" : "conv");
    snprintf(out, out_len, "lp_%s_kernel<%s, %d, %d>", fam, dt, d->kh, d->stride);
    return 0;
}

// is conv -> BatchNorm (training statistics) -> activation served as ONE product (gather GEMM + a finishing kernel that
// holds the whole map of its channels)?  Only for the small maps of conv_small.hip.
int ghm_conv_bn_fused_supported(const ghm_conv_desc* d, int32_t dtype) {
    return (sm_use(d, 0, dtype) || lp_fwd_splitk_bn_ok(d, dtype)) ? 1 : 0;
}

// y = act(bn(conv(x) + bias)) with batch statistics over (N, Ho, Wo): conv_out (fp32 NCHW, required: the BatchNorm
// backward reads it) receives conv(x) + bias, mean / inv the statistics (running statistics updated when run_mean != NULL,
// as ghm_bn_stats), y (fp32, may be NULL) and / or yq (q tensor, may be NULL) the layer output.
int ghm_conv2d_bn_fwd_lp_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, const void* wq,
                           const float* bias, float* conv_out, float* y, int64_t y_nstride, void* yq, int64_t yq_nstride,
                           const float* gamma, const float* beta, float* mean, float* inv, float* run_mean, float* run_inv,
                           float eps, float run_alpha, int32_t act, float alpha, int32_t dtype) {
    GHM_CHECK(ghm_conv_bn_fused_supported(d, dtype), "ghm_conv2d_bn_fwd_lp_q: geometry / dtype not served (ask ghm_conv_bn_fused_supported)");
    GHM_CHECK(conv_out != nullptr && (y != nullptr || yq != nullptr), "ghm_conv2d_bn_fwd_lp_q: conv_out and one of y / yq are required");
    GHM_CHECK((((uintptr_t)xq | (uintptr_t)yq) & 15) == 0, "ghm_conv2d_bn_fwd_lp_q: q tensors are 16-byte aligned");
    SmBn bn{gamma, beta, mean, inv, run_mean, run_inv, eps, run_alpha, y, (long)y_nstride};
    if (!sm_use(d, 0, dtype))
        return lp_fwd_splitk_bn(ctx, d, xq, (long)xq_nstride, wq, bias, conv_out, yq, (long)yq_nstride, act, alpha, dtype, &bn);
    return sm_conv(ctx, d, 0, xq, (long)xq_nstride, nullptr, wq, bias, conv_out, d->y_nstride, yq, (long)yq_nstride, act, alpha, 0,
                   dtype, &bn);
}

}  // extern "C"
