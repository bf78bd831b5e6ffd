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
    float* pool_out;            // POOL: dense [N, R, Hout/2, Wout/2] maximum of act(conv + bias) over 2x2 windows ...
    unsigned char* pool_mask;   // ... and the 4-bit arg-max mask of every window (bit 2*dr + dc; all ties set)
    uint2* out_q;               // optional q copy of the result (of the POOLED result when POOL): half-units of 4 channels
    long out_q_nstride;         // 16-byte units between samples
    int q_dt;                   // GHM_DTYPE_BF16 / GHM_DTYPE_F16
};

typedef float f32x2t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned thin_pack2(float a, float b, int dt) {
    const f32x2t v = {a, b};
    return dt == GHM_DTYPE_BF16 ? __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2t))
                                : __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2t));
}

// a half unit (4 consecutive channels of one pixel) of a q tensor; dt == 3 ('bf16x3', the split-fp32 mode): the three bf16
// pieces of each value in three planes ``ps2`` half-units apart (csrc/conv_split.hip, elementwise_q.hip q_store8)
__device__ __forceinline__ void thin_qstore4(uint2* qo, float v0, float v1, float v2, float v3, int dt, long ps2) {
    if (dt == 3 || dt == 4) {         // (4 = 'bf16x2': the first two pieces)
        unsigned p[2][3];
        const float in[2][2] = {{v0, v1}, {v2, v3}};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x2t v = {in[t][0], in[t][1]};
            p[t][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2t));
            f32x2t r = {v[0] - __uint_as_float(p[t][0] << 16), v[1] - __uint_as_float(p[t][0] & 0xffff0000u)};
            p[t][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2t));
            f32x2t r2 = {r[0] - __uint_as_float(p[t][1] << 16), r[1] - __uint_as_float(p[t][1] & 0xffff0000u)};
            p[t][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2t));
        }
        qo[0] = make_uint2(p[0][0], p[1][0]);
        qo[ps2] = make_uint2(p[0][1], p[1][1]);
        if (dt == 3) qo[2 * ps2] = make_uint2(p[0][2], p[1][2]);
    } else {
        *qo = make_uint2(thin_pack2(v0, v1, dt), thin_pack2(v2, v3, dt));
    }
}

// the three (two) bf16 pieces of four values as half units: p[piece] = {v0 v1, v2 v3}
__device__ __forceinline__ void thin_split4(float v0, float v1, float v2, float v3, uint2 (&pc)[3]) {
    unsigned p[2][3];
    const float in[2][2] = {{v0, v1}, {v2, v3}};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x2t v = {in[t][0], in[t][1]};
        p[t][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2t));
        f32x2t r = {v[0] - __uint_as_float(p[t][0] << 16), v[1] - __uint_as_float(p[t][0] & 0xffff0000u)};
        p[t][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2t));
        f32x2t r2 = {r[0] - __uint_as_float(p[t][1] << 16), r[1] - __uint_as_float(p[t][1] & 0xffff0000u)};
        p[t][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2t));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) pc[q] = make_uint2(p[0][q], p[1][q]);
}
typedef unsigned thin_u2 __attribute__((ext_vector_type(2)));
// lanes l and l + 32 hold the two halves of the q units of pixels a and b: after two lane swaps the lower lane holds the WHOLE
// unit of pixel a and the upper lane the whole unit of pixel b -- one 16-byte store per lane instead of two 8-byte ones
__device__ __forceinline__ uint4 thin_pair_units(uint2 pa, uint2 pb) {
    const thin_u2 r0 = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
    const thin_u2 r1 = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
    return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}

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
//   POOL (d_conv1 -> LeakyRectify -> MaxPool2D, architectures/dcgan.py:42-47): a group is TWO output rows x 64 pixels
//   (NS = 2): the lane that owns pixels (2l, 2l+1) of both rows owns one whole pooling window, so the epilogue takes
//   the maximum in registers and stores the pooled value (4 B, 128 contiguous bytes per 32 lanes) and the 4-bit arg-max
//   mask -- the 537 MB full-resolution activation is never written.
// WS = 2: EIGHT MFMA waves; the waves (2g, 2g + 1) work on the same pixel group and own RB of its 2 * RB row blocks
// each.  With four waves (one per SIMD) a wave's MFMA burst (64 cycles each in fp32) and its VALU epilogue (bias,
// activation, pooling, stores: as many cycles again -- SQ counters, d_conv1: 1.7 M MFMAs = 44 us of matrix pipe, 22.7 M
// VALU instructions = 37 us of vector pipe, kernel 127 us) run one after the other; two waves per SIMD with half the
// accumulators each (<= 170 VGPRs) let one wave's epilogue run under the other's MFMAs.
template <int KSTEPS, int RB, int NS, bool ACC, bool POOL = false, int WS = 1, bool QOUT = false>
__global__ __launch_bounds__((4 * WS + 1) * 64, WS == 2 ? 1 : 2) void fanout_kernel(const FanoutArgs a) {
    constexpr int NMW = 4 * WS;                                   // MFMA waves; wave NMW is the loader
    typedef typename StoreVec<NS>::type vec_t;
    extern __shared__ float lds[];
    float* sbias = lds;                                           // [RB*WS*32]
    int* stab = reinterpret_cast<int*>(lds + RB * WS * 32);       // [2][2*KSTEPS]: LDS offset, weight offset
    float* rows = lds + RB * WS * 32 + 4 * KSTEPS;                // [2][CS*KRT*LW]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    const int rowsz = a.CS * a.KRT * a.LW;
    if (threadIdx.x < RB * WS * 32) sbias[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;
    const int rbo = WS == 2 ? (wv & 1) * RB : 0;                  // this wave's first row block
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
    if (wv == NMW && r < NI) produce(r, rows);
    __syncthreads();          // tables + first rows (nothing else is in flight yet)

    // MFMA waves: the whole weight matrix as A fragments, lane (l, h) holds A[rb*32 + l][2j + h]
    float A[RB][KSTEPS];
    int off[KSTEPS];
    if (wv < NMW) {
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            off[j] = stab[2 * j + h] + NS * l * a.ss;      // this lane's first pixel of a group
            const int wi = stab[2 * KSTEPS + 2 * j + h];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float w = a.wp[wi >= 0 ? ((rb + rbo) * 32 + l) * a.w_rs + wi : 0];      // branch-free: clamp, mask
                A[rb][j] = wi >= 0 ? w : 0.f;
            }
        }
    }
    const int GPR = a.Wout / (NS * 32);       // pixel groups per output row
    const int G = POOL ? (a.RPI / 2) * GPR : a.RPI * GPR;
    const int HWout = a.Hout * a.Wout;
    // store addressing: wave-uniform 64-bit base (scalar) + 32-bit per-lane byte offset
    const unsigned plane = (unsigned)HWout * 4u;
    const unsigned lane_out = (4u * h * (unsigned)HWout + NS * l) * 4u;
    // linear / relu / leaky relu as one select: v > 0 ? v : slope * v
    const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);

    for (int it = 0; r < NI; ++it, r += gridDim.x) {
        const float* cur = rows + (it & 1) * rowsz;
        if (wv == NMW) {
            if (r + (int)gridDim.x < NI) produce(r + gridDim.x, rows + ((it + 1) & 1) * rowsz);
        } else {
            const int n = r / IPI, u0 = (r - n * IPI) * a.RPI;
            for (int g = wv / WS; g < G; g += 4) {
                if constexpr (POOL) {
                    static_assert(!POOL || (NS == 2 && !ACC), "pooled fan-out: 2 pixels x 2 rows per lane");
                    const int rp = g / GPR, x0 = (g - rp * GPR) * (NS * 32);
                    const int ri = 2 * rp;
                    const float* bsrc = cur + (ri * a.LW + x0) * a.ss;
                    f32x16 acc[2][NS][RB];              // [row of the pair][pixel of the pair][row block]
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < NS; ++k)
#pragma unroll
                            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                                for (int e = 0; e < 16; ++e) acc[q][k][rb][e] = 0.f;
#pragma unroll
                    for (int j = 0; j < KSTEPS; ++j) {
                        float B[2][NS];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int k = 0; k < NS; ++k) B[q][k] = bsrc[q * a.LW * a.ss + off[j] + k * a.ss];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int k = 0; k < NS; ++k)
#pragma unroll
                                for (int rb = 0; rb < RB; ++rb)
                                    acc[q][k][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][j], B[q][k], acc[q][k][rb], 0, 0, 0);
                    }
                    const int Wp = a.Wout / 2;
                    const long HWp = (long)(a.Hout / 2) * Wp;
                    const long pbase = (long)n * a.R * HWp + (long)((u0 + ri) / 2) * Wp + x0 / 2 + l;
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        float bv[16];
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const float4 t = *reinterpret_cast<const float4*>(sbias + (rb + rbo) * 32 + 8 * e4 + 4 * h);
                            bv[4 * e4] = t.x; bv[4 * e4 + 1] = t.y; bv[4 * e4 + 2] = t.z; bv[4 * e4 + 3] = t.w;
                        }
                        // a lane owns one window per channel row; lanes (2t, 2t+1) trade every other row so that each
                        // stores TWO adjacent pooled pixels of one row: 8-byte value + 2-byte mask stores, half as many
                        const bool odd = l & 1;
                        float qv[4];                    // pooled values of rows 4g .. 4g+3 (+ 4h) of this lane's OWN window
#pragma unroll
                        for (int e2 = 0; e2 < 8; ++e2) {
                            // rows ea (kept by even lanes) and eb (kept by odd lanes): two registers live at a time
                            const int ea = 2 * e2, eb = 2 * e2 + 1;
                            float pm[2];
                            unsigned pk[2];
#pragma unroll
                            for (int z = 0; z < 2; ++z) {
                                const int e = 2 * e2 + z;
                                float v[4];
#pragma unroll
                                for (int q = 0; q < 2; ++q)
#pragma unroll
                                    for (int k = 0; k < 2; ++k) {
                                        const float t = acc[q][k][rb][e] + bv[e];
                                        v[2 * q + k] = t > 0.f ? t : slope * t;
                                    }
                                pm[z] = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                                pk[z] = (v[0] == pm[z] ? 1u : 0u) | (v[1] == pm[z] ? 2u : 0u) | (v[2] == pm[z] ? 4u : 0u) |
                                        (v[3] == pm[z] ? 8u : 0u) | (pm[z] > 0.f ? GHM_POOL_SIGN : 0u);
                            }
                            if (QOUT) {
                                qv[2 * (e2 & 1)] = pm[0];
                                qv[2 * (e2 & 1) + 1] = pm[1];
                            }
                            if (QOUT && (e2 & 1)) {         // q unit (channel block, pooled pixel), half h: 4 consecutive channels
                                uint2* qo = a.out_q + 2 * ((long)n * a.out_q_nstride + (long)((rb + rbo) * 4 + (e2 >> 1)) * HWp +
                                                           (long)((u0 + ri) / 2) * Wp + x0 / 2 + l) + h;
                                thin_qstore4(qo, qv[0], qv[1], qv[2], qv[3], a.q_dt, 2 * (long)a.N * a.out_q_nstride);
                            }
                            const float keep = odd ? pm[1] : pm[0], send = odd ? pm[0] : pm[1];
                            const unsigned keepk = odd ? pk[1] : pk[0], sendk = odd ? pk[0] : pk[1];
                            const float recv = __shfl_xor(send, 1, 64);
                            const unsigned recvk = (unsigned)__shfl_xor((int)sendk, 1, 64);
                            const int e = odd ? eb : ea;
                            const int row = (rb + rbo) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            const long o = pbase - (odd ? 1 : 0) + (long)row * HWp;      // first pixel of the lane pair
                            if (a.pool_out) *reinterpret_cast<float2*>(a.pool_out + o) = odd ? make_float2(recv, keep) : make_float2(keep, recv);
                            *reinterpret_cast<unsigned short*>(a.pool_mask + o) =
                                (unsigned short)(odd ? (recvk | (keepk << 8)) : (keepk | (recvk << 8)));
                        }
                    }
                    continue;
                }
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
                        const float4 t = *reinterpret_cast<const float4*>(sbias + (rb + rbo) * 32 + 8 * e4 + 4 * h);
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
                                const unsigned row = (rb + rbo) * 32 + (e & 3) + 8 * (e >> 2);
                                vals[e] += reinterpret_cast<const float*>(ob + (lo + row * plane))[k];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[k][rb][e] = vals[e] > 0.f ? vals[e] : slope * vals[e];
                    }
                    if (!QOUT || a.out != nullptr) {        // (a q-only launch: every reader of the result takes the q copy)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const unsigned row = (rb + rbo) * 32 + (e & 3) + 8 * (e >> 2);
                            vec_t v;
                            float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
                            for (int k = 0; k < NS; ++k) vp[k] = acc[k][rb][e];
                            *reinterpret_cast<vec_t*>(ob + (lo + row * plane)) = v;
                        }
                    }
                    if constexpr (QOUT) {   // rows 4g .. 4g+3 (+ 4h) of pixel NS*l + k: one half unit per (g, k)
                        if ((a.q_dt == 3 || a.q_dt == 4) && NS % 2 == 0) {
                            // split modes: the half units of pixels (k, k + NS/2) paired across the wave's halves (as 8-byte
                            // stores of a lane's NS consecutive pixels the three planes were 60 of this kernel's 133 us on
                            // N8 C4 512^2 K64: 16 bytes per 64-byte segment and instruction)
                            uint4* qu = reinterpret_cast<uint4*>(a.out_q) + (long)n * a.out_q_nstride + (long)(u0 + ri) * a.Wout + x0 + NS * l;
                            const long psu = (long)a.N * a.out_q_nstride;
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int k = 0; k < NS / 2; ++k) {
                                    uint2 pa[3], pb[3];
                                    thin_split4(acc[k][rb][4 * g], acc[k][rb][4 * g + 1], acc[k][rb][4 * g + 2], acc[k][rb][4 * g + 3], pa);
                                    thin_split4(acc[k + NS / 2][rb][4 * g], acc[k + NS / 2][rb][4 * g + 1], acc[k + NS / 2][rb][4 * g + 2],
                                                acc[k + NS / 2][rb][4 * g + 3], pb);
                                    uint4* qo = qu + (long)((rb + rbo) * 4 + g) * HWout + k + (h ? NS / 2 : 0);
                                    qo[0] = thin_pair_units(pa[0], pb[0]);
                                    qo[psu] = thin_pair_units(pa[1], pb[1]);
                                    if (a.q_dt == 3) qo[2 * psu] = thin_pair_units(pa[2], pb[2]);
                                }
                        } else {
                            uint2* qb = a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(u0 + ri) * a.Wout + x0 + NS * l) + h;
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int k = 0; k < NS; ++k)
                                    thin_qstore4(qb + 2 * ((long)((rb + rbo) * 4 + g) * HWout + k), acc[k][rb][4 * g], acc[k][rb][4 * g + 1],
                                                 acc[k][rb][4 * g + 2], acc[k][rb][4 * g + 3], a.q_dt, 2 * (long)a.N * a.out_q_nstride);
                        }
                    }
                }
            }
        }
        lds_barrier();
    }
}

static int fanout_blocks_per_cu() {
    if (const char* f = GHM_OPT("GHM_FANOUT_BPC")) return atoi(f) > 0 ? atoi(f) : 1;
    return 1;
}

static bool fanout_wave_split() { return GHM_OPT("GHM_FANOUT_NO_SPLIT") == nullptr; }

// RB = row blocks of the layer (filters / 32); WS = 2: eight MFMA waves with RB / 2 row blocks each (see the kernel)
template <int KSTEPS, int RB, int NS, int WS>
static int launch_fanout_ws(ghm_ctx* ctx, const FanoutArgs& a) {
    constexpr int RBW = RB / WS;
    const int NI = a.N * (a.Hout / a.RPI);
    int blocks = ctx->num_cu * fanout_blocks_per_cu();      // persistent blocks, NI / blocks iterations each
    if (blocks > NI) blocks = NI;
    const size_t lds = (size_t)(RB * 32 + 4 * KSTEPS + 2 * a.CS * a.KRT * a.LW) * sizeof(float);
    const dim3 threads((4 * WS + 1) * 64);
    const int which = a.accumulate ? 1 : (a.out_q ? 2 : 0);         // (the q epilogue is its own instantiation: registers)
    GHM_CHECK(!(a.accumulate && a.out_q), "fanout: accumulate and a q output are not combined");
    const void* fn = which == 1 ? (const void*)fanout_kernel<KSTEPS, RBW, NS, true, false, WS>
                   : which == 2 ? (const void*)fanout_kernel<KSTEPS, RBW, NS, false, false, WS, true>
                                : (const void*)fanout_kernel<KSTEPS, RBW, NS, false, false, WS>;
    static bool opted_in[3] = {false, false, false};
    if (!opted_in[which]) {
        GHM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        opted_in[which] = true;
    }
    if (which == 1)
        hipLaunchKernelGGL((fanout_kernel<KSTEPS, RBW, NS, true, false, WS>), dim3(blocks), threads, lds, ctx->stream, a);
    else if (which == 2)
        hipLaunchKernelGGL((fanout_kernel<KSTEPS, RBW, NS, false, false, WS, true>), dim3(blocks), threads, lds, ctx->stream, a);
    else
        hipLaunchKernelGGL((fanout_kernel<KSTEPS, RBW, NS, false, false, WS>), dim3(blocks), threads, lds, ctx->stream, a);
    GHM_LAUNCH_CHECK();
    return 0;
}

template <int KSTEPS, int RB, int NS>
static int launch_fanout_t(ghm_ctx* ctx, const FanoutArgs& a) {
    // the full-resolution forms are bound by their stores (0.135 ms for 537 MB): eight waves measured 5-10 % slower there;
    // the split serves the pooled form (0.166 -> 0.137 ms), GHM_FANOUT_SPLIT_ALL=1 forces it everywhere
    if (RB % 2 == 0 && fanout_wave_split() && GHM_OPT("GHM_FANOUT_SPLIT_ALL"))
        return launch_fanout_ws<KSTEPS, RB, NS, (RB % 2 == 0 ? 2 : 1)>(ctx, a);
    return launch_fanout_ws<KSTEPS, RB, NS, 1>(ctx, a);
}

// geometry shared by the eligibility checks and the launcher: pixels per group, output rows per iteration
static void fanout_plan(int rows_out, int thin_ch, int kh, int kw, int stride, int Hout, int Wout, int* NS, int* RPI,
                        int* KRT, int* LW, size_t* lds) {
    *NS = rows_out <= 64 ? 4 : 2;
    const int gpr = Wout / (*NS * 32);
    int rpi = gpr >= 4 ? 1 : (4 + gpr - 1) / (gpr > 0 ? gpr : 1);      // >= 4 groups per iteration: one per MFMA wave
    if (const char* f = GHM_OPT("GHM_FANOUT_RPI")) rpi = atoi(f) > 0 ? atoi(f) : rpi;      // tuning: output rows per iteration
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

template <int KSTEPS, int WS, bool QOUT>
static int launch_fanout_pool_ws(ghm_ctx* ctx, const FanoutArgs& a) {
    const int NI = a.N * (a.Hout / a.RPI);
    int blocks = ctx->num_cu * fanout_blocks_per_cu();
    if (blocks > NI) blocks = NI;
    const size_t lds = (size_t)(2 * 32 + 4 * KSTEPS + 2 * a.CS * a.KRT * a.LW) * sizeof(float);
    static bool opted_in = false;
    if (!opted_in) {
        GHM_HIP(hipFuncSetAttribute((const void*)fanout_kernel<KSTEPS, 2 / WS, 2, false, true, WS, QOUT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        opted_in = true;
    }
    hipLaunchKernelGGL((fanout_kernel<KSTEPS, 2 / WS, 2, false, true, WS, QOUT>), dim3(blocks), dim3((4 * WS + 1) * 64), lds,
                       ctx->stream, a);
    GHM_LAUNCH_CHECK();
    return 0;
}

template <int KSTEPS>
static int launch_fanout_pool_t(ghm_ctx* ctx, const FanoutArgs& a) {
    if (a.out_q) {          // (eight-wave form only, up to 26 reduction rows: the others have no registers left for it)
        if constexpr (KSTEPS <= 13) {
            GHM_CHECK(fanout_wave_split(), "pooled thin forward with a q output needs the eight-wave form");
            return launch_fanout_pool_ws<KSTEPS, 2, true>(ctx, a);
        } else {
            ghm_set_error("pooled thin forward with a q output: %d reduction rows not served", a.Q);
            return -3;
        }
    }
    return fanout_wave_split() ? launch_fanout_pool_ws<KSTEPS, 2, false>(ctx, a) : launch_fanout_pool_ws<KSTEPS, 1, false>(ctx, a);
}

bool thin_fanout_pool_q_ok(const ghm_conv_desc* d) { return fanout_wave_split() && (d->C * d->kh * d->kw + 1) / 2 <= 13; }

static bool thin_enabled() { return GHM_OPT("GHM_NO_THIN") == nullptr; }
static bool fanout_act_ok(int act) { return act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU; }

// forward conv with <= 4 input channels
bool thin_fanout_fwd_ok(const ghm_conv_desc* d, int act) {
    return thin_enabled() && fanout_act_ok(act) && (long)d->N * d->Ho * d->Wo >= 32768 && (long)d->C * d->H * d->W < (1L << 30) &&
           fanout_geometry_ok(d->K, d->C, d->kh, d->kw, d->stride, d->Ho, d->Wo);
}

int thin_fanout_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                    float* y, int act, float alpha, int accumulate, void* yq, long yq_nstride, int q_dt) {
    FanoutArgs a;
    memset(&a, 0, sizeof(a));
    const int T = d->kh * d->kw;
    GHM_CHECK(!yq || (d->K % 8 == 0 && ((uintptr_t)yq & 15) == 0 && (q_dt == GHM_DTYPE_BF16 || q_dt == GHM_DTYPE_F16 || q_dt == 3 || q_dt == 4)),
              "thin forward with a q output: filters %% 8 == 0, 16-byte aligned q tensor, bf16 / f16");
    a.out_q = (uint2*)yq; a.out_q_nstride = yq_nstride; a.q_dt = q_dt;
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

// forward conv with <= 4 input channels + activation + 2x2 max-pool, fused (64 filters, stride 1, 'same')
bool thin_fanout_fwd_pool_ok(const ghm_conv_desc* d, int act) {
    if (!(thin_enabled() && fanout_act_ok(act) && d->K == 64 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W)) return false;
    const int q = d->C * d->kh * d->kw;
    const size_t lds = (size_t)(2 * 32 + 4 * 18 + 2 * d->C * (d->kh + 1) * ((((d->Wo - 1) + d->kw) + 63) & ~63)) * sizeof(float);
    return d->C <= 4 && q <= THIN_MAX_Q && d->Wo % 64 == 0 && d->Ho % 2 == 0 && (long)d->N * d->Ho * d->Wo >= 32768 &&
           (long)d->C * d->H * d->W < (1L << 30) && lds <= 150 * 1024 && GHM_OPT("GHM_NO_POOL_FUSE") == nullptr;
}

int thin_fanout_fwd_pool(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                         float* pooled, unsigned char* mask, int act, float alpha, void* yq, long yq_nstride, int q_dt) {
    FanoutArgs a;
    memset(&a, 0, sizeof(a));
    GHM_CHECK(!yq || (((uintptr_t)yq & 15) == 0 && (q_dt == GHM_DTYPE_BF16 || q_dt == GHM_DTYPE_F16 || q_dt == 3 || q_dt == 4)),
              "thin pooled forward with a q output: 16-byte aligned q tensor, bf16 / f16");
    a.out_q = (uint2*)yq; a.out_q_nstride = yq_nstride; a.q_dt = q_dt;
    const int T = d->kh * d->kw;
    a.in = x; a.wp = wp; a.bias = bias; a.out = nullptr; a.zeros = ctx->zeros;
    a.pool_out = pooled; a.pool_mask = mask;
    a.N = d->N; a.CS = d->C; a.Hin = d->H; a.Win = d->W; a.in_nstride = d->x_nstride;
    a.R = d->K; a.Hout = d->Ho; a.Wout = d->Wo; a.out_nstride = 0;
    a.ss = 1; a.T = T; a.Q = d->C * T;
    a.w_rs = 1; a.w_cs = (long)T * d->K; a.w_ts = d->K;
    a.act = act; a.alpha = alpha; a.accumulate = 0;
    a.dimin = a.djmin = -d->pad;
    for (int c = 0; c < d->C; ++c)
        for (int ta = 0; ta < d->kh; ++ta)
            for (int tb = 0; tb < d->kw; ++tb) {
                const int q = c * T + ta * d->kw + tb;
                a.ch[q] = c; a.di[q] = ta - d->pad; a.dj[q] = tb - d->pad;
            }
    a.RPI = 2;                                   // one row PAIR per iteration: 8 groups of 2 x 64 pixels at 512 columns
    a.KRT = d->kh + 1;
    a.LW = (((d->Wo - 1) + d->kw) + 63) & ~63;
    for (int q = 0; q < a.Q; ++q) {
        const int cs = q / a.T, t = q - cs * a.T;
        a.loff[q] = (a.ch[q] * a.KRT + (a.di[q] - a.dimin)) * a.LW + (a.dj[q] - a.djmin);
        a.widx[q] = (int)(cs * a.w_cs + t * a.w_ts);
    }
    const int ks = (a.Q + 1) / 2;
    if (ks <= 5) return launch_fanout_pool_t<5>(ctx, a);
    if (ks <= 13) return launch_fanout_pool_t<13>(ctx, a);
    if (ks <= 18) return launch_fanout_pool_t<18>(ctx, a);
    ghm_set_error("fanout pool: %d reduction rows unsupported", a.Q);
    return -3;
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

// ------------------------------------------------------------------------------------------------
// weight gradient of a thin layer:  G[b][q] = sum_{n, p} big[n, b, p] * thin[n, ch(q), p*ss + (di, dj)(q)]
//   thin-in  conv (C <= 4): big = dy (K channels on the output grid), thin = x, dwp[(c*T+t)*K + k] = G[k][(c,t)]
//   thin-out conv (K <= 4, stride 1, 'same'): big = x, thin = dy with the taps negated,
//                                             dwp[(c*T+t)*K + k] = G[c][(k,t)]
// MFMA: rows = big channels, columns = q, reduction = pixels.  The reduction index must be the MFMA k index,
// i.e. lanes along CHANNELS for the A operand, while HBM wants lanes along pixels -- so the big tensor goes
// through LDS: a loader wave (wave 8) DMAs [CB channels] x [PXC pixels] tiles with 16-byte
// global_load_lds (2-4 channel rows per instruction), XOR-swizzling the 16-byte slot on the SOURCE address so
// that the 32 lanes of an A-fragment read (32 channels, same pixels) hit 32 different slots; the thin rows
// are staged as in fanout_kernel by a second loader wave (wave 9; their 4-byte DMAs would otherwise clog the
// 64-entry vmcnt window of the tile loader).  The tile loader runs two tiles ahead with counted vmcnt (DMAs complete in
// order), the eight MFMA waves each reduce an eighth of the tile's pixels into their own accumulators, and
// the block combines them once at the end (fixed order) into its slice of the split-K partial buffer.
// Algorithmic HBM bytes: the big tensor once.
// ------------------------------------------------------------------------------------------------
struct ThinWgradArgs {
    const float* big;
    const float* thin;
    const float* zeros;
    float* part;              // [gridDim.x][n_out]
    int N, CB, Hb, Wb;        // big: CB channels on an Hb x Wb grid
    long big_nstride;
    int CS, Hin, Win;         // thin: CS channels on an Hin x Win grid
    long thin_nstride;
    int ss, Q;
    int n_out;
    long o_bs;                // output index = b*o_bs + oq[q]
    int dimin, djmin, KR, LW;
    int loff[THIN_MAX_Q], oq[THIN_MAX_Q];
    int ch[THIN_MAX_Q], di[THIN_MAX_Q], dj[THIN_MAX_Q];
};

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int RB, int CBQ>
__global__ __launch_bounds__(640) void thin_wgrad_kernel(const ThinWgradArgs a) {
    constexpr int CB = RB * 32;
    constexpr int PXC = 256 / RB;             // pixels per tile: 128 (64 channels) or 64 (128 channels)
    constexpr int SPR = PXC / 4;              // 16-byte slots per channel row
    constexpr int RPU = 64 / SPR;             // channel rows per DMA instruction
    constexpr int TILE = CB * PXC;            // floats
    constexpr int NDMA = TILE / 256;          // DMA instructions per tile (32)
    constexpr int PXW = PXC / 8;              // pixels per MFMA wave
    constexpr int SPH = PXW / 2;              // k-steps per wave and tile
#ifdef GHM_NOSWZ
    constexpr int SWZ = 0;
#else
    constexpr int SWZ = SPR - 1;
#endif
    static_assert(NDMA == 32 && SPH % 4 == 0, "tile geometry");
    extern __shared__ float lds[];
    float* tiles = lds;                       // [3][TILE]
    float* trows = lds + 3 * TILE;            // [2][CS*KR*LW]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    const int rowsz = a.CS * a.KR * a.LW;
    const int NR = a.N * a.Hb;                // big rows
    const int NC = a.Wb / PXC;                // tiles per row
    const int HWb = a.Hb * a.Wb, HWin = a.Hin * a.Win;
    const int G = gridDim.x;

    auto stage_thin = [&](int r, float* dst) {
        const int n = r / a.Hb, u = r - n * a.Hb;
        const float* img = a.thin + (long)n * a.thin_nstride;
        for (int ck = 0; ck < a.CS * a.KR; ++ck) {
            const int c = ck / a.KR, kr = ck - c * a.KR;
            const int y = u * a.ss + a.dimin + kr;
            const bool rowok = (unsigned)y < (unsigned)a.Hin;
            const float* src = img + (long)c * HWin + (rowok ? y : 0) * a.Win + a.djmin;
            float* d = dst + ck * a.LW;
            for (int x0 = 0; x0 < a.LW; x0 += 64) {
                const int xin = x0 + lane + a.djmin;
                const bool ok = rowok && (unsigned)xin < (unsigned)a.Win;
                const float* g = ok ? src + (x0 + lane) : a.zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(d + x0), 4, 0, 0);
            }
        }
    };
    auto stage_big = [&](int r, int c, float* dst) {
        const int n = r / a.Hb, u = r - n * a.Hb;
        // wave-uniform base (scalar) + 32-bit per-lane byte offset
        const char* src = reinterpret_cast<const char*>(a.big + (long)n * a.big_nstride + (long)u * a.Wb + c * PXC);
        int ln = lane;
        asm volatile("" : "+v"(ln));      // recompute the 32 lane offsets per tile instead of keeping them in VGPRs
        const int rin = ln / SPR, slot = ln % SPR;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int b = i * RPU + rin;
            const unsigned boff = ((unsigned)b * (unsigned)HWb + 4u * (slot ^ (b & SWZ))) * 4u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + boff),
                                             (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        }
    };

    // stage sequence of this block: rows r0, r0+G, ...; NC tiles per row
    const int r0 = blockIdx.x;
    const int nrows = (NR - r0 + G - 1) / G;
    const int NT = nrows * NC;
    if (wv == 9) {
        stage_thin(r0, trows);
        wait_vmcnt<0>();
    }
    if (wv == 8) {
        stage_big(r0, 0, tiles);
        if (NT > 1) {
            stage_big(r0 + (1 / NC) * G, 1 % NC, tiles + TILE);      // NC >= 2: tile 1 is in the same row
            wait_vmcnt<NDMA>();
        } else {
            wait_vmcnt<0>();
        }
    }
    __syncthreads();

    int offq[CBQ];
#pragma unroll
    for (int cb = 0; cb < CBQ; ++cb) {
        const int q = cb * 32 + l;
        offq[cb] = q < a.Q ? a.loff[q] : 0;       // dead columns read a finite word and are never written out
    }
    f32x16 acc[RB][CBQ];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CBQ; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][cb][e] = 0.f;

    int t = 0;
    for (int ir = 0; ir < nrows; ++ir) {
        const float* trow = trows + (ir & 1) * rowsz;
        for (int c = 0; c < NC; ++c, ++t) {
            if (wv == 9) {
                // thin rows of the next image row: issued at its predecessor's first tile, landed by its last
                if (c == 0 && ir + 1 < nrows) stage_thin(r0 + (ir + 1) * G, trows + ((ir + 1) & 1) * rowsz);
                if (c == NC - 1) wait_vmcnt<0>();
            } else if (wv == 8) {
                // issue tile t+2; tile t+1 must have landed.  DMAs complete in order: leaving at most as many
                // outstanding as tile t+2 issued means tile t+1 is in LDS.
                const int t2 = t + 2;
                if (t2 < NT) {
                    const int ir2 = t2 / NC, c2 = t2 - ir2 * NC;
                    stage_big(r0 + ir2 * G, c2, tiles + (t2 % 3) * TILE);
                    wait_vmcnt<NDMA>();
                } else {
                    wait_vmcnt<0>();
                }
            } else {
                const float* tile = tiles + (t % 3) * TILE;
                const int px0 = wv * PXW + SPH * h;                      // this lane's first pixel in the tile
                const float* bsrc = trow + (c * PXC + px0) * a.ss;
#pragma unroll
                for (int m = 0; m < SPH / 4; ++m) {
                    float4 av[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        const int b = rb * 32 + l;
                        av[rb] = *reinterpret_cast<const float4*>(tile + b * PXC + 4 * ((px0 / 4 + m) ^ (b & SWZ)));
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        float bv[CBQ];
#pragma unroll
                        for (int cb = 0; cb < CBQ; ++cb) bv[cb] = bsrc[offq[cb] + (4 * m + s4) * a.ss];
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            const float aval = s4 == 0 ? av[rb].x : (s4 == 1 ? av[rb].y : (s4 == 2 ? av[rb].z : av[rb].w));
#pragma unroll
                            for (int cb = 0; cb < CBQ; ++cb)
                                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[cb], acc[rb][cb], 0, 0, 0);
                        }
                    }
                }
            }
            lds_barrier();
        }
    }

    // combine the eight waves (fixed order): waves 4-7 -> 0-3, 2-3 -> 0-1, 1 -> 0; register images are lane-linear
    __syncthreads();
    float* red = lds;
    constexpr int IMG = RB * CBQ * 16 * 64;       // floats per wave image
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wv >= half && wv < 2 * half) {
            float* dst = red + (wv - half) * IMG;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CBQ; ++cb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((rb * CBQ + cb) * 16 + e) * 64 + lane] = acc[rb][cb][e];
        }
        __syncthreads();
        if (wv < half) {
            const float* src = red + wv * IMG;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CBQ; ++cb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[rb][cb][e] += src[((rb * CBQ + cb) * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (wv == 0) {
        float* out = a.part + (long)blockIdx.x * a.n_out;
#pragma unroll
        for (int cb = 0; cb < CBQ; ++cb) {
            const int q = cb * 32 + l;
            if (q < a.Q) {
                const int oq = a.oq[q];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int b = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        out[b * a.o_bs + oq] = acc[rb][cb][e];
                    }
            }
        }
    }
}

static bool thin_wgrad_geometry_ok(int big_ch, int thin_ch, int kh, int kw, int stride, int Hb, int Wb, int N) {
    const int q = thin_ch * kh * kw;
    if (!(big_ch == 64 || (big_ch == 128 && q <= 32)) || q > THIN_MAX_Q) return false;
    const int pxc = big_ch == 64 ? 128 : 64;
    if (Wb % pxc != 0 || Wb / pxc < 2 || (long)N * Hb * Wb < 32768) return false;
    const int LW = (((Wb - 1) * stride + kw) + 63) & ~63;
    const size_t lds = (size_t)(3 * big_ch * pxc + 2 * thin_ch * kh * LW) * sizeof(float);
    return lds <= 158 * 1024;
}

// which side is thin?  0 = neither, 1 = input channels (C <= 4), 2 = filters (K <= 4)
static int thin_wgrad_side(const ghm_conv_desc* d, const float* x, const float* dy) {
    if (!thin_enabled()) return 0;
    if (d->C <= 4 && d->K > 4 && thin_wgrad_geometry_ok(d->K, d->C, d->kh, d->kw, d->stride, d->Ho, d->Wo, d->N) &&
        ((uintptr_t)dy % 16 == 0) && d->y_nstride % 4 == 0 && (d->Ho * d->Wo) % 4 == 0)
        return 1;
    if (d->K <= 4 && d->C > 4 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W &&
        thin_wgrad_geometry_ok(d->C, d->K, d->kh, d->kw, 1, d->H, d->W, d->N) && ((uintptr_t)x % 16 == 0) &&
        d->x_nstride % 4 == 0 && (d->H * d->W) % 4 == 0)
        return 2;
    return 0;
}

bool thin_wgrad_ok(const ghm_conv_desc* d, const float* x, const float* dy) { return thin_wgrad_side(d, x, dy) != 0; }

int thin_wgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* dy, float* dwp, int accumulate) {
    const int side = thin_wgrad_side(d, x, dy);
    GHM_CHECK(side != 0, "thin_wgrad: geometry not supported");
    ThinWgradArgs a;
    memset(&a, 0, sizeof(a));
    const int T = d->kh * d->kw;
    a.zeros = ctx->zeros;
    a.N = d->N;
    a.n_out = d->C * T * d->K;
    if (side == 1) {
        a.big = dy; a.CB = d->K; a.Hb = d->Ho; a.Wb = d->Wo; a.big_nstride = d->y_nstride;
        a.thin = x; a.CS = d->C; a.Hin = d->H; a.Win = d->W; a.thin_nstride = d->x_nstride;
        a.ss = d->stride; a.Q = d->C * T; a.o_bs = 1;
        for (int c = 0; c < d->C; ++c)
            for (int ta = 0; ta < d->kh; ++ta)
                for (int tb = 0; tb < d->kw; ++tb) {
                    const int q = c * T + ta * d->kw + tb;
                    a.ch[q] = c; a.di[q] = ta - d->pad; a.dj[q] = tb - d->pad;
                    a.oq[q] = q * d->K;
                }
    } else {
        a.big = x; a.CB = d->C; a.Hb = d->H; a.Wb = d->W; a.big_nstride = d->x_nstride;
        a.thin = dy; a.CS = d->K; a.Hin = d->Ho; a.Win = d->Wo; a.thin_nstride = d->y_nstride;
        a.ss = 1; a.Q = d->K * T; a.o_bs = (long)T * d->K;
        for (int k = 0; k < d->K; ++k)
            for (int ta = 0; ta < d->kh; ++ta)
                for (int tb = 0; tb < d->kw; ++tb) {
                    const int q = k * T + ta * d->kw + tb;
                    a.ch[q] = k; a.di[q] = d->pad - ta; a.dj[q] = d->pad - tb;
                    a.oq[q] = (ta * d->kw + tb) * d->K + k;
                }
    }
    int dimax = -(1 << 20), djmax = -(1 << 20);
    a.dimin = a.djmin = 1 << 20;
    for (int q = 0; q < a.Q; ++q) {
        if (a.di[q] < a.dimin) a.dimin = a.di[q];
        if (a.di[q] > dimax) dimax = a.di[q];
        if (a.dj[q] < a.djmin) a.djmin = a.dj[q];
        if (a.dj[q] > djmax) djmax = a.dj[q];
    }
    a.KR = dimax - a.dimin + 1;
    a.LW = (((a.Wb - 1) * a.ss + (djmax - a.djmin) + 1) + 63) & ~63;
    for (int q = 0; q < a.Q; ++q) a.loff[q] = (a.ch[q] * a.KR + (a.di[q] - a.dimin)) * a.LW + (a.dj[q] - a.djmin);
    const int NR = a.N * a.Hb;
    int blocks = ctx->num_cu;
    if (blocks > NR) blocks = NR;
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, (size_t)blocks * a.n_out * sizeof(float), &ws)) return e;
    a.part = (float*)ws;
    const int pxc = a.CB == 64 ? 128 : 64;
    const size_t lds = (size_t)(3 * a.CB * pxc + 2 * a.CS * a.KR * a.LW) * sizeof(float);
    const int cbq = (a.Q + 31) / 32;
#define GHM_TW_CASE(RB_, CBQ_)                                                                              \
    if (a.CB == RB_ * 32 && cbq == CBQ_) {                                                                  \
        static bool opted = false;                                                                          \
        if (!opted) {                                                                                       \
            GHM_HIP(hipFuncSetAttribute((const void*)thin_wgrad_kernel<RB_, CBQ_>,                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));            \
            opted = true;                                                                                   \
        }                                                                                                   \
        hipLaunchKernelGGL((thin_wgrad_kernel<RB_, CBQ_>), dim3(blocks), dim3(640), lds, ctx->stream, a);   \
        GHM_LAUNCH_CHECK();                                                                                 \
    } else
    GHM_TW_CASE(2, 1)
    GHM_TW_CASE(2, 2)
    GHM_TW_CASE(4, 1) {
        ghm_set_error("thin_wgrad: no variant for %d channels x %d columns", a.CB, a.Q);
        return -3;
    }
#undef GHM_TW_CASE
    return ghm_reduce_splits(ctx, a.part, blocks, a.n_out, a.n_out, dwp, accumulate);
}

// ------------------------------------------------------------------------------------------------
// many -> few, stride 1, 'same' geometry (g_out forward; data gradient of d_conv1):
//   out[n, j, p] = act(bias[j] + sum_t sum_c W[(j,t)][c] * in[n, c, p + off_t]),   JS*T <= 32 rows (j, t)
// Taps as MFMA rows: S[(j,t)][q] = sum_c W[(j,t)][c] * in[c][q] is a 32 x pixels x CB GEMM whose B operand
// (two channels x 32 consecutive pixels per k-step) comes straight from global memory with lanes along
// pixels; the block computes S over its 16x64 output tile plus the filter halo into LDS and a second phase
// sums the T shifted planes per output pixel.  Replaces the unfused form (S through HBM: 25 planes written
// and re-read).  Algorithmic HBM bytes: the wide input once (x1.33 with the halo, mostly L2 hits).
// ------------------------------------------------------------------------------------------------
struct FaninS1Args {
    const float* in;
    const float* wp;
    const float* bias;
    float* out;
    int N, CB, H, W;
    long in_nstride, out_nstride;
    int JS, KS, T, pad_lo;        // halo origin = tile origin - pad_lo
    long w_cs;                    // weight stride between big channels
    int wbase[32];                // weight offset of row (j, t); -1 for dead rows
    int soff[32];                 // LDS offset of tap t relative to the output pixel (halo coordinates), per row
    int act;
    float alpha;
    int accumulate;
};

template <int TH, int TW, int KS, int KSTEPS>
__global__ __launch_bounds__(512) void fanin_s1_kernel(const FaninS1Args a) {
    constexpr int HH = TH + KS - 1, HW_ = TW + KS - 1, NQ = HH * HW_;
    constexpr int T = KS * KS;
    extern __shared__ float S[];          // [T*JS rows used][NQ]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int h = lane >> 5, l = lane & 31;
    const int tiles_x = a.W / TW, tiles_y = a.H / TH;
    // neighbouring tiles share halo pixels and the partly used 128-byte lines at their edges: keep them on one XCD's L2
    // (PMC: 710 MB fetched for 268 MB of input before)
    const int bid = ghm_xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / (tiles_x * tiles_y), trem = bid - n * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int y0 = ty * TH - a.pad_lo, x0 = tx * TW - a.pad_lo;      // halo origin (may be negative)
    const int HWp = a.H * a.W;
    const int R = a.JS * T;

    // A fragments: lane (l, h) holds W[row l][c = 2s + h] for every k-step s (CB/2 <= 64 registers)
    float A[KSTEPS];
    const int wb = l < R ? a.wbase[l] : -1;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const float w = a.wp[wb >= 0 ? wb + (long)(2 * s + h) * a.w_cs : 0];
        A[s] = wb >= 0 ? w : 0.f;
    }
    const float* img = a.in + (long)n * a.in_nstride + (long)h * HWp;     // this half-wave's channel parity
    constexpr int NSTRIP = (NQ + 31) / 32;
    float Bn[KSTEPS];
    bool okn = false;
    auto fetch = [&](int strip) {
        const int q = strip * 32 + l;
        const int qy = q / HW_, qx = q - qy * HW_;
        const int y = y0 + qy, x = x0 + qx;
        okn = q < NQ && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        const float* src = img + (okn ? y * a.W + x : 0);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) Bn[s] = src[(long)(2 * s) * HWp];
    };
    if (wv < NSTRIP) fetch(wv);
    for (int strip = wv; strip < NSTRIP; strip += 8) {
        float Bc[KSTEPS];
        const bool ok = okn;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) Bc[s] = ok ? Bn[s] : 0.f;
        if (strip + 8 < NSTRIP) fetch(strip + 8);      // next strip's 2 x KSTEPS lines are in flight under the MFMAs
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], Bc[s], acc, 0, 0, 0);
        const int q = strip * 32 + l;
        if (q < NQ) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < R) S[row * NQ + q] = acc[e];
            }
        }
    }
    __syncthreads();
    // shift-add: TH*TW output pixels x JS channels, consecutive threads on consecutive pixels
    for (int i = threadIdx.x; i < TH * TW * a.JS; i += 512) {
        const int j = i / (TH * TW), p = i - j * (TH * TW);
        const int py = p / TW, px = p - py * TW;
        const float* s0 = S + (j * T) * NQ + py * HW_ + px;
        float v = a.bias ? a.bias[j] : 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) v += s0[t * NQ + a.soff[t]];
        float* o = a.out + (long)n * a.out_nstride + (long)j * HWp + (ty * TH + py) * a.W + tx * TW + px;
        if (a.accumulate) v += *o;
        *o = ghm_act(v, a.act, a.alpha);
    }
}

static bool fanin_s1_geometry_ok(int thin, int big, int kh, int kw, int H, int W, int N) {
    return thin_enabled() && kh == kw && (kh == 5 || kh == 3) && thin * kh * kw <= 32 && (big == 64 || big == 128) &&
           H % 16 == 0 && W % 64 == 0 && (long)N * H * W >= 32768;
}

// forward of a conv with <= 4 filters (stride 1, same size)
bool thin_fanin_s1_fwd_ok(const ghm_conv_desc* d) {
    return d->stride == 1 && d->Ho == d->H && d->Wo == d->W && fanin_s1_geometry_ok(d->K, d->C, d->kh, d->kw, d->H, d->W, d->N);
}
// data gradient of a conv with <= 4 input channels (stride 1, same size)
bool thin_fanin_s1_dgrad_ok(const ghm_conv_desc* d) {
    return d->stride == 1 && d->Ho == d->H && d->Wo == d->W && fanin_s1_geometry_ok(d->C, d->K, d->kh, d->kw, d->H, d->W, d->N);
}

static int launch_fanin_s1(ghm_ctx* ctx, FaninS1Args& a) {
    constexpr int TH = 16, TW = 64;
    const int hw = TW + a.KS - 1, nq = (TH + a.KS - 1) * hw;
    const size_t lds = (size_t)a.JS * a.T * nq * sizeof(float);
    const dim3 grid(a.N * (a.H / TH) * (a.W / TW));
#define GHM_FS1_CASE(KS_, KST_)                                                                             \
    if (a.KS == KS_ && a.CB == 2 * KST_) {                                                                  \
        static bool opted = false;                                                                          \
        if (!opted) {                                                                                       \
            GHM_HIP(hipFuncSetAttribute((const void*)fanin_s1_kernel<TH, TW, KS_, KST_>,                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));            \
            opted = true;                                                                                   \
        }                                                                                                   \
        hipLaunchKernelGGL((fanin_s1_kernel<TH, TW, KS_, KST_>), grid, dim3(512), lds, ctx->stream, a);     \
        GHM_LAUNCH_CHECK();                                                                                 \
        return 0;                                                                                           \
    }
    GHM_FS1_CASE(5, 32)
    GHM_FS1_CASE(5, 64)
    GHM_FS1_CASE(3, 32)
    GHM_FS1_CASE(3, 64)
#undef GHM_FS1_CASE
    ghm_set_error("fanin_s1: no variant for k=%d", a.KS);
    return -3;
}

int thin_fanin_s1_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                      float* y, int act, float alpha, int accumulate) {
    FaninS1Args a;
    memset(&a, 0, sizeof(a));
    const int T = d->kh * d->kw, hw = 64 + d->kw - 1;
    a.in = x; a.wp = wp; a.bias = bias; a.out = y;
    a.N = d->N; a.CB = d->C; a.H = d->H; a.W = d->W; a.in_nstride = d->x_nstride; a.out_nstride = d->y_nstride;
    a.JS = d->K; a.KS = d->kh; a.T = T; a.pad_lo = d->pad;
    a.w_cs = (long)T * d->K;
    for (int r = 0; r < 32; ++r) a.wbase[r] = -1;
    for (int j = 0; j < d->K; ++j)
        for (int ta = 0; ta < d->kh; ++ta)
            for (int tb = 0; tb < d->kw; ++tb) {
                const int t = ta * d->kw + tb;
                a.wbase[j * T + t] = t * d->K + j;
                a.soff[t] = ta * hw + tb;                 // out(py,px) reads in(py + ta - pad, px + tb - pad)
            }
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    return launch_fanin_s1(ctx, a);
}

int thin_fanin_s1_dgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                        float* dx, int act, float alpha, int accumulate) {
    FaninS1Args a;
    memset(&a, 0, sizeof(a));
    const int T = d->kh * d->kw, hw = 64 + d->kw - 1;
    a.in = dy; a.wp = wp; a.bias = bias; a.out = dx;
    a.N = d->N; a.CB = d->K; a.H = d->H; a.W = d->W; a.in_nstride = d->y_nstride; a.out_nstride = d->x_nstride;
    a.JS = d->C; a.KS = d->kh; a.T = T; a.pad_lo = d->kh - 1 - d->pad;
    a.w_cs = 1;
    for (int r = 0; r < 32; ++r) a.wbase[r] = -1;
    for (int c = 0; c < d->C; ++c)
        for (int ta = 0; ta < d->kh; ++ta)
            for (int tb = 0; tb < d->kw; ++tb) {
                const int t = ta * d->kw + tb;
                a.wbase[c * T + t] = (c * T + t) * d->K;
                // dx(u) += W[tap] * dy(u + pad - ta): halo coordinate = (pad - ta) + pad_lo = kh - 1 - ta
                a.soff[t] = (d->kh - 1 - ta) * hw + (d->kw - 1 - tb);
            }
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    return launch_fanin_s1(ctx, a);
}
