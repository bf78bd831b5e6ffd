// fp32 convolutions on the bf16 matrix cores by operand splitting (gfx950 has no fast fp32 MFMA: v_mfma_f32_32x32x2_f32 runs
// at 1/16 of the bf16 rate).  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significant bits, bf16 has the
// exponent range of fp32):
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1
// and the product of two bf16 values is exact in fp32, so
//     x * w = sum over (i, j) of x_i * w_j.
// The kernels here run the six products with i + j <= 2 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three that
// are dropped are below 2^-24 |x w| each -- the size of the rounding fp32 arithmetic itself commits on every product.  This
// is the operand-splitting form of fp32 emulation (BF16x9 in vendor BLAS libraries keeps all nine products; six is the
// fp32-accurate subset): nothing about the data is rounded to 8 bits.  Measured against the float64 oracle the results carry
// LESS error than the fp32 MFMA kernels' (tests/test_gpu_split.py: 0.7-2.2e-7 against 1.4-2.7e-7 -- sixteen exact products
// are summed per instruction before the first rounding), at 6/16 of their matrix-core time.
//
// Layouts: a "split q tensor" is three planes (pieces 0, 1, 2), each a bf16 q tensor as include/ghm.h describes it
// ([N][C/8][H][W][8 channels], 16-byte units); a "split weight pack" is three planes of the low-precision pack
// wq[c/8][tap][Rpad][8] (conv_lp.hip).  ghm_split_pack / ghm_split_pack_weights produce them from fp32.
//
// Reference: the convolutions of architectures/dcgan.py:16-50 and architectures/p2p.py:139-290 in floatX=float32
// (experiment.5.sh:5).  An OPT-IN arithmetic form of the fp32 mode: Pix2Pix / GanStep(dtype='bf16x3'), bench.py --dtype bf16x3.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 sp_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// two fp32 values -> their three bf16 pieces, packed pairwise (low half = first value)
__device__ __forceinline__ void sp_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    f32x2 v = {a, b};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));          // RNE
    f32x2 r = {a - __uint_as_float(p0 << 16), b - __uint_as_float(p0 & 0xffff0000u)};      // exact
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 r2 = {r[0] - __uint_as_float(p1 << 16), r[1] - __uint_as_float(p1 & 0xffff0000u)};  // exact, <= 8 significant bits
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));         // exact
}

__device__ __forceinline__ void sp_split8(const float* v, u32x4& q0, u32x4& q1, u32x4& q2) {
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) sp_split2(v[2 * t], v[2 * t + 1], a[t], b[t], c[t]);
    q0 = u32x4{a[0], a[1], a[2], a[3]};
    q1 = u32x4{b[0], b[1], b[2], b[3]};
    q2 = u32x4{c[0], c[1], c[2], c[3]};
}

// a half unit (4 consecutive channels of one pixel) of a split q tensor: the NPC pieces of each value, NPC planes
// ``ps2`` half-units apart (the producer writes whole tensors: plane stride = samples x sample stride).  NPC = 2 ('bf16x2'):
// the first two pieces, x1 = bf16(x - x0) rounded to nearest -- x0 + x1 carries 16-17 significant bits
template <int NPC>
__device__ __forceinline__ void sp_qstore4(uint2* qo, float v0, float v1, float v2, float v3, long ps2) {
    unsigned a0, a1, a2, b0, b1, b2;
    sp_split2(v0, v1, a0, a1, a2);
    sp_split2(v2, v3, b0, b1, b2);
    qo[0] = make_uint2(a0, b0);
    qo[ps2] = make_uint2(a1, b1);
    if (NPC == 3) qo[2 * ps2] = make_uint2(a2, b2);
}

__device__ __forceinline__ int sp_xcd_remap(int bid, int nb) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// fp32 NCHW view -> split q tensor: one thread per 16-byte unit (8 channels of one pixel), three stores
template <int NPC>
__global__ __launch_bounds__(256) void sp_pack_kernel(const float* __restrict__ x, long x_nstride, int N, int C8, int HW,
                                                      u32x4* __restrict__ q, long q_nstride, long q_pstride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * C8 * HW) return;
    const int p = (int)(idx % HW);
    const long nc = idx / HW;
    const int cb = (int)(nc % C8), n = (int)(nc / C8);
    const float* g = x + (long)n * x_nstride + (long)cb * 8 * HW + p;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g[(long)j * HW];
    u32x4 q0, q1, q2;
    sp_split8(v, q0, q1, q2);
    u32x4* o = q + (long)n * q_nstride + (long)cb * HW + p;
    o[0] = q0;
    o[q_pstride] = q1;
    if (NPC == 3) o[2 * q_pstride] = q2;
}

// packed fp32 weights wp[c][tap][r] -> split pack (three planes of wq[c/8][tap][Rpad][8]); transposed: the data-gradient
// operand wqT[k/8][T-1-tap][Cpad][8 k] = wp[c][tap][k] (conv_lp.hip, lp_pack_batched_kernel)
template <int NPC>
__global__ __launch_bounds__(256) void sp_pack_w_kernel(const float* __restrict__ wp, u32x4* __restrict__ wq, int red, int T,
                                                        int rows, int nblk, int rpad, int transposed, long pstride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)nblk * T * rpad) return;
    const int r = (int)(idx % rpad);
    const long bt = idx / rpad;
    const int tap = (int)(bt % T), cb = (int)(bt / T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const bool ok = c < red && r < rows;
        const long src = transposed ? ((long)r * T + (T - 1 - tap)) * red + c : ((long)c * T + tap) * rows + r;
        v[j] = ok ? wp[src] : 0.f;
    }
    u32x4 q0, q1, q2;
    sp_split8(v, q0, q1, q2);
    wq[idx] = q0;
    wq[idx + pstride] = q1;
    if (NPC == 3) wq[idx + 2 * pstride] = q2;
}

// every pack of a net in ONE launch: the device table of ghm_lp_pack_batched (48-byte items, include/ghm.h); an item's three
// planes are nblk * T * rpad units apart
struct SpPackItem {
    const float* wp;
    u32x4* wq;
    int red, T, rows, nblk, rpad, transposed, block_begin, pad_;
};

template <int NPC>
__global__ __launch_bounds__(256) void sp_pack_w_batched_kernel(const SpPackItem* __restrict__ items, int n) {
    int li = 0;
    while (li + 1 < n && (int)blockIdx.x >= items[li + 1].block_begin) ++li;
    const SpPackItem it = items[li];
    const long idx = (long)(blockIdx.x - it.block_begin) * 256 + threadIdx.x;
    const long plane = (long)it.nblk * it.T * it.rpad;
    if (idx >= plane) return;
    const int r = (int)(idx % it.rpad);
    const long bt = idx / it.rpad;
    const int tap = (int)(bt % it.T), cb = (int)(bt / it.T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const bool ok = c < it.red && r < it.rows;
        const long src = it.transposed ? ((long)r * it.T + (it.T - 1 - tap)) * it.red + c : ((long)c * it.T + tap) * it.rows + r;
        v[j] = ok ? it.wp[src] : 0.f;
    }
    u32x4 q0, q1, q2;
    sp_split8(v, q0, q1, q2);
    it.wq[idx] = q0;
    it.wq[idx + plane] = q1;
    if (NPC == 3) it.wq[idx + 2 * plane] = q2;
}

// ------------------------------------------------------------------------------------------------
// Forward-form convolution (forward; stride-1 data gradient on the transposed pack; 3x3 stride-2 forward), the structure of
// lp_conv_kernel (conv_lp.hip) with three pieces of each operand in LDS.  A block owns BM output channels x (RT rows x 32
// columns) of one image.  K loop: slabs of 16 input channels; inside a slab one iteration per filter row (KS k-steps, one
// per filter column).  A k-step reads 3 A fragments per row tile and 3 B fragments per pixel tile and runs SIX MFMAs per
// (row tile, pixel tile): half the LDS reads per MFMA of the one-piece kernel.
// ------------------------------------------------------------------------------------------------
struct SpConvArgs {
    const u32x4* in_q;     // split q tensor of the conv input (forward) / output gradient (data gradient)
    long in_q_nstride;     // units between samples
    long in_q_pstride;     // units between pieces
    const u32x4* zeros;    // >= 16 bytes of zeros in HBM: the DMA source of padding pixels
    const u32x4* wq;       // split weight pack
    long wq_pstride;
    const float* bias;
    float* out;            // fp32 NCHW output (or null when only the split q copy is wanted)
    uint2* out_q;          // split q tensor of the result (half units; of the POOLED result when POOL) or null
    long out_q_nstride;    // 16-byte units between samples (planes: N x this apart)
    float* partial;
    int N, CH, H, W;       // OUTPUT grid
    int Hin, Win;
    int R, Rpad;
    long out_nstride;
    int pad, act;
    float alpha;
    int accumulate;
    int slabs_per_split;
    int ntiles;                 // (filter tile, pixel tile, sample) tiles of the launch; a block walks blockIdx.x, + gridDim.x, ...
    float* pool_out;            // POOL: [N, R, H/2, W/2] maximum of act(conv + bias) over 2x2 windows ...
    unsigned char* pool_mask;   // ... and the arg-max mask of every window (bit 2*dr + dc; all ties set; bit 4: sign)
    int cls_k;                  // CLS forms: filters (forward) / reduction channels (data gradient) per parity class
    int cls_group;              // CLS = 1: pixel tiles per dispatch group (see decode)
};

// the piece products of one multiply-add, small terms first, the leading product x0 w0 last: all (i, j) with i + j < NPC.
// NPC = 3 ('bf16x3'): six products, every term above 2^-24 of the product -- fp32-accurate.  NPC = 2 ('bf16x2'): x0 w0 + x1 w0 +
// x0 w1, what is dropped is 2^-16 of the product: 16-bit operands at half the matrix-core time, BASELINE config 4's arithmetic
template <int NPC>
struct SpProd;
template <>
struct SpProd<3> {
    static constexpr int N = 6;
    static constexpr int A[6] = {2, 1, 0, 1, 0, 0}, B[6] = {0, 1, 2, 0, 1, 0};
};
template <>
struct SpProd<2> {
    static constexpr int N = 3;
    static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0};
};

// the epilogue of the forward-form kernels: fp32 NCHW and / or the split q copy, split-K partials, or the pooled form
template <int BM, int RT, int WM, int WN, bool POOL, int TW, int TM, int TN, int NPC>
__device__ __forceinline__ void sp_conv_epilogue(const SpConvArgs& a, f32x16 (&acc)[TM][TN], f32x16 (&accc)[TM][TN], u32x4* sp_smem,
                                                 int tid, int wm, int wn, int kg, int li, int lx, int ly, int n, int r0, int y0,
                                                 int x0, int HW) {
    constexpr int RPF = 32 / TW;
    // ---- epilogue (fp32): lanes along pixels; row of element e of tile i: i*32 + (e&3) + 8*(e>>2) + 4*kg ----
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] += accc[i][j][e];
    const long P = (long)a.N * HW;
    const int ru = r0 + wm * (BM / WM);
    const int rl = ru + 4 * kg;
    if (a.partial) {
        float* const pb = a.partial + ((long)blockIdx.y * a.R + ru) * P + (long)n * HW + (long)(y0 + wn * TN * RPF) * a.W + x0;
        const unsigned lo = 4u * kg * (unsigned)P + ly * a.W + lx;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    float* rowp = pb + (long)k * P + j * RPF * a.W;
                    if (rl + k < a.R) rowp[lo] = acc[i][j][e];
                }
        return;
    }
    float* const sb = reinterpret_cast<float*>(sp_smem);          // bias through LDS (free after the last barrier)
    if (tid < BM) sb[tid] = (a.bias && r0 + tid < a.R) ? a.bias[r0 + tid] : 0.f;
    __syncthreads();
    const float* const lb = sb + wm * (BM / WM) + 4 * kg;
    const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
    const bool pwl = a.act == GHM_ACT_LINEAR || a.act == GHM_ACT_RELU || a.act == GHM_ACT_LRELU;
    if constexpr (POOL) {
        // 2x2 max-pool of act(conv + bias): the row pair of a window is in one lane, the column pair in lanes (2t, 2t+1)
        static_assert(!POOL || (TN % 2 == 0 && TW == 32), "pooled epilogue: row pairs inside a wave, 32-column tiles (stride 1)");
        const int Wp = a.W / 2;
        const long HWp = (long)(a.H / 2) * Wp;
        const long pix = (long)((y0 + wn * TN) / 2) * Wp + (x0 + li) / 2;
        const long base = ((long)n * a.R + rl) * HWp + pix;
        const bool even = (li & 1) == 0;
        uint2* const qb = a.out_q ? a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(ru / 8) * HWp + pix) + kg : nullptr;
        const long ps2 = 2 * (long)a.N * a.out_q_nstride;
#pragma unroll
        for (int j2 = 0; j2 < TN / 2; ++j2)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float mq[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = 4 * g + t, k = i * 32 + t + 8 * g;
                        float v0 = acc[i][2 * j2][e] + lb[k], v1 = acc[i][2 * j2 + 1][e] + lb[k];
                        v0 = v0 > 0.f ? v0 : slope * v0;
                        v1 = v1 > 0.f ? v1 : slope * v1;
                        const float w0 = __shfl_xor(v0, 1, 64), w1 = __shfl_xor(v1, 1, 64);
                        const float m = fmaxf(fmaxf(v0, v1), fmaxf(w0, w1));
                        const unsigned mk = (v0 == m ? 1u : 0u) | (w0 == m ? 2u : 0u) | (v1 == m ? 4u : 0u) | (w1 == m ? 8u : 0u) |
                                            (m > 0.f ? GHM_POOL_SIGN : 0u);
                        mq[t] = m;
                        if (even && rl + k < a.R) {
                            const long o = base + (long)k * HWp + j2 * Wp;
                            if (a.pool_out) a.pool_out[o] = m;
                            a.pool_mask[o] = (unsigned char)mk;
                        }
                    }
                    if (qb && even && rl + i * 32 + 8 * g < a.R)
                        sp_qstore4<NPC>(qb + 2 * ((long)(i * 4 + g) * HWp + j2 * Wp), mq[0], mq[1], mq[2], mq[3], ps2);
                }
        return;
    }
    float* const ub = a.out ? a.out + (long)n * a.out_nstride + (long)ru * HW + (long)(y0 + wn * TN * RPF) * a.W + x0 : nullptr;
    const unsigned lo = 4u * kg * (unsigned)HW + ly * a.W + lx;
    const bool full = r0 + BM <= a.R;
    uint2* const qb = a.out_q ? a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(ru / 8) * HW +
                                                 (long)(y0 + wn * TN * RPF + ly) * a.W + x0 + lx) + kg : nullptr;
    const long ps2 = 2 * (long)a.N * a.out_q_nstride;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                v[e] = acc[i][j][e] + lb[k];
            }
            if (a.accumulate) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (full || rl + k < a.R) v[e] += (ub + (long)k * HW + j * RPF * a.W)[lo];
                }
            }
            if (pwl) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = ghm_act(v[e], a.act, a.alpha);
            }
            if (ub) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (full || rl + k < a.R) (ub + (long)k * HW + j * RPF * a.W)[lo] = v[e];
                }
            }
            if (qb) {       // e = 4g .. 4g+3: four consecutive channels of this lane's pixel = half a q unit (kg picks the half)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (full || rl + i * 32 + 8 * g < a.R)
                        sp_qstore4<NPC>(qb + 2 * ((long)(i * 4 + g) * HW + j * RPF * a.W), v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], ps2);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// sp_conv2_kernel (round 5): the tile, the LDS images and the arithmetic of sp_conv_kernel with a BRANCH-FREE, software-
// pipelined K loop.  What the ISA of the first form showed (one wave per SIMD, so nothing hides a wave's own bubbles): every
// iteration began with ~440 scalar / vector instructions in 75 basic blocks -- each exec-masked LDS-DMA instruction sat behind
// its own branch -- issued with the matrix pipe idle, and the first filter column's fragments were read behind serial
// ``s_waitcnt lgkmcnt(0)``: together the 40 % of the cycles in which SQ_VALU_MFMA_BUSY_CYCLES did not count.  Here
//   * every wave issues the SAME number of DMA instructions per iteration, unconditionally: a lane outside the image reads the
//     zero unit, a chunk outside the patch / weight tile lands in a 1 KB scratch chunk of LDS -- no exec masks, no branches;
//     addresses are a uniform base + a per-lane 32-bit offset fixed at kernel start (weights) or a per-lane pointer walked by
//     per-lane increments (patch: 0 for padding lanes);
//   * the whole slab (KS filter rows x KS columns) is one basic block; the DMA of iteration it + 2 and the fragment reads of
//     the next k-step are placed between the MFMAs of the current one (sched_group_barrier);
//   * the barrier of an iteration sits at the end of its second-last column: behind it the last column's MFMAs run while the
//     first fragments of the next iteration are read and the weights two iterations ahead are requested, so a DMA has a full
//     iteration to land and no fragment read is exposed;
//   * past the end of the contraction the pipeline keeps issuing (clamped to valid addresses): no peeled tail.
// ------------------------------------------------------------------------------------------------
template <int KS, int ST, int BM, int RT, int WM, int WN, int TW, int NP>
struct SpGeo2 {
    static constexpr int RPF = 32 / TW, ROWS = RT * RPF;
    static constexpr int PH = (ROWS - 1) * ST + KS, PW = (TW - 1) * ST + KS;
    static constexpr int PU1 = 2 * PH * PW, PCH = (PU1 + 63) / 64, PUP = PCH * 64;   // patch units / 64-unit chunks / padded
    static constexpr int WU1 = 2 * KS * BM, NI = WU1 / 64;
    static constexpr int NW = WM * WN;
    static constexpr int NQ = (PCH + NW - 1) / NW;                 // patch DMA instructions per wave and piece
    static constexpr int NIW = (NP * NI + NW - 1) / NW;            // weight DMA instructions per wave and iteration
    static constexpr int WUNITS = NP * WU1, PUNITS = NP * PUP;
    static constexpr int SCR = 2 * WUNITS + 2 * PUNITS;            // the scratch chunk
    static constexpr int LDS_BYTES = (SCR + 64 + 32) * 16;         // + 512 bytes: the epilogue's bias staging
};

// ABL (tuning only, wrong results): 1 = no DMA inside the loop, 2 = no waits / barriers inside the loop, 4 = no fragment reads
// inside the loop -- what each costs beside the MFMA stream itself
// CLS (3x3 stride 1 only): the collapsed form of BilinearUpsample2DLayer(2) -> 3x3 conv (conv_bilinear.hip) is a 3x3 convolution
// with 4K filters ordered (parity class p q, k) in which class (p, q) has no taps in filter row 0 when p = 1 and none in filter
// column 0 when q = 1 (9 / 6 / 6 / 4 of the 9 taps): the structural zeros are SKIPPED, k-step by k-step.
//   CLS = 1 (forward): the class is a property of the block's filter tile (a.cls_k filters per class, a multiple of BM): its
//     slabs run filter rows p .. 2 and columns q .. 2;
//   CLS = 2 (data gradient on the transposed pack, whose taps are flipped): the class is a property of the SLAB (a.cls_k
//     reduction channels per class): a slab of class (p, q) runs filter rows 0 .. 2 - p and columns 0 .. 2 - q.
// Same pipeline, same products in the same order as the full kernel minus the k-steps whose weights are all zero: results are
// bit-identical to CLS = 0 (tests/test_gpu_split.py).
template <int V>
struct SpIC {
    static constexpr int value = V;
};

template <int KS, int ST, int BM, int RT, int WM, int WN, bool POOL, int TW, int NP, int ABL = 0, int CLS = 0>
__global__ __launch_bounds__(WM * WN * 64, 1) void sp_conv2_kernel(const SpConvArgs a) {
    static_assert(CLS == 0 || (KS == 3 && ST == 1 && !POOL && ABL == 0), "class forms: 3x3 stride 1");
    typedef SpGeo2<KS, ST, BM, RT, WM, WN, TW, NP> G;
    typedef SpProd<NP> PR;
    constexpr int T = KS * KS;
    constexpr int RPF = G::RPF, ROWS = G::ROWS;
    constexpr int TM = BM / (WM * 32), TN = RT / WN;
    constexpr int PH = G::PH, PW = G::PW, PU1 = G::PU1, PCH = G::PCH, PUP = G::PUP;
    // stride 2: a patch row is staged de-interleaved -- its even columns first, then its odd ones -- so that the 32 pixel lanes
    // of a fragment (image columns 2 lx + b) read CONSECUTIVE units of one parity plane: with the row as it lies in HBM the lanes
    // are 32 bytes apart and every ds_read_b128 is a two-way bank conflict
    constexpr int PWE = (PW + 1) / 2;
    constexpr int WU1 = G::WU1, NI = G::NI, NW = G::NW, NQ = G::NQ, NIW = G::NIW;
    constexpr int WUNITS = G::WUNITS, PUNITS = G::PUNITS, SCR = G::SCR;
    constexpr int NMF = PR::N * TM * TN;             // MFMAs per k-step
    constexpr int NRD = NP * (TM + TN);              // fragment reads per k-step
    static_assert(TM >= 1 && TN >= 1 && KS >= 3, "tile / pipeline shape");
    extern __shared__ __attribute__((aligned(16))) u32x4 sp_smem[];
    u32x4* const Wl = sp_smem;                         // [2 buffers][piece][2 ch-blocks][KS][BM]
    u32x4* const Pl = sp_smem + 2 * WUNITS;            // [2 buffers][piece][PUP: 2 ch-blocks x PH x PW, padded to whole chunks]
    u32x4* const Bl = sp_smem + SCR + 64;              // the epilogue's bias staging (apart from every DMA target: a persistent
                                                       // block's epilogue runs with the next tile's first requests in flight)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int kg = lane >> 5, li = lane & 31;
    const int ntr = (a.R + BM - 1) / BM;
    const int tiles_x = a.W / TW, tiles_y = a.H / ROWS;
    const int lx = li % TW, ly = li / TW;
    const int HW = a.H * a.W, HWin = a.Hin * a.Win;
    const int nslabs = a.CH / 16;
    const int s_begin = blockIdx.y * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);
    // PERSISTENT tiles: a block walks tiles v = blockIdx.x, + gridDim.x, ... < a.ntiles (grid = ntiles: one tile each, the
    // split-K plans).  The K loop's pipeline flows across the tile boundary: during a tile's last slab the "next slab" it stages
    // is the first slab of the block's next tile, so that tile starts on operands already in LDS -- the prologue (address
    // set-up, a DMA round trip, a barrier, the first fragment reads: the larger part of ~7 us per 35-70 us tile) is paid once
    // per block instead of once per tile
    struct Tile {
        int r0, n, y0, x0;
    };
    auto decode = [&](int v) {
        // CLS = 1: the classes' tiles cost 9 : 6 : 6 : 4.  Dispatch order = groups of a.cls_group pixel tiles; inside a group the
        // filter tiles slowest, i.e. class by class, heaviest first, so that the CUs that finish a 9-tap tile late pick up the
        // 4-tap ones -- and the blocks that read one input patch (one pixel tile, every filter tile) stay within a couple of
        // rounds of each other and, the group size being a multiple of 8, on ONE XCD's L2 (blocks go round-robin over the XCDs;
        // no remap).  (Filter tiles slowest over the WHOLE launch re-read the input once per filter tile from HBM: 515 MB per
        // launch on the decoder layers against 117 for the plain kernel, profiles/r06_pmc_traffic_3stream_bf16x3.json.)
        int L = CLS == 1 ? v : sp_xcd_remap(v, a.ntiles);
        Tile t;
        if (CLS == 1) {
            const int gs = a.cls_group, span = ntr * gs;
            const int g = L / span, rem = L - g * span;
            t.r0 = (rem / gs) * BM;
            L = g * gs + rem % gs;
        } else {
            t.r0 = (L % ntr) * BM;
            L /= ntr;
        }
        const int tx = L % tiles_x;
        L /= tiles_x;
        const int ty = L % tiles_y;
        t.n = L / tiles_y;
        t.y0 = ty * ROWS;
        t.x0 = tx * TW;
        return t;
    };
    int v = blockIdx.x;
    Tile tc = decode(v);
    // CLS = 1: first filter row / column of this block's class (one tile per block in the class forms)
    const int cls1 = CLS == 1 ? tc.r0 / a.cls_k : 0;
    const int fa0 = CLS == 1 ? (cls1 >> 1) : 0, b0 = CLS == 1 ? (cls1 & 1) : 0;

    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // ---- weight DMA: instruction i of this wave copies chunk j = wave + i * NW of the NP * NI chunks of a filter row ----
    // address = (filter row base + wave-uniform chunk offset: scalar registers) + lane * 16
    const unsigned lane16 = lane * 16;
    unsigned wu[NIW];            // chunk's byte offset from the filter row's base (wave-uniform)
    int wl[NIW];                 // LDS unit offset inside a weight buffer
    bool wreal[NIW];
    int wcol[NIW];               // the chunk's filter column (class forms: structurally zero columns are not fetched)
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int j = wave + i * NW;
        wreal[i] = j < NP * NI;
        const int jj = wreal[i] ? j : 0;
        const int p = jj / NI, w = jj - p * NI;
        const int ci = w / (BM / 64), h = w - ci * (BM / 64);
        const int cb = ci / KS, b = ci - cb * KS;
        wu[i] = (unsigned)(((long)p * a.wq_pstride + ((long)cb * T + b) * a.Rpad + h * 64) * 16);
        wl[i] = p * WU1 + w * 64;
        wcol[i] = b;
        if (CLS == 1 && b < b0) wreal[i] = false;          // forward class with q = 1: column 0 is all zeros
    }
    const long wrow = (long)KS * a.Rpad * 16;             // bytes between filter rows of a slab
    const long wslab = (long)2 * T * a.Rpad * 16;         // ... between slabs
    // filter row ``fa`` of slab ``s`` of the filter tile at ``r0`` -> weight buffer at unit offset ``tog``
    auto dma_w = [&](int r0, int s, int fa, int tog, int nc = KS) {
        const char* const src = (const char*)(a.wq + r0) + (long)s * wslab + fa * wrow;
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            if (CLS != 0) {
                // a chunk of a structurally zero column (or a surplus instruction) reads the zero unit into the scratch chunk
                const bool real = wreal[i] && (CLS != 2 || wcol[i] < nc);
                __builtin_amdgcn_global_load_lds(real ? (gptr_t)(src + wu[i] + lane16) : (gptr_t)a.zeros,
                                                 (lptr_t)(sp_smem + (real ? wl[i] + tog : SCR)), 16, 0, 0);
            } else if (ABL & 16)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + wu[i] + lane16), (lptr_t)(sp_smem + (wreal[i] ? wl[i] + tog : SCR)), 4, 0, 0);
            else if (ABL & 8)
                __builtin_amdgcn_global_load_lds((gptr_t)a.zeros, (lptr_t)(sp_smem + (wreal[i] ? wl[i] + tog : SCR)), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((gptr_t)(src + wu[i] + lane16), (lptr_t)(sp_smem + (wreal[i] ? wl[i] + tog : SCR)), 16, 0, 0);
    };

    // ---- patch DMA: instruction q of this wave and piece copies chunk c = wave + q * NW of the PCH chunks ----
    const char* pp[NQ];          // this lane's unit of the slab staged last, piece 0 (the zero unit for padding lanes)
    unsigned pok = 0;            // bit q: the lane's unit of chunk q is inside the image (else it stays on the zero unit)
    int pcb[NQ], pyx[NQ];        // the unit's place in the patch: channel block, (row << 16 | column) -- the same for every tile
    int pl[NQ];
    bool preal[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = wave + q * NW;
        preal[q] = c < PCH;
        const int e = c * 64 + lane;
        const int cb = e / (PH * PW), rem = e - cb * (PH * PW);
        const int py = rem / PW, pxs = rem - py * PW;
        const int px = ST == 2 ? (pxs < PWE ? 2 * pxs : 2 * (pxs - PWE) + 1) : pxs;      // LDS slot -> patch column
        pcb[q] = e < PU1 ? cb : -1;
        pyx[q] = (py << 16) | px;
        pl[q] = c * 64;
    }
    // this lane's patch pointers of tile t, slab s_begin
    auto patch_of = [&](const Tile& t) {
        const u32x4* const ibase = a.in_q + (long)t.n * a.in_q_nstride + (long)s_begin * 2 * HWin;
        pok = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int y = t.y0 * ST + (pyx[q] >> 16) - a.pad, x = t.x0 * ST + (pyx[q] & 0xffff) - a.pad;
            const bool ok = pcb[q] >= 0 && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
            pp[q] = ok ? (const char*)(ibase + (long)pcb[q] * HWin + y * a.Win + x) : (const char*)a.zeros;
            pok |= ok ? 1u << q : 0u;
        }
    };
    patch_of(tc);
    const long pstep = a.in_q_pstride * 16;       // bytes between the pieces of a q tensor
    const long sstep = (long)2 * HWin * 16;       // ... between slabs
    auto dma_p = [&](int tog, bool advance) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const bool ok = (pok >> q) & 1u;
            pp[q] += (advance && ok) ? sstep : 0;
            const char* g = pp[q];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (ABL & 16)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sp_smem + (preal[q] ? 2 * WUNITS + tog + p * PUP + pl[q] : SCR)), 4, 0, 0);
                else if (ABL & 8)
                    __builtin_amdgcn_global_load_lds((gptr_t)a.zeros, (lptr_t)(sp_smem + (preal[q] ? 2 * WUNITS + tog + p * PUP + pl[q] : SCR)), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sp_smem + (preal[q] ? 2 * WUNITS + tog + p * PUP + pl[q] : SCR)), 16, 0, 0);
                g += ok ? pstep : 0;
            }
        }
    };

    f32x16 acc[TM][TN], accc[TM][TN];

    struct Frag {
        u32x4 a[NP][TM], b[NP][TN];
    };
    // (class forms: the block's first filter row / column folded into the fragment bases; rd() takes indices relative to them)
    const int wlane = kg * KS * BM + wm * (BM / WM) + li + b0 * BM;
    const int plane = kg * PH * PW + ((wn * TN * RPF + ly) * ST) * PW + (ST == 2 ? lx : lx * ST) + fa0 * PW + b0;
    // fragments of (filter row fa, column b): weights from the buffer at Wb, patch from the buffer at Pb; in the order of
    // their first use (the products run small terms first -- NP = 3: a2 b0, a1 b1, a0 b2, a1 b0, a0 b1, a0 b0)
    auto rd = [&](Frag& f, const u32x4* Wb, const u32x4* Pb, int fa, int b) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[NP - 1 - p][i] = Wb[(NP - 1 - p) * WU1 + b * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                f.b[p][j] = Pb[p * PUP + fa * PW + j * RPF * ST * PW + (ST == 2 ? (b & 1) * PWE + (b >> 1) : b)];
        }
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
        for (int pr = 0; pr < PR::N; ++pr) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (pr < PR::N - 1)
                        accc[i][j] = sp_mfma(f.a[PR::A[pr]][i], f.b[PR::B[pr]][j], accc[i][j]);
                    else
                        acc[i][j] = sp_mfma(f.a[0][i], f.b[0][j], acc[i][j]);
                }
        }
    };

    // ---- prologue (once per block): filter row 0 and the patch of the first slab, then filter row 1; the first fragments ----
    const int s_last = s_end - 1;
    // (CLS = 2: filter columns a slab of class (p, q) has = 3 - q)
    auto ncols_of = [&](int s) { return CLS == 2 ? 3 - ((s / (a.cls_k / 16)) & 1) : KS; };
    if (s_begin < s_end) {
        dma_w(tc.r0, s_begin, fa0, 0, ncols_of(s_begin));
        dma_p(0, false);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (s_begin < s_end) dma_w(tc.r0, s_begin, fa0 + 1, WUNITS, ncols_of(s_begin));
    int w0 = 0, p0 = 0;          // unit offsets of the weight buffer of the slab's filter row 0 / of the slab's patch buffer
    Frag cur;
    rd(cur, Wl + wlane, Pl + plane, 0, 0);

    for (;;) {                   // tiles of this block
        const int vn = v + gridDim.x;
        // (the eight-wave shape -- 256 registers per wave -- has no room for the tile state: one tile per block; so have the
        // class forms, whose next tile may be of another class)
        const bool more = NW == 4 && CLS == 0 && vn < a.ntiles && s_begin < s_end;
        const Tile tn = decode(more ? vn : v);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = accc[i][j][e] = 0.f;

        // slabs [s_lo, s_hi) with NR filter rows x NC filter columns each (the full kernel: KS x KS); rows / columns relative to
        // (fa0, b0).  Whatever (NR, NC) the NEXT slab has, its first two filter rows are relative rows 0 and 1 (NR >= 2)
        auto run = [&](auto NR_, auto NC_, int s_lo, int s_hi) {
            constexpr int NR = decltype(NR_)::value, NC = decltype(NC_)::value;
            static_assert(NR >= 2 && NC >= 2, "at least two filter rows / columns per slab");
            for (int s = s_lo; s < s_hi; ++s) {
                // what this slab stages ahead: the next slab of the tile -- or, in the tile's last slab, the first slab of the
                // block's next tile (at the very end: itself again, harmless)
                const bool last = s == s_last;
                const int sn = last ? (more ? s_begin : s) : s + 1;
                const int rn = (last && more) ? tn.r0 : tc.r0;
                const u32x4* const Pc = Pl + p0 + plane;
                const u32x4* const Pn = Pl + (PUNITS - p0) + plane;
#pragma unroll
                for (int fa = 0; fa < NR; ++fa) {
                    const int wc = (fa & 1) ? WUNITS - w0 : w0;         // this filter row's weight buffer
                    const u32x4* const Wc = Wl + wc + wlane;
                    const u32x4* const Wn = Wl + (WUNITS - wc) + wlane;
#pragma unroll
                    for (int b = 0; b < NC; ++b) {
                        Frag nx;
                        int ndma = 0;
                        if (ABL & 4) {
                            nx = cur;
                        } else if (b + 1 < NC) {
                            rd(nx, Wc, Pc, fa, b + 1);
                        } else {                                         // the next iteration's first column
                            rd(nx, Wn, fa + 1 < NR ? Pc : Pn, fa + 1 < NR ? fa + 1 : 0, 0);
                        }
                        if (b + 1 == NC && !(ABL & 1)) {
                            // the filter row two iterations ahead replaces this one's (all of its fragments were read before the barrier)
                            if (fa + 2 < NR)
                                dma_w(tc.r0, s, fa0 + fa + 2, wc, NC);
                            else
                                dma_w(rn, sn, fa0 + fa + 2 - NR, wc, ncols_of(sn));
                            ndma = NIW;
                        }
                        if (fa == 0 && b == 0 && !(ABL & 1)) {           // the next slab's patch
                            if (last && more)
                                patch_of(tn);
                            dma_p(PUNITS - p0, !last);
                            ndma = NP * NQ;
                        }
                        mm(cur);
                        // fragment reads, then DMA, one behind each MFMA, front-loaded: the next k-step starts on fragments read long ago
#pragma unroll
                        for (int m_ = 0; m_ < NRD; ++m_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
#pragma unroll
                        for (int m_ = 0; m_ < ndma; ++m_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                        }
                        if (b == NC - 2 && !(ABL & 2)) {
                            // everything but the patch requested in this iteration has landed; all reads of this filter row are done
                            if (fa == 0)
                                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NP * NQ) : "memory");
                            else
                                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        cur = nx;
                    }
                }
                if (NR & 1) w0 = WUNITS - w0;
                p0 = PUNITS - p0;
            }
        };
        if constexpr (CLS == 0) {
            run(SpIC<KS>{}, SpIC<KS>{}, s_begin, s_end);
        } else if constexpr (CLS == 1) {
            switch (cls1) {
                case 0: run(SpIC<3>{}, SpIC<3>{}, s_begin, s_end); break;
                case 1: run(SpIC<3>{}, SpIC<2>{}, s_begin, s_end); break;
                case 2: run(SpIC<2>{}, SpIC<3>{}, s_begin, s_end); break;
                default: run(SpIC<2>{}, SpIC<2>{}, s_begin, s_end); break;
            }
        } else {
            const int spc = a.cls_k / 16;                  // slabs per class
            run(SpIC<3>{}, SpIC<3>{}, max(s_begin, 0), min(s_end, spc));
            run(SpIC<3>{}, SpIC<2>{}, max(s_begin, spc), min(s_end, 2 * spc));
            run(SpIC<2>{}, SpIC<3>{}, max(s_begin, 2 * spc), min(s_end, 3 * spc));
            run(SpIC<2>{}, SpIC<2>{}, max(s_begin, 3 * spc), min(s_end, 4 * spc));
        }

        sp_conv_epilogue<BM, RT, WM, WN, POOL, TW, TM, TN, NP>(a, acc, accc, Bl, tid, wm, wn, kg, li, lx, ly, tc.n, tc.r0, tc.y0, tc.x0, HW);
        if (!more) break;
        v = vn;
        tc = tn;
    }
    // DMA requested past the end of the last tile may still be in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// Weight gradient (3x3 stride 1 / 2, 5x5 stride 1; 'same') from split q tensors: dwp[(c, tap)][k] = sum over pixels of
// x[c, pixel * ST + tap - pad] * dy[k, pixel] -- the structure of lp_wgrad_q_kernel (conv_lp.hip): the contraction runs over
// pixels while a q unit holds 8 channels of one pixel, so both MFMA operands are read from LDS with the transposing read
// ds_read_b64_tr_b16; x rows live in a ring (each output row brings ST new rows), the dy strip is double-buffered, both
// staged global -> LDS by DMA as they lie in HBM, three pieces each.  Block = CHT x CT x KS waves: a wave owns the KS taps
// of ONE filter row of one (32 channels, 32 filters) tile; per 16-pixel k-step it reads 3 dy fragments and 3 KS x fragments
// and runs 6 KS MFMAs.
// ------------------------------------------------------------------------------------------------
struct SpWgradArgs {
    const u32x4* xq;
    long xq_ns, xq_ps;     // units between samples / pieces of x  [piece][N][C/8][H][W]
    const u32x4* dyq;
    long dyq_ns, dyq_ps;
    const u32x4* zeros;
    float* out;
    int N, C, H, W, K, Ho, Wo;
    int rows_per_split, splits_per_col;
    long split_stride;
    int accumulate;
    int cls_k;             // CLS form: filters per parity class (the collapsed bilinear convolution, sp_conv2_kernel)
    int xcd;               // 1: the (channel tile, filter tile) blocks of one strip run on ONE XCD (they share its x and dy rows)
    // sp_wgrad_pooled_kernel: dy in the sparse-instruction operand form (ghm_maxpool2_mask_bwd_compress_q)
    const u32x4* cq;       // [piece][N][K/8][Ho][Wo/2] half-width rows
    long cq_ns, cq_ps;
    const u32x4* cidx;     // [N][K/8][Ho][Wo/32] units of 8 x u16 column bits
    const int* cflags;     // [N * Ho]: the row has a tied window row -> take it from the dense dyq
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// TUNING ONLY (results wrong): -DGHM_WGRAD_ABLATE=<bits> compiles parts of sp_wgrad_kernel's row loop out: 1 = no DMA inside the
// loop, 2 = no fragment reads inside the loop (the first row's fragments are reused), 4 = no waits / barriers inside the loop,
// 16 = every DMA instruction of the loop copies the (cache-resident) zero unit: the instructions and LDS writes without the memory traffic
#ifndef GHM_WGRAD_ABLATE
#define GHM_WGRAD_ABLATE 0
#endif

// The transposing LDS read ds_read_b64_tr_b16 as inline assembly (round 5).  The compiler orders an LDS read it cannot
// disambiguate behind every LDS-DMA in flight: with __builtin_amdgcn_ds_read_tr16_b64_v4i16 it put ``s_waitcnt vmcnt(0)`` in front of the first fragment read of every output row
// -- directly behind the DMA requests of the next row, whose whole latency was exposed once per row.  As assembly the read is
// invisible to that pass; the kernel waits for its own fragments (sp_tr_wait: ``s_waitcnt lgkmcnt(0)`` tied to the registers).
typedef unsigned long long sp_u64;
struct SpTrFrag {
    sp_u64 lo, hi;
};
__device__ __forceinline__ void sp_tr_issue(SpTrFrag& f, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:256" : "=&v"(f.lo), "=&v"(f.hi) : "v"(lds_addr));
}
__device__ __forceinline__ u32x4 sp_tr_bits(const SpTrFrag& f) {
    return u32x4{(unsigned)f.lo, (unsigned)(f.lo >> 32), (unsigned)f.hi, (unsigned)(f.hi >> 32)};
}
template <int N>
__device__ __forceinline__ void sp_tr_wait(SpTrFrag (&f)[N]) {
    static_assert(N == 2 || N == 3 || N == 5, "fragment sets of 2, 3 or 5");
    if constexpr (N == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi));
    else if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi),
                     "+v"(f[3].lo), "+v"(f[3].hi), "+v"(f[4].lo), "+v"(f[4].hi));
}

// The KS tap fragments of an x row are windows of ONE 12-pixel span shifted by a pixel each (stride 1: a lane's 8 pixels start
// at tap t): the span is read once (three transposing reads instead of 2 KS) and the windows are cut from its registers -- even
// taps are register pairs as they lie, odd taps one v_alignbit per register.  Same operand bits, same MFMA order.
#ifndef GHM_WGRAD_SPAN
#define GHM_WGRAD_SPAN 1
#endif
struct SpTrSpan {
    sp_u64 s0, s1, s2;
};
__device__ __forceinline__ void sp_span_issue(SpTrSpan& f, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %3\n\tds_read_b64_tr_b16 %1, %3 offset:256\n\tds_read_b64_tr_b16 %2, %3 offset:512"
                 : "=&v"(f.s0), "=&v"(f.s1), "=&v"(f.s2)
                 : "v"(lds_addr));
}
__device__ __forceinline__ void sp_span_wait(SpTrSpan& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.s0), "+v"(f.s1), "+v"(f.s2)); }
template <int T>
__device__ __forceinline__ u32x4 sp_span_tap(const SpTrSpan& f) {
    static_assert(T >= 0 && T <= 4, "taps 0 .. 4 of a 12-pixel span");
    const unsigned v[6] = {(unsigned)f.s0, (unsigned)(f.s0 >> 32), (unsigned)f.s1, (unsigned)(f.s1 >> 32), (unsigned)f.s2, (unsigned)(f.s2 >> 32)};
    if constexpr (T % 2 == 0) {
        return u32x4{v[T / 2], v[T / 2 + 1], v[T / 2 + 2], v[T / 2 + 3]};
    } else {
        constexpr int b = (T - 1) / 2;
        return u32x4{__builtin_amdgcn_alignbit(v[b + 1], v[b], 16), __builtin_amdgcn_alignbit(v[b + 2], v[b + 1], 16),
                     __builtin_amdgcn_alignbit(v[b + 3], v[b + 2], 16), __builtin_amdgcn_alignbit(v[b + 4], v[b + 3], 16)};
    }
}

// CLS = 1 (3x3 stride 1): the filters are 4 parity classes of a.cls_k (class (p, q) of the collapsed bilinear convolution has no
// taps in filter row 0 when p = 1, none in filter column 0 when q = 1): a block's filter tile lies in ONE class; the waves of a
// structurally-zero filter row only help staging, and every wave skips the zero column -- fragment reads and MFMAs.  The
// skipped taps leave their zero accumulators in dwp (the expansion multiplies them by zero coefficients anyway).
// LA = 2 (round 6; 3x3 stride 1, where it fits the LDS): the operands of output row i + 2 are requested while row i is
// multiplied -- an output row of a 32-pixel strip is ~1.5 us of MFMAs (1 us in the class forms), the order of the DMA latency it
// had to hide with one row of lookahead.  Ring of KS + 2 ST rows, three dy buffers; every wave issues the SAME number of DMA
// instructions per row (surplus ones copy the zero unit into a scratch chunk), so the end-of-row wait is a compile-time
// ``vmcnt(instructions of this row)``: row i + 1 has landed, row i + 2 stays in flight.  Same products, same order.
template <int KS, int ST, int CHT, int CT, int SPX, int NP, int CLS = 0, int LA = 1>
__global__ __launch_bounds__(CHT * CT * KS * 64, 1) void sp_wgrad_kernel(const SpWgradArgs a) {
    static_assert((ST == 1 || (ST == 2 && KS == 3)) && (SPX == 64 || SPX == 32 || SPX == 16) && (KS == 3 || KS == 5), "variants");
    static_assert(CLS == 0 || (KS == 3 && ST == 1), "class form: 3x3 stride 1");
    static_assert(LA == 1 || LA == 2, "rows of lookahead");
    constexpr int T = KS * KS, PADK = KS / 2;
    constexpr int NWAVES = CHT * CT * KS;
    constexpr int NPAR = ST;                          // column-parity planes of an x row
    constexpr int XPIX = ST == 1 ? SPX + KS - 1 : SPX + 1;   // pixels per plane
    constexpr int XCH = (XPIX + 15) / 16;             // 16-pixel DMA pieces per plane
    constexpr int PLB = XCH * 16 * 64;                // bytes per plane
    constexpr int ROWB = CHT * NPAR * PLB;            // bytes per ring row and piece
    constexpr int NR = KS + LA * ST;                  // ring rows: KS live + LA x ST arriving
    constexpr int NYB = LA + 1;                       // dy buffers
    constexpr int YTB = SPX * 64;                     // bytes per dy filter tile
    constexpr int YB = CT * YTB;                      // bytes per dy buffer and piece
    constexpr int KSTEPS = SPX / 16;
    constexpr int XTOT = NP * CHT * NPAR * XCH, YTOT = NP * CT * (SPX / 16);      // DMA instructions per x row / dy strip
    constexpr int NX = (XTOT + NWAVES - 1) / NWAVES, NY = (YTOT + NWAVES - 1) / NWAVES;   // ... per wave
    extern __shared__ __attribute__((aligned(16))) char sp_wsmem[];
    char* const Xl = sp_wsmem;                        // [ring slot][piece][ROWB]
    char* const Yl = sp_wsmem + NR * NP * ROWB;       // [buffer][piece][YB]
    char* const Sl = Yl + NYB * NP * YB;              // LA = 2: 1 KB scratch, the target of the surplus DMA instructions
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / (CHT * CT);                 // this wave's filter row
    const int wrem = wave % (CHT * CT);
    const int hh = wrem / CT, ww = wrem % CT;         // this wave's channel group / filter tile
    const int kg = lane >> 5, li = lane & 31;
    // Blocks are handed to the eight XCDs round-robin in launch order: as launched, the blocks that share a strip's x rows
    // (same channel tile) and dy rows (same filter tile) sit on different XCDs and every L2 fetches them again.  Re-numbered
    // so that an XCD walks whole strips: all (channel tile, filter tile) blocks of a strip behind one L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd) {
        const int gxy = gridDim.x * gridDim.y;
        int L = sp_xcd_remap(bx + gridDim.x * by + gxy * bz, gxy * gridDim.z);
        bz = L / gxy;
        L -= bz * gxy;
        by = L / gridDim.x;
        bx = L - by * gridDim.x;
    }
    const int c0 = bx * (32 * CHT), k0 = by * (32 * CT);
    const int strips = a.Wo / SPX;
    const int col = bz / a.splits_per_col, sp = bz - col * a.splits_per_col;
    const int n = col / strips, j0 = (col - n * strips) * SPX;
    const int i_begin = sp * a.rows_per_split, i_end = min(a.Ho, i_begin + a.rows_per_split);
    const int HWx = a.H * a.W, HWy = a.Ho * a.Wo;
    const int xs0 = j0 * ST - PADK;                   // image column of local column 0

    // DMA lane roles inside a 16-pixel piece: pixel pxi, channel block cb4 of the 32-channel group
    const int pxi = lane >> 2, cb4 = lane & 3;
    const u32x4* const xbase = a.xq + (long)n * a.xq_ns + (long)(c0 / 8 + cb4) * HWx;
    const u32x4* const ybase = a.dyq + (long)n * a.dyq_ns + (long)(k0 / 8 + cb4) * HWy;

    auto stage_xrow = [&](int y) {
        const int slot = (y + NR) % NR;
        const bool rok = (unsigned)y < (unsigned)a.H;
#pragma unroll
        for (int p0 = 0; p0 < NP * CHT * NPAR * XCH; p0 += NWAVES) {
            const int pq = p0 + wave;
            if (pq < NP * CHT * NPAR * XCH) {
                const int piece = pq / (CHT * NPAR * XCH), p = pq - piece * (CHT * NPAR * XCH);
                const int pl = p / XCH, ch = p - pl * XCH;        // plane = (channel group, parity)
                const int g = pl / NPAR, par = pl - g * NPAR;
                const int pp = ch * 16 + pxi;                        // pixel inside the plane
                const int x = xs0 + (ST == 2 ? 2 * pp + par : pp);
                const bool ok = rok && (unsigned)x < (unsigned)a.W;
                const u32x4* src = (ok && !(GHM_WGRAD_ABLATE & 16)) ? xbase + piece * a.xq_ps + (long)(g * 4) * HWx + (long)y * a.W + x : a.zeros;
                if (pp < XPIX)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xl + (slot * NP + piece) * ROWB + pl * PLB + ch * 1024), 16, 0, 0);
            } else if (LA == 2) {
                __builtin_amdgcn_global_load_lds((gptr_t)a.zeros, (lptr_t)Sl, 16, 0, 0);
            }
        }
    };
    auto stage_dy = [&](int i, int buf) {
#pragma unroll
        for (int p0 = 0; p0 < NP * CT * (SPX / 16); p0 += NWAVES) {
            const int pq = p0 + wave;
            if (pq < NP * CT * (SPX / 16)) {
                const int piece = pq / (CT * (SPX / 16)), p = pq - piece * (CT * (SPX / 16));
                const int ct = p / (SPX / 16), ch = p - ct * (SPX / 16);
                const u32x4* src = (GHM_WGRAD_ABLATE & 16) ? a.zeros : ybase + piece * a.dyq_ps + (long)(ct * 4) * HWy + (long)i * a.Wo + j0 + ch * 16 + pxi;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Yl + (buf * NP + piece) * YB + ct * YTB + ch * 1024), 16, 0, 0);
            } else if (LA == 2) {
                __builtin_amdgcn_global_load_lds((gptr_t)a.zeros, (lptr_t)Sl, 16, 0, 0);
            }
        }
    };

    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (i_begin < i_end) {
#pragma unroll
        for (int fa = 0; fa < KS; ++fa) stage_xrow(i_begin * ST + fa - PADK);
        stage_dy(i_begin, 0);
        if (LA == 2 && i_begin + 1 < i_end) {
#pragma unroll
            for (int r = 0; r < ST; ++r) stage_xrow(i_begin * ST - PADK + KS + r);
            stage_dy(i_begin + 1, 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing (transposing reads): 16-lane group g16 -> rows (channels / filters) 16 * (g16 & 1) .. + 15;
    // lane t of the group supplies the address of pixel key = t >> 2, channel quad t & 3
    const int g16 = lane >> 4, lt = lane & 15;
    const int key = lt >> 2, quad = lt & 3;
    const int lane_off = (8 * kg + key) * 64 + (g16 & 1) * 32 + quad * 8;      // + 4 pixels (256 B) for the upper half
    const char* const ylane = Yl + ww * YTB + lane_off;
    const char* const xlane = Xl + hh * NPAR * PLB + lane_off;

    {
        const unsigned lds0 = (unsigned)(size_t)(lptr_t)sp_wsmem;
        const unsigned xl0 = lds0 + hh * NPAR * PLB + lane_off;
        const unsigned yl0 = lds0 + NR * NP * ROWB + ww * YTB + lane_off;
        // T0: first active filter column of this block's class; ACT: does this wave's filter row have any taps?
        auto rows = [&](auto T0_, auto ACT_) {
            constexpr int T0 = decltype(T0_)::value;
            constexpr bool ACT = decltype(ACT_)::value != 0;
            // (LA = 2: the operands of row i + 1 landed a whole row ago, so its FIRST fragments are read under row i's last MFMAs --
            // with one row of lookahead every row starts with an exposed LDS round trip)
            constexpr bool SPAN = GHM_WGRAD_SPAN && ST == 1;
            SpTrFrag af[2][SPAN ? 1 : KS], bf[2][NP];
            SpTrSpan as[2];
            auto read_x = [&](unsigned xr, int ks, int p, int slot) {
                if constexpr (SPAN) {
                    sp_span_issue(as[slot], xr + p * ROWB + (ks * 16) * 64);
                } else {
#pragma unroll
                    for (int fb = T0; fb < KS; ++fb) {
                        const int par = ST == 2 ? (fb & 1) : 0;
                        const int shift = ST == 2 ? (fb >> 1) : fb;
                        sp_tr_issue(af[slot][fb], xr + p * ROWB + par * PLB + (ks * 16 + shift) * 64);
                    }
                }
            };
            auto x_tap = [&](int slot, auto t_) -> u32x4 {
                constexpr int t = decltype(t_)::value;
                if constexpr (SPAN)
                    return sp_span_tap<t>(as[slot]);
                else
                    return sp_tr_bits(af[slot][t]);
            };
            auto read_dy = [&](unsigned yb, int ks, int slot) {
#pragma unroll
                for (int q = 0; q < NP; ++q) sp_tr_issue(bf[slot][q], yb + q * YB + ks * 1024);
            };
            auto xr_of = [&](int i) { return xl0 + (((i * ST + wr - PADK + NR) % NR) * NP) * ROWB; };
            auto yb_of = [&](int i) { return yl0 + ((i - i_begin) % NYB) * NP * YB; };
            constexpr bool CARRY = LA == 2 && ((KSTEPS * NP) % 2 == 0) && (KSTEPS % 2 == 0);     // slot parities line up across rows
            for (int i = i_begin; i < i_end; ++i) {
                const bool ahead = i + LA < i_end;
                if (ahead && !(GHM_WGRAD_ABLATE & 1)) {
#pragma unroll
                    for (int r = 0; r < ST; ++r) stage_xrow((i + LA - 1) * ST - PADK + KS + r);     // the rows output row i + LA adds
                    stage_dy(i + LA, (i + LA - i_begin) % NYB);
                }
                if constexpr (ACT) {
                    const unsigned xr = xr_of(i), yb = yb_of(i);
                    if ((!CARRY && !(GHM_WGRAD_ABLATE & 2)) || i == i_begin) {
                        read_dy(yb, 0, 0);
                        read_x(xr, 0, NP - 1, 0);
                        if (GHM_WGRAD_ABLATE & 2) {
                            read_dy(yb, 0, 1);
                            read_x(xr, 0, NP - 1, 1);
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                        for (int pi = 0; pi < NP; ++pi) {
                            const int p = NP - 1 - pi, ph = ks * NP + pi;
                            // this phase's fragments have arrived (requested one phase ago) ...
                            if constexpr (SPAN)
                                sp_span_wait(as[ph & 1]);
                            else if constexpr (T0 == 0)
                                sp_tr_wait(af[ph & 1]);
                            else
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[ph & 1][1].lo), "+v"(af[ph & 1][1].hi), "+v"(af[ph & 1][2].lo),
                                             "+v"(af[ph & 1][2].hi));
                            if (pi == 0) sp_tr_wait(bf[ks & 1]);
                            // ... the next phase's are requested behind them
                            if (GHM_WGRAD_ABLATE & 2) {
                            } else if (pi + 1 < NP) {
                                read_x(xr, ks, p - 1, (ph + 1) & 1);
                            } else if (ks + 1 < KSTEPS) {
                                read_dy(yb, ks + 1, (ks + 1) & 1);
                                read_x(xr, ks + 1, NP - 1, (ph + 1) & 1);
                            } else if (CARRY && i + 1 < i_end) {        // the next row's first fragments
                                read_dy(yb_of(i + 1), 0, 0);
                                read_x(xr_of(i + 1), 0, NP - 1, 0);
                            }
                            {
                                u32x4 xt[KS];
                                if constexpr (T0 == 0) xt[0] = x_tap(ph & 1, SpIC<0>{});
                                xt[1] = x_tap(ph & 1, SpIC<1>{});
                                xt[2] = x_tap(ph & 1, SpIC<2>{});
                                if constexpr (KS == 5) {
                                    xt[3] = x_tap(ph & 1, SpIC<3>{});
                                    xt[4] = x_tap(ph & 1, SpIC<4>{});
                                }
#pragma unroll
                                for (int q = 0; q <= NP - 1 - p; ++q)
#pragma unroll
                                    for (int t = T0; t < KS; ++t) acc[t] = sp_mfma(xt[t], sp_tr_bits(bf[ks & 1][q]), acc[t]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (CARRY && ph == KSTEPS * NP - 2) {
                                // the row's barrier, one phase early: every LDS read of row i has been issued (the last phase's
                                // fragments are in registers behind lgkmcnt(0)) and row i + 1 has landed for every wave -- behind it
                                // the last phase's MFMAs run while row i + 1's first fragments are read
                                if (ahead)
                                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ST * NX + NY) : "memory");
                                else
                                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            }
                        }
                    }
                }
                // row i + 1 has landed (LA = 2: what this row requested, row i + 2, may stay in flight); everybody is done with row i
                if (CARRY) {
                    if constexpr (!ACT) {           // (a wave that only stages meets the same barrier)
                        if (ahead)
                            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(ST * NX + NY) : "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                } else if (GHM_WGRAD_ABLATE & 4) {
                } else if (LA == 2 && ahead) {
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(ST * NX + NY) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
        };
        if constexpr (CLS == 0) {
            rows(SpIC<0>{}, SpIC<1>{});
        } else {
            const int cls = k0 / a.cls_k;                  // (p, q) of this block's filter tile
            if (wr < (cls >> 1))
                rows(SpIC<0>{}, SpIC<0>{});
            else if (cls & 1)
                rows(SpIC<1>{}, SpIC<1>{});
            else
                rows(SpIC<0>{}, SpIC<1>{});
        }
    }

    // ---- epilogue: lane = filter k0 + 32 ww + li; rows = channels c0 + 32 hh + (e & 3) + 8 (e >> 2) + 4 kg ----
    float* const ob = a.out + (long)bz * a.split_stride;
    const int kcol = k0 + ww * 32 + li;
    if (kcol >= a.K) return;
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = c0 + hh * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
            const int tap = wr * KS + t;
            if (c < a.C) {
                float* o = ob + ((long)c * T + tap) * a.K + kcol;
                float v = acc[t][e];
                if (a.accumulate) v += *o;
                *o = v;
            }
        }
}


// ------------------------------------------------------------------------------------------------
// 5x5 weight gradient of a conv -> LeakyRectify -> MaxPool2D(2) layer (DCGAN discriminator, architectures/dcgan.py:42-60) on the
// SPARSE matrix instruction.  The gradient of the conv's output is the max-pool backward of the pooled gradient: one non-zero
// per window row at most (ties aside), i.e. two values in any four consecutive pixels of a row -- v_smfmac_f32_32x32x32_bf16's
// 2:4 operand with the PIXELS as contraction index, 32 pixels per instruction at the dense instruction's issue cost
// (tools/smfmac_probe.hip, profiles/r06_smfmac_probe.txt).  A = dy (rows = filters): a lane's 8 compressed values are half-pixels
// 4 kg .. + 3 and 8 + 4 kg .. + 3 of a 16-half-pixel chunk, two transposing reads of the half-width row cq; its index word puts
// value j of a group at pixel 2 j + (column bit).  B = x (columns = channels): a lane's 16 pixels 16 kg .. + 15 shifted by the
// tap, cut from one 20-pixel span (five transposing reads).  The three pieces of a value share the pattern: the same six
// products as the dense kernel, in its order.  A row with a tied window row (cflags) is taken from the dense q tensor with the
// dense instruction in the same operand roles, into the same accumulators.  Block = 32 channels x 64 filters, 10 waves
// (5 filter rows x 2 filter tiles), staging and split-K as sp_wgrad_kernel<5, 1, 1, 2, SPX, NP>.
// ------------------------------------------------------------------------------------------------
struct SpTrSpan5 {
    sp_u64 s0, s1, s2, s3, s4;
};
__device__ __forceinline__ void sp_span5_issue(SpTrSpan5& f, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %5\n\tds_read_b64_tr_b16 %1, %5 offset:256\n\tds_read_b64_tr_b16 %2, %5 offset:512\n\t"
                 "ds_read_b64_tr_b16 %3, %5 offset:768\n\tds_read_b64_tr_b16 %4, %5 offset:1024"
                 : "=&v"(f.s0), "=&v"(f.s1), "=&v"(f.s2), "=&v"(f.s3), "=&v"(f.s4)
                 : "v"(lds_addr));
}
__device__ __forceinline__ void sp_span5_wait(SpTrSpan5& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.s0), "+v"(f.s1), "+v"(f.s2), "+v"(f.s3), "+v"(f.s4));
}
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
template <int T>
__device__ __forceinline__ u32x8 sp_span5_tap(const SpTrSpan5& f) {
    static_assert(T >= 0 && T <= 4, "taps 0 .. 4 of a 20-pixel span");
    const unsigned v[10] = {(unsigned)f.s0, (unsigned)(f.s0 >> 32), (unsigned)f.s1, (unsigned)(f.s1 >> 32), (unsigned)f.s2,
                            (unsigned)(f.s2 >> 32), (unsigned)f.s3, (unsigned)(f.s3 >> 32), (unsigned)f.s4, (unsigned)(f.s4 >> 32)};
    u32x8 r;
    if constexpr (T % 2 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = v[T / 2 + j];
    } else {
        constexpr int b = (T - 1) / 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __builtin_amdgcn_alignbit(v[b + 1 + j], v[b + j], 16);
    }
    return r;
}
__device__ __forceinline__ f32x16 sp_smfmac(u32x4 a, u32x8 b, f32x16 c, int idx) {
    return __builtin_amdgcn_smfmac_f32_32x32x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x16, b), c, idx, 0, 0);
}

template <int SPX, int NP>
__global__ __launch_bounds__(640, 1) void sp_wgrad_pooled_kernel(const SpWgradArgs a) {
    static_assert(SPX == 64 || SPX == 32, "strips of 64 or 32 pixels");
    constexpr int KS = 5, T = 25, PADK = 2, CT = 2, NWAVES = CT * KS;
    constexpr int XPIX = SPX + KS - 1, XCH = (XPIX + 15) / 16;
    constexpr int ROWB = XCH * 16 * 64;              // bytes per ring row and piece (one 32-channel group)
    constexpr int NR = KS + 1, NYB = 2;
    constexpr int YTB = SPX * 64, YB = CT * YTB;      // dy buffer sized for a DENSE row; a compressed row uses half of every tile
    constexpr int KSTEPS = SPX / 16, K2 = SPX / 32, NCH = SPX / 32;
    constexpr int IB = 1024;                          // index words of a row: CT x 4 units x NCH chunks x 16 bytes <= 256
    extern __shared__ __attribute__((aligned(16))) char sp_wsmem[];
    char* const Xl = sp_wsmem;
    char* const Yl = sp_wsmem + NR * NP * ROWB;
    char* const Il = Yl + NYB * NP * YB;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / CT, ww = wave % CT;
    const int kg = lane >> 5, li = lane & 31;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd) {
        const int gxy = gridDim.x * gridDim.y;
        int L = sp_xcd_remap(bx + gridDim.x * by + gxy * bz, gxy * gridDim.z);
        bz = L / gxy;
        L -= bz * gxy;
        by = L / gridDim.x;
        bx = L - by * gridDim.x;
    }
    const int c0 = bx * 32, k0 = by * (32 * CT);
    const int strips = a.Wo / SPX;
    const int col = bz / a.splits_per_col, sp = bz - col * a.splits_per_col;
    const int n = col / strips, j0 = (col - n * strips) * SPX;
    const int i_begin = sp * a.rows_per_split, i_end = min(a.Ho, i_begin + a.rows_per_split);
    const int HWx = a.H * a.W, HWy = a.Ho * a.Wo, HWc = a.Ho * (a.Wo / 2), Wc = a.Wo / 2, Wi = a.Wo / 32;
    const int xs0 = j0 - PADK;

    const int pxi = lane >> 2, cb4 = lane & 3;
    const u32x4* const xbase = a.xq + (long)n * a.xq_ns + (long)(c0 / 8 + cb4) * HWx;
    const u32x4* const ybase = a.dyq + (long)n * a.dyq_ns + (long)(k0 / 8 + cb4) * HWy;
    const u32x4* const cbase = a.cq + (long)n * a.cq_ns + (long)(k0 / 8 + cb4) * HWc;
    const int* const fl = a.cflags + (long)n * a.Ho;

    auto stage_xrow = [&](int y) {
        const int slot = (y + NR) % NR;
        const bool rok = (unsigned)y < (unsigned)a.H;
#pragma unroll
        for (int p0 = 0; p0 < NP * XCH; p0 += NWAVES) {
            const int pq = p0 + wave;
            if (pq < NP * XCH) {
                const int piece = pq / XCH, ch = pq - piece * XCH;
                const int pp = ch * 16 + pxi;
                const int x = xs0 + pp;
                const bool ok = rok && (unsigned)x < (unsigned)a.W;
                const u32x4* src = ok ? xbase + piece * a.xq_ps + (long)y * a.W + x : a.zeros;
                if (pp < XPIX) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xl + (slot * NP + piece) * ROWB + ch * 1024), 16, 0, 0);
            }
        }
    };
    // dy row i into buffer buf: dense (the row has a tie) or half-width + its index words
    auto stage_dy = [&](int i, int buf, bool dense) {
        if (dense) {
#pragma unroll
            for (int p0 = 0; p0 < NP * CT * (SPX / 16); p0 += NWAVES) {
                const int pq = p0 + wave;
                if (pq < NP * CT * (SPX / 16)) {
                    const int piece = pq / (CT * (SPX / 16)), p = pq - piece * (CT * (SPX / 16));
                    const int ct = p / (SPX / 16), ch = p - ct * (SPX / 16);
                    const u32x4* src = ybase + piece * a.dyq_ps + (long)(ct * 4) * HWy + (long)i * a.Wo + j0 + ch * 16 + pxi;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Yl + (buf * NP + piece) * YB + ct * YTB + ch * 1024), 16, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int p0 = 0; p0 < NP * CT * (SPX / 32); p0 += NWAVES) {
                const int pq = p0 + wave;
                if (pq < NP * CT * (SPX / 32)) {
                    const int piece = pq / (CT * (SPX / 32)), p = pq - piece * (CT * (SPX / 32));
                    const int ct = p / (SPX / 32), ch = p - ct * (SPX / 32);
                    const u32x4* src = cbase + piece * a.cq_ps + (long)(ct * 4) * HWc + (long)i * Wc + j0 / 2 + ch * 16 + pxi;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Yl + (buf * NP + piece) * YB + ct * YTB + ch * 1024), 16, 0, 0);
                }
            }
            if (wave == NWAVES - 1 && lane < CT * 4 * NCH) {       // lane = (filter tile, 8-filter unit, chunk)
                const int ct = lane / (4 * NCH), u = (lane / NCH) & 3, ch = lane % NCH;
                const u32x4* src = a.cidx + (((long)n * (a.K / 8) + k0 / 8 + ct * 4 + u) * a.Ho + i) * Wi + j0 / 32 + ch;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Il + buf * IB), 16, 0, 0);
            }
        }
    };

    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    bool cur_dense = false;
    if (i_begin < i_end) {
        cur_dense = fl[i_begin] != 0;
#pragma unroll
        for (int fa = 0; fa < KS; ++fa) stage_xrow(i_begin + fa - PADK);
        stage_dy(i_begin, 0, cur_dense);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int g16 = lane >> 4, lt = lane & 15;
    const int key = lt >> 2, quad = lt & 3;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)sp_wsmem;
    const unsigned sub = (g16 & 1) * 32 + quad * 8;
    // dense fragments: a lane's 8 pixels 8 kg .. + 7 of a 16-pixel step; sparse: x 16 kg .. + 15 (+ tap) of a 32-pixel step,
    // compressed dy half-pixels 4 kg .. + 3 and 8 + 4 kg .. + 3 of its 16
    const unsigned xd0 = lds0 + (8 * kg + key) * 64 + sub, xs_0 = lds0 + (16 * kg + key) * 64 + sub;
    const unsigned yd0 = lds0 + NR * NP * ROWB + ww * YTB + (8 * kg + key) * 64 + sub;
    const unsigned yc0 = lds0 + NR * NP * ROWB + ww * YTB + (4 * kg + key) * 64 + sub;
    const unsigned il0 = lds0 + NR * NP * ROWB + NYB * NP * YB + ((ww * 4 + (li >> 3)) * NCH) * 16 + (li & 7) * 2;

    for (int i = i_begin; i < i_end; ++i) {
        const bool ahead = i + 1 < i_end;
        const bool next_dense = ahead && fl[i + 1] != 0;
        if (ahead && !(GHM_WGRAD_ABLATE & 1)) {
            stage_xrow(i + 1 - PADK + KS - 1);
            stage_dy(i + 1, (i + 1 - i_begin) & 1, next_dense);
        }
        const unsigned xrow = (((i + wr - PADK + NR) % NR) * NP) * ROWB;
        const unsigned ybuf = (((i - i_begin) & 1) * NP) * YB;
        if (!cur_dense) {
            // ---- sparse row: K2 steps of 32 pixels ----
            SpTrSpan5 as[2];
            SpTrFrag bf[2][NP];
            unsigned iw[2];
            auto read_dy = [&](int ks2, int slot) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512"
                                 : "=&v"(bf[slot][q].lo), "=&v"(bf[slot][q].hi)
                                 : "v"(yc0 + ybuf + q * YB + ks2 * 1024));
                // (assembly like the fragment reads: a read the compiler sees waits for every LDS-DMA in flight)
                asm volatile("ds_read_u16 %0, %1" : "=&v"(iw[slot]) : "v"(il0 + ((i - i_begin) & 1) * IB + ks2 * 16));
            };
            // index word of the instruction: value j of a group sits at pixel 2 j + (its column bit)
            auto index_of = [&](unsigned w) {
                unsigned b8 = ((w >> (4 * kg)) & 0xfu) | (((w >> (8 + 4 * kg)) & 0xfu) << 4);
                b8 = (b8 | (b8 << 4)) & 0x0f0fu;
                b8 = (b8 | (b8 << 2)) & 0x3333u;
                b8 = (b8 | (b8 << 1)) & 0x5555u;
                return (int)(0x8888u | b8);
            };
            auto read_x = [&](int ks2, int p, int slot) { sp_span5_issue(as[slot], xs_0 + xrow + p * ROWB + ks2 * 2048); };
            read_dy(0, 0);
            read_x(0, NP - 1, 0);
#pragma unroll
            for (int ks2 = 0; ks2 < K2; ++ks2) {
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    const int p = NP - 1 - pi, ph = ks2 * NP + pi;
                    sp_span5_wait(as[ph & 1]);
                    if (pi == 0) {
                        sp_tr_wait(bf[ks2 & 1]);
                        asm volatile("" : "+v"(iw[ks2 & 1]));
                    }
                    const int ixw = index_of(iw[ks2 & 1]);
                    if (pi + 1 < NP) {
                        read_x(ks2, p - 1, (ph + 1) & 1);
                    } else if (ks2 + 1 < K2) {
                        read_dy(ks2 + 1, (ks2 + 1) & 1);
                        read_x(ks2 + 1, NP - 1, (ph + 1) & 1);
                    }
                    // tap by tap (one 8-register window of the span live at a time: the kernel sits at the 168 registers of three
                    // waves per SIMD); the products of a tap go to ITS accumulator, in the dense kernel's order
                    auto tap = [&](auto t_) {
                        constexpr int t = decltype(t_)::value;
                        const u32x8 xt = sp_span5_tap<t>(as[ph & 1]);
#pragma unroll
                        for (int q = 0; q <= NP - 1 - p; ++q) acc[t] = sp_smfmac(sp_tr_bits(bf[ks2 & 1][q]), xt, acc[t], ixw);
                    };
                    tap(SpIC<0>{}); tap(SpIC<1>{}); tap(SpIC<2>{}); tap(SpIC<3>{}); tap(SpIC<4>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // ---- a row with a tied window row: the dense q tensor, dense instruction, same roles (A = dy, B = x) ----
            SpTrSpan as[2];
            SpTrFrag bf[2][NP];
            auto read_dy = [&](int ks, int slot) {
#pragma unroll
                for (int q = 0; q < NP; ++q) sp_tr_issue(bf[slot][q], yd0 + ybuf + q * YB + ks * 1024);
            };
            auto read_x = [&](int ks, int p, int slot) { sp_span_issue(as[slot], xd0 + xrow + p * ROWB + ks * 1024); };
            read_dy(0, 0);
            read_x(0, NP - 1, 0);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    const int p = NP - 1 - pi, ph = ks * NP + pi;
                    sp_span_wait(as[ph & 1]);
                    if (pi == 0) sp_tr_wait(bf[ks & 1]);
                    if (pi + 1 < NP) {
                        read_x(ks, p - 1, (ph + 1) & 1);
                    } else if (ks + 1 < KSTEPS) {
                        read_dy(ks + 1, (ks + 1) & 1);
                        read_x(ks + 1, NP - 1, (ph + 1) & 1);
                    }
                    u32x4 xt[KS];
                    xt[0] = sp_span_tap<0>(as[ph & 1]);
                    xt[1] = sp_span_tap<1>(as[ph & 1]);
                    xt[2] = sp_span_tap<2>(as[ph & 1]);
                    xt[3] = sp_span_tap<3>(as[ph & 1]);
                    xt[4] = sp_span_tap<4>(as[ph & 1]);
#pragma unroll
                    for (int q = 0; q <= NP - 1 - p; ++q)
#pragma unroll
                        for (int t = 0; t < KS; ++t) acc[t] = sp_mfma(sp_tr_bits(bf[ks & 1][q]), xt[t], acc[t]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!(GHM_WGRAD_ABLATE & 4)) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        cur_dense = next_dense;
    }

    // ---- epilogue: lane = channel c0 + li; rows = filters k0 + 32 ww + (e & 3) + 8 (e >> 2) + 4 kg: four consecutive filters ----
    float* const ob = a.out + (long)bz * a.split_stride;
    const int c = c0 + li;
    if (c >= a.C) return;
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kf = k0 + ww * 32 + 8 * g + 4 * kg;
            if (kf >= a.K) continue;
            float4* o = reinterpret_cast<float4*>(ob + ((long)c * T + wr * KS + t) * a.K + kf);
            float4 v = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
            if (a.accumulate) {
                const float4 w = *o;
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            *o = v;
        }
}

// ------------------------------------------------------------------------------------------------
// Data gradient of a 3x3 / stride-2 / pad-1 convolution (U-Net encoder, PatchGAN; architectures/p2p.py:20-21) from the split
// q tensor of dy -- the structure of lp_dgrad_s2_kernel (conv_lp.hip): dx[c, 2i+pu, 2j+pv] only receives the taps whose
// parity matches, so the four output parity classes are four small stride-1 gathers over the SAME dy patch; a block
// computes all four classes of BM channels x (RT x 32) class pixels, every k-step (16 dy channels, one tap) feeds exactly
// one class.  The (TN + 1) x 2 distinct dy fragments of a wave's rows are read once per slab (three pieces each) and kept
// in registers; a tap reads the three pieces of its weight fragment and runs six MFMAs per pixel tile.
// a.in_q = dy [N, CH=K, Hc, Wc], a.out = dx [N, R=C, 2Hc, 2Wc] (a.H, a.W = dx grid; a.Hin, a.Win = class grid).
// ------------------------------------------------------------------------------------------------
struct SpDgradS2Extra {
    const float* dact_y;      // out *= act'(dact_y): the backward of the producer's nonlinearity in this epilogue (or null)
    long dact_nstride;
    float dact_alpha;
    const uint2* dact_q;      // ... or the same slope from the SIGN of the first piece of the producer's q copy (half units of 4
    long dact_q_nstride;      // channels; units between samples): 2 bytes per element in this epilogue's own register layout
};

// epilogue of the stride-2 data-gradient kernels: the four parity classes of a lane's class pixel interleaved into dx
// (fp32 and / or the split q copy; bias, activation, the producer's activation backward; or split-K partials)
template <int BM, int WM, int WN, int TM, int TN, int NP>
__device__ __forceinline__ void sp_dgrad_s2_epilogue(const SpConvArgs& a, const SpDgradS2Extra& x, f32x16 (&acc)[4][TM][TN],
                                                     f32x16 (&accc)[4][TM][TN], u32x4* sp_smem, int tid, int wm, int wn, int kg,
                                                     int li, int n, int r0, int i0, int j0, int HWx) {
    // ---- epilogue: lane = class pixel (row i0 + wn*TN + j, column j0 + li); its two column parities are adjacent:
    // one 8-byte store per (row parity, channel) ----
    static_assert(TN == 1, "one class row per wave (the prefetched epilogue operand is sized for it)");
    const long P = (long)a.N * HWx;
    const int ru = r0 + wm * (BM / WM), rl = ru + 4 * kg;
    const long rstride = a.partial ? P : (long)HWx;
    const unsigned lo = 4u * kg * (unsigned)rstride + 2u * li;
    const bool plain = a.partial != nullptr;
    const long rowpix = (long)(2 * (i0 + wn * TN)) * a.W + 2 * j0;
    float* const ub = a.partial ? a.partial + ((long)blockIdx.y * a.R + ru) * P + (long)n * HWx + rowpix
                                : (a.out ? a.out + (long)n * a.out_nstride + (long)ru * HWx + rowpix : nullptr);
    const float* const yb = x.dact_y ? x.dact_y + (long)n * x.dact_nstride + (long)ru * HWx + rowpix : nullptr;
    // what the epilogue READS -- the gradient it accumulates into (U-Net skip connections) or the producer's activation (the
    // PatchGAN's conv -> LeakyRectify -> conv chains), never both -- is fetched in ONE batch before anything else: 32 loads in
    // flight per lane instead of a round trip per channel row (the launches with such an operand took 1.5x the bare product:
    // 0.107 against 0.069 ms on N4 C64 256^2)
    const bool do_acc = !plain && a.accumulate, do_y = !plain && yb != nullptr, do_yq = !plain && x.dact_q != nullptr;
    const float* const auxb = do_y ? yb : (do_acc ? ub : nullptr);
    const long auxs = do_y ? (long)HWx : rstride;
    float2 aux[2][TM][16];
    if (auxb) {
#pragma unroll
        for (int pu = 0; pu < 2; ++pu)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    aux[pu][i][e] = rl + k < a.R ? *reinterpret_cast<const float2*>(auxb + (long)k * auxs + pu * a.W + lo)
                                                 : make_float2(0.f, 0.f);
                }
    }
    if (do_yq) {
        // the producer's q copy, piece 0: the half unit (channel block ru/8 + 4i + g, half kg) of the lane's two pixels per row
        // parity -- 16 loads of 8 bytes instead of 32 (kept in the slots of ``aux``: the two operands never come together)
        const uint2* const qd = x.dact_q + 2 * ((long)n * x.dact_q_nstride + (long)(ru / 8) * HWx + rowpix + 2 * li) + kg;
#pragma unroll
        for (int pu = 0; pu < 2; ++pu)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2* q = qd + 2 * ((long)(i * 4 + g) * HWx + pu * a.W);
                    const bool ok = rl + i * 32 + 8 * g < a.R;
                    const uint2 h0 = ok ? q[0] : make_uint2(0u, 0u), h1 = ok ? q[2] : make_uint2(0u, 0u);
                    aux[pu][i][2 * g] = make_float2(__uint_as_float(h0.x), __uint_as_float(h0.y));
                    aux[pu][i][2 * g + 1] = make_float2(__uint_as_float(h1.x), __uint_as_float(h1.y));
                }
    }
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[cl][i][j][e] += accc[cl][i][j][e];

    float* const sb = reinterpret_cast<float*>(sp_smem);
    if (tid < BM) sb[tid] = (a.bias && !a.partial && r0 + tid < a.R) ? a.bias[r0 + tid] : 0.f;
    __syncthreads();
    const float* const lb = sb + wm * (BM / WM) + 4 * kg;
    {
        constexpr int j = 0;
        // q output: unit (channel block ru/8 + 4i + g, pixel (2*(i0+..)+pu, 2*(j0+li) + {0, 1})), half kg
        uint2* const qrow = (a.out_q && !a.partial) ? a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(ru / 8) * HWx + rowpix + 2 * li) + kg
                                                     : nullptr;
        const long ps2 = 2 * (long)a.N * a.out_q_nstride;
        float2 qv[4];
#pragma unroll
        for (int pu = 0; pu < 2; ++pu)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (rl + k >= a.R) continue;
                    float2 v = make_float2(acc[pu * 2 + 0][i][j][e], acc[pu * 2 + 1][i][j][e]);
                    float2* o = ub ? reinterpret_cast<float2*>(ub + (long)k * rstride + pu * a.W + lo) : nullptr;
                    if (!plain) {
                        v.x += lb[k]; v.y += lb[k];
                        if (do_acc) { v.x += aux[pu][i][e].x; v.y += aux[pu][i][e].y; }      // (fp32 output present)
                        v.x = ghm_act(v.x, a.act, a.alpha);
                        v.y = ghm_act(v.y, a.act, a.alpha);
                        if (do_y) {                     // relu / leaky relu: the slope of the producer
                            v.x *= aux[pu][i][e].x > 0.f ? 1.f : x.dact_alpha;
                            v.y *= aux[pu][i][e].y > 0.f ? 1.f : x.dact_alpha;
                        }
                        if (do_yq) {                    // the same from the bf16 first piece: x > 0 <=> its int16 pattern > 0
                            const float2 h0 = aux[pu][i][2 * (e >> 2)], h1 = aux[pu][i][2 * (e >> 2) + 1];
                            const unsigned w0 = __float_as_uint((e & 2) ? h0.y : h0.x), w1 = __float_as_uint((e & 2) ? h1.y : h1.x);
                            const int s0 = (e & 1) ? ((int)w0 >> 16) : (int)(short)(w0 & 0xffffu);
                            const int s1 = (e & 1) ? ((int)w1 >> 16) : (int)(short)(w1 & 0xffffu);
                            v.x *= s0 > 0 ? 1.f : x.dact_alpha;
                            v.y *= s1 > 0 ? 1.f : x.dact_alpha;
                        }
                    }
                    if (ub) *o = v;
                    if (qrow) {           // four consecutive channels (e = 4g .. 4g+3) of the lane's two pixels
                        qv[e & 3] = v;
                        if ((e & 3) == 3) {
                            uint2* qo = qrow + 2 * ((long)(i * 4 + (e >> 2)) * HWx + pu * a.W);
                            sp_qstore4<NP>(qo, qv[0].x, qv[1].x, qv[2].x, qv[3].x, ps2);
                            sp_qstore4<NP>(qo + 2, qv[0].y, qv[1].y, qv[2].y, qv[3].y, ps2);
                        }
                    }
                }
    }
}

// TUNING ONLY (results wrong): -DGHM_DGS2_ABLATE=<bits>: 1 = no weight DMA inside the loop, 2 = no patch DMA inside the loop,
// 4 = weight fragments read once per filter row only, 8 = no waits / barriers inside the loop
#ifndef GHM_DGS2_ABLATE
#define GHM_DGS2_ABLATE 0
#endif
template <int BM, int RT, int NP>
__global__ __launch_bounds__(256, 2) void sp_dgrad_s2_kernel(const SpConvArgs a, const SpDgradS2Extra x) {
    typedef SpProd<NP> PR;
    // weights staged per filter ROW (3 taps x 2 channel blocks x BM rows x 3 pieces, double-buffered: 36 KB) and the dy patch
    // per slab (double-buffered): 68 KB in the 64 x 4 shape -- two blocks per CU, and room left for the other streams' kernels.
    // (Three weight buffers staged two rows ahead, the barrier waiting for the older row only: 146 / 167 / 140 TFLOP/s on the step's
    // three geometries, the same as this form -- the L2 round trip per row is not what bounds it.  In the step the launch is as
    // much epilogue as contraction: up to 14 bytes per output element -- producer's activation read, fp32 and split gradient
    // written -- against 576 flops; what cost most was the round trip of the epilogue's READS, now one batch: the nine launches
    // of the step 1.13 -> 0.97 ms, the bare products sum to 0.74.)
    constexpr int T = 9, WM = 2, WN = 2;
    constexpr int TM = BM / (WM * 32), TN = RT / WN;
    constexpr int PH = RT + 1, PW = 33;
    constexpr int PU1 = 2 * PH * PW, PUNITS = NP * PU1;
    constexpr int WU1 = 2 * 3 * BM, WUNITS = NP * WU1;           // one filter row: [cb][3 taps][BM]
    constexpr int NQ = (PU1 + 255) / 256;
    constexpr int NI = WU1 / 64;
    extern __shared__ __attribute__((aligned(16))) u32x4 sp_smem[];
    u32x4* const Wl = sp_smem;
    u32x4* const Pl = sp_smem + 2 * WUNITS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int kg = lane >> 5, li = lane & 31;
    const int Hc = a.Hin, Wc = a.Win;
    const int ntr = (a.R + BM - 1) / BM;
    const int tiles_x = Wc / 32, tiles_y = Hc / RT;
    int L = sp_xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = (L % ntr) * BM;
    L /= ntr;
    const int tx = L % tiles_x;
    L /= tiles_x;
    const int ty = L % tiles_y;
    const int n = L / tiles_y;
    const int i0 = ty * RT, j0 = tx * 32;
    const int HWc = Hc * Wc, HWx = a.H * a.W;
    const int nslabs = a.CH / 16;
    const int s_begin = blockIdx.y * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);

    int p_off[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * 256;
        const int cb = e / (PH * PW), rem = e - cb * (PH * PW);
        const int py = rem / PW, px = rem - py * PW;
        const int y = i0 + py, xx = j0 + px;
        const bool ok = e < PU1 && y < Hc && xx < Wc;
        p_off[q] = ok ? cb * HWc + y * Wc + xx : -1;
    }
    const u32x4* ibase = a.in_q + (long)n * a.in_q_nstride + (long)s_begin * 2 * HWc;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto stage_patch = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q * 256 + wave * 64 < PU1) {
                    const u32x4* g = p_off[q] >= 0 ? ibase + p * a.in_q_pstride + p_off[q] : a.zeros;
                    if (tid + q * 256 < PU1)
                        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Pl + buf * PUNITS + p * PU1 + q * 256 + wave * 64), 16, 0, 0);
                }
            }
        ibase += 2 * HWc;
    };
    // filter row fa of slab s in the transposed pack: taps tw = 3 fa .. 3 fa + 2
    auto stage_weights = [&](int s, int fa, int buf) {
        const u32x4* src = a.wq + ((long)(2 * s) * T + 3 * fa) * a.Rpad + r0 + lane;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int w0 = 0; w0 < NI; w0 += 4) {
                const int w = w0 + wave;
                if (w < NI) {
                    const int ci = w / (BM / 64), h = w - ci * (BM / 64);        // ci = cb * 3 + tap in the row
                    const int cb = ci / 3, b = ci - cb * 3;
                    const u32x4* g = src + p * a.wq_pstride + ((long)cb * T + b) * a.Rpad + h * 64;
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Wl + buf * WUNITS + p * WU1 + ci * BM + h * 64), 16, 0, 0);
                }
            }
    };

    // [parity class pu*2+pv]; leading product and correction products apart (see sp_conv_kernel)
    f32x16 acc[4][TM][TN], accc[4][TM][TN];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[cl][i][j][e] = accc[cl][i][j][e] = 0.f;

    if (s_begin < s_end) {
        stage_weights(s_begin, 0, 0);
        stage_patch(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int wlane = kg * 3 * BM + wm * (BM / WM) + li;
    const int plane = kg * PH * PW + (wn * TN) * PW + li;
    int it = 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int pbuf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        const u32x4* Pb = Pl + pbuf * PUNITS + plane;
        // the nine taps only ever read the dy fragment of a class pixel at (row, column) offsets {0, 1} x {0, 1}: the
        // (TN + 1) x 2 distinct fragments of the wave's rows are read once per slab (three pieces) and kept in registers
        u32x4 bf[NP][TN + 1][2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j <= TN; ++j) {
                bf[p][j][0] = Pb[p * PU1 + j * PW];
                bf[p][j][1] = Pb[p * PU1 + j * PW + 1];
            }
#pragma unroll
        for (int fa = 0; fa < 3; ++fa, ++it) {
            const int wbuf = it & 1;
            if (GHM_DGS2_ABLATE & 1) {
            } else if (fa + 1 < 3)
                stage_weights(s, fa + 1, wbuf ^ 1);
            else if (more)
                stage_weights(s + 1, 0, wbuf ^ 1);
            if (fa == 0 && more && !(GHM_DGS2_ABLATE & 2)) stage_patch(pbuf ^ 1);
            const u32x4* Wb = Wl + wbuf * WUNITS + wlane;
            u32x4 af[2][NP][TM];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i) af[0][p][i] = Wb[p * WU1 + i * 32];
            // tw = 3 fa + b: tap index in the transposed pack = 8 - original tap (ta, tb); the tap feeds class
            // (ta != 1, tb != 1) from dy[i + (ta == 0)][j + (tb == 0)].  Weight fragments are read one tap ahead.
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (b + 1 < 3 && !(GHM_DGS2_ABLATE & 4)) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[(b + 1) & 1][p][i] = Wb[p * WU1 + (b + 1) * BM + i * 32];
                } else if (b + 1 < 3) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[(b + 1) & 1][p][i] = af[b & 1][p][i];
                }
                const int tw = 3 * fa + b;
                const int ta = (8 - tw) / 3, tb = (8 - tw) % 3;
                const int cl = (ta == 1 ? 0 : 2) + (tb == 1 ? 0 : 1);
                const int ro = ta == 0 ? 1 : 0, co = tb == 0 ? 1 : 0;
#pragma unroll
                for (int pr = 0; pr < PR::N; ++pr) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (pr < PR::N - 1)
                                accc[cl][i][j] = sp_mfma(af[b & 1][PR::A[pr]][i], bf[PR::B[pr]][j + ro][co], accc[cl][i][j]);
                            else
                                acc[cl][i][j] = sp_mfma(af[b & 1][0][i], bf[0][j + ro][co], acc[cl][i][j]);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(GHM_DGS2_ABLATE & 8)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }
    sp_dgrad_s2_epilogue<BM, WM, WN, TM, TN, NP>(a, x, acc, accc, sp_smem, tid, wm, wn, kg, li, n, r0, i0, j0, HWx);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct SpPlan {
    bool ok;
    int np;            // pieces per operand: 3 ('bf16x3', six products) or 2 ('bf16x2', three products)
    int bm, rt, wm, wn, tw, splits, slabs_per_split, grid;
    int blocks;        // blocks launched along x: grid (= tiles) or fewer, persistent ones (sp_conv2_kernel)
    size_t lds;
};

int sp_rpad(int r) { return (r + 127) / 128 * 128; }
int sp_nblk(int red) { return (red + 15) / 16 * 2; }      // channel blocks of a pack (as the low-precision packs: whole slabs)

size_t sp_lds_bytes(int ks, int st, int bm, int rt, int tw, int np) {
    const int rows = rt * (32 / tw);
    const int ph = (rows - 1) * st + ks, pw = (tw - 1) * st + ks;
    return ((size_t)2 * np * (2 * ks * bm + (2 * ph * pw + 63) / 64 * 64) + 64 + 32) * 16;      // SpGeo2::LDS_BYTES
}

// forward form: CH reduction channels, R output channels, (H, W) output grid
SpPlan sp_plan(int N, int CH, int H, int W, int R, int ks, int st, int num_cu, int np = 3, bool bm64 = false) {
    SpPlan p;
    memset(&p, 0, sizeof(p));
    p.np = np;
    if (GHM_OPT("GHM_NO_SPLIT")) return p;
    if (!((ks == 3 && (st == 1 || st == 2)) || (ks == 5 && st == 1))) return p;
    // one block per CU (three pieces of both operands, double-buffered, fill the LDS): 3x3 stride 1 takes 128 filters x 8 rows
    // with eight waves; 5x5 and stride 2 take 64 filters with four
    p.tw = W % 32 == 0 ? 32 : (W % 16 == 0 ? 16 : 8);
    if (p.tw != 32)
        if (const char* f = GHM_OPT("GHM_SPLIT_NO_NARROW")) {       // 1: all narrow maps; 3 / 5: those of that filter size only (tuning)
            const int v = atoi(f);
            if (v == 1 || v == ks) return p;
        }
    if (p.tw == 32) {
        // 3x3 stride 1, 128 filters: eight waves of 2 x 2 tiles (four waves of 4 x 2 tiles -- a quarter fewer fragment reads per
        // MFMA -- measured 0-19 % slower alone: 212 against 231 TFLOP/s on the N4 C128 128^2 K256 data gradient)
        if (ks == 3 && st == 1) {
            p.bm = (R >= 96 && !bm64 && !GHM_OPT("GHM_SPLIT_BM64")) ? 128 : 64;
            if (p.bm == 128 && H % 8 == 0 && !GHM_OPT("GHM_SPLIT_BM128")) {
                // a launch of 128 .. 255 tiles of 128 filters leaves half the CUs idle (the N4 C1024 64^2 K256 forward: 180
                // TFLOP/s alone, 250 in 256 tiles of 64 filters; the N4 K512 -> C1024 32^2 data gradient 175 -> 239): whole
                // rounds of blocks over the CUs, priced with the two shapes' rates alone (235 / 220 TFLOP/s) -- the 64-filter
                // shape makes twice the tiles at half the work each
                const long g128 = (long)((R + 127) / 128) * (W / 32) * (H / 8) * N, g64 = (long)((R + 63) / 64) * (W / 32) * (H / 8) * N;
                const double t128 = (double)((g128 + num_cu - 1) / num_cu) / 235.0, t64 = 0.5 * (double)((g64 + num_cu - 1) / num_cu) / 220.0;
                if (g128 >= num_cu / 2 && t64 < t128) p.bm = 64;
            }
            p.rt = 8; p.wm = p.bm == 128 ? 2 : 1; p.wn = 4;
        }
        else if (ks == 5) { p.bm = 64; p.rt = 8; p.wm = 1; p.wn = 4; }
        else {      // 3x3 stride 2: the patch footprint allows 64 filters x 4 rows x 32 columns on four waves (one per SIMD: DESIGN 4d,
                    // round-6 counters).  GHM_SPLIT_S2_W8: eight waves of one tile each -- two per SIMD, twice the fragment reads per MFMA:
                    // measured 287.4 / 286.0 img/s against 290.5 / 289.3 in the joint step, not the default
            p.bm = 64; p.rt = 4; p.wm = 2; p.wn = GHM_OPT("GHM_SPLIT_S2_W8") ? 4 : 2;
        }
    } else {        // narrow maps: 64 filters x (8 rows x 16 columns | 8 x 8): fragments of 2 x 16 / 4 x 8 pixels, four waves
        p.bm = 64; p.rt = p.tw == 16 ? 4 : 2; p.wm = 2; p.wn = 2;
    }
    const int rows = p.rt * (32 / p.tw);
    if (R < 32 || (W % p.tw) || (H % rows) || (CH % 16) || CH < 16) return p;
    p.lds = sp_lds_bytes(ks, st, p.bm, p.rt, p.tw, np);
    if (p.lds > 160 * 1024) return p;
    const int ntr = (R + p.bm - 1) / p.bm;
    p.grid = ntr * (W / p.tw) * (H / rows) * N;
    const int nslabs = CH / 16;
    p.splits = 1;
    if (p.grid < num_cu / 2) {
        p.splits = (num_cu + p.grid - 1) / p.grid;
        const int maxs = nslabs / 2 > 0 ? nslabs / 2 : 1;
        if (p.splits > maxs) p.splits = maxs;
    }
    if (const char* f = GHM_OPT("GHM_SPLIT_SPLITS")) p.splits = atoi(f) < nslabs ? (atoi(f) > 0 ? atoi(f) : 1) : nslabs;
    p.slabs_per_split = (nslabs + p.splits - 1) / p.splits;
    p.splits = (nslabs + p.slabs_per_split - 1) / p.slabs_per_split;
    // persistent blocks (single-pass plans): GHM_SPLIT_PERSIST = blocks as a fraction of the CUs (0: one block per tile)
    p.blocks = p.grid;
    if (p.splits == 1 && p.wm * p.wn == 4) {
        const char* f = GHM_OPT("GHM_SPLIT_PERSIST");
        const double frac = f ? atof(f) : 1.0;
        int nb = (int)(frac * num_cu) / 8 * 8;
        if (frac > 0 && nb >= 8 && nb < p.grid) p.blocks = nb;
    }
    p.ok = true;
    return p;
}

static inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

int sp_pack(ghm_ctx* ctx, const float* x, long x_nstride, int N, int C, int HW, void* q, long q_nstride, long q_pstride, int np) {
    const long total = (long)N * (C / 8) * HW;
    if (total == 0) return 0;
    if (np == 3)
        hipLaunchKernelGGL(sp_pack_kernel<3>, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, x, x_nstride, N, C / 8, HW,
                           (u32x4*)q, q_nstride, q_pstride);
    else
        hipLaunchKernelGGL(sp_pack_kernel<2>, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, x, x_nstride, N, C / 8, HW,
                           (u32x4*)q, q_nstride, q_pstride);
    GHM_LAUNCH_CHECK();
    return 0;
}

#define GHM_SP_PIECES_OK(np) GHM_CHECK((np) == 2 || (np) == 3, "split convolution: pieces must be 3 ('bf16x3') or 2 ('bf16x2')")

// > 64 KB of dynamic LDS needs the attribute once per kernel (not per launch: it is a driver call on the step's issue path)
template <typename K>
int sp_set_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return 0;
    static const void* seen[64];
    static int nseen = 0;
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == (const void*)kernel) return 0;
    GHM_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (nseen < 64) seen[nseen++] = (const void*)kernel;
    return 0;
}

// one tile shape of the forward-form kernel, for 3 or 2 pieces per operand
template <int KS, int ST, int BM, int RT, int WM, int WN, bool POOL, int TW>
int sp_launch_variant(ghm_ctx* ctx, dim3 g, size_t lds, const SpConvArgs& a, int np, int cls = 0) {
    static_assert(SpGeo2<KS, ST, BM, RT, WM, WN, TW, 3>::LDS_BYTES <= 160 * 1024, "LDS");
    if (cls) {          // the class forms of the collapsed bilinear convolution (structural zero taps skipped)
        if constexpr (KS == 3 && ST == 1 && !POOL && TW == 32) {
            // (forward class form: 64-filter tiles only -- the eight-wave 128-filter shape spills 232 bytes per lane with it;
            // the data-gradient form fits it with 20)
#define GHM_CLS_CASE(NP_, C_)                                                                                                     \
            if constexpr (BM == 64 || C_ == 2)                                                                                    \
                if (np == NP_ && cls == C_) {                                                                                     \
                    if (int e = sp_set_lds(sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, NP_, 0, C_>, lds)) return e;         \
                    hipLaunchKernelGGL((sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, NP_, 0, C_>), g, dim3(WM * WN * 64), lds, \
                                       ctx->stream, a);                                                                           \
                    GHM_LAUNCH_CHECK();                                                                                           \
                    return 0;                                                                                                     \
                }
            GHM_CLS_CASE(3, 1) GHM_CLS_CASE(3, 2) GHM_CLS_CASE(2, 1) GHM_CLS_CASE(2, 2)
#undef GHM_CLS_CASE
        }
        ghm_set_error("split-fp32 convolution: no class form for this tile shape");
        return -3;
    }
#ifdef GHM_SPLIT_ABLATION
    if constexpr ((KS == 5 || (KS == 3 && ST == 1)) && BM == 64 && RT == 8 && !POOL && TW == 32) {
        const char* f = GHM_OPT("GHM_SPLIT_ABLATE");
        const int abl = f ? atoi(f) : 0;
#define GHM_ABL_CASE(A_)                                                                                              \
        if (abl == A_ && np == 3) {                                                                                   \
            if (int e = sp_set_lds(sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, 3, A_>, lds)) return e;           \
            hipLaunchKernelGGL((sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, 3, A_>), g, dim3(WM * WN * 64), lds, ctx->stream, a); \
            GHM_LAUNCH_CHECK();                                                                                       \
            return 0;                                                                                                 \
        }
        GHM_ABL_CASE(1) GHM_ABL_CASE(2) GHM_ABL_CASE(4) GHM_ABL_CASE(7)
#undef GHM_ABL_CASE
    }
#endif
    if (np == 3) {
        if (int e = sp_set_lds(sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, 3>, lds)) return e;
        hipLaunchKernelGGL((sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, 3>), g, dim3(WM * WN * 64), lds, ctx->stream, a);
    } else {
        if (int e = sp_set_lds(sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, 2>, lds)) return e;
        hipLaunchKernelGGL((sp_conv2_kernel<KS, ST, BM, RT, WM, WN, POOL, TW, 2>), g, dim3(WM * WN * 64), lds, ctx->stream, a);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

// in32 != null: the fp32 operand is split into the launch's workspace first
int sp_launch_conv(ghm_ctx* ctx, const SpPlan& pl, SpConvArgs a, int ks, int st, const float* in32, long in32_nstride, bool pool,
                   int cls = 0) {
    a.slabs_per_split = pl.slabs_per_split;
    a.zeros = (const u32x4*)ctx->zeros;
    a.partial = nullptr;
    const long plane = (long)a.N * (a.CH / 8) * a.Hin * a.Win;
    const int NP = pl.np;
    const size_t qbytes = in32 ? align256((size_t)NP * plane * 16) : 0;
    const size_t pbytes = pl.splits > 1 ? (size_t)pl.splits * a.R * a.N * a.H * a.W * sizeof(float) : 0;
    if (qbytes + pbytes) {
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, qbytes + pbytes, &ws)) return e;
        if (in32) {
            a.in_q = (const u32x4*)ws;
            a.in_q_nstride = (long)(a.CH / 8) * a.Hin * a.Win;
            a.in_q_pstride = plane;
            if (int e = sp_pack(ctx, in32, in32_nstride, a.N, a.CH, a.Hin * a.Win, ws, a.in_q_nstride, plane, NP)) return e;
        }
        if (pbytes) a.partial = (float*)((char*)ws + qbytes);
    }
    GHM_CHECK(!(pool && pl.splits > 1), "split-fp32 pooled convolution needs a single-pass plan");
    GHM_CHECK(!(a.out_q && pl.splits > 1), "split-fp32 convolution: a q output needs a single-pass plan (ask ghm_split_q_direct)");
    GHM_CHECK(pool || a.out || a.out_q, "split-fp32 convolution: no output");
    GHM_CHECK(!(a.accumulate && !a.out), "accumulate needs the fp32 output");
    GHM_CHECK(!a.out_q || (a.R % 8 == 0 && ((uintptr_t)a.out_q & 15) == 0), "q output: a multiple of 8 channels, 16-byte aligned");
    const dim3 g(cls ? pl.grid : pl.blocks, pl.splits);       // (the class forms run one tile per block)
    a.ntiles = pl.grid;
#define GHM_SP_CASE(KS_, ST_, BM_, RT_, WM_, WN_, POOL_, TW_)                                                    \
    if (ks == KS_ && st == ST_ && pl.bm == BM_ && pl.rt == RT_ && pl.wm == WM_ && pl.wn == WN_ && pool == POOL_ && pl.tw == TW_) {  \
        if (int e = sp_launch_variant<KS_, ST_, BM_, RT_, WM_, WN_, POOL_, TW_>(ctx, g, pl.lds, a, pl.np, cls)) return e;  \
    } else
    GHM_SP_CASE(5, 1, 64, 8, 1, 4, false, 32)
    GHM_SP_CASE(5, 1, 64, 8, 1, 4, true, 32)
    GHM_SP_CASE(3, 1, 128, 8, 2, 4, false, 32)
    GHM_SP_CASE(3, 1, 64, 8, 1, 4, false, 32)
    GHM_SP_CASE(3, 1, 128, 8, 2, 4, true, 32)
    GHM_SP_CASE(3, 1, 64, 8, 1, 4, true, 32)
    GHM_SP_CASE(3, 2, 64, 4, 2, 2, false, 32)
    GHM_SP_CASE(3, 2, 64, 4, 2, 4, false, 32)
    GHM_SP_CASE(5, 1, 64, 4, 2, 2, false, 16)
    GHM_SP_CASE(3, 1, 64, 4, 2, 2, false, 16)
    GHM_SP_CASE(3, 2, 64, 4, 2, 2, false, 16)
    GHM_SP_CASE(5, 1, 64, 2, 2, 2, false, 8)
    GHM_SP_CASE(3, 1, 64, 2, 2, 2, false, 8)
    GHM_SP_CASE(3, 2, 64, 2, 2, 2, false, 8) {
        ghm_set_error("no split-fp32 convolution variant for k=%d s=%d bm=%d rt=%d pool=%d", ks, st, pl.bm, pl.rt, (int)pool);
        return -3;
    }
#undef GHM_SP_CASE
    if (pl.splits > 1)
        return ghm_splitk_finish(ctx, a.partial, pl.splits, a.out, a.bias, a.N, a.R, a.H, a.W, a.out_nstride, a.act, a.alpha,
                                 a.accumulate);
    return 0;
}

// ---- weight gradient plan: one round of resident blocks (a block per CU) ----
struct SpWPlan {
    bool ok;
    int np;
    int cht, ct, spx, splits_per_col, rows_per_split, ncols;
    int la;            // rows of DMA lookahead (2: 3x3 stride 1 on 32-pixel strips)
    size_t lds;
};

size_t sp_wgrad_lds(int ks, int st, int cht, int ct, int spx, int NP, int la = 1) {
    const int xpix = st == 1 ? spx + ks - 1 : spx + 1, xch = (xpix + 15) / 16;
    const size_t rowb = (size_t)cht * st * xch * 16 * 64, yb = (size_t)ct * spx * 64;
    return NP * ((ks + la * st) * rowb + (la + 1) * yb) + (la == 2 ? 1024 : 0);
}

SpWPlan sp_wplan(const ghm_conv_desc* d, int num_cu, int np = 3) {
    SpWPlan v;
    memset(&v, 0, sizeof(v));
    v.np = np;
    if (GHM_OPT("GHM_NO_SPLIT") || GHM_OPT("GHM_NO_SPLIT_WGRAD")) return v;
    const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2);
    const bool k5 = d->kh == 5 && d->kw == 5 && d->pad == 2 && d->stride == 1;
    if (!k3 && !k5) return v;
    if (d->Ho != (d->H + d->stride - 1) / d->stride || d->Wo != (d->W + d->stride - 1) / d->stride) return v;
    if (d->Wo % 16 || d->C % 8 || d->K % 8 || (d->Wo % 32 && GHM_OPT("GHM_SPLIT_NO_NARROW"))) return v;
    const int narrow = d->Wo % 32 ? 16 : 32;    // 16-wide maps: strips of 16 pixels, one k-step per output row
    if (k5) {                                   // 10 waves: 5 filter rows x 2 filter tiles of one 32-channel group
        if (d->K % 64 || d->C % 32) return v;
        v.cht = 1; v.ct = 2; v.spx = d->Wo % 64 == 0 ? 64 : narrow;
    } else if (d->stride == 1) {                // 12 waves: 3 filter rows x (2 channel groups x 2 filter tiles)
        if (d->K % 64 || d->C % 64) return v;
        v.cht = 2; v.ct = 2; v.spx = narrow;
    } else {                                    // stride 2 (two parity planes per x row): 12 waves, 1 x 4
        if (d->K % 128 || d->C % 32) return v;
        v.cht = 1; v.ct = 4; v.spx = narrow;
    }
    // (two rows of lookahead: built and measured in round 6, NOT the default -- the joint step 300.1 / 299.5 / 296.5 img/s with it
    // against 299.8 / 301.0 / 298.3 without, the class-form launches 0.96 against 0.92 ms alone: the end-of-row wait is not what
    // holds this kernel.  GHM_SPLIT_WGRAD_LA2=1 turns it on; tests/test_gpu_split.py covers both.)
    v.la = (k3 && d->stride == 1 && v.spx == 32 && GHM_OPT("GHM_SPLIT_WGRAD_LA2")) ? 2 : 1;
    v.lds = sp_wgrad_lds(d->kh, d->stride, v.cht, v.ct, v.spx, np, v.la);
    if (v.lds > 160 * 1024) return v;
    v.ncols = d->N * (d->Wo / v.spx);
    const long tiles = (long)(d->C / (32 * v.cht)) * (d->K / (32 * v.ct)) * v.ncols;
    long S = num_cu / tiles;
    if (const char* f = GHM_OPT("GHM_SPLIT_WGRAD_ROUNDS")) S = (long)(atof(f) * num_cu / tiles);      // tuning
    const long minrows = d->kh == 5 ? 8 : 4;
    const long max_by_work = d->Ho / minrows > 0 ? d->Ho / minrows : 1;
    if (S > max_by_work) S = max_by_work;
    if (S < 1) S = 1;
    v.rows_per_split = (int)((d->Ho + S - 1) / S);
    v.splits_per_col = (d->Ho + v.rows_per_split - 1) / v.rows_per_split;
    v.ok = true;
    return v;
}

int sp_launch_wgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const SpWPlan& v, const void* xq, long xq_ns, long xq_ps,
                    const void* dyq, long dyq_ns, long dyq_ps, float* dwp, void* workspace, int accumulate, int cls_k = 0) {
    SpWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.cls_k = cls_k;
    a.xq = (const u32x4*)xq; a.xq_ns = xq_ns; a.xq_ps = xq_ps; a.dyq = (const u32x4*)dyq; a.dyq_ns = dyq_ns; a.dyq_ps = dyq_ps;
    a.zeros = (const u32x4*)ctx->zeros;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.Ho = d->Ho; a.Wo = d->Wo;
    a.rows_per_split = v.rows_per_split; a.splits_per_col = v.splits_per_col;
    a.xcd = GHM_OPT("GHM_SPLIT_WGRAD_NO_XCD") ? 0 : 1;
    const long n = (long)d->C * d->kh * d->kw * d->K;
    const int splits = v.ncols * v.splits_per_col;
    if (splits > 1) {
        GHM_CHECK(workspace != nullptr, "split-fp32 weight gradient needs a workspace for %d splits", splits);
        a.out = (float*)workspace; a.split_stride = n; a.accumulate = 0;
    } else {
        a.out = dwp; a.split_stride = 0; a.accumulate = accumulate;
    }
    const dim3 grid(d->C / (32 * v.cht), d->K / (32 * v.ct), splits);
    if (cls_k) {        // the class form of the collapsed bilinear convolution: 3x3 stride 1, 32-pixel strips
        GHM_CHECK(d->kh == 3 && d->stride == 1 && v.cht == 2 && v.ct == 2 && v.spx == 32 && cls_k % 64 == 0 && d->K == 4 * cls_k,
                  "split-fp32 weight gradient: no class form for this geometry");
#define GHM_SPW_CLS(NP_, LA_)                                                                                           \
        if (v.np == NP_ && v.la == LA_) {                                                                               \
            if (int e = sp_set_lds(sp_wgrad_kernel<3, 1, 2, 2, 32, NP_, 1, LA_>, v.lds)) return e;                      \
            hipLaunchKernelGGL((sp_wgrad_kernel<3, 1, 2, 2, 32, NP_, 1, LA_>), grid, dim3(2 * 2 * 3 * 64), v.lds, ctx->stream, a); \
        }
        GHM_SPW_CLS(3, 1) GHM_SPW_CLS(3, 2) GHM_SPW_CLS(2, 1) GHM_SPW_CLS(2, 2)
#undef GHM_SPW_CLS
        GHM_LAUNCH_CHECK();
        if (splits > 1) return ghm_reduce_splits(ctx, (const float*)workspace, splits, n, n, dwp, accumulate);
        return 0;
    }
#define GHM_SPW_CASE(KS_, ST_, CHT_, CT_, SPX_, LA_)                                                                \
    if (d->kh == KS_ && d->stride == ST_ && v.cht == CHT_ && v.ct == CT_ && v.spx == SPX_ && v.la == LA_) {        \
        if (v.np == 3) {                                                                                            \
            if (int e = sp_set_lds(sp_wgrad_kernel<KS_, ST_, CHT_, CT_, SPX_, 3, 0, LA_>, v.lds)) return e;         \
            hipLaunchKernelGGL((sp_wgrad_kernel<KS_, ST_, CHT_, CT_, SPX_, 3, 0, LA_>), grid, dim3(CHT_ * CT_ * KS_ * 64), v.lds, \
                               ctx->stream, a);                                                                   \
        } else {                                                                                                  \
            if (int e = sp_set_lds(sp_wgrad_kernel<KS_, ST_, CHT_, CT_, SPX_, 2, 0, LA_>, v.lds)) return e;         \
            hipLaunchKernelGGL((sp_wgrad_kernel<KS_, ST_, CHT_, CT_, SPX_, 2, 0, LA_>), grid, dim3(CHT_ * CT_ * KS_ * 64), v.lds, \
                               ctx->stream, a);                                                                   \
        }                                                                                                         \
        GHM_LAUNCH_CHECK();                                                                                       \
    } else
    GHM_SPW_CASE(3, 1, 2, 2, 32, 2)
    GHM_SPW_CASE(3, 1, 2, 2, 32, 1)
    GHM_SPW_CASE(3, 2, 1, 4, 32, 1)
    GHM_SPW_CASE(5, 1, 1, 2, 64, 1)
    GHM_SPW_CASE(5, 1, 1, 2, 32, 1)
    GHM_SPW_CASE(3, 1, 2, 2, 16, 1)
    GHM_SPW_CASE(3, 2, 1, 4, 16, 1)
    GHM_SPW_CASE(5, 1, 1, 2, 16, 1) {
        ghm_set_error("no split-fp32 weight gradient variant for k=%d s=%d cht=%d ct=%d spx=%d", d->kh, d->stride, v.cht, v.ct, v.spx);
        return -3;
    }
#undef GHM_SPW_CASE
    if (splits > 1) return ghm_reduce_splits(ctx, (const float*)workspace, splits, n, n, dwp, accumulate);
    return 0;
}

// the pooled-gradient form (sp_wgrad_pooled_kernel): 5x5 stride 1, 32 channels x 64 filters, strips of 64 or 32 pixels
bool sp_wgrad_pooled_ok(const ghm_conv_desc* d, const SpWPlan& v) {
    return v.ok && d->kh == 5 && d->kw == 5 && d->stride == 1 && v.cht == 1 && v.ct == 2 && (v.spx == 64 || v.spx == 32) && d->Wo % 32 == 0 &&
           d->Ho % 2 == 0 && d->K % 64 == 0 && !GHM_OPT("GHM_NO_SPARSE_WGRAD");
}

int sp_launch_wgrad_pooled(ghm_ctx* ctx, const ghm_conv_desc* d, const SpWPlan& v, const void* xq, long xq_ns, long xq_ps,
                           const void* dyq, long dyq_ns, long dyq_ps, const void* cq, long cq_ns, long cq_ps, const void* cidx,
                           const int* cflags, float* dwp, void* workspace, int accumulate) {
    SpWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.xq = (const u32x4*)xq; a.xq_ns = xq_ns; a.xq_ps = xq_ps; a.dyq = (const u32x4*)dyq; a.dyq_ns = dyq_ns; a.dyq_ps = dyq_ps;
    a.cq = (const u32x4*)cq; a.cq_ns = cq_ns; a.cq_ps = cq_ps; a.cidx = (const u32x4*)cidx; a.cflags = cflags;
    a.zeros = (const u32x4*)ctx->zeros;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.Ho = d->Ho; a.Wo = d->Wo;
    a.rows_per_split = v.rows_per_split; a.splits_per_col = v.splits_per_col;
    a.xcd = GHM_OPT("GHM_SPLIT_WGRAD_NO_XCD") ? 0 : 1;
    const long n = (long)d->C * d->kh * d->kw * d->K;
    const int splits = v.ncols * v.splits_per_col;
    if (splits > 1) {
        GHM_CHECK(workspace != nullptr, "split-fp32 weight gradient needs a workspace for %d splits", splits);
        a.out = (float*)workspace; a.split_stride = n; a.accumulate = 0;
    } else {
        a.out = dwp; a.split_stride = 0; a.accumulate = accumulate;
    }
    const dim3 grid(d->C / 32, d->K / 64, splits);
    const size_t lds = v.lds + 2 * 1024;
#define GHM_SPWP_CASE(SPX_, NP_)                                                                              \
    if (v.spx == SPX_ && v.np == NP_) {                                                                       \
        if (int e = sp_set_lds(sp_wgrad_pooled_kernel<SPX_, NP_>, lds)) return e;                             \
        hipLaunchKernelGGL((sp_wgrad_pooled_kernel<SPX_, NP_>), grid, dim3(640), lds, ctx->stream, a);        \
    }
    GHM_SPWP_CASE(64, 3) GHM_SPWP_CASE(32, 3) GHM_SPWP_CASE(64, 2) GHM_SPWP_CASE(32, 2)
#undef GHM_SPWP_CASE
    GHM_LAUNCH_CHECK();
    if (splits > 1) return ghm_reduce_splits(ctx, (const float*)workspace, splits, n, n, dwp, accumulate);
    return 0;
}

// ---- 3x3 stride-2 data gradient ----
SpPlan sp_plan_dgrad_s2(const ghm_conv_desc* d, int num_cu, int np = 3) {
    SpPlan p;
    memset(&p, 0, sizeof(p));
    p.np = np;
    // (a first form with all nine taps of a slab staged at once -- 142 KB of LDS, one block per CU -- was 2-25 % faster than
    // dgrad_s2_patch_kernel alone and 1.6 % SLOWER in the step: it shut the other streams' kernels out of its CU.  This one
    // stages a filter row at a time: 55 KB, two blocks per CU; alone 154-177 TFLOP/s fp32-equivalent against 107-115, joint
    // step 254.2 -> 269.5 img/s.  GHM_NO_SPLIT_DGRAD_S2 switches it off.)
    if (GHM_OPT("GHM_NO_SPLIT") || GHM_OPT("GHM_NO_SPLIT_DGRAD_S2")) return p;
    if (!(d->stride == 2 && d->kh == 3 && d->kw == 3 && d->pad == 1 && d->H == 2 * d->Ho && d->W == 2 * d->Wo)) return p;
    if (d->Wo % 32 || d->K % 16 || d->K < 16 || d->C < 32 || (d->x_nstride & 1) || ((d->H * d->W) & 1)) return p;
    p.bm = 64;
    // 64 channels x 2 class rows: eight accumulator tiles per wave (leading + correction) -> two blocks per CU.  (Round 5 built
    // the opposite trade -- 64 x 4 rows x 32 columns on four waves, one block per CU, the pipelined loop of sp_conv2_kernel with a
    // slab of weights double-buffered, 170 instead of 300 bytes of staging per MFMA -- and measured it SLOWER: 98 / 137 / 168
    // against 155 / 172 / 182 TFLOP/s on the N8 C64 256^2 / C128 128^2 / C256 64^2 layers, the step 262 against 270 img/s.  These
    // layers contract over only 8-32 slabs and write four times the pixels they read: a block is 12-50 us of MFMAs between a
    // prologue and a store-heavy epilogue, and with one block per CU nothing runs under those; two co-resident blocks hide them.)
    p.rt = 2;
    if (d->Ho % p.rt) return p;
    p.lds = (size_t)2 * np * (2 * 3 * p.bm + 2 * (p.rt + 1) * 33) * 16;
    p.grid = ((d->C + p.bm - 1) / p.bm) * (d->Wo / 32) * (d->Ho / p.rt) * d->N;
    const int nslabs = d->K / 16;
    p.splits = 1;
    if (p.grid < num_cu / 2) {
        p.splits = (num_cu + p.grid - 1) / p.grid;
        const int maxs = nslabs / 2 > 0 ? nslabs / 2 : 1;
        if (p.splits > maxs) p.splits = maxs;
    }
    p.slabs_per_split = (nslabs + p.splits - 1) / p.splits;
    p.splits = (nslabs + p.slabs_per_split - 1) / p.slabs_per_split;
    p.ok = true;
    return p;
}

int sp_launch_dgrad_s2(ghm_ctx* ctx, const SpPlan& pl, SpConvArgs a, const SpDgradS2Extra& x, const float* dy32, long dy32_nstride) {
    a.slabs_per_split = pl.slabs_per_split;
    a.zeros = (const u32x4*)ctx->zeros;
    a.partial = nullptr;
    const long plane = (long)a.N * (a.CH / 8) * a.Hin * a.Win;
    const int NP = pl.np;
    const size_t qbytes = dy32 ? align256((size_t)NP * plane * 16) : 0;
    const size_t pbytes = pl.splits > 1 ? (size_t)pl.splits * a.R * a.N * a.H * a.W * sizeof(float) : 0;
    if (qbytes + pbytes) {
        void* ws = nullptr;
        if (int e = ghm_scratch(ctx, qbytes + pbytes, &ws)) return e;
        if (dy32) {
            a.in_q = (const u32x4*)ws;
            a.in_q_nstride = (long)(a.CH / 8) * a.Hin * a.Win;
            a.in_q_pstride = plane;
            if (int e = sp_pack(ctx, dy32, dy32_nstride, a.N, a.CH, a.Hin * a.Win, ws, a.in_q_nstride, plane, NP)) return e;
        }
        if (pbytes) a.partial = (float*)((char*)ws + qbytes);
    }
    GHM_CHECK(!((x.dact_y || x.dact_q) && pl.splits > 1), "split-fp32 stride-2 data gradient + activation derivative needs a single-pass plan");
    GHM_CHECK(!(a.out_q && pl.splits > 1), "split-fp32 stride-2 data gradient: a q output needs a single-pass plan");
    GHM_CHECK(a.out || a.out_q, "split-fp32 stride-2 data gradient: no output");
    GHM_CHECK(!(a.accumulate && !a.out), "accumulate needs the fp32 output");
    GHM_CHECK(!a.out_q || (a.R % 8 == 0 && ((uintptr_t)a.out_q & 15) == 0), "q output: a multiple of 8 channels, 16-byte aligned");
    const dim3 g(pl.grid, pl.splits);
    if (NP == 3) {
        if (int e = sp_set_lds(sp_dgrad_s2_kernel<64, 2, 3>, pl.lds)) return e;
        hipLaunchKernelGGL((sp_dgrad_s2_kernel<64, 2, 3>), g, dim3(256), pl.lds, ctx->stream, a, x);
    } else {
        if (int e = sp_set_lds(sp_dgrad_s2_kernel<64, 2, 2>, pl.lds)) return e;
        hipLaunchKernelGGL((sp_dgrad_s2_kernel<64, 2, 2>), g, dim3(256), pl.lds, ctx->stream, a, x);
    }
    GHM_LAUNCH_CHECK();
    if (pl.splits > 1)
        return ghm_splitk_finish(ctx, a.partial, pl.splits, a.out, a.bias, a.N, a.R, a.H, a.W, a.out_nstride, a.act, a.alpha,
                                 a.accumulate);
    return 0;
}

bool sp_fwd_geom(const ghm_conv_desc* d) {
    return d->kh == d->kw && 2 * d->pad == d->kh - 1 && ((d->stride == 1 && d->Ho == d->H && d->Wo == d->W) ||
                                                          (d->stride == 2 && d->H == 2 * d->Ho && d->W == 2 * d->Wo));
}


static int sp_dgrad_s2(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const void* dyq, long dyq_ns, long dyq_ps,
                       const void* wqT, const float* bias, float* dx, int act, float alpha, int accumulate, const float* dact_y,
                       long dact_nstride, float dact_alpha, void* dxq, long dxq_ns, int np, const void* dact_q = nullptr,
                       long dact_q_ns = 0) {
    const SpPlan pl = sp_plan_dgrad_s2(d, ctx->num_cu, np);
    GHM_CHECK(pl.ok, "split-fp32 stride-2 data gradient: geometry not served");
    SpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)dyq; a.in_q_nstride = dyq_ns; a.in_q_pstride = dyq_ps;
    a.wq = (const u32x4*)wqT; a.bias = bias; a.out = dx; a.out_q = (uint2*)dxq; a.out_q_nstride = dxq_ns;
    a.N = d->N; a.CH = d->K; a.H = d->H; a.W = d->W; a.Hin = d->Ho; a.Win = d->Wo;
    a.R = d->C; a.Rpad = sp_rpad(d->C); a.wq_pstride = (long)sp_nblk(d->K) * 9 * a.Rpad;
    a.out_nstride = d->x_nstride; a.pad = d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    const SpDgradS2Extra x{dact_y, dact_nstride, dact_alpha, (const uint2*)dact_q, dact_q_ns};
    GHM_CHECK(!(dact_q && (dact_y || accumulate || d->C % 8)), "split-fp32 stride-2 data gradient: the slope from a q copy excludes "
              "the fp32 operand and accumulation, and needs whole 8-channel blocks");
    return sp_launch_dgrad_s2(ctx, pl, a, x, dyq ? nullptr : dy, d->y_nstride);
}

}  // namespace

// ---- C ABI (include/ghm.h) ----
extern "C" {

// kind 0: forward, 1: data gradient (stride 1, on the transposed pack)
int ghm_split_supported(const ghm_conv_desc* d, int32_t kind) {
    if (!d) return 0;
    if (kind == 0) return sp_fwd_geom(d) && sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ghm_plan_cus()).ok;
    if (kind == 1) {
        if (d->stride == 2) return sp_plan_dgrad_s2(d, ghm_plan_cus()).ok;
        return d->stride == 1 && sp_fwd_geom(d) && sp_plan(d->N, d->K, d->H, d->W, d->C, d->kh, 1, ghm_plan_cus()).ok;
    }
    if (kind == 2) return sp_wplan(d, ghm_plan_cus()).ok;
    return 0;
}

// may the product write the split q copy of its result from its own epilogue (a single-pass plan, whole q units)?
// kind 0 forward, 1 data gradient
int ghm_split_q_direct(const ghm_conv_desc* d, int32_t kind) {
    if (!d || !ghm_split_supported(d, kind)) return 0;
    if (kind == 0) return d->K % 8 == 0 && sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ghm_plan_cus()).splits == 1;
    if (kind == 1) {
        if (d->C % 8) return 0;
        if (d->stride == 2) return sp_plan_dgrad_s2(d, ghm_plan_cus()).splits == 1;
        return sp_plan(d->N, d->K, d->H, d->W, d->C, d->kh, 1, ghm_plan_cus()).splits == 1;
    }
    return 0;
}

// bytes of the workspace ghm_conv2d_wgrad_split needs (split partials, summed in fixed order)
int ghm_conv2d_wgrad_split_workspace(const ghm_conv_desc* d, size_t* bytes) {
    GHM_CHECK(d && bytes, "null argument");
    const SpWPlan v = sp_wplan(d, ghm_plan_cus());
    const size_t n = (size_t)d->C * d->kh * d->kw * d->K;
    *bytes = v.ok ? (size_t)v.ncols * v.splits_per_col * n * sizeof(float) : 16;
    return 0;
}

// dwp (+)= the weight gradient in the packed layout wp[c][tap][k], from the split q tensors of x and dy
int ghm_conv2d_wgrad_split(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, int64_t xq_pstride,
                           const void* dyq, int64_t dyq_nstride, int64_t dyq_pstride, float* dwp, void* workspace,
                           int32_t accumulate, int32_t pieces) {
    GHM_CHECK(ctx && d && xq && dyq && dwp, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_split_supported(d, 2), "ghm_conv2d_wgrad_split: geometry not served (ask ghm_split_supported)");
    GHM_CHECK((((uintptr_t)xq | (uintptr_t)dyq) & 15) == 0, "ghm_conv2d_wgrad_split: q tensors are 16-byte aligned");
    const SpWPlan v = sp_wplan(d, ghm_plan_cus(), pieces);
    return sp_launch_wgrad(ctx, d, v, xq, (long)xq_nstride, (long)xq_pstride, dyq, (long)dyq_nstride, (long)dyq_pstride, dwp,
                           workspace, accumulate);
}

// can the weight gradient of this conv -> activation -> MaxPool2D(2) layer take the pooled gradient in the sparse-instruction
// operand form (ghm_maxpool2_mask_bwd_compress_q)?
int ghm_conv2d_wgrad_pooled_split_supported(const ghm_conv_desc* d) {
    if (!d || !ghm_split_supported(d, 2)) return 0;
    const SpWPlan v = sp_wplan(d, ghm_plan_cus());
    return sp_wgrad_pooled_ok(d, v) ? 1 : 0;
}

// ghm_conv2d_wgrad_split for a layer whose dy is the max-pool backward of a pooled gradient: rows without a tied window row
// are contracted on the sparse matrix instruction from cq / cidx, the others (cflags) from the dense dyq -- same products
int ghm_conv2d_wgrad_pooled_split(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, int64_t xq_pstride,
                                  const void* dyq, int64_t dyq_nstride, int64_t dyq_pstride, const void* cq, int64_t cq_nstride,
                                  int64_t cq_pstride, const void* cidx, const int32_t* cflags, float* dwp, void* workspace,
                                  int32_t accumulate, int32_t pieces) {
    GHM_CHECK(ctx && d && xq && dyq && cq && cidx && cflags && dwp, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK((((uintptr_t)xq | (uintptr_t)dyq | (uintptr_t)cq | (uintptr_t)cidx) & 15) == 0, "ghm_conv2d_wgrad_pooled_split: 16-byte aligned tensors");
    const SpWPlan v = sp_wplan(d, ghm_plan_cus(), pieces);
    GHM_CHECK(ghm_split_supported(d, 2) && sp_wgrad_pooled_ok(d, v),
              "ghm_conv2d_wgrad_pooled_split: geometry not served (ask ghm_conv2d_wgrad_pooled_split_supported)");
    return sp_launch_wgrad_pooled(ctx, d, v, xq, (long)xq_nstride, (long)xq_pstride, dyq, (long)dyq_nstride, (long)dyq_pstride, cq,
                                  (long)cq_nstride, (long)cq_pstride, cidx, cflags, dwp, workspace, accumulate);
}

int ghm_split_weight_bytes(const ghm_conv_desc* d, int32_t transposed, size_t* bytes, int32_t pieces) {
    GHM_CHECK(d && bytes, "null argument");
    GHM_SP_PIECES_OK(pieces);
    const int red = transposed ? d->K : d->C, rows = transposed ? d->C : d->K;
    *bytes = (size_t)pieces * sp_nblk(red) * d->kh * d->kw * sp_rpad(rows) * 16;
    return 0;
}

int ghm_split_pack_weights(ghm_ctx* ctx, const ghm_conv_desc* d, const float* wp, void* wq, int32_t transposed, int32_t pieces) {
    GHM_CHECK(ctx && d && wp && wq, "null argument");
    GHM_SP_PIECES_OK(pieces);
    const int red = transposed ? d->K : d->C, rows = transposed ? d->C : d->K;
    const int T = d->kh * d->kw, nblk = sp_nblk(red), rpad = sp_rpad(rows);
    const long plane = (long)nblk * T * rpad;
    if (pieces == 3)
        hipLaunchKernelGGL(sp_pack_w_kernel<3>, dim3(ceil_div(plane, 256)), dim3(256), 0, ctx->stream, wp, (u32x4*)wq, red, T, rows,
                           nblk, rpad, transposed ? 1 : 0, plane);
    else
        hipLaunchKernelGGL(sp_pack_w_kernel<2>, dim3(ceil_div(plane, 256)), dim3(256), 0, ctx->stream, wp, (u32x4*)wq, red, T, rows,
                           nblk, rpad, transposed ? 1 : 0, plane);
    GHM_LAUNCH_CHECK();
    return 0;
}

// fp32 NCHW view -> split q tensor (``pieces`` planes of q_pstride units)
int ghm_split_pack(ghm_ctx* ctx, const float* x, int64_t x_nstride, int32_t N, int32_t C, int32_t HW, void* q, int64_t q_nstride,
                   int64_t q_pstride, int32_t pieces) {
    GHM_CHECK(ctx && x && q && C % 8 == 0, "ghm_split_pack: null argument or channels not a multiple of 8");
    GHM_SP_PIECES_OK(pieces);
    return sp_pack(ctx, x, x_nstride, N, C, HW, q, q_nstride, q_pstride, pieces);
}

int ghm_split_pack_batched(ghm_ctx* ctx, const void* table, int32_t n_items, int32_t total_blocks, int32_t pieces) {
    static_assert(sizeof(SpPackItem) == 48, "table layout is part of the ABI (see ghm.h)");
    GHM_CHECK(ctx && table, "null argument");
    GHM_SP_PIECES_OK(pieces);
    if (n_items <= 0 || total_blocks <= 0) return 0;
    if (pieces == 3)
        hipLaunchKernelGGL(sp_pack_w_batched_kernel<3>, dim3(total_blocks), dim3(256), 0, ctx->stream, (const SpPackItem*)table, n_items);
    else
        hipLaunchKernelGGL(sp_pack_w_batched_kernel<2>, dim3(total_blocks), dim3(256), 0, ctx->stream, (const SpPackItem*)table, n_items);
    GHM_LAUNCH_CHECK();
    return 0;
}

// Conv2DLayer -> {linear, relu, lrelu} -> MaxPool2DLayer(2) as one kernel (architectures/dcgan.py:42-47): the contract of
// ghm_conv2d_fwd_pool (pooled fp32 tensor or NULL, arg-max mask with the sign bit)
int ghm_split_pool_supported(const ghm_conv_desc* d, int32_t act) {
    if (!d || !(act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU)) return 0;
    if (d->stride != 1 || !sp_fwd_geom(d) || (d->Ho & 1) || (d->Wo & 1)) return 0;
    const SpPlan pl = sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, 1, ghm_plan_cus());
    return pl.ok && pl.splits == 1 && pl.tw == 32;
}

int ghm_conv2d_fwd_pool_split(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const void* xq, int64_t xq_nstride,
                              int64_t xq_pstride, const void* wq, const float* bias, float* pooled, void* pooledq,
                              int64_t pooledq_nstride, uint8_t* mask, int32_t act, float alpha, int32_t pieces) {
    GHM_CHECK(ctx && d && (x || xq) && wq && mask, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_split_pool_supported(d, act), "ghm_conv2d_fwd_pool_split: not served (ask ghm_split_pool_supported)");
    const SpPlan pl = sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, 1, ctx->num_cu, pieces);
    SpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)xq; a.in_q_nstride = xq_nstride; a.in_q_pstride = xq_pstride;
    a.wq = (const u32x4*)wq; a.bias = bias; a.pool_out = pooled; a.pool_mask = mask;
    a.out_q = (uint2*)pooledq; a.out_q_nstride = pooledq_nstride;
    a.N = d->N; a.CH = d->C; a.H = d->Ho; a.W = d->Wo; a.Hin = d->H; a.Win = d->W;
    a.R = d->K; a.Rpad = sp_rpad(d->K); a.wq_pstride = (long)sp_nblk(d->C) * d->kh * d->kw * a.Rpad;
    a.pad = d->pad; a.act = act; a.alpha = alpha;
    return sp_launch_conv(ctx, pl, a, d->kh, 1, xq ? nullptr : x, d->x_nstride, true);
}

// y = act(conv(x, W) + b) with fp32 operands and results; wq = ghm_split_pack_weights(transposed = 0).  xq != null: the
// input as a split q tensor (ghm_split_pack) instead of x.
int ghm_conv2d_fwd_split(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const void* xq, int64_t xq_nstride,
                         int64_t xq_pstride, const void* wq, const float* bias, float* y, void* yq, int64_t yq_nstride,
                         int32_t act, float alpha, int32_t accumulate, int32_t pieces) {
    GHM_CHECK(ctx && d && (x || xq) && wq && (y || yq), "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_split_supported(d, 0), "ghm_conv2d_fwd_split: geometry not served (ask ghm_split_supported)");
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    const SpPlan pl = sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ctx->num_cu, pieces);
    SpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)xq; a.in_q_nstride = xq_nstride; a.in_q_pstride = xq_pstride;
    a.wq = (const u32x4*)wq; a.bias = bias; a.out = y; a.out_q = (uint2*)yq; a.out_q_nstride = yq_nstride;
    a.N = d->N; a.CH = d->C; a.H = d->Ho; a.W = d->Wo; a.Hin = d->H; a.Win = d->W;
    a.R = d->K; a.Rpad = sp_rpad(d->K); a.wq_pstride = (long)sp_nblk(d->C) * d->kh * d->kw * a.Rpad;
    a.out_nstride = d->y_nstride; a.pad = d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    return sp_launch_conv(ctx, pl, a, d->kh, d->stride, xq ? nullptr : x, d->x_nstride, false);
}

// dx = act(conv^T(dy, W) + b) of a stride-1 'same' convolution; wqT = ghm_split_pack_weights(transposed = 1)
int ghm_conv2d_dgrad_split(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const void* dyq, int64_t dyq_nstride,
                           int64_t dyq_pstride, const void* wqT, const float* bias, float* dx, void* dxq, int64_t dxq_nstride,
                           int32_t act, float alpha, int32_t accumulate, int32_t pieces) {
    GHM_CHECK(ctx && d && (dy || dyq) && wqT && (dx || dxq), "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_split_supported(d, 1), "ghm_conv2d_dgrad_split: geometry not served (ask ghm_split_supported)");
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    if (d->stride == 2)
        return sp_dgrad_s2(ctx, d, dy, dyq, dyq_nstride, dyq_pstride, wqT, bias, dx, act, alpha, accumulate, nullptr, 0, 0.f, dxq,
                           (long)dxq_nstride, pieces);
    const SpPlan pl = sp_plan(d->N, d->K, d->H, d->W, d->C, d->kh, 1, ctx->num_cu, pieces);
    SpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)dyq; a.in_q_nstride = dyq_nstride; a.in_q_pstride = dyq_pstride;
    a.wq = (const u32x4*)wqT; a.bias = bias; a.out = dx; a.out_q = (uint2*)dxq; a.out_q_nstride = dxq_nstride;
    a.N = d->N; a.CH = d->K; a.H = d->H; a.W = d->W; a.Hin = d->H; a.Win = d->W;
    a.R = d->C; a.Rpad = sp_rpad(d->C); a.wq_pstride = (long)sp_nblk(d->K) * d->kh * d->kw * a.Rpad;
    a.out_nstride = d->x_nstride; a.pad = d->kh - 1 - d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    return sp_launch_conv(ctx, pl, a, d->kh, 1, dyq ? nullptr : dy, d->y_nstride, false);
}

// ---- the collapsed form of BilinearUpsample2DLayer(2) -> 3x3 conv (conv_bilinear.hip; d: the 3x3 'same' stride-1 descriptor on
// the coarse grid with d->K = 4 x the layer's filters, ordered (parity class, k)) with the structurally zero taps of the classes
// skipped: the same products in the same order as ghm_conv2d_{fwd, dgrad, wgrad}_split on the zero-padded collapsed weights.
// kind 0 forward, 1 data gradient, 2 weight gradient
int ghm_blconv_split_supported(const ghm_conv_desc* d, int32_t kind) {
    if (!d || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->K % 4 || GHM_OPT("GHM_BLCONV_NO_SKIP")) return 0;
    const int ck = d->K / 4;
    if (ck % 64) return 0;
    if (kind == 0) {
        const SpPlan pl = sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, 3, 1, ghm_plan_cus(), 3, true);
        return sp_fwd_geom(d) && pl.ok && pl.tw == 32 && pl.bm == 64;
    }
    if (kind == 1) {
        const SpPlan pl = sp_plan(d->N, d->K, d->H, d->W, d->C, 3, 1, ghm_plan_cus(), 3, GHM_OPT("GHM_BLCONV_DGRAD_BM64") != nullptr);
        return sp_fwd_geom(d) && pl.ok && pl.tw == 32;
    }
    if (kind == 2) {
        const SpWPlan v = sp_wplan(d, ghm_plan_cus());
        return v.ok && v.cht == 2 && v.ct == 2 && v.spx == 32;
    }
    return 0;
}

int ghm_blconv_fwd_split(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, int64_t xq_pstride, const void* wq,
                         const float* bias, float* y, int32_t pieces) {
    GHM_CHECK(ctx && d && xq && wq && y, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_blconv_split_supported(d, 0), "ghm_blconv_fwd_split: geometry not served (ask ghm_blconv_split_supported)");
    const int ck = d->K / 4;
    const SpPlan pl = sp_plan(d->N, d->C, d->Ho, d->Wo, d->K, 3, 1, ctx->num_cu, pieces, true);
    SpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)xq; a.in_q_nstride = xq_nstride; a.in_q_pstride = xq_pstride;
    a.wq = (const u32x4*)wq; a.bias = bias; a.out = y;
    a.N = d->N; a.CH = d->C; a.H = d->Ho; a.W = d->Wo; a.Hin = d->H; a.Win = d->W;
    a.R = d->K; a.Rpad = sp_rpad(d->K); a.wq_pstride = (long)sp_nblk(d->C) * 9 * a.Rpad;
    a.out_nstride = d->y_nstride; a.pad = 1;
    a.act = GHM_ACT_LINEAR; a.cls_k = ck;
    {   // dispatch groups of about two rounds of blocks: a power-of-two number of pixel tiles, a multiple of 8 where possible
        const int per = pl.grid / ((d->K + pl.bm - 1) / pl.bm), ntr = (d->K + pl.bm - 1) / pl.bm;
        int gs = 8;
        while (gs * 2 * ntr <= 2 * ctx->num_cu && per % (gs * 2) == 0) gs *= 2;
        if (per % gs) gs = per;
        if (const char* f = GHM_OPT("GHM_BLCONV_GROUP")) gs = (atoi(f) > 0 && per % atoi(f) == 0) ? atoi(f) : per;
        a.cls_group = gs;
    }
    // the classes' tiles cost 9 : 6 : 6 : 4: a launch of at most one block per CU ends with its 9-tap tiles while the CUs of
    // the 4-tap ones idle -- two halves of the contraction per tile give every CU a heavy and a light block
    SpPlan p2 = pl;
    const int nslabs = d->C / 16;
    if (p2.splits == 1 && p2.grid <= ctx->num_cu && nslabs >= 16 && !GHM_OPT("GHM_BLCONV_NO_SPLITK")) {
        p2.splits = 2;
        p2.slabs_per_split = (nslabs + 1) / 2;
    }
    return sp_launch_conv(ctx, p2, a, 3, 1, nullptr, 0, false, 1);
}

int ghm_blconv_dgrad_split(ghm_ctx* ctx, const ghm_conv_desc* d, const void* dyq, int64_t dyq_nstride, int64_t dyq_pstride,
                           const void* wqT, float* dx, int32_t accumulate, int32_t pieces) {
    GHM_CHECK(ctx && d && dyq && wqT && dx, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_blconv_split_supported(d, 1), "ghm_blconv_dgrad_split: geometry not served (ask ghm_blconv_split_supported)");
    const SpPlan pl = sp_plan(d->N, d->K, d->H, d->W, d->C, 3, 1, ctx->num_cu, pieces, GHM_OPT("GHM_BLCONV_DGRAD_BM64") != nullptr);
    SpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)dyq; a.in_q_nstride = dyq_nstride; a.in_q_pstride = dyq_pstride;
    a.wq = (const u32x4*)wqT; a.out = dx;
    a.N = d->N; a.CH = d->K; a.H = d->H; a.W = d->W; a.Hin = d->H; a.Win = d->W;
    a.R = d->C; a.Rpad = sp_rpad(d->C); a.wq_pstride = (long)sp_nblk(d->K) * 9 * a.Rpad;
    a.out_nstride = d->x_nstride; a.pad = 1;
    a.act = GHM_ACT_LINEAR; a.accumulate = accumulate; a.cls_k = d->K / 4;
    return sp_launch_conv(ctx, pl, a, 3, 1, nullptr, 0, false, 2);
}

int ghm_blconv_wgrad_split(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, int64_t xq_pstride,
                           const void* dyq, int64_t dyq_nstride, int64_t dyq_pstride, float* dwp, void* workspace,
                           int32_t accumulate, int32_t pieces) {
    GHM_CHECK(ctx && d && xq && dyq && dwp, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_blconv_split_supported(d, 2), "ghm_blconv_wgrad_split: geometry not served (ask ghm_blconv_split_supported)");
    const SpWPlan v = sp_wplan(d, ghm_plan_cus(), pieces);
    return sp_launch_wgrad(ctx, d, v, xq, (long)xq_nstride, (long)xq_pstride, dyq, (long)dyq_nstride, (long)dyq_pstride, dwp,
                           workspace, accumulate, d->K / 4);
}

// dx = conv^T(dy, W) * act'(y) of a 3x3 stride-2 convolution: the producer's relu / leaky-relu backward in the epilogue (the
// PatchGAN's conv -> LeakyRectify -> conv chains, p2p.py:285-286); 1 where served (single-pass plans)
int ghm_split_dgrad_dact_supported(const ghm_conv_desc* d) {
    if (!d || d->stride != 2) return 0;
    const SpPlan pl = sp_plan_dgrad_s2(d, ghm_plan_cus());
    return pl.ok && pl.splits == 1;
}

int ghm_conv2d_dgrad_dact_split(ghm_ctx* ctx, const ghm_conv_desc* d, const void* dyq, int64_t dyq_nstride, int64_t dyq_pstride,
                                const void* wqT, float* dx, void* dxq, int64_t dxq_nstride, const float* y, int64_t y_nstride,
                                int32_t act, float alpha, int32_t pieces) {
    GHM_CHECK(ctx && d && dyq && wqT && (dx || dxq) && y, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_split_dgrad_dact_supported(d), "ghm_conv2d_dgrad_dact_split: not served (ask ghm_split_dgrad_dact_supported)");
    GHM_CHECK(act == GHM_ACT_RELU || act == GHM_ACT_LRELU, "ghm_conv2d_dgrad_dact_split: relu / leaky relu");
    return sp_dgrad_s2(ctx, d, nullptr, dyq, (long)dyq_nstride, (long)dyq_pstride, wqT, nullptr, dx, GHM_ACT_LINEAR, 0.f, 0, y,
                       (long)y_nstride, act == GHM_ACT_RELU ? 0.f : alpha, dxq, (long)dxq_nstride, pieces);
}

// the same with the slope taken from the producer's q copy (first piece plane at ``yq``, ``yq_nstride`` units between samples):
// the producer's fp32 activation need not exist (engine.py: a conv -> LeakyRectify -> conv chain whose every reader takes q)
int ghm_conv2d_dgrad_dact_split_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* dyq, int64_t dyq_nstride, int64_t dyq_pstride,
                                  const void* wqT, float* dx, void* dxq, int64_t dxq_nstride, const void* yq, int64_t yq_nstride,
                                  int32_t act, float alpha, int32_t pieces) {
    GHM_CHECK(ctx && d && dyq && wqT && (dx || dxq) && yq, "null argument");
    GHM_SP_PIECES_OK(pieces);
    GHM_CHECK(ghm_split_dgrad_dact_supported(d) && d->C % 8 == 0,
              "ghm_conv2d_dgrad_dact_split_q: not served (ask ghm_split_dgrad_dact_supported; C % 8 == 0)");
    GHM_CHECK(act == GHM_ACT_RELU || act == GHM_ACT_LRELU, "ghm_conv2d_dgrad_dact_split_q: relu / leaky relu");
    return sp_dgrad_s2(ctx, d, nullptr, dyq, (long)dyq_nstride, (long)dyq_pstride, wqT, nullptr, dx, GHM_ACT_LINEAR, 0.f, 0, nullptr,
                       0, act == GHM_ACT_RELU ? 0.f : alpha, dxq, (long)dxq_nstride, pieces, yq, (long)yq_nstride);
}

}  // extern "C"
