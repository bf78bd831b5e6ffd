// Reduced-precision matrix-core convolutions for gfx950 (MI355X): bf16 or fp16 operands on
// v_mfma_f32_32x32x16_{bf16,f16} (2.5 PFLOP/s dense peak, 16x the fp32 MFMA rate), fp32 accumulate, fp32 storage.
//
// BASELINE configs 4 / 5 (bf16 / fp16 joint step).  The reference computes in floatX=float32 (experiment.5.sh:5), so
// this is an ADDITIONAL arithmetic mode of the same layers (Conv2DLayer: architectures/dcgan.py:22,42, p2p.py:20-21
// and their gradients), never the default.  Rule (restated in oracle/lp.py): both operands of every convolution
// product are rounded to bf16 / fp16 with round-to-nearest-even, products are exact, sums are fp32; activations,
// gradients, BatchNorm, losses, master weights and the optimiser stay fp32 in HBM.
//
// Why the activations stay fp32 in HBM: every convolution of the step is followed by fp32 element-wise work
// (BatchNorm statistics, pooling, losses) that wants the unrounded value, and with 9..25 taps of reuse per staged
// element the conversion costs a few VALU instructions per 100 MFMA cycles.  So the kernels read the same NCHW fp32
// tensors as the fp32 path and convert while staging to LDS:
//   * activations / output gradients (MFMA "B" operand, lanes along pixels): a thread gathers the 8 channels of one
//     pixel (8 dword loads, each coalesced over the pixels of the wave), packs them with v_cvt_pk_*_f32 into one
//     16-byte LDS unit -> the LDS patch is [8-channel block][row][col][8 ch], and a B fragment (8 consecutive k for
//     one pixel) is ONE conflict-free ds_read_b128 at a compile-time offset per tap, as in the fp32 patch kernel;
//   * weights (MFMA "A" operand): packed once per step into wq[c/8][tap][r][8 ch] (ghm_lp_pack_weights), so that a
//     slab's rows are contiguous 16-byte units and go global -> LDS by DMA (global_load_lds, 16 B per lane);
//   * weight gradient: the contraction runs over pixels, so 8 consecutive pixels of one channel are the 16-byte unit;
//     they ARE contiguous in NCHW, so a thread loads an aligned window of 16 (stride 2: 24) floats once and emits the
//     k shifted copies (one per filter column) the (channel, tap) rows need -- the im2col tile exists only in LDS.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int DT>
struct Lp;
template <>
struct Lp<GHM_DTYPE_BF16> {
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <>
struct Lp<GHM_DTYPE_F16> {
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));      // v_cvt_pk_f16_f32 (RNE)
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <int DT>
__device__ __forceinline__ u32x4 lp_pack8(const float* v) {
    u32x4 w;
    w.x = Lp<DT>::pack2(v[0], v[1]);
    w.y = Lp<DT>::pack2(v[2], v[3]);
    w.z = Lp<DT>::pack2(v[4], v[5]);
    w.w = Lp<DT>::pack2(v[6], v[7]);
    return w;
}

__device__ __forceinline__ int lp_xcd_remap(int bid, int nb) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 packed wp[c][tap][r]  ->  wq[c/8][tap][r (padded to 128)][8 ch]  (16-byte units)
//   transposed = 0: the forward operand (reduction over the conv's input channels c, rows = filters)
//   transposed = 1: the data-gradient operand wqT[k/8][T-1-tap][c (padded)][8 k] = wp[c][tap][k]
//                   (reduction over the filters k, rows = input channels, taps flipped: the adjoint convolution)
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void lp_pack_kernel(const float* __restrict__ wp, u32x4* __restrict__ wq, int C, int T,
                                                      int R, int nblk, int Rpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)nblk * T * Rpad) return;
    const int r = (int)(idx % Rpad);
    const long bt = idx / Rpad;
    const int tap = (int)(bt % T), cb = (int)(bt / T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        v[j] = (c < C && r < R) ? wp[((long)c * T + tap) * R + r] : 0.f;
    }
    wq[idx] = lp_pack8<DT>(v);
}

template <int DT>
__global__ __launch_bounds__(256) void lp_pack_t_kernel(const float* __restrict__ wp, u32x4* __restrict__ wq, int C, int T,
                                                        int K, int nblk, int Cpad) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)nblk * T * Cpad) return;
    const int c = (int)(idx % Cpad);
    const long bt = idx / Cpad;
    const int tapT = (int)(bt % T), kb = (int)(bt / T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb * 8 + j;
        v[j] = (c < C && k < K) ? wp[((long)c * T + (T - 1 - tapT)) * K + k] : 0.f;
    }
    wq[idx] = lp_pack8<DT>(v);
}

// every pack of a net in ONE launch: a device table of {wp, wq, red, T, rows, nblk, rpad, transposed, first block};
// a block (256 units) finds its item by a linear scan of the (<= 128-entry) table
struct LpPackItem {
    const float* wp;
    u32x4* wq;
    int red, T, rows, nblk, rpad, transposed, block_begin, pad_;
};

template <int DT>
__global__ __launch_bounds__(256) void lp_pack_batched_kernel(const LpPackItem* __restrict__ items, int n) {
    int li = 0;
    while (li + 1 < n && (int)blockIdx.x >= items[li + 1].block_begin) ++li;
    const LpPackItem it = items[li];
    const long idx = (long)(blockIdx.x - it.block_begin) * 256 + threadIdx.x;
    if (idx >= (long)it.nblk * it.T * it.rpad) return;
    const int r = (int)(idx % it.rpad);
    const long bt = idx / it.rpad;
    const int tap = (int)(bt % it.T), cb = (int)(bt / it.T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        const bool ok = c < it.red && r < it.rows;
        // plain: wp[c][tap][r] (reduction over the conv's input channels c, rows = filters)
        // transposed: element (k = c, tapT = tap, row = input channel r) = wp[r][T-1-tap][k]
        const long src = it.transposed ? ((long)r * it.T + (it.T - 1 - tap)) * it.red + c
                                       : ((long)c * it.T + tap) * it.rows + r;
        v[j] = ok ? it.wp[src] : 0.f;
    }
    it.wq[idx] = lp_pack8<DT>(v);
}

// ------------------------------------------------------------------------------------------------
// Forward-form convolution (forward pass; stride-1 data gradient on the transposed pack; 3x3 stride-2 forward).
// A block owns BM output channels x (RT rows x 32 columns) of one image.  K loop: slabs of 16 input channels; inside
// a slab one iteration per filter ROW a (KS k-steps, one per filter column b, each a 16-deep MFMA):
//   weights of (slab, a): [2 ch-blocks][KS][BM] 16-byte units by DMA, double-buffered, issued one iteration ahead;
//   input patch of the slab: [2 ch-blocks][PH][PW] units, gathered + converted through registers during the slab's
//   first iteration into the other patch buffer.
// Accumulators: TM x TN tiles of 32 x 32 (rows = channels, columns = 32 pixels of one row).
// ------------------------------------------------------------------------------------------------
struct LpConvArgs {
    const u32x4* in_q;     // the conv input (forward) / output gradient (data gradient) as a q tensor (include/ghm.h)
    long in_q_nstride;     // units (16 B = 8 channels of one pixel) between samples
    const u32x4* zeros;    // >= 16 bytes of zeros in HBM: the DMA source of padding pixels
    const u32x4* wq;
    const float* bias;
    float* out;            // fp32 NCHW output or null
    uint2* out_q;          // q-tensor output (half units: 4 channels of one pixel) or null
    long out_q_nstride;    // units between samples
    float* partial;
    int N, CH, H, W;       // OUTPUT grid
    int Hin, Win;
    int R, Rpad;
    long out_nstride;
    int pad, act;
    float alpha;
    int accumulate;
    int slabs_per_split;
    float* pool_out;            // POOL: dense [N, R, H/2, W/2] maximum of act(conv + bias) over 2x2 windows (or null) ...
    unsigned char* pool_mask;   // ... and the 4-bit arg-max mask of every window (bit 2*dr + dc; all ties set)
    // stride-2 data gradient only: out *= act'(dact_y): the backward of the producer's nonlinearity in this epilogue
    const float* dact_y;
    long dact_nstride;
    int dact;
    float dact_alpha;
    int debug;                  // tuning only (GHM_ABLATE bits: 1 no patch loads after the first slab, 2 no MFMAs, 4 no stores, 8 no weight loads after the first tile)
};

template <int DT>
__device__ __forceinline__ uint2 lp_pack4(float a, float b, float c, float d) {
    return make_uint2(Lp<DT>::pack2(a, b), Lp<DT>::pack2(c, d));
}

// fp32 NCHW view -> q tensor (the producer-side rounding as a pass of its own: test / fallback path; the step's
// producers write their q copy from their own epilogues).  One thread per 16-byte unit.
template <int DT>
__global__ __launch_bounds__(256) void q_pack_kernel(const float* __restrict__ x, long x_nstride, int N, int C8, int HW,
                                                     u32x4* __restrict__ q, long q_nstride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * C8 * HW) return;
    const int p = (int)(idx % HW);
    const long nc = idx / HW;
    const int cb = (int)(nc % C8), n = (int)(nc / C8);
    const float* g = x + (long)n * x_nstride + (long)cb * 8 * HW + p;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g[(long)j * HW];
    q[(long)n * q_nstride + (long)cb * HW + p] = lp_pack8<DT>(v);
}

// q tensor -> fp32 NCHW view (tests, and consumers that have no q path)
template <int DT>
__global__ __launch_bounds__(256) void q_unpack_kernel(const u32x4* __restrict__ q, long q_nstride, int N, int C8, int HW,
                                                       float* __restrict__ x, long x_nstride) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * C8 * HW) return;
    const int p = (int)(idx % HW);
    const long nc = idx / HW;
    const int cb = (int)(nc % C8), n = (int)(nc / C8);
    const u32x4 u = q[(long)n * q_nstride + (long)cb * HW + p];
    float* g = x + (long)n * x_nstride + (long)cb * 8 * HW + p;
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (DT == GHM_DTYPE_BF16) {
            g[(long)(2 * j) * HW] = __uint_as_float(w[j] << 16);
            g[(long)(2 * j + 1) * HW] = __uint_as_float(w[j] & 0xffff0000u);
        } else {
            const f16x2 h = __builtin_bit_cast(f16x2, w[j]);
            g[(long)(2 * j) * HW] = (float)h[0];
            g[(long)(2 * j + 1) * HW] = (float)h[1];
        }
    }
}

// the 2x2 maximum of act(v) over rows (v0, v1) of this lane and of the neighbour column's lane, and its arg-max mask
__device__ __forceinline__ float lp_pool2(float v0, float v1, unsigned& mask) {
    const float w0 = __shfl_xor(v0, 1, 64), w1 = __shfl_xor(v1, 1, 64);      // the neighbour column's two rows
    const float m = fmaxf(fmaxf(v0, v1), fmaxf(w0, w1));
    mask = (v0 == m ? 1u : 0u) | (w0 == m ? 2u : 0u) | (v1 == m ? 4u : 0u) | (w1 == m ? 8u : 0u) | (m > 0.f ? GHM_POOL_SIGN : 0u);
    return m;
}

template <int DT, int KS, int ST, int BM, int RT, int WM, int WN, int TW, bool POOL = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void lp_conv_kernel(const LpConvArgs a) {
    // TW = columns of the pixel tile (32, or 16 / 8 for narrow maps): the 32 pixel lanes of a fragment cover RPF = 32 / TW
    // consecutive rows of TW columns; the block's tile is (RT * RPF) rows x TW columns
    constexpr int T = KS * KS;
    constexpr int RPF = 32 / TW, ROWS = RT * RPF;
    constexpr int TM = BM / (WM * 32), TN = RT / WN;
    constexpr int PH = (ROWS - 1) * ST + KS, PW = (TW - 1) * ST + KS;
    constexpr int PUNITS = 2 * PH * PW;
    constexpr int WUNITS = 2 * KS * BM;
    constexpr int NW = WM * WN, NT = NW * 64;       // waves / threads per block: 4 waves, or 8 for the 16-row tile (the
                                                    // weight tile of a filter row is then staged once for twice the pixels)
    constexpr int NQ = (PUNITS + NT - 1) / NT;
    constexpr int NI = WUNITS / 64;                 // DMA wave-instructions per weight tile
    static_assert((NW == 4 || NW == 8) && TM >= 1 && TN >= 1, "4 or 8 waves");
    __shared__ __attribute__((aligned(16))) u32x4 smem[2 * WUNITS + 2 * PUNITS];
    u32x4* const Wl = smem;
    u32x4* const Pl = smem + 2 * WUNITS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int kg = lane >> 5, li = lane & 31;
    const int ntr = (a.R + BM - 1) / BM;
    const int tiles_x = a.W / TW, tiles_y = a.H / ROWS;
    int L = lp_xcd_remap(blockIdx.x, gridDim.x);
    const int r0 = (L % ntr) * BM;
    L /= ntr;
    const int tx = L % tiles_x;
    L /= tiles_x;
    const int ty = L % tiles_y;
    const int n = L / tiles_y;
    const int y0 = ty * ROWS, x0 = tx * TW;
    const int lx = li % TW, ly = li / TW;                 // this lane's pixel inside a fragment
    const int HW = a.H * a.W, HWin = a.Hin * a.Win;
    const int nslabs = a.CH / 16;
    const int s_begin = blockIdx.y * a.slabs_per_split;
    const int s_end = min(nslabs, s_begin + a.slabs_per_split);

    // ---- patch staging: item e = (channel block, patch row, patch column) is ONE 16-byte unit of the q tensor, brought
    // global -> LDS by DMA (no registers, no conversion); padding pixels come from a zero unit in HBM ----
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    int p_off[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * NT;
        const int cb = e / (PH * PW), rem = e - cb * (PH * PW);
        const int py = rem / PW, px = rem - py * PW;
        const int y = y0 * ST + py - a.pad, x = x0 * ST + px - a.pad;
        const bool ok = e < PUNITS && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        p_off[q] = ok ? cb * HWin + y * a.Win + x : -1;
    }
    const u32x4* ibase = a.in_q + (long)n * a.in_q_nstride + (long)s_begin * 2 * HWin;
    auto stage_patch = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q * NT + wave * 64 < PUNITS) {                   // wave-uniform: this wave has items in group q
                const u32x4* g = p_off[q] >= 0 ? ibase + p_off[q] : a.zeros;
                if (tid + q * NT < PUNITS)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Pl + buf * PUNITS + q * NT + wave * 64), 16, 0, 0);
            }
        }
        ibase += 2 * HWin;
    };
    // ---- weight DMA: tile (slab s, filter row fa) = chunks (cb, b) of BM consecutive 16-byte rows ----
    auto stage_weights = [&](int s, int fa, int buf) {
        const u32x4* src = a.wq + ((long)(2 * s) * T + fa * KS) * a.Rpad + r0 + lane;
#pragma unroll
        for (int w0 = 0; w0 < NI; w0 += NW) {
            const int w = w0 + wave;
            if (w < NI) {
                const int ci = w / (BM / 64), h = w - ci * (BM / 64);
                const int cb = ci / KS, b = ci - cb * KS;
                const u32x4* g = src + ((long)cb * T + b) * a.Rpad + h * 64;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Wl + buf * WUNITS + ci * BM + h * 64), 16, 0, 0);
            }
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
        stage_weights(s_begin, 0, 0);
        stage_patch(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int wlane = kg * KS * BM + wm * (BM / WM) + li;
    const int plane = kg * PH * PW + ((wn * TN * RPF + ly) * ST) * PW + lx * ST;
    int it = 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int pbuf = (s - s_begin) & 1;
        const bool next_slab = (s + 1) < s_end;
        for (int fa = 0; fa < KS; ++fa, ++it) {
            const int wbuf = it & 1;
            if (!(a.debug & 8)) {
                if (fa + 1 < KS)
                    stage_weights(s, fa + 1, wbuf ^ 1);
                else if (next_slab)
                    stage_weights(s + 1, 0, wbuf ^ 1);
            }
            if (fa == 0 && next_slab && !(a.debug & 1)) stage_patch(pbuf ^ 1);
            const u32x4* Wb = Wl + wbuf * WUNITS + wlane;
            const u32x4* Pb = Pl + pbuf * PUNITS + plane + fa * PW;
            if (!(a.debug & 2)) {
                // fragments are read one filter column ahead of their MFMAs (the compiler, left alone, sinks every read to
                // just before its use and the wave then eats the full LDS latency once per k-step: 54 % of the MFMA rate)
                u32x4 af[2][TM], bf[2][TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[0][i] = Wb[i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[0][j] = Pb[j * RPF * ST * PW];
                if (a.debug & 32) {          // tuning: no fragment reads after the first -- the bare MFMA + barrier loop
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[1][i] = af[0][i];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[1][j] = bf[0][j];
                }
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    if (b + 1 < KS && !(a.debug & 32)) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[(b + 1) & 1][i] = Wb[(b + 1) * BM + i * 32];
#pragma unroll
                        for (int j = 0; j < TN; ++j) bf[(b + 1) & 1][j] = Pb[j * RPF * ST * PW + b + 1];
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = Lp<DT>::mfma(af[b & 1][i], bf[b & 1][j], acc[i][j]);
                    if (b + 1 < KS) {                       // one of the next column's reads behind each of this column's MFMAs
#pragma unroll
                        for (int m_ = 0; m_ < TM + TN; ++m_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // keep the one-column lead
                }
            }
            if (!(a.debug & 16)) {           // (tuning bit 16: no waits / barriers in the loop)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }

    // ---- epilogue: lanes along pixels; row of element e of tile i: i*32 + (e&3) + 8*(e>>2) + 4*kg.  The four values
    // e = 4g .. 4g+3 of a lane are four consecutive channels of one pixel: half of a q unit (kg picks the half) ----
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
    float* const sb = reinterpret_cast<float*>(smem);          // bias through LDS (free after the last barrier)
    if (tid < BM) sb[tid] = (a.bias && r0 + tid < a.R) ? a.bias[r0 + tid] : 0.f;
    __syncthreads();
    const float* const lb = sb + wm * (BM / WM) + 4 * kg;
    if (a.debug & 4) return;
    if constexpr (POOL) {
        // 2x2 max-pool of act(conv + bias) in the epilogue: the row pair of a window is in one lane (TN consecutive
        // rows per wave), the column pair in lanes (2t, 2t+1); even lanes store the maximum and the arg-max mask
        static_assert(TN % 2 == 0 && ST == 1 && TW == 32, "pooled epilogue: row pairs inside a wave, 32-column tiles");
        const int Wp = a.W / 2;
        const long HWp = (long)(a.H / 2) * Wp;
        const long pix = (long)((y0 + wn * TN) / 2) * Wp + (x0 + li) / 2;
        const long base = ((long)n * a.R + rl) * HWp + pix;
        uint2* const qb = a.out_q ? a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(ru / 8) * HWp + pix) + kg : nullptr;
        const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
        const bool even = (li & 1) == 0;
#pragma unroll
        for (int j2 = 0; j2 < TN / 2; ++j2)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float m[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = 4 * g + t, k = i * 32 + t + 8 * g;
                        float v0 = acc[i][2 * j2][e] + lb[k], v1 = acc[i][2 * j2 + 1][e] + lb[k];
                        v0 = v0 > 0.f ? v0 : slope * v0;
                        v1 = v1 > 0.f ? v1 : slope * v1;
                        unsigned mk;
                        m[t] = lp_pool2(v0, v1, mk);
                        if (even && rl + k < a.R) {
                            const long o = base + (long)k * HWp + j2 * Wp;
                            if (a.pool_out) a.pool_out[o] = m[t];
                            a.pool_mask[o] = (unsigned char)mk;
                        }
                    }
                    if (qb && even && rl + i * 32 + 8 * g < a.R)
                        qb[2 * ((long)(i * 4 + g) * HWp + j2 * Wp)] = lp_pack4<DT>(m[0], m[1], m[2], m[3]);
                }
        return;
    }
    float* const ub = a.out ? a.out + (long)n * a.out_nstride + (long)ru * HW + (long)(y0 + wn * TN * RPF) * a.W + x0 : nullptr;
    const unsigned lo = 4u * kg * (unsigned)HW + ly * a.W + lx;
    uint2* const qb = a.out_q ? a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(ru / 8) * HW +
                                                 (long)(y0 + wn * TN * RPF + ly) * a.W + x0 + lx) + kg : nullptr;
    const bool pwl = a.act == GHM_ACT_LINEAR || a.act == GHM_ACT_RELU || a.act == GHM_ACT_LRELU;
    const float slope = a.act == GHM_ACT_LINEAR ? 1.f : (a.act == GHM_ACT_RELU ? 0.f : a.alpha);
    const bool full = r0 + BM <= a.R;
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
            if (a.accumulate) {           // (fp32 output present: checked by the host)
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
            if (qb) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (full || rl + i * 32 + 8 * g < a.R)
                        qb[2 * ((long)(i * 4 + g) * HW + j * RPF * a.W)] = lp_pack4<DT>(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Data gradient of a 3x3 / stride-2 / pad-1 convolution (U-Net encoder, PatchGAN; architectures/p2p.py:20-21).
// dx[c, 2i+pu, 2j+pv] only receives the taps whose parity matches (1, 2, 2 or 4 of the 9): the four output parity
// classes are four small stride-1 gathers over the SAME dy patch.  A block computes all four classes of BM channels x
// (RT x 32) class pixels; every k-step (16 dy channels, one tap) feeds exactly one class -> 9 k-steps per slab, no
// zero-insertion waste.  Operands: dy patch [2 ch-blocks][RT+1][33] units gathered + converted through registers,
// weights = the transposed pack wqT[k/8][8-tap][c][8 k] (all 9 taps of a slab by DMA).  The two column parities of a
// pixel pair live in the same lane, so the epilogue stores 8 contiguous bytes per lane.
// a.in = dy [N, CH=K, Hc, Wc], a.out = dx [N, R=C, 2Hc, 2Wc] (a.H, a.W = dx grid; a.Hin, a.Win = class grid).
// ------------------------------------------------------------------------------------------------
template <int DT, int BM, int RT>
__global__ __launch_bounds__(256, BM * RT >= 256 ? 2 : 3) void lp_dgrad_s2_kernel(const LpConvArgs a) {
    constexpr int T = 9, WM = 2, WN = 2;
    constexpr int TM = BM / (WM * 32), TN = RT / WN;
    constexpr int PH = RT + 1, PW = 33;
    constexpr int PUNITS = 2 * PH * PW;
    constexpr int WUNITS = 2 * T * BM;
    constexpr int NQ = (PUNITS + 255) / 256;
    constexpr int NI = WUNITS / 64;
    __shared__ __attribute__((aligned(16))) u32x4 smem[2 * WUNITS + 2 * PUNITS];
    u32x4* const Wl = smem;
    u32x4* const Pl = smem + 2 * WUNITS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int kg = lane >> 5, li = lane & 31;
    const int Hc = a.Hin, Wc = a.Win;
    const int ntr = (a.R + BM - 1) / BM;
    const int tiles_x = Wc / 32, tiles_y = Hc / RT;
    int L = lp_xcd_remap(blockIdx.x, gridDim.x);
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
        const int y = i0 + py, x = j0 + px;
        const bool ok = e < PUNITS && y < Hc && x < Wc;
        p_off[q] = ok ? cb * HWc + y * Wc + x : -1;
    }
    const u32x4* ibase = a.in_q + (long)n * a.in_q_nstride + (long)s_begin * 2 * HWc;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // the dy patch of a slab: one 16-byte q unit per (channel block, row, column), global -> LDS by DMA
    auto stage_patch = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q * 256 + wave * 64 < PUNITS) {
                const u32x4* g = p_off[q] >= 0 ? ibase + p_off[q] : a.zeros;
                if (tid + q * 256 < PUNITS)
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Pl + buf * PUNITS + q * 256 + wave * 64), 16, 0, 0);
            }
        }
        ibase += 2 * HWc;
    };
    auto stage_weights = [&](int s, int buf) {
        const u32x4* src = a.wq + (long)(2 * s) * T * a.Rpad + r0 + lane;
#pragma unroll
        for (int w0 = 0; w0 < NI; w0 += 4) {
            const int w = w0 + wave;
            if (w < NI) {
                const int ci = w / (BM / 64), h = w - ci * (BM / 64);        // ci = cb * T + tap
                const u32x4* g = src + (long)ci * a.Rpad + h * 64;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Wl + buf * WUNITS + ci * BM + h * 64), 16, 0, 0);
            }
        }
    };

    f32x16 acc[4][TM][TN];                          // [parity class pu*2+pv]
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[cl][i][j][e] = 0.f;

    if (s_begin < s_end) {
        stage_weights(s_begin, 0);
        stage_patch(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int wlane = kg * T * BM + wm * (BM / WM) + li;
    const int plane = kg * PH * PW + (wn * TN) * PW + li;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more) {
            stage_weights(s + 1, buf ^ 1);
            if (!(a.debug & 1)) stage_patch(buf ^ 1);
        }
        const u32x4* Wb = Wl + buf * WUNITS + wlane;
        const u32x4* Pb = Pl + buf * PUNITS + plane;
        // tw: tap index in the transposed pack = 8 - original tap (ta, tb); the tap feeds class (ta != 1, tb != 1) from
        // dy[i + (ta == 0)][j + (tb == 0)].  Fragments are read one tap ahead of their MFMAs.
        auto p_off_of = [](int tw) {
            const int ta = (8 - tw) / 3, tb = (8 - tw) % 3;
            return (ta == 0 ? 1 : 0) * PW + (tb == 0 ? 1 : 0);
        };
        // the nine taps only ever read the dy fragment of a class pixel at (row, column) offsets {0, 1} x {0, 1}: the
        // (TN + 1) x 2 distinct fragments of the wave's rows are read ONCE per slab and kept in registers (9 * TN LDS reads
        // became 2 * (TN + 1): the loop was LDS-read-bound, 27 fragment reads for 18 MFMAs in the <64, 4> shape)
        u32x4 af[2][TM], bf[TN + 1][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Wb[i * 32];
#pragma unroll
        for (int j = 0; j <= TN; ++j) {
            bf[j][0] = Pb[j * PW];
            bf[j][1] = Pb[j * PW + 1];
        }
        (void)p_off_of;
        if (!(a.debug & 2))
#pragma unroll
        for (int tw = 0; tw < T; ++tw) {
            if (tw + 1 < T) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(tw + 1) & 1][i] = Wb[(tw + 1) * BM + i * 32];
            }
            const int ta = (8 - tw) / 3, tb = (8 - tw) % 3;
            const int cl = (ta == 1 ? 0 : 2) + (tb == 1 ? 0 : 1);
            const int ro = ta == 0 ? 1 : 0, co = tb == 0 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[cl][i][j] = Lp<DT>::mfma(af[tw & 1][i], bf[j + ro][co], acc[cl][i][j]);
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch distance at one tap (register budget)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane = class pixel (row i0 + wn*TN + j, column j0 + li); its two column parities are adjacent:
    // one 8-byte store per (row parity, channel).  Wave-uniform row pointers + a 32-bit lane offset; bias through LDS.
    const long P = (long)a.N * HWx;
    const int ru = r0 + wm * (BM / WM), rl = ru + 4 * kg;
    float* const sb = reinterpret_cast<float*>(smem);
    if (tid < BM) sb[tid] = (a.bias && !a.partial && r0 + tid < a.R) ? a.bias[r0 + tid] : 0.f;
    __syncthreads();
    const float* const lb = sb + wm * (BM / WM) + 4 * kg;
    const long rstride = a.partial ? P : (long)HWx;
    const unsigned lo = 4u * kg * (unsigned)rstride + 2u * li;
    const bool plain = a.partial != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long rowpix = (long)(2 * (i0 + wn * TN + j)) * a.W + 2 * j0;
        float* const ub = a.partial ? a.partial + ((long)blockIdx.y * a.R + ru) * P + (long)n * HWx + rowpix
                                    : (a.out ? a.out + (long)n * a.out_nstride + (long)ru * HWx + rowpix : nullptr);
        // q output: unit (channel block ru/8 + 4i + g, pixel (2*(i0+..)+pu, 2*(j0+li) + {0, 1})), half kg
        uint2* const qrow = (a.out_q && !a.partial) ? a.out_q + 2 * ((long)n * a.out_q_nstride + (long)(ru / 8) * HWx + rowpix + 2 * li) + kg
                                                     : nullptr;
        float2 qv[4];
        const float* const yb = a.dact_y ? a.dact_y + (long)n * a.dact_nstride + (long)ru * HWx + rowpix : nullptr;
#pragma unroll
        for (int pu = 0; pu < 2; ++pu)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i * 32 + (e & 3) + 8 * (e >> 2);
                    if (rl + k >= a.R) continue;
                    float2 v = make_float2(acc[pu * 2 + 0][i][j][e], acc[pu * 2 + 1][i][j][e]);
                    if ((a.debug & 4) && v.x != 12345.f) continue;
                    float2* o = reinterpret_cast<float2*>(ub + (long)k * rstride + pu * a.W + lo);
                    if (!plain) {
                        v.x += lb[k]; v.y += lb[k];
                        if (a.accumulate) { const float2 old = *o; v.x += old.x; v.y += old.y; }      // (fp32 output present)
                        v.x = ghm_act(v.x, a.act, a.alpha);
                        v.y = ghm_act(v.y, a.act, a.alpha);
                        if (yb) {                       // relu / leaky relu: the slope of the producer
                            const float2 yy = *reinterpret_cast<const float2*>(yb + (long)k * HWx + pu * a.W + lo);
                            v.x *= yy.x > 0.f ? 1.f : a.dact_alpha;
                            v.y *= yy.y > 0.f ? 1.f : a.dact_alpha;
                        }
                    }
                    if (ub) *o = v;
                    if (qrow) {           // four consecutive channels (e = 4g .. 4g+3) of the lane's two pixels
                        qv[e & 3] = v;
                        if ((e & 3) == 3) {
                            uint2* qo = qrow + 2 * ((long)(i * 4 + (e >> 2)) * HWx + pu * a.W);
                            qo[0] = lp_pack4<DT>(qv[0].x, qv[1].x, qv[2].x, qv[3].x);
                            qo[2] = lp_pack4<DT>(qv[0].y, qv[1].y, qv[2].y, qv[3].y);
                        }
                    }
                }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dwp[(c, tap)][k] = sum over pixels of x[c, pixel + tap] * dy[k, pixel].
// GEMM rows = (channel, tap) of CBW = 128 / T channels, columns = BN filters, contraction = the SPX = 32 * NSEG
// pixels of one output-row segment per slab.  LDS tiles hold 8-pixel units: A [128 rows][SPX/8 (+1 pad)], B
// [BN][SPX/8 (+1 pad)]; the odd row stride makes the 32 rows of a fragment read hit distinct bank groups.
// ------------------------------------------------------------------------------------------------
struct LpWgradArgs {
    const float* x;
    const float* dy;
    float* out;
    int N, C, H, W;
    long x_nstride;
    int K, Ho, Wo;
    long y_nstride;
    int pad;
    int CT;
    int slabs_per_split;
    long split_stride;
    int accumulate;
};

template <int DT, int KS, int ST, int BN, int WM, int WN, int NSEG>
__global__ __launch_bounds__(256, 2) void lp_wgrad_kernel(const LpWgradArgs a) {
    constexpr int T = KS * KS;
    constexpr int CBW = 128 / T, ROWS = CBW * T, BM = 128;
    constexpr int SPX = 32 * NSEG, NCH = SPX / 8, LD = NCH + 1;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NA = CBW * KS * NCH;              // A staging items: (channel, filter row, pixel chunk)
    constexpr int AQ = (NA + 255) / 256;
    constexpr int WIN = ST == 1 ? 4 : 6;            // float4 per aligned input window (16 / 24 floats)
    constexpr int NB = BN * NCH;                    // B staging items: (filter, pixel chunk)
    constexpr int BQ = (NB + 255) / 256;
    constexpr int ASZ = BM * LD, BSZ = BN * LD;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) u32x4 smem[2 * (ASZ + BSZ)];
    u32x4* const Al = smem;
    u32x4* const Bl = smem + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kg = lane >> 5, li = lane & 31;
    // blocks that share a dy tile (same filter tile and pixel split, different row tiles) become neighbours on one XCD
    const int Lg = lp_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int bx = Lg % gridDim.x, by = (Lg / gridDim.x) % gridDim.y, bz = Lg / (gridDim.x * gridDim.y);
    const int c0 = bx * CBW, k0 = by * BN;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int segs_per_row = a.Wo / SPX;
    const int total_slabs = a.N * a.Ho * segs_per_row;
    const int s_begin = bz * a.slabs_per_split;
    const int s_end = min(s_begin + a.slabs_per_split, total_slabs);

    // rows ROWS..127 of the A tiles are never written: zero them once (both buffers)
    for (int e = tid; e < 2 * ASZ; e += 256) Al[e] = u32x4{0u, 0u, 0u, 0u};

    // ---- A staging constants: item (c, fa, q) -> window of input row (i*ST + fa - pad), columns from (j0+8q)*ST - 4
    int a_c[AQ], a_fa[AQ], a_q[AQ];
    bool a_valid[AQ];
#pragma unroll
    for (int p = 0; p < AQ; ++p) {
        const int e = tid + p * 256;
        const int c = e / (KS * NCH), rem = e - c * (KS * NCH);
        a_c[p] = c;
        a_fa[p] = rem / NCH;
        a_q[p] = rem - a_fa[p] * NCH;
        a_valid[p] = e < NA && (c0 + c) < a.C;
    }
    int sn, si, sj;
    {
        const int row = s_begin / segs_per_row;
        sj = (s_begin - row * segs_per_row) * SPX;
        sn = row / a.Ho;
        si = row - sn * a.Ho;
    }
    float aw[AQ][WIN * 4];
    float bw[BQ][8];
    auto load_slab = [&]() {
        const float* xb = a.x + (long)sn * a.x_nstride;
#pragma unroll
        for (int p = 0; p < AQ; ++p) {
            const int y = si * ST + a_fa[p] - a.pad;
            const int xs = (sj + 8 * a_q[p]) * ST - 4;
            const bool rok = a_valid[p] && (unsigned)y < (unsigned)a.H;
            const float* rowp = xb + (long)(c0 + (rok ? a_c[p] : 0)) * HW + (long)(rok ? y : 0) * a.W;
#pragma unroll
            for (int g = 0; g < WIN; ++g) {
                const int xg = xs + 4 * g;
                const bool ok = rok && xg >= 0 && xg < a.W;
                const float4 v = *reinterpret_cast<const float4*>(rowp + (ok ? xg : 0));
                aw[p][4 * g + 0] = ok ? v.x : 0.f;
                aw[p][4 * g + 1] = ok ? v.y : 0.f;
                aw[p][4 * g + 2] = ok ? v.z : 0.f;
                aw[p][4 * g + 3] = ok ? v.w : 0.f;
            }
        }
        const float* yb = a.dy + (long)sn * a.y_nstride + (long)si * a.Wo + sj;
#pragma unroll
        for (int p = 0; p < BQ; ++p) {
            const int e = tid + p * 256;
            const int f = e / NCH, q = e - f * NCH;
            const bool ok = e < NB && (k0 + f) < a.K;
            const float* g = yb + (long)(ok ? k0 + f : 0) * HoWo + 8 * q;
            const float4 v0 = *reinterpret_cast<const float4*>(g), v1 = *reinterpret_cast<const float4*>(g + 4);
            bw[p][0] = ok ? v0.x : 0.f; bw[p][1] = ok ? v0.y : 0.f; bw[p][2] = ok ? v0.z : 0.f; bw[p][3] = ok ? v0.w : 0.f;
            bw[p][4] = ok ? v1.x : 0.f; bw[p][5] = ok ? v1.y : 0.f; bw[p][6] = ok ? v1.z : 0.f; bw[p][7] = ok ? v1.w : 0.f;
        }
        sj += SPX;
        if (sj >= a.Wo) {
            sj = 0;
            if (++si >= a.Ho) {
                si = 0;
                ++sn;
            }
        }
        if (sn >= a.N) { sn = 0; si = 0; sj = 0; }          // past the end: keep addresses valid
    };
    auto store_slab = [&](int buf) {
        u32x4* Ab = Al + buf * ASZ;
        u32x4* Bb = Bl + buf * BSZ;
#pragma unroll
        for (int p = 0; p < AQ; ++p) {
            if (tid + p * 256 < NA) {
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = aw[p][4 + t * ST + b - KS / 2];     // pad == KS / 2 (checked by the host)
                    Ab[((a_c[p] * KS + a_fa[p]) * KS + b) * LD + a_q[p]] = lp_pack8<DT>(v);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < BQ; ++p) {
            const int e = tid + p * 256;
            if (e < NB) {
                const int f = e / NCH, q = e - f * NCH;
                Bb[f * LD + q] = lp_pack8<DT>(bw[p]);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    __syncthreads();                                    // the zero fill precedes the first tile stores
    if (s_begin < s_end) {
        load_slab();
        store_slab(0);
    }
    __syncthreads();
    const int alane = (wm * (BM / WM) + li) * LD + kg;
    const int blane = (wn * (BN / WN) + li) * LD + kg;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = (s + 1) < s_end;
        if (more) load_slab();
        const u32x4* Ab = Al + buf * ASZ + alane;
        const u32x4* Bb = Bl + buf * BSZ + blane;
#pragma unroll
        for (int ks = 0; ks < NCH / 2; ++ks) {
            u32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Ab[i * 32 * LD + ks * 2];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bb[j * 32 * LD + ks * 2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Lp<DT>::mfma(af[i], bf[j], acc[i][j]);
        }
        if (more) store_slab(buf ^ 1);
        __syncthreads();
    }

    float* ob = a.out + (long)bz * a.split_stride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = k0 + wn * (BN / WN) + j * 32 + li;
        if (col >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rt = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
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

// ------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 convolution (stride 1 or 2, pad 1) from q tensors: dwp[(c, tap)][k] = sum over pixels of
// x[c, pixel * ST + tap - 1] * dy[k, pixel].  The contraction runs over pixels while a q unit holds 8 CHANNELS of one
// pixel, so both MFMA operands are read from LDS with the transposing read ds_read_b64_tr_b16 (lane t of a 16-lane
// group supplies the address of 4 channels of pixel t / 4; it receives channel t of 4 consecutive pixels):
//   x image   [ring row][channel half][column parity (ST = 2)][pixel][32 channels]     (64 B per pixel)
//   dy image  [buffer][filter tile][pixel][32 filters]                                  (64 B per pixel)
// both staged global -> LDS by DMA as they lie in HBM (no conversion, no im2col expansion: a tap is an address offset).
// Block = 8 waves: CHT groups of 32 channels x CT tiles of 32 filters (CHT * CT = 8); a wave owns the nine taps of one
// (channel group, filter tile) = 9 accumulator tiles.  A block walks down ONE column strip of SPX output pixels, one
// output row per slab; the x rows live in a ring (each slab brings ST new rows), the dy strip is double-buffered.
// ------------------------------------------------------------------------------------------------
struct LpWgradQArgs {
    const u32x4* xq;
    long xq_ns;            // units between samples of x  [N][C/8][H][W]
    const u32x4* dyq;
    long dyq_ns;           // units between samples of dy [N][K/8][Ho][Wo]
    const u32x4* zeros;
    float* out;
    int N, C, H, W, K, Ho, Wo;
    int rows_per_split, splits_per_col;
    long split_stride;
    int accumulate;
    int debug;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 lp_tr_read8(const char* lds_lo, const char* lds_hi) {
    typedef s16x4 __attribute__((address_space(3))) * lp4_t;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4_t)lds_lo);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4_t)lds_hi);
    u32x4 r;
    r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    return r;
}
// The transposing read as inline assembly (round 5; see sp_tr_issue in conv_split.hip): behind the builtin above the compiler
// drains vmcnt in front of the first fragment read of every output row -- directly behind the DMA requests of the next row.
// The kernel waits for its own fragments instead (s_waitcnt lgkmcnt(0) tied to the fragment registers).
struct LpTrFrag {
    unsigned long long lo, hi;
};
__device__ __forceinline__ void lp_tr_issue(LpTrFrag& f, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:256" : "=&v"(f.lo), "=&v"(f.hi) : "v"(lds_addr));
}
__device__ __forceinline__ u32x4 lp_tr_bits(const LpTrFrag& f) {
    return u32x4{(unsigned)f.lo, (unsigned)(f.lo >> 32), (unsigned)f.hi, (unsigned)(f.hi >> 32)};
}
template <int N>
__device__ __forceinline__ void lp_tr_wait(LpTrFrag (&f)[N], LpTrFrag& g) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g.lo), "+v"(g.hi));
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].lo), "+v"(f[i].hi));      // (volatile: stays behind the wait)
}

template <int DT, int KS, int ST, int CHT, int CT, bool RS, int SPX>
__global__ __launch_bounds__(CHT * CT * (RS ? KS : 1) * 64, 1) void lp_wgrad_q_kernel(const LpWgradQArgs a) {
    // RS: the waves are additionally split over the KS filter ROWS (5x5: a wave owns the 5 taps of one row; 3x3: all 9 taps)
    static_assert((ST == 1 || (ST == 2 && KS == 3)) && (SPX == 64 || SPX == 32 || SPX == 16) && (KS == 3 || KS == 5), "variants");
    constexpr int T = KS * KS, PADK = KS / 2;
    constexpr int NWAVES = CHT * CT * (RS ? KS : 1);
    constexpr int TPW = RS ? KS : T;                  // accumulator tiles (taps) per wave
    constexpr int NPAR = ST;                          // column-parity planes of an x row
    constexpr int XPIX = ST == 1 ? SPX + KS - 1 : SPX + 1;   // pixels per plane
    constexpr int XCH = (XPIX + 15) / 16;             // 16-pixel DMA pieces per plane
    constexpr int PLB = XCH * 16 * 64;                // bytes per plane
    constexpr int ROWB = CHT * NPAR * PLB;            // bytes per ring row
    constexpr int NR = KS + ST;                       // ring rows: KS live + ST arriving
    constexpr int YTB = SPX * 64;                     // bytes per dy filter tile
    constexpr int YB = CT * YTB;                      // bytes per dy buffer
    constexpr int KSTEPS = SPX / 16;
    __shared__ __attribute__((aligned(16))) char smem[NR * ROWB + 2 * YB];
    char* const Xl = smem;
    char* const Yl = smem + NR * ROWB;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = RS ? wave / (CHT * CT) : 0;        // this wave's filter row (RS)
    const int wrem = wave % (CHT * CT);
    const int hh = wrem / CT, ww = wrem % CT;         // this wave's channel group / filter tile
    const int kg = lane >> 5, li = lane & 31;
    const int c0 = blockIdx.x * (32 * CHT), k0 = blockIdx.y * (32 * CT);
    const int strips = a.Wo / SPX;
    const int col = blockIdx.z / a.splits_per_col, sp = blockIdx.z - col * a.splits_per_col;
    const int n = col / strips, j0 = (col - n * strips) * SPX;
    const int i_begin = sp * a.rows_per_split, i_end = min(a.Ho, i_begin + a.rows_per_split);
    const int HWx = a.H * a.W, HWy = a.Ho * a.Wo;
    const int xs0 = j0 * ST - PADK;                   // image column of local column 0

    // DMA lane roles inside a 16-pixel piece: pixel pxi, channel block cb4 of the 32-channel group
    const int pxi = lane >> 2, cb4 = lane & 3;
    const u32x4* const xbase = a.xq + (long)n * a.xq_ns + (long)(c0 / 8 + cb4) * HWx;
    const u32x4* const ybase = a.dyq + (long)n * a.dyq_ns + (long)(k0 / 8 + cb4) * HWy;

    // bring image row y of x (all channel groups, parities) into ring slot (y + NR) % NR; rows outside the image are zero
    auto stage_xrow = [&](int y) {
        const int slot = (y + NR) % NR;
        const bool rok = (unsigned)y < (unsigned)a.H;
#pragma unroll
        for (int p0 = 0; p0 < CHT * NPAR * XCH; p0 += NWAVES) {
            const int p = p0 + wave;
            if (p < CHT * NPAR * XCH) {
                const int pl = p / XCH, ch = p - pl * XCH;        // plane = (channel group, parity)
                const int g = pl / NPAR, par = pl - g * NPAR;
                const int pp = ch * 16 + pxi;                        // pixel inside the plane
                const int x = xs0 + (ST == 2 ? 2 * pp + par : pp);
                const bool ok = rok && (unsigned)x < (unsigned)a.W;
                const u32x4* src = ok ? xbase + (long)(g * 4) * HWx + (long)y * a.W + x : a.zeros;
                if (pp < XPIX)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xl + slot * ROWB + pl * PLB + ch * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_dy = [&](int i, int buf) {
#pragma unroll
        for (int p0 = 0; p0 < CT * (SPX / 16); p0 += NWAVES) {
            const int p = p0 + wave;
            if (p < CT * (SPX / 16)) {
                const int ct = p / (SPX / 16), ch = p - ct * (SPX / 16);
                const u32x4* src = ybase + (long)(ct * 4) * HWy + (long)i * a.Wo + j0 + ch * 16 + pxi;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Yl + buf * YB + ct * YTB + ch * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (i_begin < i_end) {
        // cold start: the KS x rows of the first output row, its dy strip
#pragma unroll
        for (int fa = 0; fa < KS; ++fa) stage_xrow(i_begin * ST + fa - PADK);
        stage_dy(i_begin, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing (transposing reads): 16-lane group g16 -> rows (channels / filters) 16 * (g16 & 1) .. + 15;
    // lane t of the group supplies the address of pixel key = t >> 2, channel quad t & 3
    const int g16 = lane >> 4, lt = lane & 15;
    const int key = lt >> 2, quad = lt & 3;
    const int lane_off = (8 * kg + key) * 64 + (g16 & 1) * 32 + quad * 8;      // + 4 pixels (256 B) for the upper half
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)Xl;
    const unsigned yl0 = lds0 + (unsigned)(Yl - Xl) + ww * YTB + lane_off;
    const unsigned xl0 = lds0 + hh * NPAR * PLB + lane_off;

    for (int i = i_begin; i < i_end; ++i) {
        const int buf = (i - i_begin) & 1;
        if (i + 1 < i_end && !(a.debug & 1)) {
#pragma unroll
            for (int r = 0; r < ST; ++r) stage_xrow(i * ST - PADK + KS + r);        // the rows the next slab adds
            stage_dy(i + 1, buf ^ 1);
        }
        unsigned xr[RS ? 1 : KS];
#pragma unroll
        for (int fa = 0; fa < (RS ? 1 : KS); ++fa) xr[fa] = xl0 + ((i * ST + (RS ? wr : fa) - PADK + NR) % NR) * ROWB;
        const unsigned yb = yl0 + buf * YB;
        if (!(a.debug & 2)) {
            // the fragments of k-step ks + 1 are requested behind the wait for those of k-step ks, in front of its MFMAs
            constexpr int NF = RS ? KS : T;               // A fragments per k-step and wave
            LpTrFrag af[2][NF], bf[2];
            auto read_step = [&](int ks, int slot) {
                lp_tr_issue(bf[slot], yb + ks * 1024);
#pragma unroll
                for (int fa = 0; fa < (RS ? 1 : KS); ++fa)
#pragma unroll
                    for (int fb = 0; fb < KS; ++fb) {
                        // local column of pixel t' and tap column fb: t' * ST + fb  ->  (parity plane, pixel in plane)
                        const int par = ST == 2 ? (fb & 1) : 0;
                        const int shift = ST == 2 ? (fb >> 1) : fb;
                        lp_tr_issue(af[slot][fa * KS + fb], xr[fa] + par * PLB + (ks * 16 + shift) * 64);
                    }
            };
            read_step(0, 0);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                lp_tr_wait(af[ks & 1], bf[ks & 1]);
                if (ks + 1 < KSTEPS) read_step(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int t = 0; t < NF; ++t) acc[t] = Lp<DT>::mfma(lp_tr_bits(af[ks & 1][t]), lp_tr_bits(bf[ks & 1]), acc[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane = filter k0 + 32 ww + li; rows = channels c0 + 32 hh + (e & 3) + 8 (e >> 2) + 4 kg ----
    if (a.debug & 4) return;
    float* const ob = a.out + (long)blockIdx.z * a.split_stride;
    const int kcol = k0 + ww * 32 + li;
    if (kcol >= a.K) return;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = c0 + hh * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
            const int tap = RS ? wr * KS + t : t;
            if (c < a.C) {
                float* o = ob + ((long)c * T + tap) * a.K + kcol;
                float v = acc[t][e];
                if (a.accumulate) v += *o;
                *o = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct LpPlan {
    bool ok;
    int bm, rt, tw, splits, slabs_per_split, grid;
};

LpPlan lp_plan(int N, int CH, int H, int W, int R, int ks, int st, int num_cu, bool allow_rt16 = true) {
    LpPlan p;
    p.ok = false;
    if (GHM_OPT("GHM_NO_LP")) return p;
    if (!((ks == 3 && (st == 1 || st == 2)) || (ks == 5 && st == 1))) return p;
    // 64-filter-row blocks (1 x 4 waves, two blocks per CU) everywhere.  In isolation the 128-row tile (and its 16-row /
    // 8-wave form on the 5x5 layers) is the faster kernel on the big layers, but the train step runs three streams of
    // kernels on the one GPU, and there the smaller blocks win on every class -- measured in the step (tools/instep_sweep.sh,
    // bf16, ms per step, same box): 128-row rule 6.96; 64 on the 5x5 layers 6.80; on the 3x3 s1 layers 6.85; on the 3x3 s2
    // layers 6.95; on all 6.74.  GHM_LP_BM=128 restores the old rule (128 rows from 96 filters up) for sweeps.
    p.bm = 64;
    if (const char* f = GHM_OPT("GHM_LP_BM")) p.bm = (atoi(f) == 128 && R >= 96) ? 128 : 64;
    if (const char* f = GHM_OPT("GHM_LP_BM128_MASK")) {                               // tuning: 1 = 5x5, 2 = 3x3 s1, 4 = 3x3 s2
        const int cls = ks == 5 ? 1 : (st == 1 ? 2 : 4);
        if ((atoi(f) & cls) && R >= 96) p.bm = 128;
    }
    // pixel tile: 8 x 32 (stride 2: 4 x 32); narrow maps: 8 x 16 or 8 x 8 (fragments of 2 x 16 / 4 x 8 pixels)
    p.tw = W % 32 == 0 ? 32 : (W % 16 == 0 ? 16 : 8);
    p.rt = p.tw == 32 ? (st == 2 ? 4 : 8) : (p.tw == 16 ? 4 : 2);
    // 16-row tile (8 waves, one block per CU): the weight tile of every filter row is staged once for twice the pixels.
    // Measured (bf16, TFLOP/s, 8-row -> 16-row): 5x5 N8 C64 256^2 1172 -> 1248, N8 C128 128^2 1180 -> 1285; 3x3 layers
    // 1097 -> 1095, 1068 -> 1025, 941 -> 898 (their weight tiles are small: the 8-wave barrier costs more than the
    // staging saves) -> 5x5 only, from two rounds of 8-row blocks up.  GHM_LP_RT16 / GHM_LP_NO_RT16 force either.
    if (allow_rt16 && p.tw == 32 && st == 1 && p.bm == 128 && H % 16 == 0 && GHM_OPT("GHM_LP_NO_RT16") == nullptr) {
        const long g8 = (long)((R + p.bm - 1) / p.bm) * (W / 32) * (H / 8) * N;
        if ((ks == 5 && g8 >= 2L * num_cu) || GHM_OPT("GHM_LP_RT16")) p.rt = 16;
    }
    const int rows = p.rt * (32 / p.tw);
    if (R < 32 || (W % p.tw) || (H % rows) || (CH % 16) || CH < 16) return p;
    if (GHM_OPT("GHM_LP_NO_NARROW") && p.tw != 32) return p;
    const int ntr = (R + p.bm - 1) / p.bm;
    p.grid = ntr * (W / p.tw) * (H / rows) * N;
    const int nslabs = CH / 16;
    p.splits = 1;
    if (p.grid < num_cu) {      // a block per CU or more: the partial tensors cost more than the idle slots (tools/s2_sweep.sh)
        p.splits = (2 * num_cu + p.grid - 1) / p.grid;
        const int maxs = nslabs / 2 > 0 ? nslabs / 2 : 1;
        if (p.splits > maxs) p.splits = maxs;
    }
    if (const char* f = GHM_OPT("GHM_LP_SPLITS")) p.splits = atoi(f) < nslabs ? (atoi(f) > 0 ? atoi(f) : 1) : nslabs;
    p.slabs_per_split = (nslabs + p.splits - 1) / p.splits;
    p.splits = (nslabs + p.slabs_per_split - 1) / p.slabs_per_split;
    p.ok = true;
    return p;
}

static inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

template <int DT>
int lp_q_pack(ghm_ctx* ctx, const float* x, long x_nstride, int N, int C, int HW, void* q, long q_nstride) {
    const long total = (long)N * (C / 8) * HW;
    if (total == 0) return 0;
    hipLaunchKernelGGL((q_pack_kernel<DT>), dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, x, x_nstride, N, C / 8, HW,
                       (u32x4*)q, q_nstride);
    GHM_LAUNCH_CHECK();
    return 0;
}

// workspace of one launch: [q copy of an fp32 input (fp32-input entry points only)][split-K partials]
template <int DT>
int lp_prepare(ghm_ctx* ctx, LpConvArgs& a, int splits, const float* x32, long x32_nstride) {
    a.zeros = (const u32x4*)ctx->zeros;
    a.partial = nullptr;
    const size_t qbytes = x32 ? align256((size_t)a.N * (a.CH / 8) * a.Hin * a.Win * 16) : 0;
    const size_t pbytes = splits > 1 ? (size_t)splits * a.R * a.N * a.H * a.W * sizeof(float) : 0;
    if (qbytes + pbytes == 0) return 0;
    void* ws = nullptr;
    if (int e = ghm_scratch(ctx, qbytes + pbytes, &ws)) return e;
    if (x32) {
        a.in_q = (const u32x4*)ws;
        a.in_q_nstride = (long)(a.CH / 8) * a.Hin * a.Win;
        if (int e = lp_q_pack<DT>(ctx, x32, x32_nstride, a.N, a.CH, a.Hin * a.Win, ws, a.in_q_nstride)) return e;
    }
    if (pbytes) a.partial = (float*)((char*)ws + qbytes);
    return 0;
}

// after a split-K launch: the fixed-order reduction writes the fp32 output; a requested q output is packed from it
static bool lp_sm_finish_ok(int R, long pixels) { return R % 8 == 0 && pixels <= 8192 && GHM_OPT("GHM_NO_SM_FINISH") == nullptr; }

template <int DT>
int lp_finish_splits(ghm_ctx* ctx, const LpConvArgs& a, int splits, const SmBn* bn = nullptr) {
    if (lp_sm_finish_ok(a.R, (long)a.N * a.H * a.W))     // sum + bias + activation + fp32 / q outputs (+ BatchNorm) in ONE launch
        return sm_finish_launch(ctx, a.partial, splits, a.R, a.N, a.H * a.W, a.bias, a.out, a.out_nstride, a.accumulate, a.act,
                                a.alpha, a.out_q, a.out_q_nstride, DT, bn);
    GHM_CHECK(bn == nullptr, "split-K low-precision convolution: BatchNorm epilogue not served for this geometry");
    GHM_CHECK(a.out != nullptr, "split-K low-precision convolution needs an fp32 output (ask ghm_lp_q_direct)");
    if (int e = ghm_splitk_finish(ctx, a.partial, splits, a.out, a.bias, a.N, a.R, a.H, a.W, a.out_nstride, a.act, a.alpha,
                                  a.accumulate))
        return e;
    if (a.out_q) return lp_q_pack<DT>(ctx, a.out, a.out_nstride, a.N, a.R, a.H * a.W, a.out_q, a.out_q_nstride);
    return 0;
}

template <int DT>
int lp_launch_conv(ghm_ctx* ctx, const LpPlan& pl, LpConvArgs a, int ks, int st, const float* x32 = nullptr, long x32_nstride = 0,
                   const SmBn* bn = nullptr) {
    GHM_CHECK(bn == nullptr || pl.splits > 1, "lp_launch_conv: the BatchNorm epilogue belongs to the split-K form");
    a.slabs_per_split = pl.slabs_per_split;
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    GHM_CHECK(!(a.accumulate && !a.out), "accumulate needs the fp32 output");
    GHM_CHECK(!a.out_q || a.R % 8 == 0, "q output needs a multiple of 8 channels");
    if (int e = lp_prepare<DT>(ctx, a, pl.splits, x32, x32_nstride)) return e;
    const dim3 g(pl.grid, pl.splits);
#define GHM_LP_CASE(KS_, ST_, BM_, RT_, WM_, WN_, TW_)                                                              \
    if (ks == KS_ && st == ST_ && pl.bm == BM_ && pl.tw == TW_ && pl.rt == RT_) {                                   \
        hipLaunchKernelGGL((lp_conv_kernel<DT, KS_, ST_, BM_, RT_, WM_, WN_, TW_>), g, dim3(WM_ * WN_ * 64), 0, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                                         \
    } else
    GHM_LP_CASE(5, 1, 128, 16, 2, 4, 32)
    GHM_LP_CASE(3, 1, 128, 16, 2, 4, 32)
    GHM_LP_CASE(5, 1, 128, 8, 2, 2, 32)
    GHM_LP_CASE(5, 1, 64, 8, 1, 4, 32)
    GHM_LP_CASE(3, 1, 128, 8, 2, 2, 32)
    GHM_LP_CASE(3, 1, 64, 8, 1, 4, 32)
    GHM_LP_CASE(3, 2, 128, 4, 2, 2, 32)
    GHM_LP_CASE(3, 2, 64, 4, 1, 4, 32)
    GHM_LP_CASE(5, 1, 128, 4, 2, 2, 16)
    GHM_LP_CASE(5, 1, 64, 4, 1, 4, 16)
    GHM_LP_CASE(3, 1, 128, 4, 2, 2, 16)
    GHM_LP_CASE(3, 1, 64, 4, 1, 4, 16)
    GHM_LP_CASE(3, 2, 128, 4, 2, 2, 16)
    GHM_LP_CASE(3, 2, 64, 4, 1, 4, 16)
    GHM_LP_CASE(5, 1, 128, 2, 2, 2, 8)
    GHM_LP_CASE(5, 1, 64, 2, 2, 2, 8)
    GHM_LP_CASE(3, 1, 128, 2, 2, 2, 8)
    GHM_LP_CASE(3, 1, 64, 2, 2, 2, 8)
    GHM_LP_CASE(3, 2, 128, 2, 2, 2, 8)
    GHM_LP_CASE(3, 2, 64, 2, 2, 2, 8) {
        ghm_set_error("no lp_conv variant for k=%d s=%d bm=%d tw=%d rt=%d", ks, st, pl.bm, pl.tw, pl.rt);
        return -3;
    }
#undef GHM_LP_CASE
    if (pl.splits > 1) return lp_finish_splits<DT>(ctx, a, pl.splits, bn);
    return 0;
}

// 3x3 stride-2 pad-1 data gradient on the transposed pack
LpPlan lp_plan_dgrad_s2(const ghm_conv_desc* d, int num_cu) {
    LpPlan p;
    p.ok = false;
    if (GHM_OPT("GHM_NO_LP") || GHM_OPT("GHM_NO_LP_DGRAD_S2")) return p;
    if (!(d->stride == 2 && d->kh == 3 && d->kw == 3 && d->pad == 1 && d->H == 2 * d->Ho && d->W == 2 * d->Wo)) return p;
    if (d->Wo % 32 || d->K % 16 || d->K < 16 || d->C < 32 || (d->x_nstride & 1) || ((d->H * d->W) & 1)) return p;
    // tile = (channels, class rows): 0 = 128 x 2, 1 = 64 x 4, 2 = 64 x 2.  Measured (tools/s2_dgrad_quick.sh, bf16, warm):
    //   N8 C64 256^2 K128: 0.127 / 0.063 / 0.051 ms   N8 C128 128^2 K256: 0.047 / 0.040 / 0.042   N8 C256 64^2 K512: 0.053 / 0.045 / 0.032
    // K is short on these layers (8 - 32 slabs): a block is mostly prologue + epilogue, and the smallest tile (3 blocks per
    // CU, twice the blocks) hides them best; the larger tiles stay for the sweep (GHM_LP_DGRAD_S2_TILE)
    static const int tiles[3][2] = {{128, 2}, {64, 4}, {64, 2}};
    static const int order[3] = {2, 1, 0};
    int forced = -1;
    if (const char* f = GHM_OPT("GHM_LP_DGRAD_S2_TILE")) forced = *f ? atoi(f) : -1;
    bool found = false;
    for (int o = 0; o < 3 && !found; ++o) {
        const int t = forced >= 0 ? (forced > 2 ? 2 : forced) : order[o];
        if (tiles[t][0] == 128 && d->C < 96 && forced < 0) continue;
        if (d->Ho % tiles[t][1] == 0) {
            p.bm = tiles[t][0]; p.rt = tiles[t][1];
            p.grid = ((d->C + p.bm - 1) / p.bm) * (d->Wo / 32) * (d->Ho / p.rt) * d->N;
            found = true;
        }
        if (forced >= 0) break;
    }
    if (!found) return p;
    const int nslabs = d->K / 16;
    p.splits = 1;
    if (p.grid < num_cu) {
        p.splits = (2 * num_cu + p.grid - 1) / p.grid;
        const int maxs = nslabs / 2 > 0 ? nslabs / 2 : 1;
        if (p.splits > maxs) p.splits = maxs;
    }
    if (const char* f = GHM_OPT("GHM_LP_DGRAD_S2_SPLITS")) p.splits = atoi(f) < 1 ? 1 : atoi(f);
    p.slabs_per_split = (nslabs + p.splits - 1) / p.splits;
    p.splits = (nslabs + p.slabs_per_split - 1) / p.slabs_per_split;
    p.ok = true;
    return p;
}

template <int DT>
int lp_launch_dgrad_s2(ghm_ctx* ctx, const LpPlan& pl, LpConvArgs a, const float* dy32 = nullptr, long dy32_nstride = 0) {
    a.slabs_per_split = pl.slabs_per_split;
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    GHM_CHECK(!(a.accumulate && !a.out), "accumulate needs the fp32 output");
    GHM_CHECK(!a.out_q || a.R % 8 == 0, "q output needs a multiple of 8 channels");
    if (int e = lp_prepare<DT>(ctx, a, pl.splits, dy32, dy32_nstride)) return e;
    const dim3 g(pl.grid, pl.splits);
    if (pl.bm == 128)
        hipLaunchKernelGGL((lp_dgrad_s2_kernel<DT, 128, 2>), g, dim3(256), 0, ctx->stream, a);
    else if (pl.rt == 4)
        hipLaunchKernelGGL((lp_dgrad_s2_kernel<DT, 64, 4>), g, dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((lp_dgrad_s2_kernel<DT, 64, 2>), g, dim3(256), 0, ctx->stream, a);
    GHM_LAUNCH_CHECK();
    if (pl.splits > 1) return lp_finish_splits<DT>(ctx, a, pl.splits);
    return 0;
}

struct LpWPlan {
    bool ok;
    int bn, nseg, splits, slabs_per_split, row_tiles;
};

LpWPlan lp_wplan(const ghm_conv_desc* d, int num_cu) {
    LpWPlan v;
    v.ok = false;
    if (GHM_OPT("GHM_NO_LP") || GHM_OPT("GHM_NO_LP_WGRAD")) return v;
    const bool k_ok = (d->kh == 3 && d->kw == 3 && (d->stride == 1 || d->stride == 2)) ||
                      (d->kh == 5 && d->kw == 5 && d->stride == 1);
    if (!k_ok || d->pad != d->kh / 2 || d->Wo % 32 || d->W % 4 || d->K < 32 || d->C * d->kh * d->kw < 96) return v;
    if (d->x_nstride % 4 || d->y_nstride % 4 || (d->Ho * d->Wo) % 4 || (d->H * d->W) % 4) return v;
    if (d->Ho != (d->H + d->stride - 1) / d->stride) return v;
    const int T = d->kh * d->kw;
    v.bn = d->K >= 96 ? 128 : 64;
    v.nseg = (d->Wo % 64 == 0 && GHM_OPT("GHM_LP_WGRAD_SEG1") == nullptr) ? 2 : 1;
    v.row_tiles = ceil_div(d->C, 128 / T);
    const long tiles = (long)v.row_tiles * ceil_div(d->K, v.bn);
    const long slabs = (long)d->N * d->Ho * (d->Wo / (32 * v.nseg));
    long want = (2L * num_cu) / tiles;
    if (const char* f = GHM_OPT("GHM_LP_WGRAD_SPLITS")) want = atol(f);
    const long max_by_work = slabs / 4 > 0 ? slabs / 4 : 1;         // at least 4 slabs per split
    long S = want < max_by_work ? want : max_by_work;
    if (S < 1) S = 1;
    if (S > 1024) S = 1024;
    v.slabs_per_split = (int)((slabs + S - 1) / S);
    v.splits = (int)((slabs + v.slabs_per_split - 1) / v.slabs_per_split);
    v.ok = true;
    return v;
}

template <int DT>
int lp_launch_wgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const LpWPlan& v, const float* x, const float* dy, float* dwp,
                    void* workspace, int accumulate) {
    LpWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.x_nstride = d->x_nstride;
    a.K = d->K; a.Ho = d->Ho; a.Wo = d->Wo; a.y_nstride = d->y_nstride;
    a.pad = d->pad;
    a.CT = d->C * d->kh * d->kw;
    a.slabs_per_split = v.slabs_per_split;
    const long n = (long)a.CT * a.K;
    if (v.splits > 1) {
        GHM_CHECK(workspace != nullptr, "lp wgrad needs a workspace for %d splits", v.splits);
        a.out = (float*)workspace; a.split_stride = n; a.accumulate = 0;
    } else {
        a.out = dwp; a.split_stride = 0; a.accumulate = accumulate;
    }
    const dim3 grid(v.row_tiles, ceil_div(a.K, v.bn), v.splits);
#define GHM_LPW_CASE(KS_, ST_, BN_, WM_, WN_)                                                                        \
    if (d->kh == KS_ && d->stride == ST_ && v.bn == BN_) {                                                           \
        if (v.nseg == 2)                                                                                             \
            hipLaunchKernelGGL((lp_wgrad_kernel<DT, KS_, ST_, BN_, WM_, WN_, 2>), grid, dim3(256), 0, ctx->stream, a); \
        else                                                                                                         \
            hipLaunchKernelGGL((lp_wgrad_kernel<DT, KS_, ST_, BN_, WM_, WN_, 1>), grid, dim3(256), 0, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                                          \
    } else
    GHM_LPW_CASE(5, 1, 128, 2, 2)
    GHM_LPW_CASE(5, 1, 64, 4, 1)
    GHM_LPW_CASE(3, 1, 128, 2, 2)
    GHM_LPW_CASE(3, 1, 64, 4, 1)
    GHM_LPW_CASE(3, 2, 128, 2, 2)
    GHM_LPW_CASE(3, 2, 64, 4, 1) {
        ghm_set_error("no lp_wgrad variant for k=%d s=%d bn=%d", d->kh, d->stride, v.bn);
        return -3;
    }
#undef GHM_LPW_CASE
    if (v.splits > 1) return ghm_reduce_splits(ctx, (const float*)workspace, v.splits, n, n, dwp, accumulate);
    return 0;
}

// ---- weight gradient from q tensors (lp_wgrad_q_kernel): 3x3, pad 1, stride 1 / 2 ----
struct LpWQPlan {
    bool ok;
    int cht, ct, spx, splits_per_col, rows_per_split, ncols;
};

LpWQPlan lp_wqplan(const ghm_conv_desc* d, int num_cu) {
    LpWQPlan v;
    v.ok = false;
    if (GHM_OPT("GHM_NO_LP") || GHM_OPT("GHM_NO_LP_WGRAD") || GHM_OPT("GHM_NO_LP_WGRAD_Q")) return v;
    const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2);
    const bool k5 = d->kh == 5 && d->kw == 5 && d->pad == 2 && d->stride == 1;
    if (!k3 && !k5) return v;
    if (d->Ho != (d->H + d->stride - 1) / d->stride || d->Wo != (d->W + d->stride - 1) / d->stride) return v;
    // 16-wide maps: strips of 16 pixels, ONE 16-pixel k-step per output row and barrier -- latency-bound (a slab waits for
    // its DMA about as long as it computes), and still several times the fp32 kernels these layers ran on before
    const bool narrow = d->Wo % 32 != 0;
    if (d->Wo % 16 || d->C % 8 || d->K % 8 || (narrow && GHM_OPT("GHM_LP_NO_NARROW_WGRAD"))) return v;
    if (k5) {                   // 10 waves: 5 filter rows x 2 filter tiles of one 32-channel group
        if (d->K % 64 || d->C % 32) return v;
        v.cht = 1; v.ct = 2;
    } else if (d->K % 128 == 0 && d->C % 64 == 0) { v.cht = 2; v.ct = 4; }
    else if (d->stride == 1 && d->K % 64 == 0 && d->C % 128 == 0) { v.cht = 4; v.ct = 2; }
    else return v;
    v.spx = d->Wo % 64 == 0 ? 64 : (narrow ? 16 : 32);
    if (const char* f = GHM_OPT("GHM_LP_WGRAD_SPX"))                                  // tuning: narrower strips (less LDS per block)
        if (atoi(f) == 32 && v.spx == 64) v.spx = 32;
    v.ncols = d->N * (d->Wo / v.spx);
    const long tiles = (long)(d->C / (32 * v.cht)) * (d->K / (32 * v.ct)) * v.ncols;
    // ONE round of resident blocks (a block per CU: 8-10 waves, up to 135 KB of LDS): every further split writes and
    // re-reads another partial copy of the weight gradient -- measured (N4 C256 256^2 K64): 256 blocks 797 TFLOP/s, 512
    // blocks 601, 128 blocks 507; the same optimum at one round for the 128^2, 64^2 and 5x5 layers
    long S = num_cu / tiles;
    if (const char* f = GHM_OPT("GHM_LP_WGRAD_ROUNDS")) S = (long)(atof(f) * num_cu / tiles);      // tuning
    if (const char* f = GHM_OPT("GHM_LP_WGRAD_SPLITS")) S = atol(f);
    const long minrows = d->kh == 5 ? 8 : 4;                 // rows per split >= the cold start's KS x rows (and then some)
    const long max_by_work = d->Ho / minrows > 0 ? d->Ho / minrows : 1;
    if (S > max_by_work) S = max_by_work;
    if (S < 1) S = 1;
    v.rows_per_split = (int)((d->Ho + S - 1) / S);
    v.splits_per_col = (d->Ho + v.rows_per_split - 1) / v.rows_per_split;
    v.ok = true;
    return v;
}

template <int DT>
int lp_launch_wgrad_q(ghm_ctx* ctx, const ghm_conv_desc* d, const LpWQPlan& v, const void* xq, long xq_ns, const void* dyq,
                      long dyq_ns, float* dwp, void* workspace, int accumulate) {
    LpWgradQArgs a;
    memset(&a, 0, sizeof(a));
    a.xq = (const u32x4*)xq; a.xq_ns = xq_ns; a.dyq = (const u32x4*)dyq; a.dyq_ns = dyq_ns;
    a.zeros = (const u32x4*)ctx->zeros;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.Ho = d->Ho; a.Wo = d->Wo;
    a.rows_per_split = v.rows_per_split; a.splits_per_col = v.splits_per_col;
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    const long n = (long)d->C * d->kh * d->kw * d->K;
    const int splits = v.ncols * v.splits_per_col;
    if (splits > 1) {
        GHM_CHECK(workspace != nullptr, "lp wgrad (q) needs a workspace for %d splits", splits);
        a.out = (float*)workspace; a.split_stride = n; a.accumulate = 0;
    } else {
        a.out = dwp; a.split_stride = 0; a.accumulate = accumulate;
    }
    const dim3 grid(d->C / (32 * v.cht), d->K / (32 * v.ct), splits);
#define GHM_LPWQ_CASE(KS_, ST_, CHT_, CT_, RS_, SPX_)                                                                    \
    if (d->kh == KS_ && d->stride == ST_ && v.cht == CHT_ && v.ct == CT_ && v.spx == SPX_) {                            \
        hipLaunchKernelGGL((lp_wgrad_q_kernel<DT, KS_, ST_, CHT_, CT_, RS_, SPX_>), grid,                               \
                           dim3(CHT_ * CT_ * (RS_ ? KS_ : 1) * 64), 0, ctx->stream, a);                                \
        GHM_LAUNCH_CHECK();                                                                                            \
    } else
    GHM_LPWQ_CASE(3, 1, 2, 4, false, 64)
    GHM_LPWQ_CASE(3, 1, 2, 4, false, 32)
    GHM_LPWQ_CASE(3, 1, 4, 2, false, 64)
    GHM_LPWQ_CASE(3, 1, 4, 2, false, 32)
    GHM_LPWQ_CASE(3, 2, 2, 4, false, 64)
    GHM_LPWQ_CASE(3, 2, 2, 4, false, 32)
    GHM_LPWQ_CASE(5, 1, 1, 2, true, 64)
    GHM_LPWQ_CASE(3, 1, 2, 4, false, 16)
    GHM_LPWQ_CASE(3, 1, 4, 2, false, 16)
    GHM_LPWQ_CASE(3, 2, 2, 4, false, 16)
    GHM_LPWQ_CASE(5, 1, 1, 2, true, 16)
    GHM_LPWQ_CASE(5, 1, 1, 2, true, 32) {
        ghm_set_error("no lp_wgrad_q variant for s=%d cht=%d ct=%d spx=%d", d->stride, v.cht, v.ct, v.spx);
        return -3;
    }
#undef GHM_LPWQ_CASE
    if (splits > 1) return ghm_reduce_splits(ctx, (const float*)workspace, splits, n, n, dwp, accumulate);
    return 0;
}

// forward-form geometry of the three uses: kind 0 forward, kind 1 data gradient (stride 1 only), kind 2 weight gradient
bool lp_fwd_geom(const ghm_conv_desc* d) {
    return d->kh == d->kw && ((d->stride == 1 && d->Ho == d->H && d->Wo == d->W) ||
                              (d->stride == 2 && d->Ho * 2 == d->H && d->Wo * 2 == d->W && d->pad == 1));
}

int rpad128(int r) { return (r + 127) / 128 * 128; }

}  // namespace

bool lp_dgrad_s2_single_pass(const ghm_conv_desc* d, int dtype) {
    if (dtype != GHM_DTYPE_BF16 && dtype != GHM_DTYPE_F16) return false;
    const LpPlan pl = lp_plan_dgrad_s2(d, ghm_plan_cus());
    return pl.ok && pl.splits == 1 && (d->x_nstride & 1) == 0;
}

// the operands / results of one low-precision product: each tensor as fp32 NCHW and / or as a q tensor (include/ghm.h)
struct LpIO {
    const float* in32;          // fp32 input (packed to a q copy in the workspace first) -- or --
    const void* inq;            // the input as a q tensor
    long inq_ns;
    float* out32;               // fp32 output or null
    void* outq;                 // q output or null
    long outq_ns;
};

static int lp_check_io(const LpIO& io, const char* who) {
    GHM_CHECK((io.in32 != nullptr) != (io.inq != nullptr), "%s: exactly one of the fp32 / q input", who);
    GHM_CHECK(io.out32 != nullptr || io.outq != nullptr, "%s: no output", who);
    GHM_CHECK((((uintptr_t)io.inq | (uintptr_t)io.outq) & 15) == 0, "%s: q tensors are 16-byte aligned", who);
    return 0;
}

static int lp_dgrad_s2_io(ghm_ctx* ctx, const ghm_conv_desc* d, const LpIO& io, const void* wqT, const float* bias, int act,
                          float alpha, int accumulate, const float* dact_y, long dact_nstride, int dact, float dact_alpha,
                          int dtype, bool single_pass) {
    const LpPlan pl = lp_plan_dgrad_s2(d, ctx->num_cu);
    GHM_CHECK(pl.ok && (!single_pass || pl.splits == 1), "lp stride-2 data gradient: plan not served");
    LpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)io.inq; a.in_q_nstride = io.inq_ns;
    a.wq = (const u32x4*)wqT; a.bias = bias; a.out = io.out32; a.out_q = (uint2*)io.outq; a.out_q_nstride = io.outq_ns;
    a.N = d->N; a.CH = d->K; a.H = d->H; a.W = d->W; a.Hin = d->Ho; a.Win = d->Wo;
    a.R = d->C; a.Rpad = rpad128(d->C); a.out_nstride = d->x_nstride; a.pad = d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    a.dact_y = dact_y; a.dact_nstride = dact_nstride; a.dact = dact; a.dact_alpha = dact_alpha;
    return dtype == GHM_DTYPE_BF16 ? lp_launch_dgrad_s2<GHM_DTYPE_BF16>(ctx, pl, a, io.in32, d->y_nstride)
                                   : lp_launch_dgrad_s2<GHM_DTYPE_F16>(ctx, pl, a, io.in32, d->y_nstride);
}

int lp_dgrad_s2_dact(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const void* wqT, float* dx, const float* dact_y,
                     long dact_nstride, int dact, float dact_alpha, int dtype) {
    const LpIO io{dy, nullptr, 0, dx, nullptr, 0};
    return lp_dgrad_s2_io(ctx, d, io, wqT, nullptr, GHM_ACT_LINEAR, 0.f, 0, dact_y, dact_nstride, dact, dact_alpha, dtype, true);
}

static bool lp_pool_act_ok(int act) { return act == GHM_ACT_LINEAR || act == GHM_ACT_RELU || act == GHM_ACT_LRELU; }

bool lp_conv_pool_supported(const ghm_conv_desc* d, int act, int dtype) {
    if (dtype != GHM_DTYPE_BF16 && dtype != GHM_DTYPE_F16) return false;
    if (!(lp_pool_act_ok(act) && d->stride == 1 && d->kh == d->kw && d->Ho == d->H && d->Wo == d->W && d->H % 2 == 0)) return false;
    const LpPlan pl = lp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, 1, ghm_plan_cus(), false);
    return pl.ok && pl.tw == 32 && pl.splits == 1 && GHM_OPT("GHM_NO_POOL_FUSE") == nullptr;
}

static int lp_fwd_pool_io(ghm_ctx* ctx, const ghm_conv_desc* d, const LpIO& io, const void* wq, const float* bias,
                          unsigned char* mask, int act, float alpha, int dtype) {
    const LpPlan pl = lp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, 1, ctx->num_cu, false);      // (pooled epilogue: 8-row tile)
    GHM_CHECK(pl.ok && pl.tw == 32 && pl.splits == 1, "lp_conv_fwd_pool: geometry not served");
    GHM_CHECK(!io.outq || d->K % 8 == 0, "q output needs a multiple of 8 channels");
    LpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)io.inq; a.in_q_nstride = io.inq_ns;
    a.wq = (const u32x4*)wq; a.bias = bias; a.out = nullptr;
    a.out_q = (uint2*)io.outq; a.out_q_nstride = io.outq_ns;
    a.N = d->N; a.CH = d->C; a.H = d->Ho; a.W = d->Wo; a.Hin = d->H; a.Win = d->W;
    a.R = d->K; a.Rpad = rpad128(d->K); a.out_nstride = 0; a.pad = d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = 0;
    a.slabs_per_split = pl.slabs_per_split;
    a.pool_out = io.out32; a.pool_mask = mask;
    if (const char* f = GHM_OPT("GHM_ABLATE")) a.debug = atoi(f);
    if (int e = dtype == GHM_DTYPE_BF16 ? lp_prepare<GHM_DTYPE_BF16>(ctx, a, 1, io.in32, d->x_nstride)
                                        : lp_prepare<GHM_DTYPE_F16>(ctx, a, 1, io.in32, d->x_nstride))
        return e;
    const dim3 g(pl.grid, 1);
#define GHM_LPP_CASE(DT_, KS_, BM_, WM_, WN_)                                                                          \
    if (dtype == DT_ && d->kh == KS_ && pl.bm == BM_) {                                                               \
        hipLaunchKernelGGL((lp_conv_kernel<DT_, KS_, 1, BM_, 8, WM_, WN_, 32, true>), g, dim3(256), 0, ctx->stream, a); \
        GHM_LAUNCH_CHECK();                                                                                           \
        return 0;                                                                                                     \
    }
    GHM_LPP_CASE(GHM_DTYPE_BF16, 5, 128, 2, 2)
    GHM_LPP_CASE(GHM_DTYPE_BF16, 5, 64, 1, 4)
    GHM_LPP_CASE(GHM_DTYPE_BF16, 3, 128, 2, 2)
    GHM_LPP_CASE(GHM_DTYPE_BF16, 3, 64, 1, 4)
    GHM_LPP_CASE(GHM_DTYPE_F16, 5, 128, 2, 2)
    GHM_LPP_CASE(GHM_DTYPE_F16, 5, 64, 1, 4)
    GHM_LPP_CASE(GHM_DTYPE_F16, 3, 128, 2, 2)
    GHM_LPP_CASE(GHM_DTYPE_F16, 3, 64, 1, 4)
#undef GHM_LPP_CASE
    ghm_set_error("no pooled lp_conv variant for k=%d bm=%d", d->kh, pl.bm);
    return -3;
}

int lp_conv_fwd_pool(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const void* wq, const float* bias, float* pooled,
                     unsigned char* mask, int act, float alpha, int dtype) {
    const LpIO io{x, nullptr, 0, pooled, nullptr, 0};
    return lp_fwd_pool_io(ctx, d, io, wq, bias, mask, act, alpha, dtype);
}

static int lp_fwd_io(ghm_ctx* ctx, const ghm_conv_desc* d, const LpIO& io, const void* wq, const float* bias, int act,
                     float alpha, int accumulate, int dtype, const SmBn* bn = nullptr) {
    GHM_CHECK(ghm_lp_supported(d, 0, dtype), "low-precision forward convolution: geometry / dtype not served by the "
              "matrix-core kernels (ask ghm_lp_supported first)");
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    if (sm_use(d, 0, dtype))            // small maps: gather GEMM + finishing kernel (conv_small.hip)
        return sm_conv(ctx, d, 0, io.inq, io.inq_ns, io.in32, wq, bias, io.out32, d->y_nstride, io.outq, io.outq_ns, act, alpha,
                       accumulate, dtype, nullptr);
    const LpPlan pl = lp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ctx->num_cu);
    LpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)io.inq; a.in_q_nstride = io.inq_ns;
    a.wq = (const u32x4*)wq; a.bias = bias; a.out = io.out32; a.out_q = (uint2*)io.outq; a.out_q_nstride = io.outq_ns;
    a.N = d->N; a.CH = d->C; a.H = d->Ho; a.W = d->Wo; a.Hin = d->H; a.Win = d->W;
    a.R = d->K; a.Rpad = rpad128(d->K); a.out_nstride = d->y_nstride; a.pad = d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    return dtype == GHM_DTYPE_BF16 ? lp_launch_conv<GHM_DTYPE_BF16>(ctx, pl, a, d->kh, d->stride, io.in32, d->x_nstride, bn)
                                   : lp_launch_conv<GHM_DTYPE_F16>(ctx, pl, a, d->kh, d->stride, io.in32, d->x_nstride, bn);
}

bool lp_fwd_splitk_bn_ok(const ghm_conv_desc* d, int dtype) {
    if ((dtype != GHM_DTYPE_BF16 && dtype != GHM_DTYPE_F16) || !lp_fwd_geom(d)) return false;
    const LpPlan pl = lp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ghm_plan_cus());
    // (the BatchNorm form of the finishing kernel walks the whole map of its 8 channels in one block: 16 x 16 maps, not more)
    return pl.ok && pl.splits > 1 && (long)d->N * d->Ho * d->Wo <= 1024 && lp_sm_finish_ok(d->K, (long)d->N * d->Ho * d->Wo);
}

int lp_fwd_splitk_bn(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, long xq_ns, const void* wq, const float* bias,
                     float* conv_out, void* yq, long yq_ns, int act, float alpha, int dtype, const SmBn* bn) {
    GHM_CHECK(lp_fwd_splitk_bn_ok(d, dtype), "lp_fwd_splitk_bn: not served");
    const LpIO io{nullptr, xq, xq_ns, conv_out, yq, yq_ns};
    return lp_fwd_io(ctx, d, io, wq, bias, act, alpha, 0, dtype, bn);
}

static int lp_dgrad_io(ghm_ctx* ctx, const ghm_conv_desc* d, const LpIO& io, const void* wqT, const float* bias, int act,
                       float alpha, int accumulate, int dtype) {
    GHM_CHECK(ghm_lp_supported(d, 1, dtype), "low-precision data gradient: geometry / dtype not served (ask ghm_lp_supported)");
    GHM_CHECK(!(accumulate && act != GHM_ACT_LINEAR), "accumulate needs a linear epilogue");
    if (sm_use(d, 1, dtype))
        return sm_conv(ctx, d, 1, io.inq, io.inq_ns, io.in32, wqT, bias, io.out32, d->x_nstride, io.outq, io.outq_ns, act, alpha,
                       accumulate, dtype, nullptr);
    if (d->stride == 2)
        return lp_dgrad_s2_io(ctx, d, io, wqT, bias, act, alpha, accumulate, nullptr, 0, 0, 0.f, dtype, false);
    // the data gradient of a stride-1 conv is the forward conv K -> C with flipped taps (folded into the
    // transposed pack) and padding k-1-pad
    const LpPlan pl = lp_plan(d->N, d->K, d->H, d->W, d->C, d->kh, 1, ctx->num_cu);
    LpConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in_q = (const u32x4*)io.inq; a.in_q_nstride = io.inq_ns;
    a.wq = (const u32x4*)wqT; a.bias = bias; a.out = io.out32; a.out_q = (uint2*)io.outq; a.out_q_nstride = io.outq_ns;
    a.N = d->N; a.CH = d->K; a.H = d->H; a.W = d->W; a.Hin = d->H; a.Win = d->W;
    a.R = d->C; a.Rpad = rpad128(d->C); a.out_nstride = d->x_nstride; a.pad = d->kh - 1 - d->pad;
    a.act = act; a.alpha = alpha; a.accumulate = accumulate;
    return dtype == GHM_DTYPE_BF16 ? lp_launch_conv<GHM_DTYPE_BF16>(ctx, pl, a, d->kh, 1, io.in32, d->y_nstride)
                                   : lp_launch_conv<GHM_DTYPE_F16>(ctx, pl, a, d->kh, 1, io.in32, d->y_nstride);
}

extern "C" {

int ghm_lp_supported(const ghm_conv_desc* d, int32_t kind, int32_t dtype) {
    if (dtype != GHM_DTYPE_BF16 && dtype != GHM_DTYPE_F16) return 0;
    if ((kind == 0 || kind == 1) && sm_use(d, kind, dtype)) return 1;
    if (kind == 0) return lp_fwd_geom(d) && lp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ghm_plan_cus()).ok;
    if (kind == 1) {
        if (d->stride == 2) return lp_plan_dgrad_s2(d, ghm_plan_cus()).ok;
        return d->stride == 1 && d->kh == d->kw && d->Ho == d->H && d->Wo == d->W &&
               lp_plan(d->N, d->K, d->H, d->W, d->C, d->kh, 1, ghm_plan_cus()).ok;
    }
    if (kind == 2) return lp_wplan(d, ghm_plan_cus()).ok;
    return 0;
}

int ghm_lp_weight_bytes(const ghm_conv_desc* d, int32_t transposed, size_t* bytes) {
    const int T = d->kh * d->kw;
    const int red = transposed ? d->K : d->C, rows = transposed ? d->C : d->K;
    *bytes = (size_t)((red + 15) / 16 * 2) * T * rpad128(rows) * 16;
    return 0;
}

int ghm_lp_pack_weights(ghm_ctx* ctx, const ghm_conv_desc* d, const float* wp, void* wq, int32_t dtype,
                        int32_t transposed) {
    GHM_CHECK(dtype == GHM_DTYPE_BF16 || dtype == GHM_DTYPE_F16, "ghm_lp_pack_weights: dtype %d", dtype);
    const int T = d->kh * d->kw;
    const int red = transposed ? d->K : d->C, rows = transposed ? d->C : d->K;
    const int nblk = (red + 15) / 16 * 2, rp = rpad128(rows);
    const long total = (long)nblk * T * rp;
    const dim3 g(ceil_div(total, 256)), b(256);
    if (!transposed) {
        if (dtype == GHM_DTYPE_BF16)
            hipLaunchKernelGGL((lp_pack_kernel<GHM_DTYPE_BF16>), g, b, 0, ctx->stream, wp, (u32x4*)wq, d->C, T, d->K, nblk, rp);
        else
            hipLaunchKernelGGL((lp_pack_kernel<GHM_DTYPE_F16>), g, b, 0, ctx->stream, wp, (u32x4*)wq, d->C, T, d->K, nblk, rp);
    } else {
        if (dtype == GHM_DTYPE_BF16)
            hipLaunchKernelGGL((lp_pack_t_kernel<GHM_DTYPE_BF16>), g, b, 0, ctx->stream, wp, (u32x4*)wq, d->C, T, d->K, nblk, rp);
        else
            hipLaunchKernelGGL((lp_pack_t_kernel<GHM_DTYPE_F16>), g, b, 0, ctx->stream, wp, (u32x4*)wq, d->C, T, d->K, nblk, rp);
    }
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_lp_pack_batched(ghm_ctx* ctx, const void* table, int32_t n_items, int32_t total_blocks, int32_t dtype) {
    static_assert(sizeof(LpPackItem) == 48, "table layout is part of the ABI (see ghm.h)");
    GHM_CHECK(dtype == GHM_DTYPE_BF16 || dtype == GHM_DTYPE_F16, "ghm_lp_pack_batched: dtype %d", dtype);
    if (n_items <= 0 || total_blocks <= 0) return 0;
    if (dtype == GHM_DTYPE_BF16)
        hipLaunchKernelGGL((lp_pack_batched_kernel<GHM_DTYPE_BF16>), dim3(total_blocks), dim3(256), 0, ctx->stream,
                           (const LpPackItem*)table, n_items);
    else
        hipLaunchKernelGGL((lp_pack_batched_kernel<GHM_DTYPE_F16>), dim3(total_blocks), dim3(256), 0, ctx->stream,
                           (const LpPackItem*)table, n_items);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_q_pack(ghm_ctx* ctx, const float* x, int64_t x_nstride, int32_t N, int32_t C, int32_t HW, void* q,
               int64_t q_nstride, int32_t dtype) {
    GHM_CHECK((dtype == GHM_DTYPE_BF16 || dtype == GHM_DTYPE_F16) && C % 8 == 0 && ((uintptr_t)q & 15) == 0,
              "ghm_q_pack: dtype bf16 / f16, channels %% 8 == 0, 16-byte aligned q");
    return dtype == GHM_DTYPE_BF16 ? lp_q_pack<GHM_DTYPE_BF16>(ctx, x, x_nstride, N, C, HW, q, q_nstride)
                                   : lp_q_pack<GHM_DTYPE_F16>(ctx, x, x_nstride, N, C, HW, q, q_nstride);
}

int ghm_q_unpack(ghm_ctx* ctx, const void* q, int64_t q_nstride, int32_t N, int32_t C, int32_t HW, float* x,
                 int64_t x_nstride, int32_t dtype) {
    GHM_CHECK((dtype == GHM_DTYPE_BF16 || dtype == GHM_DTYPE_F16) && C % 8 == 0 && ((uintptr_t)q & 15) == 0,
              "ghm_q_unpack: dtype bf16 / f16, channels %% 8 == 0, 16-byte aligned q");
    const long total = (long)N * (C / 8) * HW;
    if (total == 0) return 0;
    if (dtype == GHM_DTYPE_BF16)
        hipLaunchKernelGGL((q_unpack_kernel<GHM_DTYPE_BF16>), dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream,
                           (const u32x4*)q, (long)q_nstride, N, C / 8, HW, x, (long)x_nstride);
    else
        hipLaunchKernelGGL((q_unpack_kernel<GHM_DTYPE_F16>), dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream,
                           (const u32x4*)q, (long)q_nstride, N, C / 8, HW, x, (long)x_nstride);
    GHM_LAUNCH_CHECK();
    return 0;
}

int ghm_lp_q_direct(const ghm_conv_desc* d, int32_t kind, int32_t dtype) {
    // does the kernel of this product write its q output (and may it skip the fp32 one)?  Not in the split-K form.
    if (!ghm_lp_supported(d, kind, dtype)) return 0;
    if ((kind == 0 || kind == 1) && sm_use(d, kind, dtype)) return 1;      // the finishing kernel writes the q copy
    if (kind == 0)
        return (d->K % 8 == 0) && (lp_plan(d->N, d->C, d->Ho, d->Wo, d->K, d->kh, d->stride, ghm_plan_cus()).splits == 1 ||
                                   lp_sm_finish_ok(d->K, (long)d->N * d->Ho * d->Wo));
    if (kind == 1) {
        if (d->C % 8) return 0;
        if (d->stride == 2) return lp_plan_dgrad_s2(d, ghm_plan_cus()).splits == 1;
        return lp_plan(d->N, d->K, d->H, d->W, d->C, d->kh, 1, ghm_plan_cus()).splits == 1 || lp_sm_finish_ok(d->C, (long)d->N * d->H * d->W);
    }
    return 0;
}

int ghm_conv2d_fwd_lp(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const void* wq, const float* bias, float* y,
                      int32_t act, float alpha, int32_t accumulate, int32_t dtype) {
    const LpIO io{x, nullptr, 0, y, nullptr, 0};
    if (int e = lp_check_io(io, "ghm_conv2d_fwd_lp")) return e;
    return lp_fwd_io(ctx, d, io, wq, bias, act, alpha, accumulate, dtype);
}

int ghm_conv2d_fwd_lp_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, const void* wq,
                        const float* bias, float* y, void* yq, int64_t yq_nstride, int32_t act, float alpha,
                        int32_t accumulate, int32_t dtype) {
    const LpIO io{nullptr, xq, (long)xq_nstride, y, yq, (long)yq_nstride};
    if (int e = lp_check_io(io, "ghm_conv2d_fwd_lp_q")) return e;
    return lp_fwd_io(ctx, d, io, wq, bias, act, alpha, accumulate, dtype);
}

int ghm_conv2d_dgrad_lp(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const void* wqT, const float* bias,
                        float* dx, int32_t act, float alpha, int32_t accumulate, int32_t dtype) {
    const LpIO io{dy, nullptr, 0, dx, nullptr, 0};
    if (int e = lp_check_io(io, "ghm_conv2d_dgrad_lp")) return e;
    return lp_dgrad_io(ctx, d, io, wqT, bias, act, alpha, accumulate, dtype);
}

int ghm_conv2d_dgrad_lp_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* dyq, int64_t dyq_nstride, const void* wqT,
                          const float* bias, float* dx, void* dxq, int64_t dxq_nstride, int32_t act, float alpha,
                          int32_t accumulate, int32_t dtype) {
    const LpIO io{nullptr, dyq, (long)dyq_nstride, dx, dxq, (long)dxq_nstride};
    if (int e = lp_check_io(io, "ghm_conv2d_dgrad_lp_q")) return e;
    return lp_dgrad_io(ctx, d, io, wqT, bias, act, alpha, accumulate, dtype);
}

int ghm_conv2d_dgrad_dact_lp_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* dyq, int64_t dyq_nstride, const void* wqT,
                               float* dx, void* dxq, int64_t dxq_nstride, const float* dact_y, int64_t dact_nstride,
                               int32_t dact, float dact_alpha, int32_t dtype) {
    const LpIO io{nullptr, dyq, (long)dyq_nstride, dx, dxq, (long)dxq_nstride};
    if (int e = lp_check_io(io, "ghm_conv2d_dgrad_dact_lp_q")) return e;
    GHM_CHECK(lp_dgrad_s2_single_pass(d, dtype) && (dact == GHM_ACT_RELU || dact == GHM_ACT_LRELU),
              "ghm_conv2d_dgrad_dact_lp_q: not served (ask ghm_dgrad_dact_supported: form 3)");
    return lp_dgrad_s2_io(ctx, d, io, wqT, nullptr, GHM_ACT_LINEAR, 0.f, 0, dact_y, (long)dact_nstride, dact, dact_alpha,
                          dtype, true);
}

int ghm_conv2d_fwd_pool_lp_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, const void* wq,
                             const float* bias, float* pooled, void* pooledq, int64_t pooledq_nstride, uint8_t* mask,
                             int32_t act, float alpha, int32_t dtype) {
    const LpIO io{nullptr, xq, (long)xq_nstride, pooled, pooledq, (long)pooledq_nstride};
    if (int e = lp_check_io(io, "ghm_conv2d_fwd_pool_lp_q")) return e;
    GHM_CHECK(lp_conv_pool_supported(d, act, dtype), "ghm_conv2d_fwd_pool_lp_q: geometry not served");
    return lp_fwd_pool_io(ctx, d, io, wq, bias, mask, act, alpha, dtype);
}

int ghm_conv2d_wgrad_lp_workspace(const ghm_conv_desc* d, size_t* bytes) {
    const LpWPlan v = lp_wplan(d, ghm_plan_cus());
    const size_t n = (size_t)d->C * d->kh * d->kw * d->K;
    size_t b = (v.ok && v.splits > 1) ? (size_t)v.splits * n * sizeof(float) : 16;
    const LpWQPlan q = lp_wqplan(d, ghm_plan_cus());          // the q-operand kernel splits by column strips
    if (q.ok && (size_t)q.ncols * q.splits_per_col * n * sizeof(float) > b) b = (size_t)q.ncols * q.splits_per_col * n * sizeof(float);
    *bytes = b;
    return 0;
}

int ghm_lp_wgrad_q_supported(const ghm_conv_desc* d, int32_t dtype) {
    return (dtype == GHM_DTYPE_BF16 || dtype == GHM_DTYPE_F16) && lp_wqplan(d, ghm_plan_cus()).ok;
}

int ghm_conv2d_wgrad_lp_q(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, int64_t xq_nstride, const void* dyq,
                          int64_t dyq_nstride, float* dwp, void* workspace, int32_t accumulate, int32_t dtype) {
    GHM_CHECK(ghm_lp_wgrad_q_supported(d, dtype), "ghm_conv2d_wgrad_lp_q: geometry / dtype not served (ask ghm_lp_wgrad_q_supported)");
    GHM_CHECK((((uintptr_t)xq | (uintptr_t)dyq) & 15) == 0, "ghm_conv2d_wgrad_lp_q: q tensors are 16-byte aligned");
    const LpWQPlan v = lp_wqplan(d, ghm_plan_cus());
    return dtype == GHM_DTYPE_BF16
               ? lp_launch_wgrad_q<GHM_DTYPE_BF16>(ctx, d, v, xq, (long)xq_nstride, dyq, (long)dyq_nstride, dwp, workspace, accumulate)
               : lp_launch_wgrad_q<GHM_DTYPE_F16>(ctx, d, v, xq, (long)xq_nstride, dyq, (long)dyq_nstride, dwp, workspace, accumulate);
}

int ghm_conv2d_wgrad_lp(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* dy, float* dwp,
                        void* workspace, int32_t accumulate, int32_t dtype) {
    GHM_CHECK(ghm_lp_supported(d, 2, dtype), "ghm_conv2d_wgrad_lp: geometry / dtype not served (ask ghm_lp_supported)");
    const LpWPlan v = lp_wplan(d, ghm_plan_cus());
    return dtype == GHM_DTYPE_BF16 ? lp_launch_wgrad<GHM_DTYPE_BF16>(ctx, d, v, x, dy, dwp, workspace, accumulate)
                                   : lp_launch_wgrad<GHM_DTYPE_F16>(ctx, d, v, x, dy, dwp, workspace, accumulate);
}

}  // extern "C"
