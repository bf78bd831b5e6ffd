// v_smfmac_f32_32x32x32_bf16 on gfx950: (1) where a compressed A value lands in the contraction (one-hot scan: which B element
// it meets, per lane half, compressed slot and 2-bit index), (2) its rate against the dense v_mfma_f32_32x32x16_bf16 with random
// operands (the dense pipe is power-managed).  Tuning probe for a sparse form of the 5x5 weight gradient.
//   hipcc --offload-arch=gfx950 -O3 tools/smfmac_probe.hip -o tools/smfmac_probe && tools/smfmac_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

// block b = ((hA * 8 + s) * 4 + iv) * 32 + hB * 16 + jj: A = 1.0 in lane (row 0, half hA), compressed slot s, its index iv (the
// slot's partner in the group gets a different index); B = 1.0 in lane (column 0, half hB), element jj.  hit[b] = sum |D|.
template <int ABID>
__global__ void scan(float* hit) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int jj = b & 15, hB = (b >> 4) & 1, iv = (b >> 5) & 3, s = (b >> 7) & 7, hA = (b >> 10) & 1;
    unsigned short av[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bv[16];
    for (int i = 0; i < 16; ++i) bv[i] = 0;
    unsigned idx = 0;
    for (int g = 0; g < 4; ++g) {            // default indices of a group: (0, 1) ... or around the probed slot
        int i0 = 0, i1 = 1;
        if (g == s / 2) {
            if ((s & 1) == 0) { i0 = iv; i1 = iv == 3 ? 2 : 3; if (i1 <= i0) i1 = (i0 + 1) & 3; }
            else { i1 = iv; i0 = iv == 0 ? 1 : 0; }
        }
        idx |= (unsigned)(i0 | (i1 << 2)) << (4 * g);
    }
    if (lane == hA * 32) av[s] = 0x3f80;     // bf16 1.0
    if (lane == hB * 32) bv[jj] = 0x3f80;
    u32x4 a4;
    u32x8 b8;
    for (int i = 0; i < 4; ++i) a4[i] = av[2 * i] | ((unsigned)av[2 * i + 1] << 16);
    for (int i = 0; i < 8; ++i) b8[i] = bv[2 * i] | ((unsigned)bv[2 * i + 1] << 16);
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int idxv = ABID ? (int)(idx << 16) : (int)idx;
    acc = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x16, b8), acc, idxv, 0, ABID);
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += fabsf(acc[e]);
    for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) hit[b] = t;
}

// dense layout check with the same one-hot idea: A lane (row 0, half hA) element ja, B lane (col 0, half hB) element jb
__global__ void scan_dense(float* hit) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int jb = b & 7, hB = (b >> 3) & 1, ja = (b >> 4) & 7, hA = (b >> 7) & 1;
    unsigned short av[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane == hA * 32) av[ja] = 0x3f80;
    if (lane == hB * 32) bv[jb] = 0x3f80;
    u32x4 a4, b4;
    for (int i = 0; i < 4; ++i) { a4[i] = av[2 * i] | ((unsigned)av[2 * i + 1] << 16); b4[i] = bv[2 * i] | ((unsigned)bv[2 * i + 1] << 16); }
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc, 0, 0, 0);
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += fabsf(acc[e]);
    for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) hit[b] = t;
}

// rate: 4 waves per block, one block per CU-slot, NACC independent accumulators, ITERS x NACC instructions per wave
template <bool SPARSE>
__global__ __launch_bounds__(256) void rate(const unsigned* src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a4;
    u32x8 b8;
    for (int i = 0; i < 4; ++i) a4[i] = src[(lane * 13 + i * 7 + blockIdx.x) & 4095];
    for (int i = 0; i < 8; ++i) b8[i] = src[(lane * 29 + i * 3 + 1000 + blockIdx.x) & 4095];
    const int idx = 0x4e4e4e4e;              // (2, 3), (0, 1) alternating: any valid pattern
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (SPARSE)
                acc[a] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x16, b8), acc[a], idx, 0, 0);
            else
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, u32x4{b8[0], b8[1], b8[2], b8[3]}), acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* hit;
    (void)hipMalloc(&hit, 4096 * sizeof(float));
    std::vector<float> h(4096);
    for (int abid = 0; abid < 2; ++abid) {
        (void)hipMemset(hit, 0, 4096 * sizeof(float));
        if (abid) hipLaunchKernelGGL(scan<1>, dim3(2048), dim3(64), 0, 0, hit); else hipLaunchKernelGGL(scan<0>, dim3(2048), dim3(64), 0, 0, hit);
        (void)hipMemcpy(h.data(), hit, 2048 * sizeof(float), hipMemcpyDeviceToHost);
        printf("== sparse, ABID=%d (index bits %s): for A lane half hA, compressed slot s, index iv -> the B (half, element) it multiplies\n", abid, abid ? "[31:16]" : "[15:0]");
        for (int hA = 0; hA < 2; ++hA)
            for (int s = 0; s < 8; ++s) {
                printf("  hA=%d s=%d:", hA, s);
                for (int iv = 0; iv < 4; ++iv) {
                    printf("  iv=%d->", iv);
                    int n = 0;
                    for (int q = 0; q < 32; ++q)
                        if (h[((hA * 8 + s) * 4 + iv) * 32 + q] != 0.f) { printf("(h%d,e%d)", q >> 4, q & 15); ++n; }
                    if (!n) printf("none");
                }
                printf("\n");
            }
    }
    hipLaunchKernelGGL(scan_dense, dim3(256), dim3(64), 0, 0, hit);
    (void)hipMemcpy(h.data(), hit, 256 * sizeof(float), hipMemcpyDeviceToHost);
    printf("== dense 32x32x16: A (half, element) -> B (half, element)\n");
    for (int hA = 0; hA < 2; ++hA) {
        printf("  hA=%d:", hA);
        for (int ja = 0; ja < 8; ++ja)
            for (int q = 0; q < 16; ++q)
                if (h[(hA * 8 + ja) * 16 + q] != 0.f) printf(" e%d->(h%d,e%d)", ja, q >> 3, q & 7);
        printf("\n");
    }
    unsigned* src;
    (void)hipMalloc(&src, 4096 * 4);
    std::vector<unsigned> r(4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (auto& v : r) {
            // two bf16 values in (-2, 2): random sign / mantissa, exponent 126..127
            unsigned lo = data ? ((rand() & 0x80ff) | (0x3f00 + ((rand() & 1) << 7))) : 0, hi = data ? ((rand() & 0x80ff) | (0x3f00 + ((rand() & 1) << 7))) : 0;
            v = lo | (hi << 16);
        }
        (void)hipMemcpy(src, r.data(), 4096 * 4, hipMemcpyHostToDevice);
        for (int sp = 0; sp < 2; ++sp)
            for (int rep = 0; rep < 2; ++rep) {
                const int iters = 20000, blocks = 256 * 2;
                (void)hipEventRecord(e0, 0);
                if (sp) hipLaunchKernelGGL(rate<true>, dim3(blocks), dim3(256), 0, 0, src, hit, iters);
                else hipLaunchKernelGGL(rate<false>, dim3(blocks), dim3(256), 0, 0, src, hit, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                const double inst = (double)blocks * 4 * iters * 4;
                const double flop = inst * 2.0 * 32 * 32 * (sp ? 32 : 16);
                printf("%s %s data: %.3f ms, %.1f G instr/s, %.1f TFLOP/s (logical), %.2f cycles/instr/SIMD at 2.4 GHz\n", sp ? "smfmac 32x32x32" : "mfma   32x32x16",
                       data ? "random" : "zero  ", ms, inst / ms / 1e6, flop / ms / 1e9, 2.4e9 * (ms / 1e3) / (inst / 1024.0));
            }
    }
    return 0;
}
