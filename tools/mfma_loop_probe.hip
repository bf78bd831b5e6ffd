// What slows a 1-wave-per-SIMD bf16 MFMA loop: operand rotation, interleaved LDS fragment reads, barriers (tuning probe for
// csrc/conv_split.hip).  hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o tools/mfma_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// MODE bit 0: fragments re-read from LDS every k-step (12 ds_read_b128 per 24 MFMAs); bit 1: barrier every 5 k-steps;
// bit 2: random operand data instead of zeros; bit 3: 10 global_load_lds (16 B / lane) per wave and iteration into a spare
// LDS region, waited for (vmcnt(0)) in front of the iteration's barrier -- the staging traffic of sp_conv_kernel
template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* src, float* out, int iters) {
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(MODE & 4) ? i : 0];
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    const u32x4* L = lds + threadIdx.x % 64;
    u32x4 af[2][3][2], bf[2][3][2];
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 2; ++i) { af[0][p][i] = L[(p * 2 + i) * 64]; bf[0][p][i] = L[(6 + p * 2 + i) * 64]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int cur = b & 1, nxt = cur ^ 1;
            if (MODE & 1) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        af[nxt][p][i] = L[((b + 1) * 12 + p * 2 + i) * 64];
                        bf[nxt][p][i] = L[((b + 1) * 12 + 6 + p * 2 + i) * 64 + 1024];
                    }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < 2; ++i) { af[nxt][p][i] = af[cur][p][i]; bf[nxt][p][i] = bf[cur][p][i]; }
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int a = (NACC == 8 && pr == 5) ? 4 + i * 2 + j : i * 2 + j;
                        acc[a] = mf(af[cur][PA[pr]][i], bf[cur][PB[pr]][j], acc[a]);
                    }
            }
            if (MODE & 1) {
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE & 8) && b == 0) {
                typedef const __attribute__((address_space(1))) void* gptr_t;
                typedef __attribute__((address_space(3))) void* lptr_t;
                const u32x4* g = src + ((it * 2560 + blockIdx.x * 64 + threadIdx.x) & 0x1ffff);
#pragma unroll
                for (int q = 0; q < 10; ++q)
                    __builtin_amdgcn_global_load_lds((gptr_t)(g + q * 256), (lptr_t)(lds + 4096 + (threadIdx.x / 64) * 640 + q * 64), 16, 0, 0);
            }
        }
        if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE, int NACC>
void run(const char* name, const u32x4* src, float* d, int reps = 4) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 256;
    (void)hipFuncSetAttribute((const void*)probe<MODE, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(256), 140 * 1024, 0, src, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(256), 140 * 1024, 0, src, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)reps * blocks * 4 * iters * 120.0 * 2.0 * 32 * 32 * 16;
    printf("%-64s %8.1f TFLOP/s\n", name, total / (ms * 1e-3) / 1e12);
}
int main() {
    u32x4* src; float* d;
    (void)hipMalloc(&src, (size_t)0x20000 * 16 + 4096 * 16); (void)hipMalloc(&d, 64);
    unsigned* h = (unsigned*)malloc(4096 * 16);
    for (int i = 0; i < 4096 * 4; ++i) { unsigned r = (unsigned)rand(); h[i] = (r & 0x807f807f) | 0x3f003f00; }   // bf16 values near 1
    (void)hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    run<0, 4>("registers only, 4 accumulators, zero data", src, d);
    run<0, 8>("registers only, 8 accumulators, zero data", src, d);
    run<4, 8>("registers only, 8 accumulators, random data", src, d);
    run<1, 8>("+ 12 ds_read_b128 per 24 MFMAs, zero data", src, d);
    run<5, 8>("+ 12 ds_read_b128 per 24 MFMAs, random data", src, d);
    run<3, 8>("+ reads + barrier every 120 MFMAs, zero data", src, d);
    run<7, 8>("+ reads + barrier every 120 MFMAs, random data", src, d);
    run<15, 8>("+ 40 KB of LDS DMA per block and iteration, random data", src, d, 50);
    run<7, 8>("the same, SUSTAINED (0.6 s of launches), random data", src, d, 200);
    run<3, 8>("the same, SUSTAINED (0.6 s of launches), zero data", src, d, 200);
    return 0;
}
