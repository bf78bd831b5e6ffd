// What v_mfma_f32_32x32x16_bf16 sustains on this box: a register-only MFMA loop (no memory, no LDS) at 1 / 2 / 4 waves per
// SIMD with 4 independent accumulators per wave, timed with HIP events after a 100 ms warm-up.  The nominal dense peak is
// 2.5 PFLOP/s at 2.4 GHz (MI355X_MICROARCH.md); under a sustained all-CU MFMA load the shader clock is power-managed.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o /tmp/mfma_bf16_probe && /tmp/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int FP32>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x + e); y[e] = (__bf16)(float)(blockIdx.x + e); }
    float fx = (float)threadIdx.x, fy = (float)blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (FP32) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[0] = s;
}
template <int FP32>
void run(const char* name, int blocks_per_cu, float* d) {
    int cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int w = 0; w < 6; ++w) hipLaunchKernelGGL(probe<FP32>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters);      // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<FP32>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops_per_mfma = FP32 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    const double total = (double)reps * cus * blocks_per_cu * 4 /*waves*/ * iters * 32.0 * flops_per_mfma;
    printf("%-28s %d wave(s)/SIMD: %8.1f TFLOP/s  (%.2f ms per launch)\n", name, blocks_per_cu, total / (ms * 1e-3) / 1e12, ms / reps);
}
int main() {
    float* d;
    hipMalloc(&d, 64);
    for (int b = 1; b <= 4; b *= 2) run<0>("v_mfma_f32_32x32x16_bf16", b, d);
    for (int b = 1; b <= 2; b *= 2) run<1>("v_mfma_f32_32x32x2_f32", b, d);
    return 0;
}
