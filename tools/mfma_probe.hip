// Calibration probe: what does "100 %" of the fp32 MFMA pipe look like on this box, and what do an LDS
// fragment read per k-step and a barrier per slab cost?   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(float* out, int iters) {
    __shared__ float lds[2 * 16 * 264];
    for (int i = threadIdx.x; i < 2 * 16 * 264; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lane = threadIdx.x & 63;
    float a0 = lane * 0.01f, a1 = lane * 0.02f, b0 = lane * 0.03f, b1 = lane * 0.04f;
    const float* A = lds + (lane & 31) + (lane >> 5) * 132;
    const float* B = lds + 16 * 132 + (lane & 31) + (lane >> 5) * 132;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (MODE >= 1) {
                a0 = A[ks * 264]; a1 = A[ks * 264 + 32]; b0 = B[ks * 264]; b1 = B[ks * 264 + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 3) {   // emulate the staging stores of one slab
            for (int q = 0; q < 10; ++q) lds[(it & 1) * 16 * 264 + q * 256 + threadIdx.x] = a0 + q;
            __syncthreads();
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("mfma only (regs)", blocks, d);
        run<1>("+ LDS fragment reads per k-step", blocks, d);
        run<2>("+ barrier per 8 k-steps", blocks, d);
        run<3>("+ 10 ds_write + 2nd barrier per slab", blocks, d);
    }
    return 0;
}
