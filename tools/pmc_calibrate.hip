// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the conv kernels use
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").  Each kernel
// streams a 1 GiB buffer (4x the Infinity Cache) exactly once:
//   read_b32      plain global_load_dword, lanes consecutive            (bf16 patch gathers, direct kernels)
//   read_b128     plain global_load_dwordx4                             (weight-gradient windows, element-wise)
//   dma_b32       global_load_lds_dword  (4-byte global->LDS DMA)       (fp32 patch kernels, thin fan-out loader)
//   dma_b128      global_load_lds_dwordx4                               (weight tiles of the patch kernels)
//   write_b32 / write_b128  stores
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o /tmp/pmc_cal
//               rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal_f -- /tmp/pmc_cal   (and WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void read_b32(const float* __restrict__ p, float* out, long n) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void read_b128(const float4* __restrict__ p, float* out, long n4) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void dma_b32(const float* __restrict__ p, float* out, long n) {
    __shared__ float lds[256 * 8];
    float s = 0.f;
    const int wave = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 2048; base + 2048 <= n; base += (long)gridDim.x * 2048) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(p + base + q * 256 + threadIdx.x), (lptr_t)(lds + q * 256 + wave * 64), 4, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        s += lds[threadIdx.x];
        __syncthreads();
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void dma_b128(const float* __restrict__ p, float* out, long n) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4];
    float s = 0.f;
    const int wave = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 4096; base + 4096 <= n; base += (long)gridDim.x * 4096) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(p + base + (q * 256 + threadIdx.x) * 4), (lptr_t)(lds + (q * 256 + wave * 64) * 4), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        s += lds[threadIdx.x];
        __syncthreads();
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void write_b32(float* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 1.f;
}
__global__ void write_b128(float4* p, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const long n = 1L << 28;      // 1 GiB of floats
    float *p, *out;
    hipMalloc(&p, n * 4);
    hipMalloc(&out, 256);
    hipMemset(p, 0, n * 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        read_b32<<<4096, 256>>>(p, out, n);
        read_b128<<<4096, 256>>>((const float4*)p, out, n / 4);
        dma_b32<<<4096, 256>>>(p, out, n);
        dma_b128<<<4096, 256>>>(p, out, n);
        write_b32<<<4096, 256>>>(p, n);
        write_b128<<<4096, 256>>>((float4*)p, n / 4);
    }
    hipDeviceSynchronize();
    printf("each kernel moved %ld bytes\n", n * 4);
    return 0;
}
