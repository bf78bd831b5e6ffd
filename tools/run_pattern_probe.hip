// HBM read rate of the tiled access pattern of thin_wgrad_kernel: a persistent block per CU walks image rows and reads,
// per tile, RUN bytes from each of NP channel planes (planes 1 MB apart).  tools only -- not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/run_pattern_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NP, int RUN4>          // planes per tile, float4 per run
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ x, float* __restrict__ out, int N, int C, int H, int W4) {
    // x: [N][C][H][W4] float4; tile = NP planes x RUN4 float4; a block walks rows r = blockIdx.x, += gridDim.x
    float acc = 0.f;
    const int tid = threadIdx.x;
    const int per_tile = NP * RUN4;                 // float4 per tile
    for (int r = blockIdx.x; r < N * H; r += gridDim.x) {
        const int n = r / H, u = r % H;
        for (int cg = 0; cg < C / NP; ++cg)
            for (int t = 0; t < W4 / RUN4; ++t) {
                for (int e = tid; e < per_tile; e += 256) {
                    const int p = e / RUN4, j = e % RUN4;
                    const float4 v = x[(((long)n * C + cg * NP + p) * H + u) * W4 + t * RUN4 + j];
                    acc += v.x + v.y + v.z + v.w;
                }
            }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int NP, int RUN4>
void run(const char* name, const float4* x, float* out, int N, int C, int H, int W4, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<NP, RUN4>), dim3(blocks), dim3(256), 0, 0, x, out, N, C, H, W4);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<NP, RUN4>), dim3(blocks), dim3(256), 0, 0, x, out, N, C, H, W4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double bytes = (double)N * C * H * W4 * 16;
    printf("%-34s blocks %4d  %.3f ms  %.2f TB/s\n", name, blocks, ms, bytes / ms / 1e9);
}

int main() {
    const int N = 8, C = 64, H = 512, W4 = 128;          // d_conv1's dy: 537 MB
    float4* x; float* out;
    hipMalloc(&x, (size_t)N * C * H * W4 * 16); hipMalloc(&out, 256);
    hipMemset(x, 0, (size_t)N * C * H * W4 * 16);
    for (int blocks : {256, 512, 1024}) {
        run<64, 32>("64 planes x 512 B (thin_wgrad<64>)", x, out, N, C, H, W4, blocks);
        run<32, 64>("32 planes x 1 KB", x, out, N, C, H, W4, blocks);
        run<16, 128>("16 planes x 2 KB (whole rows)", x, out, N, C, H, W4, blocks);
        run<128 / 2, 16>("64 planes x 256 B", x, out, N, C, H, W4, blocks);
        run<1, 128>("1 plane x 2 KB (linear rows)", x, out, N, C, H, W4, blocks);
    }
    return 0;
}
