#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // hypothesis A: lane t of a 16-lane group supplies the address of key t/4, column quad t%4 of a [4][16] block
    int addr_elems;
    if (mode == 0) addr_elems = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;       // dense 4x16 blocks per group
    else addr_elems = (l >> 4) * 256 + ((l & 15) >> 2) * 64 + (l & 3) * 4;                // key stride 64 elements
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr_elems));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
