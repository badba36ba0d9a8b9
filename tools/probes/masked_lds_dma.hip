// Probe: what does global_load_lds_dwordx4 do for lanes that are masked off (EXEC = 0)?  The LDS image is prefilled with 0xAAAAAAAA, every lane's
// source chunk holds its lane id, lanes with (lane & 15) >= 10 are masked.  Prints the LDS image per 16-byte slot afterwards.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/masked_lds_dma.hip -o /tmp/masked && /tmp/masked
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
__global__ void k(const unsigned int* src, unsigned int* out) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xAAAAAAAAu;
    __syncthreads();
    if ((lane & 15) < 10) __builtin_amdgcn_global_load_lds(GLB_PTR(src + lane * 4), LDS_PTR(lds), 16, 0, 0);
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
    unsigned int h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = i / 4;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 64; ++s) printf("%s%x", s % 16 ? " " : "\n", h[s * 4]);
    printf("\n");
    return 0;
}
