// Probe: semantics of `buffer_load_dwordx4 ... offen offset:IMM lds` on gfx950 (LDS-DMA through a buffer descriptor).
//  (1) is the instruction offset added to BOTH the memory address and the LDS address?   (2) does num_records = 0 suppress the
//  fetch (zeros, no fault)?   (3) is the SGPR offset left out of the range check?
// build+run: hipcc --offload-arch=gfx950 -O2 -o /tmp/buffer_lds tools/probes/buffer_lds.hip && /tmp/buffer_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ void k(const int* a, int nrec, int soff, int* out) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 4096; i += 64) reinterpret_cast<int*>(smem)[i] = -1;
    __syncthreads();
    i32x4 r; r[0] = (int)(size_t)a; r[1] = (int)((size_t)a >> 32); r[2] = nrec; r[3] = 0x00020000;
    unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(4096));
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" :: "v"(voff), "s"(r), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = reinterpret_cast<int*>(smem)[i];
}
int main() {
    std::vector<int> h(1 << 16);
    for (int i = 0; i < (1 << 16); ++i) h[i] = i;
    int *a, *o;
    hipMalloc(&a, h.size() * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    struct { int nrec, soff; const char* what; } cases[] = {{1 << 18, 0, "in range"}, {1 << 18, 8192, "soffset 8192"}, {0, 0, "num_records 0"}, {1500, 0, "num_records 1500 (voffset+1024 range check)"}, {2048, 1 << 17, "num_records 2048, soffset 128 KiB"}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, a, c.nrec, c.soff, o);
        std::vector<int> r(4096);
        hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
        int first = -1, cnt = 0;
        for (int i = 0; i < 4096; ++i) if (r[i] != -1) { if (first < 0) first = i; ++cnt; }
        printf("%-45s: %d dwords written, first at LDS byte %d; lane0 dwords %d %d, lane 20: %d, lane 63: %d\n", c.what, cnt, first * 4, first >= 0 ? r[first] : 0, first >= 0 ? r[first + 1] : 0,
               first >= 0 ? r[first + 80] : 0, first >= 0 ? r[first + 252] : 0);
    }
    return 0;
}
