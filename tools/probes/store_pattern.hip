// Write-rate probe: the same 512 MiB written by 2048 waves with (a) 1 KiB contiguous per wave instruction, (b) 16 rows x 64 B per instruction (row pitch 1 KiB:
// the image -> token kernel's fp32 output), (c) 16 rows x 64 B with row pitch 512 B (its bf16 outputs), (d) 8 rows x 128 B.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern.hip -o gpurun_out/store_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void wr(float* out, long rows_per_wave_iter, int iters, long wave_stride_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long w = (long)blockIdx.x * 8 + wave;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int it = 0; it < iters; ++it) {
        float* base = out + ((long)it * gridDim.x * 8 + w) * 16 * 256;          // 16 rows x 256 floats per wave and iteration
        if (MODE == 3) {                                                          // the image -> token kernel's order: workgroup = (prompt, part of 4), its 8 waves stride through the prompt's 4096 rows
            const int prompt = blockIdx.x >> 2, part = blockIdx.x & 3;
            base = out + ((long)prompt * 4096 + ((part * 8 + wave) + 32l * it) * 16) * 256;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            long off;
            if (MODE == 0) off = (long)t * 256 + lane * 4;                          // one whole row (1 KiB) per instruction
            else if (MODE == 1) off = (long)(lane & 15) * 256 + t * 16 + (lane >> 4) * 4;   // 16 rows x 64 B
            else if (MODE == 3) off = (long)(lane & 15) * 256 + t * 16 + (lane >> 4) * 4;
            else off = (long)((lane & 7) + 8 * (t & 1)) * 256 + (t >> 1) * 32 + (lane >> 3) * 4;   // 8 rows x 128 B
            *reinterpret_cast<f32x4*>(base + off) = v;
        }
    }
}
int main() {
    const long bytes = 256l << 20;   // 64 prompts x 4096 rows x 1 KiB
    float* d; hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256, iters = bytes / (grid * 8 * 16 * 1024);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            if (mode == 0) wr<0><<<grid, 512>>>(d, 0, iters, 0);
            if (mode == 1) wr<1><<<grid, 512>>>(d, 0, iters, 0);
            if (mode == 2) wr<2><<<grid, 512>>>(d, 0, iters, 0);
            if (mode == 3) wr<3><<<grid, 512>>>(d, 0, iters, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("mode %d: %.1f us, %.2f TB/s\n", mode, best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
