// Micro-probe of the 256x256 GEMM inner loop on one launch of 256 workgroups x 8 waves: what does each ingredient cost?
//   MODE 0: 64 MFMA 16x16x32 per iteration on register-resident fragments
//   MODE 1: + 24 ds_read_b128 (conflict-free swizzled image) feeding the fragments, free running
//   MODE 2: + one s_barrier per iteration
//   MODE 3: the staggered four-slot schedule (two groups, four barriers), no DMA
//   MODE 4: MODE 3 + 8 global_load_lds per wave per iteration from an L2-resident source
//   MODE 5: MODE 4 with the DMA streaming new A / B tiles every iteration (A panel shared by 64 workgroups, B panel by 4, as the
//           grouped tile order of the real kernel), zeros or random data (MFMA power draw depends on the operand bits)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ bf16x8_t frag(const char* tile, int row, int ks, int g) {
    const int c = (ks * 4 + g) ^ (row & 7);
    return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + (c << 4));
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, float* __restrict__ out, int iters, long it_stride = 0,
                                             long a_panel = 0, long b_panel = 0, long b_base = 32768) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
    for (int i = tid; i < 131072 / 4; i += 512) reinterpret_cast<unsigned int*>(smem)[i] = 0x3f803f80u;
    __syncthreads();
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8_t a8[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a8[i] = frag(smem, wm * 128 + i * 16 + (lane & 15), 0, lane >> 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = frag(smem + 32768, wn * 64 + j * 16 + (lane & 15), 0, lane >> 4);
    const long lane_off = ((wave * 4) * 8 + (lane >> 3)) * 128 + ((lane & 7) << 4);
    const char* gsrc = src + (size_t)blockIdx.x * 65536 + lane_off;
    // workgroup -> XCD is round robin; inside an XCD 32 workgroups = 4 A panels x 8 B panels (the real kernel's grouped order)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const char* ga = MODE == 5 ? src + (local & 3) * a_panel + lane_off : gsrc;
    const char* gb = MODE == 5 ? src + b_base + (xcd * 8 + (local >> 2)) * b_panel + lane_off : gsrc + 32768;
    auto mmas = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto reads = [&](int buf, int ks) {
        const char* Ab = smem + buf * 65536;
        const char* Bb = Ab + 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = frag(Bb, wn * 64 + j * 16 + (lane & 15), ks, lane >> 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) a8[i] = frag(Ab, wm * 128 + i * 16 + (lane & 15), ks, lane >> 4);
    };
    auto dma = [&](int buf, int it) {
        const long o = (long)it * it_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(ga + o + i * 1024), LDS_PTR(smem + buf * 65536 + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(gb + o + i * 1024), LDS_PTR(smem + buf * 65536 + 32768 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    if (MODE >= 3 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (MODE == 0) {
            mmas(); mmas();
        } else if (MODE == 1 || MODE == 2) {
            if (MODE == 2) __builtin_amdgcn_s_barrier();
            reads(buf, 0); mmas(); reads(buf, 1); mmas();
        } else {
            if (MODE >= 4) dma(buf ^ 1, it + 1);
            reads(buf, 0);
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier();
            mmas();
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier();
            reads(buf, 1);
            if (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier();
            mmas();
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier();
        }
    }
    if (MODE >= 3 && grp == 0) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[tid] = s;
}

// register-resident MFMA only: 16x16x32 (SHAPE 16) or 32x32x16 (SHAPE 32), operands taken once from LDS (zeros or random)
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int SHAPE>
__global__ __launch_bounds__(512) void probe_regs(const char* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    const bf16x8_t* p = reinterpret_cast<const bf16x8_t*>(src) + tid * 12;
    bf16x8_t a8[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a8[i] = p[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = p[8 + j];
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[(i + ks) & 7], b[(j + ks) & 3], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 12345.678f) out[tid] = s;
}
template <int SHAPE> static void run_regs(const char* src, float* out, const char* name) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_regs<SHAPE><<<blocks, 512>>>(src, out, 200);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        probe_regs<SHAPE><<<blocks, 512>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 2.0 * 256 * 256 * 64 * (double)iters * blocks;
    printf("%-52s %8.3f ms  %7.1f TF/s  %6.3f us per K-tile\n", name, best, flops / best / 1e9, best * 1e3 / iters);
}

template <int MODE> static void run(const char* src, float* out, const char* name) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 512, 131072>>>(src, out, 200);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        probe<MODE><<<blocks, 512, 131072>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 2.0 * 256 * 256 * 64 * (double)iters * blocks;
    printf("%-52s %8.3f ms  %7.1f TF/s  %6.3f us per K-tile\n", name, best, flops / best / 1e9, best * 1e3 / iters);
}

static void run_stream(const char* src, float* out, const char* name, int iters, long a_panel, long b_panel, long b_base) {
    const int blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        probe<5><<<blocks, 512, 131072>>>(src, out, iters, 32768, a_panel, b_panel, b_base);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 2.0 * 256 * 256 * 64 * (double)iters * blocks;
    printf("%-52s %8.3f ms  %7.1f TF/s  %6.3f us per K-tile\n", name, best, flops / best / 1e9, best * 1e3 / iters);
}

int main() {
    char* src; float* out;
    const int sit = 512;                                   // streamed K-tiles
    const long panel = (long)(sit + 2) * 32768;            // one A or B panel: 32 KiB per K-tile
    const long total = 68 * panel;                         // 4 A panels + 64 B panels
    hipMalloc(&src, total); hipMemset(src, 0, total); hipMalloc(&out, 4096);
    run_regs<16>(src, out, "R16: registers only, 16x16x32, zeros");
    run_regs<32>(src, out, "R32: registers only, 32x32x16, zeros");
    run<0>(src, out, "0: 64 MFMA / K-tile, register fragments");
    run<1>(src, out, "1: + 24 ds_read_b128, free running");
    run<2>(src, out, "2: + one s_barrier per K-tile");
    run<3>(src, out, "3: staggered four-slot schedule, no DMA");
    run<4>(src, out, "4: staggered + 8 global_load_lds per wave (L2-resident)");
    run_stream(src, out, "5: staggered + DMA streaming A/B panels, zeros", sit, panel, panel, 4 * panel);
    {
        std::vector<unsigned short> h(total / 2);
        unsigned int x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 16) & 0x83ffu)); }  // random sign/mantissa, |v| ~ 0.01..2
        hipMemcpy(src, h.data(), total, hipMemcpyHostToDevice);
    }
    run_stream(src, out, "5r: same, random operands", sit, panel, panel, 4 * panel);
    run<4>(src, out, "4r: L2-resident DMA, random operands");
    run_regs<16>(src, out, "R16r: registers only, 16x16x32, random");
    run_regs<32>(src, out, "R32r: registers only, 32x32x16, random");
    return 0;
}
