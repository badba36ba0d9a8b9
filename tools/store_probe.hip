// Store-pattern probe for the GEMM epilogue: every workgroup (512 threads, one per CU, `rounds` times) writes one 256 x 256 bf16 tile
// (128 KiB) of a [M, ldc] matrix in the access pattern of
//   0: the LDS-staged epilogue of gemm256_kernel  -- a thread writes 16 B, 32 consecutive lanes cover one 512 B row
//   1: the direct epilogue of gemm256p_kernel      -- per wave instruction 16 rows x 64 B (4 lanes x 16 B), the other half of each
//                                                     128 B line by the next instruction
//   2: direct, full lines                          -- per wave instruction 8 rows x 128 B (8 lanes x 16 B)
//   3: as 1 but the two halves of a line written by the SAME instruction pairs of i (control: 16 rows x 64 B, line completed 8 instr later)
// Reports chip-wide write throughput.   build: hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o gpurun_out/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned short* C, long ldc, int tiles_n, int rounds, unsigned int seed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, g4 = lane >> 4, mm = lane & 15;
    u32x4 v = {seed + tid, seed * 3 + tid, seed * 5 + tid, seed * 7 + tid};
    for (int r = 0; r < rounds; ++r) {
        const int t = r * gridDim.x + blockIdx.x;
        const long m0 = (long)(t / tiles_n) * 256, n0 = (long)(t % tiles_n) * 256;
        if (MODE == 0) {
            for (int half = 0; half < 2; ++half)
                for (int pass = 0; pass < 8; ++pass) {
                    const int row = half * 128 + pass * 16 + tid / 32, c0 = (tid % 32) * 8;
                    *reinterpret_cast<u32x4*>(C + (m0 + row) * ldc + n0 + c0) = v;
                }
        } else if (MODE == 1) {
            unsigned short* cp = C + (m0 + wm * 128 + mm) * ldc + n0 + wn * 64 + 8 * g4;
            for (int i = 0; i < 8; ++i) {
                *reinterpret_cast<u32x4*>(cp + (long)(16 * i) * ldc) = v;
                *reinterpret_cast<u32x4*>(cp + (long)(16 * i) * ldc + 32) = v;
            }
        } else if (MODE == 2) {
            // lane pair (mm, mm^1) shares two rows: even lane writes the low 64 B half of the pair's rows, odd lane the high half
            const int piece = (mm & 1) * 4 + g4;   // 16-byte piece of the 128 B line
            unsigned short* cp = C + (m0 + wm * 128 + (mm & ~1)) * ldc + n0 + wn * 64 + 8 * piece;
            for (int i = 0; i < 8; ++i) {
                *reinterpret_cast<u32x4*>(cp + (long)(16 * i) * ldc) = v;
                *reinterpret_cast<u32x4*>(cp + (long)(16 * i + 1) * ldc) = v;
            }
        } else {
            unsigned short* cp = C + (m0 + wm * 128 + mm) * ldc + n0 + wn * 64 + 8 * g4;
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(cp + (long)(16 * i) * ldc) = v;
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(cp + (long)(16 * i) * ldc + 32) = v;
        }
        v.x += 1;
    }
}

int main(int argc, char** argv) {
    const int tiles_m = 64, tiles_n = 20, rounds = 5, G = 256;   // vit.lin1: 16384 x 5120, 1280 tiles = 5 rounds of 256
    const long ldc = (long)tiles_n * 256;
    unsigned short* C;
    hipMalloc(&C, (size_t)tiles_m * 256 * ldc * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)rounds * G * 131072.0;
    for (int mode = 0; mode < 4; ++mode) {
        for (int g : {256, 32, 8}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) probe<0><<<g, 512>>>(C, ldc, tiles_n, rounds, rep);
                if (mode == 1) probe<1><<<g, 512>>>(C, ldc, tiles_n, rounds, rep);
                if (mode == 2) probe<2><<<g, 512>>>(C, ldc, tiles_n, rounds, rep);
                if (mode == 3) probe<3><<<g, 512>>>(C, ldc, tiles_n, rounds, rep);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double b = bytes * g / G;
            printf("mode %d  workgroups %3d: %8.1f us  %7.2f TB/s chip  %7.1f GB/s per workgroup  (%.1f us per 128 KiB tile)\n", mode, g, best * 1e3,
                   b / best / 1e9, b / g / best / 1e6, best * 1e3 / rounds);
        }
    }
    return 0;
}
