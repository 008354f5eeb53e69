// How many workgroups of 256 threads does the chip REALLY hold at once?  (development tool)
// Every workgroup stamps its start on the 100 MHz wall clock, spins ~30 us and leaves; the host counts the workgroups
// that started within 5 us of the first one.  Variants: VGPR budget (launch bounds / register pressure), dynamic LDS.
//   hipcc --offload-arch=gfx950 -O3 -o occprobe occprobe.hip && ./occprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int kRegs>
__global__ __launch_bounds__(256) void k_probe(long long* start, float* sink, int spin_ticks) {
    extern __shared__ int lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    float acc[kRegs];
#pragma unroll
    for (int i = 0; i < kRegs; ++i) acc[i] = (float)(threadIdx.x + i);
    while (wall_clock64() - t0 < spin_ticks) {
#pragma unroll
        for (int i = 0; i < kRegs; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kRegs; ++i) s += acc[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x & 7];
}

// SGPR budget: clobber s<N> so that the kernel's next_free_sgpr is N + 1 (the hardware adds VCC / FLAT_SCRATCH / XNACK_MASK)
#define SGPR_PROBE(N)                                                                                   \
    __global__ __launch_bounds__(256) void k_sgpr_##N(long long* start, float* sink, int spin_ticks) { \
        const long long t0 = wall_clock64();                                                            \
        if (threadIdx.x == 0) start[blockIdx.x] = t0;                                                   \
        asm volatile("s_mov_b32 s" #N ", 0" ::: "s" #N);                                               \
        while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);                           \
    }
SGPR_PROBE(60) SGPR_PROBE(72) SGPR_PROBE(73) SGPR_PROBE(76) SGPR_PROBE(80) SGPR_PROBE(88) SGPR_PROBE(95)

template <typename K>
static void run_sgpr(const char* name, K kern) {
    const int grid = 4096;
    long long* d; float* sink;
    hipMalloc(&d, grid * sizeof(long long)); hipMalloc(&sink, 4);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, sink, 3000);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), d, grid * sizeof(long long), hipMemcpyDeviceToHost);
    const long long first = *std::min_element(h.begin(), h.end());
    int early = 0;
    for (long long t : h) early += (t - first) < 500;
    printf("%-28s occupancy API %d per CU; started within 5 us of the first: %d\n", name, per_cu, early);
    hipFree(d); hipFree(sink);
}

template <int kRegs>
static void run(const char* name, size_t lds, int grid) {
    long long* d; float* sink;
    hipMalloc(&d, grid * sizeof(long long)); hipMalloc(&sink, 4);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_probe<kRegs>, 256, lds);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k_probe<kRegs>);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_probe<kRegs>, dim3(grid), dim3(256), lds, 0, d, sink, 3000);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), d, grid * sizeof(long long), hipMemcpyDeviceToHost);
    const long long first = *std::min_element(h.begin(), h.end());
    int early = 0;
    for (long long t : h) early += (t - first) < 500;
    printf("%-28s regs %3d  lds %6zu  grid %d: occupancy API %d per CU; started within 5 us of the first: %d\n", name, fa.numRegs, lds, grid, per_cu, early);
    hipFree(d); hipFree(sink);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, maxThreadsPerMultiProcessor %d, regsPerBlock %d, sharedMemPerMultiprocessor %zu\n", p.name, p.multiProcessorCount,
           p.maxThreadsPerMultiProcessor, p.regsPerBlock, (size_t)p.sharedMemPerMultiprocessor);
    run<8>("small", 0, 4096);
    run<8>("small + 10 KB LDS", 10336, 4096);
    run<8>("small + 17.5 KB LDS", 17504, 4096);
    run<8>("small + 20 KB LDS", 20480, 4096);
    run<40>("~50 VGPR", 0, 4096);
    run<56>("~64 VGPR", 0, 4096);
    run<64>("~72 VGPR", 0, 4096);
    run<8>("small, grid 2048", 0, 2048);
    run_sgpr("next_free_sgpr 61", k_sgpr_60);
    run_sgpr("next_free_sgpr 73", k_sgpr_72);
    run_sgpr("next_free_sgpr 74", k_sgpr_73);
    run_sgpr("next_free_sgpr 77", k_sgpr_76);
    run_sgpr("next_free_sgpr 81", k_sgpr_80);
    run_sgpr("next_free_sgpr 89", k_sgpr_88);
    run_sgpr("next_free_sgpr 96", k_sgpr_95);
    return 0;
}
