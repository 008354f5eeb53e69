// Read-bandwidth microbenchmark: what a pure streaming read of N bytes can reach on this chip, as a
// function of loads in flight per lane, workgroup size, grid size and access pattern.  Development
// aid for K1/K4 tuning (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pattern 0: grid-stride over 16-byte words, U independent loads in flight per lane
template <int U, int NT>
__global__ void k_read_flat(const uint4* __restrict__ p, size_t n16, uint4* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) {
                typedef unsigned int v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_nontemporal_load((const v4u*)(p + i + u * stride));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = p[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

// pattern 1: "rows": a wave owns R rows of row_bytes (rows `row_stride` rows apart), walks them in
// 1 KiB tiles with R loads in flight + the next tile prefetched (K1's access pattern)
template <int R, int PF>
__global__ void k_read_rows(const char* __restrict__ base, int n_rows, int row_bytes, int row_stride, uint4* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int g0 = wave * R;                       // group of R logical rows
    if (g0 >= n_rows) return;
    const char* row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int j = g0 + r; if (j >= n_rows) j = n_rows - 1;
        // logical row j -> physical row: j = p * F + f  ->  f * P + p  with F = row_stride rows apart
        const int F = row_stride, P = n_rows / row_stride;
        const int pp = j / F, f = j - pp * F;
        row[r] = base + (size_t)(f * P + pp) * row_bytes;
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 cur[R], nxt[R];
    const int tiles = row_bytes >> 10;
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = *(const uint4*)(row[r] + lane * 16);
    if (PF) {
        for (int t = 0; t < tiles; ++t) {
            const int no = (t + 1 < tiles ? t + 1 : t) * 1024 + lane * 16;
#pragma unroll
            for (int r = 0; r < R; ++r) nxt[r] = *(const uint4*)(row[r] + no);
#pragma unroll
            for (int r = 0; r < R; ++r) { acc.x ^= cur[r].x; acc.y ^= cur[r].y; acc.z ^= cur[r].z; acc.w ^= cur[r].w; cur[r] = nxt[r]; }
        }
    } else {
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int r = 0; r < R; ++r) { acc.x ^= cur[r].x; acc.y ^= cur[r].y; acc.z ^= cur[r].z; acc.w ^= cur[r].w; }
            if (t + 1 < tiles) {
#pragma unroll
                for (int r = 0; r < R; ++r) cur[r] = *(const uint4*)(row[r] + (t + 1) * 1024 + lane * 16);
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}


// pattern 2: a wave owns NR consecutive by-patch rows and streams them ONE AFTER THE OTHER (a single
// sequential stream per wave, DEPTH tiles in flight) - the shape of a similarity kernel that keeps
// the previous row in registers instead of reading R rows side by side
template <int NR, int DEPTH>
__global__ void k_read_seq(const char* __restrict__ base, int n_rows, int row_bytes, int row_stride, int overlap,
                           uint4* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int g0 = wave * (NR - overlap);
    if (g0 >= n_rows) return;
    const int F = row_stride, P = n_rows / row_stride;
    const int tiles = row_bytes >> 10;
    const int total = NR * tiles;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 buf[DEPTH];
    auto addr = [&](int it) {
        int r = it / tiles, t = it - r * tiles;
        int j = g0 + r; if (j >= n_rows) j = n_rows - 1;
        const int pp = j / F, f = j - pp * F;
        return base + (size_t)(f * P + pp) * row_bytes + t * 1024 + lane * 16;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = *(const uint4*)addr(d < total ? d : total - 1);
    for (int it = 0; it < total; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint4 v = buf[d];
            const int nx = it + d + DEPTH;
            buf[d] = *(const uint4*)addr(nx < total ? nx : total - 1);
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

template <class F>
static float time_us(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 302) * 1000000ull / 16 * 16;
    // two buffers, alternated, so that a buffer larger than the Infinity Cache is really cold
    char *a, *b; uint4* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16;
    int flip = 0;
    printf("buffer %.1f MB (two buffers alternated)\n", bytes / 1e6);
#define FLAT(U, NT, TPB, BLOCKS) { \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_flat<U, NT>), dim3(BLOCKS), dim3(TPB), 0, 0, (const uint4*)(flip ? a : b), n16, sink); }, 10); \
        printf("flat  U=%d nt=%d tpb=%4d blocks=%6d : %7.1f us  %7.1f GB/s\n", U, NT, TPB, BLOCKS, us, bytes / us / 1e3); }
    FLAT(1, 0, 256, 256 * 8) FLAT(2, 0, 256, 256 * 8) FLAT(4, 0, 256, 256 * 8) FLAT(8, 0, 256, 256 * 8)
    FLAT(4, 0, 256, 256 * 4) FLAT(4, 0, 256, 256 * 16) FLAT(4, 0, 256, 256 * 32) FLAT(8, 0, 256, 256 * 4)
    FLAT(4, 1, 256, 256 * 8) FLAT(8, 1, 256, 256 * 8) FLAT(4, 0, 512, 256 * 4) FLAT(4, 0, 1024, 256 * 2) FLAT(16, 0, 256, 256 * 4)
    const int row_bytes = 8192, F = 64, n_rows = (int)(bytes / row_bytes) / F * F;
#define ROWS(R, TPB) ROWSP(R, TPB, 1)
#define ROWSP(R, TPB, PF) { \
        const int waves = (n_rows + R - 1) / R, blocks = (waves * 64 + TPB - 1) / TPB; \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_rows<R, PF>), dim3(blocks), dim3(TPB), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, sink); }, 10); \
        printf("rows  R=%d pf=%d tpb=%4d blocks=%6d : %7.1f us  %7.1f GB/s\n", R, PF, TPB, blocks, us, (double)n_rows * row_bytes / us / 1e3); }
    ROWSP(5, 256, 0) ROWSP(4, 256, 0) ROWSP(3, 256, 0) ROWSP(2, 256, 0) ROWSP(1, 256, 0) ROWSP(5, 512, 0) ROWSP(5, 128, 0) ROWSP(3, 256, 1) ROWSP(9, 256, 0) ROWSP(9, 256, 1)
    ROWS(1, 256) ROWS(2, 256) ROWS(4, 256) ROWS(5, 256) ROWS(8, 256) ROWS(4, 128) ROWS(4, 512) ROWS(2, 512)
#define SEQ(NR, DEPTH, TPB, OV) { \
        const int waves = (n_rows + (NR - OV) - 1) / (NR - OV), blocks = (waves * 64 + TPB - 1) / TPB; \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_seq<NR, DEPTH>), dim3(blocks), dim3(TPB), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, OV, sink); }, 10); \
        printf("seq   NR=%2d depth=%d tpb=%4d ov=%d blocks=%6d : %7.1f us  %7.1f GB/s (unique bytes)\n", NR, DEPTH, TPB, OV, blocks, us, (double)n_rows * row_bytes / us / 1e3); }
    SEQ(4, 2, 256, 0) SEQ(8, 2, 256, 0) SEQ(16, 2, 256, 0) SEQ(8, 4, 256, 0) SEQ(16, 4, 256, 0) SEQ(8, 2, 64, 0) SEQ(16, 4, 64, 0) SEQ(16, 8, 64, 0)
    SEQ(5, 2, 256, 1) SEQ(9, 2, 256, 1) SEQ(17, 2, 256, 1) SEQ(9, 4, 256, 1) SEQ(17, 4, 256, 1) SEQ(9, 4, 64, 1) SEQ(17, 4, 64, 1) SEQ(17, 8, 64, 1) SEQ(33, 4, 64, 1) SEQ(33, 8, 64, 1)
    return 0;
}
