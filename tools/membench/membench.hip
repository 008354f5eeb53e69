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
template <int R, int PF, int ROT = 0>
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
    // ROT: every wave starts at its own tile and wraps around (waves that start together do not walk the
    // same 1 KiB column of every row together)
    const int rot = ROT == 1 ? (wave & (tiles - 1)) : ROT == 2 ? ((wave * 5 + (wave >> 3)) & (tiles - 1)) : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = *(const uint4*)(row[r] + rot * 1024 + lane * 16);
    if (PF) {
        for (int t = 0; t < tiles; ++t) {
            const int no = (((t + 1 < tiles ? t + 1 : t) + rot) & (tiles - 1)) * 1024 + lane * 16;
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


// pattern 1c: K1's shape (R rows of one patch side by side) with the waves mapped patch-fastest: adjacent waves read
// ADJACENT rows of the same frames (contiguous 8 KiB pieces) instead of later frames of the same patch
template <int R>
__global__ void k_read_rows_pf(const char* __restrict__ base, int n_rows, int row_bytes, int F, uint4* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int P = n_rows / F, groups = F / R;
    if (wave >= P * groups) return;
    const int pp = wave % P, g = wave / P;
    const char* row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) row[r] = base + (size_t)((g * R + r) * P + pp) * row_bytes;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 cur[R], nxt[R];
    const int tiles = row_bytes >> 10;
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = *(const uint4*)(row[r] + lane * 16);
    for (int t = 0; t < tiles; ++t) {
        const int no = (t + 1 < tiles ? t + 1 : t) * 1024 + lane * 16;
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[r] = *(const uint4*)(row[r] + no);
#pragma unroll
        for (int r = 0; r < R; ++r) { acc.x ^= cur[r].x; acc.y ^= cur[r].y; acc.z ^= cur[r].z; acc.w ^= cur[r].w; cur[r] = nxt[r]; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

// pattern 1d: frame-major pairs with NP consecutive patches per wave: rows (f-1, p0..p0+NP-1) and (f, p0..p0+NP-1)
template <int NP>
__global__ void k_read_pairs_np(const char* __restrict__ base, int n_rows, int row_bytes, int P, uint4* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int groups = P / NP, F = n_rows / P;
    if (wave >= groups * (F - 1)) return;
    const int g = wave % groups, f = wave / groups + 1;
    const char* row[2 * NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        row[r] = base + (size_t)((f - 1) * P + g * NP + r) * row_bytes;
        row[NP + r] = base + (size_t)(f * P + g * NP + r) * row_bytes;
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 cur[2 * NP], nxt[2 * NP];
    const int tiles = row_bytes >> 10;
#pragma unroll
    for (int r = 0; r < 2 * NP; ++r) cur[r] = *(const uint4*)(row[r] + lane * 16);
    for (int t = 0; t < tiles; ++t) {
        const int no = (t + 1 < tiles ? t + 1 : t) * 1024 + lane * 16;
#pragma unroll
        for (int r = 0; r < 2 * NP; ++r) nxt[r] = *(const uint4*)(row[r] + no);
#pragma unroll
        for (int r = 0; r < 2 * NP; ++r) { acc.x ^= cur[r].x; acc.y ^= cur[r].y; acc.z ^= cur[r].z; acc.w ^= cur[r].w; cur[r] = nxt[r]; }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

// pattern 1b: frame-major sweep: wave i reads row i and row i + P (its partner in the next frame), memory order:
// every row is read twice, the second time P rows (4.7 MB) after the first - from the Infinity Cache, if it
// delivers on top of the HBM stream
template <int PF>
__global__ void k_read_pairs(const char* __restrict__ base, int n_rows, int row_bytes, int P, uint4* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave + P >= n_rows) return;
    const char* ra = base + (size_t)wave * row_bytes;
    const char* rb = base + (size_t)(wave + P) * row_bytes;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int tiles = row_bytes >> 10;
    uint4 ca = *(const uint4*)(ra + lane * 16), cb = *(const uint4*)(rb + lane * 16);
    for (int t = 0; t < tiles; ++t) {
        const int no = (t + 1 < tiles ? t + 1 : t) * 1024 + lane * 16;
        const uint4 na = *(const uint4*)(ra + no), nb = *(const uint4*)(rb + no);
        acc.x ^= ca.x ^ cb.x; acc.y ^= ca.y ^ cb.y; acc.z ^= ca.z ^ cb.z; acc.w ^= ca.w ^ cb.w;
        ca = na; cb = nb;
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

// pattern 3: copy - every word is read, a fraction `keep_256`/256 of the rows is written (row = 8 KiB),
// stores optionally non-temporal: the traffic mix of the merge+compaction pass
template <int NT, int REV = 0>
__global__ void k_copy_rows(const uint4* __restrict__ src, uint4* __restrict__ dst, int n_rows, int keep_256) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int waves = (gridDim.x * blockDim.x) >> 6;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    int out_row = 0;
    for (int r0 = wave; r0 < n_rows; r0 += waves) {
        const int r = REV ? n_rows - 1 - r0 : r0;          // REV: from the end of the buffer (what a previous read pass fetched last)
        const bool keep = ((r * 37) & 255) < keep_256;
        const uint4* p = src + (size_t)r * 512 + lane;
        uint4 v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (NT) { const v4u q = __builtin_nontemporal_load((const v4u*)(p + t * 64)); v[t] = make_uint4(q.x, q.y, q.z, q.w); }
            else v[t] = p[t * 64];
        }
        out_row = (int)(((long long)r * keep_256) >> 8);
        uint4* q = dst + (size_t)out_row * 512 + lane;
        if (keep) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (NT) { v4u w = {v[t].x, v[t].y, v[t].z, v[t].w}; __builtin_nontemporal_store(w, (v4u*)(q + t * 64)); }
                else q[t * 64] = v[t];
            }
        } else {                                       // dropped row: still read in full (the merge pass folds it)
            uint4 x = v[0];
#pragma unroll
            for (int t = 1; t < 8; ++t) { x.x ^= v[t].x; x.y ^= v[t].y; x.z ^= v[t].z; x.w ^= v[t].w; }
            if (x.x == 0x12345678u && x.w == 0x9abcdef0u) q[0] = x;
        }
    }
}

// pattern 4: the same copy with the merge kernel's decomposition: a wave owns ONE 1 KiB column tile of
// SLOTS consecutive by-patch rows (rows 4.7 MB apart), DEPTH pieces in flight
template <int SLOTS, int DEPTH, int NT>
__global__ void k_copy_coltile(const char* __restrict__ src, char* __restrict__ dst, int n_rows, int row_stride, int keep_256) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);      // 0..7
    const int g0 = blockIdx.x * SLOTS;
    const int F = row_stride, P = n_rows / row_stride;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s0 = 0; s0 < SLOTS; s0 += DEPTH) {
        uint4 v[DEPTH];
        int rr[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            int j = g0 + min(s0 + u, SLOTS - 1); if (j >= n_rows) j = n_rows - 1;      // (past the group's end: its last row again, an L2 hit)
            const int pp = j / F, f = j - pp * F;
            rr[u] = f * P + pp;
            const char* p = src + (size_t)rr[u] * 8192 + tile * 1024 + lane * 16;
            if (NT) { const v4u q = __builtin_nontemporal_load((const v4u*)p); v[u] = make_uint4(q.x, q.y, q.z, q.w); }
            else v[u] = *(const uint4*)p;
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int r = rr[u];
            if (((r * 37) & 255) < keep_256) {
                char* q = dst + (size_t)(((long long)r * keep_256) >> 8) * 8192 + tile * 1024 + lane * 16;
                if (NT) { v4u w = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(w, (v4u*)q); }
                else *(uint4*)q = v[u];
            } else { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if (acc.x == 0x12345678u && acc.w == 0x9abcdef0u) *(uint4*)dst = acc;
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

// `pre` (untimed) then `f` (timed), per repetition: the state a consumer finds right after a producer pass over its input
template <class G, class F>
static float time_after_us(G pre, F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    pre(); f(); pre(); f();
    CK(hipDeviceSynchronize());
    float total = 0;
    for (int i = 0; i < reps; ++i) {
        pre();
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        total += ms;
    }
    return total * 1e3f / reps;
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
#define ROWSR(R, TPB, ROT) { \
        const int waves = (n_rows + R - 1) / R, blocks = (waves * 64 + TPB - 1) / TPB; \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_rows<R, 1, ROT>), dim3(blocks), dim3(TPB), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, sink); }, 10); \
        printf("rows  R=%d pf=1 rot=%d tpb=%4d blocks=%6d : %7.1f us  %7.1f GB/s\n", R, ROT, TPB, blocks, us, (double)n_rows * row_bytes / us / 1e3); }
    ROWSR(5, 256, 0) ROWSR(5, 256, 1) ROWSR(5, 256, 2) ROWSR(4, 256, 0) ROWSR(4, 256, 1) ROWSR(4, 256, 2) ROWSR(5, 256, 0) ROWSR(5, 256, 1) ROWSR(5, 256, 2)
    for (int rep = 0; rep < 2; ++rep) {
        { const int waves = n_rows / 4, blocks = (waves * 64 + 255) / 256;
          float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_rows_pf<4>), dim3(blocks), dim3(256), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, sink); }, 10);
          printf("rows  R=4 pf=1 patch-fastest waves : %7.1f us  %7.1f GB/s\n", us, (double)n_rows * row_bytes / us / 1e3); }
        { const int waves = n_rows / 2, blocks = (waves * 64 + 255) / 256;
          float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_rows_pf<2>), dim3(blocks), dim3(256), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, sink); }, 10);
          printf("rows  R=2 pf=1 patch-fastest waves : %7.1f us  %7.1f GB/s\n", us, (double)n_rows * row_bytes / us / 1e3); }
        { const int waves = n_rows / 8, blocks = (waves * 64 + 255) / 256;
          float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_rows_pf<8>), dim3(blocks), dim3(256), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, sink); }, 10);
          printf("rows  R=8 pf=1 patch-fastest waves : %7.1f us  %7.1f GB/s\n", us, (double)n_rows * row_bytes / us / 1e3); }
    }
    { const int P = n_rows / F;
      for (int rep = 0; rep < 2; ++rep) {
        { const int waves = (P / 4) * (F - 1), blocks = (waves * 64 + 255) / 256;
          float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_pairs_np<4>), dim3(blocks), dim3(256), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, P, sink); }, 10);
          printf("pairs frame-major 4 patches x 2 frames per wave : %7.1f us  %7.1f GB/s (unique bytes)\n", us, (double)n_rows * row_bytes / us / 1e3); }
        { const int waves = (P / 2) * (F - 1), blocks = (waves * 64 + 255) / 256;
          float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_pairs_np<2>), dim3(blocks), dim3(256), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, P, sink); }, 10);
          printf("pairs frame-major 2 patches x 2 frames per wave : %7.1f us  %7.1f GB/s (unique bytes)\n", us, (double)n_rows * row_bytes / us / 1e3); }
      } }
    { const int P = n_rows / F;
      for (int tpb : {256, 512, 64}) {
        const int blocks = (n_rows * 64 + tpb - 1) / tpb;
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_pairs<1>), dim3(blocks), dim3(tpb), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, P, sink); }, 10);
        printf("pairs frame-major tpb=%4d : %7.1f us  %7.1f GB/s (unique bytes; every row read twice)\n", tpb, us, (double)n_rows * row_bytes / us / 1e3);
      } }
    if (argc > 2) return 0;
    ROWS(1, 256) ROWS(2, 256) ROWS(4, 256) ROWS(5, 256) ROWS(8, 256) ROWS(4, 128) ROWS(4, 512) ROWS(2, 512)
#define SEQ(NR, DEPTH, TPB, OV) { \
        const int waves = (n_rows + (NR - OV) - 1) / (NR - OV), blocks = (waves * 64 + TPB - 1) / TPB; \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_read_seq<NR, DEPTH>), dim3(blocks), dim3(TPB), 0, 0, (const char*)(flip ? a : b), n_rows, row_bytes, F, OV, sink); }, 10); \
        printf("seq   NR=%2d depth=%d tpb=%4d ov=%d blocks=%6d : %7.1f us  %7.1f GB/s (unique bytes)\n", NR, DEPTH, TPB, OV, blocks, us, (double)n_rows * row_bytes / us / 1e3); }
    SEQ(4, 2, 256, 0) SEQ(8, 2, 256, 0) SEQ(16, 2, 256, 0) SEQ(8, 4, 256, 0) SEQ(16, 4, 256, 0) SEQ(8, 2, 64, 0) SEQ(16, 4, 64, 0) SEQ(16, 8, 64, 0)
    SEQ(5, 2, 256, 1) SEQ(9, 2, 256, 1) SEQ(17, 2, 256, 1) SEQ(9, 4, 256, 1) SEQ(17, 4, 256, 1) SEQ(9, 4, 64, 1) SEQ(17, 4, 64, 1) SEQ(17, 8, 64, 1) SEQ(33, 4, 64, 1) SEQ(33, 8, 64, 1)
    char* c; CK(hipMalloc(&c, bytes));
#define COPY(NT, KEEP, BLOCKS) { \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_copy_rows<NT>), dim3(BLOCKS), dim3(256), 0, 0, (const uint4*)(flip ? a : b), (uint4*)c, n_rows, KEEP); }, 10); \
        const double moved = (double)n_rows * row_bytes * (1.0 + KEEP / 256.0); \
        printf("copy  nt=%d keep=%3d/256 blocks=%5d : %7.1f us  %7.1f GB/s (read + written bytes)\n", NT, KEEP, BLOCKS, us, moved / us / 1e3); }
#define COLT(SLOTS, DEPTH, NT, KEEP) { \
        dim3 grid((n_rows + SLOTS - 1) / SLOTS, 2); \
        float us = time_us([&] { flip ^= 1; hipLaunchKernelGGL((k_copy_coltile<SLOTS, DEPTH, NT>), grid, dim3(256), 0, 0, (const char*)(flip ? a : b), c, n_rows, F, KEEP); }, 10); \
        const double moved = (double)n_rows * row_bytes * (1.0 + KEEP / 256.0); \
        printf("colt  slots=%2d depth=%d nt=%d keep=%3d/256 : %7.1f us  %7.1f GB/s (read + written bytes)\n", SLOTS, DEPTH, NT, KEEP, us, moved / us / 1e3); }
    // the merge pass's real situation: the similarity pass has just read the whole source (ascending), the Infinity Cache holds
    // its last 256 MiB.  An event pair around one launch adds ~7 us of command-processor round trips to every line here
    // (the "pre only" line shows it): compare lines with each other, not with the back-to-back loops above.
#define WARM(NT, REV, KEEP, BLOCKS) { \
        auto pre = [&] { flip ^= 1; hipLaunchKernelGGL((k_read_flat<1, 0>), dim3(2048), dim3(256), 0, 0, (const uint4*)(flip ? a : b), n16, sink); }; \
        float us = time_after_us(pre, [&] { hipLaunchKernelGGL((k_copy_rows<NT, REV>), dim3(BLOCKS), dim3(256), 0, 0, (const uint4*)(flip ? a : b), (uint4*)c, n_rows, KEEP); }, 10); \
        float cold = time_after_us(pre, [&] { hipLaunchKernelGGL((k_copy_rows<NT, REV>), dim3(BLOCKS), dim3(256), 0, 0, (const uint4*)(flip ? b : a), (uint4*)c, n_rows, KEEP); }, 10); \
        const double moved = (double)n_rows * row_bytes * (1.0 + KEEP / 256.0); \
        printf("copy after a read pass of the SAME buffer: nt=%d rev=%d keep=%3d/256 blocks=%5d : %7.1f us  %7.1f GB/s | of the OTHER buffer: %7.1f us\n", NT, REV, KEEP, BLOCKS, us, moved / us / 1e3, cold); }
#define WARMCOLT(SLOTS, DEPTH, NT, KEEP) { \
        dim3 grid((n_rows + SLOTS - 1) / SLOTS, 2); \
        auto pre = [&] { flip ^= 1; hipLaunchKernelGGL((k_read_flat<1, 0>), dim3(2048), dim3(256), 0, 0, (const uint4*)(flip ? a : b), n16, sink); }; \
        float us = time_after_us(pre, [&] { hipLaunchKernelGGL((k_copy_coltile<SLOTS, DEPTH, NT>), grid, dim3(256), 0, 0, (const char*)(flip ? a : b), c, n_rows, F, KEEP); }, 10); \
        float cold = time_after_us(pre, [&] { hipLaunchKernelGGL((k_copy_coltile<SLOTS, DEPTH, NT>), grid, dim3(256), 0, 0, (const char*)(flip ? b : a), c, n_rows, F, KEEP); }, 10); \
        const double moved = (double)n_rows * row_bytes * (1.0 + KEEP / 256.0); \
        printf("colt after a read pass of the SAME buffer: slots=%2d depth=%d nt=%d keep=%3d/256 : %7.1f us  %7.1f GB/s | of the OTHER buffer: %7.1f us\n", SLOTS, DEPTH, NT, KEEP, us, moved / us / 1e3, cold); }
    { auto pre = [&] { flip ^= 1; hipLaunchKernelGGL((k_read_flat<1, 0>), dim3(2048), dim3(256), 0, 0, (const uint4*)(flip ? a : b), n16, sink); };
      float us = time_after_us(pre, [&] { hipLaunchKernelGGL((k_read_flat<1, 0>), dim3(1), dim3(64), 0, 0, (const uint4*)a, (size_t)64, sink); }, 10);
      printf("event pair around an (almost) empty launch after the read pass: %7.1f us\n", us);
      us = time_after_us(pre, [&] { hipLaunchKernelGGL((k_read_flat<1, 0>), dim3(2048), dim3(256), 0, 0, (const uint4*)(flip ? a : b), n16, sink); }, 10);
      float cold = time_after_us(pre, [&] { hipLaunchKernelGGL((k_read_flat<1, 0>), dim3(2048), dim3(256), 0, 0, (const uint4*)(flip ? b : a), n16, sink); }, 10);
      printf("flat read after a read pass of the SAME buffer: %7.1f us | of the OTHER buffer: %7.1f us\n", us, cold); }
    WARM(1, 0, 77, 2048) WARM(1, 1, 77, 2048) WARM(1, 1, 77, 4096) WARM(0, 1, 77, 2048) WARM(1, 1, 0, 2048) WARM(1, 0, 0, 2048)
    WARMCOLT(37, 4, 1, 77) WARMCOLT(37, 4, 0, 77)
    if (argc > 3) return 0;
    COLT(37, 4, 1, 77) COLT(36, 4, 1, 77) COLT(19, 4, 1, 77) COLT(32, 4, 1, 77) COLT(32, 4, 0, 77) COLT(32, 8, 1, 77) COLT(16, 4, 1, 77) COLT(64, 4, 1, 77) COLT(32, 2, 1, 77)
    COPY(0, 256, 2048) COPY(1, 256, 2048) COPY(0, 77, 2048) COPY(1, 77, 2048) COPY(1, 77, 4096) COPY(1, 77, 1024) COPY(1, 128, 2048)
    return 0;
}
