// A torch-free host of the C ABI: plain HIP runtime + libframefusion_hip.so, one merge step
// (ff_merge_step) on a deterministic [L, d] bf16 input, results printed for tests/test_gpu_abi_host.py
// to compare with the Python host on the same data.
//   hipcc --offload-arch=gfx950 -O2 -I include tests/abi_host/host.cpp -L framefusion_amd -lframefusion_hip
//         -Wl,-rpath,$PWD/framefusion_amd -o tests/abi_host/host
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "framefusion_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define FF(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: %s\n", #x, ff_error_string(rc_)); return 3; } } while (0)

static uint16_t bf16_of(float f) {            // f is exactly representable (multiples of 1/8 in +-4)
    uint32_t b; memcpy(&b, &f, 4);
    return (uint16_t)(b >> 16);
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 12, P = argc > 2 ? atoi(argv[2]) : 24, d = argc > 3 ? atoi(argv[3]) : 128;
    const int pre = 5, post = 7, L = pre + F * P + post;
    // deterministic activations on the dyadic grid: frame f of patch p is frame f-1 plus a small
    // step on a few features, so neighbouring frames are similar; text rows are unrelated
    std::vector<uint16_t> h((size_t)L * d);
    std::vector<int64_t> pt(L, -1);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<float> row(d);
    for (int i = 0; i < L; ++i) {
        const bool vis = i >= pre && i < pre + F * P;
        for (int c = 0; c < d; ++c) {
            float v;
            if (vis && i - pre >= P) {                                  // previous frame, same patch
                const uint16_t pb = h[(size_t)(i - P) * d + c];
                uint32_t b = (uint32_t)pb << 16; float pv; memcpy(&pv, &b, 4);
                const uint32_t r = rnd();
                v = pv + ((r % 16) == 0 ? ((r >> 4) % 2 ? 0.125f : -0.125f) : 0.f);
                if ((r % 97) == 0) v = (float)((int)((r >> 8) % 33) - 16) * 0.125f;     // a changed feature
            } else {
                v = (float)((int)(rnd() % 33) - 16) * 0.125f;
            }
            if (v > 4.f) v = 4.f; if (v < -4.f) v = -4.f;
            h[(size_t)i * d + c] = bf16_of(v);
        }
        if (vis) pt[i] = (i - pre) % P;
    }
    void *dh, *dout, *dpt, *dorder, *dinv, *dsim, *dmember, *ddst, *dkeep, *dstats, *dws, *dpt_out;
    const size_t wsb = ff_workspace_bytes(L, P);
    CK(hipMalloc(&dh, h.size() * 2)); CK(hipMalloc(&dout, h.size() * 2)); CK(hipMalloc(&dpt, L * 8)); CK(hipMalloc(&dpt_out, L * 8));
    CK(hipMalloc(&dorder, L * 4)); CK(hipMalloc(&dinv, L * 4)); CK(hipMalloc(&dsim, L * 4)); CK(hipMalloc(&dmember, L)); CK(hipMalloc(&ddst, L * 4));
    CK(hipMalloc(&dkeep, L)); CK(hipMalloc(&dstats, FF_STAT_WORDS * 8)); CK(hipMalloc(&dws, wsb));
    CK(hipMemcpy(dh, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpt, pt.data(), L * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dstats, 0, FF_STAT_WORDS * 8)); CK(hipMemset(dws, 0, wsb));
    hipStream_t st; CK(hipStreamCreate(&st));
    if (ff_abi_version() != FF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 4; }
    ff_aux_t aux[1] = {{dpt, dpt_out, 8, 1}};
    const double thr = 0.6015625;                                        // bf16(0.6), main.py:113
    FF(ff_merge_step(dh, /*addend=*/nullptr, dout, FF_BF16, L, d, L, (const int64_t*)dpt, P, /*order_valid=*/0, thr, /*sub=*/0.7, /*ratio_lb=*/0.1,
                     (int32_t*)dorder, (int32_t*)dinv, dsim, (uint8_t*)dmember, (int32_t*)ddst, (uint8_t*)dkeep, (int64_t*)dstats, nullptr,
                     /*seq=*/1, aux, 1, /*hint_pre=*/0, /*hint_frames=*/0, /*order_next=*/nullptr, /*inv_next=*/nullptr, dws, wsb, st));
    CK(hipStreamSynchronize(st));
    std::vector<int64_t> stats(FF_STAT_WORDS);
    std::vector<uint8_t> keep(L);
    CK(hipMemcpy(stats.data(), dstats, FF_STAT_WORDS * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(keep.data(), dkeep, L, hipMemcpyDeviceToHost));
    const int64_t l_out = stats[FF_STAT_LOUT];
    std::vector<uint16_t> out((size_t)l_out * d);
    std::vector<int64_t> pt_out(l_out);
    CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pt_out.data(), dpt_out, l_out * 8, hipMemcpyDeviceToHost));
    uint64_t fnv = 1469598103934665603ull;
    for (uint16_t v : out) { fnv ^= v; fnv *= 1099511628211ull; }
    uint64_t fnv_pt = 1469598103934665603ull;
    for (int64_t v : pt_out) { fnv_pt ^= (uint64_t)v; fnv_pt *= 1099511628211ull; }
    printf("L %d NV %lld FTN %lld COUNT %lld BRANCH %lld K %lld LOUT %lld\n", L, (long long)stats[FF_STAT_NV], (long long)stats[FF_STAT_FTN],
           (long long)stats[FF_STAT_COUNT], (long long)stats[FF_STAT_BRANCH], (long long)stats[FF_STAT_K], (long long)l_out);
    printf("HIDDEN_FNV %016llx\nPTYPE_FNV %016llx\nKEEP ", (unsigned long long)fnv, (unsigned long long)fnv_pt);
    for (int i = 0; i < L; ++i) putchar(keep[i] ? '1' : '0');
    putchar('\n');
    // the input, so the Python side runs on identical bytes
    FILE* f = fopen(argc > 4 ? argv[4] : "/tmp/abi_host_input.bin", "wb");
    if (!f) return 5;
    fwrite(h.data(), 2, h.size(), f);
    fclose(f);
    return 0;
}
