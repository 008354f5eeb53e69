// A torch-free host of the C ABI: plain HIP runtime + libframefusion_hip.so, one merge step
// (ff_merge_step) on a deterministic [L, d] bf16 input, results printed for tests/test_gpu_abi_host.py
// to compare with the Python host on the same data.
//   host F P d input.bin          one ff_merge_step, patch types as the only aux tensor
//   host F P d input.bin full     two merge calls through the call context (ff_ctx_merge): the first with an
//                                 `addend` (the residual add formed in the passes), M-RoPE style [3, L, 128] cos / sin
//                                 aux tensors and the frame-major layout hint; the second on the compacted output with
//                                 the maintained order (order_valid = 1), driven as plan / wait / apply with exactly sized
//                                 outputs; then the first call once more on TWO contexts and two streams at once
//                                 (submit, submit, collect, collect)
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


static uint64_t fnv_words16(const std::vector<uint16_t>& v) {
    uint64_t f = 1469598103934665603ull;
    for (uint16_t x : v) { f ^= x; f *= 1099511628211ull; }
    return f;
}
static uint64_t fnv_words64(const std::vector<int64_t>& v) {
    uint64_t f = 1469598103934665603ull;
    for (int64_t x : v) { f ^= (uint64_t)x; f *= 1099511628211ull; }
    return f;
}
// addend row i, column c: a grid value the Python side recomputes (tests/test_gpu_abi_host.py: addend_of)
static inline float addend_at(int i, int c) { return (float)((i * 31 + c * 17) % 17 - 8) * 0.125f; }
// rotary-like table value for (plane, position, column), on the grid
static inline float table_at(int which, int plane, int i, int c) { return (float)((which * 5 + plane * 7 + i * 3 + c) % 33 - 16) * 0.0625f; }

static int full_mode(int F, int P, int d, int pre, int L, const std::vector<uint16_t>& h, const std::vector<int64_t>& pt,
                     const char* path) {
    const int dh = 128, planes = 3;
    // hidden = base + addend exactly (all values multiples of 1/8 well inside bf16's exact range)
    std::vector<uint16_t> base((size_t)L * d), add((size_t)L * d), tab[2];
    for (int i = 0; i < L; ++i)
        for (int c = 0; c < d; ++c) {
            const uint32_t b = (uint32_t)h[(size_t)i * d + c] << 16;
            float v; memcpy(&v, &b, 4);
            const float a = addend_at(i, c);
            base[(size_t)i * d + c] = bf16_of(v - a);
            add[(size_t)i * d + c] = bf16_of(a);
        }
    for (int w = 0; w < 2; ++w) {
        tab[w].resize((size_t)planes * L * dh);
        for (int p = 0; p < planes; ++p)
            for (int i = 0; i < L; ++i)
                for (int c = 0; c < dh; ++c) tab[w][((size_t)p * L + i) * dh + c] = bf16_of(table_at(w, p, i, c));
    }
    void *dbase, *dadd, *dout, *dout2, *dpt, *dpt_out, *dpt_out2, *dtab[2], *dtab_out[2], *dtab_out2[2];
    void *dorder, *dorder_next, *dinv, *dinv_next, *dsim, *dmember, *ddst, *dkeep, *dstats, *dws;
    int64_t* stats_host;
    const size_t wsb = ff_workspace_bytes(L, P), hb = (size_t)L * d * 2, tb = (size_t)planes * L * dh * 2;
    CK(hipMalloc(&dbase, hb)); CK(hipMalloc(&dadd, hb)); CK(hipMalloc(&dout, hb)); CK(hipMalloc(&dout2, hb));
    CK(hipMalloc(&dpt, L * 8)); CK(hipMalloc(&dpt_out, L * 8)); CK(hipMalloc(&dpt_out2, L * 8));
    for (int w = 0; w < 2; ++w) { CK(hipMalloc(&dtab[w], tb)); CK(hipMalloc(&dtab_out[w], tb)); CK(hipMalloc(&dtab_out2[w], tb)); }
    CK(hipMalloc(&dorder, L * 4)); CK(hipMalloc(&dorder_next, L * 4)); CK(hipMalloc(&dinv, L * 4)); CK(hipMalloc(&dinv_next, L * 4));
    CK(hipMalloc(&dsim, L * 4)); CK(hipMalloc(&dmember, L)); CK(hipMalloc(&ddst, L * 4)); CK(hipMalloc(&dkeep, L));
    CK(hipMalloc(&dstats, FF_STAT_WORDS * 8)); CK(hipMalloc(&dws, wsb));
    CK(hipHostMalloc((void**)&stats_host, FF_HOST_WORDS * 8, hipHostMallocDefault));
    memset(stats_host, 0, FF_HOST_WORDS * 8);
    CK(hipMemcpy(dbase, base.data(), hb, hipMemcpyHostToDevice)); CK(hipMemcpy(dadd, add.data(), hb, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpt, pt.data(), L * 8, hipMemcpyHostToDevice));
    for (int w = 0; w < 2; ++w) CK(hipMemcpy(dtab[w], tab[w].data(), tb, hipMemcpyHostToDevice));
    CK(hipMemset(dstats, 0, FF_STAT_WORDS * 8)); CK(hipMemset(dws, 0, wsb));
    hipStream_t st; CK(hipStreamCreate(&st));

    ff_ctx_t ctx; memset(&ctx, 0, sizeof ctx);
    ctx.cap = L; ctx.order = (int32_t*)dorder; ctx.order_next = (int32_t*)dorder_next; ctx.inv = (int32_t*)dinv; ctx.inv_next = (int32_t*)dinv_next;
    ctx.sim = dsim; ctx.member = (uint8_t*)dmember; ctx.dst = (int32_t*)ddst; ctx.keep = (uint8_t*)dkeep;
    ctx.stats = (int64_t*)dstats; ctx.stats_host = stats_host; ctx.ws = dws; ctx.ws_bytes = wsb;

    ff_merge_call_t call; memset(&call, 0, sizeof call);
    ff_merge_result_t res[2];
    call.hidden = dbase; call.addend = dadd; call.hidden_out = dout; call.patch_type = (const int64_t*)dpt;
    call.dtype = FF_BF16; call.L = L; call.d = d; call.L_cap = L; call.patch_num = P; call.order_valid = 0;
    call.threshold = 0.6015625; call.sub = 0.7; call.ratio_lb = 0.01; call.force_k = -1; call.fold = FF_FOLD_SEQUENTIAL;
    call.hint_pre = pre; call.hint_frames = F; call.stream = st; call.n_aux = 3;
    call.aux[0] = ff_aux_t{dpt, dpt_out, 8, 1};
    call.aux[1] = ff_aux_t{dtab[0], dtab_out[0], dh * 2, planes};
    call.aux[2] = ff_aux_t{dtab[1], dtab_out[1], dh * 2, planes};
    // the first call as submit + collect (ABI v9: everything enqueued, the wait split off - what a host that keeps two samples in
    // flight calls; here with one sample: a collect before any submit and a second submit in between must be refused)
    if (ff_ctx_merge_collect(&ctx, &call, &res[0]) != FF_ERR_STATE) { fprintf(stderr, "collect without submit was not refused\n"); return 6; }
    FF(ff_ctx_merge_submit(&ctx, &call));
    if (ff_ctx_merge_wait(&ctx, &call, &res[0]) != FF_ERR_STATE) { fprintf(stderr, "wait on a submitted call was not refused\n"); return 6; }
    FF(ff_ctx_merge_collect(&ctx, &call, &res[0]));
    const int64_t l1 = res[0].l_out;
    std::vector<uint8_t> keep1(L);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(keep1.data(), dkeep, L, hipMemcpyDeviceToHost));
    // second call: the compacted sequence, its patch types and tables, the order the first call left (order_valid = 1)
    ff_merge_call_t call2 = call;
    call2.hidden = dout; call2.addend = nullptr; call2.hidden_out = dout2; call2.patch_type = (const int64_t*)dpt_out;
    call2.L = l1; call2.L_cap = l1; call2.order_valid = 1; call2.hint_frames = 0; call2.sub = 0.55;
    call2.aux[0] = ff_aux_t{dpt_out, dpt_out2, 8, 1};
    // the first call wrote its tables as [planes, L_cap = L, dh], of which the first l1 rows of every plane are the second
    // call's [planes, l1, dh] input: passed as they lie, the planes L rows apart (ff_aux_t.src_outer_bytes, ABI v8)
    call2.aux[1] = ff_aux_t{dtab_out[0], dtab_out2[0], dh * 2, planes, (int64_t)L * dh * 2};
    call2.aux[2] = ff_aux_t{dtab_out[1], dtab_out2[1], dh * 2, planes, (int64_t)L * dh * 2};
    const int64_t swaps_before = ctx.swaps;
    // ... through the exact-output flow: K1, then plan + wait for l_out, size the outputs to it (L_cap = l_out: the aux
    // planes come out l_out rows apart), then the merge kernel
    FF(ff_ctx_merge_begin(&ctx, &call2));
    if (ff_ctx_merge_apply(&ctx, &call2, &res[1]) != FF_ERR_STATE) { fprintf(stderr, "apply before wait was not refused\n"); return 6; }
    FF(ff_ctx_merge_wait(&ctx, &call2, &res[1]));
    const int64_t l2 = res[1].l_out;
    call2.L_cap = l2;
    FF(ff_ctx_merge_apply(&ctx, &call2, &res[1]));
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> o1((size_t)l1 * d), o2((size_t)l2 * d), t2((size_t)planes * l2 * dh);
    std::vector<int64_t> p2(l2);
    std::vector<uint8_t> keep2(l1);
    CK(hipMemcpy(o1.data(), dout, o1.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(keep2.data(), dkeep, l1, hipMemcpyDeviceToHost));
    printf("L %d ORDER_REUSED %d\n", L, (int)(ctx.swaps == swaps_before + (l2 != l1) ? 1 : 0));
    for (int c = 0; c < 2; ++c)
        printf("CALL %d NV %lld FTN %lld COUNT %lld BRANCH %lld K %lld LOUT %lld UNHINTED %lld\n", c, (long long)res[c].nv, (long long)res[c].ftn,
               (long long)res[c].count, (long long)res[c].branch, (long long)res[c].k, (long long)res[c].l_out, (long long)res[c].unhinted);
    printf("HIDDEN1_FNV %016llx\n", (unsigned long long)fnv_words16(o1));
    if (l2 != l1) {
        CK(hipMemcpy(o2.data(), dout2, o2.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(p2.data(), dpt_out2, l2 * 8, hipMemcpyDeviceToHost));
        printf("HIDDEN2_FNV %016llx\nPTYPE2_FNV %016llx\n", (unsigned long long)fnv_words16(o2), (unsigned long long)fnv_words64(p2));
        for (int w = 0; w < 2; ++w) {
            for (int p = 0; p < planes; ++p)
                CK(hipMemcpy(t2.data() + (size_t)p * l2 * dh, (char*)dtab_out2[w] + (size_t)p * l2 * dh * 2, (size_t)l2 * dh * 2, hipMemcpyDeviceToHost));
            printf("TABLE%d_FNV %016llx\n", w, (unsigned long long)fnv_words16(t2));
        }
    }
    printf("KEEP1 "); for (int i = 0; i < L; ++i) putchar(keep1[i] ? '1' : '0'); putchar('\n');
    printf("KEEP2 "); for (int i = 0; i < l1; ++i) putchar(keep2[i] ? '1' : '0'); putchar('\n');
    // Two samples in flight from this one host thread (ABI v9): the first call again, on context A and on a second context B
    // with its own scratch and its own stream - submit(A), submit(B), collect(A), collect(B).  Same input, so both must give
    // the first call's output again.
    {
        void *border, *border_next, *binv, *binv_next, *bsim, *bmember, *bdst, *bkeep, *bstats, *bws, *bout, *bpt_out, *btab_out[2];
        int64_t* bstats_host;
        CK(hipMalloc(&border, L * 4)); CK(hipMalloc(&border_next, L * 4)); CK(hipMalloc(&binv, L * 4)); CK(hipMalloc(&binv_next, L * 4));
        CK(hipMalloc(&bsim, L * 4)); CK(hipMalloc(&bmember, L)); CK(hipMalloc(&bdst, L * 4)); CK(hipMalloc(&bkeep, L));
        CK(hipMalloc(&bstats, FF_STAT_WORDS * 8)); CK(hipMalloc(&bws, wsb)); CK(hipMalloc(&bout, hb)); CK(hipMalloc(&bpt_out, L * 8));
        for (int w = 0; w < 2; ++w) CK(hipMalloc(&btab_out[w], tb));
        CK(hipHostMalloc((void**)&bstats_host, FF_HOST_WORDS * 8, hipHostMallocDefault));
        memset(bstats_host, 0, FF_HOST_WORDS * 8);
        CK(hipMemset(bstats, 0, FF_STAT_WORDS * 8)); CK(hipMemset(bws, 0, wsb));
        hipStream_t st_b; CK(hipStreamCreateWithFlags(&st_b, hipStreamNonBlocking));
        ff_ctx_t ctx_b; memset(&ctx_b, 0, sizeof ctx_b);
        ctx_b.cap = L; ctx_b.order = (int32_t*)border; ctx_b.order_next = (int32_t*)border_next; ctx_b.inv = (int32_t*)binv;
        ctx_b.inv_next = (int32_t*)binv_next; ctx_b.sim = bsim; ctx_b.member = (uint8_t*)bmember; ctx_b.dst = (int32_t*)bdst;
        ctx_b.keep = (uint8_t*)bkeep; ctx_b.stats = (int64_t*)bstats; ctx_b.stats_host = bstats_host; ctx_b.ws = bws; ctx_b.ws_bytes = wsb;
        ff_merge_call_t call_b = call;
        call_b.hidden_out = bout; call_b.stream = st_b;
        call_b.aux[0] = ff_aux_t{dpt, bpt_out, 8, 1};
        call_b.aux[1] = ff_aux_t{dtab[0], btab_out[0], dh * 2, planes};
        call_b.aux[2] = ff_aux_t{dtab[1], btab_out[1], dh * 2, planes};
        ff_merge_result_t ra, rb;
        FF(ff_ctx_reset(&ctx, st));
        FF(ff_ctx_merge_submit(&ctx, &call));
        FF(ff_ctx_merge_submit(&ctx_b, &call_b));
        FF(ff_ctx_merge_collect(&ctx, &call, &ra));
        FF(ff_ctx_merge_collect(&ctx_b, &call_b, &rb));
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st_b));
        std::vector<uint16_t> oa((size_t)l1 * d), ob((size_t)l1 * d);
        int ok = ra.l_out == l1 && rb.l_out == l1;
        if (ok) {
            CK(hipMemcpy(oa.data(), dout, oa.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ob.data(), bout, ob.size() * 2, hipMemcpyDeviceToHost));
            ok = fnv_words16(oa) == fnv_words16(o1) && fnv_words16(ob) == fnv_words16(o1);
        }
        printf("PAIR_OK %d\n", ok);
    }
    FILE* f = fopen(path, "wb");
    if (!f) return 5;
    fwrite(h.data(), 2, h.size(), f);
    fclose(f);
    return 0;
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
    if (argc > 5 && !strcmp(argv[5], "full")) return full_mode(F, P, d, pre, L, h, pt, argv[4]);
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
