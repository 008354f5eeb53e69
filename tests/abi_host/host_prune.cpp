// A torch-free host of the C ABI, the PRUNE half and the layout entry points (tests/test_gpu_abi_host.py):
//   host_prune prune S d H H_kv dh num start n_img k dir
//       importance of the last `num` queries from the un-repeated GQA keys + the prune (framefusion/main.py:61-101 fed by
//       utils.py:27-57), twice: ff_ctx_last_query_importance + ff_ctx_prune, then ff_ctx_prune_from_qk (one crossing) on a reset
//       context.  Inputs are generated here and written to dir/{hidden,q,k}.bin for the Python side.
//   host_prune layout F P d pre post dir
//       ff_token_span on prompt ids -> ff_fill_patch_type -> one merge call (ff_ctx_merge_submit / _collect) on that patch_type.
//   hipcc --offload-arch=gfx950 -O2 -I include tests/abi_host/host_prune.cpp -L framefusion_amd -lframefusion_hip
//         -Wl,-rpath,$PWD/framefusion_amd -o host_prune
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "framefusion_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define FF(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: %s\n", #x, ff_error_string(rc_)); return 3; } } while (0)

static uint16_t bf16_of(float f) { uint32_t b; memcpy(&b, &f, 4); return (uint16_t)(b >> 16); }      // (exactly representable values only)
// position-weighted sum mod 2^64 (numpy computes the same in one vectorised line: large outputs)
static uint64_t fnv16(const std::vector<uint16_t>& v) { uint64_t f = 0; for (size_t i = 0; i < v.size(); ++i) f += (uint64_t)(i + 1) * v[i]; return f; }
static uint64_t fnv64(const std::vector<int64_t>& v) { uint64_t f = 1469598103934665603ull; for (int64_t x : v) { f ^= (uint64_t)x; f *= 1099511628211ull; } return f; }
static uint64_t fnv8(const std::vector<uint8_t>& v) { uint64_t f = 1469598103934665603ull; for (uint8_t x : v) { f ^= x; f *= 1099511628211ull; } return f; }

// multiples of 1/8 in [-2, 2): a hash of (tag, index)
static std::vector<uint16_t> grid_values(size_t n, uint32_t tag) {
    std::vector<uint16_t> v(n);
    uint32_t x = tag * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; v[i] = bf16_of((float)((int)((x >> 11) & 31) - 16) * 0.125f); }
    return v;
}
static int dump(const std::string& path, const void* p, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return 5;
    fwrite(p, 1, bytes, f);
    fclose(f);
    return 0;
}

struct Scratch {
    ff_ctx_t ctx;
    int init(int64_t cap, hipStream_t) {
        memset(&ctx, 0, sizeof ctx);
        cap = (cap + 63) & ~(int64_t)63;
        const size_t wsb = ff_workspace_bytes(cap, 1);
        void* p;
        ctx.cap = cap;
        CK(hipMalloc(&p, cap * 4)); ctx.order = (int32_t*)p;
        CK(hipMalloc(&p, cap * 4)); ctx.order_next = (int32_t*)p;
        CK(hipMalloc(&p, cap * 4)); ctx.inv = (int32_t*)p;
        CK(hipMalloc(&p, cap * 4)); ctx.inv_next = (int32_t*)p;
        CK(hipMalloc(&p, cap * 4)); ctx.sim = p;
        CK(hipMalloc(&p, cap)); ctx.member = (uint8_t*)p;
        CK(hipMalloc(&p, cap * 4)); ctx.dst = (int32_t*)p;
        CK(hipMalloc(&p, cap)); ctx.keep = (uint8_t*)p;
        CK(hipMalloc(&p, FF_STAT_WORDS * 8)); ctx.stats = (int64_t*)p;
        CK(hipMemset(p, 0, FF_STAT_WORDS * 8));
        CK(hipMalloc(&p, wsb)); ctx.ws = p; ctx.ws_bytes = wsb;
        CK(hipMemset(p, 0, wsb));
        ctx.stats_host = (int64_t*)ff_host_alloc(FF_HOST_WORDS * 8);         // coherent: outputs may go by mail
        return ctx.stats_host ? 0 : 2;
    }
};

static int prune_mode(int argc, char** argv) {
    if (argc < 12) return 1;
    const int64_t S = atoll(argv[2]), d = atoll(argv[3]), H = atoll(argv[4]), Hk = atoll(argv[5]), dh = atoll(argv[6]), num = atoll(argv[7]);
    const int64_t start = atoll(argv[8]), n_img = atoll(argv[9]), k = atoll(argv[10]);
    const std::string dir = argv[11];
    const int64_t l_out = S - n_img + k;
    std::vector<uint16_t> hidden = grid_values((size_t)S * d, 1), q = grid_values((size_t)H * num * dh, 2), key = grid_values((size_t)Hk * S * dh, 3);
    if (dump(dir + "/hidden.bin", hidden.data(), hidden.size() * 2) || dump(dir + "/q.bin", q.data(), q.size() * 2) || dump(dir + "/k.bin", key.data(), key.size() * 2)) return 5;
    std::vector<int64_t> ids(S);
    for (int64_t i = 0; i < S; ++i) ids[i] = i;
    hipStream_t st; CK(hipStreamCreate(&st));
    Scratch sc;
    if (sc.init(S, st)) return 2;
    void *dh_, *dq, *dk, *dout[2], *dids, *dids_out[2], *lqws;
    const size_t lqb = ff_last_query_workspace_bytes(FF_BF16, H, num, S, dh);
    CK(hipMalloc(&dh_, hidden.size() * 2)); CK(hipMalloc(&dq, q.size() * 2)); CK(hipMalloc(&dk, key.size() * 2));
    CK(hipMalloc(&dids, S * 8)); CK(hipMalloc(&lqws, lqb));
    for (int x = 0; x < 2; ++x) { CK(hipMalloc(&dout[x], (size_t)l_out * d * 2)); CK(hipMalloc(&dids_out[x], l_out * 8)); }
    CK(hipMemcpy(dh_, hidden.data(), hidden.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dk, key.data(), key.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dids, ids.data(), S * 8, hipMemcpyHostToDevice));
    const double scale = 1.0 / sqrt((double)dh);

    ff_prune_call_t pc; memset(&pc, 0, sizeof pc);
    pc.hidden = dh_; pc.hidden_out = dout[0]; pc.dtype = FF_BF16; pc.S = S; pc.d = d; pc.L_cap = l_out;
    pc.start = start; pc.n_img = n_img; pc.k = k; pc.stream = st; pc.n_aux = 1;
    pc.aux[0] = ff_aux_t{dids, dids_out[0], 8, 1, 0};
    // (a) the attention hook, then the prune call that consumes its importance and select tables
    FF(ff_ctx_last_query_importance(&sc.ctx, dq, dk, FF_BF16, H, Hk, num, S, dh, 0, 0, scale, 1, nullptr, sc.ctx.sim, start, n_img, k, lqws, lqb, st));
    pc.attn_w = sc.ctx.sim; pc.w_dtype = FF_BF16; pc.H = 1; pc.num = 1; pc.tables_ready = 1;
    // a short output buffer is an argument error, not an out-of-bounds write
    pc.L_cap = l_out - 1;
    if (l_out > 0 && ff_ctx_prune(&sc.ctx, &pc) != FF_ERR_ARG) { fprintf(stderr, "a short L_cap was not refused\n"); return 6; }
    pc.L_cap = l_out;
    FF(ff_ctx_prune(&sc.ctx, &pc));
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> imp(S), oa((size_t)l_out * d), ob((size_t)l_out * d);
    std::vector<int64_t> ka(l_out), kb(l_out);
    std::vector<uint8_t> keep(S);
    CK(hipMemcpy(imp.data(), sc.ctx.sim, S * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(oa.data(), dout[0], oa.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ka.data(), dids_out[0], l_out * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(keep.data(), sc.ctx.keep, S, hipMemcpyDeviceToHost));
    // (b) the same as ONE crossing on the reset context
    FF(ff_ctx_reset(&sc.ctx, st));
    ff_lq_args_t lq; memset(&lq, 0, sizeof lq);
    lq.q_last = dq; lq.k = dk; lq.dtype = FF_BF16; lq.H = H; lq.H_kv = Hk; lq.num = num; lq.dh = dh; lq.scale = scale; lq.causal = 1;
    lq.ws = lqws; lq.ws_bytes = lqb;
    ff_prune_call_t pb = pc;
    pb.hidden_out = dout[1]; pb.aux[0].dst = dids_out[1]; pb.attn_w = nullptr; pb.tables_ready = 0;
    FF(ff_ctx_prune_from_qk(&sc.ctx, &pb, &lq));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ob.data(), dout[1], ob.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(kb.data(), dids_out[1], l_out * 8, hipMemcpyDeviceToHost));
    // the workspace protocol: the select tables are clean again
    std::vector<uint8_t> ws(sc.ctx.ws_bytes);
    CK(hipMemcpy(ws.data(), sc.ctx.ws, ws.size(), hipMemcpyDeviceToHost));
    // (the level-0 table at the front, the per-slice level-1 tables down from the end; what lies between is the plan's tag and
    // granule words, which are not zero by design)
    size_t dirty = 0;
    const size_t l0_bytes = 16 * 260 * 4, tail = ((size_t)(sc.ctx.cap + 4095) / 4096 + 1) * 65536 * 4;
    for (size_t i = 0; i < l0_bytes; ++i) dirty += ws[i] != 0;
    for (size_t i = ws.size() - (tail < ws.size() ? tail : 0); i < ws.size(); ++i) dirty += ws[i] != 0;
    printf("L_OUT %lld WS_DIRTY_BYTES %zu CTX_DIRTY %lld\n", (long long)l_out, dirty, (long long)sc.ctx.dirty);
    printf("IMP_FNV %016llx\nKEEP_FNV %016llx\n", (unsigned long long)fnv16(imp), (unsigned long long)fnv8(keep));
    printf("HIDDEN_A_FNV %016llx\nHIDDEN_B_FNV %016llx\n", (unsigned long long)fnv16(oa), (unsigned long long)fnv16(ob));
    printf("KEPT_A_FNV %016llx\nKEPT_B_FNV %016llx\n", (unsigned long long)fnv64(ka), (unsigned long long)fnv64(kb));
    return 0;
}

static int layout_mode(int argc, char** argv) {
    if (argc < 8) return 1;
    const int F = atoi(argv[2]), P = atoi(argv[3]), d = atoi(argv[4]), pre = atoi(argv[5]), post = atoi(argv[6]);
    const std::string dir = argv[7];
    const int L = pre + F * P + post;
    const int64_t kImageToken = 151655;                      // (any id: the packers look it up in the config)
    std::vector<int64_t> ids(L);
    for (int i = 0; i < L; ++i) ids[i] = (i >= pre && i < pre + F * P) ? kImageToken : 1000 + i % 97;
    // frames that drift slowly: frame f of patch p = frame f - 1 with a few columns stepped
    std::vector<uint16_t> h = grid_values((size_t)L * d, 7);
    for (int i = pre + P; i < pre + F * P; ++i)
        for (int c = 0; c < d; ++c)
            if ((i * 7 + c) % 5 != 0) h[(size_t)i * d + c] = h[(size_t)(i - P) * d + c];       // (four columns of five keep the previous frame's value)
    if (dump(dir + "/hidden.bin", h.data(), h.size() * 2)) return 5;
    hipStream_t st; CK(hipStreamCreate(&st));
    Scratch sc;
    if (sc.init(L, st)) return 2;
    void *dids, *dpt, *dpt_out, *dh_, *dout;
    int64_t* span = (int64_t*)ff_host_alloc(8 * 8);
    CK(hipMalloc(&dids, L * 8)); CK(hipMalloc(&dpt, L * 8)); CK(hipMalloc(&dpt_out, L * 8)); CK(hipMalloc(&dh_, h.size() * 2)); CK(hipMalloc(&dout, h.size() * 2));
    CK(hipMemcpy(dids, ids.data(), L * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dh_, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    // llava_video/modeling_llava_video.py:332-336: where(ids == image token) -> first / last -> the patch_type list
    FF(ff_token_span((const int64_t*)dids, L, kImageToken, span, st));
    CK(hipStreamSynchronize(st));
    const int64_t first = span[0], last = span[1], count = span[2];
    if (first != pre || last != pre + F * P - 1 || count != F * P) { fprintf(stderr, "span %lld %lld %lld\n", (long long)first, (long long)last, (long long)count); return 6; }
    ff_segment_t seg{(int32_t)first, (int32_t)count, 0, (int32_t)P};
    FF(ff_fill_patch_type((int64_t*)dpt, L, &seg, 1, st));
    CK(hipStreamSynchronize(st));
    std::vector<int64_t> pt(L);
    CK(hipMemcpy(pt.data(), dpt, L * 8, hipMemcpyDeviceToHost));
    // one merge call on that patch_type, everything enqueued by one crossing (the one-launch kernel when it fits)
    ff_merge_call_t call; memset(&call, 0, sizeof call);
    ff_merge_result_t res;
    call.hidden = dh_; call.hidden_out = dout; call.patch_type = (const int64_t*)dpt; call.dtype = FF_BF16; call.L = L; call.d = d; call.L_cap = L;
    call.patch_num = P; call.threshold = 0.6015625; call.sub = 0.7; call.ratio_lb = 0.1; call.force_k = -1; call.fold = FF_FOLD_SEQUENTIAL;
    call.hint_pre = first; call.hint_frames = count / P; call.stream = st; call.n_aux = 1;
    call.aux[0] = ff_aux_t{dpt, dpt_out, 8, 1, 0};
    const int one = ff_ctx_merge_one_launch(&sc.ctx, &call);
    FF(ff_ctx_merge_submit(&sc.ctx, &call));
    FF(ff_ctx_merge_collect(&sc.ctx, &call, &res));
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> o((size_t)res.l_out * d);
    std::vector<int64_t> po(res.l_out);
    CK(hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(po.data(), dpt_out, res.l_out * 8, hipMemcpyDeviceToHost));
    printf("L %d ONE_LAUNCH %d APPLIED %lld LOUT %lld BRANCH %lld COUNT %lld UNHINTED %lld\n", L, one, (long long)res.applied, (long long)res.l_out,
           (long long)res.branch, (long long)res.count, (long long)res.unhinted);
    printf("PTYPE_FNV %016llx\nHIDDEN_FNV %016llx\nPTYPE_OUT_FNV %016llx\n", (unsigned long long)fnv64(pt), (unsigned long long)fnv16(o), (unsigned long long)fnv64(po));
    // the same call once more with the outputs BY MAIL (what the Python host does for exactly sized outputs): launch without any
    // output, mail buffers of a wrong length (slot 1: refused), read l_out from the result block the kernel publishes behind its
    // barrier, hand over buffers of exactly l_out rows (ff_ctx_merge_apply mails slot 2 and reads the acknowledgement: no launch)
    FF(ff_ctx_reset(&sc.ctx, st));
    void *dout2, *dpt_out2;
    CK(hipMalloc(&dout2, h.size() * 2)); CK(hipMalloc(&dpt_out2, L * 8));
    CK(hipMemset(dout2, 0, h.size() * 2)); CK(hipMemset(dpt_out2, 0, L * 8));
    ff_merge_call_t late = call;
    ff_merge_result_t res2;
    late.late_outputs = 1; late.hidden_out = nullptr; late.L_cap = 0; late.n_aux = 0;
    const int one2 = ff_ctx_merge_one_launch(&sc.ctx, &late);
    int slot = 0;
    if (one2) {
        FF(ff_ctx_merge_submit(&sc.ctx, &late));
        late.hidden_out = dout2; late.L_cap = res.l_out + 1; late.n_aux = 1;
        late.aux[0] = ff_aux_t{dpt, dpt_out2, 8, 1, 0};
        FF(ff_ctx_merge_mail(&sc.ctx, &late));                 // a guess that does not come true
        FF(ff_ctx_merge_collect(&sc.ctx, &late, &res2));
        if (res2.applied != 2 || res2.l_out != res.l_out) { fprintf(stderr, "late collect: applied %lld l_out %lld\n", (long long)res2.applied, (long long)res2.l_out); return 7; }
        late.L_cap = res2.l_out;
        FF(ff_ctx_merge_apply(&sc.ctx, &late, &res2));
        slot = (int)(sc.ctx.stats_host[FF_STAT_ACK] & 3);
        CK(hipStreamSynchronize(st));
        std::vector<uint16_t> o2((size_t)res2.l_out * d);
        std::vector<int64_t> po2(res2.l_out);
        CK(hipMemcpy(o2.data(), dout2, o2.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(po2.data(), dpt_out2, res2.l_out * 8, hipMemcpyDeviceToHost));
        printf("LATE_HIDDEN_FNV %016llx\nLATE_PTYPE_OUT_FNV %016llx\n", (unsigned long long)fnv16(o2), (unsigned long long)fnv64(po2));
    }
    printf("LATE_ONE_LAUNCH %d\nLATE_MAIL_SLOT %d\n", one2, slot);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "prune")) return prune_mode(argc, argv);
    if (argc > 1 && !strcmp(argv[1], "layout")) return layout_mode(argc, argv);
    fprintf(stderr, "usage: host_prune prune|layout ...\n");
    return 1;
}
