// C-ABI glue: version, error strings, workspace sizing and the fused merge step that enqueues
// K0 -> K1 -> K2+K3 -> K4 from a single host call (one FrameFusion.forward merge call,
// framefusion/main.py:104-138).
#include <string.h>

#include "ff_common.h"
#include "ff_resident.h"
#include "ff_source_hash.h"

namespace ff {
int launch_plan_merge(const void* sim, int dtype, const int32_t* order, const int32_t* inv, int64_t L, double thr,
                      double sub, double ratio_lb, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, size_t ws_bytes, bool have_tables, int64_t* host_mapped, int64_t seq,
                      hipStream_t st, long long force_k);
int launch_plan_prune(const void* imp, int dtype, int64_t S, int64_t start, int64_t n_img, int64_t k,
                      uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws, size_t ws_bytes,
                      bool have_tables, hipStream_t st);
size_t plan_ws_front_bytes(int64_t L);
size_t plan_ws_tail_bytes(int64_t L);
int32_t* ws_scratch_ints(void* ws, int64_t L);
int* ws_l0(void* ws);
int* ws_t16_end(void* ws, size_t ws_bytes);
void table_regions(void* ws, size_t ws_bytes, int64_t L, void** a, size_t* a_bytes, void** b, size_t* b_bytes);
int zero_tables(void* ws, size_t ws_bytes, int64_t L, hipStream_t st);
int launch_merge_compact(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                         const int32_t* order, const uint8_t* member, int fold, const int32_t* dst,
                         const uint8_t* keep, const ff_aux_t* aux_host, int n_aux, int32_t* order_next,
                         int32_t* inv_next, int64_t* stats, hipStream_t st, bool skip_identity, void* zero_a,
                         size_t zero_a_bytes, const void* zero_keys, int64_t zero_n, int zero_key_dt, int* t16_end,
                         const int32_t* src, int64_t l_out, int64_t guard_lout = -1);
int launch_similarity_any(const void* hidden, const void* addend, int dtype, int64_t L, int64_t d, const int64_t* ptype,
                          int32_t* order, int32_t* inv, int64_t* stats, void* sim, int* l0, int* t16_end, double thr,
                          int64_t hint_pre, int64_t hint_patches, int64_t hint_frames, hipStream_t st);
int gather_mask(const void* mask, void* out, int64_t elem_bytes, int64_t L, int64_t L_cap, const int32_t* dst,
                const int64_t* stats, int32_t* scratch, ff_stream_t stream);
int launch_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S, void* importance,
                     int64_t lo, int64_t hi, int* l0, int* t16_end, hipStream_t st);
}  // namespace ff

extern "C" int ff_abi_version(void) { return FF_ABI_VERSION; }
// (the marker lets a host read the stamp from the FILE, without loading it: a process that has already dlopen()ed a stale
// copy gets the same handle back for the same path, however often the file is rebuilt)
static const char kSourceStamp[] = "FFSRCHASH:" FF_SOURCE_HASH;
extern "C" const char* ff_source_hash(void) { return kSourceStamp + 10; }

extern "C" const char* ff_error_string(int code) {
    switch (code) {
        case FF_OK: return "ok";
        case FF_ERR_ARG: return "bad argument (null pointer, negative size or unknown dtype)";
        case FF_ERR_ALIGN: return "pointer or row size not 16-byte aligned";
        case FF_ERR_UNSUPPORTED: return "size outside the supported range";
        case FF_ERR_WORKSPACE: return "workspace too small (see ff_workspace_bytes)";
        case FF_ERR_DEVICE: return "a device-side check failed or the result block was never published";
        case FF_ERR_STATE: return "context call out of order (finish without begin)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" size_t ff_workspace_bytes(int64_t L, int64_t patch_num) {
    (void)patch_num;
    if (L < 0) return 0;
    // front: level-0 select table + fp32 level rows (ff_plan.hip), the per-slice rows of the order
    // kernel; tail (down from the end): the per-slice level-1 select tables
    return ff::plan_ws_front_bytes(L) + ((size_t)(L / 4096) + 1) * 64 + 256 + ff::plan_ws_tail_bytes(L);
}

// first half of a merge call: K0 (unless order_valid / hinted) + K1, which also accumulates the select tables of the call
static int merge_begin(const void* hidden, const void* addend, int dtype, int64_t L, int64_t d, const int64_t* patch_type,
                              int64_t patch_num, int order_valid, double threshold, int32_t* order, int32_t* inv,
                              void* sim, int64_t* stats, int64_t seq, int64_t hint_pre, int64_t hint_frames, void* ws,
                              size_t ws_bytes, ff_stream_t stream) {
    if (!hidden || !patch_type || !order || !inv || !sim || !stats || !ws) return FF_ERR_ARG;
    if (ws_bytes < ff_workspace_bytes(L, patch_num)) return FF_ERR_WORKSPACE;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((uintptr_t)addend & 15) || ((d * esz) & 15) || ((uintptr_t)ws & 15)) return FF_ERR_ALIGN;
    if (L >= (1ll << 31) || d * esz >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    // frame-major hint: the similarity kernel derives (and verifies) the order itself, K0 is skipped
    const bool hinted = !order_valid && hint_frames > 0 && hint_pre >= 0 && patch_num >= 1 &&
                        hint_pre + hint_frames * patch_num <= L;
    if (!order_valid && !hinted) {
        int rc = ff_build_order(patch_type, L, patch_num, order, inv, stats, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    if (L == 0) return FF_OK;
    // the similarity kernel also accumulates the select tables of this call (zero on entry: cleared by
    // the previous call's merge kernel)
    (void)seq;
    return ff::launch_similarity_any(hidden, addend, dtype, L, d, patch_type, order, inv, stats, sim, ff::ws_l0(ws),
                                     ff::ws_t16_end(ws, ws_bytes), threshold, hint_pre, patch_num,
                                     hinted ? hint_frames : 0, (hipStream_t)stream);
}

// second half of a merge call: select (policy or forced k) + run merge (fold 1: main.py's sequential
// rounding, 2: the baseline's fp32 mean) + compaction
static int merge_finish(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                        double threshold, double sub, double ratio_lb, long long force_k, int fold,
                        const int32_t* order, const int32_t* inv, const void* sim, uint8_t* member, int32_t* dst,
                        uint8_t* keep, int64_t* stats, int64_t* stats_host_mapped, int64_t seq, const ff_aux_t* aux_host,
                        int n_aux, int32_t* order_next, int32_t* inv_next, void* ws, size_t ws_bytes,
                        ff_stream_t stream, int phase = 3, int64_t min_cap = -1, bool guarded = false) {
    // phase: 1 = the plan only (no output field is looked at), 2 = the merge kernel only (behind a plan whose result the caller
    // has seen: L_cap >= min_cap = its l_out is enough), 3 = both (L_cap >= L: the output length is not known yet - or `guarded`:
    // buffers of a GUESSED length; the merge kernel writes nothing unless the plan's l_out is exactly L_cap)
    if (!hidden || (!hidden_out && (phase & 2)) || !order || !inv || !sim || !member || !dst || !keep || !stats || !ws) return FF_ERR_ARG;
    if ((order_next == nullptr) != (inv_next == nullptr)) return FF_ERR_ARG;
    if (ws_bytes < ff_workspace_bytes(L, 1)) return FF_ERR_WORKSPACE;
    if ((phase & 2) && L_cap < (phase == 3 ? (guarded ? 1 : L) : min_cap)) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (n_aux < 0 || n_aux > FF_MAX_AUX || (n_aux > 0 && !aux_host)) return FF_ERR_ARG;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((uintptr_t)addend & 15) || ((phase & 2) && ((uintptr_t)hidden_out & 15)) || ((d * esz) & 15)) return FF_ERR_ALIGN;
    if (order_next && ((uintptr_t)member & 15)) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    if (((uintptr_t)member & 7) || ((uintptr_t)dst & 15) || ((uintptr_t)keep & 15) || ((uintptr_t)order & 15) ||
        ((uintptr_t)sim & 15) || ((uintptr_t)ws & 15) || ((uintptr_t)inv & 15))
        return FF_ERR_ALIGN;
    // when the select folds nothing the merge kernel exits at once: see ff_merge_finish in the header;
    // its extra workgroups clear the select tables for the next call either way
    void *za, *zb;
    size_t zab, zbb;
    ff::table_regions(ws, ws_bytes, L, &za, &zab, &zb, &zbb);
    int rc = FF_OK;
    if (phase & 1)
        rc = ff::launch_plan_merge(sim, dtype, order, inv, L, threshold, sub, ratio_lb, member, dst, keep, stats, ws, ws_bytes,
                                   true, stats_host_mapped, seq, (hipStream_t)stream, force_k);
    if (rc || !(phase & 2)) return rc;
    return ff::launch_merge_compact(hidden, addend, hidden_out, dtype, L, d, L_cap, order, member, fold, dst, keep, aux_host,
                                    n_aux, order_next, inv_next, stats, (hipStream_t)stream, true, za, zab, sim, L, dtype,
                                    ff::ws_t16_end(ws, ws_bytes), nullptr, -1, (guarded && phase == 3 && L_cap < L) ? L_cap : -1);
}

extern "C" int ff_merge_step(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                             const int64_t* patch_type, int64_t patch_num, int order_valid, double threshold,
                             double sub, double ratio_lb, int32_t* order, int32_t* inv, void* sim, uint8_t* member,
                             int32_t* dst, uint8_t* keep, int64_t* stats, int64_t* stats_host_mapped,
                             int64_t seq, const ff_aux_t* aux_host, int n_aux, int64_t hint_pre,
                             int64_t hint_frames, int32_t* order_next, int32_t* inv_next, void* ws, size_t ws_bytes,
                             ff_stream_t stream) {
    int rc = merge_begin(hidden, addend, dtype, L, d, patch_type, patch_num, order_valid, threshold, order, inv, sim, stats,
                         seq, hint_pre, hint_frames, ws, ws_bytes, stream);
    if (rc) return rc;
    return merge_finish(hidden, addend, hidden_out, dtype, L, d, L_cap, threshold, sub, ratio_lb, -1, FF_FOLD_SEQUENTIAL, order, inv,
                        sim, member, dst, keep, stats, stats_host_mapped, seq, aux_host, n_aux, order_next, inv_next, ws, ws_bytes,
                        stream);
}

// one prune call (main.py:61-101): head mean (+ select tables) -> plan -> gather by output rows
static int prune_step(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t S, int64_t d, int64_t L_cap,
                             const void* attn_w, int w_dtype, int64_t H, int64_t num, void* importance,
                             int tables_ready,
                             int64_t start, int64_t n_img, int64_t k, uint8_t* member, int32_t* dst, uint8_t* keep,
                             int64_t* stats, const ff_aux_t* aux_host, int n_aux, void* ws, size_t ws_bytes,
                             ff_stream_t stream) {
    if (!attn_w || !importance || !hidden || !hidden_out || !member || !dst || !keep || !stats || !ws) return FF_ERR_ARG;
    if (S < 0 || d < 1 || H < 1 || num < 1 || start < 0 || n_img < 0 || start + n_img > S || k < 0) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (w_dtype != FF_F32 && w_dtype != FF_BF16 && w_dtype != FF_F16) return FF_ERR_ARG;
    if (S >= (1ll << 29)) return FF_ERR_UNSUPPORTED;
    if (L_cap < S - n_img + (k <= n_img ? k : n_img)) return FF_ERR_ARG;       // (the gather writes that many rows)
    if (ws_bytes < ff_workspace_bytes(S, 1)) return FF_ERR_WORKSPACE;
    if (((uintptr_t)attn_w & 15) || ((uintptr_t)importance & 15) || ((uintptr_t)member & 7) || ((uintptr_t)dst & 15) ||
        ((uintptr_t)keep & 15) || ((uintptr_t)ws & 15))
        return FF_ERR_ALIGN;
    if (S == 0) return FF_OK;
    hipStream_t st = (hipStream_t)stream;
    const void* imp = attn_w;
    bool have_tables = tables_ready != 0;          // a [S] importance whose producer already filled the tables
    if (H * num != 1 || !have_tables) {
        // head mean (main.py:69-70) + the select tables of its output in the same launch.  A ready-made [S] importance
        // without tables takes the same kernel (the mean of one row is the row): one launch instead of the stand-alone
        // plan's memsets + table kernel (84.5 -> 59.2 us per prune call at the C3 shape, r03_k5_experiments.txt)
        int rc = ff::launch_head_mean(attn_w, w_dtype, H, num, S, importance, start, start + n_img, ff::ws_l0(ws),
                                      ff::ws_t16_end(ws, ws_bytes), st);
        if (rc) return rc;
        imp = importance;
        have_tables = true;
    }
    int rc = ff::launch_plan_prune(imp, w_dtype, S, start, n_img, k, member, dst, keep, stats, ws, ws_bytes, have_tables, st);
    if (rc) return rc;
    void *za = nullptr, *zb = nullptr;
    size_t zab = 0, zbb = 0;
    if (have_tables) ff::table_regions(ws, ws_bytes, S, &za, &zab, &zb, &zbb);
    (void)zb; (void)zbb;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((uintptr_t)addend & 15) || ((uintptr_t)hidden_out & 15) || ((d * esz) & 15)) return FF_ERR_ALIGN;
    return ff::launch_merge_compact(hidden, addend, hidden_out, dtype, S, d, L_cap, nullptr, member, FF_FOLD_DROP, dst, keep,
                                    aux_host, n_aux, nullptr, nullptr, nullptr, st, false, za, zab, have_tables ? imp : nullptr,
                                    S, w_dtype, ff::ws_t16_end(ws, ws_bytes),
                                    // (the plan wrote src[] = the inverse of dst[]: the gather walks the OUTPUT rows; exactly k of
                                    // the n_img positions of the range are kept, so the output length is the caller's arithmetic)
                                    k <= n_img ? ff::ws_scratch_ints(ws, S) : nullptr, S - n_img + k);
}

// ---- call context (ABI v7): one host call per FrameFusion.forward call ------------------------------------
#include <time.h>
#if defined(__x86_64__)
#include <immintrin.h>
static inline void cpu_relax() { _mm_pause(); }
#else
static inline void cpu_relax() {}
#endif

static inline int64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

static int ctx_check(const ff_ctx_t* c, int64_t L) {
    if (!c || !c->order || !c->order_next || !c->inv || !c->inv_next || !c->sim || !c->member || !c->dst || !c->keep ||
        !c->stats || !c->stats_host || !c->ws)
        return FF_ERR_ARG;
    if (L < 0 || L > c->cap) return FF_ERR_ARG;
    if (c->ws_bytes < ff_workspace_bytes(c->cap, 1)) return FF_ERR_WORKSPACE;
    return FF_OK;
}

// restore the workspace protocol after a call that died half-way (zeroed tables, zeroed stats)
static int ctx_clean(ff_ctx_t* c, hipStream_t st) {
    if (!c->dirty) return FF_OK;
    hipError_t e = hipMemsetAsync(c->ws, 0, c->ws_bytes, st);
    if (e == hipSuccess) e = hipMemsetAsync(c->stats, 0, FF_STAT_WORDS * sizeof(int64_t), st);
    if (e != hipSuccess) return (int)e;
    c->dirty = 0;
    c->order_len = 0;          // stats[NV] / stats[FTN] went with the reset: K0 (or the hinted K1) rebuilds them
    c->cur_nv = c->cur_ftn = 0;
    c->last_L = 0;
    return FF_OK;
}

extern "C" int ff_ctx_reset(ff_ctx_t* c, ff_stream_t stream) {
    if (!c) return FF_ERR_ARG;
    c->order_len = 0;
    c->cur_nv = c->cur_ftn = 0;
    c->in_flight = 0;
    c->res_active = 0;
    c->last_L = 0;
    return ctx_clean(c, (hipStream_t)stream);
}

extern "C" void* ff_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return nullptr;
    memset(p, 0, bytes);
    return p;
}
extern "C" void ff_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

extern "C" size_t ff_abi_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(ff_ctx_t);
        case 1: return sizeof(ff_merge_call_t);
        case 2: return sizeof(ff_merge_result_t);
        case 3: return sizeof(ff_prune_call_t);
        case 4: return sizeof(ff_aux_t);
        case 5: return sizeof(ff_lq_args_t);
        default: return 0;
    }
}

static int ctx_begin(ff_ctx_t* c, const ff_merge_call_t* a, bool hinted) {
    hipStream_t st = (hipStream_t)a->stream;
    int rc = ctx_clean(c, st);
    if (rc) return rc;
    const int order_valid = (a->order_valid && c->order_len == a->L) ? 1 : 0;
    c->last_L = 0;
    c->seq += 1;
    c->dirty = 1;                // until finish has enqueued the kernel that clears the select tables
    c->in_flight = 1;
    return merge_begin(a->hidden, a->addend, (int)a->dtype, a->L, a->d, a->patch_type, a->patch_num, order_valid,
                          a->threshold, c->order, c->inv, c->sim, c->stats, c->seq, a->hint_pre,
                          hinted ? a->hint_frames : 0, c->ws, c->ws_bytes, a->stream);
}

extern "C" int ff_ctx_merge_begin(ff_ctx_t* c, const ff_merge_call_t* a) {
    if (!a) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (a->L == 0) return FF_ERR_ARG;
    // a planned / waited-for / submitted call must be finished (or the context reset) first: its result block would never be
    // matched and its order swap lost.  A call that was only begun may be begun again (the workspace is reset: ctx_begin).
    if (c->in_flight >= 2) return FF_ERR_STATE;
    if (c->res_off > 0) c->res_off -= 1;
    return ctx_begin(c, a, true);
}

// Wait for the pinned result block; no HIP call on the fast path.  A call whose stream is empty is answered in ~70 us:
// pure spin (pause).  Inside a real prefill the host runs far ahead of the device and the answer is tens of milliseconds
// away: after 200 us the loop yields its time slice on every turn, after 2 ms it sleeps 50 us per turn - other ranks and
// threads of an oversubscribed host keep their cores.  The stream is looked at after 1 ms and then every ~50 us of
// waiting: a drained or failed stream ends the wait.
#include <sched.h>
static int ctx_wait(ff_ctx_t* c, hipStream_t st, int64_t* waited_ns, int word = FF_STAT_SEQ, int shift = 0) {
    volatile int64_t* host = c->stats_host + (word - FF_STAT_SEQ);      // (the loop below names the word FF_STAT_SEQ)
    const int64_t seq = c->seq;
    const int64_t t0 = now_ns();
    int64_t next_query = t0 + 1000000;               // first look at the stream after 1 ms
    int rc = FF_OK;
    int phase = 0;                                   // 0 spin, 1 yield, 2 sleep
    for (uint32_t spins = 0;; ++spins) {
        if ((__atomic_load_n(&host[FF_STAT_SEQ], __ATOMIC_ACQUIRE) >> shift) == seq) break;
        if (phase == 0) {
            cpu_relax();
            if ((spins & 255u) != 255u) continue;
        } else if (phase == 1) {
            sched_yield();
        } else {
            timespec ts{0, 50000};
            nanosleep(&ts, nullptr);
        }
        const int64_t t = now_ns();
        phase = t - t0 < 200000 ? 0 : (t - t0 < 2000000 ? 1 : 2);
        if (t < next_query) continue;
        next_query = t + 50000;
        hipError_t e = hipStreamQuery(st);
        if (e == hipErrorNotReady) continue;
        // the stream drained (or failed): the block is there now or never will be
        if ((__atomic_load_n(&host[FF_STAT_SEQ], __ATOMIC_ACQUIRE) >> shift) == seq) break;
        rc = e == hipSuccess ? FF_ERR_DEVICE : (int)e;
        break;
    }
    *waited_ns = now_ns() - t0;
    return rc;
}

static int ctx_finish_enqueue(ff_ctx_t* c, const ff_merge_call_t* a, int phase = 3, int64_t min_cap = -1, bool guarded = false) {
    if (a->fold != FF_FOLD_SEQUENTIAL && a->fold != FF_FOLD_MEAN) return FF_ERR_ARG;
    int rc = merge_finish(a->hidden, a->addend, a->hidden_out, (int)a->dtype, a->L, a->d, a->L_cap, a->threshold, a->sub,
                          a->ratio_lb, a->force_k < 0 ? -1 : (long long)a->force_k, (int)a->fold, c->order, c->inv, c->sim,
                          c->member, c->dst, c->keep, c->stats, c->stats_host, c->seq, a->aux, (int)a->n_aux, c->order_next,
                          c->inv_next, c->ws, c->ws_bytes, a->stream, phase, min_cap, guarded);
    if (rc) return rc;
    if (phase & 2) c->dirty = 0;           // (the merge kernel is what clears the select tables)
    return rc;
}

// everything of a finish that follows the argument checks; `enqueued`: the kernels of the first attempt are already on the stream
static int ctx_after_result(ff_ctx_t* c, const ff_merge_call_t* a, const ff_merge_result_t* r);
static int ctx_finish(ff_ctx_t* c, const ff_merge_call_t* a, ff_merge_result_t* r, bool enqueued, int phase = 3) {
    int rc;
    c->in_flight = 0;
    hipStream_t st = (hipStream_t)a->stream;
    r->unhinted = 0;
    r->wait_ns = 0;
    r->applied = 0;
    bool resident = c->res_active == 1;            // the first attempt is the one-launch kernel
    bool pending = false;
    const bool guarded = c->res_active == 2;       // ... or three launches, the merge kernel blind into buffers of a guessed length
    c->res_active = 0;
    for (int attempt = 0;; ++attempt) {
        if (!(enqueued && attempt == 0)) {
            rc = ctx_finish_enqueue(c, a, phase, -1, guarded && phase == 3);
            if (rc) { c->dirty = 1; c->order_len = 0; return rc; }
        }
        int64_t waited = 0;
        rc = ctx_wait(c, st, &waited);
        r->wait_ns += waited;
        if (rc) { c->dirty = 1; c->order_len = 0; r->error = 0; return rc; }
        const int64_t* h = c->stats_host;
        const int64_t err = h[FF_STAT_ERROR];
        r->error = err;
        const bool layout = (err & FF_ERR_BIT_LAYOUT) && !(err & ~(int64_t)(FF_ERR_BIT_LAYOUT | FF_ERR_BIT_RESIDENT));
        const bool gave_up = resident && (err & FF_ERR_BIT_RESIDENT) && !(err & ~(int64_t)FF_ERR_BIT_RESIDENT);
        if ((layout || gave_up) && attempt == 0) {
            // layout: patch_type is not the frame-major layout the hint described - everything this call enqueued is void, repeat it
            // through K0.  gave_up: the one-launch kernel left at its barrier (or found another sequence than the host described);
            // its tables and barrier words are in no defined state.  Either way: reset the workspace, repeat through the three
            // launches - blind into the caller's buffers when they hold L rows, else the plan only (ff_ctx_merge_apply follows)
            if (gave_up) c->res_off = 32;
            c->dirty = 1;
            rc = ctx_begin(c, a, !layout);
            c->in_flight = 0;
            if (rc) { c->dirty = 1; c->order_len = 0; return rc; }
            if (layout) r->unhinted = 1;
            if (resident) phase = (phase != 1 && a->hidden_out && a->L_cap >= a->L) ? 3 : 1;
            resident = false;
            continue;
        }
        if (err) {
            c->dirty = 1; c->order_len = 0;
            return FF_ERR_DEVICE;
        }
        r->nv = h[FF_STAT_NV];
        r->ftn = h[FF_STAT_FTN];
        r->count = h[FF_STAT_COUNT];
        r->branch = h[FF_STAT_BRANCH];
        r->k = h[FF_STAT_K];
        r->l_out = h[FF_STAT_LOUT];
        if (resident) {
            if (h[FF_STAT_APPLIED] == 1) c->dirty = 0;  // (the one-launch kernel leaves the select tables alone)
            else if (h[FF_STAT_APPLIED] == 2) pending = true;      // the kernel waits for outputs of l_out rows, by mail
            else phase = 1;                              // the plan only: the merge kernel is still to come
        } else if (guarded && phase == 3 && r->l_out != a->L && r->l_out != a->L_cap) {
            phase = 1;                                   // the blind merge kernel found another length than its buffers': nothing written
        }
        break;
    }
    if (pending) {                         // ff_ctx_merge_apply hands over / confirms the outputs; no launch
        c->in_flight = 3;
        c->res_active = 3;
        r->applied = 2;
        return FF_OK;
    }
    if (phase == 1) {                      // the merge kernel is still to come (ff_ctx_merge_apply)
        c->in_flight = 3;
        return FF_OK;
    }
    r->applied = 1;
    return ctx_after_result(c, a, r);
}

// what follows the merge kernel's enqueue once the result is known: mask gather, order swap
static int ctx_after_result(ff_ctx_t* c, const ff_merge_call_t* a, const ff_merge_result_t* r) {
    int rc;
    // the attention mask follows once the call is known to be valid and to fold something: nothing for an attempt whose layout
    // hint was wrong, nothing when the sequence stays as it is (the caller keeps its own mask)
    if (a->mask && r->l_out != a->L) {
        rc = ff::gather_mask(a->mask, a->mask_out, a->mask_elem_bytes, a->L, a->L_cap, c->dst, c->stats,
                            ff::ws_scratch_ints(c->ws, c->cap), a->stream);
        if (rc) { c->dirty = 1; c->order_len = 0; return rc; }
    }
    if (r->l_out != a->L) {
        // the merge kernel wrote the by-patch order of the compacted sequence: it is the current one now
        int32_t* t = c->order; c->order = c->order_next; c->order_next = t;
        t = c->inv; c->inv = c->inv_next; c->inv_next = t;
        c->swaps += 1;
        c->order_len = r->l_out;
        c->cur_nv = r->nv - (a->L - r->l_out);          // (what the merge kernel leaves in stats[NV] / stats[FTN])
        c->cur_ftn = r->ftn - (a->L - r->l_out);
    } else {
        c->order_len = a->L;       // nothing folded: the order describes the unchanged sequence
        c->cur_nv = r->nv;
        c->cur_ftn = r->ftn;
    }
    c->last_L = a->L;              // (what ff_ctx_gather_mask may be asked for)
    c->last_l_out = r->l_out;
    return FF_OK;
}

static int ctx_finish_check(ff_ctx_t* c, const ff_merge_call_t* a, ff_merge_result_t* r) {
    if (!a || !r) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (c->in_flight != 1) return FF_ERR_STATE;
    if (a->mask && !a->mask_out) return FF_ERR_ARG;
    return FF_OK;
}

extern "C" int ff_ctx_merge_finish(ff_ctx_t* c, const ff_merge_call_t* a, ff_merge_result_t* r) {
    int rc = ctx_finish_check(c, a, r);
    if (rc) return rc;
    return ctx_finish(c, a, r, false);
}

// Is this call one for the one-launch kernel (ff_resident.hip)?  The host must know the by-patch order's length: from the layout
// hint, or from the previous call of the prefill.
static bool ctx_resident(const ff_ctx_t* c, const ff_merge_call_t* a, int64_t* nv, int64_t* ftn, bool* hinted) {
    if (c->res_off > 0 || a->L < 1 || a->d < 1 || (c->cap & 3)) return false;      // (whole 16-byte words behind order / inv / sim)
    const bool ov = a->order_valid && c->order_len == a->L && c->cur_nv > 0 && !c->dirty;
    const bool hint = !ov && a->hint_frames > 0 && a->hint_pre >= 0 && a->patch_num >= 1 &&
                      a->hint_pre + a->hint_frames * a->patch_num <= a->L;
    if (!ov && !hint) return false;
    if (hint && a->patch_num < 32) return false;           // (the kernel's closed form of a position's slot: 32 positions, two frames)
    *nv = ov ? c->cur_nv : a->hint_frames * a->patch_num;
    *ftn = ov ? c->cur_ftn : *nv;
    *hinted = hint;
    if (a->fold != FF_FOLD_SEQUENTIAL || a->n_aux < 0 || a->n_aux > FF_MAX_AUX) return false;
    if (((uintptr_t)a->hidden & 15) || ((uintptr_t)a->hidden_out & 15) || !a->hidden || !a->patch_type) return false;
    if ((uintptr_t)a->addend & 15) return false;
    return ff::merge_resident_fits((int)a->dtype, a->L, a->d, *nv, a->addend != nullptr, hint, (int)a->fold);
}

extern "C" int ff_ctx_merge_one_launch(const ff_ctx_t* c, const ff_merge_call_t* a) {
    if (!a || ctx_check(c, a->L) || c->in_flight >= 2) return 0;
    int64_t nv, ftn;
    bool hinted;
    ff_ctx_t probe = *c;
    if (probe.res_off > 0) probe.res_off -= 1;          // (what ff_ctx_merge_submit will see)
    return ctx_resident(&probe, a, &nv, &ftn, &hinted) ? 1 : 0;
}

// A merge call in two halves that do NOT wait in between: submit = everything enqueued (outputs allocated up front),
// collect = the wait for the result block + the context's bookkeeping.  A host thread that submits sample B (on another
// stream) before it collects sample A keeps two samples in flight: A's plan bubble and kernel ramps sit under B's streaming
// pass (what two replicas on two threads achieve in the reference's demo, llava_video_compare.py:217-223).
// When the call fits on the chip (ctx_resident) "everything" is ONE kernel.
extern "C" int ff_ctx_merge_submit(ff_ctx_t* c, const ff_merge_call_t* a) {
    if (!a) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (c->in_flight >= 2) return FF_ERR_STATE;
    int64_t nv = 0, ftn = 0;
    bool hinted = false;
    if (c->res_off > 0) c->res_off -= 1;
    if (!ctx_resident(c, a, &nv, &ftn, &hinted)) {
        if (a->late_outputs) return FF_ERR_ARG;          // (outputs by mail: the one-launch kernel only)
        c->res_off += c->res_off > 0 ? 1 : 0;            // (ff_ctx_merge_begin counts the call)
        rc = ff_ctx_merge_begin(c, a);
        if (rc) return rc;
        if (a->mask && !a->mask_out) rc = FF_ERR_ARG;
        // buffers of fewer than L rows: exactly sized outputs for a GUESSED length (the top-k branch's, main.py:122) - the merge
        // kernel goes out blind all the same and writes nothing unless the plan's l_out is that length (collect: applied = 0)
        const bool guarded = a->hidden_out && a->L_cap < a->L;
        if (!rc) rc = ctx_finish_enqueue(c, a, 3, -1, guarded);
        if (rc) { c->in_flight = 0; c->dirty = 1; c->order_len = 0; return rc; }
        c->in_flight = 4;                    // submitted: only ff_ctx_merge_collect may follow
        c->res_active = guarded ? 2 : 0;
        return FF_OK;
    }
    if (a->mask && !a->mask_out && a->hidden_out) return FF_ERR_ARG;
    const bool late = a->late_outputs != 0;
    for (int x = 0; x < ((a->hidden_out && !late) ? (int)a->n_aux : 0); ++x)
        if (!a->aux[x].src || !a->aux[x].dst || a->aux[x].row_bytes < 1 || a->aux[x].outer < 1 || a->aux[x].src_outer_bytes < 0) return FF_ERR_ARG;
    hipStream_t st = (hipStream_t)a->stream;
    rc = ctx_clean(c, st);
    if (rc) return rc;
    const int order_valid = hinted ? 0 : 1;
    c->last_L = 0;
    c->seq += 1;
    c->dirty = 1;                // until the kernel is known to have cleared the select tables (collect) / the merge kernel is enqueued (apply)
    ff::ResLaunch p;
    p.hidden = a->hidden; p.addend = a->addend; p.hidden_out = a->hidden_out; p.dtype = (int)a->dtype; p.L = a->L; p.d = a->d; p.L_cap = a->L_cap;
    p.nv = nv; p.ftn = ftn; p.ptype = a->patch_type; p.order = c->order; p.inv = c->inv;
    p.hint_pre = a->hint_pre; p.hint_patches = a->patch_num; p.hint_frames = order_valid ? 0 : a->hint_frames;
    p.sim = c->sim; p.member = c->member; p.keep = c->keep; p.dst = c->dst; p.order_next = c->order_next; p.inv_next = c->inv_next;
    p.stats = c->stats; p.host_mapped = c->stats_host; p.seq = c->seq; p.aux = a->aux; p.n_aux = (int)a->n_aux;
    p.thr = a->threshold; p.sub = a->sub; p.ratio_lb = a->ratio_lb; p.force_k = a->force_k < 0 ? -1 : (long long)a->force_k;
    p.ws = c->ws; p.ws_bytes = c->ws_bytes;
    p.mail = nullptr;
    if (late) {
        // the outputs follow by mail: the slot is invalidated first (plain host stores; the kernel compares the sequence word)
        volatile int64_t* m = c->stats_host + FF_MAIL_WORD;
        m[0] = 0;
        m[FF_MAIL_WORDS] = 0;
        c->stats_host[FF_STAT_ACK] = 0;
        p.mail = c->stats_host + FF_MAIL_WORD;
        p.hidden_out = nullptr; p.L_cap = 0; p.n_aux = 0;
    }
    rc = ff::launch_merge_resident(p, st);
    if (rc) { c->order_len = 0; return rc; }
    c->in_flight = 4;
    c->res_active = 1;
    return FF_OK;
}

// the outputs of `a` into the next free mail slot of the call (1: before the result is known, 2: sized to it)
static int ctx_mail(ff_ctx_t* c, const ff_merge_call_t* a) {
    if (a->n_aux < 0 || a->n_aux > FF_MAX_AUX) return FF_ERR_ARG;
    if ((uintptr_t)a->hidden_out & 15) return FF_ERR_ALIGN;
    for (int x = 0; x < (a->hidden_out ? (int)a->n_aux : 0); ++x)
        if (!a->aux[x].src || !a->aux[x].dst || a->aux[x].row_bytes < 1 || a->aux[x].outer < 1 || a->aux[x].src_outer_bytes < 0) return FF_ERR_ARG;
    static_assert(sizeof(ff_aux_t) == 40 && FF_MAIL_WORDS == 4 + 5 * FF_MAX_AUX, "a mail slot holds the auxiliary entries whole");
    int64_t* m = c->stats_host + FF_MAIL_WORD;
    int slot = 1;
    if (m[0] == c->seq * 4 + 1) {
        if (m[FF_MAIL_WORDS] == c->seq * 4 + 2) return FF_ERR_STATE;          // both slots of this call are written
        slot = 2;
        m += FF_MAIL_WORDS;
    }
    m[1] = (int64_t)(uintptr_t)a->hidden_out;
    m[2] = a->hidden_out ? a->L_cap : 0;
    m[3] = a->hidden_out ? a->n_aux : 0;
    const bool any = a->hidden_out != nullptr;
    for (int x = 0; x < FF_MAX_AUX; ++x) {
        const bool on = any && x < a->n_aux;
        int64_t* e = m + 4 + 5 * x;
        e[0] = on ? (int64_t)(uintptr_t)a->aux[x].src : 0;
        e[1] = on ? (int64_t)(uintptr_t)a->aux[x].dst : 0;
        e[2] = on ? a->aux[x].row_bytes : 0;
        e[3] = on ? a->aux[x].outer : 0;
        e[4] = on ? a->aux[x].src_outer_bytes : 0;
    }
    __atomic_store_n(&m[0], c->seq * 4 + slot, __ATOMIC_RELEASE);
    return FF_OK;
}

extern "C" int ff_ctx_merge_mail(ff_ctx_t* c, const ff_merge_call_t* a) {
    if (!a) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (!((c->in_flight == 4 && c->res_active == 1) || (c->in_flight == 3 && c->res_active == 3))) return FF_ERR_STATE;
    return ctx_mail(c, a);
}

extern "C" int ff_ctx_merge_collect(ff_ctx_t* c, const ff_merge_call_t* a, ff_merge_result_t* r) {
    if (!a || !r) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (c->in_flight != 4) return FF_ERR_STATE;
    if (a->mask && !a->mask_out) return FF_ERR_ARG;
    return ctx_finish(c, a, r, true);
}

// ---- the merge call for exactly sized outputs: the host sees l_out BEFORE the merge kernel is enqueued --------------------------
// begin (K1) -> ff_ctx_merge_wait (plan kernel enqueued, then the wait for the result block; a wrong layout
// hint is repeated through K0 here) -> the host sizes its outputs to l_out -> ff_ctx_merge_apply (merge kernel, mask gather,
// order swap).  Costs the GPU the host's reaction time between plan and merge kernel (~10-20 us) and saves the input-length
// output buffers (or the copy out of them).
extern "C" int ff_ctx_merge_wait(ff_ctx_t* c, const ff_merge_call_t* a, ff_merge_result_t* r) {
    if (!a || !r) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (c->in_flight != 1) return FF_ERR_STATE;
    rc = ctx_finish_enqueue(c, a, 1);                  // the plan kernel behind K1; no output field is looked at
    if (rc) { c->in_flight = 0; c->dirty = 1; c->order_len = 0; return rc; }
    return ctx_finish(c, a, r, true, 1);               // (leaves in_flight = 3 on success)
}

extern "C" int ff_ctx_merge_apply(ff_ctx_t* c, const ff_merge_call_t* a, const ff_merge_result_t* r) {
    if (!a || !r) return FF_ERR_ARG;
    int rc = ctx_check(c, a->L);
    if (rc) return rc;
    if (c->in_flight != 3) return FF_ERR_STATE;
    if (a->mask && !a->mask_out) return FF_ERR_ARG;
    if (r->l_out != c->stats_host[FF_STAT_LOUT] || r->l_out < 0 || r->l_out > a->L) return FF_ERR_ARG;      // not this call's result
    if (c->res_active == 3) {
        // a one-launch kernel is waiting, rows in hand, for outputs of l_out rows: the call block's go out by mail unless a mail
        // that holds them is on its way already (then the call block must name the same buffers), and the kernel's answer is
        // awaited (~2 PCIe round trips).  "None came in time" (a host that took > 4 ms): the merge kernel follows as below.
        if (!a->hidden_out || a->L_cap < r->l_out) return FF_ERR_ARG;
        const int64_t* m = c->stats_host + FF_MAIL_WORD;
        int sent = 0;
        for (int slot = 2; slot >= 1 && !sent; --slot) {
            const int64_t* ms = m + (slot - 1) * FF_MAIL_WORDS;
            // (slot 1 was written before the result was known: it counts when it holds exactly l_out rows - a guess that came
            // true - or a whole input; a longer guess is not what a caller of exactly sized outputs wants written)
            if (ms[0] == c->seq * 4 + slot && ms[1] && (slot == 2 ? ms[2] >= r->l_out : (ms[2] == r->l_out || ms[2] >= a->L))) sent = slot;
        }
        if (sent && (m[(sent - 1) * FF_MAIL_WORDS + 1] != (int64_t)(uintptr_t)a->hidden_out || m[(sent - 1) * FF_MAIL_WORDS + 2] != a->L_cap))
            return FF_ERR_ARG;
        if (!sent) {
            rc = ctx_mail(c, a);
            if (rc) return rc;
        }
        c->res_active = 0;
        c->in_flight = 0;
        int64_t waited = 0;
        rc = ctx_wait(c, (hipStream_t)a->stream, &waited, FF_STAT_ACK, 2);
        if (rc) { c->dirty = 1; c->order_len = 0; return rc; }
        if ((c->stats_host[FF_STAT_ACK] & 3) != 3) {
            c->dirty = 0;
            return ctx_after_result(c, a, r);
        }
    }
    c->in_flight = 0;
    rc = ctx_finish_enqueue(c, a, 2, r->l_out);
    if (rc) { c->dirty = 1; c->order_len = 0; return rc; }
    return ctx_after_result(c, a, r);
}

extern "C" int ff_ctx_gather_mask(ff_ctx_t* c, const void* mask, void* mask_out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                                  ff_stream_t stream) {
    int rc = ctx_check(c, L);
    if (rc) return rc;
    // only behind the merge call whose keep set is still in the context: same L, finished, folded something, nothing begun since
    if (c->in_flight || c->dirty || c->last_L != L || L == 0 || c->last_l_out == L || L_cap < c->last_l_out) return FF_ERR_STATE;
    return ff::gather_mask(mask, mask_out, elem_bytes, L, L_cap, c->dst, c->stats, ff::ws_scratch_ints(c->ws, c->cap), stream);
}

extern "C" int ff_ctx_prune(ff_ctx_t* c, const ff_prune_call_t* a) {
    if (!a) return FF_ERR_ARG;
    int rc = ctx_check(c, a->S);
    if (rc) return rc;
    if (c->in_flight) return FF_ERR_STATE;               // (a merge call of this context is still open)
    if (a->mask && !a->mask_out) return FF_ERR_ARG;
    hipStream_t st = (hipStream_t)a->stream;
    if (a->tables_ready != 0 && a->tables_ready != 1) return FF_ERR_ARG;
    if (!a->tables_ready) {
        rc = ctx_clean(c, st);     // (tables announced by ff_ctx_expect_tables but not used: start from zero)
        if (rc) return rc;
    }
    c->last_L = 0;
    c->dirty = 1;
    c->order_len = 0;              // the sequence changes and no order is maintained through a prune
    rc = prune_step(a->hidden, a->addend, a->hidden_out, (int)a->dtype, a->S, a->d, a->L_cap, a->attn_w, (int)a->w_dtype,
                       a->H, a->num, c->sim, (int)a->tables_ready, a->start, a->n_img, a->k, c->member, c->dst, c->keep,
                       c->stats, a->aux, (int)a->n_aux, c->ws, c->ws_bytes, a->stream);
    if (rc) return rc;
    c->dirty = 0;
    if (a->mask)
        rc = ff::gather_mask(a->mask, a->mask_out, a->mask_elem_bytes, a->S, a->L_cap, c->dst, c->stats,
                            ff::ws_scratch_ints(c->ws, c->cap), a->stream);
    return rc;
}

// hook + prune from one host call: importance (+ select tables) -> plan -> gather, all enqueued here
extern "C" int ff_ctx_prune_from_qk(ff_ctx_t* c, const ff_prune_call_t* a, const ff_lq_args_t* q) {
    if (!a || !q) return FF_ERR_ARG;
    int rc = ctx_check(c, a->S);
    if (rc) return rc;
    if (c->in_flight) return FF_ERR_STATE;
    if (a->S < 1 || a->start < 0 || a->n_img < 0 || a->start + a->n_img > a->S || a->k < 0 || a->k > a->n_img) return FF_ERR_ARG;
    if (q->dtype != FF_F32 && q->dtype != FF_BF16 && q->dtype != FF_F16) return FF_ERR_ARG;
    rc = ctx_clean(c, (hipStream_t)a->stream);          // (select tables zero on entry)
    if (rc) return rc;
    c->seq += 1;
    c->dirty = 1;                                       // until the prune below has enqueued the kernel that clears the tables
    c->order_len = 0;
    c->last_L = 0;
    rc = ff_last_query_attention(q->q_last, q->k, (int)q->dtype, q->H, q->H_kv, q->num, a->S, q->dh, q->k_head_stride, q->k_key_stride,
                                 q->scale, (int)q->causal, q->bias, nullptr, c->sim, a->start, a->start + a->n_img, c->ws, c->ws_bytes,
                                 q->ws, q->ws_bytes, a->stream);
    if (rc) return rc;
    ff_prune_call_t b = *a;
    b.attn_w = c->sim;
    b.w_dtype = q->dtype;
    b.H = 1;
    b.num = 1;
    b.tables_ready = 1;
    return ff_ctx_prune(c, &b);
}
