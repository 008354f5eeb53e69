// Plan + K4 in ONE launch (16-bit activations, at most 65 536 tokens): what a merge call enqueues behind the
// similarity kernel (framefusion/main.py:112-138: select, run detection, keep mask, fold, compaction).
//
// Why.  As two launches the plan kernel (72 workgroups, a chain of dependent round trips over < 1 MB) leaves the chip
// idle for ~8 us, and the merge kernel behind it then pays its own launch, the dispatch of ~2 000 workgroups and the
// ramp of their first requests (profiles/r03_k4_probes.txt: 4.0 us for the empty grid).  Here the merge kernel's
// workgroups are dispatched WHILE the plan runs: they fetch what does not depend on the plan (their window of the
// by-patch order), then wait for a flag the plan's last workgroup raises, and go.
//
// Shape of the grid (one-dimensional, workgroups of 256 threads):
//     [ merge main: n_main x ny, the first ceil(L / 256) of them run the plan FIRST | auxiliary rows | next order | table clearing ]
// The dispatcher hands workgroups out in index order, so every plan workgroup is resident before any workgroup that
// only waits is: a waiting workgroup never holds a place the plan still needs.  A plan workgroup goes on as a main
// workgroup, so the main workgroups alone are sized to the chip (as in the stand-alone merge kernel).  Nothing depends on
// that order for SAFETY: a waiter gives up after ~0.1 s, sets FF_ERR_BIT_BARRIER and returns - the host then falls back
// to the three-launch form (ff_abi.hip).
//
// Hand-over (two of them, ff_merge_body.h).  DECISION: plan workgroup 0 publishes (k-th key, tie slot, nv) as soon as it
// has them; a main workgroup derives its member flags from the similarities itself and requests its first rows.  DONE: each
// plan workgroup, after its last store: barrier, then ONE lane does a RELEASE fetch_add at agent scope on the arrival
// counter (the release writes this XCD's dirty L2 lines back - the plan wrote ~0.4 MB in all); the last arriver resets
// the counter and stores the call's sequence number (host-side counter of the context: strictly increasing, so the flags
// never need clearing) into 64 flag words 128 bytes apart.  A waiter polls ONE of the copies (relaxed, agent scope: the
// load goes past the non-coherent L2).  NO acquire fence on the waiting side: `buffer_inv sc1` in every wave of the
// streaming pass cost 100 us; the one line that can be stale (stats) is read with agent-scope loads instead.
//
// Measured (profiles/r04_fused.txt, profiles/EXPERIMENTS.md 4.1-4.6): 141.8 us (DONE only) / 145.8 us (with the early
// decision) against 137.6-138.2 us for two launches at 64 x 576 x 4096 - the plan runs 3 us slower inside a busy grid, the
// hand-over costs 1.9 us, and that is more than the launch + dispatch it hides.  OFF unless FF_FUSED=1 /
// ff_set_fused_launch(1); bit-exact either way (tests/test_gpu_random_sweep.py).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "ff_common.h"
#include "ff_merge_body.h"
#include "ff_plan_fast.h"

namespace ff {

PlanParams merge_plan_params(int dtype, double thr, double sub, double ratio_lb, long long force_k);
int merge_slots_for(int places, int64_t L, int ny);
uint32_t* ws_tag(void* ws);
unsigned long long* ws_agg(void* ws);
unsigned int* ws_arrive(void* ws);
unsigned long long* ws_flags(void* ws);
int* ws_l0(void* ws);
int32_t* ws_scratch_ints(void* ws, int64_t L);
int* ws_t16_end(void* ws, size_t ws_bytes);

constexpr int kFusedThreads = kMergeThreads;          // 256: one by-patch slot + one position per plan thread

struct FusedPlanArgs {
    const void* values;
    int cap;
    PlanParams pp;
    const int* l0;
    int* t16_end;
    int64_t* stats;
    const int32_t* inv;
    int L;
    uint8_t* member;
    uint8_t* keep;
    int32_t* dst;
    unsigned long long* agg;
    uint32_t* tagword;
    int64_t* host_mapped;
    int64_t seq;
    int n_plan;
    unsigned int* arrive;
    unsigned long long* flags;
    int dbg;
    long long* dbg_buf;
};

struct FusedMergeArgs {
    const char* hidden;
    const char* addend;
    char* out;
    uint32_t row_bytes;
    int L;
    int64_t L_cap;
    const int32_t* order;
    int fold;
    AuxPack aux;
    int n_main, ny, n_aux_blocks, n_next_blocks;
    int32_t* order_next;
    int32_t* inv_next;
    ZeroJob zero;
    int slots;
};

// amdgpu_num_sgpr(80): the two argument blocks would keep 87 SGPRs live, and this chip holds 8 waves per SIMD only up to 80
// SGPRs per wave (VCC / FLAT_SCRATCH / XNACK_MASK included) - 7 up to 96, whatever hipOccupancyMaxActiveBlocksPerMultiprocessor
// says (tools/occprobe: 2 048 / 1 792 / 1 536 workgroups really resident at next_free_sgpr <= 74 / 77..89 / 96).  With 7
// the main workgroups do not fit the chip at once and the late ones are a second round: +18 us.
template <int DT, bool kAdd>
__global__ __launch_bounds__(kFusedThreads) __attribute__((amdgpu_num_sgpr(80))) void k_plan_merge(const FusedPlanArgs pa, const FusedMergeArgs ma) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int b = (int)blockIdx.x;
    if (b < pa.n_plan) {
        if ((pa.dbg & 4) && b == 0 && threadIdx.x == 0) {
            pa.stats[FF_STAT_T_ORDER] = wall_clock64();
        }
        plan_fast_body<DT, kRowSlicesLds, kFusedThreads>(pa.values, pa.cap, pa.pp, pa.l0, pa.t16_end, pa.stats, pa.inv, pa.L,
                                                         pa.member, pa.keep, pa.dst, pa.agg, pa.tagword, pa.host_mapped, pa.seq,
                                                         lds_raw, b, pa.n_plan, pa.flags);
        __syncthreads();                           // every store of this workgroup has been issued and acknowledged
        if ((pa.dbg & 4) && threadIdx.x == 0 && b == pa.n_plan - 1) pa.stats[FF_STAT_T_ORDER + 1] = wall_clock64();
        if (threadIdx.x < kWave) {
            unsigned int old = 0;
            if (threadIdx.x == 0) {
                if (pa.dbg & 2) old = __hip_atomic_fetch_add(pa.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else old = __hip_atomic_fetch_add(pa.arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            old = (unsigned int)__builtin_amdgcn_readfirstlane((int)old);
            if (old == (unsigned int)pa.n_plan - 1u) {
                if ((pa.dbg & 4) && threadIdx.x == 0) pa.stats[FF_STAT_T_ORDER + 2] = wall_clock64();
                if (threadIdx.x == 0) __hip_atomic_store(pa.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pa.dbg & 2)
                    __hip_atomic_store(pa.flags + (size_t)threadIdx.x * kFlagStride, (unsigned long long)pa.seq, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                else
                    __hip_atomic_store(pa.flags + (size_t)threadIdx.x * kFlagStride, (unsigned long long)pa.seq, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ... and goes on as main workgroup b of the merge (its places stay in use: the main workgroups alone fill the chip)
    }
    // the merge kernel's own coordinates: main workgroups first (x fastest, as its 2-D grid is dispatched), then the rest
    const int i = b;
    const int n_mains = ma.n_main * ma.ny;
    const int bx = i < n_mains ? i % ma.n_main : ma.n_main + (i - n_mains);
    const int by = i < n_mains ? i / ma.n_main : 0;
    merge_compact_body<DT, kAdd, true>(ma.hidden, ma.addend, ma.out, ma.row_bytes, ma.L, ma.L_cap, ma.order, pa.member, ma.fold,
                                       pa.dst, pa.keep, ma.aux, ma.n_main, ma.n_aux_blocks, ma.n_next_blocks, ma.order_next,
                                       ma.inv_next, pa.stats, pa.stats, ma.zero, ma.slots, bx, by,
                                       FusedWait{pa.flags, (unsigned long long)pa.seq, pa.stats, pa.values, pa.pp.thr_key, pa.dbg, pa.dbg_buf});
}

template <int DT, bool kAdd>
static int fused_places(size_t lds, int n_slices) {
    static std::atomic<int> cache_all[kMaxDevices][kRowSlicesLds + 1];       // (the LDS request grows with the slices of the call)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    std::atomic<int>* cache = cache_all[0] + n_slices - dev;                // cache[dev] below = cache_all[dev][n_slices]
    if (dev >= 0 && dev < kMaxDevices) {
        cache = &cache_all[dev][n_slices] - dev;
        const int got = cache[dev].load(std::memory_order_relaxed);
        if (got > 0) return got;
    }
    int per_cu = 0, places = 2048;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_plan_merge<DT, kAdd>, kFusedThreads, lds) == hipSuccess && per_cu >= 1)
        places = per_cu * prop.multiProcessorCount;
    if (dev >= 0 && dev < kMaxDevices) cache[dev].store(places, std::memory_order_relaxed);
    return places;
}

// FF_FUSED=0 in the environment keeps the three-launch form (A/B measurements); fused_disable() is the run-time switch
static std::atomic<int> g_fused{-1};
void fused_disable() { g_fused.store(0, std::memory_order_relaxed); }
static int fused_enabled() {
    int enabled = g_fused.load(std::memory_order_relaxed);
    if (enabled < 0) {
        const char* e = getenv("FF_FUSED");
        enabled = (e && e[0] == '1') ? 1 : 0;          // off unless asked for: measured slower than two launches (header comment)
        g_fused.store(enabled, std::memory_order_relaxed);
    }
    return enabled;
}
int fused_set(int on) {
    const int prev = fused_enabled();
    if (on >= 0) g_fused.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
bool fused_applies(int dtype, int64_t L, const int32_t* inv, const int32_t* order, const int64_t* stats) {
    return fused_enabled() && (dtype == FF_BF16 || dtype == FF_F16) && L > 0 && L <= (int64_t)kRowSlicesLds * kSelSlice && inv && order && stats;
}

// Same contract as launch_plan_merge (have_tables = true) followed by launch_merge_compact (skip_identity = true).
int launch_plan_merge_fused(const void* sim, int dtype, const int32_t* order, const int32_t* inv, int64_t L, double thr, double sub,
                            double ratio_lb, long long force_k, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                            void* ws, size_t ws_bytes, int64_t* host_mapped, int64_t seq, const void* hidden, const void* addend,
                            void* hidden_out, int64_t d, int64_t L_cap, int fold, const ff_aux_t* aux_host, int n_aux,
                            int32_t* order_next, int32_t* inv_next, void* zero_a, size_t zero_a_bytes, hipStream_t st) {
    FusedPlanArgs pa;
    pa.dbg_buf = (long long*)(((uintptr_t)ws_scratch_ints(ws, L) + 7) & ~(uintptr_t)7);
    pa.values = sim; pa.cap = (int)L;
    pa.pp = merge_plan_params(dtype, thr, sub, ratio_lb, force_k);
    pa.pp.n_slices = (int)((L + kSelSlice - 1) / kSelSlice);
    pa.l0 = ws_l0(ws); pa.t16_end = ws_t16_end(ws, ws_bytes); pa.stats = stats; pa.inv = inv; pa.L = (int)L;
    pa.member = member; pa.keep = keep; pa.dst = dst; pa.agg = ws_agg(ws); pa.tagword = ws_tag(ws);
    pa.host_mapped = host_mapped; pa.seq = seq;
    pa.n_plan = (int)((L + kFusedThreads - 1) / kFusedThreads);
    pa.arrive = ws_arrive(ws); pa.flags = ws_flags(ws);
    { const char* e = getenv("FF_FUSED_DBG"); pa.dbg = e ? atoi(e) : 0; }

    FusedMergeArgs ma;
    ma.aux.n = n_aux;
    for (int x = 0; x < n_aux; ++x) ma.aux.a[x] = aux_host[x];
    for (int x = n_aux; x < FF_MAX_AUX; ++x) ma.aux.a[x] = ff_aux_t{nullptr, nullptr, 0, 0, 0};
    const int64_t row_bytes = d * 2;
    const int nblk = (int)((row_bytes + 1023) / 1024);
    ma.hidden = (const char*)hidden; ma.addend = (const char*)addend; ma.out = (char*)hidden_out;
    ma.row_bytes = (uint32_t)row_bytes; ma.L = (int)L; ma.L_cap = L_cap; ma.order = order; ma.fold = fold;
    ma.ny = (nblk + kMergeWaves - 1) / kMergeWaves;
    const size_t lds = plan_fast_lds_bytes(pa.pp.n_slices, kFusedThreads);     // (one KiB of level-1 rows per slice of the call)
    int places;
    const int ns = pa.pp.n_slices;
    if (dtype == FF_BF16) places = addend ? fused_places<FF_BF16, true>(lds, ns) : fused_places<FF_BF16, false>(lds, ns);
    else places = addend ? fused_places<FF_F16, true>(lds, ns) : fused_places<FF_F16, false>(lds, ns);
    ma.slots = merge_slots_for(places, L, ma.ny);
    ma.n_main = (int)((L + ma.slots - 1) / ma.slots);
    if ((int64_t)ma.n_main * ma.ny < pa.n_plan) return FF_ERR_UNSUPPORTED;      // (never: a main workgroup owns <= 53 slots, a plan workgroup 256)
    ma.n_aux_blocks = n_aux ? (int)((L + kMergeWaves * 4 - 1) / (kMergeWaves * 4)) : 0;
    ma.order_next = order_next; ma.inv_next = inv_next;
    ma.n_next_blocks = order_next ? (int)((L + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 0;
    ma.zero = ZeroJob{(uint4*)zero_a, (int)(zero_a_bytes / 16), sim, (int)L, dtype, pa.t16_end, 0};
    ma.zero.n_blocks = zero_a ? (int)((L + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 0;
    if (!zero_a) ma.zero.keys = nullptr;
    if (pa.dbg & 8) {
        static int once = 0;
        if (!once++) fprintf(stderr, "[ff_fused] places %d n_plan %d slots %d n_main %d ny %d aux %d next %d zero %d lds %zu\n", places, pa.n_plan,
                             ma.slots, ma.n_main, ma.ny, ma.n_aux_blocks, ma.n_next_blocks, ma.zero.n_blocks, lds);
    }
    const unsigned grid = (unsigned)(ma.n_main * ma.ny + ma.n_aux_blocks + ma.n_next_blocks + ma.zero.n_blocks);
#define FF_FUSED_LAUNCH(DT, ADD) \
    hipLaunchKernelGGL((k_plan_merge<DT, ADD>), dim3(grid), dim3(kFusedThreads), lds, st, pa, ma)
    if (dtype == FF_BF16) { if (addend) FF_FUSED_LAUNCH(FF_BF16, true); else FF_FUSED_LAUNCH(FF_BF16, false); }
    else { if (addend) FF_FUSED_LAUNCH(FF_F16, true); else FF_FUSED_LAUNCH(FF_F16, false); }
#undef FF_FUSED_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace ff

extern "C" int ff_set_fused_launch(int on) { return ff::fused_set(on); }
