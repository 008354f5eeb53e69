// K4 - run merge + compaction, the second (and last) streaming pass over the activations.
// Replaces index_add_ + divide (framefusion/main.py:304-317) and the bool-mask gathers of
// hidden_states, cos/sin (or position ids) and patch_type (main.py:132-138, 161-178), which the
// reference runs as 6+ kernels with host syncs, by one launch that reads every surviving or
// folded row exactly once and writes every output row exactly once.
//
// Work decomposition: by INPUT rows, not output rows - a wave owns about `slots` consecutive slots t of
// the by-patch order (then the non-visual tail) for one 1 KiB column block.  Member slots are
// read by the wave that owns their anchor: the boundary between two waves' ranges sits on the non-member
// slot nearest to a multiple of `slots`.  Because every input row is read by exactly one wave per column
// block and the ranges are even (headline call: 37 +- 3.9 slots, longest 53), the HBM read load is
// balanced no matter how long individual runs are (an output-row decomposition would leave the
// longest run as the tail).
//
// Arithmetic: the anchor accumulates its members in by-patch
// order with a rounding to the activation dtype T after EVERY add, then one rounded divide by
// T(n+1) - the order CPU index_add_ applies; fp32-accumulate-then-round differs on ~23 % of
// elements by more than 1e-3 relative.
#include <atomic>

#include "ff_common.h"
#include "ff_merge_body.h"

namespace ff {

template <int DT, bool kAdd>
__global__ __launch_bounds__(kMergeThreads) void k_merge_compact(
    const char* __restrict__ hidden, const char* __restrict__ addend, char* __restrict__ out, uint32_t row_bytes, int L, int64_t L_cap,
    const int32_t* __restrict__ order, const uint8_t* __restrict__ member, int fold,
    const int32_t* __restrict__ dst, const uint8_t* __restrict__ keep, AuxPack aux, int n_main,
    int n_aux_blocks, int n_next_blocks, int32_t* __restrict__ order_next, int32_t* __restrict__ inv_next,
    int64_t* __restrict__ stats, const int64_t* __restrict__ identity_stats, ZeroJob zero, int slots, long long guard_lout) {
    merge_compact_body<DT, kAdd>(hidden, addend, out, row_bytes, L, L_cap, order, member, fold, dst, keep, aux, n_main,
                                 n_aux_blocks, n_next_blocks, order_next, inv_next, stats, identity_stats, zero, slots,
                                 (int)blockIdx.x, (int)blockIdx.y, guard_lout);
}

// The prune's gather by OUTPUT rows (prune_gather_body) + the short roles of a prune launch: auxiliary rows, table clearing.
template <int DT, bool kAdd>
__global__ __launch_bounds__(kMergeThreads) void k_prune_gather(
    const char* __restrict__ hidden, const char* __restrict__ addend, char* __restrict__ out, uint32_t row_bytes, int L, int64_t L_cap,
    int l_out, const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const uint8_t* __restrict__ keep, AuxPack aux,
    int n_main, int n_aux_blocks, ZeroJob zero, int rows) {
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (bx >= n_main + n_aux_blocks) {
        if (by == 0) role_clear_tables(zero, bx - n_main - n_aux_blocks);
        return;
    }
    if (bx >= n_main) {
        if (by == 0) role_aux_rows(aux, bx - n_main, L, L_cap, keep, dst);
        return;
    }
    prune_gather_body<DT, kAdd>(hidden, addend, out, row_bytes, L, l_out, src, rows, bx, by);
}

// ---- square attention-mask gather (main.py:137-138, 99-100): out[r, c] = mask[src[r], src[c]] ----------------
// Round 2 ran one workgroup per INPUT row with a dst[] lookup per input element: L^2 index reads and 2-byte scattered
// stores (2.7 GB of traffic for a bf16 mask at 37 k tokens).  Two levels now: k_invert_dst turns dst[] (row of every
// kept position) into src[] (position of every output row) once, then one workgroup per OUTPUT row walks the output
// columns - src[] read as whole words, the mask row gathered through it (ascending, mostly neighbouring elements),
// 16-byte stores.  What is left is the mask itself: L_out^2 elements written, the kept part of L_out rows read.
// (A variant that reads the input rows as whole aligned 16-byte chunks and stages the survivors of a 4 KiB segment in LDS
// before aligned stores was built and measured: 243 / 1153 us against 171 / 847 us for this one at 12.5 k / 36.9 k tokens -
// three barriers and two LDS atomics per 4 KiB cost more than the 2-byte gathers; profiles/r03_mask_gather.txt.)
__global__ __launch_bounds__(256) void k_invert_dst(const int32_t* __restrict__ dst, int L, int32_t* __restrict__ src) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) {
        const int r = dst[i];
        if (r >= 0) src[r] = i;
    }
}

template <int EB>
__global__ __launch_bounds__(256) void k_gather_mask(const char* __restrict__ mask, char* __restrict__ out, int L, int64_t L_cap,
                                                     const int32_t* __restrict__ src, const int64_t* __restrict__ stats) {
    constexpr int V = 16 / EB;                               // elements per 16-byte store
    const int L_out = (int)stats[FF_STAT_LOUT];
    const int r = blockIdx.x;
    if (r >= L_out) return;
    const char* row_in = mask + (int64_t)src[r] * L * EB;
    char* row_out = out + (int64_t)r * L_cap * EB;
    const bool wide = (((uintptr_t)row_out | (uintptr_t)((int64_t)L_cap * EB)) & 15) == 0;
    for (int c0 = threadIdx.x * V; c0 < L_out; c0 += 256 * V) {
        if (wide && c0 + V <= L_out) {
            int idx[V];
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const int4 w = *(const int4*)(src + c0 + e);     // (src is 16-byte aligned, c0 a multiple of V >= 4... V = 2: below)
                idx[e] = w.x; idx[e + 1] = w.y; idx[e + 2] = w.z; idx[e + 3] = w.w;
            }
            uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if constexpr (EB == 1) pk[e >> 2] |= (uint32_t)(uint8_t)row_in[idx[e]] << (8 * (e & 3));
                else if constexpr (EB == 2) pk[e >> 1] |= (uint32_t)((const uint16_t*)row_in)[idx[e]] << (16 * (e & 1));
                else pk[e] = ((const uint32_t*)row_in)[idx[e]];
            }
            *(uint4*)(row_out + (int64_t)c0 * EB) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        } else {
            for (int e = 0; e < V && c0 + e < L_out; ++e) {
                const int c = src[c0 + e];
                if constexpr (EB == 1) row_out[c0 + e] = row_in[c];
                else if constexpr (EB == 2) ((uint16_t*)row_out)[c0 + e] = ((const uint16_t*)row_in)[c];
                else ((uint32_t*)row_out)[c0 + e] = ((const uint32_t*)row_in)[c];
            }
        }
    }
}

// 8-byte elements (fp64 masks): two per 16-byte store
__global__ __launch_bounds__(256) void k_gather_mask8(const char* __restrict__ mask, char* __restrict__ out, int L, int64_t L_cap,
                                                      const int32_t* __restrict__ src, const int64_t* __restrict__ stats) {
    const int L_out = (int)stats[FF_STAT_LOUT];
    const int r = blockIdx.x;
    if (r >= L_out) return;
    const uint64_t* row_in = (const uint64_t*)(mask + (int64_t)src[r] * L * 8);
    uint64_t* row_out = (uint64_t*)(out + (int64_t)r * L_cap * 8);
    for (int c = threadIdx.x; c < L_out; c += 256) row_out[c] = row_in[src[c]];
}

// Slots per workgroup, chosen per launch (measured at 64 x 576 x 4096, K4 in the step; 32 was the fixed value):
//  * the waves are long streams, so a launch whose workgroups do not all fit on the chip at once ends with a
//    nearly empty extra round: 32 slots = 2 304 workgroups on 2 048 places 79 us, 36 slots (2 048) 73.6 us,
//    40 slots (1 844) 75.8 us, 48 slots (1 536) 80.5 us.  The main workgroups are sized to ~97 % of ONE round's
//    places (the rest is for the short aux / order / table blocks passing through) whenever that fits;
//  * all workgroups start together and advance at the same pace: when the slot count shares a factor with the
//    frames per patch they all sit on the same few frames - the same few MB - at any time (64 frames: 16 / 24 /
//    32 slots 34.5 us, 19 / 21 slots 31 us at 64 x 210 x 3584; 36 / 38 slots 73.6 us, 37 slots 71 us above).
//    A prime count staggers the workgroups over the frames whatever the frame count is.
template <int DT, bool kAdd>
static int merge_places() {
    // workgroups of this instantiation the CURRENT device holds at once (cached per device: several
    // replicas on several GPUs may share the process)
    static std::atomic<int> cache[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    if (dev >= 0 && dev < kMaxDevices) {
        const int got = cache[dev].load(std::memory_order_relaxed);
        if (got > 0) return got;
    }
    int per_cu = 0, places = 2048;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_merge_compact<DT, kAdd>, kMergeThreads, 0) == hipSuccess &&
        per_cu >= 1)
        places = per_cu * prop.multiProcessorCount;
    if (dev >= 0 && dev < kMaxDevices) cache[dev].store(places, std::memory_order_relaxed);
    return places;
}
// (`places`: workgroups of the launching kernel the device holds at once, less whatever else must fit next to them)
int merge_slots_for(int places, int64_t L, int ny) {
    static const int primes[] = {17, 19, 23, 29, 31, 37, 41, 43, 47, 53};
    const double want = (double)L * ny / (0.975 * (double)places);
    for (int p : primes)
        if ((double)p >= want) return p;
    // One round does not hold the call.  Whole rounds of long workgroups only pay when they are WHOLE (128 x 576 x 4096:
    // 37 slots = 1.996 rounds 155 us, 11 slots 163, 7 slots 164, 5 slots 158; 96 x 576 x 4096: 29 slots = 1.91 rounds 127 us,
    // 7 slots 121.5) and the rows are at most two column groups wide (64 x 576 x 8192, four groups: 37 slots = 2.0 rounds
    // 166 us, 29 162, 11 159, 7 158): otherwise many short workgroups (profiles/r03_k5_experiments.txt)
    if (ny <= 2)
        for (int r = 2; r <= 3; ++r)
            for (int p : primes)
                if ((double)p >= want / r) {
                    if (want / r >= 0.985 * p) return p;
                    break;
                }
    return 7;
}
static int merge_slots(int dtype, bool add, int64_t L, int ny) {
    int places;
    switch (dtype) {
        case FF_F32: places = add ? merge_places<FF_F32, true>() : merge_places<FF_F32, false>(); break;
        case FF_BF16: places = add ? merge_places<FF_BF16, true>() : merge_places<FF_BF16, false>(); break;
        default: places = add ? merge_places<FF_F16, true>() : merge_places<FF_F16, false>();
    }
    return merge_slots_for(places, L, ny);
}

int launch_merge_compact(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                         const int32_t* order, const uint8_t* member, int fold, const int32_t* dst,
                         const uint8_t* keep, const ff_aux_t* aux_host, int n_aux, int32_t* order_next,
                         int32_t* inv_next, int64_t* stats, hipStream_t st, bool skip_identity, void* zero_a,
                         size_t zero_a_bytes, const void* zero_keys, int64_t zero_n, int zero_key_dt, int* t16_end,
                         const int32_t* src, int64_t l_out, int64_t guard_lout) {
    AuxPack pack;
    pack.n = keep ? n_aux : 0;
    for (int x = 0; x < pack.n; ++x) pack.a[x] = aux_host[x];
    for (int x = pack.n; x < FF_MAX_AUX; ++x) pack.a[x] = ff_aux_t{nullptr, nullptr, 0, 0, 0};
    const int64_t row_bytes = d * (dtype == FF_F32 ? 4 : 2);
    const int nblk = (int)((row_bytes + 1023) / 1024);
    const int ny = (nblk + kMergeWaves - 1) / kMergeWaves;
    const int n_aux_blocks = pack.n ? (int)((L + kMergeWaves * 4 - 1) / (kMergeWaves * 4)) : 0;
    if (fold == FF_FOLD_DROP && !order && src && l_out >= 0) {
        // a prune: by output rows (src[] = the plan's inverse of dst[]); `rows` kept rows x 4 column tiles per workgroup.
        // 8 / 16 / 32 rows: 62.0 / 63.0 / 71.5 us at the 72B shape, 13.0 / 11.4 / 11.2 at the Qwen2-VL shape, 30.0 / 30.5 / 29.9 at
        // C2's threshold cascade - against 79.1 / 17.2 / 33.8 for the walk over the input slots (profiles/r05_prune_gather.txt)
        constexpr int rows = 16;
        const int n_main = (int)((l_out + rows - 1) / rows);
        ZeroJob zero{(uint4*)zero_a, (int)(zero_a_bytes / 16), zero_keys, (int)zero_n, zero_key_dt, t16_end, 0};
        if (zero_a) zero.n_blocks = zero_keys ? (int)((zero_n + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 1;
        const dim3 grid((unsigned)(n_main + n_aux_blocks + zero.n_blocks), (unsigned)ny);
        if (grid.x == 0) return FF_OK;
#define FF_PG_LAUNCH(DT, ADD)                                                                                                  \
    hipLaunchKernelGGL((k_prune_gather<DT, ADD>), grid, dim3(kMergeThreads), 0, st, (const char*)hidden, (const char*)addend,  \
                       (char*)hidden_out, (uint32_t)row_bytes, (int)L, L_cap, (int)l_out, src, dst, keep, pack, n_main,        \
                       n_aux_blocks, zero, rows)
        switch (dtype) {
            case FF_F32: if (addend) FF_PG_LAUNCH(FF_F32, true); else FF_PG_LAUNCH(FF_F32, false); break;
            case FF_BF16: if (addend) FF_PG_LAUNCH(FF_BF16, true); else FF_PG_LAUNCH(FF_BF16, false); break;
            default: if (addend) FF_PG_LAUNCH(FF_F16, true); else FF_PG_LAUNCH(FF_F16, false);
        }
#undef FF_PG_LAUNCH
        return (int)hipGetLastError();
    }
    const int slots = merge_slots(dtype, addend != nullptr, L, ny);
    const int n_main = (int)((L + slots - 1) / slots);
    if (!order || !stats) order_next = nullptr;
    const int n_next_blocks = order_next ? (int)((L + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 0;
    ZeroJob zero{(uint4*)zero_a, (int)(zero_a_bytes / 16), zero_keys, (int)zero_n, zero_key_dt, t16_end, 0};
    if (zero_a) zero.n_blocks = zero_keys ? (int)((zero_n + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 1;
    const dim3 grid((unsigned)(n_main + n_aux_blocks + n_next_blocks + zero.n_blocks), (unsigned)ny);
    const char* h = (const char*)hidden;
    char* o = (char*)hidden_out;
    const int64_t* ident = (skip_identity && stats) ? stats : nullptr;
#define FF_MC_LAUNCH(DT, ADD)                                                                                          \
    hipLaunchKernelGGL((k_merge_compact<DT, ADD>), grid, dim3(kMergeThreads), 0, st, h, (const char*)addend, o,          \
                       (uint32_t)row_bytes, (int)L, L_cap, order, member, fold, dst, keep, pack, n_main, n_aux_blocks,  \
                       n_next_blocks, order_next, inv_next, stats, ident, zero, slots, (long long)guard_lout)
    switch (dtype) {
        case FF_F32: if (addend) FF_MC_LAUNCH(FF_F32, true); else FF_MC_LAUNCH(FF_F32, false); break;
        case FF_BF16: if (addend) FF_MC_LAUNCH(FF_BF16, true); else FF_MC_LAUNCH(FF_BF16, false); break;
        default: if (addend) FF_MC_LAUNCH(FF_F16, true); else FF_MC_LAUNCH(FF_F16, false);
    }
#undef FF_MC_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace ff

extern "C" int ff_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d,
                                int64_t L_cap, const int32_t* order, const uint8_t* member, int fold,
                                const int32_t* dst, const uint8_t* keep, const ff_aux_t* aux_host, int n_aux,
                                ff_stream_t stream) {
    if (n_aux > 0 && !keep) return FF_ERR_ARG;
    if (!hidden || !hidden_out || !member || !dst || L < 0 || d < 1 || L_cap < 0) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (n_aux < 0 || n_aux > FF_MAX_AUX || (n_aux > 0 && !aux_host)) return FF_ERR_ARG;
    for (int x = 0; x < n_aux; ++x)
        if (!aux_host[x].src || !aux_host[x].dst || aux_host[x].row_bytes < 1 || aux_host[x].outer < 1 || aux_host[x].src_outer_bytes < 0)
            return FF_ERR_ARG;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((uintptr_t)hidden_out & 15) || ((d * esz) & 15)) return FF_ERR_ALIGN;
    if (L >= (1ll << 29) || d * esz >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (L == 0) return FF_OK;
    return ff::launch_merge_compact(hidden, nullptr, hidden_out, dtype, L, d, L_cap, order, member, fold, dst, keep, aux_host, n_aux,
                                    nullptr, nullptr, nullptr,
                                    (hipStream_t)stream, false, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, -1, -1);
}

// Square attention-mask gather (main.py:137-138, 99-100): out[r, c] = mask[src[r], src[c]] for the kept rows / columns; dst: the
// plan's row of every position (-1 = dropped); stats: the plan's result block (FF_STAT_LOUT is read on the device); scratch: [L]
// int32, 16-byte aligned (receives the position of every output row).  Two launches.
namespace ff {
int gather_mask(const void* mask, void* out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                const int32_t* dst, const int64_t* stats, int32_t* scratch, ff_stream_t stream) {
    if (!mask || !out || !dst || !stats || !scratch || L < 0 || L_cap < 0) return FF_ERR_ARG;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return FF_ERR_ARG;
    if (L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if ((uintptr_t)scratch & 15) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ff::k_invert_dst, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, dst, (int)L, scratch);
    const char* m = (const char*)mask;
    char* o = (char*)out;
    switch ((int)elem_bytes) {
        case 1: hipLaunchKernelGGL(ff::k_gather_mask<1>, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats); break;
        case 2: hipLaunchKernelGGL(ff::k_gather_mask<2>, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats); break;
        case 4: hipLaunchKernelGGL(ff::k_gather_mask<4>, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats); break;
        default: hipLaunchKernelGGL(ff::k_gather_mask8, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats);
    }
    return (int)hipGetLastError();
}
}  // namespace ff

// ---- stand-alone token gathers for the reference's public position handlers (main.py:142-178) -------------------
// position_embedding_handler_at_pruning(pe, keep_indexs): pe[..., keep_indexs, :] - rows in the order of an index
// tensor; position_embedding_handler_at_merging(pe, token_mask): pe[..., token_mask[0], :] - compaction by a boolean
// row.  The hot path never calls them (its merge kernel gathers the position tensors on the way); they exist so that
// code written against the reference's method surface keeps working.
namespace ff {
__global__ __launch_bounds__(256) void k_rows_by_index(AuxPack aux, const int64_t* __restrict__ index, int n, int L) {
    const int r = blockIdx.x * kMergeWaves * 4 + wave_id() * 4 + (lane_id() >> 4);
    const int sub = lane_id() & 15;
    if (r >= n) return;
    int64_t i = index[r];
    if (i < 0) i += L;                                       // torch indexing accepts negative indices
    if (i < 0 || i >= L) return;
    for (int x = 0; x < aux.n; ++x) {
        const ff_aux_t& ax = aux.a[x];
        for (int64_t ou = 0; ou < ax.outer; ++ou)
            copy_row(aux_src_row(ax, ou, i, L), (char*)ax.dst + (ou * n + r) * ax.row_bytes, ax.row_bytes, sub, 16);
    }
}
__global__ __launch_bounds__(256) void k_rows_by_dst(AuxPack aux, const int32_t* __restrict__ dst, int L, int64_t L_cap) {
    const int i = blockIdx.x * kMergeWaves * 4 + wave_id() * 4 + (lane_id() >> 4);
    const int sub = lane_id() & 15;
    if (i >= L) return;
    const int r = dst[i];
    if (r < 0) return;
    for (int x = 0; x < aux.n; ++x) {
        const ff_aux_t& ax = aux.a[x];
        for (int64_t ou = 0; ou < ax.outer; ++ou)
            copy_row(aux_src_row(ax, ou, i, L), (char*)ax.dst + (ou * L_cap + r) * ax.row_bytes, ax.row_bytes, sub, 16);
    }
}
int launch_scan_keep(const uint8_t* keep, int64_t L, int32_t* dst, int64_t* stats, hipStream_t st);
}  // namespace ff

static int pack_aux(const ff_aux_t* aux_host, int n_aux, ff::AuxPack& pack) {
    if (n_aux < 1 || n_aux > FF_MAX_AUX || !aux_host) return FF_ERR_ARG;
    pack.n = n_aux;
    for (int x = 0; x < FF_MAX_AUX; ++x) pack.a[x] = x < n_aux ? aux_host[x] : ff_aux_t{nullptr, nullptr, 0, 0, 0};
    for (int x = 0; x < n_aux; ++x)
        if (!pack.a[x].src || !pack.a[x].dst || pack.a[x].row_bytes < 1 || pack.a[x].outer < 1 || pack.a[x].src_outer_bytes < 0)     // (any row size: copy_row)
            return FF_ERR_ARG;
    return FF_OK;
}

extern "C" int ff_gather_tokens_by_index(const int64_t* index, int64_t n, int64_t L, const ff_aux_t* aux_host, int n_aux,
                                         ff_stream_t stream) {
    if (n < 0 || L < 0 || (n > 0 && !index)) return FF_ERR_ARG;
    if (n >= (1ll << 31) || L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    ff::AuxPack pack;
    int rc = pack_aux(aux_host, n_aux, pack);
    if (rc) return rc;
    if (n == 0) return FF_OK;
    const unsigned blocks = (unsigned)((n + ff::kMergeWaves * 4 - 1) / (ff::kMergeWaves * 4));
    hipLaunchKernelGGL(ff::k_rows_by_index, dim3(blocks), dim3(ff::kMergeThreads), 0, (hipStream_t)stream, pack, index, (int)n, (int)L);
    return (int)hipGetLastError();
}

extern "C" int ff_gather_tokens_by_mask(const uint8_t* keep, int64_t L, int64_t L_cap, int32_t* dst, int64_t* stats,
                                        const ff_aux_t* aux_host, int n_aux, ff_stream_t stream) {
    if (!keep || !dst || !stats || L < 0 || L_cap < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31) - 65536) return FF_ERR_UNSUPPORTED;
    if (((uintptr_t)keep & 15) || ((uintptr_t)dst & 15)) return FF_ERR_ALIGN;
    ff::AuxPack pack;
    int rc = pack_aux(aux_host, n_aux, pack);
    if (rc) return rc;
    if (L == 0) return FF_OK;
    rc = ff::launch_scan_keep(keep, L, dst, stats, (hipStream_t)stream);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((L + ff::kMergeWaves * 4 - 1) / (ff::kMergeWaves * 4));
    hipLaunchKernelGGL(ff::k_rows_by_dst, dim3(blocks), dim3(ff::kMergeThreads), 0, (hipStream_t)stream, pack, dst, (int)L, L_cap);
    return (int)hipGetLastError();
}
