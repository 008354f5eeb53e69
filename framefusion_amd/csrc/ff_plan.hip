// K2+K3 - the integer "plan" between the two streaming passes, ONE launch: threshold count, budget
// decision, top-k (radix select with a lowest-index tie rule), member flags, keep mask and the
// compaction scan.  Everything here works on <= L 2-byte similarities and L-sized index arrays (a few
// hundred KB, cache-resident); the branch the reference takes on the host after two .item() syncs
// (framefusion/main.py:112-127) is decided on the device, in double, exactly as python does.
//
// Replaces: main.py:112-127 (select), find_contigious_latter_index (main.py:351-380), the
// unique/where/repeat_interleave index algebra of merge_tokens_and_get_mask (main.py:269-301)
// and the keep-mask construction (main.py:278-279) - 17 host syncs in the reference.
//
// Why one launch is enough.  The select needs two global facts: the k-th largest key and how many
// entries equal to it belong to the top k.  Both come out of small histograms that the PRODUCER of
// the values (the similarity kernel, the importance kernels, or k_tables for the stand-alone entry
// points) accumulates while it runs: the top byte of the order-preserving key (level 0, 16 copies)
// and, per slice of 4096 values, the top 16 bits (level 1).  For 16-bit activation dtypes that is
// the whole key.  Every workgroup of k_plan re-derives the same decision from those tables
// (deterministic, no communication), locates the slot t* of the last tie member that is taken (the
// per-slice tie counts give the slice, one 4096-value scan gives the slot), and from then on
// "is slot t folded?" is a pure function of (value[t], t).  The compaction scan needs, for the
// workgroup that owns positions [4096 g, 4096 g + 4096), the number of kept positions before it:
// every workgroup walks ALL slots once (coalesced values + order[], 36 slots per thread at
// 64 x 576), counts the kept ones whose position precedes its range and drops the keep flags of its
// own range into LDS on the way - so there is no dependency between workgroups at all.
// fp32 activations need two more radix levels (bits 15..0): k_hist_level, one launch each, before
// k_plan.  Run lengths are not materialised: the merge kernel derives them from 64 member flags.
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "ff_common.h"
#include "ff_plan_fast.h"

namespace ff {

constexpr int kEpt = 16;                          // values per thread in k_hist_level / k_tables

// 16 consecutive T values starting at j0 (j0 % 16 == 0): the raw 16-byte words first (so the loads
// can be issued before anything they do not depend on), then the order-preserving keys; entries at
// or beyond `n` are flagged invalid.  `cap` = number of readable elements of the array.
template <int DT> struct RawKeys { uint4 w[Act<DT>::kBytes == 2 ? 2 : 4]; };

template <int DT>
__device__ inline RawKeys<DT> load_raw(const void* __restrict__ v, int j0, int cap) {
    using A = Act<DT>;
    constexpr int W = A::kBytes == 2 ? 2 : 4;
    RawKeys<DT> r;
    if (j0 + kEpt <= cap) {
        const uint4* p = (const uint4*)((const char*)v + (size_t)j0 * A::kBytes);
#pragma unroll
        for (int q = 0; q < W; ++q) r.w[q] = p[q];
    } else {
        uint32_t x[W * 4];
#pragma unroll
        for (int q = 0; q < W * 4; ++q) x[q] = 0;
        for (int e = 0; e < kEpt; ++e) {
            if (j0 + e < cap) {
                const uint32_t b = A::bits1(v, j0 + e);
                if constexpr (A::kBytes == 2) x[e >> 1] |= b << (16 * (e & 1));
                else x[e] = b;
            }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) r.w[q] = make_uint4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    }
    return r;
}

template <int DT>
__device__ inline void keys_of(const RawKeys<DT>& r, int j0, int n, uint32_t* key, uint32_t& valid_mask) {
    using A = Act<DT>;
    if constexpr (A::kBytes == 2) {
        const uint32_t w[8] = {r.w[0].x, r.w[0].y, r.w[0].z, r.w[0].w, r.w[1].x, r.w[1].y, r.w[1].z, r.w[1].w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            key[2 * q] = order_key<DT>(w[q] & 0xffffu);
            key[2 * q + 1] = order_key<DT>(w[q] >> 16);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            key[4 * q] = order_key<DT>(r.w[q].x); key[4 * q + 1] = order_key<DT>(r.w[q].y);
            key[4 * q + 2] = order_key<DT>(r.w[q].z); key[4 * q + 3] = order_key<DT>(r.w[q].w);
        }
    }
    const int left = n - j0;
    valid_mask = left >= kEpt ? 0xffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
}

// key of value t of the array (any t < cap)
template <int DT>
__device__ inline uint32_t key_at(const void* __restrict__ v, int t) { return order_key<DT>(Act<DT>::bits1(v, t)); }

constexpr int kSliceThreads = 256;                // k_hist_level / k_tables: one slice per workgroup
constexpr int kPlanThreads = 1024;                // k_plan: one slice per workgroup, 4 values per thread

struct Resolved {
    bool topk;
    long long k;
    int count;
    uint32_t prefix;     // leading `levels` bytes of the k-th key
    int remaining;       // entries still to take inside the prefix
};

template <int NT>
struct SelLds {
    int tot[256];
    int part[NT / 256][256];
    int scratch[NT / kWave + 1];
    int bcast[8];
};

// pick the bin, from the top, in which the running count reaches `remaining` (s.tot filled)
template <int NT>
__device__ inline int pick_from_tot(SelLds<NT>& s, int remaining, int& above) {
    const int lane = lane_id();
    __syncthreads();
    if (wave_id() == 0) {
        const int top = 255 - 4 * lane;
        const int v0 = s.tot[top], v1 = s.tot[top - 1], v2 = s.tot[top - 2], v3 = s.tot[top - 3];
        const int sum = v0 + v1 + v2 + v3;
        const int incl = wave_incl_scan(sum);
        const unsigned long long hit = __ballot(incl >= remaining);
        const int first = __ffsll((long long)hit) - 1;
        if (lane == first) {
            int ab = incl - sum, bin = top;
            if (ab + v0 >= remaining) { bin = top; }
            else if (ab + v0 + v1 >= remaining) { ab += v0; bin = top - 1; }
            else if (ab + v0 + v1 + v2 >= remaining) { ab += v0 + v1; bin = top - 2; }
            else { ab += v0 + v1 + v2; bin = top - 3; }
            s.bcast[0] = bin;
            s.bcast[1] = ab;
        }
    }
    __syncthreads();
    const int bin = s.bcast[0];
    above = s.bcast[1];
    __syncthreads();
    return bin;
}

// Re-derive the decision and the first `levels` bytes of the k-th key from the tables:
// level 0 = l0 (kL0Copies copies), level 1 = the T16 rows of the level-0 bin, levels 2.. = the
// per-slice rows k_hist_level wrote into lv (fp32 only).  All NT threads call.
template <int NT>
__device__ inline Resolved resolve(const PlanParams& pp, const int* __restrict__ l0, int* t16_end,
                                   const int* __restrict__ lv, int levels, long long ftn, int nv, SelLds<NT>& s) {
    constexpr int NQ = NT / 256;
    const int tid = threadIdx.x, c = tid & 255, q = tid >> 8;
    Resolved r;
    r.prefix = 0; r.remaining = 0; r.count = 0;
    // level-0 column sums: copy q, q + NQ, ... of column c; the count column rides along
    int t0 = 0, cnt = 0;
#pragma unroll
    for (int x = 0; x < kL0Copies / NQ; ++x) t0 += l0[(q + x * NQ) * kL0Stride + c];
    if (tid < kL0Copies) cnt = l0[tid * kL0Stride + 256];
    s.part[q][c] = t0;
    r.count = block_sum_i<NT / kWave>(cnt, s.scratch);                 // (two barriers inside)
    if (tid < 256) {
        int a = 0;
#pragma unroll
        for (int x = 0; x < NQ; ++x) a += s.part[x][tid];
        s.tot[tid] = a;
    }
    if (pp.mode == 0 && pp.k_given >= 0) {
        // fixed-sparsity policy: the caller fixed k (modeling_qwen2_baseline.py:920,1001)
        r.topk = true;
        r.k = pp.k_given > nv ? (long long)nv : pp.k_given;
    } else if (pp.mode == 0) {
        // main.py:114-116 in double, as python: ratio = count / ftn ; ratio < sub ?
        const double ratio = ftn > 0 ? (double)r.count / (double)ftn : 0.0;
        r.topk = !(ratio < pp.sub);
        long long k = 0;
        if (r.topk) {
            k = (long long)(pp.sub * (double)ftn);          // int(sub * ftn), main.py:122
            if (k > nv) k = nv;
            if (k < 0) k = 0;
        }
        r.k = k;
    } else {
        r.topk = true;
        r.k = pp.k_given;
    }
    if (!r.topk || r.k <= 0 || levels <= 0) return r;       // uniform over the whole grid
    r.remaining = (int)r.k;
    int above;
    int bin = pick_from_tot(s, r.remaining, above);
    r.prefix = (uint32_t)bin;
    r.remaining -= above;
    if (levels == 1) return r;
    {   // level 1: column c of the T16 rows of bin, summed over slices q, q + NQ, ... and the copies
        int a = 0;
        for (int g = q; g < pp.n_slices; g += NQ) {
            const int* row = t16_slice(t16_end, g) + t16_bin((r.prefix << 8) | (uint32_t)c);
#pragma unroll
            for (int x = 0; x < kT16Copies; ++x) a += row[x * 65536];
        }
        s.part[q][c] = a;
        __syncthreads();
        if (tid < 256) {
            int b = 0;
#pragma unroll
            for (int x = 0; x < NQ; ++x) b += s.part[x][tid];
            s.tot[tid] = b;
        }
        bin = pick_from_tot(s, r.remaining, above);
        r.prefix = (r.prefix << 8) | (uint32_t)bin;
        r.remaining -= above;
    }
    for (int l = 2; l < levels; ++l) {
        const int* tab = lv + (size_t)(l - 2) * pp.n_slices * 256;
        int a = 0;
        for (int g = q; g < pp.n_slices; g += NQ) a += tab[g * 256 + c];
        s.part[q][c] = a;
        __syncthreads();
        if (tid < 256) {
            int b = 0;
#pragma unroll
            for (int x = 0; x < NQ; ++x) b += s.part[x][tid];
            s.tot[tid] = b;
        }
        bin = pick_from_tot(s, r.remaining, above);
        r.prefix = (r.prefix << 8) | (uint32_t)bin;
        r.remaining -= above;
    }
    return r;
}

// ---- k_tables: the level-0 / level-1 tables of an existing value array ----------------------------
// (stand-alone plan entry points and importances handed over by the caller; the fused merge step
// gets the same tables from the similarity kernel's epilogue).  Tables must be zero on entry.
template <int DT>
__global__ __launch_bounds__(kSliceThreads) void k_tables(const void* __restrict__ values, int cap, int lo, int hi,
                                                          const int64_t* __restrict__ stats, uint32_t thr_key,
                                                          int* __restrict__ l0, int* t16_end) {
    using A = Act<DT>;
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * kSelSlice + tid * kEpt;
    const RawKeys<DT> raw = load_raw<DT>(values, j0, cap);
    if (hi < 0) hi = (int)stats[FF_STAT_NV];
    uint32_t key[kEpt], valid;
    keys_of<DT>(raw, j0, hi, key, valid);
    int* tab = l0 + (blockIdx.x & (kL0Copies - 1)) * kL0Stride;
    int* t16 = t16_slice(t16_end, blockIdx.x) + (tid & (kT16Copies - 1)) * 65536;
    int n_ge = 0;
#pragma unroll
    for (int e = 0; e < kEpt; ++e) {
        const bool in = ((valid >> e) & 1u) && (j0 + e >= lo);
        n_ge += (in && key[e] >= thr_key && key[e] != nan_key<DT>()) ? 1 : 0;
        wave_agg_add<3>(tab, key[e] >> (A::kKeyBits - 8), in);
        // neighbouring lanes use different copies, so the fold is per copy
        if (in) atomicAdd(&t16[t16_bin(key[e] >> (A::kKeyBits - 16))], 1);
    }
    n_ge = wave_sum_i(n_ge);
    if (lane_id() == 0 && n_ge) atomicAdd(&tab[256], n_ge);
}

// ---- k_hist_level: radix levels 2 and 3 of fp32 keys ----------------------------------------------
// Slice g histograms byte `level` of the keys that match the (level)-byte prefix of the k-th key,
// one row per slice in lv[level - 2], no atomics across workgroups.
struct HistLds {
    SelLds<kSliceThreads> sel;
    int hist[kSliceThreads / kWave][2][256];
};

template <int DT>
__global__ __launch_bounds__(kSliceThreads) void k_hist_level(
    const void* __restrict__ values, int cap, PlanParams pp, int level, const int64_t* __restrict__ stats,
    const int* __restrict__ l0, int* t16_end, int* __restrict__ lv) {
    using A = Act<DT>;
    __shared__ HistLds s;
    const int tid = threadIdx.x, w = wave_id(), cp = lane_id() & 1;
    const int j0 = blockIdx.x * kSelSlice + tid * kEpt;
    const RawKeys<DT> raw = load_raw<DT>(values, j0, cap);
    const int nv = (int)stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    const int lo = pp.mode == 0 ? 0 : pp.lo, hi = pp.mode == 0 ? nv : pp.hi;
    const Resolved r = resolve<kSliceThreads>(pp, l0, t16_end, lv, level, ftn, nv, s.sel);
    if (!r.topk || r.k <= 0) return;          // uniform over the whole grid
    __syncthreads();
    for (int x = tid; x < (kSliceThreads / kWave) * 2 * 256; x += kSliceThreads) (&s.hist[0][0][0])[x] = 0;
    __syncthreads();
    uint32_t key[kEpt], valid;
    keys_of<DT>(raw, j0, hi, key, valid);
    const int shift = A::kKeyBits - 8 * (level + 1);
#pragma unroll
    for (int e = 0; e < kEpt; ++e) {
        const bool in = ((valid >> e) & 1u) && (j0 + e >= lo);
        if (in && (key[e] >> (shift + 8)) == r.prefix) atomicAdd(&s.hist[w][cp][(key[e] >> shift) & 255u], 1);
    }
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int qq = 0; qq < kSliceThreads / kWave; ++qq) t += s.hist[qq][0][tid] + s.hist[qq][1][tid];
    lv[((size_t)(level - 2) * pp.n_slices + blockIdx.x) * 256 + tid] = t;
}

// ---- k_plan ---------------------------------------------------------------------------------------
// Workgroup g owns by-patch slots AND sequence positions [4096 g, 4096 g + 4096), 4 of each per thread.
// mode 0 (merge): slot t of the by-patch order (visual slots [0, nv), then the non-visual tail) is folded
//   iff it is selected and t > 0 (slot 0 has no predecessor: the reference would wrap to order[-1],
//   main.py:290; reachable only when top-k exceeds the number of valid pairs).  `inv` is the inverse
//   of the by-patch order (slot of every sequence position), maintained next to `order` by its
//   producers (K0, the hinted similarity kernel, the merge kernel's next-order blocks);
// mode 1 (prune): identity order (inv = NULL), position i in [lo, hi) is DROPPED iff it is not selected.
//
// Latency structure: the kernel is a chain of dependent steps on a cold chip (its predecessor streamed
// hundreds of MB), so every load that does not depend on the decision is issued in the FIRST round:
// the level-0 tables, - speculatively - the level-1 rows of the top byte the k-th key is expected
// to have (pp.p0_guess: the threshold's; a wrong guess costs one more round trip), inv[] of my
// positions, and ALL values (8 consecutive slots per thread per chunk, up to kRegChunks x 8192
// slots kept in registers: the tie slot t* then comes out of registers, whatever slice it is in).
// Wave 0 resolves the levels from LDS.  Each workgroup classifies only its own 4096 slots (member
// flags) and 4096 positions (keep flags, values gathered through inv[]); the number of kept positions
// BEFORE its range comes from the other workgroups' totals: every workgroup publishes its own total
// as ONE 8-byte {tag, count} granule (agent-scope store, the data is the flag) and sums the granules
// of its predecessors - all workgroups publish at about the same time, so this is one hop, not a
// chain.  The tag is a per-workspace launch counter kept on the device (graph-replay safe).
constexpr int kRegChunks = 5;               // 40 960 slots in registers; longer sequences load the rest on the fly
constexpr int kChunkStride = kPlanThreads * 8;
constexpr int kMaxPlanGroups = 240;         // every workgroup must be resident (one per CU): L < 983 040

struct PlanLds {
    SelLds<kPlanThreads> sel;
};

template <int DT>
__device__ inline void chunk_keys(const uint4* kw, uint32_t* key) {
    if constexpr (Act<DT>::kBytes == 2) {
        const uint32_t w[4] = {kw[0].x, kw[0].y, kw[0].z, kw[0].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            key[2 * e] = order_key<DT>(w[e] & 0xffffu);
            key[2 * e + 1] = order_key<DT>(w[e] >> 16);
        }
    } else {
        key[0] = order_key<DT>(kw[0].x); key[1] = order_key<DT>(kw[0].y); key[2] = order_key<DT>(kw[0].z); key[3] = order_key<DT>(kw[0].w);
        key[4] = order_key<DT>(kw[1].x); key[5] = order_key<DT>(kw[1].y); key[6] = order_key<DT>(kw[1].z); key[7] = order_key<DT>(kw[1].w);
    }
}

// An odd number of 2-byte values ends in the middle of a dword, which the range check of the buffer
// loads zeroes as a whole: the last value is loaded on its own (by every thread: a broadcast) and
// patched into the chunk that holds it.
template <int DT>
__device__ inline void patch_last(uint32_t* key, int t0, int cap, uint32_t last_key) {
    if constexpr (Act<DT>::kBytes == 2) {
        if (cap & 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (t0 + e == cap - 1) key[e] = last_key;
        }
    }
}

// raw values of the 8 slots [t0, t0 + 8): raw buffer loads (range-checked per dword: whatever lies
// beyond the array reads 0), so the loads of all chunks are straight-line code and go out back to back
template <int DT>
__device__ inline void load_chunk(__amdgpu_buffer_rsrc_t values, int t0, uint4* kw) {
    constexpr int KW = Act<DT>::kBytes == 2 ? 1 : 2;
#pragma unroll
    for (int x = 0; x < KW; ++x) kw[x] = buf_load16(values, (uint32_t)t0 * (uint32_t)Act<DT>::kBytes + 16u * x);
}

template <int DT>
__global__ __launch_bounds__(kPlanThreads) void k_plan(
    const void* __restrict__ values, int cap, PlanParams pp, const int* __restrict__ l0, int* t16_end,
    const int* __restrict__ lv, int64_t* __restrict__ stats, const int32_t* __restrict__ inv, int L,
    uint8_t* __restrict__ member, uint8_t* __restrict__ keep, int32_t* __restrict__ dst,
    unsigned long long* agg, uint32_t* tagword, int64_t* host_mapped, int64_t seq) {
    using A = Act<DT>;
    constexpr int kLevels = A::kKeyBits / 8;
    constexpr int KW = A::kBytes == 2 ? 1 : 2;
    constexpr int NW = kPlanThreads / kWave;
    __shared__ PlanLds s;
    const int tid = threadIdx.x, lane = lane_id();
    const int base = blockIdx.x * kSelSlice;
#ifdef FF_PLAN_PROBE
    long long stamp[6];
    stamp[0] = wall_clock64();
#endif

    // ---- round 1: everything that does not depend on the decision; the small tables first (loads
    // return in order: the first barrier then only waits for them, not for the value chunks).  Raw
    // values only: any arithmetic on them here would make the compiler wait before the later loads
    // are even issued.
    const long long nv_raw = stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    const uint32_t tag_prev = *tagword;
    const int i0 = base + tid * 4;                // my positions
    const __amdgpu_buffer_rsrc_t inv_rsrc = make_rsrc(inv ? (const void*)inv : values, inv ? (uint32_t)L * 4u : 0u);
    const uint4 inv4 = buf_load16(inv_rsrc, (uint32_t)i0 * 4u);
    const uint32_t last_bits = A::bits1(values, max(cap - 1, 0));
    __builtin_amdgcn_sched_barrier(0);            // the table loads stay AHEAD of the chunk loads in issue order
    const __amdgpu_buffer_rsrc_t val_rsrc = make_rsrc(values, (uint32_t)cap * (uint32_t)A::kBytes);
    uint4 kw[kRegChunks][KW];
#pragma unroll
    for (int x = 0; x < kRegChunks; ++x) load_chunk<DT>(val_rsrc, tid * 8 + x * kChunkStride, kw[x]);
    __builtin_amdgcn_sched_barrier(0);
    // the values of my positions' slots: one dependent gather, in flight while the levels are resolved
    int pslot[4] = {(int)inv4.x, (int)inv4.y, (int)inv4.z, (int)inv4.w};
    if (!inv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pslot[e] = i0 + e;
    }
    uint32_t pbits[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pbits[e] = A::bits1(values, min(max(pslot[e], 0), cap - 1));
    __builtin_amdgcn_sched_barrier(0);
    const int nv = pp.mode == 0 ? (int)nv_raw : L;
    const int lo = pp.mode == 0 ? 0 : pp.lo, hi = pp.mode == 0 ? nv : pp.hi;
    const uint32_t last_key = order_key<DT>(last_bits);

    // the 8 keys of register chunk x (wave-uniform x): select the raw words, extract once
    auto keys_of_chunk = [&](int x, int t0, uint32_t* key) {
        uint4 sel[KW];
#pragma unroll
        for (int w = 0; w < KW; ++w) sel[w] = kw[0][w];
#pragma unroll
        for (int y = 1; y < kRegChunks; ++y)
            if (x == y) {
#pragma unroll
                for (int w = 0; w < KW; ++w) sel[w] = kw[y][w];
            }
        if (x >= kRegChunks) load_chunk<DT>(val_rsrc, t0, sel);
        chunk_keys<DT>(sel, key);
        patch_last<DT>(key, t0, cap, last_key);
    };

    // ---- decision + k-th key -------------------------------------------------------------------------
    Resolved r;
    int sl = 0, want = 0;                         // the want-th (1-based) entry equal to kth inside slice sl is the last one taken
    {
        r = resolve<kPlanThreads>(pp, l0, t16_end, lv, kLevels, ftn, nv, s.sel);
        if (r.topk && r.k > 0) {
            // entries equal to kth per slice: the last level's rows count exactly those
            const int per = (pp.n_slices + kPlanThreads - 1) / kPlanThreads;
            const int g0 = tid * per, g1 = min(g0 + per, pp.n_slices);
            auto ties_of = [&](int g) {
                int cnt = 0;
                if constexpr (kLevels == 2) {
                    const int* row = t16_slice(t16_end, g) + t16_bin(r.prefix);
#pragma unroll
                    for (int x = 0; x < kT16Copies; ++x) cnt += row[x * 65536];
                } else {
                    cnt = lv[((size_t)(kLevels - 3) * pp.n_slices + g) * 256 + (r.prefix & 255u)];
                }
                return cnt;
            };
            int mine = 0;
            for (int g = g0; g < g1; ++g) mine += ties_of(g);
            int total;
            const int before_me = block_excl_scan<NW>(mine, s.sel.scratch, total);
            if (before_me < r.remaining && r.remaining <= before_me + mine) {      // exactly one thread
                int acc = before_me;
                for (int g = g0; g < g1; ++g) {
                    const int cnt = ties_of(g);
                    if (r.remaining <= acc + cnt) { s.sel.bcast[5] = g; s.sel.bcast[6] = r.remaining - acc; break; }
                    acc += cnt;
                }
            }
            __syncthreads();
            sl = s.sel.bcast[5];
            want = s.sel.bcast[6];
        }
    }
    const bool topk = r.topk && r.k > 0;
    const uint32_t kth = r.prefix;
    const int need = r.remaining;                 // entries equal to kth that belong to the top k (>= 1)
#ifdef FF_PLAN_PROBE
    stamp[1] = wall_clock64();
#endif

    // ---- t*: the slot of the want-th entry equal to kth inside slice sl (lowest-index tie rule) ----------
    int tstar = -1;
    if (topk) {
        const int cstar = sl >> 1;                // the chunk index that covers slice sl (chunk stride = 2 slices)
        const int t0 = tid * 8 + cstar * kChunkStride;
        uint32_t key[8];
        keys_of_chunk(cstar, t0, key);
        unsigned tie = 0;
        if (t0 >= sl * kSelSlice && t0 < (sl + 1) * kSelSlice) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int t = t0 + e;
                if (t >= lo && t < hi && key[e] == kth) tie |= 1u << e;
            }
        }
        int slice_total;
        const int ex = block_excl_scan<NW>(__popc(tie), s.sel.scratch, slice_total);
        if (ex < want && want <= ex + __popc(tie)) {
            int seen = ex;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if ((tie >> e) & 1u) { if (++seen == want) s.sel.bcast[0] = t0 + e; }
        }
        __syncthreads();
        tstar = s.sel.bcast[0];
    }
#ifdef FF_PLAN_PROBE
    stamp[2] = wall_clock64();
#endif
    auto folded = [&](uint32_t key, int t) {            // is slot t (value key) folded (mode 0) / dropped (mode 1)?
        const bool in = t >= lo && t < hi;
        bool sel;
        if (r.topk) sel = topk && in && (key > kth || (key == kth && t <= tstar));
        else sel = in && key >= pp.thr_key && key != nan_key<DT>();
        return pp.mode == 0 ? (sel && t > 0) : (in && !sel);
    };
    // ---- member flags of my slots: 8 consecutive slots for half of the threads, from the register chunk
    {
        const int cm = (int)blockIdx.x >> 1;          // the chunk that covers my slice
        const int t0 = tid * 8 + cm * kChunkStride;
        if (t0 >= base && t0 < base + kSelSlice && t0 < L) {
            uint32_t key[8];
            keys_of_chunk(cm, t0, key);
            uint32_t lo4 = 0, hi4 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo4 |= (folded(key[e], t0 + e) ? 1u : 0u) << (8 * e);
                hi4 |= (folded(key[4 + e], t0 + 4 + e) ? 1u : 0u) << (8 * e);
            }
            if (t0 + 8 <= L) {
                *(uint2*)(member + t0) = make_uint2(lo4, hi4);
            } else {
                for (int e = 0; e < 8 && t0 + e < L; ++e) member[t0 + e] = ((e < 4 ? lo4 >> (8 * e) : hi4 >> (8 * (e - 4))) & 1u);
            }
        }
    }
    // ---- keep flags of my positions + their compaction scan inside the workgroup
    uint32_t kb = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i0 + e < L && !folded(order_key<DT>(pbits[e]), pslot[e])) kb |= 1u << (8 * e);
    int span_total;
    const int ex = block_excl_scan<NW>(__popc(kb), s.sel.scratch, span_total);
#ifdef FF_PLAN_PROBE
    stamp[3] = wall_clock64();
#endif
    // ---- kept positions before my range = the totals of the workgroups before me
    uint32_t tag = tag_prev + 1u;
    if (tag == 0u) tag = 1u;
    if (tid == 0)
        __hip_atomic_store(&agg[blockIdx.x], ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)span_total,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < kWave) {
        int sum = 0;
        for (int g0 = 0; g0 < (int)blockIdx.x; g0 += kWave) {
            const int gg = g0 + lane;
            const bool need_it = gg < (int)blockIdx.x;
            unsigned long long v = 0;
            for (int spins = 0;; ++spins) {
                if (need_it) v = __hip_atomic_load(&agg[gg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!need_it || (uint32_t)(v >> 32) == tag)) break;
                __builtin_amdgcn_s_sleep(1);
                if (spins > (1 << 22)) {             // a predecessor never arrived: report, do not hang
                    if (lane == 0) atomicOr((unsigned long long*)(stats + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_BARRIER);
                    break;
                }
            }
            sum += need_it ? (int)(uint32_t)v : 0;
        }
        sum = wave_sum_i(sum);
        if (lane == 0) s.sel.bcast[1] = sum;
    }
    __syncthreads();
    const int before = s.sel.bcast[1];
    int p = before + ex;
    if (i0 < L) {
        int d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kp = (kb >> (8 * e)) & 1u;
            d[e] = kp ? p : -1;
            if (kp && pp.src_out && i0 + e < L) pp.src_out[p] = i0 + e;
            p += kp;
        }
        if (i0 + 4 <= L) {
            *(uint4*)(dst + i0) = make_uint4(d[0], d[1], d[2], d[3]);
            *(uint32_t*)(keep + i0) = kb;
        } else {
            for (int e = 0; e < 4 && i0 + e < L; ++e) { dst[i0 + e] = d[e]; keep[i0 + e] = (kb >> (8 * e)) & 1u; }
        }
    }
    if (blockIdx.x == gridDim.x - 1) {
        if (tid == 0) {
            *tagword = tag;                          // (every workgroup has read the old value: they all published)
#ifdef FF_PLAN_PROBE
            stamp[4] = wall_clock64();
#pragma unroll
            for (int x = 1; x < 5; ++x) stats[FF_STAT_T_PLAN + x] = stamp[x] - stamp[0];
#endif
            const int l_out = before + span_total;
            stats[FF_STAT_LOUT] = l_out;
            stats[FF_STAT_MERGED] = L - l_out;
            if (pp.mode == 0) {
                const double ratio = ftn > 0 ? (double)r.count / (double)ftn : 0.0;
                stats[FF_STAT_COUNT] = r.count;
                stats[FF_STAT_BRANCH] = r.topk ? 1 : 0;
                stats[FF_STAT_BELOW_LB] = (!r.topk && ratio < pp.ratio_lb) ? 1 : 0;
            } else {
                stats[FF_STAT_NV] = L;
            }
            stats[FF_STAT_K] = r.k;
            stats[FF_STAT_KTH_KEY] = kth;
            stats[FF_STAT_TIES_TAKEN] = topk ? need : 0;
        }
        __syncthreads();
        if (host_mapped && tid < kWave) publish(stats, host_mapped, seq);
    }
}

// ---- k_plan_fast: the plan for 16-bit values of at most 163 840 tokens: ff_plan_fast.h ---------------------
#ifndef FF_FAST_THREADS
#define FF_FAST_THREADS 512
#endif
constexpr int kFastThreads = FF_FAST_THREADS;
constexpr int kFastSpan = kFastThreads;            // slots / positions per workgroup

template <int DT, int RS>
__global__ __launch_bounds__(kFastThreads) void k_plan_fast(
    const void* __restrict__ values, int cap, PlanParams pp, const int* __restrict__ l0, int* t16_end,
    int64_t* __restrict__ stats, const int32_t* __restrict__ inv, int L,
    uint8_t* __restrict__ member, uint8_t* __restrict__ keep, int32_t* __restrict__ dst,
    unsigned long long* agg, uint32_t* tagword, int64_t* host_mapped, int64_t seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    plan_fast_body<DT, RS, kFastThreads>(values, cap, pp, l0, t16_end, stats, inv, L, member, keep, dst, agg, tagword,
                                         host_mapped, seq, lds_raw, (int)blockIdx.x, (int)gridDim.x);
}

// ---- explicit merge set (merge_tokens_and_get_mask, main.py:243-319) --------------------------------
__global__ __launch_bounds__(256) void k_mark_index(const int64_t* __restrict__ merge_index, int n_merge,
                                                    const int64_t* __restrict__ stats, uint8_t* __restrict__ member) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_merge) return;
    const int64_t j = merge_index[q];
    if (j > 0 && j < stats[FF_STAT_NV]) member[j] = 1;
}
__global__ __launch_bounds__(256) void k_keep_from_member(const uint8_t* __restrict__ member,
                                                          const int32_t* __restrict__ order, int L,
                                                          uint8_t* __restrict__ keep) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    keep[order ? order[t] : t] = member[t] ? 0 : 1;
}

// dst[] from given keep bytes: workgroup g recounts keep[0, 4096 g) itself and scans its own span
constexpr int kScanThreads = 256;
constexpr int kScanSpan = kScanThreads * kEpt;
__global__ __launch_bounds__(kScanThreads) void k_scan(const uint8_t* __restrict__ keep, int L,
                                                       int32_t* __restrict__ dst, int64_t* __restrict__ stats) {
    __shared__ int scratch[kScanThreads / kWave + 1];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * kScanSpan;
    int before = 0;
    for (int off = tid * 16; off < base; off += kScanThreads * 16) {
        const uint4 k4 = *(const uint4*)(keep + off);
        before += __popc(k4.x) + __popc(k4.y) + __popc(k4.z) + __popc(k4.w);
    }
    before = block_sum_i<kScanThreads / kWave>(before, scratch);
    const int i0 = base + tid * kEpt;
    const int n_here = min(max(L - i0, 0), kEpt);
    uint32_t kb[4] = {0, 0, 0, 0};
    if (n_here == kEpt) {
        const uint4 k4 = *(const uint4*)(keep + i0);
        kb[0] = k4.x; kb[1] = k4.y; kb[2] = k4.z; kb[3] = k4.w;
    } else {
        for (int e = 0; e < n_here; ++e) kb[e >> 2] |= (uint32_t)keep[i0 + e] << (8 * (e & 3));
    }
    const int mine = __popc(kb[0]) + __popc(kb[1]) + __popc(kb[2]) + __popc(kb[3]);
    int span_total;
    int pos = before + block_excl_scan<kScanThreads / kWave>(mine, scratch, span_total);
    for (int e = 0; e < n_here; ++e) {
        const int kp = (kb[e >> 2] >> (8 * (e & 3))) & 1u;
        dst[i0 + e] = kp ? pos : -1;
        pos += kp;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const int l_out = before + span_total;
        stats[FF_STAT_LOUT] = l_out;
        stats[FF_STAT_MERGED] = L - l_out;
    }
}

// ---- launchers (also used by the merge / prune steps in ff_abi.hip) ------------------------------------------
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// Workspace layout: [kL0Ints ints: level-0 tables][16 ints: launch tag][2 x kMaxPlanGroups ints: the
// workgroup totals][2 x G x 256 ints: fp32 level rows][L ints: inverse order of the stand-alone entry
// points][K0's per-slice rows, ff_order.hip] ... free ... [G slices of level-1 tables, down from the
// end];  G = ceil(L / 4096).
constexpr int kWsResBarInts = 1024;                            // barrier words of the one-launch merge kernel (ff_resident.hip)
constexpr int kWsFixedInts = kL0Ints + 16 + 2 * 512 + kWsResBarInts;          // (room for 512 workgroup totals)
size_t plan_ws_front_bytes(int64_t L) {
    const size_t G = (size_t)((L + kSelSlice - 1) / kSelSlice) + 1;
    const size_t b = ((size_t)kWsFixedInts + 2 * G * 256 + (size_t)L + 16) * sizeof(int) + 256;
    return (b + 255) & ~(size_t)255;
}
size_t plan_ws_tail_bytes(int64_t L) {
    const size_t G = (size_t)((L + kSelSlice - 1) / kSelSlice) + 1;
    return G * kT16SliceInts * sizeof(int);
}
int* ws_l0(void* ws) { return (int*)ws; }
int* ws_t16_end(void* ws, size_t ws_bytes) { return (int*)((char*)ws + (ws_bytes & ~(size_t)15)); }
uint32_t* ws_tag(void* ws) { return (uint32_t*)((int*)ws + kL0Ints); }
unsigned long long* ws_agg(void* ws) { return (unsigned long long*)((int*)ws + kL0Ints + 16); }
struct ResBar;
ResBar* ws_resbar(void* ws) { return (ResBar*)((int*)ws + kL0Ints + 16 + 2 * 512); }
static int* ws_levels(void* ws) { return (int*)ws + kWsFixedInts; }
int32_t* ws_scratch_ints(void* ws, int64_t L);
static int32_t* ws_inv(void* ws, int64_t L) {
    const size_t G = (size_t)((L + kSelSlice - 1) / kSelSlice) + 1;
    return (int32_t*)(((uintptr_t)(ws_levels(ws) + 2 * G * 256) + 15) & ~(uintptr_t)15);
}

// [L] ints of the workspace that only the stand-alone plan entry points use (their inverse order): free scratch
// for the fused step's followers (the attention-mask gather)
int32_t* ws_scratch_ints(void* ws, int64_t L) { return ws_inv(ws, L); }

__global__ __launch_bounds__(256) void k_invert(const int32_t* __restrict__ order, int L, int32_t* __restrict__ inv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < L) inv[order[t]] = t;
}

// the regions a call over L values dirties (zeroed again by the merge kernel or by memsets)
void table_regions(void* ws, size_t ws_bytes, int64_t L, void** a, size_t* a_bytes, void** b, size_t* b_bytes) {
    const size_t G = (size_t)cdiv(L, kSelSlice);
    *a = ws;
    *a_bytes = (size_t)kL0Ints * sizeof(int);
    *b = t16_slice(ws_t16_end(ws, ws_bytes), (int)G - 1);
    *b_bytes = G * kT16SliceInts * sizeof(int);
}
int zero_tables(void* ws, size_t ws_bytes, int64_t L, hipStream_t st) {
    void *a, *b;
    size_t ab, bb;
    table_regions(ws, ws_bytes, L, &a, &ab, &b, &bb);
    hipError_t e = hipMemsetAsync(a, 0, ab, st);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bb, st);
    return (int)e;
}

template <int DT>
static uint32_t host_thr_key(double thr);   // order-preserving key of T(thr) computed on the host
template <> uint32_t host_thr_key<FF_F32>(double thr) {
    float f = thr == 0.0 ? -0.0f : (float)thr;
    uint32_t b; memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
template <> uint32_t host_thr_key<FF_BF16>(double thr) {
    float f = thr == 0.0 ? -0.0f : (float)thr;   // already bf16-valued: low 16 bits are zero
    uint32_t b; memcpy(&b, &f, 4);
    b >>= 16;
    return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}
template <> uint32_t host_thr_key<FF_F16>(double thr) {
    _Float16 h = (_Float16)(thr == 0.0 ? -0.0f : (float)thr);
    uint16_t b; memcpy(&b, &h, 2);
    return (b & 0x8000u) ? ((uint32_t)(~b) & 0xffffu) : ((uint32_t)b | 0x8000u);
}

// values + tables -> member, keep, dst, stats.  have_tables == false: build them here (k_tables) and
// restore the zero state afterwards (memsets) - the stand-alone entry points.
template <int DT>
static int launch_plan(const void* values, PlanParams pp, bool have_tables, int64_t cap, int64_t L,
                       const int32_t* inv, uint8_t* member, uint8_t* keep, int32_t* dst, int64_t* stats,
                       void* ws, size_t ws_bytes, int64_t* host_mapped, int64_t seq, hipStream_t st) {
    constexpr int kLevels = Act<DT>::kKeyBits / 8;
    const unsigned G = cdiv(L, kSelSlice);
    if (G > (unsigned)kMaxPlanGroups) return FF_ERR_UNSUPPORTED;      // every workgroup must be resident
    pp.n_slices = (int)G;
    int* l0 = ws_l0(ws);
    int* t16_end = ws_t16_end(ws, ws_bytes);
    int* lv = ws_levels(ws);
    if (!have_tables) {
        int rc = zero_tables(ws, ws_bytes, L, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tables<DT>, dim3(G), dim3(kSliceThreads), 0, st, values, (int)cap, pp.mode == 0 ? 0 : pp.lo,
                           pp.mode == 0 ? -1 : pp.hi, (const int64_t*)stats, pp.thr_key, l0, t16_end);
    }
    for (int level = 2; level < kLevels; ++level)
        hipLaunchKernelGGL(k_hist_level<DT>, dim3(G), dim3(kSliceThreads), 0, st, values, (int)cap, pp, level,
                           (const int64_t*)stats, (const int*)l0, t16_end, lv);
    if constexpr (kLevels == 2) {
        if (L <= kFastMaxL) {
            // per-device, per-instantiation: the dynamic LDS limit is an attribute of the function ON a device
            static std::atomic<bool> attr_set[kMaxDevices][3][2];
            const bool big = L > (int64_t)kRowSlicesLds * kSelSlice;
            const int rs = big ? kFastSlicesBig : kRowSlicesLds;
            const size_t lds = plan_fast_lds_bytes(rs, kFastThreads);
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
            if (dev < 0 || !attr_set[dev][DT][big].load(std::memory_order_acquire)) {
                const void* fn = big ? (const void*)k_plan_fast<DT, kFastSlicesBig> : (const void*)k_plan_fast<DT, kRowSlicesLds>;
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
                if (e != hipSuccess) return (int)e;
                if (dev >= 0) attr_set[dev][DT][big].store(true, std::memory_order_release);
            }
            if (big)
                hipLaunchKernelGGL((k_plan_fast<DT, kFastSlicesBig>), dim3(cdiv(L, kFastSpan)), dim3(kFastThreads), lds, st, values,
                                   (int)cap, pp, (const int*)l0, t16_end, stats, inv, (int)L, member, keep, dst, ws_agg(ws),
                                   ws_tag(ws), host_mapped, seq);
            else
                hipLaunchKernelGGL((k_plan_fast<DT, kRowSlicesLds>), dim3(cdiv(L, kFastSpan)), dim3(kFastThreads), lds, st, values,
                                   (int)cap, pp, (const int*)l0, t16_end, stats, inv, (int)L, member, keep, dst, ws_agg(ws),
                                   ws_tag(ws), host_mapped, seq);
            int rc0 = (int)hipGetLastError();
            if (rc0) return rc0;
            if (!have_tables) return zero_tables(ws, ws_bytes, L, st);
            return FF_OK;
        }
    }
    hipLaunchKernelGGL(k_plan<DT>, dim3(G), dim3(kPlanThreads), 0, st, values, (int)cap, pp, (const int*)l0, t16_end,
                       (const int*)lv, stats, inv, (int)L, member, keep, dst, ws_agg(ws), ws_tag(ws), host_mapped, seq);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    if (!have_tables) return zero_tables(ws, ws_bytes, L, st);
    return FF_OK;
}

// the plan parameters of a merge call (threshold key, expected top byte of the k-th key); n_slices is the launcher's
PlanParams merge_plan_params(int dtype, double thr, double sub, double ratio_lb, long long force_k) {
    PlanParams pp;
    pp.mode = 0; pp.lo = 0; pp.hi = -1; pp.k_given = force_k; pp.sub = sub; pp.ratio_lb = ratio_lb; pp.n_slices = 0;
    pp.src_out = nullptr;
    // the k-th similarity of a video sits in the binade of typical thresholds: [0.5, 1)
    const double guess_value = force_k >= 0 ? 0.75 : thr;
    pp.p0_guess = (int)(dtype == FF_F32 ? host_thr_key<FF_F32>(guess_value) >> 24
                        : dtype == FF_BF16 ? host_thr_key<FF_BF16>(guess_value) >> 8 : host_thr_key<FF_F16>(guess_value) >> 8);
    pp.thr_key = dtype == FF_F32 ? host_thr_key<FF_F32>(thr) : dtype == FF_BF16 ? host_thr_key<FF_BF16>(thr) : host_thr_key<FF_F16>(thr);
    return pp;
}

// inv == nullptr: the inverse of `order` is built here (stand-alone entry points)
int launch_plan_merge(const void* sim, int dtype, const int32_t* order, const int32_t* inv, int64_t L, double thr,
                      double sub, double ratio_lb, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, size_t ws_bytes, bool have_tables, int64_t* host_mapped, int64_t seq,
                      hipStream_t st, long long force_k) {
    if (!inv) {
        int32_t* tmp = ws_inv(ws, L);
        hipLaunchKernelGGL(k_invert, dim3(cdiv(L, 256)), dim3(256), 0, st, order, (int)L, tmp);
        inv = tmp;
    }
    const PlanParams pp = merge_plan_params(dtype, thr, sub, ratio_lb, force_k);
    switch (dtype) {
        case FF_F32: return launch_plan<FF_F32>(sim, pp, have_tables, L, L, inv, member, keep, dst, stats, ws, ws_bytes, host_mapped, seq, st);
        case FF_BF16: return launch_plan<FF_BF16>(sim, pp, have_tables, L, L, inv, member, keep, dst, stats, ws, ws_bytes, host_mapped, seq, st);
        default: return launch_plan<FF_F16>(sim, pp, have_tables, L, L, inv, member, keep, dst, stats, ws, ws_bytes, host_mapped, seq, st);
    }
}

// the plan parameters of a prune call (top-k over [start, start + n_img)); n_slices is the launcher's
PlanParams prune_plan_params(int dtype, int64_t start, int64_t n_img, int64_t k) {
    PlanParams pp;
    pp.mode = 1; pp.lo = (int)start; pp.hi = (int)(start + n_img); pp.k_given = k; pp.sub = 0; pp.ratio_lb = 0;
    pp.thr_key = 0xffffffffu; pp.n_slices = 0;
    pp.src_out = nullptr;
    // importances are probabilities of ~1/S: guess the binade of 1/n_img
    const double guess_value = n_img > 0 ? 1.0 / (double)n_img : 1.0;
    pp.p0_guess = (int)(dtype == FF_F32 ? host_thr_key<FF_F32>(guess_value) >> 24
                        : dtype == FF_BF16 ? host_thr_key<FF_BF16>(guess_value) >> 8 : host_thr_key<FF_F16>(guess_value) >> 8);
    return pp;
}

// Also writes src[] = the inverse of dst[] (position of every output row) into ws_scratch_ints(ws, S): the prune's gather walks it
int launch_plan_prune(const void* imp, int dtype, int64_t S, int64_t start, int64_t n_img, int64_t k,
                      uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws, size_t ws_bytes,
                      bool have_tables, hipStream_t st) {
    PlanParams pp = prune_plan_params(dtype, start, n_img, k);
    pp.src_out = ws_scratch_ints(ws, S);
    switch (dtype) {
        case FF_F32: return launch_plan<FF_F32>(imp, pp, have_tables, S, S, nullptr, member, keep, dst, stats, ws, ws_bytes, nullptr, 0, st);
        case FF_BF16: return launch_plan<FF_BF16>(imp, pp, have_tables, S, S, nullptr, member, keep, dst, stats, ws, ws_bytes, nullptr, 0, st);
        default: return launch_plan<FF_F16>(imp, pp, have_tables, S, S, nullptr, member, keep, dst, stats, ws, ws_bytes, nullptr, 0, st);
    }
}

// dst[] (row of every kept position, -1 else) + stats[LOUT / MERGED] from keep bytes (0 / 1)
int launch_scan_keep(const uint8_t* keep, int64_t L, int32_t* dst, int64_t* stats, hipStream_t st) {
    hipLaunchKernelGGL(k_scan, dim3(cdiv(L, kScanSpan)), dim3(kScanThreads), 0, st, keep, (int)L, dst, stats);
    return (int)hipGetLastError();
}

int launch_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                           uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, hipStream_t st) {
    hipError_t e = hipMemsetAsync(member, 0, (size_t)L, st);
    if (e != hipSuccess) return (int)e;
    if (n_merge > 0)
        hipLaunchKernelGGL(k_mark_index, dim3(cdiv(n_merge, 256)), dim3(256), 0, st, merge_index, (int)n_merge, stats, member);
    hipLaunchKernelGGL(k_keep_from_member, dim3(cdiv(L, 256)), dim3(256), 0, st, member, order, (int)L, keep);
    hipLaunchKernelGGL(k_scan, dim3(cdiv(L, kScanSpan)), dim3(kScanThreads), 0, st, (const uint8_t*)keep, (int)L, dst, stats);
    return (int)hipGetLastError();
}

}  // namespace ff

static int check_plan_args(const void* a, const void* b, const void* c, const void* d, const void* e,
                           int64_t L, void* ws, size_t ws_bytes) {
    if (!a || !b || !c || !d || !e || !ws || L < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31) - 65536) return FF_ERR_UNSUPPORTED;
    if (ws_bytes < ff_workspace_bytes(L, 1)) return FF_ERR_WORKSPACE;
    return FF_OK;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int ff_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double threshold,
                             double sub, double ratio_lb, uint8_t* member, int32_t* dst, uint8_t* keep,
                             int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(sim, member, dst, keep, stats, L, ws, ws_bytes);
    if (rc) return rc;
    if (!order) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (!aligned16(sim) || !aligned16(dst) || !aligned16(keep) || !aligned16(ws) || !aligned16(order) ||
        ((uintptr_t)member & 7))
        return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    return ff::launch_plan_merge(sim, dtype, order, nullptr, L, threshold, sub, ratio_lb, member, dst, keep, stats, ws,
                                 ws_bytes, false, nullptr, 0, (hipStream_t)stream, -1);
}

extern "C" int ff_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                                  uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws,
                                  size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(order, member, dst, keep, stats, L, ws, ws_bytes);
    if (rc) return rc;
    if (n_merge < 0 || (n_merge > 0 && !merge_index)) return FF_ERR_ARG;
    if (!aligned16(dst) || !aligned16(keep)) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    return ff::launch_plan_from_index(merge_index, n_merge, order, L, member, dst, keep, stats, (hipStream_t)stream);
}

