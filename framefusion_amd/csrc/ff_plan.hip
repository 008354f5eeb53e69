// K2+K3 - the integer "plan" between the two streaming passes: threshold count, budget decision,
// top-k (radix select with a lowest-index tie rule), member flags, keep mask and the compaction
// scan.  Everything here works on <= L 2-byte similarities and L-sized index arrays (a few hundred
// KB, L2-resident); the branch the reference takes on the host after two .item() syncs
// (framefusion/main.py:112-127) is decided on the device, in double, exactly as python does.
//
// Replaces: main.py:112-127 (select), find_contigious_latter_index (main.py:351-380), the
// unique/where/repeat_interleave index algebra of merge_tokens_and_get_mask (main.py:269-301)
// and the keep-mask construction (main.py:278-279) - 17 host syncs in the reference.
//
// Launches, split by what is inherently global (details at "multi-workgroup select" below):
//   k_hist_level  per 4096-value slice: one byte of the radix select per launch (level 0 comes for
//                 free from the similarity kernel in the fused path); nothing runs in the
//                 threshold branch beyond re-deriving the decision;
//   k_flags       per slice: member[t] for every by-patch slot and the scatter
//                 keep[order[t]] = !member[t] - one CU would take a cycle per scattered byte,
//                 many CUs do not notice it;
//   k_scan        many workgroups, no communication: workgroup g recounts keep[0, 4096 g) itself
//                 (<= L bytes, L2-resident) and scans its own 4096 positions into dst[]; the last
//                 workgroup knows L_out and publishes the result block.
// Run lengths are not materialised: the merge kernel derives them from 64 member flags at a time.
#include <stdlib.h>
#include <string.h>

#include "ff_common.h"

namespace ff {

constexpr int kEpt = 16;                          // values per thread: two/four 16-byte loads

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

template <int DT>
__device__ inline void load_keys(const void* __restrict__ v, int j0, int n, uint32_t* key, uint32_t& valid_mask) {
    const RawKeys<DT> r = load_raw<DT>(v, j0, n);
    keys_of<DT>(r, j0, n, key, valid_mask);
}

// key of a T-valued float (the threshold): entries with key >= this and not NaN satisfy sim >= thr
template <int DT>
__device__ inline uint32_t key_of_value(float x) {
    if constexpr (DT == FF_F32) return order_key<DT>(__float_as_uint(x));
    else if constexpr (DT == FF_BF16) return order_key<DT>(__float_as_uint(x) >> 16);
    else { _Float16 h = (_Float16)x; return order_key<DT>((uint32_t)__builtin_bit_cast(uint16_t, h)); }
}

template <int DT> __device__ inline uint32_t nan_key() { return Act<DT>::kKeyBits == 32 ? 0xffffffffu : 0xffffu; }

// ---- multi-workgroup select ---------------------------------------------------------------------
// The decision data is a handful of small histograms, so instead of one workgroup walking all
// similarities several times, every workgroup of every stage re-derives the (deterministic)
// decision from the partial histograms written by the previous stage:
//   level 0 : top-byte histogram + count(sim >= thr), accumulated by the similarity kernel itself
//             (64 copies, atomics spread over its whole run) or by k_hist_level(0);
//   level l : k_hist_level(l) - slice g (4096 values) histograms byte l of the keys that match the
//             l-byte prefix of the k-th key, one row per slice, no atomics across workgroups;
//   k_flags : resolves the k-th key from all levels; the last level's per-slice rows also give the
//             number of entries equal to it in earlier slices (lowest-index tie rule without a
//             global scan); then member flags + the keep scatter for its slice.
constexpr int kSliceThreads = 256;
constexpr int kSlice = kSliceThreads * kEpt;      // 4096 values per workgroup
constexpr int kRowStride = 260;                   // 256 bins + count + pad
constexpr int kL0Copies = 16;

struct PlanParams {
    int mode;            // 0: merge (threshold / top-k decided from the count), 1: prune (top-k given)
    int lo, hi;          // value range the selection runs over (merge: hi < 0 -> [0, Nv))
    long long k_given;   // prune: k; merge: >= 0 forces top-k with this k, -1 = threshold/budget policy
    double sub, ratio_lb;
    uint32_t thr_key;
    int l0_rows;         // rows of the level-0 table (64 copies or G slices)
    int n_slices;
};

struct Resolved {
    bool topk;
    long long k;
    int count;
    uint32_t prefix;     // leading `levels_done` bytes of the k-th key
    int remaining;       // entries still to take inside the prefix
};

struct SliceLds {
    int hist[kSliceThreads / kWave][2][256];
    int tot[256];
    int scratch[kSliceThreads / kWave + 1];
    int bcast[4];
};

// pick the bin, from the top, in which the running count reaches `remaining` (s.tot filled)
__device__ inline int pick_from_tot(SliceLds& s, int remaining, int& above) {
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

// Column `tid` of the level-0 table summed over its rows, plus this thread's share of the count
// column.  Straight-line loads (all 64 in flight) for the table the similarity kernel fills.
__device__ inline void sum_l0(const PlanParams& pp, const int* __restrict__ l0, int& col, int& cnt_part) {
    const int tid = threadIdx.x;
    int t0 = 0;
    if (pp.l0_rows == kL0Copies) {
        int v[kL0Copies];
#pragma unroll
        for (int q = 0; q < kL0Copies; ++q) v[q] = l0[q * kRowStride + tid];
#pragma unroll
        for (int q = 0; q < kL0Copies; ++q) t0 += v[q];
    } else {
#pragma unroll 16
        for (int q = 0; q < pp.l0_rows; ++q) t0 += l0[q * kRowStride + tid];
    }
    int c = 0;
    for (int q = tid; q < pp.l0_rows; q += kSliceThreads) c += l0[q * kRowStride + 256];
    col = t0;
    cnt_part = c;
}

// Column `tid` of every row of level table `l` (1-based) into regs[0..n_slices) is not possible with
// a runtime count, so levels are summed on the fly; `before` = the rows of slices < my_slice only.
__device__ inline void sum_level(const int* __restrict__ tab, int n_slices, int my_slice, int& all, int& before) {
    const int tid = threadIdx.x;
    int a = 0, b = 0;
#pragma unroll 16
    for (int g = 0; g < n_slices; ++g) {
        const int x = tab[g * 256 + tid];
        a += x;
        b += g < my_slice ? x : 0;
    }
    all = a;
    before = b;
}

// Re-derive the decision and the first `levels` bytes of the k-th key from the partial tables.
// l0col / l0cnt: this thread's sums from sum_l0; lvcol[l-1]: column sums of level l (sum_level).
__device__ inline Resolved resolve(const PlanParams& pp, int l0col, int l0cnt, const int* lvcol, int levels,
                                   long long ftn, int nv, SliceLds& s) {
    const int tid = threadIdx.x;
    Resolved r;
    r.prefix = 0; r.remaining = 0; r.count = 0;
    s.tot[tid] = l0col;
    const int c = block_sum_i<kSliceThreads / kWave>(l0cnt, s.scratch);
    r.count = c;
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
    if (!r.topk || r.k <= 0 || levels <= 0) return r;
    r.remaining = (int)r.k;
    for (int l = 0; l < levels; ++l) {
        if (l > 0) {
            __syncthreads();
            s.tot[tid] = lvcol[l - 1];
        }
        int above;
        const int bin = pick_from_tot(s, r.remaining, above);
        r.prefix = (r.prefix << 8) | (uint32_t)bin;
        r.remaining -= above;
    }
    return r;
}

__device__ inline void slice_range(const PlanParams& pp, int nv, int& lo, int& hi) {
    lo = pp.mode == 0 ? 0 : pp.lo;
    hi = pp.mode == 0 ? nv : pp.hi;
}

// level `level` histogram of slice blockIdx.x (level 0 also counts values >= thr)
template <int DT>
__device__ inline void hist_level_body(const void* values, int cap, const PlanParams& pp, int level,
                                       const int64_t* stats, const int* l0, int* lv, int* l0_out, SliceLds& s) {
    using A = Act<DT>;
    const int tid = threadIdx.x, w = wave_id(), cp = lane_id() & 1;
    // every load this workgroup needs is independent of the others: issue them all first
    const int j0 = blockIdx.x * kSlice + tid * kEpt;
    const RawKeys<DT> raw = load_raw<DT>(values, j0, cap);
    const int nv = (int)stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    int l0col = 0, l0cnt = 0, lvcol[3] = {0, 0, 0}, dummy;
    if (level > 0) {
        sum_l0(pp, l0, l0col, l0cnt);
        for (int l = 1; l < level; ++l) sum_level(lv + (size_t)(l - 1) * pp.n_slices * 256, pp.n_slices, 0, lvcol[l - 1], dummy);
    }
    int lo, hi;
    slice_range(pp, nv, lo, hi);
    uint32_t prefix = 0;
    if (level > 0) {
        const Resolved r = resolve(pp, l0col, l0cnt, lvcol, level, ftn, nv, s);
        if (!r.topk || r.k <= 0) return;          // uniform over the whole grid
        prefix = r.prefix;
        __syncthreads();
    }
    for (int x = tid; x < (kSliceThreads / kWave) * 2 * 256; x += kSliceThreads) (&s.hist[0][0][0])[x] = 0;
    __syncthreads();
    uint32_t key[kEpt], valid;
    keys_of<DT>(raw, j0, hi, key, valid);
    const int shift = A::kKeyBits - 8 * (level + 1);
    int c = 0;
#pragma unroll
    for (int e = 0; e < kEpt; ++e) {
        const bool in = ((valid >> e) & 1u) && (j0 + e >= lo);
        if (!in) continue;
        if (level == 0) {
            c += (key[e] >= pp.thr_key && key[e] != nan_key<DT>()) ? 1 : 0;
            atomicAdd(&s.hist[w][cp][key[e] >> shift], 1);
        } else if ((key[e] >> (shift + 8)) == prefix) {
            atomicAdd(&s.hist[w][cp][(key[e] >> shift) & 255u], 1);
        }
    }
    if (level == 0) c = block_sum_i<kSliceThreads / kWave>(c, s.scratch);
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int q = 0; q < kSliceThreads / kWave; ++q) t += s.hist[q][0][tid] + s.hist[q][1][tid];
    if (level == 0) {
        l0_out[blockIdx.x * kRowStride + tid] = t;
        if (tid == 0) l0_out[blockIdx.x * kRowStride + 256] = c;
    } else {
        lv[((size_t)(level - 1) * pp.n_slices + blockIdx.x) * 256 + tid] = t;
    }
}

template <int DT>
__global__ __launch_bounds__(kSliceThreads) void k_hist_level(
    const void* __restrict__ values, int cap, PlanParams pp, int level, const int64_t* __restrict__ stats,
    const int* __restrict__ l0, int* __restrict__ lv, int* __restrict__ l0_out) {
    __shared__ SliceLds s;
    hist_level_body<DT>(values, cap, pp, level, stats, l0, lv, l0_out, s);
}

// ---- k_flags ----------------------------------------------------------------------------------------
// Slice g of the by-patch slots (then the non-visual tail of `order`).  Slot 0 never folds: it has
// no predecessor (the reference would wrap to order[-1], main.py:290; reachable only when top-k
// exceeds the number of valid pairs).
template <int DT>
__device__ inline void flags_body(const void* values, int cap, const PlanParams& pp, const int* l0, const int* lv,
                                  int64_t* stats, const int32_t* order, int L, uint8_t* member, uint8_t* keep,
                                  SliceLds& s) {
    using A = Act<DT>;
    constexpr int kLevels = A::kKeyBits / 8;
    const int tid = threadIdx.x;
    // every load is independent of the others: values, row indices, counts and all tables first
    const int j0 = blockIdx.x * kSlice + tid * kEpt;
    const RawKeys<DT> raw = load_raw<DT>(values, j0, cap);
    int ord[kEpt];
    const int n_here = min(max(L - j0, 0), kEpt);
    if (order) {
        if (n_here == kEpt) {
            const uint4* po = (const uint4*)(order + j0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 o4 = po[q];
                ord[4 * q] = o4.x; ord[4 * q + 1] = o4.y; ord[4 * q + 2] = o4.z; ord[4 * q + 3] = o4.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < kEpt; ++e) ord[e] = e < n_here ? order[j0 + e] : 0;
        }
    } else {
#pragma unroll
        for (int e = 0; e < kEpt; ++e) ord[e] = j0 + e;
    }
    const int nv = (int)stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    int l0col, l0cnt, lvcol[3] = {0, 0, 0}, lvbefore[3] = {0, 0, 0};
    sum_l0(pp, l0, l0col, l0cnt);
#pragma unroll
    for (int l = 1; l < kLevels; ++l)
        sum_level(lv + (size_t)(l - 1) * pp.n_slices * 256, pp.n_slices, (int)blockIdx.x, lvcol[l - 1], lvbefore[l - 1]);
    int lo, hi;
    slice_range(pp, nv, lo, hi);
    const Resolved r = resolve(pp, l0col, l0cnt, lvcol, kLevels, ftn, nv, s);
    const bool select_topk = r.topk;
    const uint32_t kth = r.prefix;
    const int need = r.remaining;                 // entries equal to kth that belong to the top k
    // entries equal to kth in earlier slices: the last level's rows count exactly those, and the
    // thread that owns column (kth & 255) already holds their sum
    __syncthreads();
    if (tid == (int)(kth & 255u)) s.bcast[3] = lvbefore[kLevels - 2];
    __syncthreads();
    const int ties_before = (select_topk && r.k > 0) ? s.bcast[3] : 0;
    const int n_flag = pp.mode == 0 ? nv : L;
    uint32_t key[kEpt], valid;
    keys_of<DT>(raw, j0, n_flag, key, valid);
    int mine = 0;
    if (select_topk && r.k > 0) {
#pragma unroll
        for (int e = 0; e < kEpt; ++e)
            mine += (((valid >> e) & 1u) && j0 + e >= lo && j0 + e < hi && key[e] == kth) ? 1 : 0;
    }
    int slice_ties;
    int rank = ties_before + block_excl_scan<kSliceThreads / kWave>(mine, s.scratch, slice_ties);
    uint32_t mem = 0;
#pragma unroll
    for (int e = 0; e < kEpt; ++e) {
        const int j = j0 + e;
        if (!((valid >> e) & 1u)) continue;
        const bool in = j >= lo && j < hi;
        bool sel;
        if (select_topk) {
            sel = false;
            if (in && r.k > 0) {
                if (key[e] > kth) sel = true;
                else if (key[e] == kth) { sel = rank < need; ++rank; }
            }
        } else {
            sel = key[e] >= pp.thr_key && key[e] != nan_key<DT>();
        }
        const bool m = pp.mode == 0 ? (sel && j > 0) : (in && !sel);
        if (m) mem |= 1u << e;
    }
    if (j0 < L) {
        if (n_here == kEpt && (((uintptr_t)(member + j0)) & 15) == 0) {
            uint32_t mb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mb[q] = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) mb[q] |= ((mem >> (4 * q + bb)) & 1u) << (8 * bb);
            }
            *(uint4*)(member + j0) = make_uint4(mb[0], mb[1], mb[2], mb[3]);
        } else {
            for (int e = 0; e < n_here; ++e) member[j0 + e] = (mem >> e) & 1u;
        }
#pragma unroll
        for (int e = 0; e < kEpt; ++e)
            if (e < n_here) keep[ord[e]] = ((mem >> e) & 1u) ? 0 : 1;
    }
    if (blockIdx.x == 0 && tid == 0) {
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
        stats[FF_STAT_TIES_TAKEN] = (select_topk && r.k > 0) ? need : 0;
    }
}

template <int DT>
__global__ __launch_bounds__(kSliceThreads) void k_flags(
    const void* __restrict__ values, int cap, PlanParams pp, const int* __restrict__ l0, const int* __restrict__ lv,
    int64_t* __restrict__ stats, const int32_t* __restrict__ order, int L,
    uint8_t* __restrict__ member, uint8_t* __restrict__ keep) {
    __shared__ SliceLds s;
    flags_body<DT>(values, cap, pp, l0, lv, stats, order, L, member, keep, s);
}

// Explicit merge set (merge_tokens_and_get_mask, main.py:243-319): member bytes were zeroed, set them.
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

// ---- k_scan -------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanSpan = kScanThreads * kEpt;     // 4096 positions per workgroup

// Copy of the result block into device-visible pinned host memory: one lane per word (a single
// store instruction crosses PCIe once), a system fence, then the sequence word the host polls.
__device__ inline void publish(int64_t* __restrict__ stats, int64_t* host_mapped, int64_t seq) {
    const int lane = threadIdx.x;          // called by wave 0
    if (lane < FF_STAT_WORDS && lane != FF_STAT_SEQ)
        __hip_atomic_store(&host_mapped[lane], stats[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane == FF_STAT_ERROR) stats[lane] = 0;     // reported; the next call starts clean
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(&host_mapped[FF_STAT_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ inline void scan_body(const uint8_t* keep, int L, int32_t* dst, int64_t* stats, int64_t* host_mapped,
                                 int64_t seq, int* zero_me, int* scratch) {
    const int tid = threadIdx.x;
    // the level-0 statistics table the NEXT call's similarity kernel will accumulate into
    if (zero_me)
        for (int x = blockIdx.x * kScanThreads + tid; x < kL0Copies * kRowStride; x += gridDim.x * kScanThreads)
            zero_me[x] = 0;
    const int base = blockIdx.x * kScanSpan;
    // kept positions before my span: keep bytes are 0/1, so popcount of the words counts them
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
    if (n_here > 0) {
        int d[kEpt];
#pragma unroll
        for (int e = 0; e < kEpt; ++e) {
            const int kp = (kb[e >> 2] >> (8 * (e & 3))) & 1u;
            d[e] = kp ? pos : -1;
            pos += kp;
        }
        if (n_here == kEpt) {
            uint4* p = (uint4*)(dst + i0);
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = make_uint4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
        } else {
#pragma unroll
            for (int e = 0; e < kEpt; ++e)
                if (e < n_here) dst[i0 + e] = d[e];
        }
    }
    if (blockIdx.x == gridDim.x - 1) {
        if (tid == 0) {
            const int l_out = before + span_total;
            stats[FF_STAT_LOUT] = l_out;
            stats[FF_STAT_MERGED] = L - l_out;
        }
        __syncthreads();
        if (host_mapped && tid < kWave) publish(stats, host_mapped, seq);
    }
}

__global__ __launch_bounds__(kScanThreads) void k_scan(const uint8_t* __restrict__ keep, int L,
                                                       int32_t* __restrict__ dst, int64_t* __restrict__ stats,
                                                       int64_t* host_mapped, int64_t seq, int* __restrict__ zero_me) {
    __shared__ int scratch[kScanThreads / kWave + 1];
    scan_body(keep, L, dst, stats, host_mapped, seq, zero_me, scratch);
}

// ---- fused plan: the radix levels, the flags and the scan in ONE launch ----------------------------------
// The three stages need the results of ALL workgroups of the previous stage, so they are separated by
// a grid barrier instead of a kernel boundary: only ceil(L / 4096) workgroups exist (9 at 64 x 576),
// all co-resident on a 256-CU chip, and a kernel boundary costs a cold start per stage (the two
// streaming passes flush the instruction lines of these tiny kernels out of L2 every call).
// Barrier = the counter form of the release/acquire hand-off (cdna_hip_programming.md, G16): stores ->
// __syncthreads -> lane 0: agent release + vmcnt(0) + relaxed agent add, relaxed poll with s_sleep,
// agent acquire -> __syncthreads -> plain loads.  The counter is monotonic within a call and the
// last stage clears the other parity's counter for the next call; a bounded spin turns a lost
// workgroup into an error word instead of a hang.
constexpr int kMaxFusedSlices = 64;

__device__ inline void grid_barrier(int* counter, int target, int64_t* stats) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 24)) { stats[FF_STAT_ERROR] = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int DT>
__global__ __launch_bounds__(kSliceThreads) void k_plan_fused(
    const void* values, int cap, PlanParams pp, const int* l0, int* lv, int64_t* stats, const int32_t* order, int L,
    uint8_t* member, uint8_t* keep, int32_t* dst, int64_t* host_mapped, int64_t seq, int* zero_l0,
    int* bar, int* bar_next) {
    constexpr int kLevels = Act<DT>::kKeyBits / 8;
    __shared__ SliceLds s;
    const int G = (int)gridDim.x;
    int phase = 0;
    for (int level = 1; level < kLevels; ++level) {
        hist_level_body<DT>(values, cap, pp, level, stats, l0, lv, (int*)nullptr, s);
        grid_barrier(bar, ++phase * G, stats);
    }
    flags_body<DT>(values, cap, pp, l0, lv, stats, order, L, member, keep, s);
    grid_barrier(bar, ++phase * G, stats);
    if (blockIdx.x == 0 && threadIdx.x == 0) *bar_next = 0;
    scan_body(keep, L, dst, stats, host_mapped, seq, zero_l0, s.scratch);
}

// ---- launchers (also used by the fused step in ff_abi.hip) ------------------------------------------
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// Workspace layout (ints): [2][kL0Copies][kRowStride] level-0 tables filled by the similarity kernel
// (double-buffered by call parity), 64 ints for the grid-barrier counters, [G][kRowStride] level-0 rows for the stand-alone entry points,
// [3][G][256] rows of levels 1..3;  G = ceil(L / 4096).
constexpr int kBarrierInts = 64;
size_t plan_ws_bytes(int64_t L) {
    const size_t G = (size_t)((L + kSlice - 1) / kSlice) + 1;
    const size_t b = (2 * (size_t)kL0Copies * kRowStride + kBarrierInts + G * kRowStride + 3 * G * 256) * sizeof(int) + 256;
    return (b + 255) & ~(size_t)255;
}
// the two grid-barrier counters sit at a FIXED offset (they persist across calls of any length)
static int* ws_barrier(void* ws, int64_t seq) {
    return (int*)ws + 2 * (size_t)kL0Copies * kRowStride + 16 * (seq & 1);
}
int* ws_l0_copies(void* ws, int64_t seq) { return (int*)ws + (size_t)(seq & 1) * kL0Copies * kRowStride; }
static int* ws_l0_rows(void* ws) { return (int*)ws + 2 * (size_t)kL0Copies * kRowStride + kBarrierInts; }
static int* ws_levels(void* ws, int64_t L) {
    const size_t G = (size_t)((L + kSlice - 1) / kSlice) + 1;
    return ws_l0_rows(ws) + G * kRowStride;
}

static int launch_scan(const uint8_t* keep, int64_t L, int32_t* dst, int64_t* stats, int64_t* host_mapped,
                       int64_t seq, int* zero_me, hipStream_t st) {
    hipLaunchKernelGGL(k_scan, dim3(cdiv(L, kScanSpan)), dim3(kScanThreads), 0, st, keep, (int)L, dst, stats,
                       host_mapped, seq, zero_me);
    return (int)hipGetLastError();
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

struct FusedTail {          // what the fused launch needs to also run the scan stage
    int32_t* dst;
    int64_t* host_mapped;
    int64_t seq;
    int* zero_l0;
    int* bar;
    int* bar_next;
    bool done;
};

// values/selection -> member, keep (+ dst when fused).  l0 == nullptr: level 0 is computed here.
template <int DT>
static int launch_select_flags(const void* values, PlanParams pp, const int* l0, int64_t cap, int64_t L,
                               const int32_t* order, uint8_t* member, uint8_t* keep, int64_t* stats, void* ws,
                               FusedTail* fused, hipStream_t st) {
    constexpr int kLevels = Act<DT>::kKeyBits / 8;
    const unsigned G = cdiv(L, kSlice);
    pp.n_slices = (int)G;
    int* lv = ws_levels(ws, L);
    if (!l0) {
        int* rows = ws_l0_rows(ws);
        pp.l0_rows = (int)G;
        hipLaunchKernelGGL(k_hist_level<DT>, dim3(G), dim3(kSliceThreads), 0, st, values, (int)cap, pp, 0, stats,
                           (const int*)nullptr, lv, rows);
        l0 = rows;
    }
    if (fused && G <= (unsigned)kMaxFusedSlices) {
        // radix levels + flags + scan in one launch (grid barriers between the stages)
        hipLaunchKernelGGL(k_plan_fused<DT>, dim3(G), dim3(kSliceThreads), 0, st, values, (int)cap, pp, l0, lv, stats,
                           order, (int)L, member, keep, fused->dst, fused->host_mapped, fused->seq, fused->zero_l0,
                           fused->bar, fused->bar_next);
        fused->done = true;
        return (int)hipGetLastError();
    }
    for (int level = 1; level < kLevels; ++level)
        hipLaunchKernelGGL(k_hist_level<DT>, dim3(G), dim3(kSliceThreads), 0, st, values, (int)cap, pp, level, stats, l0, lv,
                           (int*)nullptr);
    hipLaunchKernelGGL(k_flags<DT>, dim3(G), dim3(kSliceThreads), 0, st, values, (int)cap, pp, l0, (const int*)lv, stats, order,
                       (int)L, member, keep);
    return (int)hipGetLastError();
}

// l0_copies: the table the similarity kernel filled for this call (fused path) or nullptr.
int launch_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double thr, double sub,
                      double ratio_lb, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, const int* l0_copies, int* zero_next, int64_t* host_mapped, int64_t seq,
                      hipStream_t st, long long force_k) {
    PlanParams pp;
    pp.mode = 0; pp.lo = 0; pp.hi = -1; pp.k_given = force_k; pp.sub = sub; pp.ratio_lb = ratio_lb;
    pp.l0_rows = kL0Copies; pp.n_slices = 0;
    int rc;
    static int use_fused = -1;
    if (use_fused < 0) { const char* e = getenv("FF_PLAN_FUSED"); use_fused = e ? atoi(e) : 0; }
    FusedTail tail{dst, host_mapped, seq, zero_next, ws_barrier(ws, seq), ws_barrier(ws, seq + 1), false};
    FusedTail* ft = (use_fused && l0_copies) ? &tail : nullptr;     // only with the per-call workspace protocol
    switch (dtype) {
        case FF_F32:
            pp.thr_key = host_thr_key<FF_F32>(thr);
            rc = launch_select_flags<FF_F32>(sim, pp, l0_copies, L, L, order, member, keep, stats, ws, ft, st);
            break;
        case FF_BF16:
            pp.thr_key = host_thr_key<FF_BF16>(thr);
            rc = launch_select_flags<FF_BF16>(sim, pp, l0_copies, L, L, order, member, keep, stats, ws, ft, st);
            break;
        default:
            pp.thr_key = host_thr_key<FF_F16>(thr);
            rc = launch_select_flags<FF_F16>(sim, pp, l0_copies, L, L, order, member, keep, stats, ws, ft, st);
    }
    if (rc) return rc;
    if (tail.done) return FF_OK;
    return launch_scan(keep, L, dst, stats, host_mapped, seq, zero_next, st);
}

int launch_plan_prune(const void* imp, int dtype, int64_t S, int64_t start, int64_t n_img, int64_t k,
                      uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws,
                      int64_t* host_mapped, int64_t seq, hipStream_t st) {
    PlanParams pp;
    pp.mode = 1; pp.lo = (int)start; pp.hi = (int)(start + n_img); pp.k_given = k; pp.sub = 0; pp.ratio_lb = 0;
    pp.thr_key = 0; pp.l0_rows = 0; pp.n_slices = 0;
    int rc;
    switch (dtype) {
        case FF_F32: rc = launch_select_flags<FF_F32>(imp, pp, nullptr, S, S, nullptr, member, keep, stats, ws, nullptr, st); break;
        case FF_BF16: rc = launch_select_flags<FF_BF16>(imp, pp, nullptr, S, S, nullptr, member, keep, stats, ws, nullptr, st); break;
        default: rc = launch_select_flags<FF_F16>(imp, pp, nullptr, S, S, nullptr, member, keep, stats, ws, nullptr, st);
    }
    if (rc) return rc;
    return launch_scan(keep, S, dst, stats, host_mapped, seq, nullptr, st);
}

int launch_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                           uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, hipStream_t st) {
    hipError_t e = hipMemsetAsync(member, 0, (size_t)L, st);
    if (e != hipSuccess) return (int)e;
    if (n_merge > 0)
        hipLaunchKernelGGL(k_mark_index, dim3(cdiv(n_merge, 256)), dim3(256), 0, st, merge_index, (int)n_merge, stats, member);
    hipLaunchKernelGGL(k_keep_from_member, dim3(cdiv(L, 256)), dim3(256), 0, st, member, order, (int)L, keep);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_scan(keep, L, dst, stats, nullptr, 0, nullptr, st);
}

}  // namespace ff

static int check_plan_args(const void* a, const void* b, const void* c, const void* d, const void* e,
                           int64_t L, void* ws, size_t ws_bytes) {
    if (!a || !b || !c || !d || !e || !ws || L < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31) - 65536) return FF_ERR_UNSUPPORTED;
    if (ws_bytes < ff::plan_ws_bytes(L)) return FF_ERR_WORKSPACE;
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
    if (!aligned16(sim) || !aligned16(dst) || !aligned16(keep) || !aligned16(ws)) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    return ff::launch_plan_merge(sim, dtype, order, L, threshold, sub, ratio_lb, member, dst, keep, stats, ws,
                                 nullptr, nullptr, nullptr, 0, (hipStream_t)stream, -1);
}

extern "C" int ff_plan_topk(const void* sim, int dtype, const int32_t* order, int64_t L, int64_t k, uint8_t* member,
                            int32_t* dst, uint8_t* keep, int64_t* stats, void* ws, size_t ws_bytes,
                            ff_stream_t stream) {
    int rc = check_plan_args(sim, member, dst, keep, stats, L, ws, ws_bytes);
    if (rc) return rc;
    if (!order || k < 0) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (!aligned16(sim) || !aligned16(dst) || !aligned16(keep) || !aligned16(ws)) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    return ff::launch_plan_merge(sim, dtype, order, L, 0.0, 0.0, 0.0, member, dst, keep, stats, ws, nullptr, nullptr,
                                 nullptr, 0, (hipStream_t)stream, k);
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

extern "C" int ff_plan_prune(const void* importance, int dtype, int64_t S, int64_t start, int64_t n_img,
                             int64_t k, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                             void* ws, size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(importance, member, dst, keep, stats, S, ws, ws_bytes);
    if (rc) return rc;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (start < 0 || n_img < 0 || start + n_img > S || k < 0) return FF_ERR_ARG;
    if (!aligned16(importance) || !aligned16(dst) || !aligned16(keep) || !aligned16(ws)) return FF_ERR_ALIGN;
    if (S == 0) return FF_OK;
    return ff::launch_plan_prune(importance, dtype, S, start, n_img, k, member, dst, keep, stats, ws, nullptr, 0,
                                 (hipStream_t)stream);
}
