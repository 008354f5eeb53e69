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
// Three launches, split by what is inherently global:
//   k_select  ONE 16-wave workgroup: count(sim >= thr), the budget decision, and for the top-k
//             branch the k-th key (8-bit radix passes over LDS histograms) plus the index cutoff
//             among entries equal to it.  Output: a 48-byte Select record.
//   k_flags   many workgroups: member[t] for every by-patch slot from (similarity, Select) and
//             the scatter keep[order[t]] = !member[t] - one CU would take a cycle per scattered
//             byte, 256 CUs do not notice it.
//   k_scan    many workgroups, no communication: workgroup g recounts keep[0, 4096 g) itself
//             (<= L bytes, L2-resident) and scans its own 4096 positions into dst[]; the last
//             workgroup knows L_out and publishes the result block.
// Run lengths are not materialised: the merge kernel derives them from 64 member flags at a time.
#include "ff_common.h"

namespace ff {

constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / kWave;
constexpr int kEpt = 16;                          // elements per thread per round
constexpr int kRound = kSelThreads * kEpt;        // 16384
constexpr int kCopies = 2;                        // histogram copies per wave (lane & 3): fewer same-word hits

// Selection rule for entry j of values[lo, hi):
//   topk : key > kth || (key == kth && j <= tie_cut)       (k > 0)
//   else : key >= thr_key && key != NaN
struct Select {
    int topk;
    uint32_t thr_key, kth;
    int tie_cut;
    long long k;
    int lo, hi;       // range the rule applies to
    int invert;       // prune plan: member (= dropped) iff inside [lo, hi) and NOT selected
    int pad;
};

struct SelLds {
    int hist[kSelWaves][kCopies][256];
    int tot[256];
    int scratch[kSelWaves + 1];
    int bcast[4];
};

// 16 consecutive T values starting at j0 (j0 % 16 == 0) as order-preserving keys; entries at or
// beyond `n` are flagged invalid.
template <int DT>
__device__ inline void load_keys(const void* __restrict__ v, int j0, int n, uint32_t* key, uint32_t& valid_mask) {
    using A = Act<DT>;
    valid_mask = 0;
    if (j0 >= n) {
#pragma unroll
        for (int e = 0; e < kEpt; ++e) key[e] = 0;
        return;
    }
    if (j0 + kEpt <= n) {
        valid_mask = 0xffffu;
        if constexpr (A::kBytes == 2) {
            const uint4* p = (const uint4*)((const uint16_t*)v + j0);
            const uint4 a = p[0], b = p[1];
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                key[2 * q] = order_key<DT>(w[q] & 0xffffu);
                key[2 * q + 1] = order_key<DT>(w[q] >> 16);
            }
        } else {
            const uint4* p = (const uint4*)((const uint32_t*)v + j0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 a = p[q];
                key[4 * q] = order_key<DT>(a.x); key[4 * q + 1] = order_key<DT>(a.y);
                key[4 * q + 2] = order_key<DT>(a.z); key[4 * q + 3] = order_key<DT>(a.w);
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < kEpt; ++e) {
        const bool ok = j0 + e < n;
        key[e] = ok ? order_key<DT>(A::bits1(v, j0 + e)) : 0;
        valid_mask |= ok ? (1u << e) : 0u;
    }
}

// key of a T-valued float (the threshold): entries with key >= this and not NaN satisfy sim >= thr
template <int DT>
__device__ inline uint32_t key_of_value(float x) {
    if constexpr (DT == FF_F32) return order_key<DT>(__float_as_uint(x));
    else if constexpr (DT == FF_BF16) return order_key<DT>(__float_as_uint(x) >> 16);
    else { _Float16 h = (_Float16)x; return order_key<DT>((uint32_t)__builtin_bit_cast(uint16_t, h)); }
}

template <int DT> __device__ inline uint32_t nan_key() { return Act<DT>::kKeyBits == 32 ? 0xffffffffu : 0xffffu; }

template <int DT>
__device__ inline bool is_selected(const Select& sel, uint32_t key, int j) {
    if (sel.topk) return sel.k > 0 && (key > sel.kth || (key == sel.kth && j <= sel.tie_cut));
    return key >= sel.thr_key && key != nan_key<DT>();
}

__device__ inline void zero_hist(SelLds& s) {
    for (int x = threadIdx.x; x < kSelWaves * kCopies * 256; x += kSelThreads) (&s.hist[0][0][0])[x] = 0;
}

// After a histogram pass: fold the per-wave copies and pick, from the top, the bin in which the
// running count reaches `remaining`; returns the bin, `above` = entries in higher bins.
__device__ inline int pick_bin(SelLds& s, int remaining, int& above) {
    const int tid = threadIdx.x, lane = lane_id();
    __syncthreads();
    if (tid < 256) {
        int t = 0;
#pragma unroll
        for (int q = 0; q < kSelWaves; ++q)
#pragma unroll
            for (int c = 0; c < kCopies; ++c) t += s.hist[q][c][tid];
        s.tot[tid] = t;
    }
    __syncthreads();
    if (wave_id() == 0) {
        const int top = 255 - 4 * lane;   // lane covers bins top .. top-3
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

// Top-k over values[lo, hi): 8-bit radix passes, then the index of the last selected entry among
// those equal to the k-th value.  `top_hist_ready`: s.hist already holds the top-byte histogram.
template <int DT>
__device__ inline void select_topk(const void* __restrict__ values, int lo, int hi, int k, bool top_hist_ready,
                                   SelLds& s, Select& sel, int& ties_taken) {
    using A = Act<DT>;
    const int tid = threadIdx.x, w = wave_id(), cp = lane_id() & (kCopies - 1);
    const int lo_al = lo & ~(kEpt - 1);
    uint32_t prefix = 0;
    int remaining = k;
    for (int shift = A::kKeyBits - 8; shift >= 0; shift -= 8) {
        const int hi_bits = shift + 8;
        if (!(top_hist_ready && hi_bits == A::kKeyBits)) {
            zero_hist(s);
            __syncthreads();
            for (int base = lo_al; base < hi; base += kRound) {
                const int j0 = base + tid * kEpt;
                uint32_t key[kEpt], valid;
                load_keys<DT>(values, j0, hi, key, valid);
#pragma unroll
                for (int e = 0; e < kEpt; ++e) {
                    const bool in = ((valid >> e) & 1u) && (j0 + e >= lo);
                    const bool match = hi_bits >= A::kKeyBits || (key[e] >> hi_bits) == prefix;
                    if (in && match) atomicAdd(&s.hist[w][cp][(key[e] >> shift) & 255u], 1);
                }
            }
        }
        int above;
        const int bin = pick_bin(s, remaining, above);
        prefix = (prefix << 8) | (uint32_t)bin;
        remaining -= above;
    }
    sel.kth = prefix;
    ties_taken = remaining;                       // >= 1 entries equal to kth belong to the top k
    // the index of the `remaining`-th entry equal to kth, in ascending index order
    int seen = 0;
    if (tid == 0) s.bcast[2] = -1;
    __syncthreads();
    for (int base = lo_al; base < hi; base += kRound) {
        const int j0 = base + tid * kEpt;
        uint32_t key[kEpt], valid;
        load_keys<DT>(values, j0, hi, key, valid);
        int mine = 0;
#pragma unroll
        for (int e = 0; e < kEpt; ++e)
            mine += (((valid >> e) & 1u) && (j0 + e >= lo) && key[e] == prefix) ? 1 : 0;
        int round_total;
        const int before = seen + block_excl_scan<kSelWaves>(mine, s.scratch, round_total);
        if (before < remaining && before + mine >= remaining) {
            int c = before;
#pragma unroll
            for (int e = 0; e < kEpt; ++e) {
                if (((valid >> e) & 1u) && (j0 + e >= lo) && key[e] == prefix) {
                    ++c;
                    if (c == remaining) s.bcast[2] = j0 + e;
                }
            }
        }
        seen += round_total;
        if (seen >= remaining) break;             // uniform
    }
    __syncthreads();
    sel.tie_cut = s.bcast[2];
    __syncthreads();
}

// ---- k_select (merge): main.py:112-127 -------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(kSelThreads) void k_select_merge(
    const void* __restrict__ sim, double thr, double sub, double ratio_lb, int64_t* __restrict__ stats,
    Select* __restrict__ sel_out) {
    using A = Act<DT>;
    __shared__ SelLds s;
    const int tid = threadIdx.x, w = wave_id(), cp = lane_id() & (kCopies - 1);
    const uint64_t clk0 = __builtin_amdgcn_s_memtime();
    const int nv = (int)stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    Select sel;
    sel.topk = 0;
    // thr is already T-valued; +-0 compare equal as floats, so a zero threshold admits both
    sel.thr_key = key_of_value<DT>(thr == 0.0 ? -0.0f : (float)thr);
    sel.kth = 0; sel.tie_cut = -1; sel.k = 0; sel.lo = 0; sel.hi = nv; sel.invert = 0; sel.pad = 0;

    // count(sim >= T(thr)) (main.py:113; NaN compares false, -2 never passes) and the top-byte
    // histogram the top-k branch would need
    zero_hist(s);
    __syncthreads();
    int c = 0;
    for (int base = 0; base < nv; base += kRound) {
        const int j0 = base + tid * kEpt;
        uint32_t key[kEpt], valid;
        load_keys<DT>(sim, j0, nv, key, valid);
        // similarities of neighbouring pairs share their top byte (sign + exponent): fold equal
        // neighbours before touching LDS, and let one lane add for the wave's common bin -
        // otherwise every lane hammers the same histogram word (64-way serialised atomics).
        int run_bin = -1, run_cnt = 0;
#pragma unroll
        for (int e = 0; e < kEpt; ++e) {
            if (!((valid >> e) & 1u)) continue;
            c += (key[e] >= sel.thr_key && key[e] != nan_key<DT>()) ? 1 : 0;
            const int bin = (int)(key[e] >> (A::kKeyBits - 8));
            if (bin == run_bin) { ++run_cnt; continue; }
            if (run_cnt) atomicAdd(&s.hist[w][cp][run_bin], run_cnt);
            run_bin = bin; run_cnt = 1;
        }
        const int lead_bin = uniform(run_bin);
        const bool with_lead = run_cnt > 0 && run_bin == lead_bin;
        const int lead_total = wave_sum_i(with_lead ? run_cnt : 0);
        if (lane_id() == 0 && lead_total) atomicAdd(&s.hist[w][0][lead_bin], lead_total);
        if (run_cnt > 0 && !with_lead) atomicAdd(&s.hist[w][cp][run_bin], run_cnt);
    }
    const int count = block_sum_i<kSelWaves>(c, s.scratch);
    const uint64_t clk1 = __builtin_amdgcn_s_memtime();

    // main.py:114-116 in double, as python: ratio = count / ftn ; ratio < sub ?
    const double ratio = ftn > 0 ? (double)count / (double)ftn : 0.0;
    sel.topk = !(ratio < sub);
    int ties_taken = 0;
    if (sel.topk) {
        long long k = (long long)(sub * (double)ftn);      // int(sub * ftn), main.py:122
        if (k > nv) k = nv;
        if (k < 0) k = 0;
        sel.k = k;
        if (k > 0) select_topk<DT>(sim, 0, nv, (int)k, true, s, sel, ties_taken);
    }
    if (tid == 0) {
        *sel_out = sel;
        stats[FF_STAT_COUNT] = count;
        stats[FF_STAT_BRANCH] = sel.topk ? 1 : 0;
        stats[FF_STAT_K] = sel.k;
        stats[FF_STAT_BELOW_LB] = (!sel.topk && ratio < ratio_lb) ? 1 : 0;
        stats[FF_STAT_KTH_KEY] = sel.kth;
        stats[FF_STAT_TIES_TAKEN] = ties_taken;
        stats[FF_STAT_T_PLAN + 0] = (int64_t)(clk1 - clk0);
        stats[FF_STAT_T_PLAN + 1] = (int64_t)(__builtin_amdgcn_s_memtime() - clk1);
    }
}

// ---- k_select (prune): top-k of importance[start, start + n_img) (main.py:74-79) ----------------------
template <int DT>
__global__ __launch_bounds__(kSelThreads) void k_select_prune(
    const void* __restrict__ importance, int S, int start, int n_img, int k, int64_t* __restrict__ stats,
    Select* __restrict__ sel_out) {
    __shared__ SelLds s;
    Select sel;
    sel.topk = 1; sel.thr_key = 0; sel.kth = 0; sel.tie_cut = -1; sel.k = k;
    sel.lo = start; sel.hi = start + n_img; sel.invert = 1; sel.pad = 0;
    int ties_taken = 0;
    if (k >= n_img) { sel.kth = 0; sel.tie_cut = 0x7fffffff; sel.k = n_img > 0 ? n_img : 1; }   // everything selected
    else if (k > 0) select_topk<DT>(importance, start, start + n_img, k, false, s, sel, ties_taken);
    if (threadIdx.x == 0) {
        *sel_out = sel;
        stats[FF_STAT_NV] = S;
        stats[FF_STAT_K] = k;
        stats[FF_STAT_KTH_KEY] = sel.kth;
        stats[FF_STAT_TIES_TAKEN] = ties_taken;
    }
}

// ---- k_flags ----------------------------------------------------------------------------------------
// One thread per by-patch slot t (then the non-visual tail of `order`).  Slot 0 never folds: it has
// no predecessor (the reference would wrap to order[-1], main.py:290; reachable only when top-k
// exceeds the number of valid pairs).
template <int DT>
__global__ __launch_bounds__(256) void k_flags(const void* __restrict__ values, const Select* __restrict__ selp,
                                               const int64_t* __restrict__ stats, const int32_t* __restrict__ order,
                                               int L, uint8_t* __restrict__ member, uint8_t* __restrict__ keep) {
    using A = Act<DT>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const int n_flag = (int)stats[FF_STAT_NV];
    const Select sel = *selp;
    uint8_t m = 0;
    if (t < n_flag) {
        const uint32_t key = order_key<DT>(A::bits1(values, t));
        if (sel.invert) m = (t >= sel.lo && t < sel.hi) && !is_selected<DT>(sel, key, t);
        else m = is_selected<DT>(sel, key, t) && t > 0;
    }
    member[t] = m;
    keep[order ? order[t] : t] = m ? 0 : 1;
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

__device__ inline void publish(const int64_t* __restrict__ stats, int64_t* host_mapped, int64_t seq) {
    // Optional copy of the result block into device-visible pinned host memory, sequence word last.
    if (!host_mapped) return;
    for (int q = 0; q < FF_STAT_WORDS; ++q)
        if (q != FF_STAT_SEQ) __hip_atomic_store(&host_mapped[q], stats[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host_mapped[FF_STAT_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kScanThreads) void k_scan(const uint8_t* __restrict__ keep, int L,
                                                       int32_t* __restrict__ dst, int64_t* __restrict__ stats,
                                                       int64_t* host_mapped, int64_t seq) {
    __shared__ int scratch[kScanThreads / kWave + 1];
    const int tid = threadIdx.x;
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
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const int l_out = before + span_total;
        stats[FF_STAT_LOUT] = l_out;
        stats[FF_STAT_MERGED] = L - l_out;
        publish(stats, host_mapped, seq);
    }
}

// ---- launchers (also used by the fused step in ff_abi.hip) ------------------------------------------
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

static int launch_scan(const uint8_t* keep, int64_t L, int32_t* dst, int64_t* stats, int64_t* host_mapped,
                       int64_t seq, hipStream_t st) {
    hipLaunchKernelGGL(k_scan, dim3(cdiv(L, kScanSpan)), dim3(kScanThreads), 0, st, keep, (int)L, dst, stats,
                       host_mapped, seq);
    return (int)hipGetLastError();
}

int launch_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double thr, double sub,
                      double ratio_lb, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, int64_t* host_mapped, int64_t seq, hipStream_t st) {
    Select* sel = (Select*)ws;
    const dim3 fg(cdiv(L, 256));
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_select_merge<FF_F32>, dim3(1), dim3(kSelThreads), 0, st, sim, thr, sub, ratio_lb, stats, sel);
            hipLaunchKernelGGL(k_flags<FF_F32>, fg, dim3(256), 0, st, sim, sel, stats, order, (int)L, member, keep);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_select_merge<FF_BF16>, dim3(1), dim3(kSelThreads), 0, st, sim, thr, sub, ratio_lb, stats, sel);
            hipLaunchKernelGGL(k_flags<FF_BF16>, fg, dim3(256), 0, st, sim, sel, stats, order, (int)L, member, keep);
            break;
        default:
            hipLaunchKernelGGL(k_select_merge<FF_F16>, dim3(1), dim3(kSelThreads), 0, st, sim, thr, sub, ratio_lb, stats, sel);
            hipLaunchKernelGGL(k_flags<FF_F16>, fg, dim3(256), 0, st, sim, sel, stats, order, (int)L, member, keep);
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_scan(keep, L, dst, stats, host_mapped, seq, st);
}

int launch_plan_prune(const void* imp, int dtype, int64_t S, int64_t start, int64_t n_img, int64_t k,
                      uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws,
                      int64_t* host_mapped, int64_t seq, hipStream_t st) {
    Select* sel = (Select*)ws;
    const dim3 fg(cdiv(S, 256));
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_select_prune<FF_F32>, dim3(1), dim3(kSelThreads), 0, st, imp, (int)S, (int)start, (int)n_img, (int)k, stats, sel);
            hipLaunchKernelGGL(k_flags<FF_F32>, fg, dim3(256), 0, st, imp, sel, stats, (const int32_t*)nullptr, (int)S, member, keep);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_select_prune<FF_BF16>, dim3(1), dim3(kSelThreads), 0, st, imp, (int)S, (int)start, (int)n_img, (int)k, stats, sel);
            hipLaunchKernelGGL(k_flags<FF_BF16>, fg, dim3(256), 0, st, imp, sel, stats, (const int32_t*)nullptr, (int)S, member, keep);
            break;
        default:
            hipLaunchKernelGGL(k_select_prune<FF_F16>, dim3(1), dim3(kSelThreads), 0, st, imp, (int)S, (int)start, (int)n_img, (int)k, stats, sel);
            hipLaunchKernelGGL(k_flags<FF_F16>, fg, dim3(256), 0, st, imp, sel, stats, (const int32_t*)nullptr, (int)S, member, keep);
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_scan(keep, S, dst, stats, host_mapped, seq, st);
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
    return launch_scan(keep, L, dst, stats, nullptr, 0, st);
}

}  // namespace ff

static int check_plan_args(const void* a, const void* b, const void* c, const void* d, const void* e,
                           int64_t L, void* ws, size_t ws_bytes) {
    if (!a || !b || !c || !d || !e || !ws || L < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31) - ff::kRound) return FF_ERR_UNSUPPORTED;
    if (ws_bytes < 256) return FF_ERR_WORKSPACE;
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
                                 nullptr, 0, (hipStream_t)stream);
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
