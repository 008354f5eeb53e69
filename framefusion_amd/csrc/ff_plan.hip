// K2+K3 - the integer "plan" between the two streaming passes: threshold count, budget decision,
// top-k (radix select with a lowest-index tie rule), run detection and the compaction scan.
// All of it works on <= L 2-byte similarities and L int32 indices (a few hundred KB, L2-resident),
// so it is ONE 16-wave workgroup with LDS histograms/scans and no host round trip: the branch the
// reference takes on the host after two .item() syncs (framefusion/main.py:112-127) is decided
// on the device, in double, exactly as python evaluates it.
//
// Replaces: main.py:112-127 (select), find_contigious_latter_index (main.py:351-380), the
// unique/where/repeat_interleave index algebra of merge_tokens_and_get_mask (main.py:269-301)
// and the keep-mask construction (main.py:278-279) - 17 host syncs in the reference.
//
// Data movement: every pass walks its array in rounds of 16384 elements, thread t owning the 16
// consecutive elements [round*16384 + 16t, +16): 16-byte vector loads/stores, a wave touching one
// contiguous 2-4 KiB span per instruction pair, and one block scan (two barriers) per round.
//   pass A  keys  -> count(sim >= thr) + histogram of the key's top byte      (always)
//   pass B  keys  -> histogram of the next byte(s) among matching prefixes     (top-k branch)
//   pass C  keys  -> index cutoff among the entries equal to the k-th value    (top-k branch)
//   pass D  keys  -> member flags, run lengths (reverse walk), keep[] scatter, new by-patch rank
//   pass E  keep  -> dst[] (exclusive scan in sequence order), L_out
#include "ff_common.h"

namespace ff {

constexpr int kPlanThreads = 1024;
constexpr int kPlanWaves = kPlanThreads / kWave;
constexpr int kEpt = 16;                          // elements per thread per round
constexpr int kRound = kPlanThreads * kEpt;       // 16384
constexpr int kInf = 0x7fffffff;

struct PlanLds {
    int hist[kPlanWaves][256];
    int tot[256];
    int scratch[kPlanWaves + 1];
    int wmin[kPlanWaves];
    int bcast[4];
};

// 16 consecutive T values starting at j0 (j0 % 16 == 0) as order-preserving keys; entries at or
// beyond `n` get valid = false.
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

__device__ inline void zero_hist(PlanLds& s) {
    for (int x = threadIdx.x; x < kPlanWaves * 256; x += kPlanThreads) (&s.hist[0][0])[x] = 0;
}

// After a histogram pass: fold the per-wave histograms and pick, from the top, the bin in which the
// running count reaches `remaining`; returns the bin, `above` = entries in higher bins.
__device__ inline int pick_bin(PlanLds& s, int remaining, int& above) {
    const int tid = threadIdx.x, lane = lane_id();
    __syncthreads();
    if (tid < 256) {
        int t = 0;
#pragma unroll
        for (int q = 0; q < kPlanWaves; ++q) t += s.hist[q][tid];
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

// Selection state shared by the merge and prune plans: entry j in [lo, hi) is selected iff
//   key > kth || (key == kth && j <= tie_cut)          (mode top-k)
//   key >= thr_key && key != NaN                       (mode threshold)
struct Select {
    bool topk;
    uint32_t thr_key, kth;
    int tie_cut;      // largest index of a selected tie (-1: none)
    long long k;
};

// Top-k over values[lo, hi): radix select (passes B...) then the tie cutoff (pass C).
// `top_hist_ready`: s.hist already holds the top-byte histogram (pass A did it).
template <int DT>
__device__ inline void select_topk(const void* __restrict__ values, int lo, int hi, int k, bool top_hist_ready,
                                   PlanLds& s, Select& sel, int& ties_taken) {
    using A = Act<DT>;
    const int tid = threadIdx.x, w = wave_id();
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
                    if (in && match) atomicAdd(&s.hist[w][(key[e] >> shift) & 255u], 1);
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
    // pass C: the index of the `remaining`-th entry equal to kth, in ascending index order
    int seen = 0;                                 // ties in earlier rounds
    int cut = -1;
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
        int before = seen + block_excl_scan<kPlanWaves>(mine, s.scratch, round_total);
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
        if (seen >= remaining) break;             // uniform: seen is identical in every thread
    }
    __syncthreads();
    cut = s.bcast[2];
    __syncthreads();
    sel.tie_cut = cut;
}

template <int DT>
__device__ inline bool is_selected(const Select& sel, uint32_t key, int j) {
    if (sel.topk) return sel.k > 0 && (key > sel.kth || (key == sel.kth && j <= sel.tie_cut));
    return key >= sel.thr_key && key != nan_key<DT>();
}

// Pass D + E.  member(j) for j in [0, n_flag) comes from `values` + `sel` (drop = !selected inside
// [lo, hi) when `invert`, i.e. the prune plan) or from explicit byte flags.
template <int DT>
__device__ inline void plan_tail(const void* __restrict__ values, const Select& sel, bool invert, int lo, int hi,
                                 const unsigned char* __restrict__ flags, int n_flag,
                                 const int32_t* __restrict__ order, int L, bool merge_runs,
                                 int32_t* __restrict__ run_len, int32_t* __restrict__ dst,
                                 uint8_t* __restrict__ keep_out, int64_t* __restrict__ stats, PlanLds& s,
                                 uint8_t* keep_lds) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    // keep[] is written by sequence position = scattered single bytes in by-patch order: stage it
    // in LDS when it fits (one CU's L1 would otherwise take one request per byte) and stream it
    // out during the scan.
    uint8_t* keep = keep_lds ? keep_lds : keep_out;
    const uint64_t clk_d = __builtin_amdgcn_s_memtime();

    // ---- pass D: reverse walk over the by-patch positions -----------------------------------
    int carry = n_flag;                               // index of the next non-member to the right
    const int rounds = (n_flag + kRound - 1) / kRound;
    for (int r = rounds - 1; r >= 0; --r) {
        const int j0 = r * kRound + tid * kEpt;
        uint32_t mem = 0;                             // bit e: element j0+e is a member
        if (j0 < n_flag) {
            if (flags) {
#pragma unroll
                for (int e = 0; e < kEpt; ++e)
                    if (j0 + e < n_flag && flags[j0 + e]) mem |= 1u << e;
            } else {
                uint32_t key[kEpt], valid;
                load_keys<DT>(values, j0, n_flag, key, valid);
#pragma unroll
                for (int e = 0; e < kEpt; ++e) {
                    const int j = j0 + e;
                    if (!((valid >> e) & 1u)) continue;
                    bool m;
                    if (invert) m = (j >= lo && j < hi) && !is_selected<DT>(sel, key[e], j);
                    else m = is_selected<DT>(sel, key[e], j);
                    if (m) mem |= 1u << e;
                }
            }
            // by-patch position 0 has no predecessor to fold into (the reference would wrap to
            // order[-1], main.py:290; reachable only when top-k exceeds the number of valid pairs)
            if (merge_runs && j0 == 0) mem &= ~1u;
        }
        const int n_here = min(max(n_flag - j0, 0), kEpt);
        const uint32_t live = n_here >= kEpt ? 0xffffu : ((1u << n_here) - 1u);
        const uint32_t nonmem = ~mem & live;
        const int first_nm = nonmem ? j0 + __ffs((int)nonmem) - 1 : kInf;
        // suffix-min of first_nm over the threads to my right
        int v = first_nm;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int t = __shfl_down(v, o, kWave);
            if (lane + o < kWave) v = min(v, t);
        }
        int right = __shfl_down(v, 1, kWave);
        if (lane == kWave - 1) right = kInf;
        if (lane == 0) s.wmin[w] = v;
        __syncthreads();
        int round_min = kInf;
#pragma unroll
        for (int q = 0; q < kPlanWaves; ++q) {
            const int x = s.wmin[q];
            if (q > w) right = min(right, x);
            round_min = min(round_min, x);
        }
        __syncthreads();
        int next_nm = min(right, carry);
        if (j0 < n_flag) {
            int rl[kEpt];
#pragma unroll
            for (int e = kEpt - 1; e >= 0; --e) {
                const int j = j0 + e;
                if ((mem >> e) & 1u) rl[e] = -1;
                else { rl[e] = merge_runs ? next_nm - j - 1 : 0; next_nm = j; }
            }
            if (n_here == kEpt) {
                uint4* p = (uint4*)(run_len + j0);
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = make_uint4(rl[4 * q], rl[4 * q + 1], rl[4 * q + 2], rl[4 * q + 3]);
                int32_t ord[kEpt];
                if (order) {
                    const uint4* po = (const uint4*)(order + j0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint4 o4 = po[q];
                        ord[4 * q] = o4.x; ord[4 * q + 1] = o4.y; ord[4 * q + 2] = o4.z; ord[4 * q + 3] = o4.w;
                    }
#pragma unroll
                    for (int e = 0; e < kEpt; ++e) keep[ord[e]] = ((mem >> e) & 1u) ? 0 : 1;
                } else {
                    uint32_t kb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        kb[q] = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) kb[q] |= (((mem >> (4 * q + b)) & 1u) ? 0u : 1u) << (8 * b);
                    }
                    *(uint4*)(keep + j0) = make_uint4(kb[0], kb[1], kb[2], kb[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < kEpt; ++e) {
                    if (e < n_here) {
                        run_len[j0 + e] = rl[e];
                        keep[order ? order[j0 + e] : j0 + e] = ((mem >> e) & 1u) ? 0 : 1;
                    }
                }
            }
        }
        carry = min(carry, round_min);
    }
    // non-visual tail of `order`: plain copies, always kept
    for (int t = n_flag + tid; t < L; t += kPlanThreads) {
        run_len[t] = 0;
        keep[order ? order[t] : t] = 1;
    }
    __syncthreads();

    // ---- pass E: compaction scan over sequence positions ---------------------------------------
    const uint64_t clk_e = __builtin_amdgcn_s_memtime();
    int base_out = 0;
    for (int i0r = 0; i0r < L; i0r += kRound) {
        const int i0 = i0r + tid * kEpt;
        uint32_t kb[4] = {0, 0, 0, 0};
        const int n_here = min(max(L - i0, 0), kEpt);
        if (n_here == kEpt) {
            const uint4 k4 = *(const uint4*)(keep + i0);
            kb[0] = k4.x; kb[1] = k4.y; kb[2] = k4.z; kb[3] = k4.w;
            if (keep_lds) *(uint4*)(keep_out + i0) = k4;
        } else {
            for (int e = 0; e < n_here; ++e) {
                const uint8_t kv = keep[i0 + e];
                kb[e >> 2] |= (uint32_t)kv << (8 * (e & 3));
                if (keep_lds) keep_out[i0 + e] = kv;
            }
        }
        // keep bytes are 0/1: popcount of the words = number kept
        const int mine = __popc(kb[0]) + __popc(kb[1]) + __popc(kb[2]) + __popc(kb[3]);
        int round_total;
        int pos = base_out + block_excl_scan<kPlanWaves>(mine, s.scratch, round_total);
        base_out += round_total;
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
    }
    if (tid == 0) {
        stats[FF_STAT_LOUT] = base_out;
        stats[FF_STAT_MERGED] = L - base_out;
        stats[FF_STAT_T_PLAN + 2] = (int64_t)(clk_e - clk_d);
        stats[FF_STAT_T_PLAN + 3] = (int64_t)(__builtin_amdgcn_s_memtime() - clk_e);
    }
}

__device__ inline void publish(const int64_t* __restrict__ stats, int64_t* host_mapped, int64_t seq) {
    // Optional copy of the result block into device-visible pinned host memory, sequence word last.
    if (!host_mapped) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < FF_STAT_WORDS; ++q)
            if (q != FF_STAT_SEQ) __hip_atomic_store(&host_mapped[q], stats[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_mapped[FF_STAT_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int DT>
__global__ __launch_bounds__(kPlanThreads) void k_plan_merge(
    const void* __restrict__ sim, const int32_t* __restrict__ order, int L, double thr, double sub,
    double ratio_lb, int32_t* __restrict__ run_len, int32_t* __restrict__ dst, uint8_t* __restrict__ keep,
    int64_t* __restrict__ stats, int64_t* host_mapped, int64_t seq, int keep_in_lds) {
    using A = Act<DT>;
    __shared__ PlanLds s;
    extern __shared__ __attribute__((aligned(16))) uint8_t keep_lds[];
    const int tid = threadIdx.x, w = wave_id();
    const uint64_t clk0 = __builtin_amdgcn_s_memtime();
    const int nv = (int)stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    Select sel;
    sel.topk = false;
    // thr is already T-valued; +-0 compare equal as floats, so a zero threshold admits both
    sel.thr_key = key_of_value<DT>(thr == 0.0 ? -0.0f : (float)thr);
    sel.kth = 0; sel.tie_cut = -1; sel.k = 0;

    // ---- pass A: count(sim >= T(thr)) (main.py:113; NaN compares false, -2 never passes) and the
    //      top-byte histogram the top-k branch would need
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
            if (run_cnt) atomicAdd(&s.hist[w][run_bin], run_cnt);
            run_bin = bin; run_cnt = 1;
        }
        const int lead_bin = uniform(run_bin);
        const bool with_lead = run_cnt > 0 && run_bin == lead_bin;
        const int lead_total = wave_sum_i(with_lead ? run_cnt : 0);
        if (lane_id() == 0 && lead_total) atomicAdd(&s.hist[w][lead_bin], lead_total);
        if (run_cnt > 0 && !with_lead) atomicAdd(&s.hist[w][run_bin], run_cnt);
    }
    const int count = block_sum_i<kPlanWaves>(c, s.scratch);
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
        stats[FF_STAT_T_PLAN + 0] = (int64_t)(clk1 - clk0);
        stats[FF_STAT_T_PLAN + 1] = (int64_t)(__builtin_amdgcn_s_memtime() - clk1);
        stats[FF_STAT_COUNT] = count;
        stats[FF_STAT_BRANCH] = sel.topk ? 1 : 0;
        stats[FF_STAT_K] = sel.k;
        stats[FF_STAT_BELOW_LB] = (!sel.topk && ratio < ratio_lb) ? 1 : 0;
        stats[FF_STAT_KTH_KEY] = sel.kth;
        stats[FF_STAT_TIES_TAKEN] = ties_taken;
    }
    plan_tail<DT>(sim, sel, false, 0, nv, nullptr, nv, order, L, true, run_len, dst, keep, stats, s,
                  keep_in_lds ? keep_lds : nullptr);
    publish(stats, host_mapped, seq);
}

__global__ __launch_bounds__(kPlanThreads) void k_plan_from_index(
    const int64_t* __restrict__ merge_index, int n_merge, const int32_t* __restrict__ order, int L,
    int32_t* __restrict__ run_len, int32_t* __restrict__ dst, uint8_t* __restrict__ keep,
    int64_t* __restrict__ stats, unsigned char* __restrict__ flags, int keep_in_lds) {
    __shared__ PlanLds s;
    extern __shared__ __attribute__((aligned(16))) uint8_t keep_lds[];
    const int nv = (int)stats[FF_STAT_NV];
    for (int j = threadIdx.x; j < nv; j += kPlanThreads) flags[j] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < n_merge; q += kPlanThreads) {
        const int64_t j = merge_index[q];
        if (j >= 0 && j < nv) flags[j] = 1;
    }
    __syncthreads();
    Select sel{};
    plan_tail<FF_BF16>(nullptr, sel, false, 0, nv, flags, nv, order, L, true, run_len, dst, keep, stats, s,
                       keep_in_lds ? keep_lds : nullptr);
}

template <int DT>
__global__ __launch_bounds__(kPlanThreads) void k_plan_prune(
    const void* __restrict__ importance, int S, int start, int n_img, int k,
    int32_t* __restrict__ run_len, int32_t* __restrict__ dst, uint8_t* __restrict__ keep,
    int64_t* __restrict__ stats, int64_t* host_mapped, int64_t seq, int keep_in_lds) {
    __shared__ PlanLds s;
    extern __shared__ __attribute__((aligned(16))) uint8_t keep_lds[];
    Select sel;
    sel.topk = true;
    sel.thr_key = 0; sel.kth = 0; sel.tie_cut = -1;
    sel.k = k;
    int ties_taken = 0;
    if (k >= n_img) { sel.kth = 0; sel.tie_cut = kInf; sel.k = n_img > 0 ? n_img : 1; }   // everything selected
    else if (k > 0) select_topk<DT>(importance, start, start + n_img, k, false, s, sel, ties_taken);
    if (threadIdx.x == 0) {
        stats[FF_STAT_NV] = S;
        stats[FF_STAT_K] = k;
        stats[FF_STAT_KTH_KEY] = sel.kth;
        stats[FF_STAT_TIES_TAKEN] = ties_taken;
    }
    plan_tail<DT>(importance, sel, true, start, start + n_img, nullptr, S, nullptr, S, false, run_len, dst, keep,
                  stats, s, keep_in_lds ? keep_lds : nullptr);
    publish(stats, host_mapped, seq);
}

// ---- launchers (also used by the fused step in ff_abi.hip) ------------------------------------------
constexpr size_t kKeepLdsMax = 128 * 1024;

// Returns 1 (+ dynamic LDS bytes) when keep[] is staged in LDS, 0 when it stays in global memory,
// -hipError on failure.  The opt-in for > 64 KiB of dynamic LDS is set once per kernel.
template <typename K>
static int allow_big_lds(K kernel) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kKeepLdsMax);
    return e == hipSuccess ? 0 : -(int)e;
}
static int keep_lds_plan(int64_t L, size_t& lds) {
    static int ready = 0;
    if (!ready) {
        int rc = 0;
        rc = rc ? rc : allow_big_lds(k_plan_merge<FF_F32>);
        rc = rc ? rc : allow_big_lds(k_plan_merge<FF_BF16>);
        rc = rc ? rc : allow_big_lds(k_plan_merge<FF_F16>);
        rc = rc ? rc : allow_big_lds(k_plan_prune<FF_F32>);
        rc = rc ? rc : allow_big_lds(k_plan_prune<FF_BF16>);
        rc = rc ? rc : allow_big_lds(k_plan_prune<FF_F16>);
        rc = rc ? rc : allow_big_lds(k_plan_from_index);
        if (rc) return rc;
        ready = 1;
    }
    const size_t need = ((size_t)L + 15) & ~(size_t)15;
    if (need <= kKeepLdsMax) { lds = need; return 1; }
    lds = 0;
    return 0;
}

int launch_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double thr, double sub,
                      double ratio_lb, int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, int64_t* host_mapped, int64_t seq, hipStream_t st) {
    (void)ws;
    size_t lds;
    const int in_lds = keep_lds_plan(L, lds);
    if (in_lds < 0) return -in_lds;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_plan_merge<FF_F32>, dim3(1), dim3(kPlanThreads), lds, st, sim, order, (int)L, thr,
                               sub, ratio_lb, run_len, dst, keep, stats, host_mapped, seq, in_lds);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_plan_merge<FF_BF16>, dim3(1), dim3(kPlanThreads), lds, st, sim, order, (int)L, thr,
                               sub, ratio_lb, run_len, dst, keep, stats, host_mapped, seq, in_lds);
            break;
        default:
            hipLaunchKernelGGL(k_plan_merge<FF_F16>, dim3(1), dim3(kPlanThreads), lds, st, sim, order, (int)L, thr,
                               sub, ratio_lb, run_len, dst, keep, stats, host_mapped, seq, in_lds);
    }
    return (int)hipGetLastError();
}

int launch_plan_prune(const void* imp, int dtype, int64_t S, int64_t start, int64_t n_img, int64_t k,
                      int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws,
                      int64_t* host_mapped, int64_t seq, hipStream_t st) {
    (void)ws;
    size_t lds;
    const int in_lds = keep_lds_plan(S, lds);
    if (in_lds < 0) return -in_lds;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_plan_prune<FF_F32>, dim3(1), dim3(kPlanThreads), lds, st, imp, (int)S, (int)start,
                               (int)n_img, (int)k, run_len, dst, keep, stats, host_mapped, seq, in_lds);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_plan_prune<FF_BF16>, dim3(1), dim3(kPlanThreads), lds, st, imp, (int)S, (int)start,
                               (int)n_img, (int)k, run_len, dst, keep, stats, host_mapped, seq, in_lds);
            break;
        default:
            hipLaunchKernelGGL(k_plan_prune<FF_F16>, dim3(1), dim3(kPlanThreads), lds, st, imp, (int)S, (int)start,
                               (int)n_img, (int)k, run_len, dst, keep, stats, host_mapped, seq, in_lds);
    }
    return (int)hipGetLastError();
}

int launch_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                           int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws, hipStream_t st) {
    size_t lds;
    const int in_lds = keep_lds_plan(L, lds);
    if (in_lds < 0) return -in_lds;
    hipLaunchKernelGGL(k_plan_from_index, dim3(1), dim3(kPlanThreads), lds, st, merge_index, (int)n_merge, order,
                       (int)L, run_len, dst, keep, stats, (unsigned char*)ws, in_lds);
    return (int)hipGetLastError();
}

}  // namespace ff

static int check_plan_args(const void* a, const void* b, const void* c, const void* d, const void* e,
                           int64_t L, void* ws, size_t ws_bytes) {
    if (!a || !b || !c || !d || !e || !ws || L < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31) - ff::kRound) return FF_ERR_UNSUPPORTED;
    if (ws_bytes < (size_t)L) return FF_ERR_WORKSPACE;
    return FF_OK;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int ff_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double threshold,
                             double sub, double ratio_lb, int32_t* run_len, int32_t* dst, uint8_t* keep,
                             int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(sim, run_len, dst, keep, stats, L, ws, ws_bytes);
    if (rc) return rc;
    if (!order) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (!aligned16(sim) || !aligned16(order) || !aligned16(run_len) || !aligned16(dst) || !aligned16(keep))
        return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    return ff::launch_plan_merge(sim, dtype, order, L, threshold, sub, ratio_lb, run_len, dst, keep, stats, ws,
                                 nullptr, 0, (hipStream_t)stream);
}

extern "C" int ff_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                                  int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws,
                                  size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(order, run_len, dst, keep, stats, L, ws, ws_bytes);
    if (rc) return rc;
    if (n_merge < 0 || (n_merge > 0 && !merge_index)) return FF_ERR_ARG;
    if (!aligned16(order) || !aligned16(run_len) || !aligned16(dst) || !aligned16(keep)) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    return ff::launch_plan_from_index(merge_index, n_merge, order, L, run_len, dst, keep, stats, ws,
                                      (hipStream_t)stream);
}

extern "C" int ff_plan_prune(const void* importance, int dtype, int64_t S, int64_t start, int64_t n_img,
                             int64_t k, int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats,
                             void* ws, size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(importance, run_len, dst, keep, stats, S, ws, ws_bytes);
    if (rc) return rc;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (start < 0 || n_img < 0 || start + n_img > S || k < 0) return FF_ERR_ARG;
    if (!aligned16(importance) || !aligned16(run_len) || !aligned16(dst) || !aligned16(keep)) return FF_ERR_ALIGN;
    if (S == 0) return FF_OK;
    return ff::launch_plan_prune(importance, dtype, S, start, n_img, k, run_len, dst, keep, stats, ws, nullptr, 0,
                                 (hipStream_t)stream);
}
