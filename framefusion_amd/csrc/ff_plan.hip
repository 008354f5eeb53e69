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
#include "ff_common.h"

namespace ff {

constexpr int kPlanThreads = 1024;
constexpr int kPlanWaves = kPlanThreads / kWave;

struct PlanLds {
    int hist[kPlanWaves][256];
    int tot[256];
    int scratch[kPlanWaves + 1];
    int lead[kPlanThreads];      // leading ones of each thread's range
    unsigned char full[kPlanThreads];
    int bcast[4];
};

// Contiguous split of [lo, hi) over the block's threads.
__device__ inline void thread_range(int lo, int hi, int& a, int& b) {
    const int n = hi - lo;
    const int per = (n + kPlanThreads - 1) / kPlanThreads;
    a = min(lo + (int)threadIdx.x * per, hi);
    b = min(a + per, hi);
}

// k-th largest key of values[lo, hi) (k >= 1, k <= hi - lo). Returns the key; `need` = how many
// entries EQUAL to it belong to the top k (the rest of the top k are strictly greater).
template <int DT>
__device__ inline uint32_t radix_select(const void* values, int lo, int hi, int k, PlanLds& s, int& need) {
    using A = Act<DT>;
    int a, b;
    thread_range(lo, hi, a, b);
    const int w = wave_id(), lane = lane_id(), tid = threadIdx.x;
    uint32_t prefix = 0;
    int remaining = k;
    for (int shift = A::kKeyBits - 8; shift >= 0; shift -= 8) {
        for (int x = tid; x < kPlanWaves * 256; x += kPlanThreads) (&s.hist[0][0])[x] = 0;
        __syncthreads();
        const int hi_bits = shift + 8;
        for (int j = a; j < b; ++j) {
            const uint32_t key = order_key<DT>(A::bits1(values, j));
            const bool match = (hi_bits >= A::kKeyBits) || ((key >> hi_bits) == prefix);
            if (match) atomicAdd(&s.hist[w][(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid < 256) {
            int t = 0;
#pragma unroll
            for (int q = 0; q < kPlanWaves; ++q) t += s.hist[q][tid];
            s.tot[tid] = t;
        }
        __syncthreads();
        if (w == 0) {
            // lane l covers bins 255-4l .. 252-4l (descending); find the bin where the running
            // count from the top reaches `remaining`.
            const int top = 255 - 4 * lane;
            const int v0 = s.tot[top], v1 = s.tot[top - 1], v2 = s.tot[top - 2], v3 = s.tot[top - 3];
            const int sum = v0 + v1 + v2 + v3;
            const int incl = wave_incl_scan(sum);
            const unsigned long long hit = __ballot(incl >= remaining);
            const int first = __ffsll((long long)hit) - 1;
            if (lane == first) {
                int above = incl - sum, bin = top;
                if (above + v0 >= remaining) { bin = top; }
                else if (above + v0 + v1 >= remaining) { above += v0; bin = top - 1; }
                else if (above + v0 + v1 + v2 >= remaining) { above += v0 + v1; bin = top - 2; }
                else { above += v0 + v1 + v2; bin = top - 3; }
                s.bcast[0] = bin;
                s.bcast[1] = above;
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint32_t)s.bcast[0];
        remaining -= s.bcast[1];
        __syncthreads();
    }
    need = remaining;
    return prefix;
}

// From member flags m[0, n_flag) (bytes in global scratch, each thread has written exactly its own
// thread_range slice) produce run_len / keep / dst / stats.  merge_runs: a member folds into the
// nearest preceding non-member (merge); otherwise members are simply dropped (prune).
__device__ inline void plan_tail(unsigned char* __restrict__ m, int n_flag, const int32_t* __restrict__ order,
                                 int L, bool merge_runs, int32_t* __restrict__ run_len,
                                 int32_t* __restrict__ dst, uint8_t* __restrict__ keep,
                                 int64_t* __restrict__ stats, PlanLds& s) {
    const int tid = threadIdx.x;
    int a, b;
    thread_range(0, n_flag, a, b);
    // A token at by-patch position 0 has no predecessor to fold into (the reference would wrap to
    // order[-1], main.py:290; only reachable when top-k exceeds the number of valid pairs).
    if (merge_runs && a == 0 && b > 0) m[0] = 0;

    // leading ones of my range
    int lead = 0;
    while (a + lead < b && m[a + lead]) ++lead;
    s.lead[tid] = lead;
    s.full[tid] = (a < b && lead == b - a) || (a >= b && a < n_flag);
    __syncthreads();
    // ones that follow my range, across as many all-ones ranges as needed
    int carry = 0;
    if (merge_runs && a < b) {
        int t = tid + 1;
        while (t < kPlanThreads) {
            carry += s.lead[t];
            if (!s.full[t]) break;
            ++t;
        }
    }
    for (int j = b - 1; j >= a; --j) {
        if (m[j]) { run_len[j] = -1; ++carry; }
        else { run_len[j] = merge_runs ? carry : 0; carry = 0; }
    }
    // non-visual tail of `order`: plain copies
    for (int t = n_flag + tid; t < L; t += kPlanThreads) run_len[t] = 0;

    // keep mask by sequence position
    for (int i = tid; i < L; i += kPlanThreads) keep[i] = 1;
    __syncthreads();
    for (int j = a; j < b; ++j)
        if (m[j]) keep[order ? order[j] : j] = 0;
    __syncthreads();

    // compaction scan over sequence positions
    int sa, sb;
    thread_range(0, L, sa, sb);
    int mine = 0;
    for (int i = sa; i < sb; ++i) mine += keep[i];
    int total;
    int pos = block_excl_scan<kPlanWaves>(mine, s.scratch, total);
    for (int i = sa; i < sb; ++i) {
        const int kp = keep[i];
        dst[i] = kp ? pos : -1;
        pos += kp;
    }
    if (tid == 0) {
        stats[FF_STAT_LOUT] = total;
        stats[FF_STAT_MERGED] = L - total;
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
    int64_t* __restrict__ stats, unsigned char* __restrict__ flags, int64_t* host_mapped, int64_t seq) {
    using A = Act<DT>;
    __shared__ PlanLds s;
    const int tid = threadIdx.x;
    const int nv = (int)stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    const float thr_f = (float)thr;   // already T-valued
    int a, b;
    thread_range(0, nv, a, b);

    // main.py:113 - count of similarities >= T(threshold) (NaN compares false, -2 never passes)
    int c = 0;
    for (int j = a; j < b; ++j) c += (A::load1(sim, j) >= thr_f) ? 1 : 0;
    const int count = block_sum_i<kPlanWaves>(c, s.scratch);

    // main.py:114-116 in double, as python: ratio = count / ftn ; ratio < sub ?
    const double ratio = ftn > 0 ? (double)count / (double)ftn : 0.0;
    const bool use_topk = !(ratio < sub);
    long long k = 0;
    uint32_t kth = 0;
    int need = 0;
    if (use_topk) {
        k = (long long)(sub * (double)ftn);      // int(sub * ftn), main.py:122
        if (k > nv) k = nv;
        if (k < 0) k = 0;
        if (k > 0) kth = radix_select<DT>(sim, 0, nv, (int)k, s, need);
    }

    // member flags
    int taken = 0;
    if (!use_topk) {
        for (int j = a; j < b; ++j) flags[j] = (A::load1(sim, j) >= thr_f) ? 1 : 0;
    } else if (k == 0) {
        for (int j = a; j < b; ++j) flags[j] = 0;
    } else {
        int eq = 0;
        for (int j = a; j < b; ++j) eq += (order_key<DT>(A::bits1(sim, j)) == kth) ? 1 : 0;
        int eq_total;
        int rank = block_excl_scan<kPlanWaves>(eq, s.scratch, eq_total);
        for (int j = a; j < b; ++j) {
            const uint32_t key = order_key<DT>(A::bits1(sim, j));
            unsigned char f = key > kth;
            if (key == kth) { f = rank < need; ++rank; }
            flags[j] = f;
        }
        taken = need;
    }
    if (tid == 0) {
        stats[FF_STAT_COUNT] = count;
        stats[FF_STAT_BRANCH] = use_topk ? 1 : 0;
        stats[FF_STAT_K] = k;
        stats[FF_STAT_BELOW_LB] = (!use_topk && ratio < ratio_lb) ? 1 : 0;
        stats[FF_STAT_KTH_KEY] = kth;
        stats[FF_STAT_TIES_TAKEN] = taken;
    }
    plan_tail(flags, nv, order, L, true, run_len, dst, keep, stats, s);
    publish(stats, host_mapped, seq);
}

__global__ __launch_bounds__(kPlanThreads) void k_plan_from_index(
    const int64_t* __restrict__ merge_index, int n_merge, const int32_t* __restrict__ order, int L,
    int32_t* __restrict__ run_len, int32_t* __restrict__ dst, uint8_t* __restrict__ keep,
    int64_t* __restrict__ stats, unsigned char* __restrict__ flags) {
    __shared__ PlanLds s;
    const int nv = (int)stats[FF_STAT_NV];
    for (int j = threadIdx.x; j < nv; j += kPlanThreads) flags[j] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < n_merge; q += kPlanThreads) {
        const int64_t j = merge_index[q];
        if (j >= 0 && j < nv) flags[j] = 1;
    }
    __syncthreads();
    plan_tail(flags, nv, order, L, true, run_len, dst, keep, stats, s);
}

template <int DT>
__global__ __launch_bounds__(kPlanThreads) void k_plan_prune(
    const void* __restrict__ importance, int S, int start, int n_img, int k,
    int32_t* __restrict__ run_len, int32_t* __restrict__ dst, uint8_t* __restrict__ keep,
    int64_t* __restrict__ stats, unsigned char* __restrict__ flags, int64_t* host_mapped, int64_t seq) {
    using A = Act<DT>;
    __shared__ PlanLds s;
    const int lo = start, hi = start + n_img;
    uint32_t kth = 0;
    int need = 0;
    if (k > 0 && k < n_img) kth = radix_select<DT>(importance, lo, hi, k, s, need);
    // drop flags over [0, S): thread_range(0, S) slices, as plan_tail expects
    int a, b;
    thread_range(0, S, a, b);
    // rank of ties needs a scan restricted to [lo, hi)
    int eq = 0;
    if (k > 0 && k < n_img)
        for (int i = max(a, lo); i < min(b, hi); ++i) eq += (order_key<DT>(A::bits1(importance, i)) == kth) ? 1 : 0;
    int eq_total;
    int rank = block_excl_scan<kPlanWaves>(eq, s.scratch, eq_total);
    for (int i = a; i < b; ++i) {
        unsigned char drop = 0;
        if (i >= lo && i < hi) {
            if (k <= 0) drop = 1;
            else if (k >= n_img) drop = 0;
            else {
                const uint32_t key = order_key<DT>(A::bits1(importance, i));
                bool sel = key > kth;
                if (key == kth) { sel = rank < need; ++rank; }
                drop = !sel;
            }
        }
        flags[i] = drop;
    }
    if (threadIdx.x == 0) {
        stats[FF_STAT_NV] = S;
        stats[FF_STAT_K] = k;
        stats[FF_STAT_KTH_KEY] = kth;
        stats[FF_STAT_TIES_TAKEN] = need;
    }
    plan_tail(flags, S, nullptr, S, false, run_len, dst, keep, stats, s);
    publish(stats, host_mapped, seq);
}

}  // namespace ff

// The fused step (ff_abi.hip) reuses these launchers with its host-mapped result block.
namespace ff {
int launch_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double thr, double sub,
                      double ratio_lb, int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, int64_t* host_mapped, int64_t seq, hipStream_t st) {
    unsigned char* flags = (unsigned char*)ws;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_plan_merge<FF_F32>, dim3(1), dim3(kPlanThreads), 0, st, sim, order, (int)L, thr,
                               sub, ratio_lb, run_len, dst, keep, stats, flags, host_mapped, seq);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_plan_merge<FF_BF16>, dim3(1), dim3(kPlanThreads), 0, st, sim, order, (int)L, thr,
                               sub, ratio_lb, run_len, dst, keep, stats, flags, host_mapped, seq);
            break;
        default:
            hipLaunchKernelGGL(k_plan_merge<FF_F16>, dim3(1), dim3(kPlanThreads), 0, st, sim, order, (int)L, thr,
                               sub, ratio_lb, run_len, dst, keep, stats, flags, host_mapped, seq);
    }
    return (int)hipGetLastError();
}
}  // namespace ff

static int check_plan_args(const void* a, const void* b, const void* c, const void* d, const void* e,
                           int64_t L, void* ws, size_t ws_bytes) {
    if (!a || !b || !c || !d || !e || !ws || L < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (ws_bytes < (size_t)L) return FF_ERR_WORKSPACE;
    return FF_OK;
}

extern "C" int ff_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double threshold,
                             double sub, double ratio_lb, int32_t* run_len, int32_t* dst, uint8_t* keep,
                             int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(sim, run_len, dst, keep, stats, L, ws, ws_bytes);
    if (rc) return rc;
    if (!order) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
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
    if (L == 0) return FF_OK;
    hipLaunchKernelGGL(ff::k_plan_from_index, dim3(1), dim3(ff::kPlanThreads), 0, (hipStream_t)stream,
                       merge_index, (int)n_merge, order, (int)L, run_len, dst, keep, stats, (unsigned char*)ws);
    return (int)hipGetLastError();
}

namespace ff {
int launch_plan_prune(const void* imp, int dtype, int64_t S, int64_t start, int64_t n_img, int64_t k,
                      int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats, void* ws,
                      int64_t* host_mapped, int64_t seq, hipStream_t st) {
    unsigned char* flags = (unsigned char*)ws;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_plan_prune<FF_F32>, dim3(1), dim3(kPlanThreads), 0, st, imp, (int)S, (int)start,
                               (int)n_img, (int)k, run_len, dst, keep, stats, flags, host_mapped, seq);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_plan_prune<FF_BF16>, dim3(1), dim3(kPlanThreads), 0, st, imp, (int)S, (int)start,
                               (int)n_img, (int)k, run_len, dst, keep, stats, flags, host_mapped, seq);
            break;
        default:
            hipLaunchKernelGGL(k_plan_prune<FF_F16>, dim3(1), dim3(kPlanThreads), 0, st, imp, (int)S, (int)start,
                               (int)n_img, (int)k, run_len, dst, keep, stats, flags, host_mapped, seq);
    }
    return (int)hipGetLastError();
}
}  // namespace ff

extern "C" int ff_plan_prune(const void* importance, int dtype, int64_t S, int64_t start, int64_t n_img,
                             int64_t k, int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats,
                             void* ws, size_t ws_bytes, ff_stream_t stream) {
    int rc = check_plan_args(importance, run_len, dst, keep, stats, S, ws, ws_bytes);
    if (rc) return rc;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (start < 0 || n_img < 0 || start + n_img > S) return FF_ERR_ARG;
    if (S == 0) return FF_OK;
    return ff::launch_plan_prune(importance, dtype, S, start, n_img, k, run_len, dst, keep, stats, ws, nullptr, 0,
                                 (hipStream_t)stream);
}
