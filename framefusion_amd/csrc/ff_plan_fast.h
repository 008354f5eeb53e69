// The fast plan (16-bit values, <= 163 840 tokens) as a device function: the body of k_plan_fast (ff_plan.hip).
#pragma once

#include "ff_common.h"

namespace ff {

template <int DT> __device__ inline uint32_t nan_key() { return Act<DT>::kKeyBits == 32 ? 0xffffffffu : 0xffffu; }

struct PlanParams {
    int mode;            // 0: merge (threshold / top-k decided from the count), 1: prune (top-k given)
    int lo, hi;          // value range the selection runs over (merge: [0, Nv))
    long long k_given;   // prune: k; merge: >= 0 forces top-k with this k, -1 = threshold/budget policy
    double sub, ratio_lb;
    uint32_t thr_key;
    int p0_guess;        // expected top byte of the k-th key (speculative prefetch of its level-1 rows)
    int n_slices;
    int32_t* src_out;    // optional (prune): src_out[dst[i]] = i for every kept position - the inverse of dst[], what the
                         // prune's gather walks by OUTPUT rows (ff_merge_body.h, prune_gather_body)
};

// Copy of the result block into device-visible pinned host memory: one lane per word (a single
// store instruction crosses PCIe once), a system fence, then the sequence word the host polls.
__device__ inline void publish(int64_t* __restrict__ stats, int64_t* host_mapped, int64_t seq) {
    const int lane = threadIdx.x;          // called by wave 0
    if (lane < FF_STAT_T_ORDER && lane != FF_STAT_SEQ)       // (the pinned words behind are the host's: FF_MAIL_WORD)
        __hip_atomic_store(&host_mapped[lane], stats[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane == FF_STAT_ERROR) stats[lane] = 0;     // reported; the next call starts clean
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(&host_mapped[FF_STAT_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- k_plan_fast: the same plan for 16-bit values of at most 163 840 tokens -----------------------------
// The general kernel above is bound by instruction issue (16 waves on one CU walk ~2 500 instructions
// each: register-chunk selection, shuffle scans through LDS, divergent branches).  Here a workgroup of
// kFastThreads threads owns kFastThreads slots and as many positions - ONE of each per thread, loaded in
// the first round together with the tables (staging ALL values in LDS was tried: one CU pulls ~12 B/clk, the
// first barrier came at 4.9 us).  The only load that depends on the decision is the 8 KB slice that holds
// the tie slot t*; the wave scans run on the DPP network and the classification is branch-free.  The rest
// is as above: tables -> k-th key by wave 0, tie slot t*, per-workgroup totals exchanged as {tag, count}
// granules (published before the member flags are computed, so the hop overlaps with work; a workgroup
// polls its <= 127 predecessors in two passes of 64 lanes).
// NT threads per workgroup = NT slots + NT positions per workgroup; LDS: plan_fast_lds_bytes(RS, NT).
constexpr int kRowSlicesLds = 16;           // level-1 rows kept in LDS for the tie-slice search (16 slices = 65 536 tokens)
constexpr int kFastSlicesBig = 40;          // second instantiation: 40 slices = 163 840 tokens (256 frames x 576 = 147 456)
constexpr int kFastMaxL = kFastSlicesBig * 4096;   // (the level-1 rows of every slice sit in LDS: 16 or 40 KB per workgroup)
__host__ __device__ constexpr size_t plan_fast_lds_bytes(int rs, int nt) {
    return (size_t)(rs + nt / 256) * 256 * 4 + (2 * (nt / 64) + 16) * 4;
}

template <int DT, int RS, int NT>
__device__ inline void plan_fast_body(
    const void* __restrict__ values, int cap, const PlanParams& pp, const int* __restrict__ l0, int* t16_end,
    int64_t* __restrict__ stats, const int32_t* __restrict__ inv, int L,
    uint8_t* __restrict__ member, uint8_t* __restrict__ keep, int32_t* __restrict__ dst,
    unsigned long long* agg, uint32_t* tagword, int64_t* host_mapped, int64_t seq,
    unsigned char* lds_raw, const int bid, const int nblk) {
    using A = Act<DT>;
    static_assert(A::kBytes == 2, "16-bit values only");
    constexpr int NW = NT / kWave, NQ = NT / 256, kSpec = RS / NQ;
    constexpr int kTiePer = kSelSlice / NT;      // slots per thread in the tie-slot search
    // LDS: [NQ][256] partial column sums | 2 * NW scan ints | 16 broadcast ints | level-1 rows, one KiB per slice of
    // the call (at most RS; last, so that a launch allocates pp.n_slices of them: plan_fast_lds_bytes(pp.n_slices, NT))
    int (*part)[256] = (int (*)[256])lds_raw;
    int* scratch = (int*)(lds_raw + NQ * 256 * 4);                  // [2 * NW]
    int (*rows)[256] = (int (*)[256])(lds_raw + NQ * 256 * 4 + (2 * NW + 16) * 4);
    int* bcast = scratch + 2 * NW;                                              // [16]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = tid & 255, q = tid >> 8;
    const int i0 = bid * NT + tid;        // my position and my slot
#ifdef FF_PLAN_PROBE
    long long stamp[8];
    stamp[0] = wall_clock64();
    const long long wave_t0 = stamp[0];
#endif

    // ---- round 1: tables, inv[] of my position, and all values -> LDS (every load issued before any is used)
    const long long nv_raw = stats[FF_STAT_NV];
    const long long ftn = stats[FF_STAT_FTN];
    const uint32_t tag_prev = *tagword;
    int l0v[kL0Copies / NQ];
#pragma unroll
    for (int x = 0; x < kL0Copies / NQ; ++x) l0v[x] = l0[(q + x * NQ) * kL0Stride + c];
    const int l0cnt_raw = l0[(tid & (kL0Copies - 1)) * kL0Stride + 256];
    int specv[kSpec][kT16Copies];
    {
        const uint32_t bin = t16_bin(((uint32_t)pp.p0_guess << 8) | (uint32_t)c);
#pragma unroll
        for (int j = 0; j < kSpec; ++j) {
            if (j * NQ < pp.n_slices) {                                     // (uniform; the row is clamped: used for g < n_slices only)
                const int* row = t16_slice(t16_end, min(q + j * NQ, pp.n_slices - 1)) + bin;
#pragma unroll
                for (int x = 0; x < kT16Copies; ++x) specv[j][x] = row[x * 65536];
            } else {
#pragma unroll
                for (int x = 0; x < kT16Copies; ++x) specv[j][x] = 0;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t inv_rsrc = make_rsrc(inv ? (const void*)inv : values, inv ? (uint32_t)L * 4u : 0u);
    const int inv1 = __builtin_amdgcn_raw_buffer_load_b32(inv_rsrc, i0 * 4, 0, 0);
    const uint32_t my_bits = A::bits1(values, min(i0, cap - 1));          // the value of my slot
    __builtin_amdgcn_sched_barrier(0);
    const int nv = pp.mode == 0 ? (int)nv_raw : L;
    const int lo = pp.mode == 0 ? 0 : pp.lo, hi = pp.mode == 0 ? nv : pp.hi;
    {
        int colsum = 0;
#pragma unroll
        for (int x = 0; x < kL0Copies / NQ; ++x) colsum += l0v[x];
        part[q][c] = colsum;
        if (tid < kL0Copies) scratch[tid] = l0cnt_raw;
    }
    // the value of my position's slot: one dependent gather, in flight while the levels are resolved
    const int pslot = inv ? inv1 : i0;
    const uint32_t pos_bits = A::bits1(values, min(max(pslot, 0), cap - 1));
#ifdef FF_PLAN_PROBE
    stamp[1] = wall_clock64();
    __shared__ long long wstart[16], wready[16];
    if (lane == 0) { wstart[wv] = wave_t0; wready[wv] = stamp[1]; }
#endif
    __syncthreads();
#ifdef FF_PLAN_PROBE
    if (bid == nblk - 1 && tid == 0) {
        long long s_max = 0, r_max = 0, r_min = 1ll << 62;
        for (int x = 0; x < NW; ++x) { s_max = max(s_max, wstart[x]); r_max = max(r_max, wready[x]); r_min = min(r_min, wready[x]); }
        stats[FF_STAT_T_ORDER] = s_max - stamp[0]; stats[FF_STAT_T_ORDER + 1] = r_min - stamp[0]; stats[FF_STAT_T_ORDER + 2] = r_max - stamp[0];
        long long packed = 0;       // per-wave ready times, 0.1 us units, 4 bits each (clamped)
        for (int x = 0; x < NW; ++x) { long long u = (wready[x] - stamp[0]) / 40; packed |= (u > 15 ? 15 : u) << (4 * x); }
        stats[FF_STAT_T_ORDER + 3] = packed;
    }
#endif
#ifdef FF_PLAN_PROBE
    stamp[2] = wall_clock64();
#endif
    // ---- level 0: decision + top byte of the k-th key (wave 0)
    auto pick = [&](int rem, int& bin, int& above) {
        const int top = 255 - 4 * lane;
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = 0;
#pragma unroll
            for (int x = 0; x < NQ; ++x) v[e] += part[x][top - e];
        }
        const int sum = v[0] + v[1] + v[2] + v[3];
        const int incl = wave_incl_scan_dpp(sum);
        const int first = __ffsll((long long)__ballot(incl >= rem)) - 1;
        int ab = incl - sum, b = top;
        if (ab + v[0] >= rem) { b = top; }
        else if (ab + v[0] + v[1] >= rem) { ab += v[0]; b = top - 1; }
        else if (ab + v[0] + v[1] + v[2] >= rem) { ab += v[0] + v[1]; b = top - 2; }
        else { ab += v[0] + v[1] + v[2]; b = top - 3; }
        bin = __builtin_amdgcn_readlane(b, first);
        above = __builtin_amdgcn_readlane(ab, first);
    };
    if (wv == 0) {
        int cnt = lane < kL0Copies ? scratch[lane] : 0;
        cnt = __builtin_amdgcn_readlane(wave_incl_scan_dpp(cnt), 63);
        bool topk;
        long long k;
        if (pp.mode == 0 && pp.k_given >= 0) {
            topk = true;                                  // fixed-sparsity policy (modeling_qwen2_baseline.py:920,1001)
            k = pp.k_given > nv ? (long long)nv : pp.k_given;
        } else if (pp.mode == 0) {
            // main.py:114-116 in double, as python: ratio = count / ftn ; ratio < sub ?
            const double ratio = ftn > 0 ? (double)cnt / (double)ftn : 0.0;
            topk = !(ratio < pp.sub);
            k = 0;
            if (topk) {
                k = (long long)(pp.sub * (double)ftn);   // int(sub * ftn), main.py:122
                if (k > nv) k = nv;
                if (k < 0) k = 0;
            }
        } else {
            topk = true;
            k = pp.k_given;
        }
        int bin = 0, above = 0;
        if (topk && k > 0) pick((int)k, bin, above);
        if (lane == 0) { bcast[0] = topk ? 1 : 0; bcast[1] = cnt; bcast[2] = (int)k; bcast[3] = bin; bcast[4] = (int)k - above; }
    }
    __syncthreads();
#ifdef FF_PLAN_PROBE
    stamp[3] = wall_clock64();
#endif
    const bool is_topk = bcast[0] != 0;
    const int count = bcast[1], k_sel = bcast[2];
    const bool topk = is_topk && k_sel > 0;
    uint32_t kth = 0;
    int need = 0, tstar = -1;
    if (topk) {                                              // (uniform)
        // ---- level 1: the rows of the level-0 bin (the speculated ones if the guess was right)
        const int p0 = bcast[3];
        int colsum = 0;
        if (p0 == pp.p0_guess) {
#pragma unroll
            for (int j = 0; j < kSpec; ++j) {
                int a = 0;
#pragma unroll
                for (int x = 0; x < kT16Copies; ++x) a += specv[j][x];
                a = q + j * NQ < pp.n_slices ? a : 0;
                if (q + j * NQ < pp.n_slices) rows[q + j * NQ][c] = a;
                colsum += a;
            }
        } else {
            const uint32_t bin = t16_bin(((uint32_t)p0 << 8) | (uint32_t)c);
#pragma unroll
            for (int j = 0; j < kSpec; ++j) {
                int a = 0;
                if (q + j * NQ < pp.n_slices) {
                    const int* row = t16_slice(t16_end, q + j * NQ) + bin;
#pragma unroll
                    for (int x = 0; x < kT16Copies; ++x) a += row[x * 65536];
                }
                if (q + j * NQ < pp.n_slices) rows[q + j * NQ][c] = a;
                colsum += a;
            }
        }
        part[q][c] = colsum;
        __syncthreads();
        if (wv == 0) {
            int p1, above;
            pick(bcast[4], p1, above);
            const int rem = bcast[4] - above;
            // the slice that holds the rem-th entry equal to the k-th key (entries per slice = rows[g][p1])
            static_assert(RS <= kWave, "one lane per slice in the tie-slice search");
            const int ties = lane < pp.n_slices ? rows[lane][p1] : 0;
            const int tincl = wave_incl_scan_dpp(ties);
            const int hit = __ffsll((long long)__ballot(tincl >= rem)) - 1;
            const int before_hit = __builtin_amdgcn_readlane(tincl - ties, hit);
            if (lane == 0) { bcast[5] = (p0 << 8) | p1; bcast[6] = rem; bcast[7] = hit; bcast[8] = rem - before_hit; }
        }
        __syncthreads();
        kth = (uint32_t)bcast[5];
        need = bcast[6];
        const int sl = bcast[7], want = bcast[8];
        // ---- t*: the slot of the want-th entry equal to kth inside slice sl: kTiePer slots per thread (the
        // one load of the kernel that depends on the decision)
        const int t0 = sl * kSelSlice + tid * kTiePer;
        uint32_t w[kTiePer / 2];
        if (t0 + kTiePer <= cap) {
            const uint32_t* src = (const uint32_t*)((const uint16_t*)values + t0);
#pragma unroll
            for (int e = 0; e < kTiePer / 2; ++e) w[e] = src[e];
        } else {
#pragma unroll
            for (int e = 0; e < kTiePer / 2; ++e) w[e] = 0u;
            for (int e = 0; e < kTiePer; ++e)
                if (t0 + e < cap) w[e >> 1] |= A::bits1(values, t0 + e) << (16 * (e & 1));
        }
        uint32_t tie = 0;
#pragma unroll
        for (int e = 0; e < kTiePer; ++e) {
            const uint32_t key = order_key<DT>((w[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            const int t = t0 + e;
            tie |= (uint32_t)((t >= lo) & (t < hi) & (key == kth)) << e;
        }
        const int mine = __popc(tie);
        const int wincl = wave_incl_scan_dpp(mine);
        if (lane == 63) scratch[wv] = wincl;
        __syncthreads();
        int ex = wincl - mine;
#pragma unroll
        for (int x = 0; x < NW; ++x) ex += x < wv ? scratch[x] : 0;
        if (ex < want && want <= ex + mine) {
            int seen = ex, found = -1;
#pragma unroll
            for (int e = 0; e < kTiePer; ++e) {
                seen += (tie >> e) & 1u;
                found = (found < 0 && ((tie >> e) & 1u) && seen == want) ? t0 + e : found;
            }
            bcast[9] = found;
        }
        __syncthreads();
        tstar = bcast[9];
    }
#ifdef FF_PLAN_PROBE
    stamp[4] = wall_clock64();
#endif
    // is slot t (raw value bits) folded (mode 0) / dropped (mode 1)?  branch-free
    auto folded = [&](uint32_t bits, int t) -> uint32_t {
        const uint32_t key = order_key<DT>(bits);
        const bool in = (t >= lo) & (t < hi);
        const bool sel_topk = topk & in & ((key > kth) | ((key == kth) & (t <= tstar)));
        const bool sel_thr = in & (key >= pp.thr_key) & (key != nan_key<DT>());
        const bool sel = is_topk ? sel_topk : sel_thr;
        return (uint32_t)(pp.mode == 0 ? (sel & (t > 0)) : (in & !sel));
    };
    // ---- keep flag of my position (its value through inv[]) + the scan inside the workgroup
    const uint32_t kp = (uint32_t)(i0 < L) & (folded(pos_bits, pslot) ^ 1u);
    const int wincl = wave_incl_scan_dpp((int)kp);
    if (lane == 63) scratch[NW + wv] = wincl;
    uint32_t tag = tag_prev + 1u;
    if (tag == 0u) tag = 1u;
    __syncthreads();
    int span_total = 0, ex = wincl - (int)kp;
#pragma unroll
    for (int x = 0; x < NW; ++x) { span_total += scratch[NW + x]; ex += x < wv ? scratch[NW + x] : 0; }
    // ---- kept positions before my range = the totals of the workgroups before me: publish mine first
    if (tid == 0)
        __hip_atomic_store(&agg[bid], ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)span_total,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef FF_PLAN_PROBE
    stamp[5] = wall_clock64();
#endif
    // ---- member flag of my slot (while the totals travel)
    if (i0 < L) member[i0] = (uint8_t)folded(my_bits, i0);
    if (wv == 0) {
        int sum = 0;
        for (int g0 = 0; g0 < bid; g0 += kWave) {
            const int gg = g0 + lane;
            const bool need_it = gg < bid;
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
        sum = __builtin_amdgcn_readlane(wave_incl_scan_dpp(sum), 63);
        if (lane == 0) bcast[10] = sum;
    }
    __syncthreads();
    const int before = bcast[10];
#ifdef FF_PLAN_PROBE
    stamp[6] = wall_clock64();
#endif
    if (i0 < L) {
        dst[i0] = kp ? before + ex : -1;
        keep[i0] = (uint8_t)kp;
        if (pp.src_out && kp) pp.src_out[before + ex] = i0;
    }
    if (bid == nblk - 1) {
        if (tid == 0) {
            *tagword = tag;                          // (every workgroup has read the old value: they all published)
#ifdef FF_PLAN_PROBE
            stamp[7] = wall_clock64();
            for (int x = 1; x < 8; ++x) stats[FF_STAT_T_PLAN + x - 1] = stamp[x] - stamp[0];
#endif
            const int l_out = before + span_total;
            stats[FF_STAT_LOUT] = l_out;
            stats[FF_STAT_MERGED] = L - l_out;
            if (pp.mode == 0) {
                const double ratio = ftn > 0 ? (double)count / (double)ftn : 0.0;
                stats[FF_STAT_COUNT] = count;
                stats[FF_STAT_BRANCH] = is_topk ? 1 : 0;
                stats[FF_STAT_BELOW_LB] = (!is_topk && ratio < pp.ratio_lb) ? 1 : 0;
            } else {
                stats[FF_STAT_NV] = L;
            }
            stats[FF_STAT_K] = k_sel;
            stats[FF_STAT_KTH_KEY] = kth;
            stats[FF_STAT_TIES_TAKEN] = topk ? need : 0;
        }
        __syncthreads();
        if (host_mapped && tid < kWave) publish(stats, host_mapped, seq);
    }
}

}  // namespace ff
