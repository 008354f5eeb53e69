// K0 - by-patch order: a stable counting sort of the sequence positions by patch type, built by
// ONE 16-wave workgroup with its per-wave histograms in LDS (the whole input is L int64 = a few
// hundred KB, far below what a multi-workgroup sort would pay in launch boundaries).
//
// Replaces torch.where(patch_type == arange(P)[:, None]) of framefusion/main.py:208-210, which
// materialises a [P, L] bool matrix (21 MB at 64x576) and a nonzero() host sync.
//
// Layout: bins 0..P-1 = patch types, bin P = "everything else" (text, out-of-range types), so
// order[] comes out as a full permutation: the by-patch visual order followed by the remaining
// positions in sequence order.  Wave w owns the contiguous sequence segment w and the histogram
// row hist[w][*]; stability across segments comes from the column scan, stability inside a
// 64-token step from the duplicate fix-up (LDS atomics give unique but unordered slots).
// When they fit beside the histograms, the bin ids of all positions are staged in LDS as uint16:
// patch_type is then read once, coalesced, with all 16 waves' loads in flight together.
#include "ff_common.h"

namespace ff {

constexpr int kOrderThreads = 1024;
constexpr int kOrderWaves = kOrderThreads / kWave;

__device__ inline int bin_of(int64_t t, int P) { return (t >= 0 && t < P) ? (int)t : P; }

// One 64-token step of the stable placement for wave-private histogram row `my`.
__device__ inline void place_step(int* my, int key, bool valid, int i, int32_t* __restrict__ order,
                                  unsigned long long lt_mask) {
    int pre = 0, slot = 0;
    if (valid) pre = my[key];
    __builtin_amdgcn_wave_barrier();
    if (valid) slot = atomicAdd(&my[key], 1);
    // lanes sharing a key in this step got unique slots in unspecified order: redo them in lane
    // (= sequence) order. `pre` is identical for all lanes of one key.
    unsigned long long todo = __ballot(valid && slot != pre);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, kWave);
        const unsigned long long same = __ballot(valid && key == k);
        if (valid && key == k) slot = pre + __popcll(same & lt_mask);
        todo &= ~same;
    }
    if (valid) order[slot] = i;
}

// kKeysInLds: the bin id of every position is staged once in LDS as uint16 (all 16 waves load
// patch_type together, fully coalesced, many loads in flight); the histogram and placement sweeps
// are then pure LDS traffic.  Otherwise the two sweeps re-read patch_type in batches of 8 steps.
template <bool kKeysInLds>
__global__ __launch_bounds__(kOrderThreads) void k_build_order(
    const int64_t* __restrict__ ptype, int L, int P, int n_seg, int seg_len,
    int32_t* __restrict__ order, int64_t* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int bins = P + 1;
    int* hist = lds;                       // [n_seg][bins]
    int* scratch = lds + n_seg * bins;     // [kOrderWaves + 1] + misc
    int* misc = scratch + kOrderWaves + 1; // [0] = ftn
    uint16_t* keys = (uint16_t*)(misc + 7);   // [L] when kKeysInLds
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int seg_lo = w * seg_len;
    const int seg_hi = min(seg_lo + seg_len, L);
    int* my = hist + w * bins;

    for (int x = tid; x < n_seg * bins; x += kOrderThreads) hist[x] = 0;
    if (tid == 0) { misc[0] = 0; misc[1] = 0x7fffffff; misc[2] = -1; misc[3] = 0; misc[4] = 0; }
    __syncthreads();

    // phase 0: one coalesced read of patch_type by all 16 waves.  Besides staging the bin ids it
    // recognises the frame-major layout the adapters produce for the first call of a prefill
    // (pre text rows, F x [0..P-1], post text rows; llava_video.py:321-336): visual positions
    // contiguous, a whole number of frames, and type == (i - pre) mod P.  That layout has the
    // closed form order[p*F + f] = pre + f*P + p, written with coalesced stores; anything else
    // (ragged later layers, text between frames) takes the counting sort below.  All index
    // arithmetic is incremental (one integer division per thread, not per element).
    const uint64_t clk0 = __builtin_amdgcn_s_memtime();
    {
        int not_text = 0;
        int v_min = 0x7fffffff, v_max = -1, v_cnt = 0;
        for (int i0 = 0; i0 < L; i0 += 8 * kOrderThreads) {
            int64_t t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kOrderThreads + tid;
                t[u] = i < L ? ptype[i] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kOrderThreads + tid;
                if (i < L) {
                    const int bin = bin_of(t[u], P);
                    if constexpr (kKeysInLds) keys[i] = (uint16_t)bin;
                    not_text += (t[u] != -1);
                    if (bin < P) { v_min = min(v_min, i); v_max = max(v_max, i); ++v_cnt; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v_min = min(v_min, __shfl_xor(v_min, o, kWave)); v_max = max(v_max, __shfl_xor(v_max, o, kWave));
            v_cnt += __shfl_xor(v_cnt, o, kWave); not_text += __shfl_xor(not_text, o, kWave);
        }
        if (lane == 0) {
            atomicAdd(&misc[0], not_text);
            atomicMin(&misc[1], v_min); atomicMax(&misc[2], v_max); atomicAdd(&misc[3], v_cnt);
        }
        __syncthreads();
        const int pre = misc[1], last = misc[2], nv = misc[3];
        if (nv > 0 && last - pre + 1 == nv && nv % P == 0) {      // uniform
            int bad = 0;
            int e = tid % P;
            const int step = kOrderThreads % P;
            for (int i = pre + tid; i < pre + nv; i += kOrderThreads) {
                int bin;
                if constexpr (kKeysInLds) bin = keys[i];
                else bin = bin_of(ptype[i], P);
                bad |= (bin != e);
                e += step;
                e = e >= P ? e - P : e;
            }
            if (bad) atomicOr(&misc[4], 1);
            __syncthreads();
            if (misc[4] == 0) {
                const int F = nv / P;
                int p = tid / F, f = tid - p * F;
                const int step_p = kOrderThreads / F, step_f = kOrderThreads - step_p * F;
                for (int j = tid; j < nv; j += kOrderThreads) {
                    order[j] = pre + f * P + p;
                    f += step_f; p += step_p;
                    if (f >= F) { f -= F; ++p; }
                }
                for (int q = tid; q < L - nv; q += kOrderThreads) order[nv + q] = q < pre ? q : q + nv;
                if (tid == 0) {
                    stats[FF_STAT_NV] = nv;
                    stats[FF_STAT_FTN] = misc[0];
                    stats[FF_STAT_T_ORDER] = (int64_t)(__builtin_amdgcn_s_memtime() - clk0);
                    stats[FF_STAT_T_ORDER + 1] = 0;
                }
                return;
            }
        }
    }
    const uint64_t clk1 = __builtin_amdgcn_s_memtime();

    // phase 1: per-segment histogram
    if (w < n_seg) {
        if constexpr (kKeysInLds) {
            for (int i = seg_lo + lane; i < seg_hi; i += kWave) atomicAdd(&my[keys[i]], 1);
        } else {
            for (int i0 = seg_lo; i0 < seg_hi; i0 += 8 * kWave) {
                int64_t t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * kWave + lane;
                    t[u] = i < seg_hi ? ptype[i] : -1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * kWave + lane;
                    if (i < seg_hi) {
                        atomicAdd(&my[bin_of(t[u], P)], 1);
                    }
                }
            }
        }
    }
    __syncthreads();

    // phase 2: hist[s][p] <- first output slot of (type p, segment s)
    int carry = 0;
    for (int p0 = 0; p0 < bins; p0 += kOrderThreads) {
        const int p = p0 + tid;
        int col = 0;
        if (p < bins)
            for (int s = 0; s < n_seg; ++s) col += hist[s * bins + p];
        int chunk_total;
        int run = carry + block_excl_scan<kOrderWaves>(col, scratch, chunk_total);
        carry += chunk_total;
        if (p < bins) {
            if (p == P) {
                stats[FF_STAT_NV] = run;
                stats[FF_STAT_FTN] = misc[0];
            }
            for (int s = 0; s < n_seg; ++s) {
                const int c = hist[s * bins + p];
                hist[s * bins + p] = run;
                run += c;
            }
        }
    }
    __syncthreads();

    // phase 3: stable placement
    if (w < n_seg) {
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        if constexpr (kKeysInLds) {
            for (int i0 = seg_lo; i0 < seg_hi; i0 += kWave) {
                const int i = i0 + lane;
                const bool valid = i < seg_hi;
                place_step(my, valid ? (int)keys[i] : -1, valid, i, order, lt_mask);
            }
        } else {
            for (int i0 = seg_lo; i0 < seg_hi; i0 += 8 * kWave) {
                int64_t t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * kWave + lane;
                    t[u] = i < seg_hi ? ptype[i] : -1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * kWave + lane;
                    if (i0 + u * kWave < seg_hi) place_step(my, bin_of(t[u], P), i < seg_hi, i, order, lt_mask);
                }
            }
        }
    }
    if (tid == 0) {
        stats[FF_STAT_T_ORDER] = (int64_t)(clk1 - clk0);
        stats[FF_STAT_T_ORDER + 1] = (int64_t)(__builtin_amdgcn_s_memtime() - clk1);
    }
}

}  // namespace ff

extern "C" int ff_build_order(const int64_t* patch_type, int64_t L, int64_t patch_num, int32_t* order,
                              int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream) {
    (void)ws; (void)ws_bytes;
    if (!patch_type || !order || !stats || L < 0 || patch_num < 1) return FF_ERR_ARG;
    if (L >= (1ll << 31) || patch_num > 32768) return FF_ERR_UNSUPPORTED;
    if (L == 0) return FF_OK;
    const int bins = (int)patch_num + 1;
    // LDS plan: per-segment histograms [n_seg][bins] (+ the uint16 bin ids of all positions when
    // they fit, trading segments for the staging down to 4 segments).
    const size_t cap = 160 * 1024, fixed = (ff::kOrderWaves + 8) * sizeof(int);
    const size_t key_lds = ((size_t)L * sizeof(uint16_t) + 15) & ~(size_t)15;
    int n_seg = ff::kOrderWaves;
    while (n_seg > 1 && (size_t)n_seg * bins * sizeof(int) + fixed > cap - 16 * 1024) n_seg >>= 1;
    if ((size_t)n_seg * bins * sizeof(int) + fixed > cap) return FF_ERR_UNSUPPORTED;
    bool in_regs = false;   // bin ids staged in LDS
    for (int s = n_seg; s >= 4 || s == n_seg; s >>= 1) {
        if ((size_t)s * bins * sizeof(int) + fixed + key_lds <= cap) { n_seg = s; in_regs = true; break; }
        if (s == 1) break;
    }
    const size_t lds = (size_t)n_seg * bins * sizeof(int) + fixed + (in_regs ? key_lds : 0);
    int seg_len = (int)((L + n_seg - 1) / n_seg);
    seg_len = (seg_len + ff::kWave - 1) / ff::kWave * ff::kWave;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ff::k_build_order<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ff::k_build_order<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (in_regs)
        hipLaunchKernelGGL(ff::k_build_order<true>, dim3(1), dim3(ff::kOrderThreads), lds, (hipStream_t)stream,
                           patch_type, (int)L, (int)patch_num, n_seg, seg_len, order, stats);
    else
        hipLaunchKernelGGL(ff::k_build_order<false>, dim3(1), dim3(ff::kOrderThreads), lds, (hipStream_t)stream,
                           patch_type, (int)L, (int)patch_num, n_seg, seg_len, order, stats);
    return (int)hipGetLastError();
}
