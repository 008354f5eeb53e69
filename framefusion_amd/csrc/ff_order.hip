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
#include "ff_common.h"

namespace ff {

constexpr int kOrderThreads = 1024;
constexpr int kOrderWaves = kOrderThreads / kWave;

__global__ __launch_bounds__(kOrderThreads) void k_build_order(
    const int64_t* __restrict__ ptype, int L, int P, int n_seg, int seg_len,
    int32_t* __restrict__ order, int64_t* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int bins = P + 1;
    int* hist = lds;                       // [n_seg][bins]
    int* scratch = lds + n_seg * bins;     // [kOrderWaves + 1] + misc
    int* misc = scratch + kOrderWaves + 1; // [0] = ftn
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();

    for (int x = tid; x < n_seg * bins; x += kOrderThreads) hist[x] = 0;
    if (tid == 0) misc[0] = 0;
    __syncthreads();

    const int seg_lo = w * seg_len;
    const int seg_hi = min(seg_lo + seg_len, L);

    // phase 1: per-segment histogram
    if (w < n_seg) {
        int* my = hist + w * bins;
        int not_text = 0;
        for (int i0 = seg_lo; i0 < seg_hi; i0 += kWave) {
            const int i = i0 + lane;
            if (i < seg_hi) {
                const int64_t t = ptype[i];
                const int key = (t >= 0 && t < P) ? (int)t : P;
                atomicAdd(&my[key], 1);
                not_text += (t != -1);
            }
        }
        not_text = wave_sum_i(not_text);
        if (lane == 0) atomicAdd(&misc[0], not_text);
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
        int* my = hist + w * bins;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        for (int i0 = seg_lo; i0 < seg_hi; i0 += kWave) {
            const int i = i0 + lane;
            const bool valid = i < seg_hi;
            int key = -1, pre = 0, slot = 0;
            if (valid) {
                const int64_t t = ptype[i];
                key = (t >= 0 && t < P) ? (int)t : P;
                pre = my[key];
            }
            __builtin_amdgcn_wave_barrier();
            if (valid) slot = atomicAdd(&my[key], 1);
            // lanes sharing a key in this step got unique slots in unspecified order: redo them
            // in lane (= sequence) order. `pre` is identical for all lanes of one key.
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
    int n_seg = ff::kOrderWaves;
    const size_t budget = 144 * 1024;
    while (n_seg > 1 && (size_t)n_seg * bins * sizeof(int) > budget) n_seg >>= 1;
    const size_t lds = ((size_t)n_seg * bins + ff::kOrderWaves + 8) * sizeof(int);
    if (lds > 160 * 1024) return FF_ERR_UNSUPPORTED;
    int seg_len = (int)((L + n_seg - 1) / n_seg);
    seg_len = (seg_len + ff::kWave - 1) / ff::kWave * ff::kWave;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ff::k_build_order,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(ff::k_build_order, dim3(1), dim3(ff::kOrderThreads), lds, (hipStream_t)stream,
                       patch_type, (int)L, (int)patch_num, n_seg, seg_len, order, stats);
    return (int)hipGetLastError();
}
