// K1 - adjacent-pair cosine similarity in by-patch order, HBM-bound (0.25 flop/byte: plain FMA,
// no MFMA).  Replaces the two [Nv-1, d] gathers and ~11 elementwise/reduction passes of
// framefusion/main.py:216-238 + cosine_similarity (main.py:345-349) with ONE read of every row.
//
// Work decomposition: a wave owns kPairs consecutive by-patch positions j0..j0+kPairs-1, i.e. the
// kPairs+1 rows order[j0-1 .. j0+kPairs-1] (one row of overlap with the previous wave, which sits in
// the same workgroup for 3 of 4 waves and is served by L2).  It walks the feature dimension in
// 1 KiB tiles - each lane one 16-byte load per row per tile, fully coalesced - keeping kPairs+1
// independent loads in flight per tile and the next tile's loads issued before the current tile's
// arithmetic.  Per row it accumulates |x|^2 once (shared by the two pairs the row takes part in),
// per pair the sum of T-rounded products; one butterfly reduction per wave at the end.
//
// Rounding recipe, T = activation dtype:
//     dot = T(sum_fp32 T(a_i*b_i));  na = T(sqrt_fp32(sum_fp32 a_i^2));  sim = T(dot / T(na*nb))
// sqrt and divide are the correctly rounded fp32 forms (hipcc default).
#include <stdlib.h>

#include "ff_common.h"

namespace ff {

template <int DT, int kPairs, int kSimThreads>
__global__ __launch_bounds__(kSimThreads) void k_pair_similarity(
    const char* __restrict__ hidden, uint32_t row_bytes, const int64_t* __restrict__ ptype,
    const int32_t* __restrict__ order, const int64_t* __restrict__ stats, void* __restrict__ sim,
    int* __restrict__ l0, float thr) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    constexpr int R = kPairs + 1;
    constexpr int kSimWaves = kSimThreads / kWave;
    const int nv = (int)stats[FF_STAT_NV];
    const int lane = lane_id();
    const int j0 = uniform((blockIdx.x * kSimWaves + wave_id()) * kPairs);
    if (j0 >= nv) return;

    __amdgpu_buffer_rsrc_t row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int j = j0 - 1 + r;
        j = j < 0 ? 0 : (j >= nv ? nv - 1 : j);
        row[r] = make_rsrc(hidden + (int64_t)uniform(order[j]) * row_bytes, row_bytes);
    }

    float nrm[R], dot[kPairs];
#pragma unroll
    for (int r = 0; r < R; ++r) nrm[r] = 0.f;
#pragma unroll
    for (int r = 0; r < kPairs; ++r) dot[r] = 0.f;

    // 1 KiB tiles; tile t+1 is in flight while tile t is reduced
    const uint32_t lane_off = (uint32_t)lane * 16;
    const uint32_t tiles = (row_bytes + 1023u) >> 10;
    uint4 cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = buf_load16(row[r], lane_off);
    for (uint32_t t = 0; t < tiles; ++t) {
        const uint32_t noff = lane_off + (t + 1) * 1024u;      // past-the-end lanes read zeros
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[r] = buf_load16(row[r], noff);

        float prev[E], x[E];
        A::unpack(cur[0], prev);
#pragma unroll
        for (int e = 0; e < E; ++e) nrm[0] = __builtin_fmaf(prev[e], prev[e], nrm[0]);
#pragma unroll
        for (int r = 1; r < R; ++r) {
            A::unpack(cur[r], x);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                nrm[r] = __builtin_fmaf(x[e], x[e], nrm[r]);
                dot[r - 1] += A::rnd(prev[e] * x[e]);
                prev[e] = x[e];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }

#pragma unroll
    for (int r = 0; r < R; ++r) nrm[r] = wave_sum(nrm[r]);
#pragma unroll
    for (int r = 0; r < kPairs; ++r) dot[r] = wave_sum(dot[r]);

    float mine = -2.0f;
    bool have = false;
#pragma unroll
    for (int r = 0; r < kPairs; ++r) {
        if (lane == r) {
            const int j = j0 + r;
            if (j < nv) {
                float s = -2.0f;   // IGNORE_TOKEN (main.py:225-238)
                if (j > 0 && ptype[order[j - 1]] == ptype[order[j]]) {
                    const float d = A::rnd(dot[r]);
                    const float na = A::rnd(sqrtf(nrm[r]));
                    const float nb = A::rnd(sqrtf(nrm[r + 1]));
                    const float den = A::rnd(na * nb);
                    s = A::rnd(d / den);
                }
                A::store1(sim, j, s);
                mine = s;
                have = true;
            }
        }
    }
    if (l0) {
        // level-0 statistics of the select that follows (ff_plan.hip): top byte of the
        // order-preserving key + count(sim >= thr), folded over the wave's kPairs values and added
        // to one of 16 table copies so the (non-returning) atomics never pile up on one L2 word
        int* tab = l0 + (blockIdx.x & 15) * 260;
        uint32_t bits;
        if constexpr (DT == FF_F32) bits = __float_as_uint(mine);
        else if constexpr (DT == FF_BF16) bits = __float_as_uint(mine) >> 16;
        else { _Float16 h = (_Float16)mine; bits = (uint32_t)__builtin_bit_cast(uint16_t, h); }
        const int bin = (int)(order_key<DT>(bits) >> (A::kKeyBits - 8));
        const unsigned long long vm = __ballot(have);
        const int n_ge = __popcll(__ballot(have && mine >= thr));
        if (lane == 0 && n_ge) atomicAdd(&tab[256], n_ge);
        int mult = 0;
        bool leader = have;
#pragma unroll
        for (int q = 0; q < kPairs; ++q) {
            const int bq = __builtin_amdgcn_readlane(bin, q);
            if (((vm >> q) & 1ull) && bq == bin) {
                ++mult;
                if (q < lane) leader = false;
            }
        }
        if (leader) atomicAdd(&tab[bin], mult);
    }
}

template <int DT, int kPairs, int kSimThreads>
static int launch_similarity_pt(const void* hidden, int64_t L, int64_t d, const int64_t* ptype,
                                const int32_t* order, const int64_t* stats, void* sim, int* l0, float thr,
                                hipStream_t st) {
    const int64_t row_bytes = d * Act<DT>::kBytes;
    const int64_t per_block = (int64_t)(kSimThreads / kWave) * kPairs;
    const int64_t blocks = (L + per_block - 1) / per_block;
    hipLaunchKernelGGL((k_pair_similarity<DT, kPairs, kSimThreads>), dim3((unsigned)blocks), dim3(kSimThreads), 0, st,
                       (const char*)hidden, (uint32_t)row_bytes, ptype, order, stats, sim, l0, thr);
    return (int)hipGetLastError();
}

template <int DT, int kPairs>
static int launch_similarity_p(const void* hidden, int64_t L, int64_t d, const int64_t* ptype,
                               const int32_t* order, const int64_t* stats, void* sim, int* l0, float thr,
                               hipStream_t st) {
    static int waves = 0;
    if (!waves) { const char* e = getenv("FF_SIM_WAVES"); waves = e ? atoi(e) : 4; }
    if (waves == 8) return launch_similarity_pt<DT, kPairs, 512>(hidden, L, d, ptype, order, stats, sim, l0, thr, st);
    if (waves == 2) return launch_similarity_pt<DT, kPairs, 128>(hidden, L, d, ptype, order, stats, sim, l0, thr, st);
    return launch_similarity_pt<DT, kPairs, 256>(hidden, L, d, ptype, order, stats, sim, l0, thr, st);
}

static int tune_pairs() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FF_SIM_PAIRS");
        v = e ? atoi(e) : 4;
    }
    return v;
}

template <int DT>
static int launch_similarity(const void* hidden, int64_t L, int64_t d, const int64_t* ptype,
                             const int32_t* order, const int64_t* stats, void* sim, int* l0, float thr,
                             hipStream_t st) {
    switch (tune_pairs()) {
        case 2: return launch_similarity_p<DT, 2>(hidden, L, d, ptype, order, stats, sim, l0, thr, st);
        case 8: return launch_similarity_p<DT, 8>(hidden, L, d, ptype, order, stats, sim, l0, thr, st);
        default: return launch_similarity_p<DT, 4>(hidden, L, d, ptype, order, stats, sim, l0, thr, st);
    }
}

int launch_similarity_any(const void* hidden, int dtype, int64_t L, int64_t d, const int64_t* ptype,
                          const int32_t* order, const int64_t* stats, void* sim, int* l0, double thr,
                          hipStream_t st) {
    switch (dtype) {
        case FF_F32: return launch_similarity<FF_F32>(hidden, L, d, ptype, order, stats, sim, l0, (float)thr, st);
        case FF_BF16: return launch_similarity<FF_BF16>(hidden, L, d, ptype, order, stats, sim, l0, (float)thr, st);
        default: return launch_similarity<FF_F16>(hidden, L, d, ptype, order, stats, sim, l0, (float)thr, st);
    }
}

}  // namespace ff

extern "C" int ff_pair_similarity(const void* hidden, int dtype, int64_t L, int64_t d,
                                  const int64_t* patch_type, const int32_t* order, const int64_t* stats,
                                  void* sim, ff_stream_t stream) {
    if (!hidden || !patch_type || !order || !stats || !sim || L < 0 || d < 1) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((d * esz) & 15)) return FF_ERR_ALIGN;
    if (L >= (1ll << 31) || d * esz >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (L == 0) return FF_OK;
    return ff::launch_similarity_any(hidden, dtype, L, d, patch_type, order, stats, sim, nullptr, 0.0,
                                     (hipStream_t)stream);
}
