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
#include "ff_common.h"

namespace ff {

constexpr int kSimThreads = 256;
constexpr int kSimWaves = kSimThreads / kWave;

template <int DT, int kPairs>
__global__ __launch_bounds__(kSimThreads) void k_pair_similarity(
    const char* __restrict__ hidden, int64_t row_bytes, const int64_t* __restrict__ ptype,
    const int32_t* __restrict__ order, const int64_t* __restrict__ stats, void* __restrict__ sim) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    const int nv = (int)stats[FF_STAT_NV];
    const int lane = lane_id();
    const int j0 = (blockIdx.x * kSimWaves + wave_id()) * kPairs;
    if (j0 >= nv) return;

    const char* row[kPairs + 1];
#pragma unroll
    for (int r = 0; r <= kPairs; ++r) {
        int j = j0 - 1 + r;
        j = j < 0 ? 0 : (j >= nv ? nv - 1 : j);
        row[r] = hidden + (int64_t)order[j] * row_bytes + lane * 16;
    }

    float nrm[kPairs + 1], dot[kPairs];
#pragma unroll
    for (int r = 0; r <= kPairs; ++r) nrm[r] = 0.f;
#pragma unroll
    for (int r = 0; r < kPairs; ++r) dot[r] = 0.f;

    const int64_t lane_off = (int64_t)lane * 16;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 cur[kPairs + 1], nxt[kPairs + 1];
#pragma unroll
    for (int r = 0; r <= kPairs; ++r)
        cur[r] = (lane_off < row_bytes) ? *(const uint4*)(row[r]) : zero;

    for (int64_t off = 0; off < row_bytes; off += 1024) {
        const int64_t noff = off + 1024;
        const bool more = (noff + lane_off) < row_bytes;
#pragma unroll
        for (int r = 0; r <= kPairs; ++r)
            nxt[r] = more ? *(const uint4*)(row[r] + noff) : zero;

        float prev[E], x[E];
        A::unpack(cur[0], prev);
#pragma unroll
        for (int e = 0; e < E; ++e) nrm[0] = __builtin_fmaf(prev[e], prev[e], nrm[0]);
#pragma unroll
        for (int r = 1; r <= kPairs; ++r) {
            A::unpack(cur[r], x);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                nrm[r] = __builtin_fmaf(x[e], x[e], nrm[r]);
                dot[r - 1] += A::rnd(prev[e] * x[e]);
                prev[e] = x[e];
            }
        }
#pragma unroll
        for (int r = 0; r <= kPairs; ++r) cur[r] = nxt[r];
    }

#pragma unroll
    for (int r = 0; r <= kPairs; ++r) nrm[r] = wave_sum(nrm[r]);
#pragma unroll
    for (int r = 0; r < kPairs; ++r) dot[r] = wave_sum(dot[r]);

#pragma unroll
    for (int r = 0; r < kPairs; ++r) {
        if (lane == r) {
            const int j = j0 + r;
            if (j < nv) {
                float s = -2.0f;   // IGNORE_TOKEN (main.py:225-238)
                if (j > 0 && ptype[order[j - 1]] == ptype[order[j]]) {
                    const float d = A::rnd(dot[r]);
                    const float na = A::rnd(__fsqrt_rn(nrm[r]));
                    const float nb = A::rnd(__fsqrt_rn(nrm[r + 1]));
                    const float den = A::rnd(na * nb);
                    s = A::rnd(__fdiv_rn(d, den));
                }
                A::store1(sim, j, s);
            }
        }
    }
}

template <int DT>
static int launch_similarity(const void* hidden, int64_t L, int64_t d, const int64_t* ptype,
                             const int32_t* order, const int64_t* stats, void* sim, hipStream_t st) {
    constexpr int kPairs = 4;
    const int64_t row_bytes = d * Act<DT>::kBytes;
    const int64_t per_block = (int64_t)kSimWaves * kPairs;
    const int64_t blocks = (L + per_block - 1) / per_block;
    hipLaunchKernelGGL((k_pair_similarity<DT, kPairs>), dim3((unsigned)blocks), dim3(kSimThreads), 0, st,
                       (const char*)hidden, row_bytes, ptype, order, stats, sim);
    return (int)hipGetLastError();
}

}  // namespace ff

extern "C" int ff_pair_similarity(const void* hidden, int dtype, int64_t L, int64_t d,
                                  const int64_t* patch_type, const int32_t* order, const int64_t* stats,
                                  void* sim, ff_stream_t stream) {
    if (!hidden || !patch_type || !order || !stats || !sim || L < 0 || d < 1) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((d * esz) & 15)) return FF_ERR_ALIGN;
    if (L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (L == 0) return FF_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FF_F32: return ff::launch_similarity<FF_F32>(hidden, L, d, patch_type, order, stats, sim, st);
        case FF_BF16: return ff::launch_similarity<FF_BF16>(hidden, L, d, patch_type, order, stats, sim, st);
        default: return ff::launch_similarity<FF_F16>(hidden, L, d, patch_type, order, stats, sim, st);
    }
}
