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

// Frame-major layout the host expects from FrameFusion.prepare()'s scalars: `frames` frames of
// `patches` visual tokens behind `pre` other tokens, every frame typed 0..patches-1, everything else
// TEXT.  With a hint the similarity kernel computes the by-patch order in closed form
// (order[p*F + f] = pre + f*P + p) instead of waiting for the order kernels, WRITES `order` and the
// stats words K0 would have written, and verifies the hint on the way: every position is compared
// with its expected type by exactly one lane.  A mismatch sets bit FF_ERR_BIT_LAYOUT of
// stats[FF_STAT_ERROR]; the host then repeats the call through K0.
struct LayoutHint {
    int pre, patches, frames, L;
};

// kAdd: the rows are T(hidden[i] + addend[i]) - the decoder's residual add fused into the pass (both
// operands read once here and once by the merge kernel, the sum never touches memory).
// Measured and not shipped (profiles/EXPERIMENTS.md 4.8, 4.13): 8 waves per SIMD instead of 7 (66 VGPRs / 85 SGPRs here;
// amdgpu_waves_per_eu(8, 8) + amdgpu_num_sgpr(80) = 4 spills): 68.8 vs 59.5 us; the overlap row handed from wave to wave
// through LDS (two 1 KiB buffers per wave, one barrier per tile) instead of being fetched again: 60.7 vs 59.8 us.
template <int DT, int kPairs, int kSimThreads, bool kHint, bool kAdd>
__global__ __launch_bounds__(kSimThreads) void k_pair_similarity(
    const char* __restrict__ hidden, const char* __restrict__ addend, uint32_t row_bytes, const int64_t* __restrict__ ptype,
    const int32_t* __restrict__ order, const int64_t* __restrict__ stats, void* __restrict__ sim,
    int* __restrict__ l0, int* t16_end, float thr, const LayoutHint hint, int32_t* __restrict__ order_out,
    int32_t* __restrict__ inv_out, int64_t* __restrict__ stats_out) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    constexpr int R = kPairs + 1;
    constexpr int kSimWaves = kSimThreads / kWave;
    const int lane = lane_id();
    int nv;
    if constexpr (kHint) {
        nv = hint.patches * hint.frames;
        // the non-visual tail of `order` (positions in sequence order) + their type check
        const int n_tail = hint.L - nv;
        const int gtid = blockIdx.x * kSimThreads + threadIdx.x;
        for (int q = gtid; q < n_tail; q += (int)gridDim.x * kSimThreads) {
            const int i = q < hint.pre ? q : q + nv;
            if (ptype[i] != -1) atomicOr((unsigned long long*)(stats_out + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_LAYOUT);
            order_out[nv + q] = i;
            if (inv_out) inv_out[i] = nv + q;
        }
        if (gtid == 0) {
            stats_out[FF_STAT_NV] = nv;
            stats_out[FF_STAT_FTN] = nv;
        }
    } else {
        nv = (int)stats[FF_STAT_NV];
    }
    const int j0 = uniform((blockIdx.x * kSimWaves + wave_id()) * kPairs);
    if (j0 >= nv) return;

    __amdgpu_buffer_rsrc_t row[R], row2[R];
    auto set_row = [&](int r, int64_t i) {
        row[r] = make_rsrc(hidden + i * row_bytes, row_bytes);
        row2[r] = make_rsrc((kAdd ? addend : hidden) + i * row_bytes, row_bytes);
    };
    int slot_i[kPairs], slot_p[kPairs], slot_f[kPairs];       // kHint: position / type / frame of slot j0+r
    int64_t my_type = 0;
    if constexpr (kHint) {
        const int F = hint.frames, P = hint.patches;
        const int p0 = uniform(j0 / F), f0 = j0 - p0 * F;
        {   // row 0 = slot j0-1 (slot 0 again when j0 == 0)
            int pm = p0, fm = f0 - 1;
            if (fm < 0) { fm = F - 1; pm = p0 - 1; }
            if (j0 == 0) { pm = 0; fm = 0; }
            set_row(0, (int64_t)(hint.pre + fm * P + pm));
        }
        int p = p0, f = f0, last_i = hint.pre + f0 * P + p0;
#pragma unroll
        for (int r = 0; r < kPairs; ++r) {
            const bool in = j0 + r < nv;
            const int i = in ? hint.pre + f * P + p : last_i;
            slot_i[r] = i; slot_p[r] = p; slot_f[r] = f;
            last_i = i;
            set_row(r + 1, (int64_t)i);
            if (++f == F) { f = 0; ++p; }
        }
        // type of my slot, requested now so that the check at the end finds it in a register
        int my_i = slot_i[0];
#pragma unroll
        for (int r = 1; r < kPairs; ++r) my_i = lane == r ? slot_i[r] : my_i;
        if (lane < kPairs) my_type = ptype[my_i];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int j = j0 - 1 + r;
            j = j < 0 ? 0 : (j >= nv ? nv - 1 : j);
            set_row(r, (int64_t)uniform(order[j]));
        }
    }

    float nrm[R], dot[kPairs];
#pragma unroll
    for (int r = 0; r < R; ++r) nrm[r] = 0.f;
#pragma unroll
    for (int r = 0; r < kPairs; ++r) dot[r] = 0.f;

    // 1 KiB tiles; tile t+1 is in flight while tile t is reduced
    const uint32_t lane_off = (uint32_t)lane * 16;
    const uint32_t tiles = (row_bytes + 1023u) >> 10;
    uint4 cur[R], nxt[R], nxt2[kAdd ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        cur[r] = buf_load16(row[r], lane_off);
        if constexpr (kAdd) cur[r] = add16<DT>(cur[r], buf_load16(row2[r], lane_off));
    }
    for (uint32_t t = 0; t < tiles; ++t) {
        const uint32_t noff = lane_off + (t + 1) * 1024u;      // past-the-end lanes read zeros
#pragma unroll
        for (int r = 0; r < R; ++r) {
            nxt[r] = buf_load16(row[r], noff);
            if constexpr (kAdd) nxt2[r] = buf_load16(row2[r], noff);
        }

        if constexpr (DT == FF_BF16) {
            // |x|^2 straight from the packed words; the T-rounded products from the unpacked rows
            float prev[E], x[E];
#pragma unroll
            for (int r = 0; r < R; ++r) nrm[r] = A::sumsq(cur[r], nrm[r]);
            A::unpack(cur[0], prev);
#pragma unroll
            for (int r = 1; r < R; ++r) {
                A::unpack(cur[r], x);
                dot[r - 1] = A::dot_rounded(prev, x, dot[r - 1]);
#pragma unroll
                for (int e = 0; e < E; ++e) prev[e] = x[e];
            }
        } else if constexpr (DT == FF_F16) {
#pragma unroll
            for (int r = 0; r < R; ++r) nrm[r] = A::sumsq(cur[r], nrm[r]);
#pragma unroll
            for (int r = 1; r < R; ++r) dot[r - 1] = A::dot_rounded_raw(cur[r - 1], cur[r], dot[r - 1]);
        } else {
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
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (kAdd) cur[r] = add16<DT>(nxt[r], nxt2[r]);
            else cur[r] = nxt[r];
        }
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
                bool same_type;
                if constexpr (kHint) {
                    if (my_type != (int64_t)slot_p[r])
                        atomicOr((unsigned long long*)(stats_out + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_LAYOUT);
                    order_out[j] = slot_i[r];
                    if (inv_out) inv_out[slot_i[r]] = j;
                    same_type = slot_f[r] != 0;            // the previous slot is the same patch one frame earlier
                } else {
                    same_type = j > 0 && ptype[order[j - 1]] == ptype[order[j]];
                }
                if (same_type) {
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
        // select statistics of the plan kernel that follows (ff_plan.hip): the workgroup's similarities
        // meet in LDS and ONE wave folds them - the top byte of the order-preserving key + count(sim >= thr)
        // into one of kL0Copies level-0 tables, the top 16 bits into the level-1 table of the group's slice;
        // equal values folded into one non-returning atomic each (a video's similarities are a dozen
        // distinct values: per-wave folding left ~4x the atomics on the same few memory-side words)
        __shared__ float sims[kSimWaves * kPairs];
        if (lane < kPairs) sims[wave_id() * kPairs + lane] = have ? mine : __int_as_float(0x7fc00001);   // (a NaN payload no similarity has: "absent")
        __syncthreads();
        // (waves past the end of the order have exited: the barrier counts the live ones; their entries stay
        // unwritten, so the folding wave only trusts entries of slots below nv)
        const int g0 = (blockIdx.x * kSimWaves) * kPairs;            // first slot of the workgroup
        if (j0 == g0) {                                            // the workgroup's first wave is always live
            const bool got = lane < kSimWaves * kPairs && g0 + lane < nv;
            const float sv = got ? sims[lane] : -2.0f;
            uint32_t bits;
            if constexpr (DT == FF_F32) bits = __float_as_uint(sv);
            else if constexpr (DT == FF_BF16) bits = __float_as_uint(sv) >> 16;
            else { _Float16 h = (_Float16)sv; bits = (uint32_t)__builtin_bit_cast(uint16_t, h); }
            const uint32_t key = order_key<DT>(bits);
            int* tab = l0 + (blockIdx.x & (kL0Copies - 1)) * kL0Stride;
            const int n_ge = __popcll(__ballot(got && sv >= thr));
            if (lane == 0 && n_ge) atomicAdd(&tab[256], n_ge);
            wave_agg_add<3>(tab, key >> (A::kKeyBits - 8), got);
            int* t16 = t16_slice(t16_end, g0 / kSelSlice) + (blockIdx.x & (kT16Copies - 1)) * 65536;
            wave_agg_add<6>(t16, t16_bin(key >> (A::kKeyBits - 16)), got);
        }
    }
}

struct SimArgs {
    const void* hidden;
    const void* addend;       // NULL: the rows are hidden's
    int64_t L, d;
    const int64_t* ptype;
    const int32_t* order;
    const int64_t* stats;
    void* sim;
    int* l0;
    int* t16_end;
    float thr;
    LayoutHint hint;          // frames == 0: no hint
    int32_t* order_out;
    int32_t* inv_out;
    int64_t* stats_out;
};

template <int DT, int kPairs, int kSimThreads>
static int launch_similarity_pt(const SimArgs& a, hipStream_t st) {
    const int64_t row_bytes = a.d * Act<DT>::kBytes;
    const int64_t per_block = (int64_t)(kSimThreads / kWave) * kPairs;
    const int64_t blocks = (a.L + per_block - 1) / per_block;
#define FF_SIM_LAUNCH(HINT, ADD, OO, IO, SO)                                                                              \
    hipLaunchKernelGGL((k_pair_similarity<DT, kPairs, kSimThreads, HINT, ADD>), dim3((unsigned)blocks), dim3(kSimThreads), 0, \
                       st, (const char*)a.hidden, (const char*)a.addend, (uint32_t)row_bytes, a.ptype, a.order, a.stats,    \
                       a.sim, a.l0, a.t16_end, a.thr, a.hint, OO, IO, SO)
    if (a.hint.frames > 0) {
        if (a.addend) FF_SIM_LAUNCH(true, true, a.order_out, a.inv_out, a.stats_out);
        else FF_SIM_LAUNCH(true, false, a.order_out, a.inv_out, a.stats_out);
    } else {
        if (a.addend) FF_SIM_LAUNCH(false, true, (int32_t*)nullptr, (int32_t*)nullptr, (int64_t*)nullptr);
        else FF_SIM_LAUNCH(false, false, (int32_t*)nullptr, (int32_t*)nullptr, (int64_t*)nullptr);
    }
#undef FF_SIM_LAUNCH
    return (int)hipGetLastError();
}

template <int DT>
static int launch_similarity(const SimArgs& a, hipStream_t st) { return launch_similarity_pt<DT, 4, 256>(a, st); }

// hint_frames > 0: frame-major closed form (see LayoutHint); `order` and `stats` are then outputs.
int launch_similarity_any(const void* hidden, const void* addend, int dtype, int64_t L, int64_t d, const int64_t* ptype,
                          int32_t* order, int32_t* inv, int64_t* stats, void* sim, int* l0, int* t16_end, double thr,
                          int64_t hint_pre, int64_t hint_patches, int64_t hint_frames, hipStream_t st) {
    SimArgs a{hidden, addend, L, d, ptype, order, stats, sim, l0, t16_end, (float)thr,
              LayoutHint{(int)hint_pre, (int)hint_patches, (int)hint_frames, (int)L}, order, inv, stats};
    switch (dtype) {
        case FF_F32: return launch_similarity<FF_F32>(a, st);
        case FF_BF16: return launch_similarity<FF_BF16>(a, st);
        default: return launch_similarity<FF_F16>(a, st);
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
    return ff::launch_similarity_any(hidden, nullptr, dtype, L, d, patch_type, const_cast<int32_t*>(order), nullptr,
                                     const_cast<int64_t*>(stats), sim, nullptr, nullptr, 0.0, 0, 0, 0, (hipStream_t)stream);
}
