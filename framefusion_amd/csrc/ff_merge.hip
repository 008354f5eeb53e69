// K4 - run merge + compaction, the second (and last) streaming pass over the activations.
// Replaces index_add_ + divide (framefusion/main.py:304-317) and the bool-mask gathers of
// hidden_states, cos/sin (or position ids) and patch_type (main.py:132-138, 161-178), which the
// reference runs as 6+ kernels with host syncs, by one launch that reads every surviving or
// folded row exactly once and writes every output row exactly once.
//
// Work decomposition: by INPUT rows, not output rows - a wave owns kSlots consecutive slots t of
// the by-patch order (then the non-visual tail) for one 4 KiB column block.  Slots whose run_len
// is -1 are members: they are read by the wave that owns their anchor.  Because every input row
// is read by exactly one wave per column block and slots are spread evenly, the HBM read load is
// balanced no matter how long individual runs are (an output-row decomposition would leave the
// longest run as the tail).
//
// Arithmetic: the anchor accumulates its members in by-patch
// order with a rounding to the activation dtype T after EVERY add, then one rounded divide by
// T(n+1) - the order CPU index_add_ applies; fp32-accumulate-then-round differs on ~23 % of
// elements by more than 1e-3 relative.
#include "ff_common.h"

namespace ff {

constexpr int kMergeThreads = 256;
constexpr int kMergeWaves = kMergeThreads / kWave;
constexpr int kSlots = 8;     // consecutive by-patch slots per workgroup
constexpr int kChunks = 2;    // 1 KiB chunks per column block: a wave moves 2 KiB of a row at a time

struct AuxPack {
    ff_aux_t a[FF_MAX_AUX];
    int n;
};

// Copy `bytes` (multiple of 2) from src to dst with the widest unit the alignment allows,
// spread over the threads [tid, nthreads).
__device__ inline void copy_row(const char* __restrict__ src, char* __restrict__ dst, int64_t bytes,
                                int tid, int nthreads) {
    const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)bytes;
    if ((al & 15) == 0) {
        for (int64_t o = (int64_t)tid * 16; o < bytes; o += (int64_t)nthreads * 16)
            *(uint4*)(dst + o) = *(const uint4*)(src + o);
    } else if ((al & 7) == 0) {
        for (int64_t o = (int64_t)tid * 8; o < bytes; o += (int64_t)nthreads * 8)
            *(uint2*)(dst + o) = *(const uint2*)(src + o);
    } else if ((al & 3) == 0) {
        for (int64_t o = (int64_t)tid * 4; o < bytes; o += (int64_t)nthreads * 4)
            *(uint32_t*)(dst + o) = *(const uint32_t*)(src + o);
    } else {
        for (int64_t o = (int64_t)tid * 2; o < bytes; o += (int64_t)nthreads * 2)
            *(uint16_t*)(dst + o) = *(const uint16_t*)(src + o);
    }
}

// Waves are independent (no LDS, no barrier): the 4 waves of a workgroup own the same kSlots
// consecutive by-patch slots and one 2 KiB column block each, so together they read whole 8 KiB
// rows.  One coalesced load of order[t0 .. t0+63] / run_len[...] gives a wave the anchors AND
// the member rows that follow them (members are simply the next slots), so the only dependent
// index fetch is dst[] for the anchors.  Row pieces move as raw buffer loads/stores (lanes past
// the row end read 0 / are dropped); member pieces are prefetched two ahead and the next slot's
// anchor piece is requested before the current slot's members are folded.
template <int DT>
__global__ __launch_bounds__(kMergeThreads) void k_merge_compact(
    const char* __restrict__ hidden, char* __restrict__ out, uint32_t row_bytes, int L, int64_t L_cap,
    const int32_t* __restrict__ order, const int32_t* __restrict__ run_len, const int32_t* __restrict__ dst,
    AuxPack aux) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    constexpr uint32_t kBlk = kChunks * 1024u;
    const int lane = lane_id();
    const int t0 = blockIdx.x * kSlots;
    const int cb = uniform(blockIdx.y * kMergeWaves + wave_id());
    const uint32_t col = (uint32_t)cb * kBlk;
    if (col >= row_bytes) return;
    const uint32_t blk_bytes = min(kBlk, row_bytes - col);
    const uint32_t voff = (uint32_t)lane * 16;

    const int tl = t0 + lane;
    const int rl = tl < L ? run_len[tl] : -1;
    const int ord = tl < L ? (order ? order[tl] : tl) : 0;
    const int dv = (lane < kSlots && rl >= 0) ? dst[ord] : 0;

    auto row_of = [&](int t_abs) -> int {     // sequence index of by-patch slot t_abs (wave-uniform)
        const int rel = t_abs - t0;
        return rel < kWave ? __builtin_amdgcn_readlane(ord, rel) : uniform(order[t_abs]);
    };
    auto piece = [&](int i) { return make_rsrc(hidden + (int64_t)i * row_bytes + col, blk_bytes); };

    uint4 a[kChunks];
    int q = 0;
    // first live slot
    while (q < kSlots && __builtin_amdgcn_readlane(rl, q) < 0) ++q;
    if (q < kSlots) {
        const __amdgpu_buffer_rsrc_t src = piece(__builtin_amdgcn_readlane(ord, q));
#pragma unroll
        for (int c = 0; c < kChunks; ++c) a[c] = buf_load16(src, voff + c * 1024u);
    }
    while (q < kSlots) {
        const int n = __builtin_amdgcn_readlane(rl, q);
        const int i = __builtin_amdgcn_readlane(ord, q);
        const int r = __builtin_amdgcn_readlane(dv, q);
        int qn = q + 1;
        while (qn < kSlots && __builtin_amdgcn_readlane(rl, qn) < 0) ++qn;
        uint4 an[kChunks];
        if (qn < kSlots) {      // request the next slot's anchor piece now
            const __amdgpu_buffer_rsrc_t src = piece(__builtin_amdgcn_readlane(ord, qn));
#pragma unroll
            for (int c = 0; c < kChunks; ++c) an[c] = buf_load16(src, voff + c * 1024u);
        }
        if (n > 0) {
            const int t = t0 + q;
            float acc[kChunks][E];
            uint4 n0[kChunks], n1[kChunks];
            {
                const __amdgpu_buffer_rsrc_t m0 = piece(row_of(t + 1));
#pragma unroll
                for (int c = 0; c < kChunks; ++c) n0[c] = buf_load16(m0, voff + c * 1024u);
            }
            if (n > 1) {
                const __amdgpu_buffer_rsrc_t m1 = piece(row_of(t + 2));
#pragma unroll
                for (int c = 0; c < kChunks; ++c) n1[c] = buf_load16(m1, voff + c * 1024u);
            }
#pragma unroll
            for (int c = 0; c < kChunks; ++c) A::unpack(a[c], acc[c]);
            for (int m = 1; m <= n; ++m) {
                uint4 cur[kChunks];
#pragma unroll
                for (int c = 0; c < kChunks; ++c) { cur[c] = n0[c]; n0[c] = n1[c]; }
                if (m + 2 <= n) {
                    const __amdgpu_buffer_rsrc_t mr = piece(row_of(t + m + 2));
#pragma unroll
                    for (int c = 0; c < kChunks; ++c) n1[c] = buf_load16(mr, voff + c * 1024u);
                }
#pragma unroll
                for (int c = 0; c < kChunks; ++c) {
                    float x[E];
                    A::unpack(cur[c], x);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[c][e] = A::rnd(acc[c][e] + x[e]);
                }
            }
            const float div = A::rnd((float)(n + 1));
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
#pragma unroll
                for (int e = 0; e < E; ++e) acc[c][e] = A::rnd(acc[c][e] / div);
                a[c] = A::pack(acc[c]);
            }
        }
        const __amdgpu_buffer_rsrc_t dstr = make_rsrc(out + (int64_t)r * row_bytes + col, blk_bytes);
#pragma unroll
        for (int c = 0; c < kChunks; ++c) buf_store16(dstr, voff + c * 1024u, a[c]);
        // position embeddings / patch types / position ids ride along: tiny rows, same mapping
        if (cb == 0) {
            for (int x = 0; x < aux.n; ++x) {
                const ff_aux_t& ax = aux.a[x];
                for (int64_t o = 0; o < ax.outer; ++o)
                    copy_row((const char*)ax.src + (o * L + i) * ax.row_bytes,
                             (char*)ax.dst + (o * L_cap + r) * ax.row_bytes, ax.row_bytes, lane, kWave);
            }
        }
#pragma unroll
        for (int c = 0; c < kChunks; ++c) a[c] = an[c];
        q = qn;
    }
}

// out[r, c] = mask[i_r, i_c]: one workgroup per kept input row, threads over input columns.
__global__ __launch_bounds__(256) void k_gather_mask(const char* __restrict__ mask, char* __restrict__ out,
                                                     int elem_bytes, int L, int64_t L_cap,
                                                     const int32_t* __restrict__ dst) {
    const int i = blockIdx.x;
    const int r = dst[i];
    if (r < 0) return;
    for (int c = threadIdx.x; c < L; c += blockDim.x) {
        const int rc = dst[c];
        if (rc < 0) continue;
        const char* s = mask + ((int64_t)i * L + c) * elem_bytes;
        char* d = out + ((int64_t)r * L_cap + rc) * elem_bytes;
        if (elem_bytes == 2) *(uint16_t*)d = *(const uint16_t*)s;
        else if (elem_bytes == 4) *(uint32_t*)d = *(const uint32_t*)s;
        else if (elem_bytes == 1) *d = *s;
        else *(uint64_t*)d = *(const uint64_t*)s;
    }
}

// importance[s] = T(mean over H*num of attn_w[h, n, s]) accumulated in fp32 (main.py:70).
template <int DT>
__global__ __launch_bounds__(256) void k_head_mean(const void* __restrict__ w, int rows, int S,
                                                   void* __restrict__ imp) {
    using A = Act<DT>;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += A::load1(w, (int64_t)r * S + s);
    A::store1(imp, s, acc / (float)rows);
}

int launch_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                         const int32_t* order, const int32_t* run_len, const int32_t* dst,
                         const ff_aux_t* aux_host, int n_aux, hipStream_t st) {
    AuxPack pack;
    pack.n = n_aux;
    for (int x = 0; x < n_aux; ++x) pack.a[x] = aux_host[x];
    for (int x = n_aux; x < FF_MAX_AUX; ++x) pack.a[x] = ff_aux_t{nullptr, nullptr, 0, 0};
    const int64_t row_bytes = d * (dtype == FF_F32 ? 4 : 2);
    const int nblk = (int)((row_bytes + kChunks * 1024 - 1) / (kChunks * 1024));
    const dim3 blocks((unsigned)((L + kSlots - 1) / kSlots), (unsigned)((nblk + kMergeWaves - 1) / kMergeWaves));
    const char* h = (const char*)hidden;
    char* o = (char*)hidden_out;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_merge_compact<FF_F32>, blocks, dim3(kMergeThreads), 0, st, h, o,
                               (uint32_t)row_bytes, (int)L, L_cap, order, run_len, dst, pack);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_merge_compact<FF_BF16>, blocks, dim3(kMergeThreads), 0, st, h, o,
                               (uint32_t)row_bytes, (int)L, L_cap, order, run_len, dst, pack);
            break;
        default:
            hipLaunchKernelGGL(k_merge_compact<FF_F16>, blocks, dim3(kMergeThreads), 0, st, h, o,
                               (uint32_t)row_bytes, (int)L, L_cap, order, run_len, dst, pack);
    }
    return (int)hipGetLastError();
}

}  // namespace ff

extern "C" int ff_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d,
                                int64_t L_cap, const int32_t* order, const int32_t* run_len, const int32_t* dst,
                                const ff_aux_t* aux_host, int n_aux, ff_stream_t stream) {
    if (!hidden || !hidden_out || !run_len || !dst || L < 0 || d < 1 || L_cap < 0) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (n_aux < 0 || n_aux > FF_MAX_AUX || (n_aux > 0 && !aux_host)) return FF_ERR_ARG;
    for (int x = 0; x < n_aux; ++x)
        if (!aux_host[x].src || !aux_host[x].dst || aux_host[x].row_bytes < 2 || (aux_host[x].row_bytes & 1) ||
            aux_host[x].outer < 1)
            return FF_ERR_ARG;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((uintptr_t)hidden & 15) || ((uintptr_t)hidden_out & 15) || ((d * esz) & 15)) return FF_ERR_ALIGN;
    if (L >= (1ll << 29) || d * esz >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (L == 0) return FF_OK;
    return ff::launch_merge_compact(hidden, hidden_out, dtype, L, d, L_cap, order, run_len, dst, aux_host, n_aux,
                                    (hipStream_t)stream);
}

extern "C" int ff_gather_mask(const void* mask, void* out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                              const int32_t* dst, ff_stream_t stream) {
    if (!mask || !out || !dst || L < 0 || L_cap < 0) return FF_ERR_ARG;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return FF_ERR_ARG;
    if (L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (L == 0) return FF_OK;
    hipLaunchKernelGGL(ff::k_gather_mask, dim3((unsigned)L), dim3(256), 0, (hipStream_t)stream, (const char*)mask,
                       (char*)out, (int)elem_bytes, (int)L, L_cap, dst);
    return (int)hipGetLastError();
}

extern "C" int ff_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S, void* importance,
                            ff_stream_t stream) {
    if (!attn_w || !importance || H < 1 || num < 1 || S < 0) return FF_ERR_ARG;
    if (S >= (1ll << 31) || H * num >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (S == 0) return FF_OK;
    const unsigned blocks = (unsigned)((S + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(ff::k_head_mean<FF_F32>, dim3(blocks), dim3(256), 0, st, attn_w, (int)(H * num), (int)S, importance);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(ff::k_head_mean<FF_BF16>, dim3(blocks), dim3(256), 0, st, attn_w, (int)(H * num), (int)S, importance);
            break;
        case FF_F16:
            hipLaunchKernelGGL(ff::k_head_mean<FF_F16>, dim3(blocks), dim3(256), 0, st, attn_w, (int)(H * num), (int)S, importance);
            break;
        default:
            return FF_ERR_ARG;
    }
    return (int)hipGetLastError();
}
