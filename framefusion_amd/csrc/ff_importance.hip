// K5 - attention probabilities of the last `num` queries and their head mean: the only extra work
// FrameFusion adds inside the attention module (framefusion/utils.py:27-57, called from
// models/qwen2/modeling_qwen2.py:166-178 with num=1 and modeling_qwen2_vl.py:292-300 with num=4),
// feeding the prune of main.py:69-92.
//
// The reference repeats K to all H heads first (repeat_kv, modeling_qwen2.py:147) and materialises
// a [1, H, num, S] tensor; here a key row of ONE kv head is scored against the query heads of its
// GQA group, so K traffic is ~H_kv*S*dh*sizeof(T) (11 MB at S = 11k, H_kv = 4, dh = 128),
// negligible next to the activation passes.
//
// Staged rounding, T = activation dtype:
//   s = T(sum_fp32 q_i*k_i);  s = T(fp32(s) * fp32(scale));  s = T(s + bias);  p = T(exp(s - max) / sum)
#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "ff_common.h"

namespace ff {

// scores[h, n, s] as float holding T values.  grid: (ceil(S/256), H_kv, row groups); one LANE per
// key: the lane streams its own key row (dh * sizeof(T) bytes, 16 B at a time) and dots it with up
// to kRowsPerBlock query rows of the GQA group that sit in LDS as fp32 (wave-uniform broadcast
// reads) - no cross-lane reduction, stores coalesced over s.
constexpr int kRowsPerBlock = 8;

struct KStrides { int64_t head, key; };          // bytes from one kv head / one key to the next

template <int DT>
__global__ __launch_bounds__(256) void k_lq_scores(const void* __restrict__ q, const void* __restrict__ k,
                                                   int64_t k_head_stride, int64_t k_key_stride, int H, int H_kv, int num, int S, int dh, float scale,
                                                   int causal, const void* __restrict__ bias, float* __restrict__ scores) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    extern __shared__ __attribute__((aligned(16))) float q_lds[];   // [rows_here][dh]
    const int hk = blockIdx.y;
    const int group = H / H_kv;
    const int rows = group * num;
    const int r0 = blockIdx.z * kRowsPerBlock;
    const int rows_here = min(kRowsPerBlock, rows - r0);
    for (int x = threadIdx.x; x < rows_here * dh; x += blockDim.x) {
        const int r = r0 + x / dh, e = x % dh;
        const int h = hk * group + r / num, n = r % num;
        q_lds[x] = A::load1(q, ((int64_t)h * num + n) * dh + e);
    }
    __syncthreads();
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const char* krow = (const char*)k + (int64_t)hk * k_head_stride + (int64_t)s * k_key_stride;
    float acc[kRowsPerBlock];
#pragma unroll
    for (int r = 0; r < kRowsPerBlock; ++r) acc[r] = 0.f;
    for (int c = 0; c < dh; c += E) {                 // dh % E == 0 (checked by the launcher)
        float kv[E];
        A::unpack(*(const uint4*)(krow + (size_t)c * A::kBytes), kv);
#pragma unroll
        for (int r = 0; r < kRowsPerBlock; ++r) {
            if (r < rows_here) {
#pragma unroll
                for (int e = 0; e < E; ++e) acc[r] = __builtin_fmaf(q_lds[r * dh + c + e], kv[e], acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRowsPerBlock; ++r) {
        if (r < rows_here) {
            const int rr = r0 + r;
            const int h = hk * group + rr / num, n = rr % num;
            float v = A::rnd(acc[r]);
            v = A::rnd(v * scale);
            if (causal && s > S - num + n) v = A::rnd(v + (-INFINITY));
            if (bias) v = A::rnd(v + A::load1(bias, (int64_t)n * S + s));
            scores[((int64_t)h * num + n) * S + s] = v;
        }
    }
}

// One workgroup per (h, n) row: softmax over S in fp32, rounded to T.
template <int DT>
__global__ __launch_bounds__(256) void k_lq_softmax(const float* __restrict__ scores, int S,
                                                    float* __restrict__ probs_f, void* __restrict__ weights) {
    using A = Act<DT>;
    __shared__ float red[4];
    const int64_t base = (int64_t)blockIdx.x * S;
    float m = -INFINITY;
    for (int s = threadIdx.x; s < S; s += blockDim.x) m = fmaxf(m, scores[base + s]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
    if (lane_id() == 0) red[wave_id()] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int s = threadIdx.x; s < S; s += blockDim.x) sum += expf(scores[base + s] - m);
    sum = wave_sum(sum);
    if (lane_id() == 0) red[wave_id()] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const float p = A::rnd(expf(scores[base + s] - m) / sum);
        probs_f[base + s] = p;
        if (weights) A::store1(weights, base + s, p);
    }
}

// Select tables of freshly computed importances (one value per lane, 64 consecutive positions per
// wave, so the slice is wave-uniform): what the plan kernel of the prune call needs (ff_plan.hip).
template <int DT>
__device__ inline void importance_tables(float value, int s, bool in_range, int* l0, int* t16_end) {
    using A = Act<DT>;
    uint32_t bits;
    if constexpr (DT == FF_F32) bits = __float_as_uint(value);
    else if constexpr (DT == FF_BF16) bits = __float_as_uint(A::rnd(value)) >> 16;
    else { _Float16 h = (_Float16)value; bits = (uint32_t)__builtin_bit_cast(uint16_t, h); }
    const uint32_t key = order_key<DT>(bits);
    wave_agg_add<3>(l0 + (blockIdx.x & (kL0Copies - 1)) * kL0Stride, key >> (A::kKeyBits - 8), in_range);
    int* t16 = t16_slice(t16_end, uniform(s) / kSelSlice) + (wave_id() & (kT16Copies - 1)) * 65536;
    wave_agg_add<2>(t16, t16_bin(key >> (A::kKeyBits - 16)), in_range);
}

// importance[s] = T(mean over H*num of attn_w[h, n, s]) accumulated in fp32 in row order (main.py:70); with
// l0 the select tables of the values in [lo, hi) are accumulated on the way.  A workgroup owns 8 sixteen-byte
// pieces of positions; its 256 threads load 32 rows of them at once (kHmDepth such passes in flight) and
// the first threads add the rows up through LDS.
constexpr int kHmDepth = 4;
template <int DT>
__global__ __launch_bounds__(256) void k_head_mean(const void* __restrict__ w, int rows, int S,
                                                   void* __restrict__ imp, int lo, int hi, int* __restrict__ l0,
                                                   int* t16_end) {
    using A = Act<DT>;
    constexpr int E = A::kPer16, P = 8 * E;          // positions per workgroup
    __shared__ float tile[32][P + 1];
    const int t = threadIdx.x, piece = t & 7, rip = t >> 3;
    const int s0 = blockIdx.x * P + piece * E;
    const bool vec = (S % E) == 0;                   // rows keep the 16-byte alignment of the base
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(w, (uint32_t)((int64_t)rows * S * A::kBytes));
    float acc = 0.f;
    for (int r0 = 0; r0 < rows; r0 += 32 * kHmDepth) {
        float x[kHmDepth][E];
#pragma unroll
        for (int u = 0; u < kHmDepth; ++u) {
            const int r = r0 + u * 32 + rip;
            if (vec) {
                // (past the last row or the last position the range check returns zeros)
                const uint4 v = buf_load16(rs, r < rows && s0 < S ? (uint32_t)(((int64_t)r * S + s0) * A::kBytes) : 0xfffffff0u);
                A::unpack(v, x[u]);
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) x[u][e] = r < rows && s0 + e < S ? A::load1(w, (int64_t)r * S + s0 + e) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < kHmDepth; ++u) {
            if (r0 + u * 32 >= rows) break;          // (uniform)
#pragma unroll
            for (int e = 0; e < E; ++e) tile[rip][piece * E + e] = x[u][e];
            __syncthreads();
            if (t < P) {
                const int n = min(32, rows - (r0 + u * 32));
                for (int r = 0; r < n; ++r) acc += tile[r][t];
            }
            __syncthreads();
        }
    }
    if (t >= 64) return;                             // (P <= 64: the first wave finishes)
    const int s = blockIdx.x * P + t;
    const bool ok = t < P && s < S;
    const float v = A::rnd(acc / (float)rows);
    if (ok) A::store1(imp, s, v);
    if (l0) importance_tables<DT>(v, s, ok && s >= lo && s < hi, l0, t16_end);
}

int launch_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S, void* importance,
                     int64_t lo, int64_t hi, int* l0, int* t16_end, hipStream_t st) {
    const int64_t per_block = 8 * (dtype == FF_F32 ? 4 : 8);
    const unsigned blocks = (unsigned)((S + per_block - 1) / per_block);
    if ((int64_t)H * num * S * (dtype == FF_F32 ? 4 : 2) >= (1ll << 32)) return FF_ERR_UNSUPPORTED;
    switch (dtype) {
        case FF_F32:
            hipLaunchKernelGGL(k_head_mean<FF_F32>, dim3(blocks), dim3(256), 0, st, attn_w, (int)(H * num), (int)S, importance, (int)lo, (int)hi, l0, t16_end);
            break;
        case FF_BF16:
            hipLaunchKernelGGL(k_head_mean<FF_BF16>, dim3(blocks), dim3(256), 0, st, attn_w, (int)(H * num), (int)S, importance, (int)lo, (int)hi, l0, t16_end);
            break;
        case FF_F16:
            hipLaunchKernelGGL(k_head_mean<FF_F16>, dim3(blocks), dim3(256), 0, st, attn_w, (int)(H * num), (int)S, importance, (int)lo, (int)hi, l0, t16_end);
            break;
        default:
            return FF_ERR_ARG;
    }
    return (int)hipGetLastError();
}

template <int DT>
__global__ __launch_bounds__(256) void k_lq_mean(const float* __restrict__ probs_f, int rows, int S,
                                                 void* __restrict__ imp, int lo, int hi, int* __restrict__ l0, int* t16_end) {
    using A = Act<DT>;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (s < S) {
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) acc += probs_f[(int64_t)r * S + s];
        v = A::rnd(acc / (float)rows);
        A::store1(imp, s, v);
    }
    if (l0) importance_tables<DT>(v, s, s >= lo && s < hi, l0, t16_end);
}

// ---- tiled path (lanes-per-key a power of two: every real head size) -------------------------------------
// k_lq_tile: workgroup = 256 keys of one kv head x up to 16 query rows of its GQA group.  A key row is
// read by LPK = dh * sizeof(T) / 16 consecutive lanes, 16 bytes each: a wave-load covers 1 KiB of
// CONTIGUOUS keys (the old lane-per-key layout strode 256 B between lanes).  The query rows sit in LDS
// as packed T; q . k runs on v_dot2c (products of two bf16 / fp16 are exact in fp32), the LPK partial
// sums meet through xor shuffles.  Scores leave as T in a key-major [S][H*num] workspace (2 bytes per
// score instead of the 8 of the old fp32 scores + probs), together with per-(row, tile) softmax
// statistics (max, sum of exp relative to it).
// k_lq_finish: one thread per key: p = T(exp(x - M) / Sum) with the row's global (M, Sum) folded from the
// tile statistics, fp32 head/query mean -> importance, optional [H, num, S] weights, and the select
// tables of the prune's plan kernel on the way.
// (max, sum of exp) of one row from its tile statistics, folded by ONE wave (every lane gets the result): 8 tiles in
// flight per lane, running max with the sum rescaled to it, then a butterfly over the 64 lanes.  The order of the
// additions is fixed by (tiles) alone, so every workgroup that folds a row - its owner or, after a timeout, anybody -
// gets the same bits.
__device__ inline float2 fold_row_wave(const float2* __restrict__ ts, int tiles) {
    const int lane = lane_id();
    float M = -INFINITY, sum = 0.f;
    for (int t0 = lane; t0 < tiles; t0 += 8 * kWave) {
        float2 ms[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ms[u] = t0 + u * kWave < tiles ? ts[t0 + u * kWave] : make_float2(-INFINITY, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ms[u].y > 0.f) {
                const float Mn = fmaxf(M, ms[u].x);
                sum = sum * expf(M - Mn) + ms[u].y * expf(ms[u].x - Mn);      // (exp(-inf) = 0 on the first)
                M = Mn;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float Mo = __shfl_xor(M, o, kWave), so = __shfl_xor(sum, o, kWave);
        const float Mn = fmaxf(M, Mo);
        sum = (M > -INFINITY ? sum * expf(M - Mn) : 0.f) + (Mo > -INFINITY ? so * expf(Mo - Mn) : 0.f);
        M = Mn;
    }
    return make_float2(M, sum);
}

// Tag of a published granule.  The score kernel that precedes every finish launch on the stream clears the exchange
// area (lq_clear_exchange), so a granule carrying kLqTag was written by THIS launch.
constexpr uint32_t kLqTag = 0x5eed0001u;
__device__ inline void lq_clear_exchange(unsigned long long* xch, int rows_total) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int x = threadIdx.x; x < 2 * rows_total; x += blockDim.x) xch[x] = 0ull;
}

constexpr int kLqKeys = 256;
constexpr int kLqRows = 16;

template <int DT>
__device__ inline float dot16(const uint4& a, const uint4& b, float acc) {
    using A = Act<DT>;
    if constexpr (DT == FF_F32) {
        float x[4], y[4];
        A::unpack(a, x);
        A::unpack(b, y);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(x[e], y[e], acc);
        return acc;
    } else {
        acc = A::dot2(a.x, b.x, acc); acc = A::dot2(a.y, b.y, acc); acc = A::dot2(a.z, b.z, acc);
        return A::dot2(a.w, b.w, acc);
    }
}

// sum over the LPK consecutive lanes that share a key (every lane gets it): DPP permutations inside a
// row of 16 lanes (VALU rate), LDS shuffles only beyond
template <int LPK>
__device__ inline float group_sum(float v) {
    auto dpp = [](float x, int ctrl_id) {
        const int i = __float_as_int(x);
        int r;
        switch (ctrl_id) {
            case 0: r = __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, false); break;      // quad_perm [1,0,3,2]
            case 1: r = __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, false); break;      // quad_perm [2,3,0,1]
            case 2: r = __builtin_amdgcn_update_dpp(0, i, 0x141, 0xf, 0xf, false); break;     // row_half_mirror
            default: r = __builtin_amdgcn_update_dpp(0, i, 0x140, 0xf, 0xf, false); break;    // row_mirror
        }
        return __int_as_float(r);
    };
    if constexpr (LPK >= 2) v += dpp(v, 0);
    if constexpr (LPK >= 4) v += dpp(v, 1);
    if constexpr (LPK >= 8) v += dpp(v, 2);
    if constexpr (LPK >= 16) v += dpp(v, 3);
    if constexpr (LPK >= 32) v += __shfl_xor(v, 16, kWave);
    if constexpr (LPK >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

template <int DT, int LPK>
__global__ __launch_bounds__(256) void k_lq_tile(const void* __restrict__ q, const void* __restrict__ k, int64_t k_head_stride, uint32_t k_key_stride, int H, int H_kv,
                                                 int num, int S, float scale, int causal, const void* __restrict__ bias, int pitch,
                                                 void* __restrict__ scores, float2* __restrict__ tstats, int tiles,
                                                 unsigned long long* __restrict__ xch, int rows_total) {
    using A = Act<DT>;
    constexpr int KPW = kWave / LPK;
    lq_clear_exchange(xch, rows_total);                 // keys per wave-load
    __shared__ uint4 q_lds[kLqRows][LPK];
    __shared__ float sc[kLqKeys][kLqRows + 1];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int hk = blockIdx.y, group = H / H_kv, rows = group * num;
    const int r0 = blockIdx.z * kLqRows, rows_here = min(kLqRows, rows - r0);
    const int tile = blockIdx.x;
    for (int x = tid; x < kLqRows * LPK; x += 256) {
        const int row = x / LPK, part = x - row * LPK;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rows_here) v = ((const uint4*)q)[(size_t)(hk * rows + r0 + row) * LPK + part];
        q_lds[row][part] = v;
    }
    __syncthreads();
    const uint32_t row_bytes = (uint32_t)LPK * 16u;
    const __amdgpu_buffer_rsrc_t krs = make_rsrc((const char*)k + (int64_t)hk * k_head_stride, (uint32_t)(S - 1) * k_key_stride + row_bytes);
    const int part = lane & (LPK - 1), slot = lane / LPK;
#pragma unroll 2
    for (int it = 0; it < kWave / KPW; ++it) {
        const int key_local = w * kWave + it * KPW + slot;
        const int s_key = tile * kLqKeys + key_local;
        const uint4 kv = s_key < S ? buf_load16(krs, (uint32_t)s_key * k_key_stride + (uint32_t)part * 16u) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < kLqRows; ++r) {
            const float acc = group_sum<LPK>(dot16<DT>(kv, q_lds[r][part], 0.f));
            if (part == (r & (LPK - 1))) sc[key_local][r] = acc;
        }
    }
    __syncthreads();
    {   // staged rounding + the causal bias, scores out as T, key-major
        const int s_key = tile * kLqKeys + tid;
        if (s_key < S) {
            for (int r = 0; r < rows_here; ++r) {
                const int n = (r0 + r) % num;
                float v = A::rnd(sc[tid][r]);
                v = A::rnd(v * scale);
                if (causal && s_key > S - num + n) v = A::rnd(v + (-INFINITY));
                if (bias) v = A::rnd(v + A::load1(bias, (int64_t)n * S + s_key));
                sc[tid][r] = v;
                A::store1(scores, (int64_t)s_key * pitch + hk * rows + r0 + r, v);
            }
        } else {
            for (int r = 0; r < rows_here; ++r) sc[tid][r] = -INFINITY;
        }
    }
    __syncthreads();
    {   // tile statistics: 16 threads per row
        const int r = tid >> 4, sub = tid & 15;
        float m = -INFINITY;
        for (int x = sub; x < kLqKeys; x += 16) m = fmaxf(m, sc[x][r]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
        float e = 0.f;
        if (m > -INFINITY)
            for (int x = sub; x < kLqKeys; x += 16) e += expf(sc[x][r] - m);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) e += __shfl_xor(e, o, kWave);
        if (sub == 0 && r < rows_here) tstats[(size_t)(hk * rows + r0 + r) * tiles + tile] = make_float2(m, e);
    }
}

// ---- scores on the matrix cores (16-bit T, dh = 64 / 128 / 256) --------------------------------------------
// q . K^T of the last queries IS a small dense GEMM: [keys x dh] . [dh x rows] with the GQA group's rows
// padded to 32.  The dot2 kernel above is ALU-bound (16 rows x (dot + group sum) per 16 bytes of K); one
// v_mfma_f32_32x32x16 does a 32-key x 32-row x 16-dim block, which leaves the kernel waiting for K.
// Products of two bf16 / fp16 values are exact in fp32 and the accumulation is fp32: the same numbers as
// the reference's matmul up to summation order.  Wave = 64 keys (two 32-key blocks), workgroup = 256 keys.
//   A operand: lane l holds K[key = l & 31][16 kk + 8 (l >> 5) .. + 8]    (16 bytes straight from memory)
//   B operand: lane l holds Q[row = l & 31][16 kk + 8 (l >> 5) .. + 8]    (all kk kept in registers)
//   C/D      : lane l, register r: row(key) = (r & 3) + 8 (r >> 2) + 4 (l >> 5), column(query row) = l & 31
typedef short mfma_ab_t __attribute__((ext_vector_type(8)));
typedef float mfma_cd_t __attribute__((ext_vector_type(16)));

template <int DT, int NK, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_lq_mfma(const void* __restrict__ q, const void* __restrict__ k, int64_t k_head_stride, uint32_t k_key_stride, int H, int H_kv,
                                                 int num, int S, float scale, int causal, const void* __restrict__ bias, int pitch,
                                                 void* __restrict__ scores, float2* __restrict__ tstats, int tiles,
                                                 unsigned long long* __restrict__ xch, int rows_total) {
    using A = Act<DT>;
    static_assert(A::kBytes == 2, "16-bit activations");
    constexpr int kRowsPad = 32;
    __shared__ float2 wstat[WAVES][kRowsPad];
    lq_clear_exchange(xch, rows_total);
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int hk = blockIdx.y, group = H / H_kv, rows = group * num;
    const int r0 = blockIdx.z * kRowsPad, rows_here = min(kRowsPad, rows - r0);
    const int tile = blockIdx.x;
    const int col = lane & 31, half = lane >> 5;
    constexpr uint32_t row_bytes = NK * 32u;                           // dh * 2
    // B fragments: my query row's 8 values of every 16-dim step (zeros for the padding rows)
    mfma_ab_t bfrag[NK];
    {
        const __amdgpu_buffer_rsrc_t qrs = make_rsrc((const char*)q + (size_t)(hk * rows + r0) * row_bytes,
                                                     (uint32_t)rows_here * row_bytes);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const uint4 v = buf_load16(qrs, (uint32_t)col * row_bytes + (uint32_t)kk * 32u + (uint32_t)half * 16u);   // rows past the group: zeros
            bfrag[kk] = __builtin_bit_cast(mfma_ab_t, v);
        }
    }
    // (keys may be rows of a wider tensor - the [S, H_kv, dh] layout attention projections produce: any 16-byte
    // aligned strides; the resource ends with the last key's row)
    const __amdgpu_buffer_rsrc_t krs = make_rsrc((const char*)k + (int64_t)hk * k_head_stride, (uint32_t)(S - 1) * k_key_stride + row_bytes);
    const int key0 = tile * (64 * WAVES) + w * 64;
    mfma_cd_t acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        uint4 a[NK];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
            a[kk] = key0 + b * 32 + col < S ? buf_load16(krs, (uint32_t)(key0 + b * 32 + col) * k_key_stride + (uint32_t)kk * 32u + (uint32_t)half * 16u)
                                            : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[b][x] = 0.f;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            if constexpr (DT == FF_BF16)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_ab_t, a[kk]), bfrag[kk], acc[b], 0, 0, 0);
            else
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, a[kk]),
                                                               __builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, bfrag[kk]),
                                                               acc[b], 0, 0, 0);
        }
    }
    // staged rounding + the causal bias; scores out as T, key-major; statistics of my row
    const int n = (r0 + col) % num;
    float m = -INFINITY;
    float v[2][16];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s_key = key0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float x = A::rnd(acc[b][r]);
            x = A::rnd(x * scale);
            if (causal && s_key > S - num + n) x = A::rnd(x + (-INFINITY));
            if (bias && s_key < S) x = A::rnd(x + A::load1(bias, (int64_t)n * S + s_key));
            if (s_key >= S || col >= rows_here) x = -INFINITY;
            else A::store1(scores, (int64_t)s_key * pitch + hk * rows + r0 + col, x);
            v[b][r] = x;
            m = fmaxf(m, x);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, kWave));
    float e = 0.f;
    if (m > -INFINITY) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) e += expf(v[b][r] - m);
    }
    e += __shfl_xor(e, 32, kWave);
    if (half == 0) wstat[w][col] = make_float2(m, e);
    __syncthreads();
    if (tid < rows_here) {
        float M = -INFINITY, sum = 0.f;
#pragma unroll
        for (int x = 0; x < WAVES; ++x) M = fmaxf(M, wstat[x][tid].x);
#pragma unroll
        for (int x = 0; x < WAVES; ++x) {
            const float2 ms = wstat[x][tid];
            sum += ms.y > 0.f ? ms.y * expf(ms.x - M) : 0.f;
        }
        tstats[(size_t)(hk * rows + r0 + tid) * tiles + tile] = make_float2(M, sum);
    }
}

// ---- k_lq_dot: scores of at most 8 query rows per kv head (num = 1 with a GQA group of up to 8: LLaVA-Video 7B / 72B) ----
// The matrix-core kernel above wants one key per lane: a wave-load touches 32 rows x 32 bytes, and with 8 real rows out of
// the 32 padded ones 3/4 of its epilogue lanes idle - 23 us for the 72 MB of K of the 72B shape, which a bare read delivers
// in 11 us (profiles/r03_membench_small.txt).  Here 8 consecutive lanes own a key and read its row as whole 128-byte
// lines (16 bytes per lane and line, NL = dh / 64 lines): every wave-load is 8 full cache lines.  The 8 rows' partial dots
// (v_dot2c on the packed words, exact products, fp32 sums) meet in a TRANSPOSED reduction over the 8 lanes - 21 DPP /
// select operations instead of 8 butterflies - after which lane q holds the score of row q: no padded lanes in the
// epilogue.  A workgroup owns a contiguous run of keys of one kv head; a wave walks 16 keys per step with the next 16
// (4 KiB) in flight, keeps the running (max, sum of exp) of its row in registers and leaves ONE statistics entry per
// workgroup and row.
template <int DT>
__device__ inline float dot16_packed(const uint4& a, const uint4& b, float acc) {
    using A = Act<DT>;
    acc = A::dot2(a.x, b.x, acc); acc = A::dot2(a.y, b.y, acc); acc = A::dot2(a.z, b.z, acc);
    return A::dot2(a.w, b.w, acc);
}

// a[r] = lane's partial of row r; returns the sum over the 8 lanes of the key for row q = lane & 7
__device__ inline float reduce8_transposed(const float (&a)[8], int q) {
    auto hm = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, false)); };   // row_half_mirror: q <- 7 - q
    auto x2 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false)); };    // quad_perm [2,3,0,1]
    auto x1 = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false)); };    // quad_perm [1,0,3,2]
    const bool lo4 = q < 4, lo2 = (q & 2) == 0, lo1 = (q & 1) == 0;
    float b[4], c[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (lo4 ? a[i] : a[i + 4]) + hm(lo4 ? a[i + 4] : a[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) c[i] = (lo2 ? b[i] : b[i + 2]) + x2(lo2 ? b[i + 2] : b[i]);
    return (lo1 ? c[0] : c[1]) + x1(lo1 ? c[1] : c[0]);
}

template <int DT, int NL>
__global__ __launch_bounds__(256) void k_lq_dot(const void* __restrict__ q, const void* __restrict__ k, int64_t k_head_stride, uint32_t k_key_stride, int H, int H_kv,
                                                int num, int S, float scale, int causal, const void* __restrict__ bias, int pitch,
                                                void* __restrict__ scores, float2* __restrict__ tstats, int keys_per_wg,
                                                unsigned long long* __restrict__ xch, int rows_total) {
    using A = Act<DT>;
    static_assert(A::kBytes == 2, "16-bit activations");
    constexpr int U = 2, WAVES = 4, kStep = 8 * U, kStride = WAVES * kStep;  // key groups per wave and step; keys per wave / per workgroup and step
    __shared__ float2 wstat[WAVES][8];
    lq_clear_exchange(xch, rows_total);
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int hk = blockIdx.y, rows = (H / H_kv) * num;               // <= 8 (the launcher's condition)
    const int sub = lane & 7, kslot = lane >> 3;
    constexpr uint32_t row_bytes = NL * 128u;                          // dh * 2
    const uint32_t lane_off = (uint32_t)sub * 16u;
    uint4 qf[8][NL];                                                   // my 16-byte piece of every line of every query row
    {
        const __amdgpu_buffer_rsrc_t qrs = make_rsrc((const char*)q + (size_t)(hk * rows) * row_bytes, (uint32_t)rows * row_bytes);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NL; ++j) qf[r][j] = buf_load16(qrs, (uint32_t)r * row_bytes + (uint32_t)j * 128u + lane_off);   // rows past the group: zeros
    }
    const __amdgpu_buffer_rsrc_t krs = make_rsrc((const char*)k + (int64_t)hk * k_head_stride, (uint32_t)(S - 1) * k_key_stride + row_bytes);
    const int key_end = min(S, ((int)blockIdx.x + 1) * keys_per_wg);
    int base = blockIdx.x * keys_per_wg + w * kStep;
    struct Stage { uint4 v[U][NL]; };
    auto load = [&](Stage& st, int b0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = b0 + u * 8 + kslot;
#pragma unroll
            for (int j = 0; j < NL; ++j)
                st.v[u][j] = key < key_end ? buf_load16(krs, (uint32_t)key * k_key_stride + (uint32_t)j * 128u + lane_off) : make_uint4(0, 0, 0, 0);
        }
    };
    const int n = sub % num;
    const bool my_row = sub < rows;
    float m_run = -INFINITY, e_run = 0.f;
    auto compute = [&](const Stage& st, int b0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < NL; ++j) acc = dot16_packed<DT>(st.v[u][j], qf[r][j], acc);
                a[r] = acc;
            }
            float x = reduce8_transposed(a, sub);
            // staged rounding + the causal bias; scores out as T, key-major
            x = A::rnd(x);
            const int s_key = b0 + u * 8 + kslot;
            x = A::rnd(x * scale);
            if (causal && s_key > S - num + n) x = A::rnd(x + (-INFINITY));
            const bool live = my_row && s_key < key_end;
            if (bias && live) x = A::rnd(x + A::load1(bias, (int64_t)n * S + s_key));
            if (live) {
                A::store1(scores, (int64_t)s_key * pitch + hk * rows + sub, x);
                if (x > -INFINITY) {                   // running statistics of my row over my keys: one exp
                    const float dlt = x - m_run;                       // (+inf on the first)
                    const float t = expf(-fabsf(dlt));                 // exp(-inf) = 0
                    e_run = dlt > 0.f ? e_run * t + 1.f : e_run + t;
                    m_run = fmaxf(m_run, x);
                }
            }
        }
    };
    // the next step's 4 KiB are in flight while this one is multiplied; the two stages swap by name (no register copies).
    // (two steps in flight, 8 keys per step, more or fewer workgroups: all slower, profiles/r03_k5_experiments.txt)
    Stage s0, s1;
    load(s0, base);
    while (true) {
        load(s1, base + kStride); compute(s0, base); base += kStride; if (base >= key_end) break;
        load(s0, base + kStride); compute(s1, base); base += kStride; if (base >= key_end) break;
    }
    // the 8 key slots of the wave hold different keys of the same row
#pragma unroll
    for (int o = 8; o < kWave; o <<= 1) {
        const float mo = __shfl_xor(m_run, o, kWave), eo = __shfl_xor(e_run, o, kWave);
        const float mn = fmaxf(m_run, mo);
        e_run = (m_run > -INFINITY ? e_run * expf(m_run - mn) : 0.f) + (mo > -INFINITY ? eo * expf(mo - mn) : 0.f);
        m_run = mn;
    }
    if (lane < 8) wstat[w][lane] = make_float2(m_run, e_run);
    __syncthreads();
    if (tid < rows) {
        float M = -INFINITY, sum = 0.f;
#pragma unroll
        for (int x = 0; x < WAVES; ++x) M = fmaxf(M, wstat[x][tid].x);
#pragma unroll
        for (int x = 0; x < WAVES; ++x) {
            const float2 ms = wstat[x][tid];
            sum += ms.y > 0.f ? ms.y * expf(ms.x - M) : 0.f;
        }
        tstats[(size_t)(hk * rows + tid) * gridDim.x + blockIdx.x] = make_float2(M, sum);
    }
}

// k_lq_finish, two phases in one launch.  (1) The row statistics are folded ONCE: the first ceil(rows / 4) workgroups
// own four rows each (a wave per row) and publish (max, sum) as two 8-byte {value, tag} granules (agent-scope store:
// the data is the flag).  Until round 3 every workgroup folded every row (H * num * tiles pairs each: 560 KB per
// workgroup at the 72B shape, 16.5 us for a kernel that moves 5 MB).  (2) Every workgroup polls the granules into
// LDS, then normalises its keys.  Owners never wait before they publish, and workgroups are dispatched in index
// order, so the wait ends; should it not (the contract does not promise dispatch order), a workgroup folds the rows
// itself after ~1 ms - same bits, see fold_row_wave.
template <int DT, int KG>
__device__ inline void lq_finish_body(const void* __restrict__ scores, const float2* __restrict__ tstats,
                                      int rows_total, int pitch, int tiles, int S, void* __restrict__ weights,
                                      void* __restrict__ imp, int lo, int hi, int* __restrict__ l0,
                                      int* t16_end, unsigned long long* __restrict__ xch, int publish, float* row_ms) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    // row_ms: [rows_total][2] floats of dynamic LDS: the row's max and sum of exp
    __shared__ int timed_out;
    const int tid = threadIdx.x;
    {
        const int w = wave_id(), lane = lane_id();
        const int owners = min((int)gridDim.x, (rows_total + 3) / 4);
        if (tid == 0) timed_out = 0;
        if ((int)blockIdx.x < owners && publish) {          // (publish == 0: fault injection of the tests - nobody answers)
            for (int row = blockIdx.x * 4 + w; row < rows_total; row += owners * 4) {
                const float2 ms = fold_row_wave(tstats + (size_t)row * tiles, tiles);
                if (lane < 2)
                    __hip_atomic_store(&xch[2 * row + lane], ((unsigned long long)kLqTag << 32) | (unsigned long long)__float_as_uint(lane ? ms.y : ms.x),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        for (int x0 = 0; x0 < 2 * rows_total; x0 += 256) {
            const int x = x0 + tid;
            const bool need_it = x < 2 * rows_total;
            unsigned long long v = 0;
            for (int spins = 0;; ++spins) {
                if (need_it) v = __hip_atomic_load(&xch[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!need_it || (uint32_t)(v >> 32) == kLqTag)) break;
                __builtin_amdgcn_s_sleep(1);
                if (spins > (1 << 14)) { if (lane == 0) timed_out = 1; break; }
            }
            if (need_it) row_ms[x] = __uint_as_float((uint32_t)v);
        }
        __syncthreads();
        if (timed_out) {                                  // (uniform) an owner never arrived: fold here
            for (int row = w; row < rows_total; row += 4) {
                const float2 ms = fold_row_wave(tstats + (size_t)row * tiles, tiles);
                if (lane == 0) { row_ms[2 * row] = ms.x; row_ms[2 * row + 1] = ms.y; }
            }
        }
    }
    __syncthreads();
    const int s = blockIdx.x * (256 / KG) + tid / KG, sub_k = tid & (KG - 1);
    float v = 0.f;
    {
        float acc = 0.f;
        if (s < S) {
            const uint4* src = (const uint4*)((const char*)scores + (size_t)s * pitch * A::kBytes);
            for (int row0 = sub_k * E; row0 < rows_total; row0 += KG * E) {
                float x[E];
                A::unpack(src[row0 / E], x);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int row = row0 + e;
                    if (row < rows_total) {
                        const float p = A::rnd(expf(x[e] - row_ms[2 * row]) / row_ms[2 * row + 1]);
                        acc += p;
                        if (weights) A::store1(weights, (int64_t)row * S + s, p);
                    }
                }
            }
        }
        acc = group_sum<KG>(acc);
        if (s < S && imp) {
            v = A::rnd(acc / (float)rows_total);
            if (sub_k == 0) A::store1(imp, s, v);
        }
    }
    if (l0) importance_tables<DT>(v, s, sub_k == 0 && s >= lo && s < hi, l0, t16_end);
}

template <int DT, int KG>
__global__ __launch_bounds__(256) void k_lq_finish(const void* __restrict__ scores, const float2* __restrict__ tstats,
                                                   int rows_total, int pitch, int tiles, int S, void* __restrict__ weights,
                                                   void* __restrict__ imp, int lo, int hi, int* __restrict__ l0,
                                                   int* t16_end, unsigned long long* __restrict__ xch, int publish) {
    extern __shared__ __attribute__((aligned(16))) float row_ms_dyn[];
    lq_finish_body<DT, KG>(scores, tstats, rows_total, pitch, tiles, S, weights, imp, lo, hi, l0, t16_end, xch, publish, row_ms_dyn);
}

template <int DT>
static int launch_lq_general(const void* q, const void* k, KStrides ks, int64_t H, int64_t H_kv, int64_t num, int64_t S, int64_t dh,
                             double scale, int causal, const void* bias, void* weights, void* importance, void* ws, int64_t lo, int64_t hi,
                             int* l0, int* t16_end, hipStream_t st) {
    float* scores = (float*)ws;
    float* probs = scores + H * num * S;
    const int rows = (int)((H / H_kv) * num);
    const size_t lds = (size_t)kRowsPerBlock * dh * sizeof(float);
    hipLaunchKernelGGL(k_lq_scores<DT>, dim3((unsigned)((S + 255) / 256), (unsigned)H_kv,
                                             (unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)),
                       dim3(256), lds, st, q, k, ks.head, ks.key, (int)H, (int)H_kv, (int)num, (int)S, (int)dh, (float)scale, causal, bias,
                       scores);
    hipLaunchKernelGGL(k_lq_softmax<DT>, dim3((unsigned)(H * num)), dim3(256), 0, st, scores, (int)S, probs, weights);
    if (importance)
        hipLaunchKernelGGL(k_lq_mean<DT>, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, probs, (int)(H * num),
                           (int)S, importance, (int)lo, (int)hi, l0, t16_end);
    return (int)hipGetLastError();
}

template <int DT>
static int launch_lq_finish(void* scores, float2* tstats, unsigned long long* xch, int rows_total, int pitch, int tiles, int64_t S, void* weights,
                            void* importance, int64_t lo, int64_t hi, int* l0, int* t16_end, hipStream_t st) {
    const int words = pitch * Act<DT>::kBytes / 16;              // 16-byte words of one key's scores
    const size_t lds = (size_t)rows_total * 2 * sizeof(float);
    // FF_LQ_TEST_NO_PUBLISH (tests only): the row owners stay silent, every workgroup runs into its timeout and folds the
    // rows itself - the results must be the same bits (tests/test_gpu_parity.py::test_importance_owner_timeout_same_bits)
    static const int publish = getenv("FF_LQ_TEST_NO_PUBLISH") ? 0 : 1;
#define FF_LQ_FIN(KG)                                                                                                      \
    hipLaunchKernelGGL((k_lq_finish<DT, KG>), dim3((unsigned)((S * KG + 255) / 256)), dim3(256), lds, st, (const void*)scores, \
                       (const float2*)tstats, rows_total, pitch, tiles, (int)S, weights, importance, (int)lo, (int)hi, l0, t16_end, xch, publish)
    if (words >= 8) FF_LQ_FIN(4);
    else if (words >= 4) FF_LQ_FIN(2);
    else FF_LQ_FIN(1);
#undef FF_LQ_FIN
    return (int)hipGetLastError();
}

// Which path launch_lq takes - ONE predicate for the launcher and for the workspace size (they used to be two copies, and
// the size's copy lacked the H * num clause: 256 queries x 28 heads took the general path into a tiled-size workspace).
// Tiled: a key row is read by a power-of-two number of lanes, 16 bytes each, and one key's scores fit the finish
// kernel's row loop.
static bool lq_tiled(int64_t esz, int64_t H, int64_t num, int64_t dh) {
    const int64_t lpk = dh * esz / 16;
    return (dh * esz) % 16 == 0 && lpk >= 1 && lpk <= 64 && (lpk & (lpk - 1)) == 0 && H * num <= 4096;
}

size_t lq_ws_bytes(int dtype, int64_t H, int64_t num, int64_t S, int64_t dh) {
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (!lq_tiled(esz, H, num, dh)) return (size_t)(2 * H * num * S) * sizeof(float);
    const int64_t cap = (S + 63) / 64;                                         // statistics entries per row (k_lq_dot: one per 64-key step at most)
    const int64_t pitch_bytes = (H * num * esz + 15) & ~(int64_t)15;           // one key's scores: whole 16-byte words
    return (size_t)(S * pitch_bytes + H * num * cap * 8 + H * num * 16 + 16);       // scores | tile statistics | row granules
}

// One launch of k_lq_dot<DT, NL>: one round of workgroups (what the current device holds of this instantiation at once,
// cached per device like merge_places in ff_merge.hip), whole steps per workgroup, at least 64 keys per workgroup (the
// workspace has room for S / 64 statistics entries per row).  Returns the number of statistics entries per row.
template <int DT, int NL>
static int launch_lq_dot(const void* q, const void* k, KStrides ks, int64_t H, int64_t H_kv, int64_t num, int64_t S, float scale, int causal,
                         const void* bias, int pitch, void* scores, float2* tstats, unsigned long long* xch, int rows_total, hipStream_t st) {
    static std::atomic<int> cache[kMaxDevices];
    int dev = 0, places = 768;
    const bool cached = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices;
    if (cached) places = cache[dev].load(std::memory_order_relaxed);
    if (!cached || places <= 0) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        places = 768;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lq_dot<DT, NL>, 256, 0) == hipSuccess && per_cu >= 1 &&
            hipGetDeviceProperties(&prop, dev) == hipSuccess)
            places = per_cu * prop.multiProcessorCount;
        if (cached) cache[dev].store(places, std::memory_order_relaxed);
    }
    constexpr int step = 64;                                           // keys per workgroup and step
    int G = (int)(places / H_kv);
    G = G < 1 ? 1 : G;
    int kpw = (int)((S + G - 1) / G);
    kpw = (kpw + step - 1) / step * step;
    G = (int)((S + kpw - 1) / kpw);
    hipLaunchKernelGGL((k_lq_dot<DT, NL>), dim3((unsigned)G, (unsigned)H_kv), dim3(256), 0, st, q, k, ks.head, (uint32_t)ks.key, (int)H,
                       (int)H_kv, (int)num, (int)S, scale, causal, bias, pitch, scores, tstats, kpw, xch, rows_total);
    return G;
}
template <int DT>
static int launch_lq(const void* q, const void* k, KStrides ks, int64_t H, int64_t H_kv, int64_t num, int64_t S, int64_t dh,
                     double scale, int causal, const void* bias, void* weights, void* importance, void* ws, int64_t lo, int64_t hi,
                     int* l0, int* t16_end, hipStream_t st) {
    constexpr int kB = Act<DT>::kBytes;
    const int64_t lpk = dh * kB / 16;
    if (!lq_tiled(kB, H, num, dh)) return launch_lq_general<DT>(q, k, ks, H, H_kv, num, S, dh, scale, causal, bias, weights, importance, ws, lo, hi, l0, t16_end, st);
    const int rows = (int)((H / H_kv) * num), rows_total = (int)(H * num);
    const int tiles = (int)((S + kLqKeys - 1) / kLqKeys);
    void* scores = ws;
    const int64_t pitch_bytes = ((int64_t)rows_total * kB + 15) & ~(int64_t)15;
    const int pitch = (int)(pitch_bytes / kB);
    float2* tstats = (float2*)((char*)scores + S * pitch_bytes);
    const int cap = (int)((S + 63) / 64);
    unsigned long long* xch = (unsigned long long*)(tstats + (size_t)rows_total * cap);
    if constexpr (kB == 2) {
        // matrix-core scores for the head sizes of real models.  4 waves (256 keys) per workgroup = per statistics tile:
        // wider workgroups would mean fewer tiles for the finish kernel to fold (72B shape: 16.5 us at 256 keys per tile,
        // 12.3 at 512, 9.9 at 1024) but cost this kernel more than that (23.8 -> 38.5 -> 42.5 us); 2 waves: 23.7 + 24.5;
        // a streaming form (workgroup = a run of 32-key blocks, next block in flight, one statistics entry per workgroup):
        // 22.9-29.9 us at 256-1024 workgroups against 23.5 (profiles/r03_k5_experiments.txt)
        if ((dh == 64 || dh == 128) && rows <= 8) {
            const int G = dh == 64 ? launch_lq_dot<DT, 1>(q, k, ks, H, H_kv, num, S, (float)scale, causal, bias, pitch, scores, tstats, xch, rows_total, st)
                                   : launch_lq_dot<DT, 2>(q, k, ks, H, H_kv, num, S, (float)scale, causal, bias, pitch, scores, tstats, xch, rows_total, st);
            return launch_lq_finish<DT>(scores, tstats, xch, rows_total, pitch, G, S, weights, importance, lo, hi, l0, t16_end, st);
        }
        if (dh == 64 || dh == 128 || dh == 256) {
            const int64_t zg = (rows + 31) / 32;
            constexpr int waves = 4;
            const int tiles_m = (int)((S + 64 * waves - 1) / (64 * waves));
            const dim3 mgrid((unsigned)tiles_m, (unsigned)H_kv, (unsigned)zg);
#define FF_LQ_MFMA(NK)                                                                                                         \
    hipLaunchKernelGGL((k_lq_mfma<DT, NK, waves>), mgrid, dim3(64 * waves), 0, st, q, k, ks.head, (uint32_t)ks.key, (int)H, (int)H_kv, (int)num, (int)S,   \
                       (float)scale, causal, bias, pitch, scores, tstats, tiles_m, xch, rows_total)
            if (dh == 64) FF_LQ_MFMA(4);
            else if (dh == 128) FF_LQ_MFMA(8);
            else FF_LQ_MFMA(16);
#undef FF_LQ_MFMA
            return launch_lq_finish<DT>(scores, tstats, xch, rows_total, pitch, tiles_m, S, weights, importance, lo, hi, l0, t16_end, st);
        }
    }
    const dim3 grid((unsigned)tiles, (unsigned)H_kv, (unsigned)((rows + kLqRows - 1) / kLqRows));
#define FF_LQ_TILE(LPK)                                                                                                   \
    hipLaunchKernelGGL((k_lq_tile<DT, LPK>), grid, dim3(256), 0, st, q, k, ks.head, (uint32_t)ks.key, (int)H, (int)H_kv, (int)num, (int)S, (float)scale, \
                       causal, bias, pitch, scores, tstats, tiles, xch, rows_total)
    switch ((int)lpk) {
        case 1: FF_LQ_TILE(1); break;
        case 2: FF_LQ_TILE(2); break;
        case 4: FF_LQ_TILE(4); break;
        case 8: FF_LQ_TILE(8); break;
        case 16: FF_LQ_TILE(16); break;
        case 32: FF_LQ_TILE(32); break;
        default: FF_LQ_TILE(64);
    }
#undef FF_LQ_TILE
    return launch_lq_finish<DT>(scores, tstats, xch, rows_total, pitch, tiles, S, weights, importance, lo, hi, l0, t16_end, st);
}

}  // namespace ff

extern "C" int ff_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S, void* importance,
                            ff_stream_t stream) {
    if (!attn_w || !importance || H < 1 || num < 1 || S < 0) return FF_ERR_ARG;
    if (S >= (1ll << 31) || H * num >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (S == 0) return FF_OK;
    return ff::launch_head_mean(attn_w, dtype, H, num, S, importance, 0, 0, nullptr, nullptr, (hipStream_t)stream);
}

namespace ff {
int* ws_l0(void* ws);
int* ws_t16_end(void* ws, size_t ws_bytes);
}  // namespace ff

extern "C" size_t ff_last_query_workspace_bytes(int dtype, int64_t H, int64_t num, int64_t S, int64_t dh) {
    if (H < 1 || num < 1 || S < 1 || dh < 1) return 0;
    return ff::lq_ws_bytes(dtype, H, num, S, dh);
}

static int lq_attention(const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv,
                        int64_t num, int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride,
                        double scale, int causal, const void* bias, void* weights, void* importance, int64_t sel_lo, int64_t sel_hi, void* sel_ws,
                        size_t sel_ws_bytes, void* ws, size_t ws_bytes, ff_stream_t stream) {
    if (!q_last || !k || !ws || H < 1 || H_kv < 1 || num < 1 || S < 1 || dh < 1) return FF_ERR_ARG;
    if (H % H_kv) return FF_ERR_ARG;
    if (!weights && !importance) return FF_ERR_ARG;
    if (dtype != FF_F32 && dtype != FF_BF16 && dtype != FF_F16) return FF_ERR_ARG;
    if (S >= (1ll << 31) || H * num * S >= (1ll << 40)) return FF_ERR_UNSUPPORTED;
    if ((size_t)ff::kRowsPerBlock * dh * sizeof(float) > 64 * 1024) return FF_ERR_UNSUPPORTED;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((dh * esz) & 15) || ((uintptr_t)k & 15) || ((uintptr_t)q_last & 15) || ((uintptr_t)ws & 15)) return FF_ERR_ALIGN;
    if (ws_bytes < ff::lq_ws_bytes(dtype, H, num, S, dh)) return FF_ERR_WORKSPACE;
    int *l0 = nullptr, *t16_end = nullptr;
    if (sel_ws) {                                   // accumulate the select tables of importance[sel_lo, sel_hi)
        if (!importance || sel_lo < 0 || sel_hi > S || sel_lo > sel_hi) return FF_ERR_ARG;
        if (sel_ws_bytes < ff_workspace_bytes(S, 1)) return FF_ERR_WORKSPACE;
        if ((uintptr_t)sel_ws & 15) return FF_ERR_ALIGN;
        l0 = ff::ws_l0(sel_ws);
        t16_end = ff::ws_t16_end(sel_ws, sel_ws_bytes);
    }
    // key layout: 0 / 0 = contiguous [H_kv, S, dh]; else element strides (e.g. dh and H_kv * dh for the [S, H_kv, dh]
    // layout a k_proj output has before any copy)
    if ((k_head_stride == 0) != (k_key_stride == 0) || k_head_stride < 0 || k_key_stride < 0) return FF_ERR_ARG;
    ff::KStrides ks{(k_head_stride ? k_head_stride : S * dh) * esz, (k_key_stride ? k_key_stride : dh) * esz};
    if ((ks.head & 15) || (ks.key & 15)) return FF_ERR_ALIGN;
    if (ks.key < dh * esz) return FF_ERR_ARG;
    if ((S - 1) * ks.key + dh * esz >= (1ll << 32)) return FF_ERR_UNSUPPORTED;     // one kv head must fit a buffer resource
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FF_F32: return ff::launch_lq<FF_F32>(q_last, k, ks, H, H_kv, num, S, dh, scale, causal, bias, weights, importance, ws, sel_lo, sel_hi, l0, t16_end, st);
        case FF_BF16: return ff::launch_lq<FF_BF16>(q_last, k, ks, H, H_kv, num, S, dh, scale, causal, bias, weights, importance, ws, sel_lo, sel_hi, l0, t16_end, st);
        default: return ff::launch_lq<FF_F16>(q_last, k, ks, H, H_kv, num, S, dh, scale, causal, bias, weights, importance, ws, sel_lo, sel_hi, l0, t16_end, st);
    }
}

extern "C" int ff_last_query_attention(const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv,
                                       int64_t num, int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride,
                                       double scale, int causal, const void* bias, void* weights, void* importance, int64_t sel_lo, int64_t sel_hi, void* sel_ws,
                                       size_t sel_ws_bytes, void* ws, size_t ws_bytes, ff_stream_t stream) {
    return lq_attention(q_last, k, dtype, H, H_kv, num, S, dh, k_head_stride, k_key_stride, scale, causal, bias, weights, importance, sel_lo,
                        sel_hi, sel_ws, sel_ws_bytes, ws, ws_bytes, stream);
}

// The attention hook of a context whose prune call comes next: importance + the select tables of importance[start, start + n_img)
// accumulated in the context's workspace (the prune call then passes tables_ready = 1).
extern "C" int ff_ctx_last_query_importance(ff_ctx_t* c, const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv, int64_t num,
                                            int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride, double scale, int causal,
                                            const void* bias, void* importance, int64_t start, int64_t n_img, int64_t k_keep,
                                            void* ws, size_t ws_bytes, ff_stream_t stream) {
    if (!c || !c->member || !c->dst || !c->keep || !c->stats || !c->ws || !importance) return FF_ERR_ARG;
    if (S < 1 || S > c->cap || start < 0 || n_img < 0 || start + n_img > S || k_keep < 0 || k_keep > n_img) return FF_ERR_ARG;
    if (c->ws_bytes < ff_workspace_bytes(c->cap, 1)) return FF_ERR_WORKSPACE;
    if (c->dirty || c->in_flight) return FF_ERR_STATE;             // (the tables must be clean: ff_ctx_reset first)
    c->seq += 1;
    c->dirty = 1;                                                  // until the prune call has consumed (and cleared) the tables
    c->order_len = 0;
    c->last_L = 0;
    return lq_attention(q_last, k, dtype, H, H_kv, num, S, dh, k_head_stride, k_key_stride, scale, causal, bias, nullptr, importance, start,
                        start + n_img, c->ws, c->ws_bytes, ws, ws_bytes, stream);
}
