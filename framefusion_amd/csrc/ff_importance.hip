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
#include "ff_common.h"

namespace ff {

// scores[h, n, s] as float holding T values.  grid: (ceil(S/256), H_kv, row groups); one LANE per
// key: the lane streams its own key row (dh * sizeof(T) bytes, 16 B at a time) and dots it with up
// to kRowsPerBlock query rows of the GQA group that sit in LDS as fp32 (wave-uniform broadcast
// reads) - no cross-lane reduction, stores coalesced over s.
constexpr int kRowsPerBlock = 8;

template <int DT>
__global__ __launch_bounds__(256) void k_lq_scores(const void* __restrict__ q, const void* __restrict__ k,
                                                   int H, int H_kv, int num, int S, int dh, float scale,
                                                   int causal, float* __restrict__ scores) {
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
    const char* krow = (const char*)k + ((int64_t)hk * S + s) * dh * A::kBytes;
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

// importance[s] = T(mean over H*num of attn_w[h, n, s]) accumulated in fp32 (main.py:70); with l0 the
// select tables of the values in [lo, hi) are accumulated on the way.
template <int DT>
__global__ __launch_bounds__(256) void k_head_mean(const void* __restrict__ w, int rows, int S,
                                                   void* __restrict__ imp, int lo, int hi, int* __restrict__ l0,
                                                   int* t16_end) {
    using A = Act<DT>;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (s < S) {
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) acc += A::load1(w, (int64_t)r * S + s);
        v = A::rnd(acc / (float)rows);
        A::store1(imp, s, v);
    }
    if (l0) importance_tables<DT>(v, s, s >= lo && s < hi, l0, t16_end);
}

int launch_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S, void* importance,
                     int64_t lo, int64_t hi, int* l0, int* t16_end, hipStream_t st) {
    const unsigned blocks = (unsigned)((S + 255) / 256);
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
                                                 void* __restrict__ imp) {
    using A = Act<DT>;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += probs_f[(int64_t)r * S + s];
    A::store1(imp, s, acc / (float)rows);
}

template <int DT>
static int launch_lq(const void* q, const void* k, int64_t H, int64_t H_kv, int64_t num, int64_t S, int64_t dh,
                     double scale, int causal, void* weights, void* importance, void* ws, hipStream_t st) {
    float* scores = (float*)ws;
    float* probs = scores + H * num * S;
    const int rows = (int)((H / H_kv) * num);
    const size_t lds = (size_t)kRowsPerBlock * dh * sizeof(float);
    hipLaunchKernelGGL(k_lq_scores<DT>, dim3((unsigned)((S + 255) / 256), (unsigned)H_kv,
                                             (unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)),
                       dim3(256), lds, st, q, k, (int)H, (int)H_kv, (int)num, (int)S, (int)dh, (float)scale, causal,
                       scores);
    hipLaunchKernelGGL(k_lq_softmax<DT>, dim3((unsigned)(H * num)), dim3(256), 0, st, scores, (int)S, probs, weights);
    if (importance)
        hipLaunchKernelGGL(k_lq_mean<DT>, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, probs, (int)(H * num),
                           (int)S, importance);
    return (int)hipGetLastError();
}

}  // namespace ff

extern "C" int ff_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S, void* importance,
                            ff_stream_t stream) {
    if (!attn_w || !importance || H < 1 || num < 1 || S < 0) return FF_ERR_ARG;
    if (S >= (1ll << 31) || H * num >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if (S == 0) return FF_OK;
    return ff::launch_head_mean(attn_w, dtype, H, num, S, importance, 0, 0, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int ff_last_query_attention(const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv,
                                       int64_t num, int64_t S, int64_t dh, double scale, int causal,
                                       void* weights, void* importance, void* ws, size_t ws_bytes,
                                       ff_stream_t stream) {
    if (!q_last || !k || !ws || H < 1 || H_kv < 1 || num < 1 || S < 1 || dh < 1) return FF_ERR_ARG;
    if (H % H_kv) return FF_ERR_ARG;
    if (!weights && !importance) return FF_ERR_ARG;
    if (S >= (1ll << 31) || H * num * S >= (1ll << 40)) return FF_ERR_UNSUPPORTED;
    if ((size_t)ff::kRowsPerBlock * dh * sizeof(float) > 64 * 1024) return FF_ERR_UNSUPPORTED;
    const int64_t esz = dtype == FF_F32 ? 4 : 2;
    if (((dh * esz) & 15) || ((uintptr_t)k & 15)) return FF_ERR_ALIGN;
    if (ws_bytes < (size_t)(2 * H * num * S) * sizeof(float)) return FF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case FF_F32: return ff::launch_lq<FF_F32>(q_last, k, H, H_kv, num, S, dh, scale, causal, weights, importance, ws, st);
        case FF_BF16: return ff::launch_lq<FF_BF16>(q_last, k, H, H_kv, num, S, dh, scale, causal, weights, importance, ws, st);
        case FF_F16: return ff::launch_lq<FF_F16>(q_last, k, H, H_kv, num, S, dh, scale, causal, weights, importance, ws, st);
        default: return FF_ERR_ARG;
    }
}
