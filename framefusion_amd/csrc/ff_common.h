// Shared device helpers for the gfx950 kernels of libframefusion_hip.so.
// Compiled with -ffp-contract=off: every rounding below is placed by hand because the reference's
// results are defined by where it rounds to the activation dtype.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "framefusion_hip.h"

namespace ff {

constexpr int kWave = 64;  // gfx950 wavefront

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- raw buffer access (SRSRC): a wave-uniform base + byte count; lanes past the end read 0 and
// their stores are dropped by the hardware, so ragged row tails need no predication.
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// kAux: cache policy bits of the buffer instruction (2 = nt: streamed once, do not keep it cached)
template <int kAux = 0>
__device__ inline uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, 0, kAux);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int kAux = 0>
__device__ inline void buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t voffset, const uint4& x) {
    u32x4 v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voffset, 0, kAux);
}
__device__ inline int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// ---- activation dtype traits ---------------------------------------------------------------
// PER16: elements per 16-byte lane load.  rnd(x): x rounded to T (RNE), returned as float.
template <int DT> struct Act;

template <> struct Act<FF_F32> {
    static constexpr int kPer16 = 4;
    static constexpr int kBytes = 4;
    static constexpr int kKeyBits = 32;
    __device__ static inline float rnd(float x) { return x; }
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
    __device__ static inline float load1(const void* p, int64_t idx) { return ((const float*)p)[idx]; }
    __device__ static inline void store1(void* p, int64_t idx, float x) { ((float*)p)[idx] = x; }
    __device__ static inline uint32_t bits1(const void* p, int64_t idx) { return ((const uint32_t*)p)[idx]; }
};

template <> struct Act<FF_BF16> {
    static constexpr int kPer16 = 8;
    static constexpr int kBytes = 2;
    static constexpr int kKeyBits = 16;
    __device__ static inline float rnd(float x) { return (float)(__bf16)x; }  // v_cvt_pk_bf16_f32 (RNE)
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    // inputs are already T-valued floats (low 16 bits zero)
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4((__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u),
                          (__float_as_uint(f[2]) >> 16) | (__float_as_uint(f[3]) & 0xffff0000u),
                          (__float_as_uint(f[4]) >> 16) | (__float_as_uint(f[5]) & 0xffff0000u),
                          (__float_as_uint(f[6]) >> 16) | (__float_as_uint(f[7]) & 0xffff0000u));
    }
    // any floats: rounded to bf16 (RNE) on the way, two per v_cvt_pk_bf16_f32
    __device__ static inline uint4 pack_rne(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x2_t pr;
            pr.x = (__bf16)f[2 * e];
            pr.y = (__bf16)f[2 * e + 1];
            w[e] = __builtin_bit_cast(uint32_t, pr);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static inline float load1(const void* p, int64_t idx) {
        return __uint_as_float((uint32_t)((const uint16_t*)p)[idx] << 16);
    }
    // acc + a.lo*b.lo + a.hi*b.hi on packed pairs (v_dot2c_f32_bf16: products of two bf16 are exact in fp32)
    __device__ static inline float dot2(uint32_t a, uint32_t b, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
    }
    // acc + sum of squares of the 8 values of a 16-byte word: no unpacking
    __device__ static inline float sumsq(const uint4& v, float acc) {
        acc = dot2(v.x, v.x, acc); acc = dot2(v.y, v.y, acc); acc = dot2(v.z, v.z, acc); return dot2(v.w, v.w, acc);
    }
    // acc + sum_i T(prev_i * x_i): fp32 products rounded pairwise to bf16 (v_cvt_pk_bf16_f32), summed by
    // one dot2 with (1, 1) per pair instead of two unpacks + two adds
    __device__ static inline float dot_rounded(const float* prev, const float* x, float acc) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            bf16x2_t pr;
            pr.x = (__bf16)(prev[e] * x[e]);
            pr.y = (__bf16)(prev[e + 1] * x[e + 1]);
            acc = __builtin_amdgcn_fdot2_f32_bf16(pr, __builtin_bit_cast(bf16x2_t, 0x3f803f80u), acc, false);
        }
        return acc;
    }
    __device__ static inline void store1(void* p, int64_t idx, float x) {
        ((uint16_t*)p)[idx] = (uint16_t)(__float_as_uint(rnd(x)) >> 16);
    }
    __device__ static inline uint32_t bits1(const void* p, int64_t idx) { return ((const uint16_t*)p)[idx]; }
};

template <> struct Act<FF_F16> {
    static constexpr int kPer16 = 8;
    static constexpr int kBytes = 2;
    static constexpr int kKeyBits = 16;
    __device__ static inline float rnd(float x) { return (float)(_Float16)x; }  // v_cvt_f16_f32 (RNE)
    __device__ static inline void unpack2(uint32_t w, float* f) {
        half2_t h = __builtin_bit_cast(half2_t, w);
        f[0] = (float)h.x; f[1] = (float)h.y;
    }
    __device__ static inline void unpack(const uint4& v, float* f) {
        unpack2(v.x, f); unpack2(v.y, f + 2); unpack2(v.z, f + 4); unpack2(v.w, f + 6);
    }
    __device__ static inline uint32_t pack2(float a, float b) {
        half2_t h; h.x = (_Float16)a; h.y = (_Float16)b;
        return __builtin_bit_cast(uint32_t, h);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    __device__ static inline float load1(const void* p, int64_t idx) { return (float)((const _Float16*)p)[idx]; }
    __device__ static inline void store1(void* p, int64_t idx, float x) { ((_Float16*)p)[idx] = (_Float16)x; }
    __device__ static inline float dot2(uint32_t a, uint32_t b, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a), __builtin_bit_cast(half2_t, b), acc, false);
    }
    __device__ static inline float sumsq(const uint4& v, float acc) {
        acc = dot2(v.x, v.x, acc); acc = dot2(v.y, v.y, acc); acc = dot2(v.z, v.z, acc); return dot2(v.w, v.w, acc);
    }
    // acc + sum_i T(a_i * b_i) on raw words: the packed fp16 multiply rounds each (exact) product once,
    // exactly T(a*b); a dot2 with (1, 1) adds the pair
    __device__ static inline float dot_rounded_raw(const uint4& a, const uint4& b, float acc) {
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const half2_t pr = __builtin_bit_cast(half2_t, aw[e]) * __builtin_bit_cast(half2_t, bw[e]);
            acc = __builtin_amdgcn_fdot2(pr, __builtin_bit_cast(half2_t, 0x3c003c00u), acc, false);
        }
        return acc;
    }
    __device__ static inline uint32_t bits1(const void* p, int64_t idx) { return ((const uint16_t*)p)[idx]; }
};

// 16 bytes of T(a_i + b_i): the residual add the decoder performs before a reduction call
// (framefusion/models/qwen2/modeling_qwen2.py:64-67), formed in registers: fp32 add, one rounding to T
// (what torch's bf16 / fp16 add does).
template <int DT>
__device__ inline uint4 add16(const uint4& a, const uint4& b) {
    using A = Act<DT>;
    float x[A::kPer16], y[A::kPer16];
    A::unpack(a, x);
    A::unpack(b, y);
#pragma unroll
    for (int e = 0; e < A::kPer16; ++e) x[e] = A::rnd(x[e] + y[e]);
    return A::pack(x);
}

// Order-preserving unsigned key of a T value given its raw bits: larger value <=> larger key;
// every NaN maps to the maximum (torch.topk ranks NaN highest).
template <int DT> __device__ inline uint32_t order_key(uint32_t bits);
template <> __device__ inline uint32_t order_key<FF_F32>(uint32_t b) {
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
template <> __device__ inline uint32_t order_key<FF_BF16>(uint32_t b) {
    if ((b & 0x7fffu) > 0x7f80u) return 0xffffu;
    return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}
template <> __device__ inline uint32_t order_key<FF_F16>(uint32_t b) {
    if ((b & 0x7fffu) > 0x7c00u) return 0xffffu;
    return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}

// ---- wave / block primitives (wave = 64 lanes) -------------------------------------------------
__device__ inline int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ inline int wave_id() { return threadIdx.x >> 6; }

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ inline int wave_incl_scan(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        int t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
}

// Inclusive wave scan on the DPP network (row shifts + row broadcasts, gfx9 family): VALU-rate, no
// LDS round trips - for the latency-bound index kernels, where a ds_bpermute chain costs ~0.3 us.
__device__ inline int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2 and 3
    return v;
}

// Exclusive scan of one int per thread across a block of NW waves; `total` = block sum.
// `scratch` = NW + 1 ints of LDS; safe to call back to back (two barriers inside).
template <int NW>
__device__ inline int block_excl_scan(int v, int* scratch, int& total) {
    const int incl = wave_incl_scan(v);
    const int w = wave_id();
    if (lane_id() == kWave - 1) scratch[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int s = scratch[k];
        base += (k < w) ? s : 0;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// ---- select tables (ff_plan.hip; filled by the similarity / importance kernels) ------------------
// Level 0: kL0Copies copies of [256 bins of the key's top byte | count(value >= thr)], row stride
// kL0Stride ints, at the START of the workspace.  Level 1: for every slice of kSelSlice consecutive
// values, kT16Copies copies of a 65536-bin table indexed by the key's top 16 bits, laid out DOWN from
// the END of the workspace (slice g ends at ws_end - g * kT16SliceInts): both locations depend on
// neither L nor the call, so "zero on entry" survives calls of different lengths.
constexpr int kMaxDevices = 64;                   // host-side per-device caches (function attributes, occupancy)
constexpr int kSelSlice = 4096;
constexpr int kL0Copies = 16;
constexpr int kL0Stride = 260;                    // 256 bins + count + pad
constexpr int kL0Ints = kL0Copies * kL0Stride;
constexpr int kT16Copies = 1;
constexpr size_t kT16SliceInts = (size_t)kT16Copies * 65536;

__device__ __host__ inline int* t16_slice(int* t16_end, int g) { return t16_end - (size_t)(g + 1) * kT16SliceInts; }
// bin of the 16-bit key prefix inside a level-1 table: the low byte transposed as a 16 x 16 matrix, so
// that NEIGHBOURING key values (a video's similarities are a dozen adjacent bf16 values) sit 64 bytes
// apart - on different cache lines - instead of sharing one line whose atomics would serialise, while
// the 256 bins of one top byte stay one contiguous KiB for the plan kernel's row reads
__device__ __host__ inline uint32_t t16_bin(uint32_t key16) {
    return (key16 & 0xff00u) | ((key16 & 15u) << 4) | ((key16 >> 4) & 15u);
}

// table[idx] += 1 for every lane with `valid`.  Lanes that hold the same idx are folded into ONE
// non-returning atomic for the first kIters distinct values (video similarities take a dozen distinct
// bf16 values: unfolded, thousands of atomics pile up on a few memory-side words); whatever is
// left after kIters rounds adds directly.
template <int kIters>
__device__ inline void wave_agg_add(int* table, uint32_t idx, bool valid) {
    unsigned long long rem = __ballot(valid);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        if (rem == 0ull) break;                                   // wave-uniform
        const int first = __ffsll((long long)rem) - 1;
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)idx, first);
        const unsigned long long m = __ballot(valid && idx == v) & rem;
        if (lane == first) atomicAdd(&table[v], (int)__popcll(m));
        rem &= ~m;
    }
    if ((rem >> lane) & 1ull) atomicAdd(&table[idx], 1);
}

template <int NW>
__device__ inline int block_sum_i(int v, int* scratch) {
    v = wave_sum_i(v);
    if (lane_id() == 0) scratch[wave_id()] = v;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) tot += scratch[k];
    __syncthreads();
    return tot;
}

}  // namespace ff
