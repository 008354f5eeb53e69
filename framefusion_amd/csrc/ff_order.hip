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
#include <atomic>

#include "ff_common.h"

namespace ff {

size_t plan_ws_front_bytes(int64_t L);      // ff_plan.hip: the select tables occupy the front (and the tail) of the workspace

constexpr int kOrderThreads = 1024;
constexpr int kOrderWaves = kOrderThreads / kWave;

__device__ inline int bin_of(int64_t t, int P) { return (t >= 0 && t < P) ? (int)t : P; }

// One 64-token step of the stable placement for wave-private histogram row `my`.
__device__ inline void place_step(int* my, int key, bool valid, int i, int32_t* __restrict__ order,
                                  int32_t* __restrict__ inv, unsigned long long lt_mask) {
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
    if (valid) {
        order[slot] = i;
        if (inv) inv[i] = slot;
    }
}

// ---- pass 1: per-slice facts about patch_type (many workgroups, 4096 positions each) ---------------
// Everything the frame-major test needs, so the second launch can decide without touching
// patch_type again: counts, the visual span, whether consecutive visual positions step their type
// by +1 mod P inside the slice, and the slice's first/last visual (index, type) for the checks
// across slice borders.
constexpr int kStatThreads = 256;
constexpr int kStatSpan = kStatThreads * 16;
struct OrderRow {
    int not_text, v_min, v_max, v_cnt, ok, first_idx, first_type, last_idx, last_type, pad[7];
};

__global__ __launch_bounds__(kStatThreads) void k_order_stats(const int64_t* __restrict__ ptype, int L, int P,
                                                              OrderRow* __restrict__ rows) {
    __shared__ int s_first_idx[kStatThreads], s_first_type[kStatThreads];
    __shared__ int s_red[8];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * kStatSpan + tid * 16;
    int64_t t[16];
    if (i0 + 16 <= L) {
        const ulonglong2* p = (const ulonglong2*)(ptype + i0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const ulonglong2 v = p[q];
            t[2 * q] = (int64_t)v.x; t[2 * q + 1] = (int64_t)v.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = i0 + e < L ? ptype[i0 + e] : -1;
    }
    int not_text = 0, v_cnt = 0, ok = 1;
    int first_idx = 0x7fffffff, first_type = -1, last_idx = -1, last_type = -1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + e;
        if (i >= L) continue;
        not_text += (t[e] != -1);
        const int bin = bin_of(t[e], P);
        if (bin < P) {
            if (last_idx == i - 1) ok &= (bin == (last_type + 1 == P ? 0 : last_type + 1));
            if (first_idx == 0x7fffffff) { first_idx = i; first_type = bin; }
            last_idx = i; last_type = bin;
            ++v_cnt;
        }
    }
    s_first_idx[tid] = first_idx;
    s_first_type[tid] = first_type;
    if (tid < 8) s_red[tid] = (tid == 1 || tid == 5) ? 0x7fffffff : (tid == 4 ? 1 : (tid == 2 || tid == 7) ? -1 : 0);
    __syncthreads();
    if (tid + 1 < kStatThreads && last_idx >= 0 && s_first_idx[tid + 1] == last_idx + 1)
        ok &= (s_first_type[tid + 1] == (last_type + 1 == P ? 0 : last_type + 1));
    // block reduce through LDS atomics: [0] not_text [1] v_min [2] v_max [3] v_cnt [4] ok
    not_text = wave_sum_i(not_text);
    v_cnt = wave_sum_i(v_cnt);
    int vmin = first_idx, vmax = last_idx, okw = ok;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vmin = min(vmin, __shfl_xor(vmin, o, kWave)); vmax = max(vmax, __shfl_xor(vmax, o, kWave));
        okw &= __shfl_xor(okw, o, kWave);
    }
    if (lane_id() == 0) {
        atomicAdd(&s_red[0], not_text); atomicMin(&s_red[1], vmin); atomicMax(&s_red[2], vmax);
        atomicAdd(&s_red[3], v_cnt); atomicAnd(&s_red[4], okw);
    }
    __syncthreads();
    if (first_idx == s_red[1]) s_red[6] = first_type;       // type at the slice's first visual
    if (last_idx >= 0 && last_idx == s_red[2]) s_red[7] = last_type;
    __syncthreads();
    if (tid == 0) {
        OrderRow r;
        r.not_text = s_red[0]; r.v_min = s_red[1]; r.v_max = s_red[2]; r.v_cnt = s_red[3]; r.ok = s_red[4];
        r.first_idx = s_red[1]; r.first_type = s_red[3] ? s_red[6] : -1;
        r.last_idx = s_red[2]; r.last_type = s_red[3] ? s_red[7] : -1;
        for (int q = 0; q < 7; ++q) r.pad[q] = 0;
        rows[blockIdx.x] = r;
    }
}

// kKeysInLds: the bin id of every position is staged once in LDS as uint16 (all 16 waves load
// patch_type together, fully coalesced, many loads in flight); the histogram and placement sweeps
// are then pure LDS traffic.  Otherwise the two sweeps re-read patch_type in batches of 8 steps.
template <bool kKeysInLds>
__global__ __launch_bounds__(kOrderThreads) void k_build_order(
    const int64_t* __restrict__ ptype, int L, int P, int n_seg, int seg_len,
    int32_t* __restrict__ order, int32_t* __restrict__ inv, int64_t* __restrict__ stats,
    const OrderRow* __restrict__ rows, int n_rows) {
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
    if (tid == 0) { misc[0] = 0; misc[1] = 0x7fffffff; misc[2] = -1; misc[3] = 0; misc[4] = 0; misc[5] = -1; }
    __syncthreads();

    // phase 0: fold the per-slice facts of k_order_stats.  The frame-major layout the adapters
    // produce for the first call of a prefill (pre text rows, F x [0..P-1], post text rows;
    // llava_video.py:321-336) - visual positions contiguous, a whole number of frames, types
    // stepping +1 mod P from 0 - has the closed form order[p*F + f] = pre + f*P + p: every
    // workgroup of this launch writes its share with coalesced stores (one integer division per
    // thread).  Anything else (ragged later layers, text between frames) is sorted by workgroup 0.
    const uint64_t clk0 = __builtin_amdgcn_s_memtime();
    {
        int not_text = 0, v_min = 0x7fffffff, v_max = -1, v_cnt = 0, ok = 1, first_type = -1;
        for (int g = tid; g < n_rows; g += kOrderThreads) {
            const OrderRow r = rows[g];
            not_text += r.not_text; v_cnt += r.v_cnt; ok &= r.ok;
            v_min = min(v_min, r.v_min); v_max = max(v_max, r.v_max);
            if (r.v_cnt > 0) {
                // the next slice that has a visual token: adjacent indices must step the type too
                for (int h = g + 1; h < n_rows; ++h) {
                    const OrderRow nx = rows[h];
                    if (nx.v_cnt == 0) continue;
                    if (nx.first_idx == r.last_idx + 1) ok &= (nx.first_type == (r.last_type + 1 == P ? 0 : r.last_type + 1));
                    break;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v_min = min(v_min, __shfl_xor(v_min, o, kWave)); v_max = max(v_max, __shfl_xor(v_max, o, kWave));
            v_cnt += __shfl_xor(v_cnt, o, kWave); not_text += __shfl_xor(not_text, o, kWave);
            ok &= __shfl_xor(ok, o, kWave);
        }
        if (lane == 0) {
            atomicAdd(&misc[0], not_text);
            atomicMin(&misc[1], v_min); atomicMax(&misc[2], v_max); atomicAdd(&misc[3], v_cnt);
            if (!ok) atomicOr(&misc[4], 1);
        }
        __syncthreads();
        const int pre = misc[1], last = misc[2], nv = misc[3];
        for (int g = tid; g < n_rows; g += kOrderThreads)
            if (rows[g].v_cnt > 0 && rows[g].first_idx == pre) misc[5] = rows[g].first_type;
        __syncthreads();
        first_type = misc[5];
        const bool regular = nv > 0 && misc[4] == 0 && last - pre + 1 == nv && nv % P == 0 && first_type == 0;
        if (regular) {
            const int F = nv / P;
            const int j0 = (blockIdx.x * kOrderThreads + tid) * 16;
            if (j0 < nv) {
                int p = j0 / F, f = j0 - p * F;
                int v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    v[e] = pre + f * P + p;
                    if (++f == F) { f = 0; ++p; }
                }
                if (j0 + 16 <= nv) {
                    uint4* o4 = (uint4*)(order + j0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o4[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (j0 + e < nv) order[j0 + e] = v[e];
                }
                if (inv) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (j0 + e < nv) inv[v[e]] = j0 + e;
                }
            }
            for (int q = blockIdx.x * kOrderThreads + tid; q < L - nv; q += gridDim.x * kOrderThreads) {
                const int i = q < pre ? q : q + nv;
                order[nv + q] = i;
                if (inv) inv[i] = nv + q;
            }
            if (blockIdx.x == 0 && tid == 0) {
                stats[FF_STAT_NV] = nv;
                stats[FF_STAT_FTN] = misc[0];
                stats[FF_STAT_T_ORDER] = (int64_t)(__builtin_amdgcn_s_memtime() - clk0);
                stats[FF_STAT_T_ORDER + 1] = 0;
            }
            return;
        }
        if (blockIdx.x != 0) return;          // the counting sort below is a single-workgroup job
        if constexpr (kKeysInLds) {
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
                    if (i < L) keys[i] = (uint16_t)bin_of(t[u], P);
                }
            }
        }
        __syncthreads();
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
                place_step(my, valid ? (int)keys[i] : -1, valid, i, order, inv, lt_mask);
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
                    if (i0 + u * kWave < seg_hi) place_step(my, bin_of(t[u], P), i < seg_hi, i, order, inv, lt_mask);
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
                              int32_t* inv, int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream) {
    if (!patch_type || !order || !stats || !ws || L < 0 || patch_num < 1) return FF_ERR_ARG;
    if (ws_bytes < ff::plan_ws_front_bytes(L) + ((size_t)(L / ff::kStatSpan) + 1) * sizeof(ff::OrderRow)) return FF_ERR_WORKSPACE;
    if (((uintptr_t)patch_type & 15) || ((uintptr_t)order & 15) || ((uintptr_t)ws & 15)) return FF_ERR_ALIGN;
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
    // the dynamic LDS limit is an attribute of the function ON a device: cached per device
    static std::atomic<bool> attr_set[ff::kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ff::kMaxDevices) dev = -1;
    if (dev < 0 || !attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)ff::k_build_order<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ff::k_build_order<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_set[dev].store(true, std::memory_order_release);
    }
    // pass 1 (slice facts) + pass 2 (closed form on every workgroup, or the sort on workgroup 0)
    ff::OrderRow* rows = (ff::OrderRow*)((char*)ws + ff::plan_ws_front_bytes(L));
    const int n_rows = (int)((L + ff::kStatSpan - 1) / ff::kStatSpan);
    hipLaunchKernelGGL(ff::k_order_stats, dim3(n_rows), dim3(ff::kStatThreads), 0, (hipStream_t)stream, patch_type,
                       (int)L, (int)patch_num, rows);
    const unsigned nb = (unsigned)((L + ff::kOrderThreads * 16 - 1) / (ff::kOrderThreads * 16));
    if (in_regs)
        hipLaunchKernelGGL(ff::k_build_order<true>, dim3(nb), dim3(ff::kOrderThreads), lds, (hipStream_t)stream,
                           patch_type, (int)L, (int)patch_num, n_seg, seg_len, order, inv, stats, rows, n_rows);
    else
        hipLaunchKernelGGL(ff::k_build_order<false>, dim3(nb), dim3(ff::kOrderThreads), lds, (hipStream_t)stream,
                           patch_type, (int)L, (int)patch_num, n_seg, seg_len, order, inv, stats, rows, n_rows);
    return (int)hipGetLastError();
}
