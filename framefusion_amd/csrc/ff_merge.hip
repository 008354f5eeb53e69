// K4 - run merge + compaction, the second (and last) streaming pass over the activations.
// Replaces index_add_ + divide (framefusion/main.py:304-317) and the bool-mask gathers of
// hidden_states, cos/sin (or position ids) and patch_type (main.py:132-138, 161-178), which the
// reference runs as 6+ kernels with host syncs, by one launch that reads every surviving or
// folded row exactly once and writes every output row exactly once.
//
// Work decomposition: by INPUT rows, not output rows - a wave owns about `slots` consecutive slots t of
// the by-patch order (then the non-visual tail) for one 1 KiB column block.  Member slots are
// read by the wave that owns their anchor: the boundary between two waves' ranges sits on the non-member
// slot nearest to a multiple of `slots`.  Because every input row is read by exactly one wave per column
// block and the ranges are even (headline call: 37 +- 3.9 slots, longest 53), the HBM read load is
// balanced no matter how long individual runs are (an output-row decomposition would leave the
// longest run as the tail).
//
// Arithmetic: the anchor accumulates its members in by-patch
// order with a rounding to the activation dtype T after EVERY add, then one rounded divide by
// T(n+1) - the order CPU index_add_ applies; fp32-accumulate-then-round differs on ~23 % of
// elements by more than 1e-3 relative.
#include <atomic>

#include "ff_common.h"

namespace ff {

constexpr int kMergeThreads = 256;
constexpr int kMergeWaves = kMergeThreads / kWave;


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

// Waves are independent (no LDS, no barrier): the 4 waves of a workgroup own the same `slots`
// consecutive by-patch slots and one 1 KiB column tile each (16 bytes per lane), so a workgroup
// reads 4 KiB of a row at a time.  A wave treats its job as a STREAM of rows in by-patch order:
// it runs from boundary(t0) to boundary(t0 + slots) (see `boundary` in the kernel: the nearest non-member
// slot); a non-member row opens a new output row, a member row is folded into the open one.
// One coalesced load of order[] / member[] for 64 slots tells the wave the whole stream, so the
// row pieces are requested kDepth at a time, the next batch being issued BEFORE the current one
// is folded (two register batches), with no dependent index fetch in between (dst[] is only
// needed by the stores).  The additions stay sequential - a rounding after each - but the loads
// do not wait for them.  Row pieces move as raw buffer loads/stores (lanes past the row end read
// 0 / are dropped).
template <int kDepth, bool kAdd>
struct Batch {
    uint4 buf[kDepth];
    uint4 buf2[kAdd ? kDepth : 1];   // the addend's pieces (kAdd: rows are T(hidden + addend))
    int idx[kDepth];       // sequence index of each row (wave-uniform)
    int pos, take;
    unsigned mem_bits;
    bool last;
};

// Select tables to clear for the next call (ff_plan.hip).  The level-0 table (a few KB) is cleared as a
// byte range; the per-slice level-1 tables are megabytes of which a few hundred bins are non-zero, so
// they are cleared BY KEY: every value that was counted names its bin (streaming zeros over the whole
// table inside this kernel cost 8 us at 64 x 576 - measured).
struct ZeroJob {
    uint4* a;                 // level-0 table
    int a_n16;
    const void* keys;         // the values the producer counted (n of them, dtype key_dt), or NULL
    int n, key_dt;
    int* t16_end;
    int n_blocks;
};

template <int KDT>
__device__ inline void zero_by_key(const ZeroJob& z, int t) {
    using K = Act<KDT>;
    const uint32_t bin = t16_bin(order_key<KDT>(K::bits1(z.keys, t)) >> (K::kKeyBits - 16));
    int* tab = t16_slice(z.t16_end, t / kSelSlice) + bin;
#pragma unroll
    for (int x = 0; x < kT16Copies; ++x) tab[x * 65536] = 0;
}

template <int DT, bool kAdd>
__global__ __launch_bounds__(kMergeThreads) void k_merge_compact(
    const char* __restrict__ hidden, const char* __restrict__ addend, char* __restrict__ out, uint32_t row_bytes, int L, int64_t L_cap,
    const int32_t* __restrict__ order, const uint8_t* __restrict__ member, int fold,
    const int32_t* __restrict__ dst, const uint8_t* __restrict__ keep, AuxPack aux, int n_main,
    int n_aux_blocks, int n_next_blocks, int32_t* __restrict__ order_next, int32_t* __restrict__ inv_next,
    int64_t* __restrict__ stats, const int64_t* __restrict__ identity_stats, ZeroJob zero, int slots) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    constexpr int kDepth = 4;                     // row pieces requested per batch (two batches in flight)
    const int lane = lane_id();
    if ((int)blockIdx.x >= n_main + n_aux_blocks + n_next_blocks) {
        // ---- the select tables of this call have been consumed by the plan kernel: clear them for the
        // next call's producer (runs even when nothing is folded)
        if (blockIdx.y != 0) return;
        const int zb = (int)blockIdx.x - n_main - n_aux_blocks - n_next_blocks;
        if (zb == 0)
            for (int z = threadIdx.x; z < zero.a_n16; z += kMergeThreads) zero.a[z] = make_uint4(0, 0, 0, 0);
        if (zero.keys) {
            const int t0 = (zb * kMergeThreads + threadIdx.x) * 16;
            for (int t = t0; t < min(t0 + 16, zero.n); ++t) {
                if (zero.key_dt == FF_BF16) zero_by_key<FF_BF16>(zero, t);
                else if (zero.key_dt == FF_F16) zero_by_key<FF_F16>(zero, t);
                else zero_by_key<FF_F32>(zero, t);
            }
        }
        return;
    }
    // nothing folded (a merge call whose threshold set is empty, main.py:264-266): the reduced
    // sequence IS the input, the caller keeps using its own tensors and this launch writes nothing
    if (identity_stats && identity_stats[FF_STAT_MERGED] == 0) return;
    if ((int)blockIdx.x >= n_main + n_aux_blocks) {
        // ---- by-patch order of the COMPACTED sequence, for the next merge call (order maintenance):
        // the surviving slots keep their relative by-patch order and dst[] is monotonic in the
        // sequence position, so new_order = dst[order[t]] compacted over the non-member slots.
        // Same communication-free scan as k_scan: this workgroup recounts the slots before its own.
        if (blockIdx.y != 0) return;
        __shared__ int scratch[kMergeWaves + 1];
        const int tid = threadIdx.x;
        const int base = ((int)blockIdx.x - n_main - n_aux_blocks) * (kMergeThreads * 16);
        int before = 0;
        for (int off = tid * 16; off < base; off += kMergeThreads * 16) {
            const uint4 m4 = *(const uint4*)(member + off);
            before += 16 - (__popc(m4.x) + __popc(m4.y) + __popc(m4.z) + __popc(m4.w));
        }
        before = block_sum_i<kMergeWaves>(before, scratch);
        const int s0 = base + tid * 16;
        const int n_here = min(max(L - s0, 0), 16);
        unsigned nonmem = 0;
        for (int e = 0; e < n_here; ++e) nonmem |= (member[s0 + e] ? 0u : 1u) << e;
        int total;
        int pos = before + block_excl_scan<kMergeWaves>(__popc(nonmem), scratch, total);
        for (int e = 0; e < n_here; ++e) {
            if ((nonmem >> e) & 1u) {
                const int np = dst[order[s0 + e]];      // the slot's position in the compacted sequence
                order_next[pos] = np;
                if (inv_next) inv_next[np] = pos;
                ++pos;
            }
        }
        if (base + kMergeThreads * 16 >= L && tid == 0) {
            const int64_t merged = stats[FF_STAT_MERGED];
            stats[FF_STAT_NV] -= merged;        // the next call (order_valid) skips K0, which would set these
            stats[FF_STAT_FTN] -= merged;
        }
        return;
    }
    if ((int)blockIdx.x >= n_main) {
        // ---- auxiliary rows (position embeddings, patch types, position ids): plain compaction by
        // SEQUENCE position - reads coalesced, writes in increasing order.  Only blockIdx.y == 0.
        if (blockIdx.y != 0) return;
        const int i = ((int)blockIdx.x - n_main) * kMergeWaves * 4 + wave_id() * 4 + (lane >> 4);
        const int sub = lane & 15;                       // 16 lanes per row
        if (i >= L || !keep[i]) return;
        const int r = dst[i];
        for (int x = 0; x < aux.n; ++x) {
            const ff_aux_t& ax = aux.a[x];
            for (int64_t ou = 0; ou < ax.outer; ++ou)
                copy_row((const char*)ax.src + (ou * L + i) * ax.row_bytes,
                         (char*)ax.dst + (ou * L_cap + r) * ax.row_bytes, ax.row_bytes, sub, 16);
        }
        return;
    }
    // slot groups are walked from the END of the by-patch order: the similarity pass read the rows
    // in ascending order, so its most recently fetched rows - the ones the 256 MiB Infinity Cache
    // still holds - are the first ones this pass asks for
    const int t0 = (n_main - 1 - (int)blockIdx.x) * slots;
    const int cb = uniform(blockIdx.y * kMergeWaves + wave_id());     // 1 KiB column tile
    const uint32_t col = (uint32_t)cb * 1024u;
    if (col >= row_bytes) return;
    const uint32_t blk_bytes = min(1024u, row_bytes - col);
    const uint32_t voff = (uint32_t)lane * 16;
    const int t_end = min(t0 + slots, L);

    // window of 64 slots: row indices and member flags, starting `look` slots BEFORE the nominal range
    const int look = min(10, (kWave - slots) / 2);          // slots <= 53 (merge_slots): look >= 5
    const int win0 = t0 - look;
    int win = win0;
    const int sl = win0 + lane;
    const bool sl_ok = sl >= 0 && sl < L;
    int ordw = sl_ok ? (order ? order[sl] : sl) : 0;
    const bool sl_mem = sl_ok ? (member[sl] != 0) : false;
    unsigned long long memw = __ballot(sl_mem);
    const unsigned long long nonmem = __ballot(sl_ok && !sl_mem);

    // The stream of this wave is [bs, be): a run cannot be split between waves (the rounding after every add makes the fold
    // sequential), so the boundary between two slot groups moves to a non-member slot - the NEAREST one within `look` slots
    // before the nominal boundary x, else the first one at or after x (the rule until round 3, which made the longest
    // stream of the headline call 68 slots for a mean of 37; nearest: 53, profiles/r03_k4_probes.txt).  boundary(x) only
    // reads the flags of [x - look, ...): both neighbours of a boundary compute the same slot.
    auto first_nonmember_from = [&](int s0) {               // rare: no non-member left in the window
        for (int s = s0; s < L; s += kWave) {
            const unsigned long long b = __ballot(s + lane < L && member[s + lane] == 0);
            if (b) return s + (int)__ffsll((long long)b) - 1;
        }
        return L;
    };
    auto boundary = [&](int x) {                            // x in [t0, t_end]: inside the window
        if (x >= L) return L;
        const int xr = x - win0;
        const unsigned long long above = nonmem >> xr;
        const int fwd = above ? x + (int)__ffsll((long long)above) - 1 : first_nonmember_from(win0 + kWave);
        unsigned long long below = nonmem & ((1ull << xr) - 1ull);
        if (xr > look) below &= ~((1ull << (xr - look)) - 1ull);
        if (below) {
            const int bwd = win0 + 63 - (int)__builtin_clzll(below);
            if (x - bwd < fwd - x) return bwd;
        }
        return fwd;
    };
    const int bs = boundary(t0), be = boundary(t_end);
    if (bs >= be) return;                                   // the whole nominal range belongs to a neighbour's stream
    // output rows of my anchors: non-members of [bs, be), all inside the first window (bs < t_end <= win0 + 64 - look, and
    // everything from t_end to be is a member)
    const bool anchor_lane = sl_ok && !sl_mem && sl >= bs && sl < be;
    const int dv = anchor_lane ? dst[ordw] : 0;

    auto piece = [&](int i) { return make_rsrc(hidden + (int64_t)i * row_bytes + col, blk_bytes); };
    auto piece2 = [&](int i) { return make_rsrc((kAdd ? addend : hidden) + (int64_t)i * row_bytes + col, blk_bytes); };

    // request the next (up to) kDepth rows of the stream
    auto issue = [&](Batch<kDepth, kAdd>& b, int pos) {
        if (pos - win > kWave - kDepth) {                   // keep kDepth slots of lookahead in the window
            win = pos;
            ordw = (pos + lane < L) ? (order ? order[pos + lane] : pos + lane) : 0;
            memw = __ballot((pos + lane < L) ? (member[pos + lane] != 0) : false);
        }
        const int rel = pos - win;
        int take = 0;
        unsigned mem_bits = 0;
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            const int s = pos + u;
            const bool is_mem = (memw >> (rel + u)) & 1ull;
            const bool in = take == u && s < be;
            if (in) { ++take; mem_bits |= is_mem ? (1u << u) : 0u; }
            b.idx[u] = __builtin_amdgcn_readlane(ordw, rel + u);
        }
        b.pos = pos; b.take = take; b.mem_bits = mem_bits; b.last = take < kDepth;
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            const bool is_mem = (mem_bits >> u) & 1u;
            if (u < take && (fold || !is_mem)) {
                b.buf[u] = buf_load16<2>(piece(b.idx[u]), voff);
                if constexpr (kAdd) b.buf2[u] = buf_load16<2>(piece2(b.idx[u]), voff);
            }
        }
    };

    float acc[E];
    int open_r = -1, open_n = 0;                            // the output row being accumulated
    auto flush = [&]() {
        float o[E];
        if (open_n > 0 && fold == FF_FOLD_MEAN) {
            // torch .mean(dim=1): fp32 sum (which starts from +0, so an all -0 column gives +0) / N, one rounding
            const float div = (float)(open_n + 1);
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = A::rnd((acc[e] + 0.0f) / div);
        } else if (open_n > 0) {
            const float div = A::rnd((float)(open_n + 1));
            if constexpr (DT == FF_BF16) {
                // bf16 only: T(a / div) == T(a * RN(1 / div)) for EVERY bf16-valued a (acc is one: it is rounded after each
                // add) and every divisor T(k) - a quotient of two 8-bit significands is never closer than 2^-17 (relative)
                // to a bf16 rounding boundary and never on one, the product is within 2^-23 of it; checked exhaustively
                // (65 536 values x the 1 288 divisors up to T(70 000)) by tests/test_host_logic.py.  One IEEE reciprocal
                // per flush instead of eight IEEE divisions; fp16 fails the same check (11-bit significands) and keeps
                // the division, like fp32.
                const float r = 1.0f / div;
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = acc[e] * r;
                buf_store16<2>(make_rsrc(out + (int64_t)open_r * row_bytes + col, blk_bytes), voff, A::pack_rne(o));
                return;
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = A::rnd(acc[e] / div);
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = acc[e];
        }
        buf_store16<2>(make_rsrc(out + (int64_t)open_r * row_bytes + col, blk_bytes), voff, A::pack(o));
    };
    auto row_of = [&](const Batch<kDepth, kAdd>& b, int u, float* f) {          // the (summed) row piece as T-valued floats
        if constexpr (kAdd) {
            float y[E];
            A::unpack(b.buf[u], f);
            A::unpack(b.buf2[u], y);
#pragma unroll
            for (int e = 0; e < E; ++e) f[e] = A::rnd(f[e] + y[e]);
        } else {
            A::unpack(b.buf[u], f);
        }
    };
    auto fold_batch = [&](Batch<kDepth, kAdd>& b) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (u < b.take) {
                const bool is_mem = (b.mem_bits >> u) & 1u;
                if (!is_mem) {
                    if (open_r >= 0) flush();
                    open_r = __builtin_amdgcn_readlane(dv, b.pos + u - win0);
                    open_n = 0;
                    row_of(b, u, acc);
                } else if (fold == FF_FOLD_MEAN) {
                    float x[E];
                    row_of(b, u, x);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = acc[e] + x[e];
                    ++open_n;
                } else if (fold) {
                    float x[E];
                    row_of(b, u, x);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = A::rnd(acc[e] + x[e]);
                    ++open_n;
                }
            }
        }
    };

    Batch<kDepth, kAdd> b0, b1;
    int pos = bs;
    issue(b0, pos);
    pos += b0.take;
    while (true) {
        if (!b0.last) { issue(b1, pos); pos += b1.take; }
        fold_batch(b0);
        if (b0.last) break;
        if (!b1.last) { issue(b0, pos); pos += b0.take; }
        fold_batch(b1);
        if (b1.last) break;
    }
    if (open_r >= 0) flush();
}

// ---- square attention-mask gather (main.py:137-138, 99-100): out[r, c] = mask[src[r], src[c]] ----------------
// Round 2 ran one workgroup per INPUT row with a dst[] lookup per input element: L^2 index reads and 2-byte scattered
// stores (2.7 GB of traffic for a bf16 mask at 37 k tokens).  Two levels now: k_invert_dst turns dst[] (row of every
// kept position) into src[] (position of every output row) once, then one workgroup per OUTPUT row walks the output
// columns - src[] read as whole words, the mask row gathered through it (ascending, mostly neighbouring elements),
// 16-byte stores.  What is left is the mask itself: L_out^2 elements written, the kept part of L_out rows read.
// (A variant that reads the input rows as whole aligned 16-byte chunks and stages the survivors of a 4 KiB segment in LDS
// before aligned stores was built and measured: 243 / 1153 us against 171 / 847 us for this one at 12.5 k / 36.9 k tokens -
// three barriers and two LDS atomics per 4 KiB cost more than the 2-byte gathers; profiles/r03_mask_gather.txt.)
__global__ __launch_bounds__(256) void k_invert_dst(const int32_t* __restrict__ dst, int L, int32_t* __restrict__ src) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) {
        const int r = dst[i];
        if (r >= 0) src[r] = i;
    }
}

template <int EB>
__global__ __launch_bounds__(256) void k_gather_mask(const char* __restrict__ mask, char* __restrict__ out, int L, int64_t L_cap,
                                                     const int32_t* __restrict__ src, const int64_t* __restrict__ stats) {
    constexpr int V = 16 / EB;                               // elements per 16-byte store
    const int L_out = (int)stats[FF_STAT_LOUT];
    const int r = blockIdx.x;
    if (r >= L_out) return;
    const char* row_in = mask + (int64_t)src[r] * L * EB;
    char* row_out = out + (int64_t)r * L_cap * EB;
    const bool wide = (((uintptr_t)row_out | (uintptr_t)((int64_t)L_cap * EB)) & 15) == 0;
    for (int c0 = threadIdx.x * V; c0 < L_out; c0 += 256 * V) {
        if (wide && c0 + V <= L_out) {
            int idx[V];
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const int4 w = *(const int4*)(src + c0 + e);     // (src is 16-byte aligned, c0 a multiple of V >= 4... V = 2: below)
                idx[e] = w.x; idx[e + 1] = w.y; idx[e + 2] = w.z; idx[e + 3] = w.w;
            }
            uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if constexpr (EB == 1) pk[e >> 2] |= (uint32_t)(uint8_t)row_in[idx[e]] << (8 * (e & 3));
                else if constexpr (EB == 2) pk[e >> 1] |= (uint32_t)((const uint16_t*)row_in)[idx[e]] << (16 * (e & 1));
                else pk[e] = ((const uint32_t*)row_in)[idx[e]];
            }
            *(uint4*)(row_out + (int64_t)c0 * EB) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        } else {
            for (int e = 0; e < V && c0 + e < L_out; ++e) {
                const int c = src[c0 + e];
                if constexpr (EB == 1) row_out[c0 + e] = row_in[c];
                else if constexpr (EB == 2) ((uint16_t*)row_out)[c0 + e] = ((const uint16_t*)row_in)[c];
                else ((uint32_t*)row_out)[c0 + e] = ((const uint32_t*)row_in)[c];
            }
        }
    }
}

// 8-byte elements (fp64 masks): two per 16-byte store
__global__ __launch_bounds__(256) void k_gather_mask8(const char* __restrict__ mask, char* __restrict__ out, int L, int64_t L_cap,
                                                      const int32_t* __restrict__ src, const int64_t* __restrict__ stats) {
    const int L_out = (int)stats[FF_STAT_LOUT];
    const int r = blockIdx.x;
    if (r >= L_out) return;
    const uint64_t* row_in = (const uint64_t*)(mask + (int64_t)src[r] * L * 8);
    uint64_t* row_out = (uint64_t*)(out + (int64_t)r * L_cap * 8);
    for (int c = threadIdx.x; c < L_out; c += 256) row_out[c] = row_in[src[c]];
}

// Slots per workgroup, chosen per launch (measured at 64 x 576 x 4096, K4 in the step; 32 was the fixed value):
//  * the waves are long streams, so a launch whose workgroups do not all fit on the chip at once ends with a
//    nearly empty extra round: 32 slots = 2 304 workgroups on 2 048 places 79 us, 36 slots (2 048) 73.6 us,
//    40 slots (1 844) 75.8 us, 48 slots (1 536) 80.5 us.  The main workgroups are sized to ~97 % of ONE round's
//    places (the rest is for the short aux / order / table blocks passing through) whenever that fits;
//  * all workgroups start together and advance at the same pace: when the slot count shares a factor with the
//    frames per patch they all sit on the same few frames - the same few MB - at any time (64 frames: 16 / 24 /
//    32 slots 34.5 us, 19 / 21 slots 31 us at 64 x 210 x 3584; 36 / 38 slots 73.6 us, 37 slots 71 us above).
//    A prime count staggers the workgroups over the frames whatever the frame count is.
template <int DT, bool kAdd>
static int merge_places() {
    // workgroups of this instantiation the CURRENT device holds at once (cached per device: several
    // replicas on several GPUs may share the process)
    static std::atomic<int> cache[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    if (dev >= 0 && dev < kMaxDevices) {
        const int got = cache[dev].load(std::memory_order_relaxed);
        if (got > 0) return got;
    }
    int per_cu = 0, places = 2048;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_merge_compact<DT, kAdd>, kMergeThreads, 0) == hipSuccess &&
        per_cu >= 1)
        places = per_cu * prop.multiProcessorCount;
    if (dev >= 0 && dev < kMaxDevices) cache[dev].store(places, std::memory_order_relaxed);
    return places;
}
static int merge_slots(int dtype, bool add, int64_t L, int ny) {
    int places;
    switch (dtype) {
        case FF_F32: places = add ? merge_places<FF_F32, true>() : merge_places<FF_F32, false>(); break;
        case FF_BF16: places = add ? merge_places<FF_BF16, true>() : merge_places<FF_BF16, false>(); break;
        default: places = add ? merge_places<FF_F16, true>() : merge_places<FF_F16, false>();
    }
    static const int primes[] = {17, 19, 23, 29, 31, 37, 41, 43, 47, 53};
    const double want = (double)L * ny / (0.975 * (double)places);
    for (int p : primes)
        if ((double)p >= want) return p;
    // One round does not hold the call.  Whole rounds of long workgroups only pay when they are WHOLE (128 x 576 x 4096:
    // 37 slots = 1.996 rounds 155 us, 11 slots 163, 7 slots 164, 5 slots 158; 96 x 576 x 4096: 29 slots = 1.91 rounds 127 us,
    // 7 slots 121.5) and the rows are at most two column groups wide (64 x 576 x 8192, four groups: 37 slots = 2.0 rounds
    // 166 us, 29 162, 11 159, 7 158): otherwise many short workgroups (profiles/r03_k5_experiments.txt)
    if (ny <= 2)
        for (int r = 2; r <= 3; ++r)
            for (int p : primes)
                if ((double)p >= want / r) {
                    if (want / r >= 0.985 * p) return p;
                    break;
                }
    return 7;
}

int launch_merge_compact(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                         const int32_t* order, const uint8_t* member, int fold, const int32_t* dst,
                         const uint8_t* keep, const ff_aux_t* aux_host, int n_aux, int32_t* order_next,
                         int32_t* inv_next, int64_t* stats, hipStream_t st, bool skip_identity, void* zero_a,
                         size_t zero_a_bytes, const void* zero_keys, int64_t zero_n, int zero_key_dt, int* t16_end) {
    AuxPack pack;
    pack.n = keep ? n_aux : 0;
    for (int x = 0; x < pack.n; ++x) pack.a[x] = aux_host[x];
    for (int x = pack.n; x < FF_MAX_AUX; ++x) pack.a[x] = ff_aux_t{nullptr, nullptr, 0, 0};
    const int64_t row_bytes = d * (dtype == FF_F32 ? 4 : 2);
    const int nblk = (int)((row_bytes + 1023) / 1024);
    const int ny = (nblk + kMergeWaves - 1) / kMergeWaves;
    const int slots = merge_slots(dtype, addend != nullptr, L, ny);
    const int n_main = (int)((L + slots - 1) / slots);
    const int n_aux_blocks = pack.n ? (int)((L + kMergeWaves * 4 - 1) / (kMergeWaves * 4)) : 0;
    if (!order || !stats) order_next = nullptr;
    const int n_next_blocks = order_next ? (int)((L + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 0;
    ZeroJob zero{(uint4*)zero_a, (int)(zero_a_bytes / 16), zero_keys, (int)zero_n, zero_key_dt, t16_end, 0};
    if (zero_a) zero.n_blocks = zero_keys ? (int)((zero_n + kMergeThreads * 16 - 1) / (kMergeThreads * 16)) : 1;
    const dim3 grid((unsigned)(n_main + n_aux_blocks + n_next_blocks + zero.n_blocks), (unsigned)ny);
    const char* h = (const char*)hidden;
    char* o = (char*)hidden_out;
    const int64_t* ident = (skip_identity && stats) ? stats : nullptr;
#define FF_MC_LAUNCH(DT, ADD)                                                                                          \
    hipLaunchKernelGGL((k_merge_compact<DT, ADD>), grid, dim3(kMergeThreads), 0, st, h, (const char*)addend, o,          \
                       (uint32_t)row_bytes, (int)L, L_cap, order, member, fold, dst, keep, pack, n_main, n_aux_blocks,  \
                       n_next_blocks, order_next, inv_next, stats, ident, zero, slots)
    switch (dtype) {
        case FF_F32: if (addend) FF_MC_LAUNCH(FF_F32, true); else FF_MC_LAUNCH(FF_F32, false); break;
        case FF_BF16: if (addend) FF_MC_LAUNCH(FF_BF16, true); else FF_MC_LAUNCH(FF_BF16, false); break;
        default: if (addend) FF_MC_LAUNCH(FF_F16, true); else FF_MC_LAUNCH(FF_F16, false);
    }
#undef FF_MC_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace ff

extern "C" int ff_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d,
                                int64_t L_cap, const int32_t* order, const uint8_t* member, int fold,
                                const int32_t* dst, const uint8_t* keep, const ff_aux_t* aux_host, int n_aux,
                                ff_stream_t stream) {
    if (n_aux > 0 && !keep) return FF_ERR_ARG;
    if (!hidden || !hidden_out || !member || !dst || L < 0 || d < 1 || L_cap < 0) return FF_ERR_ARG;
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
    return ff::launch_merge_compact(hidden, nullptr, hidden_out, dtype, L, d, L_cap, order, member, fold, dst, keep, aux_host, n_aux,
                                    nullptr, nullptr, nullptr,
                                    (hipStream_t)stream, false, nullptr, 0, nullptr, 0, 0, nullptr);
}

extern "C" int ff_gather_mask(const void* mask, void* out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                              const int32_t* dst, const int64_t* stats, int32_t* scratch, ff_stream_t stream) {
    if (!mask || !out || !dst || !stats || !scratch || L < 0 || L_cap < 0) return FF_ERR_ARG;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return FF_ERR_ARG;
    if (L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    if ((uintptr_t)scratch & 15) return FF_ERR_ALIGN;
    if (L == 0) return FF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ff::k_invert_dst, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, dst, (int)L, scratch);
    const char* m = (const char*)mask;
    char* o = (char*)out;
    switch ((int)elem_bytes) {
        case 1: hipLaunchKernelGGL(ff::k_gather_mask<1>, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats); break;
        case 2: hipLaunchKernelGGL(ff::k_gather_mask<2>, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats); break;
        case 4: hipLaunchKernelGGL(ff::k_gather_mask<4>, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats); break;
        default: hipLaunchKernelGGL(ff::k_gather_mask8, dim3((unsigned)L), dim3(256), 0, st, m, o, (int)L, L_cap, scratch, stats);
    }
    return (int)hipGetLastError();
}

// ---- stand-alone token gathers for the reference's public position handlers (main.py:142-178) -------------------
// position_embedding_handler_at_pruning(pe, keep_indexs): pe[..., keep_indexs, :] - rows in the order of an index
// tensor; position_embedding_handler_at_merging(pe, token_mask): pe[..., token_mask[0], :] - compaction by a boolean
// row.  The hot path never calls them (its merge kernel gathers the position tensors on the way); they exist so that
// code written against the reference's method surface keeps working.
namespace ff {
__global__ __launch_bounds__(256) void k_rows_by_index(AuxPack aux, const int64_t* __restrict__ index, int n, int L) {
    const int r = blockIdx.x * kMergeWaves * 4 + wave_id() * 4 + (lane_id() >> 4);
    const int sub = lane_id() & 15;
    if (r >= n) return;
    int64_t i = index[r];
    if (i < 0) i += L;                                       // torch indexing accepts negative indices
    if (i < 0 || i >= L) return;
    for (int x = 0; x < aux.n; ++x) {
        const ff_aux_t& ax = aux.a[x];
        for (int64_t ou = 0; ou < ax.outer; ++ou)
            copy_row((const char*)ax.src + (ou * L + i) * ax.row_bytes, (char*)ax.dst + (ou * n + r) * ax.row_bytes,
                     ax.row_bytes, sub, 16);
    }
}
__global__ __launch_bounds__(256) void k_rows_by_dst(AuxPack aux, const int32_t* __restrict__ dst, int L, int64_t L_cap) {
    const int i = blockIdx.x * kMergeWaves * 4 + wave_id() * 4 + (lane_id() >> 4);
    const int sub = lane_id() & 15;
    if (i >= L) return;
    const int r = dst[i];
    if (r < 0) return;
    for (int x = 0; x < aux.n; ++x) {
        const ff_aux_t& ax = aux.a[x];
        for (int64_t ou = 0; ou < ax.outer; ++ou)
            copy_row((const char*)ax.src + (ou * L + i) * ax.row_bytes, (char*)ax.dst + (ou * L_cap + r) * ax.row_bytes,
                     ax.row_bytes, sub, 16);
    }
}
int launch_scan_keep(const uint8_t* keep, int64_t L, int32_t* dst, int64_t* stats, hipStream_t st);
}  // namespace ff

static int pack_aux(const ff_aux_t* aux_host, int n_aux, ff::AuxPack& pack) {
    if (n_aux < 1 || n_aux > FF_MAX_AUX || !aux_host) return FF_ERR_ARG;
    pack.n = n_aux;
    for (int x = 0; x < FF_MAX_AUX; ++x) pack.a[x] = x < n_aux ? aux_host[x] : ff_aux_t{nullptr, nullptr, 0, 0};
    for (int x = 0; x < n_aux; ++x)
        if (!pack.a[x].src || !pack.a[x].dst || pack.a[x].row_bytes < 2 || (pack.a[x].row_bytes & 1) || pack.a[x].outer < 1)
            return FF_ERR_ARG;
    return FF_OK;
}

extern "C" int ff_gather_tokens_by_index(const int64_t* index, int64_t n, int64_t L, const ff_aux_t* aux_host, int n_aux,
                                         ff_stream_t stream) {
    if (n < 0 || L < 0 || (n > 0 && !index)) return FF_ERR_ARG;
    if (n >= (1ll << 31) || L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    ff::AuxPack pack;
    int rc = pack_aux(aux_host, n_aux, pack);
    if (rc) return rc;
    if (n == 0) return FF_OK;
    const unsigned blocks = (unsigned)((n + ff::kMergeWaves * 4 - 1) / (ff::kMergeWaves * 4));
    hipLaunchKernelGGL(ff::k_rows_by_index, dim3(blocks), dim3(ff::kMergeThreads), 0, (hipStream_t)stream, pack, index, (int)n, (int)L);
    return (int)hipGetLastError();
}

extern "C" int ff_gather_tokens_by_mask(const uint8_t* keep, int64_t L, int64_t L_cap, int32_t* dst, int64_t* stats,
                                        const ff_aux_t* aux_host, int n_aux, ff_stream_t stream) {
    if (!keep || !dst || !stats || L < 0 || L_cap < 0) return FF_ERR_ARG;
    if (L >= (1ll << 31) - 65536) return FF_ERR_UNSUPPORTED;
    if (((uintptr_t)keep & 15) || ((uintptr_t)dst & 15)) return FF_ERR_ALIGN;
    ff::AuxPack pack;
    int rc = pack_aux(aux_host, n_aux, pack);
    if (rc) return rc;
    if (L == 0) return FF_OK;
    rc = ff::launch_scan_keep(keep, L, dst, stats, (hipStream_t)stream);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((L + ff::kMergeWaves * 4 - 1) / (ff::kMergeWaves * 4));
    hipLaunchKernelGGL(ff::k_rows_by_dst, dim3(blocks), dim3(ff::kMergeThreads), 0, (hipStream_t)stream, pack, dst, (int)L, L_cap);
    return (int)hipGetLastError();
}
