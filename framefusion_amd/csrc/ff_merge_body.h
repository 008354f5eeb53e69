// K4's body as a device function (ff_merge.hip, k_merge_compact).
#pragma once

#include "ff_common.h"

namespace ff {

constexpr int kMergeThreads = 256;
constexpr int kMergeWaves = kMergeThreads / kWave;


struct AuxPack {
    ff_aux_t a[FF_MAX_AUX];
    int n;
};
// row i of outer slice ou of an auxiliary source holding L tokens
__device__ inline const char* aux_src_row(const ff_aux_t& ax, int64_t ou, int64_t i, int64_t L) {
    return (const char*)ax.src + ou * (ax.src_outer_bytes ? ax.src_outer_bytes : L * ax.row_bytes) + i * ax.row_bytes;
}

// Copy `bytes` from src to dst with the widest unit the alignment allows (single bytes for odd sizes: 1-byte
// position ids / boolean rows of the stand-alone gathers), spread over the threads [tid, nthreads).
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
    } else if ((al & 1) == 0) {
        for (int64_t o = (int64_t)tid * 2; o < bytes; o += (int64_t)nthreads * 2)
            *(uint16_t*)(dst + o) = *(const uint16_t*)(src + o);
    } else {
        for (int64_t o = tid; o < bytes; o += nthreads) dst[o] = src[o];
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

// ---- the short roles that ride along with a merge / prune launch (one workgroup each, by == 0 only) --------------------------
// The select tables of this call have been consumed by the plan kernel: clear them for the next call's producer.
__device__ inline void role_clear_tables(const ZeroJob& zero, int zb) {
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
}
// Auxiliary rows (position embeddings, patch types, position ids): plain compaction by SEQUENCE position - reads coalesced,
// writes in increasing order.  Workgroup `ab` owns 16 positions, 16 lanes per row.
__device__ inline void role_aux_rows(const AuxPack& aux, int ab, int L, int64_t L_cap, const uint8_t* __restrict__ keep,
                                     const int32_t* __restrict__ dst) {
    const int lane = lane_id();
    const int i = ab * kMergeWaves * 4 + wave_id() * 4 + (lane >> 4);
    const int sub = lane & 15;
    if (i >= L || !keep[i]) return;
    const int r = dst[i];
    for (int x = 0; x < aux.n; ++x) {
        const ff_aux_t& ax = aux.a[x];
        for (int64_t ou = 0; ou < ax.outer; ++ou)
            copy_row(aux_src_row(ax, ou, i, L), (char*)ax.dst + (ou * L_cap + r) * ax.row_bytes, ax.row_bytes, sub, 16);
    }
}

// (bx, by): the workgroup's coordinates in the merge kernel's own 2-D grid
template <int DT, bool kAdd>
__device__ inline void merge_compact_body(
    const char* __restrict__ hidden, const char* __restrict__ addend, char* __restrict__ out, uint32_t row_bytes, int L, int64_t L_cap,
    const int32_t* __restrict__ order, const uint8_t* __restrict__ member, int fold,
    const int32_t* __restrict__ dst, const uint8_t* __restrict__ keep, const AuxPack& aux, int n_main,
    int n_aux_blocks, int n_next_blocks, int32_t* __restrict__ order_next, int32_t* __restrict__ inv_next,
    int64_t* __restrict__ stats, const int64_t* __restrict__ identity_stats, const ZeroJob& zero, int slots,
    const int bx, const int by, const long long guard_lout = -1) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    constexpr int kDepth = 4;                     // row pieces requested per batch (two batches in flight)
    const int lane = lane_id();
    if (bx >= n_main + n_aux_blocks + n_next_blocks) {
        // ---- the select tables of this call have been consumed by the plan kernel: clear them for the
        // next call's producer (runs even when nothing is folded)
        if (by != 0) return;
        role_clear_tables(zero, bx - n_main - n_aux_blocks - n_next_blocks);
        return;
    }
    // nothing folded (a merge call whose threshold set is empty, main.py:264-266): the reduced
    // sequence IS the input, the caller keeps using its own tensors and this launch writes nothing
    if (identity_stats && identity_stats[FF_STAT_MERGED] == 0) return;
    // enqueued blind into buffers of `guard_lout` rows (exactly sized outputs allocated for the length the top-k branch gives,
    // main.py:122): if the plan decided otherwise nothing is written - the host sees the result block and repeats the launch
    if (guard_lout >= 0 && identity_stats && identity_stats[FF_STAT_LOUT] != guard_lout) return;
    if (bx >= n_main + n_aux_blocks) {
        // ---- by-patch order of the COMPACTED sequence, for the next merge call (order maintenance):
        // the surviving slots keep their relative by-patch order and dst[] is monotonic in the
        // sequence position, so new_order = dst[order[t]] compacted over the non-member slots.
        // Same communication-free scan as k_scan: this workgroup recounts the slots before its own.
        if (by != 0) return;
        __shared__ int scratch[kMergeWaves + 1];
        const int tid = threadIdx.x;
        const int base = (bx - n_main - n_aux_blocks) * (kMergeThreads * 16);
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
            stats[FF_STAT_NV] = stats[FF_STAT_NV] - merged;     // the next call (order_valid) skips K0, which would set these
            stats[FF_STAT_FTN] = stats[FF_STAT_FTN] - merged;
        }
        return;
    }
    if (bx >= n_main) {
        // ---- auxiliary rows (position embeddings, patch types, position ids): plain compaction by
        // SEQUENCE position - reads coalesced, writes in increasing order.  Only by == 0.
        if (by != 0) return;
        role_aux_rows(aux, bx - n_main, L, L_cap, keep, dst);
        return;
    }
    // slot groups are walked from the END of the by-patch order: the similarity pass read the rows
    // in ascending order, so its most recently fetched rows - the ones the 256 MiB Infinity Cache
    // still holds - are the first ones this pass asks for
    const int t0 = (n_main - 1 - bx) * slots;
    const int cb = uniform(by * kMergeWaves + wave_id());     // 1 KiB column tile
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
    // is slot s (< L) folded into its predecessor?  The plan kernel's flags.
    auto is_member = [&](int s) -> bool { return member[s] != 0; };
    const bool sl_mem = sl_ok ? is_member(sl) : false;
    unsigned long long memw = __ballot(sl_mem);
    const unsigned long long nonmem = __ballot(sl_ok && !sl_mem);

    // The stream of this wave is [bs, be): a run cannot be split between waves (the rounding after every add makes the fold
    // sequential), so the boundary between two slot groups moves to a non-member slot - the NEAREST one within `look` slots
    // before the nominal boundary x, else the first one at or after x (the rule until round 3, which made the longest
    // stream of the headline call 68 slots for a mean of 37; nearest: 53, profiles/r03_k4_probes.txt).  boundary(x) only
    // reads the flags of [x - look, ...): both neighbours of a boundary compute the same slot.
    auto first_nonmember_from = [&](int s0) {               // rare: no non-member left in the window
        for (int s = s0; s < L; s += kWave) {
            const unsigned long long b = __ballot(s + lane < L && !is_member(s + lane));
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
    const int dv = anchor_lane ? dst[ordw] : 0;               // (issue() slides the window: the anchors' rows are the first window's)

    auto piece = [&](int i) { return make_rsrc(hidden + (int64_t)i * row_bytes + col, blk_bytes); };
    auto piece2 = [&](int i) { return make_rsrc((kAdd ? addend : hidden) + (int64_t)i * row_bytes + col, blk_bytes); };

    // request the next (up to) kDepth rows of the stream
    auto issue = [&](Batch<kDepth, kAdd>& b, int pos) {
        if (pos - win > kWave - kDepth) {                   // keep kDepth slots of lookahead in the window
            win = pos;
            ordw = (pos + lane < L) ? (order ? order[pos + lane] : pos + lane) : 0;
            memw = __ballot((pos + lane < L) ? is_member(pos + lane) : false);
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

// ---- the prune's gather (main.py:78-101: hidden_states[:, keep_indexs]) by OUTPUT rows -----------------------------------------
// merge_compact_body with fold = DROP walks the INPUT slots and skips the loads of the dropped ones: at the prune's keep ratios
// (29 % at the 72B shape) a batch of 4 slots holds ~1 load, a wave has 1-2 requests in flight, and the pass ran at 0.47-0.66
// of the rate the same kernel reaches on a merge (profiles/r05_kernel_fractions.txt).  Here the plan kernel also writes
// src[] (the position of every output row: the inverse of dst[]) and a workgroup owns `rows` consecutive OUTPUT rows x 4 column
// tiles: every request is a row that is kept, two batches of kDepth pieces in flight per wave, rows written in order.
template <int DT, bool kAdd>
__device__ inline void prune_gather_body(const char* __restrict__ hidden, const char* __restrict__ addend, char* __restrict__ out,
                                         uint32_t row_bytes, int L, int l_out, const int32_t* __restrict__ src, int rows,
                                         const int bx, const int by) {
    constexpr int kDepth = kAdd ? 4 : 8;
    const int lane = lane_id();
    const int r0 = bx * rows, r1 = min(r0 + rows, l_out);
    const uint32_t col = (uint32_t)uniform(by * kMergeWaves + wave_id()) * 1024u;
    if (col >= row_bytes || r0 >= r1) return;
    const uint32_t blk_bytes = min(1024u, row_bytes - col);
    const uint32_t voff = (uint32_t)lane * 16;
    // rows <= 64: one coalesced load names every source row (clamped: a row index is never trusted with an address)
    const int mine = r0 + lane < r1 ? min(max(src[r0 + lane], 0), L - 1) : 0;
    auto in_piece = [&](const char* base, int i) { return make_rsrc(base + (int64_t)i * row_bytes + col, blk_bytes); };
    uint4 a[kDepth], b[kDepth], a2[kAdd ? kDepth : 1], b2[kAdd ? kDepth : 1];
    auto issue = [&](uint4* buf, uint4* buf2, int r) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (r + u < r1) {
                const int i = __builtin_amdgcn_readlane(mine, r + u - r0);
                buf[u] = buf_load16<2>(in_piece(hidden, i), voff);
                if constexpr (kAdd) buf2[u] = buf_load16<2>(in_piece(addend, i), voff);
            }
        }
    };
    auto store = [&](const uint4* buf, const uint4* buf2, int r) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (r + u < r1) {
                uint4 v = buf[u];
                if constexpr (kAdd) v = add16<DT>(v, buf2[u]);
                buf_store16<2>(make_rsrc(out + (int64_t)(r + u) * row_bytes + col, blk_bytes), voff, v);
            }
        }
    };
    for (int r = r0; r < r1; r += 2 * kDepth) {
        issue(a, a2, r);
        issue(b, b2, r + kDepth);
        store(a, a2, r);
        store(b, b2, r + kDepth);
    }
}

}  // namespace ff
