// K4's body as a device function: run by the stand-alone merge kernel (ff_merge.hip, k_merge_compact) and by the
// fused plan + merge launch (ff_fused.hip).
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

// Fused launch (ff_fused.hip): the workgroups of this body share a grid with the plan's.  Two hand-overs, both through
// 64 replicated lines of 128 bytes in the workspace (2 048 pollers on ONE word would queue up on one memory channel; a
// workgroup polls the line its index selects):
//  * the DECISION (line words 1, 2), published by plan workgroup 0 as soon as it knows the k-th key and the tie slot t*
//    - two thirds into the plan.  "Is slot t folded?" is a pure function of (similarity[t], t) and the decision, so a main
//    workgroup derives its member flags from the similarities itself and gets its first row requests out while the
//    plan is still scanning;
//  * DONE (line word 0 = the call's sequence number), raised by the last plan workgroup to arrive: dst[] (needed by the
//    first store), keep / stats (auxiliary rows, next order) and the tables (clearing) wait for this one.
// Words carry a tag derived from the context's strictly increasing sequence number, so nothing is ever cleared.
struct FusedWait {
    const unsigned long long* flags;       // 64 lines, 16 words apart: [done | decision 0 | decision 1 | ...]
    unsigned long long seq;
    int64_t* stats;                        // FF_STAT_ERROR gets FF_ERR_BIT_BARRIER if a flag never comes
    const void* sim;                       // the similarities the decision applies to (activation dtype)
    uint32_t thr_key;
    int dbg;
    long long* dbg_buf;                    // (FF_FUSED_DBG bit 2: per-workgroup phase stamps)
};
constexpr int kFlagCopies = 64, kFlagStride = 16;

// decision word 0: tag << 35 | topk (k > 0) << 34 | is_topk << 33 | kth key << 17 | (t* + 1);  word 1: tag << 35 | nv
__device__ inline unsigned long long decision_tag(unsigned long long seq) { return seq % ((1ull << 29) - 1ull) + 1ull; }
struct Decision {
    uint32_t kth;
    int tstar, nv;
    bool is_topk, topk;
};
__device__ inline void publish_decision(unsigned long long* flags, unsigned long long seq, bool is_topk, bool topk, uint32_t kth,
                                        int tstar, int nv) {          // one wave; lane = copy
    const unsigned long long tag = decision_tag(seq) << 35;
    unsigned long long* line = flags + (size_t)(threadIdx.x & (kFlagCopies - 1)) * kFlagStride;
    __hip_atomic_store(line + 1, tag | ((unsigned long long)topk << 34) | ((unsigned long long)is_topk << 33) |
                                 ((unsigned long long)(kth & 0xffffu) << 17) | (unsigned long long)(uint32_t)(tstar + 1),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(line + 2, tag | (unsigned long long)(uint32_t)nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void flag_timeout(const FusedWait& fw) {
    atomicOr((unsigned long long*)(fw.stats + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_BARRIER);
}

// whole workgroup (call before the waves diverge): thread 0 polls, the words travel through LDS
__device__ inline Decision wait_for_decision(const FusedWait& fw, int bx) {
    __shared__ unsigned long long words[2];
    if (threadIdx.x == 0) {
        const unsigned long long* line = fw.flags + (size_t)(bx & (kFlagCopies - 1)) * kFlagStride;
        const unsigned long long tag = decision_tag(fw.seq);
        unsigned long long w0 = 0, w1 = 0;
        __builtin_amdgcn_s_sleep(100);             // (~3 us: the decision takes the plan longer than that from its first instruction)
        for (int spins = 0;; ++spins) {
            w0 = __hip_atomic_load(line + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1 = __hip_atomic_load(line + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w0 >> 35) == tag && (w1 >> 35) == tag) break;
            __builtin_amdgcn_s_sleep(2);
            if (spins > (1 << 20)) { flag_timeout(fw); w0 = w1 = 0; break; }       // (an empty decision: nothing folds)
        }
        words[0] = w0; words[1] = w1;
    }
    __syncthreads();
    const unsigned long long w0 = words[0], w1 = words[1];
    Decision d;
    d.tstar = (int)(w0 & 0x1ffffull) - 1;
    d.kth = (uint32_t)(w0 >> 17) & 0xffffu;
    d.is_topk = (w0 >> 33) & 1ull;
    d.topk = (w0 >> 34) & 1ull;
    d.nv = (int)(w1 & 0x7ffffffull);
    return d;
}

// one wave (uniform): DONE seen?  false after the time-out
__device__ inline bool wave_wait_done(const FusedWait& fw, int bx) {
    const unsigned long long* f = fw.flags + (size_t)(bx & (kFlagCopies - 1)) * kFlagStride;
    for (int spins = 0;; ++spins) {
        const unsigned long long v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((int)(v == fw.seq))) return true;
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1 << 20)) {
            if (lane_id() == 0) flag_timeout(fw);
            return false;
        }
    }
}

// whole workgroup: DONE (the roles that consume keep / dst / stats or clear the tables).
// No acquire fence behind it: `buffer_inv sc1` by every wave of a streaming kernel cost 100 us at 64 x 576 x 4096 (it empties
// the L2 under the streams, again and again).  It is not needed either: the kernel started with clean caches, and of
// everything the plan writes (member, dst, keep - whole cache lines per plan workgroup - and stats) only `stats` has been
// READ in this launch before the flag (by the plan workgroups), so only its line can sit stale in an XCD's L2: stats is
// read with agent-scope loads (stat_word), everything else misses and comes from memory, where the plan's release put it.
__device__ inline void wait_for_plan(const FusedWait& fw, int bx, bool nap = true) {
    if (threadIdx.x == 0) {
        const unsigned long long* f = fw.flags + (size_t)(bx & (kFlagCopies - 1)) * kFlagStride;
        if (nap) __builtin_amdgcn_s_sleep(100);        // (~3 us: what is waited for takes longer than that)
        for (int spins = 0;; ++spins) {
            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fw.seq) break;
            __builtin_amdgcn_s_sleep(2);
            if (spins > (1 << 20)) { flag_timeout(fw); break; }       // the plan never finished: report, do not hang
        }
    }
    __syncthreads();
    if (fw.dbg & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// a word of the result block: past the (non-coherent) L2 in the fused launch
template <bool kFused>
__device__ inline int64_t stat_word(const int64_t* stats, int which) {
    if constexpr (kFused) return __hip_atomic_load(stats + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return stats[which];
}

// (bx, by): the workgroup's coordinates in the merge kernel's own 2-D grid
template <int DT, bool kAdd, bool kFused>
__device__ inline void merge_compact_body(
    const char* __restrict__ hidden, const char* __restrict__ addend, char* __restrict__ out, uint32_t row_bytes, int L, int64_t L_cap,
    const int32_t* __restrict__ order, const uint8_t* __restrict__ member, int fold,
    const int32_t* __restrict__ dst, const uint8_t* __restrict__ keep, const AuxPack& aux, int n_main,
    int n_aux_blocks, int n_next_blocks, int32_t* __restrict__ order_next, int32_t* __restrict__ inv_next,
    int64_t* __restrict__ stats, const int64_t* __restrict__ identity_stats, const ZeroJob& zero, int slots,
    const int bx, const int by, const FusedWait& fw) {
    using A = Act<DT>;
    constexpr int E = A::kPer16;
    constexpr int kDepth = 4;                     // row pieces requested per batch (two batches in flight)
    const int lane = lane_id();
    if constexpr (kFused) {
        // every role but the main one starts with the plan's outputs (or clears what the plan reads)
        if (bx >= n_main) {
            if (by != 0) return;
            wait_for_plan(fw, bx);
        }
    }
    if (bx >= n_main + n_aux_blocks + n_next_blocks) {
        // ---- the select tables of this call have been consumed by the plan kernel: clear them for the
        // next call's producer (runs even when nothing is folded)
        if (by != 0) return;
        const int zb = bx - n_main - n_aux_blocks - n_next_blocks;
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
    if (!(kFused && bx < n_main) && identity_stats && stat_word<kFused>(identity_stats, FF_STAT_MERGED) == 0) return;
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
            const int64_t merged = stat_word<kFused>(stats, FF_STAT_MERGED);
            stats[FF_STAT_NV] = stat_word<kFused>(stats, FF_STAT_NV) - merged;     // the next call (order_valid) skips K0, which would set these
            stats[FF_STAT_FTN] = stat_word<kFused>(stats, FF_STAT_FTN) - merged;
        }
        return;
    }
    if (bx >= n_main) {
        // ---- auxiliary rows (position embeddings, patch types, position ids): plain compaction by
        // SEQUENCE position - reads coalesced, writes in increasing order.  Only by == 0.
        if (by != 0) return;
        const int i = (bx - n_main) * kMergeWaves * 4 + wave_id() * 4 + (lane >> 4);
        const int sub = lane & 15;                       // 16 lanes per row
        if (i >= L || !keep[i]) return;
        const int r = dst[i];
        for (int x = 0; x < aux.n; ++x) {
            const ff_aux_t& ax = aux.a[x];
            for (int64_t ou = 0; ou < ax.outer; ++ou)
                copy_row(aux_src_row(ax, ou, i, L), (char*)ax.dst + (ou * L_cap + r) * ax.row_bytes, ax.row_bytes, sub, 16);
        }
        return;
    }
    // slot groups are walked from the END of the by-patch order: the similarity pass read the rows
    // in ascending order, so its most recently fetched rows - the ones the 256 MiB Infinity Cache
    // still holds - are the first ones this pass asks for
    const int t0 = (n_main - 1 - bx) * slots;
    const int cb = uniform(by * kMergeWaves + wave_id());     // 1 KiB column tile
    const uint32_t col = (uint32_t)cb * 1024u;
    if (!kFused && col >= row_bytes) return;
    const uint32_t blk_bytes = col < row_bytes ? min(1024u, row_bytes - col) : 0u;
    const uint32_t voff = (uint32_t)lane * 16;
    const int t_end = min(t0 + slots, L);

    // window of 64 slots: row indices and member flags, starting `look` slots BEFORE the nominal range
    const int look = min(10, (kWave - slots) / 2);          // slots <= 53 (merge_slots): look >= 5
    const int win0 = t0 - look;
    int win = win0;
    const int sl = win0 + lane;
    const bool sl_ok = sl >= 0 && sl < L;
    int ordw = sl_ok ? (order ? order[sl] : sl) : 0;
    Decision dec{0u, -1, 0, false, false};
    if constexpr (kFused) {
        // (the order window above does not depend on the plan: its load is in flight while the workgroup waits)
        dec = wait_for_decision(fw, bx);
        if ((fw.dbg & 4) && threadIdx.x == 0) fw.dbg_buf[3 * (by * n_main + bx)] = wall_clock64();
        if (col >= row_bytes) return;
    }
    // is slot s (< L) folded into its predecessor?  Stand-alone: the plan kernel's flags.  Fused: the plan's rule
    // (plan_fast_body, `folded`) applied to the similarity itself - the flags may not have been written yet.
    auto is_member = [&](int s) -> bool {
        if constexpr (kFused && DT != FF_F32) {
            const uint32_t key = order_key<DT>(A::bits1(fw.sim, s));
            const bool in = s < dec.nv;
            const bool sel_topk = dec.topk & in & ((key > dec.kth) | ((key == dec.kth) & (s <= dec.tstar)));
            const bool sel_thr = in & (key >= fw.thr_key) & (key != 0xffffu);
            return (dec.is_topk ? sel_topk : sel_thr) & (s > 0);
        } else {
            return member[s] != 0;
        }
    };
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
    const int ordw0 = ordw;                                  // (issue() slides the window: the anchors' rows are the first window's)
    int dv = (!kFused && anchor_lane) ? dst[ordw] : 0;

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
    bool have_b1 = false;
    if constexpr (kFused) {
        // both batches are requested; the output rows (dst) are the plan's last product: wait for DONE now
        if (!b0.last) { issue(b1, pos); pos += b1.take; have_b1 = true; }
        if (!wave_wait_done(fw, bx)) return;
        if (identity_stats && stat_word<kFused>(identity_stats, FF_STAT_MERGED) == 0) return;       // nothing folded: nothing written
        dv = anchor_lane ? dst[ordw0] : 0;
    }
    while (true) {
        if (!b0.last && !have_b1) { issue(b1, pos); pos += b1.take; }
        have_b1 = false;
        fold_batch(b0);
        if (b0.last) break;
        if (!b1.last) { issue(b0, pos); pos += b0.take; }
        fold_batch(b1);
        if (b1.last) break;
    }
    if (open_r >= 0) flush();
    if constexpr (kFused) {
        if ((fw.dbg & 4) && threadIdx.x == 0) { fw.dbg_buf[3 * (by * n_main + bx) + 1] = wall_clock64(); fw.dbg_buf[3 * (by * n_main + bx) + 2] = (long long)(be - bs) << 32; }
    }
}

}  // namespace ff
