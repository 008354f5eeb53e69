// The merge call as ONE launch that reads the activations ONCE (round 6) - framefusion/main.py:104-138, 180-319.
//
// K1 -> plan -> K4 read every row twice (similarity pass, merge pass) and pay three launch ramps.  For the shapes the reference
// ships for - LLaVA-Video-7B 64 x 210 x 3584 bf16 = 96 MB, Qwen2-VL-7B 64 x 195 x 3584 = 90 MB - the whole activation fits into
// the chip's registers + LDS (256 CUs x (512 KiB VGPR + 160 KiB LDS) = 168 MiB).  One workgroup of 8 waves per CU owns a
// contiguous segment of <= kResRows slots of the by-patch order; wave w owns the 1 KiB column tile w of every row of the segment:
//
//   A  load + similarities: the rows are requested kResRL ahead of the arithmetic (the first kResRL by LDS-DMA into LDS, the rest
//      into VGPRs, 4 per row per wave) and STAY there.  |x|^2 and the T-rounded pair dots are summed per lane as the rows arrive
//      (recipe of ff_similarity.hip), per wave on the DPP network, across the waves through LDS; the segment's similarities are
//      published write-through and folded into the select tables (as K1's epilogue does).
//   -- ONE XCD-hierarchical grid barrier (static groups blockIdx & 7; bounded; abort flag) --
//   B  plan, by every workgroup for itself (nothing else crosses workgroups): decision, k-th key and ties from the tables
//      (ff_plan_fast.h's arithmetic) with the similarities 32 per thread in registers; member bits by slot; the member bitmap
//      by POSITION (closed form of the frame-major layout, or inv[]); two prefix scans.  Workgroup 0 publishes the result block.
//   C  fold + compaction FROM THE RESIDENT ROWS: a wave walks its rows in order, a non-member opens an output row, a member folds
//      into it (rounding after every add, main.py:304-317); a run that continues into the next workgroup's segment is finished
//      by its anchor's workgroup, which fetches those rows (L2 / Infinity Cache) - the only rows read twice.  Then the short
//      roles: non-visual rows, auxiliary rows (position tables, patch types), member / keep / dst, next order + inverse, table
//      clearing (behind a second, non-blocking count of the workgroups that have finished READING the tables).
//
// When the output buffers are too short for the result (L_cap < l_out: exactly sized outputs allocated for the top-k branch's
// length while the plan took the threshold branch) or absent, the launch stops after B with member / keep / dst in place:
// stats[FF_STAT_APPLIED] = 0 and the host follows with the merge kernel alone (ff_ctx_merge_apply).
//
// Measured on the bare pattern first (tools/resprobe, profiles/r06_resident_probe.txt): 31-33 us against 57.7 us for the three
// launches at the 7B layout.  Lesson of the probe: all loads first, arithmetic behind them, starts the arithmetic 9-12 us late -
// a CU holds far fewer requests than 8 waves x 55 KiB, so the load INSTRUCTIONS queue; hence the software pipeline.
#include <atomic>

#include "ff_common.h"
#include "ff_merge_body.h"
#include "ff_plan_fast.h"
#include "ff_resident.h"

namespace ff {

constexpr int kResThreads = 512;
constexpr int kResWaves = kResThreads / kWave;
constexpr int kResRV = 40;                       // rows of a segment held in VGPRs (4 per row and wave)
constexpr int kResRL = 16;                       // ... in LDS (the first ones; also the prefetch distance)
constexpr int kResRows = kResRV + kResRL;        // <= 62: a wave names its rows (and the one before) by lane
constexpr int kResKeys = 32;                     // similarities per thread in the plan
constexpr int kResMaxNv = kResThreads * kResKeys;        // 16 384
constexpr int kResMaxL = 2 * kResMaxNv;                  // 32 768 (two position words per thread)
constexpr int kResSlices = kResMaxNv / kSelSlice;        // 4 level-1 slices
static_assert(kResRows <= 62 && kResRV % 2 == 0 && kResRL % 2 == 0, "rows are handled in pairs, named by lane");

// barrier state, 128-byte lines, zero between launches (the last arriver of every stage resets its word)
struct ResBar {
    unsigned cnt[8][32];
    unsigned top[32];
    unsigned gen[8][32];
    unsigned abort_tag[32];
    unsigned readers[32];
};
static_assert(sizeof(ResBar) <= 4096, "ff_plan.hip reserves 4 KB of the workspace front");

struct ResArgs {
    const char* hidden;
    char* out;                    // NULL: plan only
    uint32_t row_bytes;
    int nt;                       // 1 KiB column tiles per row = data waves
    int L;
    int nv_expect, ftn_expect;    // what the host believes (hint: frames * patches; else the context's count)
    long long L_cap;
    const int64_t* ptype;
    int32_t* order;               // hint: written; else read
    int32_t* inv;
    int hint_pre, hint_patches, hint_frames;
    void* sim;
    int* l0;
    int* t16_end;
    float thr;
    PlanParams pp;
    uint8_t* member;
    uint8_t* keep;
    int32_t* dst;
    int32_t* order_next;
    int32_t* inv_next;
    int64_t* stats;
    int64_t* host_mapped;
    long long seq;
    AuxPack aux;
    ResBar* bar;
};

// sum of four floats over the wave at once (each total in lane 63): four independent DPP chains interleaved, so that no
// step waits for the wait states a DPP read of a just-written VGPR needs
__device__ inline void wave_sum4_dpp63(float& a, float& b, float& c, float& d) {
#define FF_STEP(CTRL) \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"
    asm volatile("s_nop 1\n\t"
                 FF_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 FF_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 FF_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 FF_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 FF_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 FF_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef FF_STEP
}

// LDS-DMA: 16 bytes per lane, buffer -> LDS, lane-linear destination (wave-uniform base in M0); lanes whose offset is out of
// range deliver zeros.  Invisible to the compiler's vmcnt bookkeeping: counted by hand where it is used.
// (the descriptor as four plain words: {base lo, base hi, bytes, flags of make_rsrc})
__device__ inline u32x4 raw_rsrc(const void* base, uint32_t bytes) {
    u32x4 r;
    r.x = (uint32_t)(uintptr_t)base; r.y = (uint32_t)((uintptr_t)base >> 32) & 0xffffu; r.z = bytes; r.w = 0x00020000u;
    return r;
}
__device__ inline void buf_load16_lds(u32x4 r, uint32_t voffset, uint32_t soffset, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voffset), "s"(r), "s"(soffset), "s"(lds_dst) : "memory");
}
// a 16-byte load the compiler does not count (it would wait for everything requested before it - the LDS-DMA rows included -
// at the first use): the caller waits by hand, naming the destination in the wait statement
__device__ inline void buf_load16_uncounted(u32x4& dstv, u32x4 r, uint32_t voffset, uint32_t soffset) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dstv) : "v"(voffset), "s"(r), "s"(soffset) : "memory");
}
template <int kAux = 0>
__device__ inline uint4 buf_load16s(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, (int)soffset, kAux);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int kAux = 0>
__device__ inline void buf_store16s(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset, const uint4& x) {
    u32x4 v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voffset, (int)soffset, kAux);
}

__device__ inline unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned add_agent(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-hierarchical grid barrier (MI355X_MICROARCH.md, price list row barrier-xcd).  Groups are STATIC (blockIdx & 7: the XCD a
// block is observed to run on - for speed only, nothing depends on it).  One thread per workgroup; every writing wave has drained
// its write-through stores and the workgroup has passed __syncthreads().  Returns false if the launch was aborted: a workgroup
// waited longer than ~2 ms (another barrier kernel holds CUs this one needs: two samples on two streams), everybody leaves.
__device__ inline bool res_barrier(ResBar* gb, int bid, int nwg, unsigned tag) {
    const int g = bid & 7;
    const unsigned gsize = (unsigned)((nwg + 7 - g) >> 3);
    const unsigned ngroups = (unsigned)(nwg < 8 ? nwg : 8);
    if (add_agent(&gb->cnt[g][0], 1u) + 1 == gsize) {
        st_agent(&gb->cnt[g][0], 0u);
        if (add_agent(&gb->top[0], 1u) + 1 == ngroups) {
            st_agent(&gb->top[0], 0u);
            for (unsigned x = 0; x < ngroups; ++x) st_agent(&gb->gen[x][0], tag);
        }
    }
    const long long t0 = wall_clock64();                                   // 100 MHz
    bool ok = true;
    for (unsigned spins = 0;; ++spins) {
        if (ld_agent(&gb->gen[g][0]) == tag) break;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 63u) == 63u) {
            if (ld_agent(&gb->abort_tag[0]) == tag) { ok = false; break; }
            if (wall_clock64() - t0 > 200000) { st_agent(&gb->abort_tag[0], tag); ok = false; break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// table[idx] += 1 for every lane with `valid` (idx relative to `table`, may be negative: the level-1 slices lie DOWN from the
// end of the workspace); equal indices folded into one atomic for the first kIters distinct values (ff_common.h, wave_agg_add)
template <int kIters>
__device__ inline void wave_agg_add_rel(int* table, int idx, bool valid) {
    unsigned long long rem = __ballot(valid);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        if (rem == 0ull) break;
        const int first = __ffsll((long long)rem) - 1;
        const int v = __builtin_amdgcn_readlane(idx, first);
        const unsigned long long m = __ballot(valid && idx == v) & rem;
        if (lane == first) atomicAdd(table + v, (int)__popcll(m));
        rem &= ~m;
    }
    if ((rem >> lane) & 1ull) atomicAdd(table + idx, 1);
}

constexpr size_t kResPartBytes = (size_t)(kResRows + 2) * kResWaves * 8;          // float2 [rows + 2][waves]
struct ResLds {
    // offsets into the dynamic LDS block (all multiples of 16)
    static constexpr size_t part = 0;
    static constexpr size_t simk = part + ((kResPartBytes + 15) & ~(size_t)15);      // u32 [64]: raw bits of my similarities
    static constexpr size_t sflag = simk + 64 * 4;                                  // int [64]: slot continues its predecessor's chain
    static constexpr size_t dpart = sflag + 64 * 4;                                 // int [2][256]
    static constexpr size_t drows = dpart + 2 * 256 * 4;                            // int [kResSlices][256]
    static constexpr size_t scratch = drows + kResSlices * 256 * 4;                 // int [32]
    static constexpr size_t bcast = scratch + 32 * 4;                               // int [16]
    static constexpr size_t slotmask = bcast + 16 * 4;                              // u32 [512]
    static constexpr size_t slotpre = slotmask + 512 * 4;                           // int [512]  members before the word
    static constexpr size_t posmask = slotpre + 512 * 4;                            // u32 [1024]
    static constexpr size_t pospre = posmask + 1024 * 4;                            // int [1024]
    static constexpr size_t sres = pospre + 1024 * 4;                               // int64 [32]
    static constexpr size_t rows = sres + 32 * 8;                                   // [kResRL][nt][1024]
};
__host__ __device__ constexpr size_t res_lds_bytes(int nt) { return ResLds::rows + (size_t)kResRL * nt * 1024; }

template <int DT, bool kHint>
__global__ __launch_bounds__(kResThreads) void k_merge_resident(const ResArgs a) {
    using A = Act<DT>;
    static_assert(A::kBytes == 2, "16-bit activations");
    constexpr int E = 8, RV = kResRV, RL = kResRL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* part = (float2*)(smem + ResLds::part);
    uint32_t* simk = (uint32_t*)(smem + ResLds::simk);
    int* sflag = (int*)(smem + ResLds::sflag);
    int (*dpart)[256] = (int (*)[256])(smem + ResLds::dpart);
    int* scratch = (int*)(smem + ResLds::scratch);
    int* bcast = (int*)(smem + ResLds::bcast);
    uint32_t* slotmask = (uint32_t*)(smem + ResLds::slotmask);
    int* slotpre = (int*)(smem + ResLds::slotpre);
    uint32_t* posmask = (uint32_t*)(smem + ResLds::posmask);
    int* pospre = (int*)(smem + ResLds::pospre);
    long long* sres = (long long*)(smem + ResLds::sres);
    unsigned char* lrows = smem + ResLds::rows;

    const int tid = threadIdx.x, lane = tid & 63, wv = uniform(tid >> 6);
    const int bid = blockIdx.x, G = gridDim.x;
    const int nt = a.nt, L = a.L;
    const unsigned tag = (unsigned)a.seq;
    const int F = a.hint_frames, P = a.hint_patches, pre = a.hint_pre;
    int nv, ftn;
    if constexpr (kHint) { nv = P * F; ftn = nv; }
    else { nv = (int)a.stats[FF_STAT_NV]; ftn = (int)a.stats[FF_STAT_FTN]; }
    // ---- the result block leaves through the first wave of one workgroup (word SEQ last: what the host polls)
    auto publish_words = [&](bool err_only, long long err) {      // called by the whole workgroup, sres[] complete
        __syncthreads();
        if (tid < FF_STAT_WORDS && tid != FF_STAT_SEQ) {
            long long vres = err_only ? 0 : sres[tid];
            if (tid == FF_STAT_ERROR) vres = err;
            if (!err_only && tid != FF_STAT_ERROR && tid < FF_STAT_T_ORDER) a.stats[tid] = vres;
            if (tid == FF_STAT_ERROR) a.stats[tid] = 0;                         // reported: the next call starts clean
            __hip_atomic_store(&a.host_mapped[tid], vres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tid < kWave) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (tid == 0) __hip_atomic_store(&a.host_mapped[FF_STAT_SEQ], (int64_t)a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // the host's picture of the sequence must be the device's (every workgroup sees the same words: all leave together)
    if (nv != a.nv_expect || ftn != a.ftn_expect || nv < 1) {
        if (bid == 0) publish_words(true, FF_ERR_BIT_RESIDENT);
        return;
    }
    const int s0 = (int)((long long)bid * nv / G), s1 = (int)((long long)(bid + 1) * nv / G);
    const int n = s1 - s0;                                   // <= kResRows (the launcher's arithmetic)
    const uint32_t rb = a.row_bytes;
    const uint32_t col = (uint32_t)wv * 1024u + (uint32_t)lane * 16u;
    const bool data_wave = wv < nt;
    // lanes past the end of a ragged last tile read zeros and store nothing: their offset is out of the buffer's range
    const uint32_t kDead = 0x7ffffff0u;
    const uint32_t vcol = col < rb ? col : kDead;
    const __amdgpu_buffer_rsrc_t hres = make_rsrc(a.hidden, (uint32_t)L * rb);
    const u32x4 hraw = raw_rsrc(a.hidden, (uint32_t)L * rb);
    auto lrow = [&](int i) { return (uint4*)(lrows + ((size_t)i * nt + wv) * 1024 + lane * 16); };
    // slot -> sequence position
    auto pos_hint = [&](int s) { const int p = s / F, f = s - p * F; return pre + f * P + p; };
    // order mode: lane l of every wave holds the position of slot s0 - 1 + l (l <= n); hint mode: closed form
    int ordw = 0;
    if constexpr (!kHint) {
        int j = s0 - 1 + lane;
        j = j < 0 ? 0 : (j >= nv ? nv - 1 : j);
        ordw = a.order[j];
    }
    uint4 v[RV];

    // ======================================================================================================================
    // A. rows in, similarities out
    if (data_wave && n > 0) {
        int hp = 0, hf = 0;                      // hint: patch / frame of the next row to request
        if constexpr (kHint) { hp = s0 / F; hf = s0 - hp * F; }
        auto next_off = [&]() -> uint32_t {                  // byte offset of row `issued` of my segment, in request order
            uint32_t o;
            if constexpr (kHint) {
                o = (uint32_t)(pre + hf * P + hp) * rb;
                if (++hf == F) { hf = 0; ++hp; }
            } else {
                o = 0;                                       // (order mode: by lane, below)
            }
            return o;
        };
        (void)next_off;
        auto row_off = [&](int i) -> uint32_t {              // i: static row index 0..R-1 (only called for i < n)
            if constexpr (kHint) return next_off();
            else return (uint32_t)__builtin_amdgcn_readlane(ordw, i + 1) * rb;
        };
        uint32_t prev_off;
        if constexpr (kHint) prev_off = (uint32_t)pos_hint(s0 > 0 ? s0 - 1 : 0) * rb;
        else prev_off = (uint32_t)__builtin_amdgcn_readlane(ordw, 0) * rb;
        u32x4 prevv;
        buf_load16_uncounted(prevv, hraw, vcol, prev_off);
        // every wave issues exactly 1 + RL + RV requests (rows past n: an out-of-range offset, answered with zeros without
        // traffic), so that the hand-counted waits below hold for every segment length
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const uint32_t lds = (uint32_t)(uintptr_t)(lrows + ((size_t)i * nt + wv) * 1024);
            if (i < n) buf_load16_lds(hraw, vcol, row_off(i), lds);
            else buf_load16_lds(hraw, kDead, 0u, lds);
        }
        float lastf[E];                                      // bf16: the previous row, unpacked
        uint4 lastw = make_uint4(0, 0, 0, 0);                // fp16: the previous row, raw (packed multiply)
        auto one = [&](const uint4& x, float& q, float& d) {
            q = A::sumsq(x, 0.f);
            if constexpr (DT == FF_BF16) {
                float y[E];
                A::unpack(x, y);
                d = A::dot_rounded(lastf, y, 0.f);
#pragma unroll
                for (int e = 0; e < E; ++e) lastf[e] = y[e];
            } else {
                d = A::dot_rounded_raw(lastw, x, 0.f);
                lastw = x;
            }
        };
        auto two = [&](const uint4& x0, const uint4& x1, int i) {          // rows i, i + 1 -> part[i + 1], part[i + 2]
            float qa, da, qb, db;
            one(x0, qa, da);
            one(x1, qb, db);
            wave_sum4_dpp63(qa, da, qb, db);
            if (lane == 63) { part[(i + 1) * kResWaves + wv] = make_float2(qa, da); part[(i + 2) * kResWaves + wv] = make_float2(qb, db); }
        };
        // LDS rows i, i + 1: first request the VGPR rows that take their place in the window, then wait until at most RL
        // requests are outstanding - the LDS-DMA rows behind these two plus the VGPR rows requested so far
#pragma unroll
        for (int i = 0; i < RL; i += 2) {
            if (i < RV) { if (RL + i < n) v[i] = buf_load16s(hres, vcol, row_off(RL + i)); else v[i] = buf_load16s(hres, kDead, 0u); }
            if (i + 1 < RV) { if (RL + i + 1 < n) v[i + 1] = buf_load16s(hres, vcol, row_off(RL + i + 1)); else v[i + 1] = buf_load16s(hres, kDead, 0u); }
            if (i == 0) {
                // (the row before my first slot was requested first: it is there when rows 0 and 1 are)
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(prevv) : "n"(RL) : "memory");
                const uint4 prev = make_uint4(prevv.x, prevv.y, prevv.z, prevv.w);
                if constexpr (DT == FF_BF16) A::unpack(prev, lastf);
                else lastw = prev;
                float q0 = A::sumsq(prev, 0.f), z0 = 0.f, z1 = 0.f, z2 = 0.f;
                wave_sum4_dpp63(q0, z0, z1, z2);
                if (lane == 63) part[wv] = make_float2(q0, 0.f);
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RL) : "memory");
            }
            if (i < n) two(*lrow(i), *lrow(i + 1), i);
        }
#pragma unroll
        for (int i = 0; i < RV; i += 2) {
            if (i + RL < RV) { if (RL + i + RL < n) v[i + RL] = buf_load16s(hres, vcol, row_off(RL + i + RL)); else v[i + RL] = buf_load16s(hres, kDead, 0u); }
            if (i + RL + 1 < RV) { if (RL + i + RL + 1 < n) v[i + RL + 1] = buf_load16s(hres, vcol, row_off(RL + i + RL + 1)); else v[i + RL + 1] = buf_load16s(hres, kDead, 0u); }
            if (RL + i < n) two(v[i], v[i + 1], RL + i);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): every row is on chip (and the compiler knows it)
    }
    // ---- index duties of the last wave (behind its rows when it has any): chain flags, layout check, order / inverse
    if (wv == kResWaves - 1) {
        const int j = s0 + lane;
        int flag = 0;
        if (lane < n) {
            if constexpr (kHint) {
                const int p = j / F, f = j - p * F, i = pre + f * P + p;
                if (a.ptype[i] != (int64_t)p) atomicOr((unsigned long long*)(a.stats + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_LAYOUT);
                a.order[j] = i;
                a.inv[i] = j;
                flag = f != 0;
            } else {
                flag = j > 0 && a.ptype[a.order[j - 1]] == a.ptype[a.order[j]];
            }
            sflag[lane] = flag;
        }
        if constexpr (kHint) {
            // the non-visual tail of `order` (positions in sequence order) + their type check
            const int n_tail = L - nv;
            for (int q = bid * kWave + lane; q < n_tail; q += G * kWave) {
                const int i = q < pre ? q : q + nv;
                if (a.ptype[i] != -1) atomicOr((unsigned long long*)(a.stats + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_LAYOUT);
                a.order[nv + q] = i;
                a.inv[i] = nv + q;
            }
        }
    }
    __syncthreads();
    // ---- similarities of my slots (wave 0: lane <-> slot), published write-through, folded into the select tables
    if (wv == 0) {
        const bool mine = lane < n;
        float sv = -2.0f;                                   // IGNORE_TOKEN (main.py:225-238)
        if (mine && sflag[lane]) {
            float qa = 0.f, qb = 0.f, d = 0.f;
            for (int w = 0; w < nt; ++w) {
                qa += part[lane * kResWaves + w].x;
                const float2 pb = part[(lane + 1) * kResWaves + w];
                qb += pb.x; d += pb.y;
            }
            const float na = A::rnd(sqrtf(qa)), nb = A::rnd(sqrtf(qb));
            sv = A::rnd(A::rnd(d) / A::rnd(na * nb));
        }
        uint32_t bits;
        if constexpr (DT == FF_BF16) bits = __float_as_uint(sv) >> 16;
        else { _Float16 h = (_Float16)sv; bits = (uint32_t)__builtin_bit_cast(uint16_t, h); }
        simk[lane] = mine ? bits : 0u;
        if (mine) __hip_atomic_store((uint16_t*)a.sim + (s0 + lane), (uint16_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t key = order_key<DT>(bits);
        int* tab = a.l0 + (bid & (kL0Copies - 1)) * kL0Stride;
        const int n_ge = __popcll(__ballot(mine && sv >= a.thr));
        if (lane == 0 && n_ge) atomicAdd(&tab[256], n_ge);
        wave_agg_add_rel<4>(tab, (int)(key >> 8), mine);
        const int g = (s0 + lane) / kSelSlice;
        wave_agg_add_rel<8>(a.t16_end, (int)t16_bin(key) - (g + 1) * (int)kT16SliceInts, mine);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) bcast[15] = res_barrier(a.bar, bid, G, tag) ? 1 : 0;
    __syncthreads();
    if (!bcast[15]) {
        // aborted: the select tables and the barrier words are in no defined state - the host resets the workspace and
        // repeats the call through the three launches
        publish_words(true, FF_ERR_BIT_RESIDENT);
        return;
    }

    // ======================================================================================================================
    // B. plan (every workgroup for itself)
    const PlanParams& pp = a.pp;
    const int c = tid & 255, q = tid >> 8;
    const int n_slices = (nv + kSelSlice - 1) / kSelSlice;
    // ---- round 1: tables, my 32 similarities, the error word
    int l0v[kL0Copies / 2];
#pragma unroll
    for (int x = 0; x < kL0Copies / 2; ++x) l0v[x] = a.l0[(q + x * 2) * kL0Stride + c];
    const int l0cnt_raw = a.l0[(tid & (kL0Copies - 1)) * kL0Stride + 256];
    int specv[kResSlices / 2];
    {
        const uint32_t bin = t16_bin(((uint32_t)pp.p0_guess << 8) | (uint32_t)c);
#pragma unroll
        for (int j = 0; j < kResSlices / 2; ++j) specv[j] = q + j * 2 < n_slices ? t16_slice(a.t16_end, q + j * 2)[bin] : 0;
    }
    const long long err_bits = (long long)__hip_atomic_load((unsigned long long*)(a.stats + FF_STAT_ERROR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        int colsum = 0;
#pragma unroll
        for (int x = 0; x < kL0Copies / 2; ++x) colsum += l0v[x];
        dpart[q][c] = colsum;
        if (tid < kL0Copies) scratch[tid] = l0cnt_raw;
    }
    // my 32 similarities (requested once the level-0 words have left their registers)
    // (a 16-byte buffer access that crosses the end of the range is out of range as a WHOLE: the last, partial group of 8 is
    // read element by element)
    const __amdgpu_buffer_rsrc_t sres_rsrc = make_rsrc(a.sim, (uint32_t)nv * 2u);
    uint4 kx[kResKeys / 8];
#pragma unroll
    for (int x = 0; x < kResKeys / 8; ++x) {
        const int t0 = tid * kResKeys + x * 8;
        if (t0 + 8 <= nv || t0 >= nv) {
            kx[x] = buf_load16s(sres_rsrc, (uint32_t)t0 * 2u, 0u);
        } else {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            for (int e = 0; t0 + e < nv; ++e) w[e >> 1] |= (uint32_t)((const uint16_t*)a.sim)[t0 + e] << (16 * (e & 1));
            kx[x] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();
    auto pick = [&](int rem, int& bin, int& above) {
        const int top = 255 - 4 * lane;
        int vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = dpart[0][top - e] + dpart[1][top - e];
        const int sum = vv[0] + vv[1] + vv[2] + vv[3];
        const int incl = wave_incl_scan_dpp(sum);
        const int first = __ffsll((long long)__ballot(incl >= rem)) - 1;
        int ab = incl - sum, b = top;
        if (ab + vv[0] >= rem) { b = top; }
        else if (ab + vv[0] + vv[1] >= rem) { ab += vv[0]; b = top - 1; }
        else if (ab + vv[0] + vv[1] + vv[2] >= rem) { ab += vv[0] + vv[1]; b = top - 2; }
        else { ab += vv[0] + vv[1] + vv[2]; b = top - 3; }
        bin = __builtin_amdgcn_readlane(b, first);
        above = __builtin_amdgcn_readlane(ab, first);
    };
    if (wv == 0) {
        int cnt = lane < kL0Copies ? scratch[lane] : 0;
        cnt = __builtin_amdgcn_readlane(wave_incl_scan_dpp(cnt), 63);
        bool topk;
        long long k;
        if (pp.k_given >= 0) {
            topk = true;                                  // fixed-sparsity policy (modeling_qwen2_baseline.py:920,1001)
            k = pp.k_given > nv ? (long long)nv : pp.k_given;
        } else {
            // main.py:114-116 in double, as python: ratio = count / ftn ; ratio < sub ?
            const double ratio = ftn > 0 ? (double)cnt / (double)ftn : 0.0;
            topk = !(ratio < pp.sub);
            k = 0;
            if (topk) {
                k = (long long)(pp.sub * (double)ftn);   // int(sub * ftn), main.py:122
                if (k > nv) k = nv;
                if (k < 0) k = 0;
            }
        }
        int bin = 0, above = 0;
        if (topk && k > 0) pick((int)k, bin, above);
        if (lane == 0) { bcast[0] = topk ? 1 : 0; bcast[1] = cnt; bcast[2] = (int)k; bcast[3] = bin; bcast[4] = (int)k - above; bcast[9] = -1; }
    }
    __syncthreads();
    const bool is_topk = bcast[0] != 0;
    const int count = bcast[1], k_sel = bcast[2];
    const bool topk = is_topk && k_sel > 0;
    uint32_t kth = 0;
    int need = 0, tstar = -1;
    // my 32 keys as masks (bit e <-> slot 32 * tid + e)
    const int t_base = tid * kResKeys;
    auto key_of = [&](int e) -> uint32_t {
        const uint4& w4 = kx[e >> 3];
        const uint32_t w = ((e >> 1) & 3) == 0 ? w4.x : ((e >> 1) & 3) == 1 ? w4.y : ((e >> 1) & 3) == 2 ? w4.z : w4.w;
        return order_key<DT>((w >> (16 * (e & 1))) & 0xffffu);
    };
    if (topk) {                                              // (uniform)
        const int p0 = bcast[3];
        int colsum = 0;
#pragma unroll
        for (int j = 0; j < kResSlices / 2; ++j) {
            int x = 0;
            if (q + j * 2 < n_slices) x = p0 == pp.p0_guess ? specv[j] : t16_slice(a.t16_end, q + j * 2)[t16_bin(((uint32_t)p0 << 8) | (uint32_t)c)];
            colsum += x;
        }
        dpart[q][c] = colsum;                                // (level 0's sums have been consumed: wave 0 only, before the barrier above)
        __syncthreads();
        if (wv == 0) {
            int p1, above;
            pick(bcast[4], p1, above);
            if (lane == 0) { bcast[5] = (p0 << 8) | p1; bcast[6] = bcast[4] - above; }
        }
        __syncthreads();
        kth = (uint32_t)bcast[5];
        need = bcast[6];
    }
    // the level-0 / level-1 words I needed are in registers or LDS now
    if (tid == 0) add_agent(&a.bar->readers[0], 1u);
    uint32_t eqm = 0, gtm = 0, thm = 0;
#pragma unroll
    for (int e = 0; e < kResKeys; ++e) {
        const uint32_t key = key_of(e);
        const bool in = t_base + e < nv;
        eqm |= (uint32_t)(in & (key == kth)) << e;
        gtm |= (uint32_t)(in & (key > kth)) << e;
        thm |= (uint32_t)(in & (key >= pp.thr_key) & (key != nan_key<DT>())) << e;
    }
    if (topk) {
        // t*: the slot of the need-th entry equal to the k-th key (ties taken in ascending by-patch position)
        const int mine = __popc(eqm);
        int total;
        const int ex = block_excl_scan<kResWaves>(mine, scratch, total);
        if (ex < need && need <= ex + mine) {
            uint32_t m = eqm;
            for (int x = ex + 1; x < need; ++x) m &= m - 1;        // drop the lowest set bit need - ex - 1 times
            bcast[9] = t_base + (__ffs((int)m) - 1);
        }
        __syncthreads();
        tstar = bcast[9];
    }
    uint32_t mm;
    if (is_topk) {
        mm = 0;
        if (topk) {
            const int upto = tstar - t_base;                // ties at slots <= t* are taken
            const uint32_t tie_take = upto >= 31 ? 0xffffffffu : upto < 0 ? 0u : ((2u << upto) - 1u);
            mm = gtm | (eqm & tie_take);
        }
    } else {
        mm = thm;
    }
    if (tid == 0) mm &= ~1u;                                // slot 0 never folds
    slotmask[tid] = mm;
    __syncthreads();
    // ---- member bits by POSITION: thread t owns positions [ppt * t, ppt * (t + 1)), ppt = 32 or 64
    const int words = L > kResMaxNv ? 2 : 1;
    uint32_t pm0 = 0u, pm1 = 0u;
    {
        const int i0 = tid * 32 * words;
        if constexpr (kHint) {
            int x = i0 - pre;                               // index inside the visual span
            int f = 0, p = 0, s = 0;                        // frame / patch / slot of position x (x <= 0: of the span's start)
            if (x > 0 && x < nv) { f = x / P; p = x - f * P; s = p * F + f; }
            auto word = [&]() {
                uint32_t m = 0;
#pragma unroll 8
                for (int e = 0; e < 32; ++e) {
                    if (x >= 0 && x < nv) {
                        m |= ((slotmask[s >> 5] >> (s & 31)) & 1u) << e;
                        s += F;
                        if (++p == P) { p = 0; ++f; s = f; }
                    }
                    ++x;
                }
                return m;
            };
            pm0 = word();
            if (words == 2) pm1 = word();
        } else {
            const __amdgpu_buffer_rsrc_t irs = make_rsrc(a.inv, (uint32_t)L * 4u);
            auto word = [&](int ib) {
                uint32_t m = 0;
#pragma unroll 2
                for (int ch = 0; ch < 8; ++ch) {
                    const int ic = ib + ch * 4;
                    uint32_t sl[4];
                    if (ic + 4 <= L || ic >= L) {           // (past L: zeros, masked below; the partial group: element by element)
                        const uint4 a0 = buf_load16s(irs, (uint32_t)ic * 4u, 0u);
                        sl[0] = a0.x; sl[1] = a0.y; sl[2] = a0.z; sl[3] = a0.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sl[e] = ic + e < L ? (uint32_t)a.inv[ic + e] : 0u;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = ib + ch * 4 + e;
                        const uint32_t s = sl[e];
                        const uint32_t bit = (i < L && s < (uint32_t)nv) ? (slotmask[s >> 5] >> (s & 31)) & 1u : 0u;
                        m |= bit << (ch * 4 + e);
                    }
                }
                return m;
            };
            pm0 = word(i0);
            if (words == 2) pm1 = word(i0 + 32);
        }
    }
    // ---- two exclusive scans in one: members before my slot word (high half) / before my position words (low half)
    int members_total;
    {
        const int packed = (__popc(mm) << 16) | (__popc(pm0) + __popc(pm1));
        int total;
        const int ex = block_excl_scan<kResWaves>(packed, scratch, total);
        slotpre[tid] = ex >> 16;
        const int pb = ex & 0xffff;
        if (words == 1) { posmask[tid] = pm0; pospre[tid] = pb; }
        else { posmask[2 * tid] = pm0; pospre[2 * tid] = pb; posmask[2 * tid + 1] = pm1; pospre[2 * tid + 1] = pb + __popc(pm0); }
        members_total = total >> 16;
        if (members_total != (total & 0xffff)) members_total = -1;          // (the two views of the member set disagree: reported)
    }
    __syncthreads();
    const int l_out = L - (members_total < 0 ? 0 : members_total);
    const bool plan_bad = members_total < 0 || (err_bits != 0);
    const bool apply = !plan_bad && a.out != nullptr && a.L_cap >= (long long)l_out;
    auto members_before_pos = [&](int i) { return pospre[i >> 5] + __popc(posmask[i >> 5] & ((1u << (i & 31)) - 1u)); };
    auto members_before_slot = [&](int t) { return slotpre[t >> 5] + __popc(slotmask[t >> 5] & ((1u << (t & 31)) - 1u)); };
    if (bid == 0) {
        if (tid < FF_STAT_WORDS) {
            long long vres = 0;
            const double ratio = ftn > 0 ? (double)count / (double)ftn : 0.0;
            switch (tid) {
                case FF_STAT_NV: vres = nv; break;
                case FF_STAT_FTN: vres = ftn; break;
                case FF_STAT_COUNT: vres = count; break;
                case FF_STAT_BRANCH: vres = is_topk ? 1 : 0; break;
                case FF_STAT_K: vres = k_sel; break;
                case FF_STAT_MERGED: vres = L - l_out; break;
                case FF_STAT_LOUT: vres = l_out; break;
                case FF_STAT_BELOW_LB: vres = (!is_topk && ratio < pp.ratio_lb) ? 1 : 0; break;
                case FF_STAT_KTH_KEY: vres = kth; break;
                case FF_STAT_TIES_TAKEN: vres = topk ? need : 0; break;
                case FF_STAT_APPLIED: vres = apply ? 1 : 0; break;
                default: break;
            }
            sres[tid] = vres;
        }
        long long e = err_bits;
        if (members_total < 0) e |= FF_ERR_BIT_RESIDENT;
        publish_words(e != 0, e);
    }
    if (plan_bad) return;                                   // (the host resets the workspace)

    // ---- member / keep / dst: the plan's arrays, a slice per workgroup (the merge kernel that follows a plan-only launch, the
    // attention-mask gather and the diagnostics read them)
    {
        const int b0 = (int)((long long)bid * L / G), b1 = (int)((long long)(bid + 1) * L / G);
        for (int i = b0 + tid; i < b1; i += kResThreads) {
            const uint32_t mbit = (posmask[i >> 5] >> (i & 31)) & 1u;
            a.keep[i] = (uint8_t)(mbit ^ 1u);
            a.dst[i] = mbit ? -1 : i - members_before_pos(i);
            a.member[i] = i < nv ? (uint8_t)((slotmask[i >> 5] >> (i & 31)) & 1u) : (uint8_t)0;
        }
    }
    const bool folded = l_out != L;
    auto readers_leave = [&]() {                            // second count: the last workgroup through resets the word for the next launch
        if (add_agent(&a.bar->readers[0], 1u) + 1 == 2u * (unsigned)G) st_agent(&a.bar->readers[0], 0u);
    };
    if (!apply) {                                           // plan only: ff_ctx_merge_apply follows (it also clears the tables)
        if (tid == 0) readers_leave();
        return;
    }

    // ======================================================================================================================
    // C. fold + compaction from the resident rows
    if (folded && data_wave) {
        const __amdgpu_buffer_rsrc_t ores = make_rsrc(a.out, (uint32_t)(a.L_cap * (long long)rb));
        // member bits of my slots and of the 64 behind them; output row of every anchor (by lane)
        const int js = s0 + lane;
        const bool mbit = lane < n && ((slotmask[js >> 5] >> (js & 31)) & 1u);
        const unsigned long long memw = __ballot(mbit);
        int dv = 0;
        if (lane < n && !mbit) {
            int i;
            if constexpr (kHint) i = pos_hint(js);
            else i = a.order[js];
            dv = i - members_before_pos(i);
        }
        float acc[E];
        int open_r = -1, open_n = 0;
        auto flush = [&]() {
            float o[E];
            const uint32_t off = (uint32_t)open_r * rb;
            if (open_n > 0) {
                const float div = A::rnd((float)(open_n + 1));
                if constexpr (DT == FF_BF16) {
                    // T(a / div) == T(a * RN(1 / div)) for every bf16-valued a and divisor T(k): ff_merge_body.h
                    const float r = 1.0f / div;
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = acc[e] * r;
                    buf_store16s<2>(ores, vcol, off, A::pack_rne(o));
                    return;
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = A::rnd(acc[e] / div);
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = acc[e];
            }
            buf_store16s<2>(ores, vcol, off, A::pack(o));
        };
        auto take = [&](const uint4& x, int i) {            // i: static row index
            if (!((memw >> i) & 1ull)) {
                if (open_r >= 0) flush();
                open_r = __builtin_amdgcn_readlane(dv, i);
                open_n = 0;
                A::unpack(x, acc);
            } else if (open_r >= 0) {                        // (leading members belong to the previous workgroup's run)
                float y[E];
                A::unpack(x, y);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = A::rnd(acc[e] + y[e]);
                ++open_n;
            }
        };
#pragma unroll 4
        for (int i = 0; i < RL; ++i)
            if (i < n) take(*lrow(i), i);
#pragma unroll
        for (int i = 0; i < RV; ++i)
            if (RL + i < n) take(v[i], RL + i);
        // the open run may go on in the following segments: those rows come from L2 / the Infinity Cache
        if (open_r >= 0) {
            for (int t = s1; t < nv;) {
                const int tt = t + lane;
                const bool mb = tt < nv && ((slotmask[tt >> 5] >> (tt & 31)) & 1u);
                const unsigned long long mw = __ballot(mb);
                const int run = mw == ~0ull ? kWave : __ffsll((long long)~mw) - 1;      // leading members of this window
                int iw = 0;
                if (lane < run) {
                    if constexpr (kHint) iw = pos_hint(tt);
                    else iw = a.order[tt];
                }
                for (int u = 0; u < run; u += 2) {
                    uint4 x[2];
#pragma unroll
                    for (int z = 0; z < 2; ++z)
                        if (u + z < run) x[z] = buf_load16s(hres, vcol, (uint32_t)__builtin_amdgcn_readlane(iw, u + z) * rb);
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        if (u + z < run) {
                            float y[E];
                            A::unpack(x[z], y);
#pragma unroll
                            for (int e = 0; e < E; ++e) acc[e] = A::rnd(acc[e] + y[e]);
                            ++open_n;
                        }
                    }
                }
                if (run < kWave) break;
                t += kWave;
            }
            flush();
        }
        // non-visual rows (kept as they are): row q of the order's tail goes to workgroup q mod G
        const int n_tail = L - nv;
        for (int qq = bid; qq < n_tail; qq += G) {
            int i;
            if constexpr (kHint) i = qq < pre ? qq : qq + nv;
            else i = a.order[nv + qq];
            const uint4 x = buf_load16s<2>(hres, vcol, (uint32_t)i * rb);
            buf_store16s<2>(ores, vcol, (uint32_t)(i - members_before_pos(i)) * rb, x);
        }
    }
    if (folded) {
        // ---- auxiliary rows (position tables, patch types): 16 lanes per kept position of my slice
        if (a.aux.n > 0) {
            const int b0 = (int)((long long)bid * L / G), b1 = (int)((long long)(bid + 1) * L / G);
            for (int i = b0 + (tid >> 4); i < b1; i += kResThreads / 16) {
                if ((posmask[i >> 5] >> (i & 31)) & 1u) continue;
                const int r = i - members_before_pos(i);
                for (int x = 0; x < a.aux.n; ++x) {
                    const ff_aux_t& ax = a.aux.a[x];
                    for (int64_t ou = 0; ou < ax.outer; ++ou)
                        copy_row(aux_src_row(ax, ou, i, L), (char*)ax.dst + (ou * a.L_cap + r) * ax.row_bytes, ax.row_bytes, tid & 15, 16);
                }
            }
        }
        // ---- by-patch order of the compacted sequence + its inverse (the next merge call skips K0)
        if (a.order_next) {
            const int b0 = (int)((long long)bid * L / G), b1 = (int)((long long)(bid + 1) * L / G);
            for (int t = b0 + tid; t < b1; t += kResThreads) {
                int rank, i;
                if (t < nv) {
                    if ((slotmask[t >> 5] >> (t & 31)) & 1u) continue;
                    rank = t - members_before_slot(t);
                    if constexpr (kHint) i = pos_hint(t);
                    else i = a.order[t];
                } else {
                    rank = t - members_total;
                    if constexpr (kHint) i = (t - nv) < pre ? (t - nv) : t;
                    else i = a.order[t];
                }
                const int np = i - members_before_pos(i);
                a.order_next[rank] = np;
                a.inv_next[np] = rank;
            }
        }
    }
    // ---- clear the select tables for the next call - once every workgroup has READ them (counted above; nobody waits for
    // long: the readers finished a fold phase ago)
    bool clear_ok = true;
    if (tid == 0) {
        for (unsigned spins = 0; ld_agent(&a.bar->readers[0]) < (unsigned)G; ++spins) {
            __builtin_amdgcn_s_sleep(2);
            if (spins > (1u << 16)) { clear_ok = false; break; }
        }
        bcast[14] = clear_ok ? 1 : 0;
    }
    __syncthreads();
    if (bcast[14]) {
        if (bid < kL0Copies)
            for (int z = tid; z < kL0Stride; z += kResThreads) a.l0[bid * kL0Stride + z] = 0;
        if (tid < n) {
            const uint32_t key = order_key<DT>(simk[tid]);
            t16_slice(a.t16_end, (s0 + tid) / kSelSlice)[t16_bin(key)] = 0;
        }
    } else if (tid == 0) {
        atomicOr((unsigned long long*)(a.stats + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_RESIDENT);       // (seen by the next call)
    }
    __syncthreads();
    if (tid == 0) {
        // second count: the last workgroup through resets the word for the next launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        readers_leave();
        if (bid == 0 && folded && a.order_next) {
            a.stats[FF_STAT_NV] = nv - (L - l_out);          // the next call (order_valid) skips K0, which would set these
            a.stats[FF_STAT_FTN] = ftn - (L - l_out);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static int res_cus() {
    static std::atomic<int> cache[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < kMaxDevices) {
        const int got = cache[dev].load(std::memory_order_relaxed);
        if (got > 0) return got;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const int cus = prop.multiProcessorCount;
    if (dev >= 0 && dev < kMaxDevices) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// Does a merge call of this shape run as the one-launch kernel?  `nv`: visual tokens as the host knows them (<= 0: unknown).
bool merge_resident_fits(int dtype, int64_t L, int64_t d, int64_t nv, bool addend, int fold) {
    // (bf16 only: the fp16 fold keeps its eight IEEE divisions per flush - ff_merge_body.h - and does not fit next to 160 pinned VGPRs)
    if (dtype != FF_BF16) return false;
    if (addend || fold != FF_FOLD_SEQUENTIAL) return false;
    const int64_t rb = d * 2;
    if (rb < 16 || rb > 8 * 1024 || (rb & 15)) return false;
    if (nv < 1 || nv > kResMaxNv || nv > L || L > kResMaxL) return false;
    if (L * rb >= (1ll << 31)) return false;
    const int cus = res_cus();
    if (cus < 8) return false;
    return (nv + cus - 1) / cus <= kResRows;
}

ResBar* ws_resbar(void* ws);

PlanParams merge_plan_params(int dtype, double thr, double sub, double ratio_lb, long long force_k);
int* ws_l0(void* ws);
int* ws_t16_end(void* ws, size_t ws_bytes);

template <int DT, bool kHint>
static int launch_res(const ResArgs& a, int cus, hipStream_t st) {
    static std::atomic<bool> attr_set[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
    if (dev < 0 || !attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)k_merge_resident<DT, kHint>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)res_lds_bytes(8));
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_merge_resident<DT, kHint>), dim3((unsigned)cus), dim3(kResThreads), res_lds_bytes(a.nt), st, a);
    return (int)hipGetLastError();
}

int launch_merge_resident(const ResLaunch& p, hipStream_t st) {
    const int cus = res_cus();
    if (cus < 8) return FF_ERR_UNSUPPORTED;
    ResArgs a;
    a.hidden = (const char*)p.hidden;
    a.out = (char*)p.hidden_out;
    a.row_bytes = (uint32_t)(p.d * 2);
    a.nt = (int)((p.d * 2 + 1023) / 1024);
    a.L = (int)p.L;
    a.nv_expect = (int)p.nv;
    a.ftn_expect = (int)p.ftn;
    a.L_cap = p.hidden_out ? p.L_cap : 0;
    a.ptype = p.ptype;
    a.order = p.order;
    a.inv = p.inv;
    a.hint_pre = (int)p.hint_pre; a.hint_patches = (int)p.hint_patches; a.hint_frames = (int)p.hint_frames;
    a.sim = p.sim;
    a.l0 = ws_l0(p.ws);
    a.t16_end = ws_t16_end(p.ws, p.ws_bytes);
    a.thr = (float)p.thr;
    a.pp = merge_plan_params(p.dtype, p.thr, p.sub, p.ratio_lb, p.force_k);
    a.pp.n_slices = (int)((p.nv + kSelSlice - 1) / kSelSlice);
    a.member = p.member; a.keep = p.keep; a.dst = p.dst;
    a.order_next = p.order_next; a.inv_next = p.inv_next;
    a.stats = p.stats; a.host_mapped = p.host_mapped; a.seq = p.seq;
    a.aux.n = p.hidden_out ? p.n_aux : 0;
    for (int x = 0; x < FF_MAX_AUX; ++x) a.aux.a[x] = x < a.aux.n ? p.aux[x] : ff_aux_t{nullptr, nullptr, 0, 0, 0};
    a.bar = ws_resbar(p.ws);
    const bool hint = p.hint_frames > 0;
    return hint ? launch_res<FF_BF16, true>(a, cus, st) : launch_res<FF_BF16, false>(a, cus, st);
}

}  // namespace ff
