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
//      published write-through.
//   -- ONE XCD-hierarchical grid barrier (static groups blockIdx & 7; bounded; abort flag).  Its arrival words also CARRY two
//      counts per workgroup - similarities in the threshold set (main.py:113), similarities above IGNORE_TOKEN - and the release
//      word hands the two totals to everybody: they ARE the decision (main.py:114-127) and the output length, so the RESULT
//      BLOCK LEAVES HERE (~22 us after the kernel's start at the 7B layout, ~12 us before the plan is through) --
//   B  plan, by every workgroup for itself (nothing else crosses workgroups; the select tables of the three-launch path are not
//      touched): all similarities 32 per thread in registers, level 0 speculated (counts above / at the guessed top byte), one
//      256-bin histogram in LDS for the low byte, ties with ff_plan_fast.h's arithmetic; member bits by slot; the member bitmap
//      by POSITION (closed form of the frame-major layout, or inv[]); two prefix scans.  The plan's member count is checked
//      against the published length (an internal assertion: device error word).
//   C  outputs + fold + compaction FROM THE RESIDENT ROWS.  The outputs come in the launch arguments or BY MAIL (two slots in the
//      pinned block, relayed into device memory by one wave): slot 1 is written by the host while the rows are being read
//      (input-length buffers, or a guessed length), slot 2 after it has read the result block (exactly l_out rows) - the kernel
//      waits for whichever holds the result, rows in hand, and acknowledges it (FF_STAT_ACK); a host that does not answer within
//      ~4 ms gets the plan only and follows with the merge kernel (ff_ctx_merge_apply).  Then a wave walks its rows in order, a
//      non-member opens an output row, a member folds into it (rounding after every add, main.py:304-317); a run that continues
//      into the next workgroup's segment is finished by its anchor's workgroup, which fetches those rows (L2 / Infinity Cache) -
//      the only rows read twice.  The short roles: non-visual rows, auxiliary rows (position tables, patch types: before the
//      fold when there are at most three (tensor, slice) pairs, else handed out by an LDS counter, the wave without rows first),
//      member / keep / dst, next order + inverse.
//
// Fold while waiting: a data wave that finds no mail with the result at the end of the plan does the fold's arithmetic at once (pass 1:
// finished rows parked in the home of each run's last row) and only stores when the outputs are known (pass 2); see C. below.
//
// A third instance (kAdd) takes call B of a decoder layer with the residual add fused in: its rows are the sums T(hidden + addend),
// two requests per row through registers the compiler counts (phase A below); everything behind phase A sees resident sums.
//
// Measured (tools/flow_stamps.py --wg on a library built with EXTRA=-DFF_RES_WGSTAMPS: first workgroup start -> last workgroup
// end on the device clock; rocprofv3's kernel duration includes the wait for the host's mail, and its tracing slows the host):
// 53-61 us at the 7B layout (13 474 -> 4 066) against 57.7 us for the three launches; the host sees the result 22 us after
// the kernel's start instead of 31 us, and a call whose guessed output length was wrong costs no second launch.
// History of a mismeasurement: until the per-workgroup stamps existed the kernel was timed by workgroup 0's FIRST wave
// ("48 us"); the wave without rows did all auxiliary rows alone and took until 75 us.
//
// The bare pattern was measured first (tools/resprobe, profiles/r06_resident_probe.txt): 31-33 us against 57.7 us for the three
// launches at the 7B layout.  Lesson of the probe: all loads first, arithmetic behind them, starts the arithmetic 9-12 us late -
// a CU holds far fewer requests than 8 waves x 55 KiB, so the load INSTRUCTIONS queue; hence the software pipeline.
#include <atomic>
#include <type_traits>

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
static_assert(kResRows <= 62 && kResRV % 2 == 0 && kResRL % 2 == 0, "rows are handled in pairs, named by lane");

// barrier state, 128-byte lines, zero between launches (the last arriver of every stage resets its word)
struct ResBar {
    // arrival words: arrivals in bits 0-11, and what the arrivers bring along - the number of their similarities in the threshold
    // set (bits 12-31) and above IGNORE_TOKEN (bits 32-51): the release word hands the two totals to everybody
    unsigned long long cnt[8][16];
    unsigned long long top[16];
    unsigned long long gen[8][16];       // release word: tag (24 bits, never 0) | threshold-set total (20) | above-ignore total (20)
    unsigned abort_tag[32];
    unsigned long long mail_tag[16];     // outputs by mail, relayed into device memory by ONE wave: 4 * seq + slot (3: none came) ...
    unsigned long long mail[32];         // ... and the slot's FF_MAIL_WORDS words {., hidden_out, L_cap, n_aux, aux[4]}
};
static_assert(sizeof(ResBar) <= 4096, "ff_plan.hip reserves 4 KB of the workspace front");

struct ResArgs {
    const char* hidden;
    const char* addend;           // kAdd: the rows are T(hidden + addend) (the decoder's residual add, modeling_qwen2.py:64-67)
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
    const int64_t* mail;          // != NULL: the outputs + auxiliary tensors come by mail (two slots of FF_MAIL_WORDS pinned host words)
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
__device__ inline unsigned long long ld_agent64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long add_agent64(unsigned long long* p, unsigned long long v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Returns the release word's two totals (threshold set << 20 | above ignore), or -1 if the launch was aborted.
__device__ inline long long res_barrier(ResBar* gb, int bid, int nwg, unsigned tag, int thr_n, int gt_n) {
    const int g = bid & 7;
    const unsigned gsize = (unsigned)((nwg + 7 - g) >> 3);
    const unsigned ngroups = (unsigned)(nwg < 8 ? nwg : 8);
    const unsigned long long tag24 = (unsigned long long)(tag % 0xffffffu) + 1ull;
    const unsigned long long mine = 1ull | ((unsigned long long)thr_n << 12) | ((unsigned long long)gt_n << 32);
    const unsigned long long now = add_agent64(&gb->cnt[g][0], mine) + mine;
    if ((unsigned)(now & 0xfffull) == gsize) {
        st_agent64(&gb->cnt[g][0], 0ull);
        const unsigned long long up = (now & ~0xfffull) | 1ull;
        const unsigned long long t = add_agent64(&gb->top[0], up) + up;
        if ((unsigned)(t & 0xfffull) == ngroups) {
            st_agent64(&gb->top[0], 0ull);
            const unsigned long long word = (tag24 << 40) | (((t >> 12) & 0xfffffull) << 20) | ((t >> 32) & 0xfffffull);
            for (unsigned x = 0; x < ngroups; ++x) st_agent64(&gb->gen[x][0], word);
        }
    }
    const long long t0 = wall_clock64();                                   // 100 MHz
    bool ok = true;
    unsigned long long w = 0;
    for (unsigned spins = 0;; ++spins) {
        w = ld_agent64(&gb->gen[g][0]);
        if ((w >> 40) == tag24) break;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 63u) == 63u) {
            if (ld_agent(&gb->abort_tag[0]) == tag) { ok = false; break; }
            if (wall_clock64() - t0 > 200000) { st_agent(&gb->abort_tag[0], tag); ok = false; break; }
        }
    }
    // (no acquire fence: everything read across workgroups behind the barrier - the similarities, the error word - was stored
    // write-through / by an atomic and is loaded with agent-scope loads, which do not look at the CU's L1; an L1 invalidation
    // costs ~1.5 us per workgroup here)
    return ok ? (long long)(w & 0xffffffffffull) : -1ll;
}

// raw bits of a T value -> order-preserving 16-bit key (negative values inverted, the others get the sign bit; NaN -> 0xffff)
__device__ inline uint32_t res_key16(uint32_t bits) {
    uint32_t k = (bits & 0x8000u) ? (~bits & 0xffffu) : (bits | 0x8000u);
    if ((bits & 0x7fffu) > 0x7f80u) k = 0xffffu;
    return k;
}
constexpr uint32_t kResIgnoreKey = 0x3fffu;      // key of IGNORE_TOKEN = -2.0 (main.py:225-238) in bf16

constexpr size_t kResPartBytes = (size_t)(kResRows + 2) * kResWaves * 8;          // float2 [rows + 2][waves]
struct ResLds {
    // offsets into the dynamic LDS block (all multiples of 16)
    static constexpr size_t part = 0;
    static constexpr size_t simk = part + ((kResPartBytes + 15) & ~(size_t)15);      // u32 [64]: raw bits of my similarities
    static constexpr size_t sflag = simk + 64 * 4;                                  // int [64]: slot continues its predecessor's chain
    static constexpr size_t dpart = sflag + 64 * 4;                                 // int [256 + 8 * 256 + 8]: the plan's histograms
    static constexpr size_t scratch = dpart + (256 + 8 * 256 + 8) * 4;              // int [32]
    static constexpr size_t bcast = scratch + 32 * 4;                               // int [16]
    static constexpr size_t slotmask = bcast + 16 * 4;                              // u32 [512 + 8]: word w at w + (w >> 6)
    static constexpr size_t slotpre = slotmask + 528 * 4;                           // int [512]  members before the word
    static constexpr size_t posmask = slotpre + 512 * 4;                            // u32 [1024]
    static constexpr size_t pospre = posmask + 1024 * 4;                            // int [1024]
    static constexpr size_t sres = pospre + 1024 * 4;                               // int64 [32]
    static constexpr size_t rows = sres + 32 * 8;                                   // [kResRL][nt][1024]
};
__host__ __device__ constexpr size_t res_lds_bytes(int nt) { return ResLds::rows + (size_t)kResRL * nt * 1024; }

template <int DT, bool kHint, bool kAdd>
__global__ __launch_bounds__(kResThreads) void k_merge_resident(const ResArgs a) {
    static_assert(!(kHint && kAdd), "rows that are sums (a fused residual add) come with a maintained order: call B follows call A");
    using A = Act<DT>;
    static_assert(A::kBytes == 2, "16-bit activations");
    constexpr int E = 8, RV = kResRV, RL = kResRL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* part = (float2*)(smem + ResLds::part);
    int* sflag = (int*)(smem + ResLds::sflag);
    int* hist0 = (int*)(smem + ResLds::dpart);                  // [256] level 0 | [8][256] level 1 | [8] counters
    int* hist1 = hist0 + 256;
    int* hcnt = hist1 + 8 * 256;
    int* scratch = (int*)(smem + ResLds::scratch);
    int* bcast = (int*)(smem + ResLds::bcast);
    uint32_t* slotmask_ = (uint32_t*)(smem + ResLds::slotmask);
    // (one pad word per 64: threads 32 positions apart read words F apart - all 64 lanes one bank at F = 64 without it)
    auto slotword = [&](int w) -> uint32_t& { return slotmask_[w + (w >> 6)]; };
    auto slotbit = [&](int t) -> uint32_t { return (slotword(t >> 5) >> (t & 31)) & 1u; };
    int* slotpre = (int*)(smem + ResLds::slotpre);
    uint32_t* posmask = (uint32_t*)(smem + ResLds::posmask);
    int* pospre = (int*)(smem + ResLds::pospre);
    long long* sres = (long long*)(smem + ResLds::sres);
    unsigned char* lrows = smem + ResLds::rows;

    const int tid = threadIdx.x, lane = tid & 63, wv = uniform(tid >> 6);
    const int bid = blockIdx.x, G = gridDim.x;
    const int nt = a.nt, L = a.L;
    const unsigned tag = (unsigned)a.seq;
    // phase stamps of workgroup 0 (100 MHz wall clock, relative to its start): stats[FF_STAT_T_PLAN ..], diagnostics
    const long long stamp0 = wall_clock64();
#ifdef FF_RES_WGSTAMPS
    // (diagnostic build, tools/flow_stamps.py --wg: earliest / latest workgroup start and end on the device's 100 MHz clock, in the
    // spare words of the barrier page)
    unsigned long long* wgdbg = (unsigned long long*)a.bar + 384;
    if (tid == 0) {
        atomicMax(&wgdbg[0], ~0ull - (unsigned long long)stamp0);
        atomicMax(&wgdbg[1], (unsigned long long)stamp0);
        if (bid == 0) wgdbg[4] = (unsigned long long)stamp0;
    }
#endif
    long long stamp[7] = {0, 0, 0, 0, 0, 0, 0};
    long long sub[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // finer stamps (stats[FF_STAT_T_ORDER ..])
    const int F = a.hint_frames, P = a.hint_patches, pre = a.hint_pre;
    int nv, ftn;
    if constexpr (kHint) { nv = P * F; ftn = nv; }
    else { nv = (int)a.stats[FF_STAT_NV]; ftn = (int)a.stats[FF_STAT_FTN]; }
    // ---- the result block leaves through the first wave of one workgroup (word SEQ last: what the host polls)
    auto publish_words = [&](bool err_only, long long err) {      // called by the whole workgroup, sres[] complete
        __syncthreads();
        if (tid < FF_STAT_T_ORDER && tid != FF_STAT_SEQ) {             // (the words behind are the host's: FF_MAIL_WORD)
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
        if constexpr (!kAdd) buf_load16_uncounted(prevv, hraw, vcol, prev_off);
        // every wave issues exactly 1 + RL + RV requests (rows past n: an out-of-range offset, answered with zeros without
        // traffic), so that the hand-counted waits below hold for every segment length
        if constexpr (!kAdd) {
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const uint32_t lds = (uint32_t)(uintptr_t)(lrows + ((size_t)i * nt + wv) * 1024);
                if (i < n) buf_load16_lds(hraw, vcol, row_off(i), lds);
                else buf_load16_lds(hraw, kDead, 0u, lds);
            }
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
        if constexpr (kAdd) {
            // Rows that are SUMS, T(hidden + addend): two requests per row, kW rows in flight, everything through registers
            // the compiler counts (no LDS-DMA: the raw halves would have to meet in LDS).  The addend halves wait in ta[], the
            // hidden halves of the rows whose home is LDS in th[] - those of the VGPR rows arrive in their home registers, which
            // are free until then; a row's sum goes home (LDS / its registers) and into the norms and dots at once.
            constexpr int kW = 8;
            static_assert(kW <= RL && RL % kW == 0, "the first window is LDS-home rows only");
            const __amdgpu_buffer_rsrc_t ares = make_rsrc(a.addend, (uint32_t)L * rb);
            uint4 th[kW], ta[kW];
            const uint4 ph = buf_load16s(hres, vcol, prev_off), pa = buf_load16s(ares, vcol, prev_off);
#pragma unroll
            for (int w = 0; w < kW; ++w) {
                const uint32_t vo = w < n ? vcol : kDead, so = w < n ? row_off(w) : 0u;
                th[w] = buf_load16s(hres, vo, so);
                ta[w] = buf_load16s(ares, vo, so);
            }
            {
                const uint4 prev = add16<DT>(ph, pa);
                if constexpr (DT == FF_BF16) A::unpack(prev, lastf);
                else lastw = prev;
                float q0 = A::sumsq(prev, 0.f), z0 = 0.f, z1 = 0.f, z2 = 0.f;
                wave_sum4_dpp63(q0, z0, z1, z2);
                if (lane == 63) part[wv] = make_float2(q0, 0.f);
            }
#pragma unroll
            for (int i = 0; i < RL + RV; i += 2) {
                uint4 x[2];
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const int r = i + z;
                    if (r < RL) { x[z] = add16<DT>(th[r % kW], ta[r % kW]); *lrow(r) = x[z]; }
                    else { v[r - RL] = add16<DT>(v[r - RL], ta[r % kW]); x[z] = v[r - RL]; }
                    const int q = r + kW;                    // the row that takes the places just freed
                    if (q < RL + RV) {
                        const uint32_t vo = q < n ? vcol : kDead, so = q < n ? row_off(q) : 0u;
                        if (q < RL) th[q % kW] = buf_load16s(hres, vo, so);
                        else v[q - RL] = buf_load16s(hres, vo, so);
                        ta[q % kW] = buf_load16s(ares, vo, so);
                    }
                }
                if (i < n) two(x[0], x[1], i);
            }
        } else {
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
    stamp[0] = wall_clock64() - stamp0;                       // rows in, norms and dots done
    // ---- similarities of my slots (wave 0: lane <-> slot), published write-through
    const PlanParams& pp = a.pp;
    int my_thr = 0, my_gt = 0;
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
        const uint32_t bits = __float_as_uint(sv) >> 16;
        if (mine) __hip_atomic_store((uint16_t*)a.sim + (s0 + lane), (uint16_t)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // what the barrier carries to everybody: how many of my similarities are in the threshold set (main.py:113) and how many
        // are real ones (above IGNORE_TOKEN) - the two totals ARE the decision and the output length (below)
        const uint32_t key = res_key16(bits);
        my_thr = __popcll(__ballot(mine && (key - pp.thr_key) < (0xffffu - pp.thr_key)));
        my_gt = __popcll(__ballot(mine && key > kResIgnoreKey));
    } else {
        // (the plan's histograms start from zero)
        for (int z = tid - kWave; z < 256 + 8 * 256 + 8; z += kResThreads - kWave) hist0[z] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const long long totals = res_barrier(a.bar, bid, G, tag, my_thr, my_gt);
        bcast[15] = totals >= 0 ? 1 : 0;
        bcast[8] = (int)((totals >> 20) & 0xfffff);
        bcast[9] = (int)(totals & 0xfffff);
    }
    __syncthreads();
    if (!bcast[15]) {
        // aborted: the select tables and the barrier words are in no defined state - the host resets the workspace and
        // repeats the call through the three launches
        publish_words(true, FF_ERR_BIT_RESIDENT);
        return;
    }

    // ======================================================================================================================
    // B. plan (every workgroup for itself)
    stamp[1] = wall_clock64() - stamp0;                       // barrier passed
    // ---- my 32 similarities (bit e <-> slot 32 * tid + e) as order-preserving keys, two per word
    // (a 16-byte buffer access that crosses the end of the range is out of range as a WHOLE, so the range is rounded up to
    // whole words: the context's `sim` holds 4 bytes per token of capacity; what lies past nv is masked by `valid`)
    const int t_base = tid * kResKeys;
    const __amdgpu_buffer_rsrc_t sres_rsrc = make_rsrc(a.sim, ((uint32_t)nv * 2u + 15u) & ~15u);
    uint4 kx[kResKeys / 8];
#pragma unroll
    for (int x = 0; x < kResKeys / 8; ++x) kx[x] = buf_load16s<16>(sres_rsrc, (uint32_t)(t_base + x * 8) * 2u, 0u);      // (aux 16 = sc1: agent scope)
    const long long err_bits = (long long)__hip_atomic_load((unsigned long long*)(a.stats + FF_STAT_ERROR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- the decision (main.py:114-127) and the output length need nothing but the two totals the barrier brought: every wave
    // works them out for itself (the same scalars from the same LDS words), and the result block leaves NOW - the host learns
    // l_out ~13 us before the plan below is through and prepares what follows the call under it
    const int count = uniform(bcast[8]), n_real = uniform(bcast[9]);
    bool is_topk;
    int k_sel;
    {
        long long k;
        if (pp.k_given >= 0) {
            is_topk = true;                               // fixed-sparsity policy (modeling_qwen2_baseline.py:920,1001)
            k = pp.k_given > nv ? (long long)nv : pp.k_given;
        } else {
            // main.py:114-116 in double, as python: ratio = count / ftn ; ratio < sub ?
            const double ratio = ftn > 0 ? (double)count / (double)ftn : 0.0;
            is_topk = !(ratio < pp.sub);
            k = 0;
            if (is_topk) {
                k = (long long)(pp.sub * (double)ftn);   // int(sub * ftn), main.py:122
                if (k > nv) k = nv;
                if (k < 0) k = 0;
            }
        }
        k_sel = uniform((int)k);
    }
    const bool topk = is_topk && k_sel > 0;
    // slot 0 never folds (it has no predecessor; its similarity is IGNORE_TOKEN, the FIRST of the ties at that value): it is in
    // the top-k set iff k exceeds the real similarities, in the threshold set iff the threshold lies below IGNORE_TOKEN
    const bool slot0_thr = (kResIgnoreKey - pp.thr_key) < (0xffffu - pp.thr_key);
    const int members_e = topk ? k_sel - (k_sel > n_real ? 1 : 0) : (is_topk ? 0 : count - (slot0_thr ? 1 : 0));
    const int l_out = L - members_e;
    const long long tag_mail = (long long)a.seq * 4;          // + slot (1, 2): a mail of this call; + 3: none came in time
    if (bid == 0 && wv == kResWaves - 1) {
        // the result block leaves through the last wave of workgroup 0: one lane per word.  APPLIED = 2: the outputs come by mail
        // and this launch waits for them behind its plan (ff_ctx_merge_apply confirms: word FF_STAT_ACK)
        const long long e = err_bits;
        if (lane < FF_STAT_T_ORDER && lane != FF_STAT_SEQ) {            // (the words behind are the host's: FF_MAIL_WORD)
            long long vres = 0;
            const double ratio = ftn > 0 ? (double)count / (double)ftn : 0.0;
            switch (lane) {
                case FF_STAT_NV: vres = nv; break;
                case FF_STAT_FTN: vres = ftn; break;
                case FF_STAT_COUNT: vres = count; break;
                case FF_STAT_BRANCH: vres = is_topk ? 1 : 0; break;
                case FF_STAT_K: vres = k_sel; break;
                case FF_STAT_MERGED: vres = L - l_out; break;
                case FF_STAT_LOUT: vres = l_out; break;
                case FF_STAT_BELOW_LB: vres = (!is_topk && ratio < pp.ratio_lb) ? 1 : 0; break;
                case FF_STAT_APPLIED: vres = l_out == L ? 1 : (a.mail ? 2 : ((a.out != nullptr && a.L_cap >= (long long)l_out) ? 1 : 0)); break;
                default: break;
            }
            if (e) vres = 0;
            if (lane == FF_STAT_ERROR) vres = e;
            if (!e && lane != FF_STAT_ERROR && lane != FF_STAT_ACK) a.stats[lane] = vres;
            if (lane == FF_STAT_ERROR) a.stats[lane] = 0;                       // reported: the next call starts clean
            if (lane != FF_STAT_ACK) __hip_atomic_store(&a.host_mapped[lane], vres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (lane == 0) __hip_atomic_store(&a.host_mapped[FF_STAT_SEQ], (int64_t)a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (err_bits) return;                                     // (the host resets the workspace and repeats the call)
    sub[3] = wall_clock64() - stamp0;                         // result block on its way
    // ---- outputs by mail: ONE wave of the launch reads the host's words (a device read of host memory is a PCIe round trip,
    // ~2.5 us, and the link takes only a few dozen of them at a time: every workgroup asking for itself cost 270 us) and relays
    // them into device memory, where everybody picks them up behind the plan.  Two slots, each written once per call (words,
    // then its tag): the first for outputs allocated before the result was known (a guessed length), the second for outputs
    // sized to the l_out published above.  The first mail that holds l_out rows is taken and acknowledged.
    const bool wants_mail = a.mail != nullptr && l_out != L;
    const bool relay_wave = wants_mail && bid == G - 1 && wv == kResWaves - 1;
    // one look at the host's slots (0: nothing usable there, 1: relayed + acknowledged, 2: slot 1 is there and does not fit)
    auto relay_try = [&](bool skip1) -> int {
        long long t = 0;
        if (lane < 2) t = __hip_atomic_load(&a.mail[lane * FF_MAIL_WORDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t1 = __shfl(t, 0), t2 = __shfl(t, 1);
        const int slot = t2 == tag_mail + 2 ? 2 : ((t1 == tag_mail + 1 && !skip1) ? 1 : 0);
        if (!slot) return 0;
        long long w = 0;
        if (lane < FF_MAIL_WORDS) w = __hip_atomic_load(&a.mail[(slot - 1) * FF_MAIL_WORDS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long ptr = __shfl(w, 1), cap = __shfl(w, 2);
        // slot 1 was written before the result was known: taken when it holds exactly l_out rows (a guessed length that came
        // true) or a whole input; slot 2 is sized to the result
        if (!ptr || (slot == 2 ? cap < (long long)l_out : (cap != (long long)l_out && cap < (long long)L))) return slot == 1 ? 2 : 0;
        if (lane < FF_MAIL_WORDS) __hip_atomic_store(&a.bar->mail[lane], (unsigned long long)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            __hip_atomic_store(&a.bar->mail_tag[0], (unsigned long long)(tag_mail + slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.host_mapped[FF_STAT_ACK], (int64_t)(tag_mail + slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return 1;
    };
    int relay_state = 0;
    if (relay_wave) relay_state = relay_try(false);                            // (one look now, under the plan; the patient ones behind it)
    const uint32_t valid = nv - t_base >= 32 ? 0xffffffffu : (nv - t_base <= 0 ? 0u : (1u << (nv - t_base)) - 1u);
    // raw bits -> keys, both halves of a word at once: negative values are inverted, the others get the sign bit; NaN -> 0xffff
    auto keys2 = [&](uint32_t w) -> uint32_t {
        const uint32_t neg = (w >> 15) & 0x00010001u;
        uint32_t k2 = w ^ (((neg << 16) - neg) | 0x80008000u);
        const uint32_t mag = w & 0x7fff7fffu;
        if ((mag & 0xffffu) > 0x7f80u) k2 |= 0x0000ffffu;
        if ((mag >> 16) > 0x7f80u) k2 |= 0xffff0000u;
        return k2;
    };
    uint32_t kk[kResKeys / 2];
#pragma unroll
    for (int x = 0; x < kResKeys / 8; ++x) {
        kk[4 * x] = keys2(kx[x].x); kk[4 * x + 1] = keys2(kx[x].y); kk[4 * x + 2] = keys2(kx[x].z); kk[4 * x + 3] = keys2(kx[x].w);
    }
    auto key_of = [&](int e) -> uint32_t { return (e & 1) ? kk[e >> 1] >> 16 : kk[e >> 1] & 0xffffu; };
    // (the masks below are built in four independent pieces: one OR chain of 32 links is latency, not work, with two waves per SIMD)
    // threshold set (main.py:113): key >= thr_key and not NaN, as ONE unsigned compare
    uint32_t thm;
    {
        const uint32_t span = 0xffffu - pp.thr_key;
        uint32_t t4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < kResKeys; ++e) t4[e & 3] |= (uint32_t)((key_of(e) - pp.thr_key) < span) << e;
        thm = ((t4[0] | t4[1]) | (t4[2] | t4[3])) & valid;
    }
    // ---- level 0, speculated: the k-th similarity of a video sits in the binade of typical thresholds (pp.p0_guess, as the
    // three-launch plan guesses): count my keys whose top byte is ABOVE / EQUAL to the guess - both halves of a word at once -
    // and let the totals say whether the guess holds.  (A 256-bin histogram of top bytes in LDS costs microseconds here: a
    // video's similarities share two or three top bytes and same-address LDS atomics serialise.)
    {
        const uint32_t g = (uint32_t)pp.p0_guess;
        int c_ge = 0, c_eq = 0;
        if (valid == 0xffffffffu) {
            const uint32_t G1 = g * 0x00010001u, G8 = g * 0x01000100u;
            int ce[2] = {0, 0}, cg[2] = {0, 0};
#pragma unroll
            for (int x = 0; x < kResKeys / 2; ++x) {
                const uint32_t tb = kk[x] & 0xff00ff00u;
                const uint32_t z = tb ^ G8;                                              // a half is zero <=> its top byte == g
                ce[x & 1] += __popc((z - 0x00010001u) & ~z & 0x80008000u);
                cg[x & 1] += __popc((((tb >> 8) | 0x01000100u) - G1) & 0x01000100u);     // bit 8 of 0x100 + byte - g
            }
            c_eq = ce[0] + ce[1]; c_ge = cg[0] + cg[1];
        } else {
#pragma unroll
            for (int e = 0; e < kResKeys; ++e) {
                const uint32_t tb = key_of(e) >> 8, ok = (valid >> e) & 1u;
                c_eq += (int)(ok & (uint32_t)(tb == g));
                c_ge += (int)(ok & (uint32_t)(tb >= g));
            }
        }
        float r0 = (float)__popc(thm), r1 = (float)(c_ge - c_eq), r2 = (float)c_eq, r3 = 0.f;   // (exact: sums below 2^24)
        wave_sum4_dpp63(r0, r1, r2, r3);
        if (lane == 63) { atomicAdd(&hcnt[0], (int)r0); atomicAdd(&hcnt[1], (int)r1); atomicAdd(&hcnt[2], (int)r2); }
    }
    __syncthreads();                                         // #1
    // pick: the bin of the rem-th largest entry of a 256-bin histogram (+ more copies of it), and the entries above it - by one wave
    auto pick = [&](const int* h, int copies, int rem, int& bin, int& above) {
        const int top = 255 - 4 * lane;
        int vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vv[e] = h[top - e];
            for (int x = 1; x < copies; ++x) vv[e] += h[x * 256 + top - e];
        }
        const int sum = vv[0] + vv[1] + vv[2] + vv[3];
        const int incl = wave_incl_scan_dpp(sum);
        const int first = __ffsll((long long)__ballot(incl >= rem)) - 1;
        int ab = incl - sum, bsel = top;
        if (ab + vv[0] >= rem) { bsel = top; }
        else if (ab + vv[0] + vv[1] >= rem) { ab += vv[0]; bsel = top - 1; }
        else if (ab + vv[0] + vv[1] + vv[2] >= rem) { ab += vv[0] + vv[1]; bsel = top - 2; }
        else { ab += vv[0] + vv[1] + vv[2]; bsel = top - 3; }
        bin = __builtin_amdgcn_readlane(bsel, first);
        above = __builtin_amdgcn_readlane(ab, first);
    };
    uint32_t kth = 0;
    int need = 0;
    uint32_t mm;
    if (topk) {                                              // (uniform)
        int p0 = pp.p0_guess, rem = k_sel - uniform(hcnt[1]);
        if (!(rem >= 1 && rem <= uniform(hcnt[2]))) {
            // the guess failed (rare): the 256-bin histogram after all - runs of equal top bytes counted in registers, one LDS
            // atomic per change
            int cur = -1, run = 0;
#pragma unroll
            for (int e = 0; e < kResKeys; ++e) {
                const int tb = ((valid >> e) & 1u) ? (int)(key_of(e) >> 8) : -1;
                if (tb != cur) {
                    if (cur >= 0) atomicAdd(&hist0[cur], run);
                    cur = tb; run = 0;
                }
                ++run;
            }
            if (cur >= 0) atomicAdd(&hist0[cur], run);
            __syncthreads();
            int above;
            pick(hist0, 1, k_sel, p0, above);
            rem = k_sel - above;
        }
        // level 1: the low bytes of the keys whose top byte is the k-th key's, 8 copies against same-bin collisions
        int* h1 = hist1 + (lane & 7) * 256;
#pragma unroll
        for (int e = 0; e < kResKeys; ++e) {
            const uint32_t key = key_of(e);
            atomicAdd(&h1[key & 0xffu], (int)(((valid >> e) & 1u) & (uint32_t)((key >> 8) == (uint32_t)p0)));
        }
        __syncthreads();                                     // #2
        if (tid < 256) {
            int sum8 = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x) sum8 += hist1[x * 256 + tid];
            hist0[tid] = sum8;                               // (level 0's bins are not needed any more)
        }
        __syncthreads();                                     // #2b
        int p1, above1;
        pick(hist0, 1, rem, p1, above1);
        kth = ((uint32_t)p0 << 8) | (uint32_t)p1;
        need = rem - above1;
        stamp[2] = wall_clock64() - stamp0;                   // decision + k-th key known
        uint32_t e4[4] = {0u, 0u, 0u, 0u}, g4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < kResKeys; ++e) {
            const uint32_t key = key_of(e);
            e4[e & 3] |= (uint32_t)(key == kth) << e;
            g4[e & 3] |= (uint32_t)(key > kth) << e;
        }
        const uint32_t eqm = ((e4[0] | e4[1]) | (e4[2] | e4[3])) & valid, gtm = ((g4[0] | g4[1]) | (g4[2] | g4[3])) & valid;
        sub[0] = wall_clock64() - stamp0;                     // key masks
        // ties at the k-th value are taken in ascending by-patch position: I take those of mine whose rank among all ties is
        // below `need`
        const int mine = __popc(eqm);
        const int wincl = wave_incl_scan_dpp(mine);
        if (lane == 63) scratch[wv] = wincl;
        __syncthreads();                                     // #3
        int ex = wincl - mine;
#pragma unroll
        for (int x = 0; x < kResWaves; ++x) ex += x < wv ? scratch[x] : 0;
        const int take_n = need - ex;
        uint32_t tie = take_n >= mine ? eqm : 0u;
        if (take_n > 0 && take_n < mine) {
            uint32_t m = eqm;
            for (int x = 0; x < take_n; ++x) { tie |= m & (0u - m); m &= m - 1; }        // the lowest take_n set bits
        }
        mm = gtm | tie;
    } else {
        stamp[2] = wall_clock64() - stamp0;
        mm = is_topk ? 0u : thm;
    }
    if (tid == 0) mm &= ~1u;                                // slot 0 never folds
    sub[1] = wall_clock64() - stamp0;                         // member bits of my slots
    slotword(tid) = mm;
    __syncthreads();                                         // #4
    // ---- member bits by POSITION: position i = 512 e + tid in round e, so that the lanes of a wave look at 64 consecutive
    // positions (their slots lie F / 32 words apart: no bank pile-up) and ONE ballot is two words of the position bitmap
    {
        const int rounds = (L + kResThreads - 1) / kResThreads;
        const __amdgpu_buffer_rsrc_t irs = make_rsrc(a.inv, (uint32_t)L * 4u);
        int pf = 0, pp_ = 0, q512 = 0, r512 = 0, shift_f = 0;
        if constexpr (kHint) {
            // x = i - pre = f * P + p inside the visual span; kept non-negative by `shift_f` whole frames for the division
            shift_f = (pre + P - 1) / P;
            const unsigned xx = (unsigned)(tid - pre + shift_f * P);
            pf = (int)(xx / (unsigned)P); pp_ = (int)(xx - (unsigned)pf * (unsigned)P);
            q512 = kResThreads / P; r512 = kResThreads - q512 * P;
        }
        auto slot_of = [&](int e, bool& in) -> int {      // the slot of position 512 e + tid (0 if it has none)
            const int i = e * kResThreads + tid;
            int sl;
            if constexpr (kHint) {
                in = (unsigned)(i - pre) < (unsigned)nv;
                sl = in ? pp_ * F + (pf - shift_f) : 0;
                pp_ += r512; pf += q512;
                if (pp_ >= P) { pp_ -= P; ++pf; }
            } else {
                const uint32_t sv = i < L ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irs, i * 4, 0, 0) : 0xffffffffu;
                in = sv < (uint32_t)nv;
                sl = in ? (int)sv : 0;
            }
            return sl;
        };
        for (int e = 0; e < rounds; e += 4) {                // (rounds past the end: positions >= L have no slot)
            bool in0, in1, in2, in3;
            const int s0_ = slot_of(e, in0), s1_ = slot_of(e + 1, in1), s2_ = slot_of(e + 2, in2), s3_ = slot_of(e + 3, in3);
            const uint32_t b0_ = slotbit(s0_), b1_ = slotbit(s1_), b2_ = slotbit(s2_), b3_ = slotbit(s3_);
            const unsigned long long w0 = __ballot(in0 && b0_), w1 = __ballot(in1 && b1_), w2 = __ballot(in2 && b2_), w3 = __ballot(in3 && b3_);
            if (lane < 8) {
                const unsigned long long wsel = (lane >> 1) == 0 ? w0 : (lane >> 1) == 1 ? w1 : (lane >> 1) == 2 ? w2 : w3;
                const int word = (e + (lane >> 1)) * 16 + wv * 2 + (lane & 1);
                if (word < 1024) posmask[word] = (uint32_t)(wsel >> (32 * (lane & 1)));
            }
        }
    }
    __syncthreads();                                         // #5
    sub[2] = wall_clock64() - stamp0;                         // position words
    // ---- two exclusive scans in one: members before my slot word (high half) / before my two position words (low half)
    int members_total;
    {
        const int nw = (L + 31) >> 5;
        const uint32_t w0 = 2 * tid < nw ? posmask[2 * tid] : 0u, w1 = 2 * tid + 1 < nw ? posmask[2 * tid + 1] : 0u;
        const int packed = (__popc(mm) << 16) | (__popc(w0) + __popc(w1));
        const int wincl = wave_incl_scan_dpp(packed);
        if (lane == 63) scratch[16 + wv] = wincl;
        __syncthreads();                                     // #6
        int ex = wincl - packed, total = 0;
#pragma unroll
        for (int x = 0; x < kResWaves; ++x) { const int t = scratch[16 + x]; ex += x < wv ? t : 0; total += t; }
        slotpre[tid] = ex >> 16;
        pospre[2 * tid] = ex & 0xffff;
        pospre[2 * tid + 1] = (ex & 0xffff) + __popc(w0);
        members_total = total >> 16;
        if (members_total != (total & 0xffff)) members_total = -1;          // (the two views of the member set disagree: reported)
    }
    // (is a mail that holds the result in already?  one look, answered by the time the barrier below is passed: see C.)
    if (tid == 0) {
        const unsigned long long v = wants_mail ? __hip_atomic_load(&a.bar->mail_tag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        bcast[12] = (long long)(v >> 2) == (long long)a.seq ? (int)(v & 3ull) : 0;
    }
    __syncthreads();                                         // #7
    stamp[3] = wall_clock64() - stamp0;                       // member bits by slot and position, prefix sums
    // (the plan and the arithmetic of the published result must agree: an internal check - the result block has left, so a
    // disagreement is reported through the device's error word, by the next call)
    const bool plan_bad = members_total != members_e || uniform(hcnt[0]) != count;
    if (plan_bad && bid == 0 && tid == 0) atomicOr((unsigned long long*)(a.stats + FF_STAT_ERROR), (unsigned long long)FF_ERR_BIT_RESIDENT);
    // ---- the outputs: in the launch arguments, or by mail.  The host mailed them before it began to wait for the result block
    // (a guessed length that came true, or input-length buffers) - then the relay wave has them already - or it sizes them to the
    // l_out it has just read and mails them now: this is where the launch waits for that, rows in hand (the alternative is a
    // second kernel that reads every row again).  Bounded: a host that does not answer within ~4 ms gets the plan only
    // (FF_STAT_ACK = 4 seq + 3: ff_ctx_merge_apply follows with the merge kernel).
    char* out_ptr = a.out;
    long long out_cap = a.L_cap;
    int aux_n = a.aux.n;
    // the auxiliary tensors as the rest of the kernel reads them: LDS copies of the ff_aux_t entries (a mail slot's layout)
    static_assert(sizeof(ff_aux_t) == 40 && FF_MAIL_WORDS == 4 + 5 * FF_MAX_AUX && FF_MAIL_WORDS <= 32, "slot layout");
    const ff_aux_t* auxe = (const ff_aux_t*)(sres + 4);
    if (!a.mail && tid == 0) {
#pragma unroll
        for (int x = 0; x < FF_MAX_AUX; ++x) {
            sres[4 + 5 * x] = (long long)(uintptr_t)a.aux.a[x].src;
            sres[5 + 5 * x] = (long long)(uintptr_t)a.aux.a[x].dst;
            sres[6 + 5 * x] = a.aux.a[x].row_bytes;
            sres[7 + 5 * x] = a.aux.a[x].outer;
            sres[8 + 5 * x] = a.aux.a[x].src_outer_bytes;
        }
    }
    if (a.mail && !wants_mail) aux_n = 0;
    // ---- The outputs are not there yet (a threshold-branch call with exactly sized outputs: the host is allocating l_out rows
    // right now, ~20 us): the fold's ARITHMETIC does not need them.  Pass 1 walks the rows as the fold proper does and parks the
    // finished row of every run in the home of the run's LAST row (a static place: the row before the one that opens the next
    // run); the run that is open at the end of the segment - with the rows it continues with in the next segments - ends in
    // tail_x.  Pass 2, once the outputs are known, only stores.  (The wave that relays the mail keeps its hands free.)
    // (not in the instance whose rows are sums: its register file is full)
    const bool two_pass = !kAdd && wants_mail && bcast[12] == 0 && data_wave && !relay_wave && n > 0;
    uint4 tail_x = make_uint4(0, 0, 0, 0);
    int tail_lane = -1;                                      // lane (= row of my segment) of the open run's anchor; -1: none
    if constexpr (!kAdd) if (two_pass) {
        const bool mbit = lane < n && slotbit(s0 + lane);
        const unsigned long long memw = __ballot(mbit);
        float acc[E];
        int open_n = 0;
        auto finished = [&]() -> uint4 {                     // (as flush() below, without the store)
            if (open_n > 0) {
                float o[E];
                const float r = 1.0f / A::rnd((float)(open_n + 1));
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = acc[e] * r;
                return A::pack_rne(o);
            }
            return A::pack(acc);
        };
        auto home = [&](int r) -> uint4& { return r < RL ? *lrow(r) : v[r - RL]; };       // r: static
#pragma unroll
        for (int r = 0; r < RL + RV; ++r) {
            if (r < n) {
                if (!((memw >> r) & 1ull)) {
                    if (tail_lane >= 0 && open_n > 0) { if (r > 0) home(r > 0 ? r - 1 : 0) = finished(); }
                    tail_lane = r;
                    open_n = 0;
                    A::unpack(home(r), acc);
                } else if (tail_lane >= 0) {                 // (leading members belong to the previous workgroup's run)
                    float y[E];
                    A::unpack(home(r), y);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] = A::rnd(acc[e] + y[e]);
                    ++open_n;
                }
            }
        }
        if (tail_lane >= 0) {
            // the open run goes on in the following segments: those rows come from L2 / the Infinity Cache, a few at a time (the
            // VGPR rows are all alive here - they hold parked results - and the wait for the host hides the round trips)
            for (int t = s1; t < nv;) {
                const int tt = t + lane;
                const bool mb = tt < nv && slotbit(tt);
                const unsigned long long mw = __ballot(mb);
                const int run = mw == ~0ull ? kWave : __ffsll((long long)~mw) - 1;
                int iw = 0;
                if (lane < run) {
                    if constexpr (kHint) iw = pos_hint(tt);
                    else iw = a.order[tt];
                }
                constexpr int kB = kAdd ? 2 : 4;
                for (int u = 0; u < run; u += kB) {
                    uint4 x[kB], xa[kAdd ? kB : 1];
#pragma unroll
                    for (int z = 0; z < kB; ++z) {
                        if (u + z < run) {
                            const uint32_t so = (uint32_t)__builtin_amdgcn_readlane(iw, u + z) * rb;
                            x[z] = buf_load16s(hres, vcol, so);
                            if constexpr (kAdd) xa[z] = buf_load16s(make_rsrc(a.addend, (uint32_t)L * rb), vcol, so);
                        }
                    }
#pragma unroll
                    for (int z = 0; z < kB; ++z) {
                        if (u + z < run) {
                            float y[E];
                            if constexpr (kAdd) x[z] = add16<DT>(x[z], xa[z]);
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
            tail_x = finished();
        }
    }
    if (wants_mail) {
        if (relay_wave && relay_state != 1) {
            const long long t0 = wall_clock64();
            while (true) {
                const int got = relay_try(relay_state == 2);
                if (got == 1) break;
                if (got == 2) relay_state = 2;
                if (wall_clock64() - t0 > 400000) {             // 4 ms
                    if (lane == 0) {
                        __hip_atomic_store(&a.bar->mail_tag[0], (unsigned long long)(tag_mail + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&a.host_mapped[FF_STAT_ACK], (int64_t)(tag_mail + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (tid == 0) {
            int got = 3;
            const long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long v = __hip_atomic_load(&a.bar->mail_tag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((long long)(v >> 2) == (long long)a.seq) { got = (int)(v & 3ull); break; }
                __builtin_amdgcn_s_sleep(2);
                if ((spins & 63u) == 63u && wall_clock64() - t0 > 1000000) break;       // (10 ms: the relay wave's bound and more)
            }
            bcast[13] = got;
        }
        __syncthreads();
        const bool have = bcast[13] == 1 || bcast[13] == 2;
        if (tid < FF_MAIL_WORDS) sres[tid] = have ? (long long)__hip_atomic_load(&a.bar->mail[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ll;
        __syncthreads();
        if (!have) {
            out_ptr = nullptr;
            aux_n = 0;
        } else {
            out_ptr = (char*)(uintptr_t)sres[1];
            out_cap = sres[2];
            const int n_mail = (int)sres[3];
            aux_n = out_ptr ? (n_mail < FF_MAX_AUX ? (n_mail < 0 ? 0 : n_mail) : FF_MAX_AUX) : 0;
        }
    }
    if (tid == 0) {
        // the (tensor, outer slice) pairs of the auxiliary rows: x << 8 | slice, with how a 16-lane group moves one row of the
        // pair (<< 16; 2: 16 bytes per lane, 1: one 8-byte word, 0: copy_row)
        int np = 0, slow = 0;
        for (int x = 0; x < aux_n; ++x) {
            const ff_aux_t& ax = auxe[x];
            const uintptr_t al = (uintptr_t)ax.src | (uintptr_t)ax.dst | (uintptr_t)ax.row_bytes | (uintptr_t)ax.src_outer_bytes;
            const int f = (ax.row_bytes <= 256 && !(al & 15)) ? 2 : ((ax.row_bytes == 8 && !(al & 7)) ? 1 : 0);
            for (int kq = 0; kq < (int)ax.outer; ++kq, ++np) {
                if (np < 8) scratch[8 + np] = (f << 16) | (x << 8) | kq;
                if (np >= 8 || f == 0) slow = 1;
            }
        }
        scratch[29] = slow;                                  // some pair takes the general row copy
        scratch[30] = np;
        scratch[31] = 0;
    }
    __syncthreads();
    sub[7] = wall_clock64() - stamp0;                         // outputs known (mail read)
    const bool apply = !plan_bad && (l_out == L || (out_ptr != nullptr && out_cap >= (long long)l_out));
    auto members_before_pos = [&](int i) { return pospre[i >> 5] + __popc(posmask[i >> 5] & ((1u << (i & 31)) - 1u)); };
    auto members_before_slot = [&](int t) { return slotpre[t >> 5] + __popc(slotword(t >> 5) & ((1u << (t & 31)) - 1u)); };
    if (bid == 0 && tid == 0) {                               // (debug words: device block only)
        a.stats[FF_STAT_KTH_KEY] = kth;
        a.stats[FF_STAT_TIES_TAKEN] = topk ? need : 0;
    }

    const bool folded = l_out != L;
    const int b0 = (int)((long long)bid * L / G), b1 = (int)((long long)(bid + 1) * L / G);      // my slice of positions / slots
    // ---- member / keep / dst: the plan's arrays (the merge kernel that follows a plan-only launch, the attention-mask gather and
    // the diagnostics read them) and - when the launch folds - the by-patch order of the compacted sequence + its inverse (the
    // next merge call skips K0).  By `nthr` threads, `rt` of them this one.
    auto index_roles = [&](int rt, int nthr) {
        for (int i = b0 + rt; i < b1; i += nthr) {
            const uint32_t mbit = (posmask[i >> 5] >> (i & 31)) & 1u;
            a.keep[i] = (uint8_t)(mbit ^ 1u);
            a.dst[i] = mbit ? -1 : i - members_before_pos(i);
            a.member[i] = i < nv ? (uint8_t)slotbit(i) : (uint8_t)0;
        }
        if (!apply || !folded || !a.order_next) return;
        for (int t = b0 + rt; t < b1; t += nthr) {
            int rank, i;
            if (t < nv) {
                if (slotbit(t)) continue;
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
    };
    if (!apply) {                                           // plan only: ff_ctx_merge_apply follows
        index_roles(tid, kResThreads);
        return;
    }
    // ---- auxiliary rows (position tables, patch types) of the kept positions of my slice.  One position = one unit of work for a
    // group of 16 lanes, handed out by an LDS counter: the wave that holds no rows starts at once, the others join when their
    // fold is done.  Per position the loads of all (tensor, outer slice) pairs go out before the first store (the pairs' table:
    // scratch[8..15], built by thread 0 above; a 16-lane group moves up to 256 bytes per pair at once).
    // (one wave alone, descriptors looked up per task, took 35 us for the 53 positions of a 7B segment: 24 us behind the fold)
    auto aux_rows = [&](auto np_c, auto u_c) {
        constexpr int NP = decltype(np_c)::value;            // pairs held in registers (the table's first NP)
        constexpr int U = decltype(u_c)::value;              // positions per turn of a group: U x NP loads in flight
        const int np = scratch[30];
        if (!folded || np <= 0) return;
        const int l16 = lane & 15, gl = lane & ~15, npos = b1 - b0;
        // the pairs once: source of position 0, destination of output row 0, bytes per row, how to move one
        const char* spz[NP];
        char* dpz[NP];
        int rbz[NP], fz[NP];
#pragma unroll
        for (int z = 0; z < NP; ++z) {
            spz[z] = nullptr; dpz[z] = nullptr; rbz[z] = 0; fz[z] = 0;
            if (z < np) {
                const int e = scratch[8 + z], x = (e >> 8) & 0xff, kq = e & 0xff;
                const ff_aux_t& ax = auxe[x];
                fz[z] = e >> 16;
                rbz[z] = (int)ax.row_bytes;
                spz[z] = aux_src_row(ax, kq, 0, L);
                dpz[z] = (char*)ax.dst + (int64_t)kq * out_cap * ax.row_bytes;
            }
        }
        while (true) {
            int p = 0;
            if (l16 == 0) p = atomicAdd(&scratch[31], U);
            p = __shfl(p, gl);
            if (p >= npos) break;
            int iu[U];
            int64_t ru[U];
            uint4 u[U][NP];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int i = b0 + p + q;
                const bool kept = p + q < npos && !((posmask[i >> 5] >> (i & 31)) & 1u);
                iu[q] = kept ? i : -1;
                ru[q] = kept ? i - members_before_pos(i) : 0;
#pragma unroll
                for (int z = 0; z < NP; ++z) {
                    u[q][z] = make_uint4(0, 0, 0, 0);
                    if (z < np && kept) {
                        const char* sp = spz[z] + (int64_t)i * rbz[z];
                        if (fz[z] == 2) { if (l16 * 16 < rbz[z]) u[q][z] = *(const uint4*)(sp + l16 * 16); }
                        else if (fz[z] == 1) { if (l16 == 0) { const uint2 w = *(const uint2*)sp; u[q][z].x = w.x; u[q][z].y = w.y; } }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
#pragma unroll
                for (int z = 0; z < NP; ++z) {
                    if (z < np && iu[q] >= 0) {
                        char* dq = dpz[z] + ru[q] * rbz[z];
                        if (fz[z] == 2) { if (l16 * 16 < rbz[z]) *(uint4*)(dq + l16 * 16) = u[q][z]; }
                        else if (fz[z] == 1) { if (l16 == 0) *(uint2*)dq = make_uint2(u[q][z].x, u[q][z].y); }
                    }
                }
            }
            if (scratch[29] || np > NP) {
                // pairs the table does not hold, odd sizes, long rows: one by one (ONE copy of the general row copy in the
                // kernel's code: unrolled into the block above it made this path crawl through the instruction cache)
                for (int q = 0; q < U; ++q) {
                    if (iu[q] < 0) continue;
                    int z = 0;
                    for (int x = 0; x < aux_n; ++x) {
                        const ff_aux_t& ax = auxe[x];
                        for (int kq = 0; kq < (int)ax.outer; ++kq, ++z)
                            if (z >= NP || (scratch[8 + z] >> 16) == 0)
                                copy_row(aux_src_row(ax, kq, iu[q], L), (char*)ax.dst + ((int64_t)kq * out_cap + ru[q]) * ax.row_bytes, ax.row_bytes, l16, 16);
                    }
                }
            }
        }
    };
    // a wave that holds no rows (rows of fewer than 8 tiles) does the short roles while the others fold: it walks past the fold
    // below at once
    const bool spare_wave = nt < kResWaves;
    // Up to three pairs (patch types + two [1, L, .] position tables: every model but the M-RoPE ones) go BEFORE the fold, by
    // every wave, two positions per 16-lane group: the memory system is quiet here and one turn is one load latency (~2 us);
    // behind the fold's stores the same turn took 5 us, and the wave without rows, started early, 14 us for its share.
    const bool aux_first = scratch[30] <= 3;
    if (aux_first) aux_rows(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});

    // ======================================================================================================================
    // C. fold + compaction from the resident rows
#ifdef FF_RES_WGSTAMPS
    long long wg_waited = 0;
#endif
    if (!kAdd && folded && data_wave && two_pass) {
        // pass 2: every run's finished row from where pass 1 parked it to its output row
        const __amdgpu_buffer_rsrc_t ores = make_rsrc(out_ptr, (uint32_t)(out_cap * (long long)rb));
        const int js = s0 + lane;
        const bool mbit = lane < n && slotbit(js);
        const unsigned long long memw = __ballot(mbit);
        int dv = 0;
        if (lane < n && !mbit) {
            int i;
            if constexpr (kHint) i = pos_hint(js);
            else i = a.order[js];
            dv = i - members_before_pos(i);
        }
        int cur = -1;
#pragma unroll
        for (int r = 0; r < RL + RV; ++r) {
            if (r < n) {
                if (!((memw >> r) & 1ull)) cur = __builtin_amdgcn_readlane(dv, r);
                if (r + 1 < n && !((memw >> (r + 1)) & 1ull) && cur >= 0)
                    buf_store16s<2>(ores, vcol, (uint32_t)cur * rb, r < RL ? *lrow(r) : v[r - RL]);
            }
        }
        if (tail_lane >= 0) buf_store16s<2>(ores, vcol, (uint32_t)__builtin_amdgcn_readlane(dv, tail_lane) * rb, tail_x);
        // non-visual rows (kept as they are): row q of the order's tail goes to workgroup q mod G
        const int n_tail = L - nv;
        for (int qq = bid; qq < n_tail; qq += G) {
            int i;
            if constexpr (kHint) i = qq < pre ? qq : qq + nv;
            else i = a.order[nv + qq];
            uint4 x = buf_load16s<2>(hres, vcol, (uint32_t)i * rb);
            if constexpr (kAdd) x = add16<DT>(x, buf_load16s<2>(make_rsrc(a.addend, (uint32_t)L * rb), vcol, (uint32_t)i * rb));
            buf_store16s<2>(ores, vcol, (uint32_t)(i - members_before_pos(i)) * rb, x);
        }
    } else if (folded && data_wave) {
        const __amdgpu_buffer_rsrc_t ores = make_rsrc(out_ptr, (uint32_t)(out_cap * (long long)rb));
        // member bits of my slots and of the 64 behind them; output row of every anchor (by lane)
        const int js = s0 + lane;
        const bool mbit = lane < n && slotbit(js);
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
            // T(a / div) == T(a * RN(1 / div)) for every bf16-valued a and divisor T(k): ff_merge_body.h
            float o[E];
            const uint32_t off = (uint32_t)open_r * rb;
            if (open_n > 0) {
                const float r = 1.0f / A::rnd((float)(open_n + 1));
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = acc[e] * r;
                buf_store16s<2>(ores, vcol, off, A::pack_rne(o));
            } else {
                buf_store16s<2>(ores, vcol, off, A::pack(acc));
            }
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
        sub[4] = wall_clock64() - stamp0;                     // LDS rows folded
        // The last run of my segment may go on in the following segments.  Those rows sit in other workgroups' registers; this
        // one fetches them again (L2 / Infinity Cache: the only rows read twice).  The first kResRL of them are requested NOW, by
        // LDS-DMA into the LDS rows just folded, and arrive under the fold of the VGPR rows: fetched behind the fold, two per
        // round trip, they were the kernel's tail (up to 10 us at the 7B layout).
        int pre_run = 0;
        if constexpr (!kAdd) {                               // (sums: both halves through registers, in the loop further down)
            const int tt = s1 + lane;
            const bool mb = tt < nv && slotbit(tt);
            const unsigned long long mw = __ballot(mb);
            const int run = mw == ~0ull ? kWave : __ffsll((long long)~mw) - 1;      // leading members behind my segment
            pre_run = run < RL ? run : RL;
            int iw = 0;
            if (lane < pre_run) {
                if constexpr (kHint) iw = pos_hint(tt);
                else iw = a.order[tt];
            }
            for (int u = 0; u < pre_run; ++u)
                buf_load16_lds(hraw, vcol, (uint32_t)__builtin_amdgcn_readlane(iw, u) * rb, (uint32_t)(uintptr_t)(lrows + ((size_t)u * nt + wv) * 1024));
        }
#pragma unroll
        for (int i = 0; i < RV; ++i)
            if (RL + i < n) take(v[i], RL + i);
        sub[5] = wall_clock64() - stamp0;                     // VGPR rows folded
        if (pre_run > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the LDS-DMA rows: not the compiler's to count)
#ifdef FF_RES_WGSTAMPS
        wg_waited = wall_clock64() - stamp0;
#endif
        if (open_r >= 0) {
            for (int u = 0; u < pre_run; ++u) {
                float y[E];
                A::unpack(*lrow(u), y);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = A::rnd(acc[e] + y[e]);
                ++open_n;
            }
            // (a run of more than kResRL rows behind my segment: the rest eight at a time)
            for (int t = s1 + pre_run; (kAdd || pre_run == RL) && t < nv;) {
                const int tt = t + lane;
                const bool mb = tt < nv && slotbit(tt);
                const unsigned long long mw = __ballot(mb);
                const int run = mw == ~0ull ? kWave : __ffsll((long long)~mw) - 1;      // leading members of this window
                int iw = 0;
                if (lane < run) {
                    if constexpr (kHint) iw = pos_hint(tt);
                    else iw = a.order[tt];
                }
                constexpr int kB = 8;                        // rows in flight (the VGPR rows are folded: their registers are free)
                for (int u = 0; u < run; u += kB) {
                    uint4 x[kB], xa[kAdd ? kB : 1];
#pragma unroll
                    for (int z = 0; z < kB; ++z) {
                        if (u + z < run) {
                            const uint32_t so = (uint32_t)__builtin_amdgcn_readlane(iw, u + z) * rb;
                            x[z] = buf_load16s(hres, vcol, so);
                            if constexpr (kAdd) xa[z] = buf_load16s(make_rsrc(a.addend, (uint32_t)L * rb), vcol, so);
                        }
                    }
#pragma unroll
                    for (int z = 0; z < kB; ++z) {
                        if (u + z < run) {
                            float y[E];
                            if constexpr (kAdd) x[z] = add16<DT>(x[z], xa[z]);
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
        sub[6] = wall_clock64() - stamp0;                     // runs continued into the next segments
        // non-visual rows (kept as they are): row q of the order's tail goes to workgroup q mod G
        const int n_tail = L - nv;
        for (int qq = bid; qq < n_tail; qq += G) {
            int i;
            if constexpr (kHint) i = qq < pre ? qq : qq + nv;
            else i = a.order[nv + qq];
            uint4 x = buf_load16s<2>(hres, vcol, (uint32_t)i * rb);
            if constexpr (kAdd) x = add16<DT>(x, buf_load16s<2>(make_rsrc(a.addend, (uint32_t)L * rb), vcol, (uint32_t)i * rb));
            buf_store16s<2>(ores, vcol, (uint32_t)(i - members_before_pos(i)) * rb, x);
        }
    }
    stamp[4] = wall_clock64() - stamp0;                       // my rows folded and written (this wave)
#ifdef FF_RES_WGSTAMPS
    if (tid == 0) {                                          // (latest first wave of the launch, relative to its workgroup's start;
        // one atomic per workgroup and word: same-address device atomics cost ~10 ns EACH - three per wave made the kernel 114 us)
        atomicMax(&wgdbg[9], (unsigned long long)(wg_waited - sub[5]));     // longest wait for the prefetched continuation rows
        atomicMax(&wgdbg[10], (unsigned long long)(sub[6] - sub[5]));      // longest continuation into the next segments
        atomicMax(&wgdbg[11], (unsigned long long)stamp[4]);               // fold + non-visual rows done
    }
#endif
    if (!spare_wave) {
        index_roles(tid, kResThreads);
        if (!aux_first) aux_rows(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
    } else if (wv == kResWaves - 1) {
        // (first in line at its SIMD's issue port: the wave it shares the SIMD with is folding, VALU-bound)
        __builtin_amdgcn_s_setprio(3);
#ifdef FF_RES_WGSTAMPS
        if (bid == 0 && lane == 0) wgdbg[6] = (unsigned long long)wall_clock64();
#endif
        index_roles(lane, kWave);
#ifdef FF_RES_WGSTAMPS
        if (bid == 0 && lane == 0) wgdbg[7] = (unsigned long long)wall_clock64();
#endif
        // (more than three pairs - M-RoPE: the wave without rows starts on them at once, the others join behind their fold)
        if (!aux_first) aux_rows(std::integral_constant<int, 8>{}, std::integral_constant<int, 5>{});
#ifdef FF_RES_WGSTAMPS
        if (bid == 0 && lane == 0) wgdbg[8] = (unsigned long long)wall_clock64();
#endif
    } else {
        if (!aux_first) aux_rows(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});      // (a data wave behind its fold: what the spare wave has not taken yet)
    }
    stamp[5] = wall_clock64() - stamp0;                       // short roles done
#ifdef FF_RES_WGSTAMPS
    __syncthreads();
    if (tid == 0) {
        const unsigned long long te = (unsigned long long)wall_clock64();
        atomicMax(&wgdbg[2], ~0ull - te);
        atomicMax(&wgdbg[3], te);
        if (bid == 0) wgdbg[5] = te;
    }
#endif
    if (bid == 0 && tid == 0) {
        stamp[6] = wall_clock64() - stamp0;
        for (int x = 0; x < 7; ++x) a.stats[FF_STAT_T_PLAN + x] = stamp[x];
        for (int x = 0; x < 8; ++x) a.stats[FF_STAT_T_ORDER + x] = sub[x];
        if (folded && a.order_next) {
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
bool merge_resident_fits(int dtype, int64_t L, int64_t d, int64_t nv, bool addend, bool hinted, int fold) {
    // (bf16 only: the fp16 fold keeps its eight IEEE divisions per flush - ff_merge_body.h - and does not fit next to 160 pinned VGPRs)
    if (dtype != FF_BF16) return false;
    // (rows that are sums - the fused residual add of call B - come with a maintained order: call B follows call A)
    if ((addend && hinted) || fold != FF_FOLD_SEQUENTIAL) return false;
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

template <int DT, bool kHint, bool kAdd>
static int launch_res(const ResArgs& a, int cus, hipStream_t st) {
    static std::atomic<bool> attr_set[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
    if (dev < 0 || !attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)k_merge_resident<DT, kHint, kAdd>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)res_lds_bytes(8));
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_merge_resident<DT, kHint, kAdd>), dim3((unsigned)cus), dim3(kResThreads), res_lds_bytes(a.nt), st, a);
    return (int)hipGetLastError();
}

int launch_merge_resident(const ResLaunch& p, hipStream_t st) {
    const int cus = res_cus();
    if (cus < 8) return FF_ERR_UNSUPPORTED;
    ResArgs a;
    a.hidden = (const char*)p.hidden;
    a.addend = (const char*)p.addend;
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
    a.pp = merge_plan_params(p.dtype, p.thr, p.sub, p.ratio_lb, p.force_k);
    a.pp.n_slices = (int)((p.nv + kSelSlice - 1) / kSelSlice);
    a.member = p.member; a.keep = p.keep; a.dst = p.dst;
    a.order_next = p.order_next; a.inv_next = p.inv_next;
    a.stats = p.stats; a.host_mapped = p.host_mapped; a.seq = p.seq;
    a.aux.n = (p.hidden_out && !p.mail) ? p.n_aux : 0;
    for (int x = 0; x < FF_MAX_AUX; ++x) a.aux.a[x] = x < a.aux.n ? p.aux[x] : ff_aux_t{nullptr, nullptr, 0, 0, 0};
    a.bar = ws_resbar(p.ws);
    a.mail = p.mail;
    const bool hint = p.hint_frames > 0;
    if (hint) return p.addend ? FF_ERR_UNSUPPORTED : launch_res<FF_BF16, true, false>(a, cus, st);
    return p.addend ? launch_res<FF_BF16, false, true>(a, cus, st) : launch_res<FF_BF16, false, false>(a, cus, st);
}

}  // namespace ff
