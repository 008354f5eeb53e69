// Probe for the "read the activations once" merge call (round 6): is the BARE pattern
//   one workgroup per CU loads its ~53 rows of the by-patch order into VGPRs (+ LDS) with everything in flight at once,
//   computes the row norms / pair dots, publishes its similarities, crosses ONE XCD-hierarchical grid barrier, reads all
//   similarities back, then folds ~70 % of its rows into the others FROM REGISTERS and writes the ~30 % that survive
// fast enough to beat K1 + plan + K4 (57.7 us at the LLaVA-Video-7B layout, 64 x 210 x 3584 bf16, profiles/r05_timeline_7b.txt)?
// Kill criterion of the round-5 review: the bare pattern must be <= 0.7 x 57.7 = 40 us at 96 MB.
// Development aid (not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -o resprobe resprobe.hip && ./resprobe [frames patches d]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int kAux = 0>
__device__ inline uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset = 0) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, (int)soffset, kAux);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int kAux = 0>
__device__ inline void buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset, const uint4& x) {
    u32x4 v; v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voffset, (int)soffset, kAux);
}
__device__ inline float rnd(float x) { return (float)(__bf16)x; }
__device__ inline void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ inline uint4 pack_rne(const float* f) {
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { bf16x2_t pr; pr.x = (__bf16)f[2 * e]; pr.y = (__bf16)f[2 * e + 1]; w[e] = __builtin_bit_cast(uint32_t, pr); }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ inline float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}
__device__ inline float sumsq(const uint4& v, float acc) {
    acc = dot2(v.x, v.x, acc); acc = dot2(v.y, v.y, acc); acc = dot2(v.z, v.z, acc); return dot2(v.w, v.w, acc);
}
__device__ inline float dot_rounded(const uint4& a, const uint4& b, float acc) {
    float x[8], y[8];
    unpack(a, x); unpack(b, y);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        bf16x2_t pr; pr.x = (__bf16)(x[e] * y[e]); pr.y = (__bf16)(x[e + 1] * y[e + 1]);
        acc = __builtin_amdgcn_fdot2_f32_bf16(pr, __builtin_bit_cast(bf16x2_t, 0x3f803f80u), acc, false);
    }
    return acc;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the wave on the DPP network (VALU latency, no LDS crossbar): the total ends up in lane 63
template <int kCtrl, int kRowMask = 0xf>
__device__ inline float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, 0xf, false));
}
__device__ inline float wave_sum_dpp63(float v) {
    v += dpp_f<0x111>(v);          // row_shr:1
    v += dpp_f<0x112>(v);          // row_shr:2
    v += dpp_f<0x114>(v);          // row_shr:4
    v += dpp_f<0x118>(v);          // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_f<0x142, 0xa>(v);     // row_bcast:15
    v += dpp_f<0x143, 0xc>(v);     // row_bcast:31
    return v;
}

// ---- XCD-hierarchical grid barrier (MI355X_MICROARCH.md, price list row barrier-xcd).  Groups are STATIC (blockIdx & 7: the XCD a
// block is observed to run on - for speed only, nothing depends on it), every spin is bounded.  State: zeroed before every launch.
struct GridBar {
    unsigned cnt[8][32];     // one 128-byte line per group
    unsigned top[32];
    unsigned gen[8][32];
    unsigned fail[32];
};
__device__ inline bool grid_barrier(GridBar* gb, int bid, int nwg, unsigned epoch) {
    // caller: every wave has drained its (write-through) stores and the workgroup has passed __syncthreads(); called by thread 0
    const int g = bid & 7;
    const unsigned gsize = (unsigned)((nwg + 7 - g) >> 3);
    const unsigned ngroups = (unsigned)(nwg < 8 ? nwg : 8);
    const unsigned old = __hip_atomic_fetch_add(&gb->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gsize * epoch) {
        const unsigned t = __hip_atomic_fetch_add(&gb->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == ngroups * epoch) {
            for (unsigned x = 0; x < ngroups; ++x) __hip_atomic_store(&gb->gen[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    bool ok = true;
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(&gb->gen[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch) break;
        __builtin_amdgcn_s_sleep(2);
        if (spins > (1u << 20)) { ok = false; __hip_atomic_store(&gb->fail[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// LDS-DMA: 16 bytes per lane, global -> LDS, lane-linear destination (wave-uniform base in M0)
__device__ inline void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct Args {
    const char* hidden;
    char* out;
    uint16_t* sim;          // [nv]
    GridBar* bar;
    int* sink;
    int nv, frames, patches, row_bytes;
    int mode;               // bit 0: skip the barrier; bit 1: skip the write phase
    int epoch;
    long long* stamps;      // [nwg][8 waves][8]
};

// sum of four floats over the wave at once (the total of each in lane 63): four independent DPP chains interleaved, so that
// no step waits for the two wait states a DPP read of a just-written VGPR needs
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

// NT: 1 KiB column tiles per row (= data waves); RL rows per workgroup in LDS (the FIRST slots), RV in VGPRs (the rest).
// 512 threads = 8 waves, 1 workgroup per CU.
template <int NT, int RV, int RL, bool kArith>
__global__ __launch_bounds__(512) void k_res_probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = RV + RL;
    static_assert(RV % 2 == 0 && RL % 2 == 0 || true, "");
    float* part = (float*)smem;                                  // [R + 1][8][2]  per-wave partial (|x|^2, dot with the previous row)
    unsigned char* lrows = smem + ((R + 1) * 16 * 4 + 15) / 16 * 16;      // [RL][NT][1024]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int s0 = (int)((long long)bid * a.nv / nwg), s1 = (int)((long long)(bid + 1) * a.nv / nwg);
    const int n = s1 - s0;                                      // <= R
    const int F = a.frames, P = a.patches;
    auto pos_of = [&](int s) { const int p = s / F, f = s - p * F; return f * P + p; };      // slot -> sequence position (frame-major)
    const uint32_t col = (uint32_t)wv * 1024u + (uint32_t)lane * 16u;
    const bool data_wave = wv < NT;
    long long st[10];
    st[8] = st[9] = 0;
    st[0] = wall_clock64();
    auto dump = [&]() { if (a.stamps && lane == 0) for (int x = 0; x < 10; ++x) a.stamps[((size_t)bid * 8 + wv) * 10 + x] = st[x]; };
    uint4 v[RV > 0 ? RV : 1];
    const uint32_t rb = (uint32_t)a.row_bytes;
    const __amdgpu_buffer_rsrc_t hres = make_rsrc(a.hidden, (uint32_t)a.nv * rb), ores = make_rsrc(a.out, (uint32_t)a.nv * rb);
    auto lrow = [&](int i) { return lrows + ((size_t)i * NT + wv) * 1024 + lane * 16; };
    if (data_wave) {
        // ---- loads and arithmetic software-pipelined, W rows ahead.  (All loads first and the arithmetic behind them - the first
        // version - let the arithmetic start 9-12 us into the kernel: a CU holds far fewer requests than 8 waves x 55 KiB, the
        // load instructions themselves queue, and a wave only gets past its last load when most of its data is already there.)
        constexpr int W = RL > 0 ? RL : 14;
        int p = s0 / F, f = s0 - p * F, issued = 0;
        auto next_row = [&]() { const uint32_t o = (uint32_t)(f * P + p) * rb; if (issued + 1 < n && ++f == F) { f = 0; ++p; } ++issued; return o; };
        const uint4 prev = buf_load16(hres, col, (uint32_t)pos_of(s0 > 0 ? s0 - 1 : 0) * rb);
#pragma unroll
        for (int i = 0; i < RL; ++i) glds16(a.hidden + next_row() + col, (uint32_t)(uintptr_t)(lrows + ((size_t)i * NT + wv) * 1024));
        if constexpr (RL == 0) {
#pragma unroll
            for (int i = 0; i < W && i < RV; ++i) v[i] = buf_load16(hres, col, next_row());
        }
        float lastf[8];
        float q0 = 0.f;
        if constexpr (kArith) { unpack(prev, lastf); q0 = sumsq(prev, 0.f); }
        st[1] = wall_clock64();
        auto one = [&](const uint4& x, float& q, float& d) {
            float y[8];
            unpack(x, y);
            q = sumsq(x, 0.f);
            d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                bf16x2_t pr; pr.x = (__bf16)(lastf[e] * y[e]); pr.y = (__bf16)(lastf[e + 1] * y[e + 1]);
                d = __builtin_amdgcn_fdot2_f32_bf16(pr, __builtin_bit_cast(bf16x2_t, 0x3f803f80u), d, false);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) lastf[e] = y[e];
        };
        auto two = [&](const uint4& x0, const uint4& x1, int i) {
            if constexpr (kArith) {
                float qa, da, qb, db;
                one(x0, qa, da);
                one(x1, qb, db);
                wave_sum4_dpp63(qa, da, qb, db);
                if (lane == 63) *(float4*)&part[((i + 1) * 8 + wv) * 2] = make_float4(qa, da, 0.f, 0.f), *(float4*)&part[((i + 2) * 8 + wv) * 2] = make_float4(qb, db, 0.f, 0.f);
            }
        };
        if constexpr (kArith) {
            float z0 = 0.f, z1 = 0.f, z2 = 0.f;
            wave_sum4_dpp63(q0, z0, z1, z2);
            if (lane == 63) part[(0 * 8 + wv) * 2] = q0;
        }
        static_assert(RL % 2 == 0 && RV % 2 == 0, "rows are handled in pairs");
        // LDS rows i, i+1: first request the VGPR rows that take their place in the window, then wait until at most RL loads are
        // outstanding (the LDS-DMA rows behind these two + the VGPR rows requested so far): hand-counted, the compiler does not see LDS-DMA
#pragma unroll
        for (int i = 0; i < RL; i += 2) {
            if (i < RV) v[i] = buf_load16(hres, col, next_row());
            if (i + 1 < RV) v[i + 1] = buf_load16(hres, col, next_row());
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RL) : "memory");
            two(*(const uint4*)lrow(i), *(const uint4*)lrow(i + 1), i);
        }
        st[8] = wall_clock64();
#pragma unroll
        for (int i = 0; i < RV; i += 2) {
            if (i + W < RV) v[i + W] = buf_load16(hres, col, next_row());
            if (i + W + 1 < RV) v[i + W + 1] = buf_load16(hres, col, next_row());
            two(v[i], v[i + 1], RL + i);
            if (i == (RV / 4) * 2) st[9] = wall_clock64();
        }
        st[2] = wall_clock64();
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): every row is on chip (and the compiler knows it)
    }
    __syncthreads();
    // ---- similarities of my slots (threads 0..n-1), published write-through
    if constexpr (kArith) {
        float na = 0.f, dd = 0.f;
        if (tid <= n) {
            float q = 0.f, d = 0.f;
            for (int w = 0; w < NT; ++w) { q += part[(tid * 8 + w) * 2]; d += part[(tid * 8 + w) * 2 + 1]; }
            na = rnd(sqrtf(q)); dd = rnd(d);
        }
        __syncthreads();
        if (tid <= n) { part[tid * 16] = na; part[tid * 16 + 1] = dd; }
        __syncthreads();
        if (tid < n) {
            const float nb = part[(tid + 1) * 16], d = part[(tid + 1) * 16 + 1];
            const float s = rnd(d / rnd(part[tid * 16] * nb));
            __hip_atomic_store(&a.sim[s0 + tid], (uint16_t)(__float_as_uint(s) >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    st[3] = wall_clock64();
    if (!(a.mode & 1)) {
        if (tid == 0) grid_barrier(a.bar, bid, nwg, (unsigned)a.epoch);
        __syncthreads();
    }
    st[4] = wall_clock64();
    // ---- the plan's stand-in: every workgroup reads all similarities (8 per lane per load) and derives something from them
    __shared__ int dec[8];
    {
        int cnt = 0;
        for (int t = tid * 8; t < a.nv; t += 512 * 8) {
            const uint4 w = *(const uint4*)(a.sim + t);
            cnt += __popc(w.x & 0x80008000u) + __popc(w.y & 0x80008000u) + __popc(w.z & 0x80008000u) + __popc(w.w & 0x80008000u);
        }
        cnt = (int)wave_sum((float)cnt);
        if (lane == 0) dec[wv] = cnt;
        __syncthreads();
    }
    int negs = 0;
    for (int w = 0; w < 8; ++w) negs += dec[w];
    st[5] = wall_clock64();
    if (a.mode & 2) { if (negs == 0x7fffffff) a.sink[0] = 1; st[6] = st[7] = st[5]; dump(); return; }
    // ---- fold + write from the resident rows: ~30 % of the slots open an output row, the others fold into it
    if (!data_wave) { st[6] = st[7] = st[5]; dump(); return; }
    const uint32_t salt = (uint32_t)negs & 1u;          // (so that the write phase depends on the plan)
    auto is_member = [&](int s) { return (((uint32_t)s * 2654435761u + salt) >> 16) % 10u >= 3u && (s % F) != 0; };
    float acc[8];
    int open_r = -1, open_n = 0;
    int out_base = (int)((long long)s0 * 3 / 10) + bid;        // (a stand-in for dst[]: about the right place)
    auto flush = [&]() {
        float o[8];
        const float r = open_n ? 1.0f / rnd((float)(open_n + 1)) : 1.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = acc[e] * r;
        buf_store16<2>(ores, col, (uint32_t)open_r * rb, pack_rne(o));
    };
    const unsigned long long memmask = __ballot(lane < n && is_member(s0 + lane) && lane > 0);      // (bit i: row i folds into the open row)
    auto take = [&](const uint4& x, int i) {
        if (!((memmask >> i) & 1ull)) {
            if (open_r >= 0) flush();
            open_r = out_base++;
            open_n = 0;
            unpack(x, acc);
        } else {
            float y[8];
            unpack(x, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = rnd(acc[e] + y[e]);
            ++open_n;
        }
    };
    for (int i = 0; i < RL && i < n; ++i) take(*(const uint4*)lrow(i), i);
#pragma unroll
    for (int i = 0; i < RV; ++i)
        if (RL + i < n) take(v[i], RL + i);
    if (open_r >= 0) flush();
    st[6] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[7] = wall_clock64();
    dump();
}

// what the three launches read and write, as bare streams (for the same-run comparison)
__global__ void k_touch(const uint4* __restrict__ p, size_t n16, int* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;
}

template <typename Fn>
static float time_us(Fn&& launch, int reps, int warm = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 64, P = argc > 2 ? atoi(argv[2]) : 210, d = argc > 3 ? atoi(argv[3]) : 3584;
    const int nv = F * P, row_bytes = d * 2;
    const size_t bytes = (size_t)nv * row_bytes;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; %d frames x %d patches x %d bf16 = %.1f MB, %d slots, %.1f per workgroup\n", prop.gcnArchName, cus, F, P, d, bytes / 1e6, nv,
           (double)nv / cus);
    char *h0, *h1, *out; uint16_t* sim; GridBar* bar; int* sink;
    CK(hipMalloc(&h0, bytes)); CK(hipMalloc(&h1, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sim, nv * 2 + 64));
    CK(hipMalloc(&bar, sizeof(GridBar))); CK(hipMalloc(&sink, 64));
    {
        std::vector<uint16_t> host(bytes / 2);
        uint32_t x = 12345;
        for (size_t i = 0; i < host.size(); ++i) { x = x * 1664525u + 1013904223u; host[i] = (uint16_t)(0x3f00 | ((x >> 20) & 0xff) | ((x >> 3) & 0x8000)); }
        CK(hipMemcpy(h0, host.data(), bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(h1, host.data(), bytes, hipMemcpyHostToDevice));
    }
    int flip = 0;
    const int per = (nv + cus - 1) / cus;
    int epoch = 0;
    long long* stamps; CK(hipMalloc(&stamps, (size_t)cus * 80 * 8));
    CK(hipMemset(bar, 0, sizeof(GridBar)));
#define RUN(NT, RV, RL, ARITH, MODE, ALT, WHAT) { \
        if (NT * 1024 >= row_bytes && (NT - 1) * 1024 < row_bytes && RV + RL >= per) { \
            const size_t lds = (((RV + RL + 1) * 16 * 4 + 15) / 16 * 16) + (size_t)RL * NT * 1024; \
            CK(hipFuncSetAttribute((const void*)k_res_probe<NT, RV, RL, ARITH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            float us = time_us([&] { if (ALT) flip ^= 1; Args a{flip ? h1 : h0, out, sim, bar, sink, nv, F, P, row_bytes, MODE, ++epoch, nullptr}; \
                                     hipLaunchKernelGGL((k_res_probe<NT, RV, RL, ARITH>), dim3(cus), dim3(512), lds, 0, a); }, 20); \
            unsigned fail = 0; CK(hipMemcpy(&fail, &bar->fail[0], 4, hipMemcpyDeviceToHost)); \
            printf("resident NT=%d RV=%2d RL=%2d lds=%6zu arith=%d mode=%d alt=%d : %7.1f us%s   %s\n", NT, RV, RL, lds, ARITH, MODE, ALT, us, fail ? "  BARRIER TIMEOUT" : "", WHAT); \
            if (MODE & 1) { CK(hipMemset(bar, 0, sizeof(GridBar))); epoch = 0; } \
        } }
    for (int rep = 0; rep < 2; ++rep) {
        float us = time_us([&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, (const uint4*)h0, bytes / 16, sink); }, 20);
        printf("flat read of the same buffer (warm in the Infinity Cache): %7.1f us\n", us);
        us = time_us([&] { flip ^= 1; hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, (const uint4*)(flip ? h1 : h0), bytes / 16, sink); }, 20);
        printf("flat read, two buffers alternated: %7.1f us\n", us);
        // 7 tiles (d = 3584)
        RUN(7, 54, 0, true, 0, 0, "all in VGPRs") RUN(7, 40, 14, true, 0, 0, "40 VGPR + 14 LDS") RUN(7, 32, 22, true, 0, 0, "32 + 22") RUN(7, 48, 6, true, 0, 0, "48 + 6")
        RUN(7, 40, 14, true, 0, 1, "alternating buffers") RUN(7, 40, 14, true, 1, 0, "no barrier") RUN(7, 40, 14, true, 2, 0, "no write phase") RUN(7, 40, 14, true, 3, 0, "no barrier, no write")
        RUN(7, 40, 14, false, 0, 0, "no arithmetic") RUN(7, 40, 14, false, 3, 0, "loads only") RUN(7, 40, 10, true, 0, 0, "C3: 49 rows") RUN(7, 36, 14, true, 0, 0, "49 rows")
        // 8 tiles (d = 4096)
        RUN(8, 40, 14, true, 0, 0, "40 + 14") RUN(8, 36, 0, true, 0, 0, "36 VGPR") RUN(8, 40, 14, false, 3, 0, "loads only")
    }
#define STAMPED(NT, RV, RL) { \
        if (NT * 1024 >= row_bytes && (NT - 1) * 1024 < row_bytes && RV + RL >= per) { \
            const size_t lds = (((RV + RL + 1) * 16 * 4 + 15) / 16 * 16) + (size_t)RL * NT * 1024; \
            CK(hipFuncSetAttribute((const void*)k_res_probe<NT, RV, RL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            for (int rep = 0; rep < 3; ++rep) { \
                Args a{h0, out, sim, bar, sink, nv, F, P, row_bytes, 0, ++epoch, stamps}; \
                hipLaunchKernelGGL((k_res_probe<NT, RV, RL, true>), dim3(cus), dim3(512), lds, 0, a); \
            } \
            CK(hipDeviceSynchronize()); \
            std::vector<long long> h((size_t)cus * 80); \
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost)); \
            long long t0 = 1ll << 62; \
            for (int b = 0; b < cus; ++b) for (int w = 0; w < NT; ++w) t0 = std::min(t0, h[((size_t)b * 8 + w) * 10]); \
            const char* names[10] = {"start", "first row there", "sims done", "published", "barrier passed", "plan done", "fold issued", "stores drained", "LDS rows done", "1/2 VGPR rows done"}; \
            const int order[10] = {0, 1, 8, 9, 2, 3, 4, 5, 6, 7}; \
            printf("stamps NT=%d RV=%d RL=%d (us after the first wave's start; data waves of all workgroups)\n", NT, RV, RL); \
            for (int xx = 0; xx < 10; ++xx) { \
                const int x = order[xx]; \
                std::vector<double> v; \
                for (int b = 0; b < cus; ++b) for (int w = 0; w < NT; ++w) v.push_back((h[((size_t)b * 8 + w) * 10 + x] - t0) / 100.0); \
                std::sort(v.begin(), v.end()); \
                printf("  %-18s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f", names[x], v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back()); \
                if (x == 1 || x == 2) { printf("   | by wave (median):"); for (int w = 0; w < NT; ++w) { std::vector<double> u; for (int b = 0; b < cus; ++b) u.push_back((h[((size_t)b * 8 + w) * 10 + x] - t0) / 100.0); std::sort(u.begin(), u.end()); printf(" %5.1f", u[u.size() / 2]); } } \
                printf("\n"); \
            } \
        } }
    STAMPED(7, 40, 14) STAMPED(7, 54, 0) STAMPED(8, 40, 14)
    return 0;
}
