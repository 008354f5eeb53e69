// ff_layout.hip - token layout on the device: where the visual span sits in
// the prompt and the patch_type row FrameFusion.prepare receives.  The reference's packers do this
// with torch.where + Python list arithmetic + an upload of L int64 (see the header for file:line);
// these are a few tens of KB of integer work, so each entry point is ONE small launch.
#include <hip/hip_runtime.h>

#include "ff_common.h"

namespace ff {

constexpr int kSpanThreads = 1024;
constexpr int kSegsPerLaunch = 128;

struct SegArgs {
    int n;
    int background;                 // 1: positions outside every segment are written as TEXT (-1)
    ff_segment_t s[kSegsPerLaunch];
};

__global__ __launch_bounds__(kSpanThreads) void k_token_span(const int64_t* __restrict__ ids, int n, int64_t token,
                                                             int64_t* __restrict__ span) {
    __shared__ int s_first, s_last, s_count;
    if (threadIdx.x == 0) {
        s_first = 0x7fffffff;
        s_last = -1;
        s_count = 0;
    }
    __syncthreads();
    int first = 0x7fffffff, last = -1, count = 0;
    for (int i = threadIdx.x; i < n; i += kSpanThreads) {
        if (ids[i] == token) {
            first = min(first, i);
            last = i;
            ++count;
        }
    }
    if (count) {
        atomicMin(&s_first, first);
        atomicMax(&s_last, last);
        atomicAdd(&s_count, count);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        span[0] = s_count ? s_first : -1;
        span[1] = s_last;
        span[2] = s_count;
    }
}

__global__ __launch_bounds__(256) void k_fill_patch_type(int64_t* __restrict__ patch_type, int L, const SegArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    int64_t v = -1;
    bool hit = false;
    for (int s = 0; s < a.n; ++s) {                  // uniform loop over launch arguments (scalar loads)
        const unsigned off = (unsigned)(i - a.s[s].begin);
        if (off < (unsigned)a.s[s].count) {
            v = (int64_t)(((unsigned)a.s[s].first + off) % (unsigned)a.s[s].period);
            hit = true;
        }
    }
    if (hit || a.background) patch_type[i] = v;
}

// One workgroup; thread t owns the contiguous chunk [t*chunk, (t+1)*chunk).  The offset of a match
// inside its run is its distance from the last non-match before it: a max-scan of "last zero
// position" over the chunks gives every thread its starting state.
__global__ __launch_bounds__(kSpanThreads) void k_patch_type_from_mask(const uint8_t* __restrict__ mask, int n,
                                                                       int patch_num, int64_t* __restrict__ patch_type,
                                                                       int64_t* __restrict__ span) {
    __shared__ int s_wave_max[kSpanThreads / kWave];
    __shared__ int s_first, s_last, s_count, s_runs, s_bad;
    if (threadIdx.x == 0) {
        s_first = 0x7fffffff;
        s_last = -1;
        s_count = s_runs = s_bad = 0;
    }
    const int chunk = (n + kSpanThreads - 1) / kSpanThreads;
    const int lo = min((int)threadIdx.x * chunk, n), hi = min(lo + chunk, n);
    int last_zero = -1;
    for (int i = lo; i < hi; ++i)
        if (!mask[i]) last_zero = i;
    // inclusive max-scan over the workgroup
    int incl = last_zero;
    for (int d = 1; d < kWave; d <<= 1) {
        const int up = __shfl_up(incl, d, kWave);
        if (lane_id() >= d) incl = max(incl, up);
    }
    if (lane_id() == kWave - 1) s_wave_max[wave_id()] = incl;
    __syncthreads();
    int before = -1;                                  // last zero position in front of this chunk
    for (int w = 0; w < wave_id(); ++w) before = max(before, s_wave_max[w]);
    {
        const int up = __shfl_up(incl, 1, kWave);
        if (lane_id() > 0) before = max(before, up);
    }
    int first = 0x7fffffff, last = -1, count = 0, runs = 0, bad = 0;
    int lz = before;
    for (int i = lo; i < hi; ++i) {
        int64_t v = -1;
        if (mask[i]) {
            const int off = i - lz - 1;
            v = off;
            first = min(first, i);
            last = i;
            ++count;
            runs += off == 0;
            if ((i == n - 1 || !mask[i + 1]) && off + 1 != patch_num) ++bad;
        } else {
            lz = i;
        }
        patch_type[i] = v;
    }
    if (count) {
        atomicMin(&s_first, first);
        atomicMax(&s_last, last);
        atomicAdd(&s_count, count);
        atomicAdd(&s_runs, runs);
        if (bad) atomicAdd(&s_bad, bad);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        span[0] = s_count ? s_first : -1;
        span[1] = s_last;
        span[2] = s_count;
        span[3] = s_runs;
        span[4] = s_bad;
    }
}

}  // namespace ff

extern "C" int ff_token_span(const int64_t* ids, int64_t n, int64_t token, int64_t* span, ff_stream_t stream) {
    if (!span || n < 0 || (n > 0 && !ids)) return FF_ERR_ARG;
    if (n >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ff::k_token_span, dim3(1), dim3(ff::kSpanThreads), 0, (hipStream_t)stream, ids, (int)n, token, span);
    return (int)hipGetLastError();
}

extern "C" int ff_fill_patch_type(int64_t* patch_type, int64_t L, const ff_segment_t* segments_host, int64_t n_segments,
                                  ff_stream_t stream) {
    if (L < 0 || n_segments < 0 || (L > 0 && !patch_type) || (n_segments > 0 && !segments_host)) return FF_ERR_ARG;
    if (L >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    for (int64_t s = 0; s < n_segments; ++s) {
        const ff_segment_t& g = segments_host[s];
        if (g.begin < 0 || g.count < 0 || g.first < 0 || g.period < 1 || (int64_t)g.begin + g.count > L) return FF_ERR_ARG;
    }
    if (L == 0) return FF_OK;
    const unsigned blocks = (unsigned)((L + 255) / 256);
    int64_t done = 0;
    do {                                              // first launch also writes the TEXT background
        ff::SegArgs a;
        a.n = (int)((n_segments - done) < ff::kSegsPerLaunch ? (n_segments - done) : ff::kSegsPerLaunch);
        a.background = done == 0;
        for (int s = 0; s < a.n; ++s) a.s[s] = segments_host[done + s];
        hipLaunchKernelGGL(ff::k_fill_patch_type, dim3(blocks), dim3(256), 0, (hipStream_t)stream, patch_type, (int)L, a);
        done += a.n;
    } while (done < n_segments);
    return (int)hipGetLastError();
}

extern "C" int ff_patch_type_from_mask(const uint8_t* mask, int64_t n, int64_t patch_num, int64_t* patch_type,
                                       int64_t* span, ff_stream_t stream) {
    if (!span || n < 0 || patch_num < 1 || (n > 0 && (!mask || !patch_type))) return FF_ERR_ARG;
    if (n >= (1ll << 31) || patch_num >= (1ll << 31)) return FF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ff::k_patch_type_from_mask, dim3(1), dim3(ff::kSpanThreads), 0, (hipStream_t)stream, mask, (int)n,
                       (int)patch_num, patch_type, span);
    return (int)hipGetLastError();
}
