// Host-side interface of the one-launch merge kernel (ff_resident.hip), used by the call context in ff_abi.hip.
#pragma once

#include "ff_common.h"

namespace ff {

struct ResLaunch {
    const void* hidden;
    const void* addend;           // optional: the rows are T(hidden + addend); only with a maintained order (no layout hint)
    void* hidden_out;             // NULL: plan only
    int dtype;
    int64_t L, d, L_cap;
    int64_t nv, ftn;              // visual / non-text tokens as the host knows them
    const int64_t* ptype;
    int32_t* order;
    int32_t* inv;
    int64_t hint_pre, hint_patches, hint_frames;      // hint_frames > 0: frame-major closed form (order / inv are written)
    void* sim;
    uint8_t* member;
    uint8_t* keep;
    int32_t* dst;
    int32_t* order_next;
    int32_t* inv_next;
    int64_t* stats;
    int64_t* host_mapped;
    int64_t seq;
    const ff_aux_t* aux;
    int n_aux;
    double thr, sub, ratio_lb;
    long long force_k;
    void* ws;
    size_t ws_bytes;
    const int64_t* mail;          // != NULL: outputs by mail (ff_ctx_merge_mail)
};

// Does a merge call of this shape run as the one-launch kernel?  `nv`: visual tokens as the host knows them (<= 0: unknown).
bool merge_resident_fits(int dtype, int64_t L, int64_t d, int64_t nv, bool addend, bool hinted, int fold);
int launch_merge_resident(const ResLaunch& p, hipStream_t st);

}  // namespace ff
