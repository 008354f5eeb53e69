// C-ABI glue: version, error strings, workspace sizing and the fused merge step that enqueues
// K0 -> K1 -> K2+K3 -> K4 from a single host call (one FrameFusion.forward merge call,
// framefusion/main.py:104-138).
#include "ff_common.h"

namespace ff {
int launch_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L, double thr, double sub,
                      double ratio_lb, int32_t* run_len, int32_t* dst, uint8_t* keep, int64_t* stats,
                      void* ws, int64_t* host_mapped, int64_t seq, hipStream_t st);
int launch_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                         const int32_t* order, const int32_t* run_len, const int32_t* dst,
                         const ff_aux_t* aux_host, int n_aux, hipStream_t st);
}  // namespace ff

extern "C" int ff_abi_version(void) { return FF_ABI_VERSION; }

extern "C" const char* ff_error_string(int code) {
    switch (code) {
        case FF_OK: return "ok";
        case FF_ERR_ARG: return "bad argument (null pointer, negative size or unknown dtype)";
        case FF_ERR_ALIGN: return "pointer or row size not 16-byte aligned";
        case FF_ERR_UNSUPPORTED: return "size outside the supported range";
        case FF_ERR_WORKSPACE: return "workspace too small (see ff_workspace_bytes)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" size_t ff_workspace_bytes(int64_t L, int64_t patch_num) {
    (void)patch_num;
    if (L < 0) return 0;
    return (size_t)((L + 255) / 256 * 256 + 256);
}

extern "C" int ff_merge_step(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                             const int64_t* patch_type, int64_t patch_num, int order_valid, double threshold,
                             double sub, double ratio_lb, int32_t* order, void* sim, int32_t* run_len,
                             int32_t* dst, uint8_t* keep, int64_t* stats, int64_t* stats_host_mapped,
                             int64_t seq, const ff_aux_t* aux_host, int n_aux, void* ws, size_t ws_bytes,
                             ff_stream_t stream) {
    if (!hidden || !hidden_out || !patch_type || !order || !sim || !run_len || !dst || !keep || !stats || !ws)
        return FF_ERR_ARG;
    if (ws_bytes < ff_workspace_bytes(L, patch_num)) return FF_ERR_WORKSPACE;
    if (L_cap < L) return FF_ERR_ARG;
    int rc;
    if (!order_valid) {
        rc = ff_build_order(patch_type, L, patch_num, order, stats, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    rc = ff_pair_similarity(hidden, dtype, L, d, patch_type, order, stats, sim, stream);
    if (rc) return rc;
    if (L == 0) return FF_OK;
    rc = ff::launch_plan_merge(sim, dtype, order, L, threshold, sub, ratio_lb, run_len, dst, keep, stats, ws,
                               stats_host_mapped, seq, (hipStream_t)stream);
    if (rc) return rc;
    return ff_merge_compact(hidden, hidden_out, dtype, L, d, L_cap, order, run_len, dst, aux_host, n_aux, stream);
}
