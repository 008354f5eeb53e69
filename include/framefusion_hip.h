/*
 * framefusion_hip.h - C ABI of libframefusion_hip.so, the MI355X (gfx950) implementation of the
 * FrameFusion token-reduction hot path.
 *
 * The reference (thu-nics/FrameFusion) is pure Python/torch and has no FFI; each entry point below
 * replaces the torch-op sequence of one reference function (file:line relative to the reference
 * tree) and is what a binding for that function would call.  The Python host in framefusion_amd/
 * binds them with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; raw sizes, no torch types;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t; 0 = the null stream): no
 *     allocation, no synchronisation, no host read of device data; outputs are caller-allocated;
 *   - return value: 0 = OK, <0 = FF_ERR_* (bad argument, nothing enqueued), >0 = hipError_t;
 *   - dtype codes: FF_F32 / FF_BF16 / FF_F16 = the activation dtype T the reference computes in;
 *   - all index outputs are int32 (sequence lengths < 2^31); patch types are int64 as in the
 *     reference's `patch_type` tensor.
 *   - sequence positions are "i" (0..L-1); by-patch positions are "j" (0..Nv-1): the visual tokens
 *     sorted by (patch type, position), the order compute_similarity_and_token_index_by_patch
 *     (framefusion/main.py:208-214) defines.
 */
#ifndef FRAMEFUSION_HIP_H
#define FRAMEFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FF_ABI_VERSION 10

enum { FF_F32 = 0, FF_BF16 = 1, FF_F16 = 2 };

enum {
    FF_OK = 0,
    FF_ERR_ARG = -1,          /* null pointer / negative size / unknown dtype            */
    FF_ERR_ALIGN = -2,        /* pointer or row size not 16-byte aligned                 */
    FF_ERR_UNSUPPORTED = -3,  /* size outside what the kernels are built for             */
    FF_ERR_WORKSPACE = -4,    /* workspace smaller than ff_workspace_bytes()             */
    FF_ERR_DEVICE = -5,       /* a device-side check failed (ff_merge_result_t.error) or the
                                 result block of a context call was never published      */
    FF_ERR_STATE = -6         /* context call out of order (finish without begin, ...)   */
};

/* Device-side result block written by the plan kernels and read back once per call by the host
 * (the only host<-device traffic of a FrameFusion.forward call). int64 each. */
enum {
    FF_STAT_NV = 0,        /* visual tokens with 0 <= type < patch_num  (len of by-patch order)   */
    FF_STAT_FTN = 1,       /* tokens with type != TEXT_TOKEN            (main.py:112)             */
    FF_STAT_COUNT = 2,     /* #{j : sim[j] >= T(threshold)}             (main.py:113)             */
    FF_STAT_BRANCH = 3,    /* 0: threshold set kept (main.py:116-120); 1: top-k (main.py:121-127) */
    FF_STAT_K = 4,         /* top-k size int(sub * ftn) when BRANCH == 1                          */
    FF_STAT_MERGED = 5,    /* tokens folded away (= L_in - L_out)                                 */
    FF_STAT_LOUT = 6,      /* output sequence length                                              */
    FF_STAT_BELOW_LB = 7,  /* 1 if BRANCH == 0 and count/ftn < ratio_lower_bound                  */
    FF_STAT_KTH_KEY = 8,   /* debug: order-preserving key of the k-th largest similarity          */
    FF_STAT_TIES_TAKEN = 9,/* debug: entries equal to the k-th value that were selected           */
    FF_STAT_SEQ = 10,      /* sequence number, copied from the call (host polling)                */
    FF_STAT_ERROR = 11,    /* bit mask of device-side checks that failed, FF_ERR_BIT_*; cleared once published */
    FF_STAT_APPLIED = 12,  /* one-launch merge call: 1 = the outputs are written (or nothing folds), 0 = the plan only -
                              member / keep / dst are in place, the merge kernel is still to come (L_cap < l_out or no outputs) */
    FF_STAT_T_ORDER = 16,  /* device block only: 8 diagnostic words (K0's cycles; sub-phase stamps of the one-launch kernel) */
    FF_STAT_T_PLAN = 24,   /* 7 words: steady-counter stamps of the plan kernel's / the one-launch kernel's phases (diagnostics) */
    FF_STAT_WORDS = 32
};

enum {
    FF_ERR_BIT_BARRIER = 1,  /* a workgroup of the plan kernel never saw a predecessor's total (bounded look-back) */
    FF_ERR_BIT_LAYOUT = 2,   /* the frame-major layout hint of ff_merge_begin does not describe
                                patch_type: the call's outputs are meaningless, repeat it unhinted  */
    FF_ERR_BIT_RESIDENT = 4  /* the one-launch merge kernel gave up (its grid barrier timed out behind another barrier kernel, or the
                                sequence is not what the host believed): the library repeats the call through the three launches */
};

typedef void* ff_stream_t; /* hipStream_t */

int ff_abi_version(void);
/* first 16 hex digits of the SHA-256 over the library's sources (the .hip files, ff_common.h, this header and
 * the Makefile, concatenated in sorted order) as they were when it was compiled: a host that has the sources
 * next to the binary can tell a stale build from a current one. */
const char* ff_source_hash(void);
const char* ff_error_string(int code);

/* Scratch bytes any entry point may need for a sequence of L tokens and `patch_num` patch types.
 * Workspace protocol: allocate it ZERO-INITIALISED and pass the SAME (ws, ws_bytes) to every call that
 * works on one sample.  It holds the select tables (a level-0 histogram at the front, per-slice
 * level-1 histograms laid out down from the end, so their place depends on neither L nor the call):
 * the producer of the values (similarity / head-mean kernel) accumulates them, the plan kernel consumes
 * them, the merge kernel of the same call clears them again - every entry point leaves the workspace
 * as it found it.  After a failed call zero it again. */
size_t ff_workspace_bytes(int64_t L, int64_t patch_num);

/* ---- K0: by-patch order --------------------------------------------------------------------
 * Replaces torch.where(patch_type == arange(P)[:, None]) (main.py:208-210).
 * order[0 .. Nv)  = sequence index of the visual tokens, stable-sorted by patch type;
 * order[Nv .. L)  = the remaining (text / out-of-range) positions in sequence order, so that
 *                   `order` is a permutation of 0..L-1 that later stages can walk uniformly.
 * inv (optional)  = its inverse: inv[order[t]] = t, the slot of every sequence position (what the
 *                   plan kernel reads to classify positions without walking the whole order).
 * stats[FF_STAT_NV], stats[FF_STAT_FTN] are written.  patch_num <= 32768; patch_type, order and ws
 * 16-byte aligned, ws >= ff_workspace_bytes(L, patch_num) (two launches: per-slice facts, then the
 * closed form of the frame-major layout on every workgroup or the counting sort on one). */
int ff_build_order(const int64_t* patch_type, int64_t L, int64_t patch_num,
                   int32_t* order, int32_t* inv, int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream);

/* ---- K1: adjacent-pair cosine similarity ----------------------------------------------------
 * Replaces the two [Nv-1, d] gathers + cosine_similarity + boundary fill (main.py:216-238,
 * 345-349).  sim[j] (dtype T, j in [0, Nv)) = T(T(sum T(a*b)) / T(T(|a|) * T(|b|))) for
 * a = hidden[order[j-1]], b = hidden[order[j]]; -2 when j == 0 or the two patch types differ.
 * hidden: [L, d] row-major, 16-byte aligned, d*sizeof(T) a multiple of 16. */
int ff_pair_similarity(const void* hidden, int dtype, int64_t L, int64_t d,
                       const int64_t* patch_type, const int32_t* order, const int64_t* stats,
                       void* sim, ff_stream_t stream);

/* ---- K2+K3: select + run detection + compaction scan ----------------------------------------
 * Replaces main.py:112-127 (threshold count, budget test, top-k) and the index algebra of
 * merge_tokens_and_get_mask / find_contigious_latter_index (main.py:269-301, 351-380).
 *   threshold : similarity_lower_bound already rounded to T by the caller (double holding T(thr))
 *   sub       : sparsity upper bound from the budget (main.py:109); ratio_lb: ratio_lower_bound
 * Decision on device, in double like python: ratio = count/ftn; ratio < sub ? threshold set
 * : top-k with k = (int64)(sub*ftn), ties at the k-th value taken in ascending j.
 * Outputs (caller-allocated):
 *   member  [L] uint8 : for slot t of the by-patch order (then the non-visual tail): 1 if token t
 *                       is folded into its nearest preceding non-member slot (its run's anchor,
 *                       main.py:282-301), else 0.  Slot 0 is never a member;
 *   dst     [L] int32 : for each SEQUENCE position i: its row in the compacted output, or -1;
 *   keep    [L] uint8 : the keep mask of main.py:278-279 by sequence position;
 *   stats             : FF_STAT_COUNT .. FF_STAT_TIES_TAKEN, FF_STAT_LOUT, FF_STAT_MERGED.
 * sim, order, dst, keep and ws must be 16-byte aligned (member 8); ws >= ff_workspace_bytes().
 * L < 983 040 (one workgroup per 4096 tokens, all resident: each publishes its kept count as one
 * 8-byte {tag, count} word and sums its predecessors' - no other inter-workgroup traffic).
 * Stand-alone form: builds the select tables and the inverse order itself (extra launches); the fused
 * step gets both from its producers. */
int ff_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L,
                  double threshold, double sub, double ratio_lb,
                  uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                  void* ws, size_t ws_bytes, ff_stream_t stream);

/* Same outputs for the FIXED-SPARSITY policy of the reference's merging baseline
 * (framefusion/models/qwen2/modeling_qwen2_baseline.py:918-1012): the k largest by-patch
 * similarities (k = floor(sparsity * ftn), computed by the caller), ties at the k-th value taken
 * in ascending j.  stats as ff_plan_merge with BRANCH = 1. */
int ff_plan_topk(const void* sim, int dtype, const int32_t* order, int64_t L, int64_t k,
                 uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                 void* ws, size_t ws_bytes, ff_stream_t stream);

/* Same outputs for an EXPLICIT merge set (the static merge_tokens_and_get_mask entry point,
 * main.py:243-319): merge_index[0..n_merge) ascending by-patch positions. */
int ff_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                       uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                       void* ws, size_t ws_bytes, ff_stream_t stream);

/* ---- prune plan (main.py:69-92) ---------------------------------------------------------------
 * importance [S] dtype T; keeps every position outside [start, start+n_img) and the k largest
 * inside it (ties -> lowest index).  Outputs as ff_plan_merge with order = identity; member[i] = 1
 * marks a DROPPED position (nothing is folded: ff_merge_compact is then called with order = NULL
 * and fold = 0).  The [S] int32 scratch inside `ws` also receives src[] = the inverse of dst[] (the
 * position of every output row), which the gather of ff_prune_step walks by output rows. */
int ff_plan_prune(const void* importance, int dtype, int64_t S, int64_t start, int64_t n_img,
                  int64_t k, uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                  void* ws, size_t ws_bytes, ff_stream_t stream);

/* ---- K4: run merge + compaction ---------------------------------------------------------------
 * Replaces index_add_ + divide (main.py:304-317) and the keep-mask gathers of hidden /
 * position embeddings / patch_type (main.py:132-138, 161-178) in one pass.
 * For every slot t with member[t] == 0, i = order[t] (order == NULL: i = t), n = number of
 * consecutive member slots after t (fold = FF_FOLD_*; with FF_FOLD_DROP members are simply dropped):
 *   out[dst[i]] = T( (..(T(h[i] + h[order[t+1]]) + ..) + h[order[t+n]]) / T(n+1) )   (n > 0, SEQUENTIAL)
 *   out[dst[i]] = T( (h[i] + h[order[t+1]] + .. + h[order[t+n]])_fp32 / (n+1) )       (n > 0, MEAN)
 *   out[dst[i]] = h[i]                                                               (n == 0)
 * and each aux tensor (viewed as [outer, L, row_bytes] bytes) is gathered the same way into
 * [outer, L_cap, row_bytes] (every kept position i goes to row dst[i]; `keep` is the plan's keep
 * mask, only read when n_aux > 0). `hidden_out` holds L_cap rows (L_cap >= L_out; L is enough). */
enum {
    FF_FOLD_DROP = 0,        /* members are dropped (prune)                                         */
    FF_FOLD_SEQUENTIAL = 1,  /* main.py:304-317: T-rounded add per member, one T-rounded divide     */
    FF_FOLD_MEAN = 2         /* modeling_qwen2_baseline.py:1034-1048: T(fp32 sum / (n+1)), .mean()  */
};

typedef struct {
    const void* src;     /* [outer, L, row_bytes]                         */
    void* dst;           /* [outer, L_cap, row_bytes]                     */
    int64_t row_bytes;   /* bytes per token (>= 1)                        */
    int64_t outer;       /* leading dims folded together (>= 1)           */
    int64_t src_outer_bytes;  /* bytes from one outer slice of src to the next; 0 = L * row_bytes (dense).  Lets
                                 the [3, 1, L_out, dh] VIEW a merge call returned for an M-RoPE table - rows dense, the
                                 three planes L_cap rows apart - go into the next call as it is (ABI v8)           */
} ff_aux_t;

#define FF_MAX_AUX 4

int ff_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d,
                     int64_t L_cap, const int32_t* order, const uint8_t* member, int fold,
                     const int32_t* dst, const uint8_t* keep, const ff_aux_t* aux_host, int n_aux,
                     ff_stream_t stream);

/* Square attention-mask gather: out[r, c] = mask[src_r, src_c] for kept rows/cols
 * (main.py:137-138, 99-100). mask: [L, L] elements of elem_bytes; out: [L_cap, L_cap]; dst: the plan's row of
 * every position (-1 = dropped); stats: the plan's result block (FF_STAT_LOUT is read on the device);
 * scratch: [L] int32, 16-byte aligned (receives the position of every output row).  Two launches. */
int ff_gather_mask(const void* mask, void* out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                   const int32_t* dst, const int64_t* stats, int32_t* scratch, ff_stream_t stream);

/* Token gathers of the reference's public position handlers (stand-alone: the fused step gathers the position
 * tensors inside its merge kernel).  aux as ff_merge_compact: every tensor viewed as [outer, L, row_bytes].
 *   by index (position_embedding_handler_at_pruning, main.py:142-158: pe[..., keep_indexs, :]): output row r =
 *     input row index[r] (negative indices count from the end, as in torch), outputs [outer, n, row_bytes];
 *   by mask  (position_embedding_handler_at_merging, main.py:161-178: pe[..., token_mask[0], :]): keep = L bytes
 *     0 / 1 (a torch.bool row), 16-byte aligned; dst [L] int32 (16-byte aligned) and stats [FF_STAT_WORDS] receive the
 *     scan (row of every kept position, FF_STAT_LOUT = number kept); outputs [outer, L_cap, row_bytes]. */
int ff_gather_tokens_by_index(const int64_t* index, int64_t n, int64_t L, const ff_aux_t* aux_host, int n_aux,
                              ff_stream_t stream);
int ff_gather_tokens_by_mask(const uint8_t* keep, int64_t L, int64_t L_cap, int32_t* dst, int64_t* stats,
                             const ff_aux_t* aux_host, int n_aux, ff_stream_t stream);

/* ---- importance (framefusion/utils.py:27-57 + main.py:69-70) -----------------------------------
 * Head/query mean of attention probabilities: attn_w [H, num, S] (T) -> importance [S] (T),
 * T(mean over H*num accumulated in fp32). */
int ff_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S,
                 void* importance, ff_stream_t stream);

/* Last-`num`-query attention probabilities with the reference's staged rounding:
 * p = T(softmax_fp32(T(T(q K^T) * scale) + bias)).  q_last [H, num, dh] contiguous; k: S keys of dh elements for
 * each of H_kv heads, key (hk, s) at element offset hk * k_head_stride + s * k_key_stride (both 0: contiguous
 * [H_kv, S, dh]; dh and H_kv * dh: the [S, H_kv, dh] layout a k_proj output has before transpose + copy; strides in
 * elements, 16-byte aligned in bytes, one head's span below 4 GiB)
 * (GQA: head h reads kv head h / (H/H_kv), the repeat_kv of modeling_qwen2.py:147 folded in),
 * weights [H, num, S] (may be NULL), importance [S] (may be NULL) = head_mean(weights).
 * The bias of utils.py:32-44 is either the causal one (`causal` != 0: -inf above the diagonal of the last
 * `num` rows, utils.py:34-38) or `bias` (optional, [num, S] of T, added to every head: the caller's
 * attn_mask already turned into T values - 0 / -inf for a boolean mask, T(mask) for an additive one,
 * utils.py:40-44) - the reference asserts that the two are not combined, so does the Python host.
 * sel_ws (optional, with importance): the workspace of the prune call that will consume the
 * importance - the select tables of importance[sel_lo, sel_hi) are accumulated on the way, and
 * ff_prune_step is then called with H = num = 1, attn_w = importance, tables_ready = 1.
 * ws: ff_last_query_workspace_bytes() bytes, 16-byte aligned, no initialisation needed (scores as T, key-major,
 * + tile statistics + row statistics).
 * Two launches when dh * sizeof(T) / 16 is a power of two (every real head size), else three. */
size_t ff_last_query_workspace_bytes(int dtype, int64_t H, int64_t num, int64_t S, int64_t dh);
int ff_last_query_attention(const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv,
                            int64_t num, int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride,
                            double scale, int causal, const void* bias,
                            void* weights, void* importance,
                            int64_t sel_lo, int64_t sel_hi, void* sel_ws, size_t sel_ws_bytes,
                            void* ws, size_t ws_bytes, ff_stream_t stream);

/* ---- token layout (the patch_type builders of the reference's multimodal packers) -------------
 * The packers find the visual span in the prompt ids with torch.where and build patch_type as a
 * Python list of L ints that is then uploaded (llava_video/modeling_llava_video.py:332-336,
 * qwenvl/modeling_qwen2_vl.py:123-127, internvl/modeling_internvl_chat.py:59-74,
 * minicpmv/modeling_minicpmv.py:92-98, nvila/llava_arch.py:51,86-88).  Here the row is written on
 * the device from a handful of segment descriptors.
 *
 * ff_token_span: span[0] = first index with ids[i] == token (-1 if none), span[1] = last such
 * index (-1), span[2] = number of matches.  `span` may be device memory or device-visible pinned
 * host memory. */
int ff_token_span(const int64_t* ids, int64_t n, int64_t token, int64_t* span, ff_stream_t stream);

/* Positions begin .. begin+count-1 get type (first + offset) % period; everything else
 * TEXT_TOKEN (-1).  Segments must not overlap. */
typedef struct {
    int32_t begin, count, first, period;
} ff_segment_t;

/* patch_type [L] int64 (the reference's dtype); segments_host is HOST memory (copied into the
 * launch arguments, any number of segments). */
int ff_fill_patch_type(int64_t* patch_type, int64_t L, const ff_segment_t* segments_host,
                       int64_t n_segments, ff_stream_t stream);

/* internvl marks the image-context tokens with a boolean mask (`selected`, :29) in which every
 * frame is one run of `patch_num` matches separated by text.  Writes patch_type directly: a
 * position inside a run gets its offset from the start of the run, every other position -1.
 * span[0..2] as ff_token_span (first, last, count of nonzero bytes), span[3] = number of runs,
 * span[4] = number of runs whose length differs from patch_num (the caller rejects the layout if
 * nonzero: the reference's list arithmetic has no meaning for it).  n < 2^31. */
int ff_patch_type_from_mask(const uint8_t* mask, int64_t n, int64_t patch_num, int64_t* patch_type,
                            int64_t* span, ff_stream_t stream);

/* ---- fused step -----------------------------------------------------------------------------
 * One FrameFusion.forward merge call (main.py:104-138): K0 (skipped when order_valid != 0) ->
 * K1 -> K2+K3 -> K4, all enqueued by one host call.  `stats_host_mapped` (may be NULL) is a
 * device-visible pinned host pointer that receives a copy of the stats block, its FF_STAT_SEQ
 * word written last with `seq`, so the host can poll instead of synchronising the stream.
 * Workspace protocol, layout hint, identity calls and order_next as for ff_merge_begin /
 * ff_merge_finish below.  One host call = three launches issued back to back: the form to use when
 * the sequence is short enough that the host, not the similarity pass, would set the pace. */
int ff_merge_step(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                  const int64_t* patch_type, int64_t patch_num, int order_valid,
                  double threshold, double sub, double ratio_lb,
                  int32_t* order, int32_t* inv, void* sim, uint8_t* member, int32_t* dst, uint8_t* keep,
                  int64_t* stats, int64_t* stats_host_mapped, int64_t seq,
                  const ff_aux_t* aux_host, int n_aux,
                  int64_t hint_pre, int64_t hint_frames, int32_t* order_next, int32_t* inv_next,
                  void* ws, size_t ws_bytes, ff_stream_t stream);

/* `addend` (optional, [L, d] like hidden; ff_merge_step / begin / finish / ff_prune_step): the rows that are
 * reduced are T(hidden[i] + addend[i]) - the residual add the decoder performs right before the call
 * (framefusion/models/qwen2/modeling_qwen2.py:64-67: hidden = residual + attention output) formed in
 * registers by both streaming passes: neither the eager add's write nor its re-read reach memory.
 * Pass the same pointer to begin and finish. */

/* The same step in two halves, so the host can allocate the output tensors while the first
 * streaming pass runs: begin = K0 (unless order_valid) + K1, finish = K2+K3 + K4.  `ws` must be the
 * SAME zero-initialised workspace for every call of a sequence (see ff_workspace_bytes) and `seq` a
 * number that differs from call to call (it is echoed in the published result block): the similarity
 * kernel accumulates the select tables of the call, the plan kernel consumes them and the merge
 * kernel clears them for the next call.  `inv` is the inverse of `order` (both [L] int32, both written
 * by begin unless order_valid).
 * Layout hint (hint_frames > 0, only looked at when order_valid == 0): the caller expects the
 * frame-major layout the reference's packers produce - hint_frames frames of patch_num visual
 * tokens typed 0..patch_num-1 behind hint_pre other tokens, TEXT (-1) everywhere else
 * (llava_video/modeling_llava_video.py:335).  The similarity kernel then computes the by-patch
 * order in closed form, writes `order` / stats[NV] / stats[FTN] itself and K0 is not launched;
 * every position's type is verified on the way and a mismatch sets FF_ERR_BIT_LAYOUT in
 * stats[FF_STAT_ERROR] (published with the stats block): the caller discards the call's outputs,
 * zeroes the workspace and repeats the call with hint_frames = 0.
 * Identity calls: when the select folds nothing (stats[FF_STAT_MERGED] == 0, e.g. an empty
 * threshold set, main.py:264-266) the reduced sequence IS the input: the merge kernel exits without
 * writing hidden_out, the aux outputs or order_next, and stats[NV]/[FTN] stay as they are - the
 * caller keeps using its input tensors and its current `order`.
 * order_next / inv_next (optional, both or neither, [L] int32): receive the by-patch order of the
 * COMPACTED sequence and its inverse, and stats[NV]/stats[FTN] are advanced to it, so the next merge
 * call on the reduced sequence can pass them as `order` / `inv` with order_valid = 1 and skip K0
 * (surviving tokens keep their relative order). */
int ff_merge_begin(const void* hidden, const void* addend, int dtype, int64_t L, int64_t d,
                   const int64_t* patch_type, int64_t patch_num, int order_valid, double threshold,
                   int32_t* order, int32_t* inv, void* sim, int64_t* stats, int64_t seq,
                   int64_t hint_pre, int64_t hint_frames, void* ws, size_t ws_bytes,
                   ff_stream_t stream);
int ff_merge_finish(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                    double threshold, double sub, double ratio_lb,
                    const int32_t* order, const int32_t* inv, const void* sim, uint8_t* member, int32_t* dst,
                    uint8_t* keep, int64_t* stats, int64_t* stats_host_mapped, int64_t seq,
                    const ff_aux_t* aux_host, int n_aux, int32_t* order_next, int32_t* inv_next,
                    void* ws, size_t ws_bytes, ff_stream_t stream);

/* ff_merge_finish for the fixed-sparsity baseline: top-k with the caller's k instead of the
 * threshold/budget policy, and `fold` = FF_FOLD_SEQUENTIAL or FF_FOLD_MEAN (the baseline averages
 * each run with .mean(dim=1), modeling_qwen2_baseline.py:1034-1048).  Same workspace protocol. */
int ff_merge_finish_topk(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                         int64_t k, int fold,
                         const int32_t* order, const int32_t* inv, const void* sim, uint8_t* member, int32_t* dst,
                         uint8_t* keep, int64_t* stats, int64_t* stats_host_mapped, int64_t seq,
                         const ff_aux_t* aux_host, int n_aux, int32_t* order_next, int32_t* inv_next,
                         void* ws, size_t ws_bytes, ff_stream_t stream);

/* One FrameFusion.forward prune call (main.py:61-101) from a single host call: head mean of the
 * attention weights [H, num, S] of dtype w_dtype (main.py:69-70: mean and top-k run in the WEIGHTS'
 * dtype, which need not be the activations'), the select + scan (ff_plan_prune) and
 * ff_merge_compact(order = NULL, fold = FF_FOLD_DROP).  H*num == 1: attn_w already IS the importance
 * [S]; tables_ready != 0 then says that its producer (ff_last_query_attention with a workspace) has
 * accumulated the select tables in `ws`.  `importance` is scratch for S values of w_dtype.  The output
 * length S - n_img + k is known to the host, so nothing is read back.  `ws` follows the workspace
 * protocol of ff_merge_begin (zero on entry, left zero). */
int ff_prune_step(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t S, int64_t d, int64_t L_cap,
                  const void* attn_w, int w_dtype, int64_t H, int64_t num, void* importance, int tables_ready,
                  int64_t start, int64_t n_img, int64_t k,
                  uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                  const ff_aux_t* aux_host, int n_aux, void* ws, size_t ws_bytes, ff_stream_t stream);


/* ---- call context (ABI v7) --------------------------------------------------------------------
 * One FrameFusion.forward call (framefusion/main.py:40-140) per host call, for hosts that keep one
 * `FrameFusion` instance per sample: the context owns nothing, it NAMES the per-sample scratch the
 * caller allocated once (so a call passes one pointer instead of ~30 scalars) and carries the state
 * the library keeps between the calls of a prefill (sequence number, which of the two order buffers
 * is current, whether the workspace needs a reset).  A context is plain host memory: zero it, fill
 * the "owner" fields, and never share it between threads without a lock (the reference's instance
 * is not thread-safe either).
 *
 * All structure members are 8 bytes wide (pointers, int64_t, double, size_t) so that any FFI can
 * fill them by offset without padding rules. */
typedef struct ff_ctx {
    /* ---- owner fields: device scratch for sequences of up to `cap` tokens ---- */
    int64_t cap;
    int32_t* order;        /* [cap] by-patch order of the CURRENT sequence (valid when order_len > 0)   */
    int32_t* order_next;   /* [cap] written by a merge call; the library swaps the pair on success      */
    int32_t* inv;          /* [cap] inverse of order                                                    */
    int32_t* inv_next;     /* [cap]                                                                     */
    void* sim;             /* [cap] x 4 bytes: similarities (T) / importance scratch of a prune call    */
    uint8_t* member;       /* [cap] 16-byte aligned                                                     */
    int32_t* dst;          /* [cap]                                                                     */
    uint8_t* keep;         /* [cap]                                                                     */
    int64_t* stats;        /* [FF_STAT_WORDS] device, zero-initialised                                  */
    int64_t* stats_host;   /* [FF_STAT_WORDS] pinned host memory the DEVICE can write (hipHostMalloc); words
                              FF_MAIL_WORD.. are written by the HOST and read by the device (ff_ctx_merge_mail)    */
    void* ws;              /* ff_workspace_bytes(cap, .) bytes, zero-initialised                        */
    size_t ws_bytes;
    /* ---- library state (zero-initialise; read-only for the owner) ---- */
    int64_t seq;           /* number of the last merge call (echoed by the device)                      */
    int64_t order_len;     /* length of the sequence `order`/`inv` describe; 0 = none                   */
    int64_t dirty;         /* a call died half-way: workspace + stats are reset by the next call        */
    int64_t in_flight;     /* 0: no call; 1: begun (K1 enqueued); 2: planned (ff_ctx_merge_plan); 3: result
                              known, merge kernel still to come (ff_ctx_merge_wait); 4: submitted
                              (ff_ctx_merge_submit: everything enqueued, ff_ctx_merge_collect pending)   */
    int64_t swaps;         /* number of order <-> order_next exchanges so far (owner mirrors its views) */
    int64_t last_L;        /* input length of the merge call that finished last and whose keep set is still in
                              the scratch (0: none) - what ff_ctx_gather_mask may be asked for (ABI v9)         */
    int64_t last_l_out;    /* ... and its output length                                                         */
    int64_t cur_nv;        /* visual / non-text tokens of the sequence `order` describes (0: not known to the host) -     */
    int64_t cur_ftn;       /* what makes a later merge call of the prefill eligible for the one-launch kernel (ABI v10)  */
    int64_t res_active;    /* the call in flight went out as the one-launch kernel                                      */
    int64_t res_off;       /* > 0: merge calls left for which the one-launch kernel is not tried (it gave up recently)  */
} ff_ctx_t;

/* Inputs of one merge call (main.py:104-138).  The same structure goes to begin and finish; the
 * output fields (hidden_out, aux[].dst, mask_out) are only read by finish, so the host may allocate
 * them while the similarity pass of begin is running. */
typedef struct ff_merge_call {
    const void* hidden;          /* [L, d] T                                                            */
    const void* addend;          /* optional [L, d] T: rows are T(hidden + addend)                      */
    void* hidden_out;            /* [L_cap, d] T                                                        */
    const int64_t* patch_type;   /* [L]                                                                 */
    int64_t dtype, L, d, L_cap, patch_num;
    int64_t order_valid;         /* != 0: patch_type is what the context's `order` was built for (the
                                    previous merge call's compacted patch_type, untouched since)        */
    double threshold, sub, ratio_lb;   /* as ff_plan_merge                                              */
    int64_t force_k;             /* < 0: the threshold / budget policy above; >= 0: top-k with this k
                                    (the fixed-sparsity baseline, as ff_merge_finish_topk)              */
    int64_t fold;                /* FF_FOLD_SEQUENTIAL (main.py) or FF_FOLD_MEAN (baseline)             */
    int64_t hint_pre, hint_frames;     /* frame-major layout hint, as ff_merge_begin                    */
    ff_stream_t stream;
    int64_t n_aux;
    ff_aux_t aux[FF_MAX_AUX];
    const void* mask;            /* optional [L, L] attention mask of mask_elem_bytes per element       */
    void* mask_out;              /* [L_cap, L_cap]: written ONLY when the call folded something
                                    (result->l_out != L); when l_out == L the sequence - and its mask -
                                    stay as they are and mask_out is left untouched (ABI v8 on)         */
    int64_t mask_elem_bytes;
    int64_t late_outputs;        /* ff_ctx_merge_submit of a one-launch call only: != 0 = hidden_out / L_cap / aux[].dst are
                                    not set yet - they follow by ff_ctx_merge_mail while the kernel reads the rows (ABI v10) */
} ff_merge_call_t;

/* What the host needs from the call (everything else stays on the device). */
typedef struct ff_merge_result {
    int64_t nv, ftn, count, branch, k, l_out;
    int64_t error;       /* FF_ERR_BIT_* of the device-side checks (0 on success)                       */
    int64_t unhinted;    /* 1: the layout hint did not describe patch_type; the call was repeated
                            through K0 inside ff_ctx_merge_finish (stop hinting for this sample)        */
    int64_t wait_ns;     /* time ff_ctx_merge_finish spent polling for the result block (diagnostics)   */
    int64_t applied;     /* 1: the merge kernel of the call is enqueued (or ran); 0: the plan only - the context is in
                            the state ff_ctx_merge_wait leaves it in and ff_ctx_merge_apply must follow (ABI v10:
                            ff_ctx_merge_collect behind a one-launch call whose output buffers were absent or too short) */
} ff_merge_result_t;

/* begin: (reset if dirty) + K0 unless order_valid/hinted + K1.  Enqueues only.
 * finish: plan + K4 (+ mask gather), then WAITS - polling the pinned result block, which the plan
 * kernel publishes before K4 starts - until L_out is known, and advances the context (order swap).
 * The poll spins on host memory without any HIP call; FFI layers should release their interpreter
 * lock around it (ctypes.CDLL does).  If nothing is published within ~1 ms the stream is queried
 * every ~50 us and a failed or drained stream ends the wait with FF_ERR_DEVICE / the hipError_t.
 * A FF_ERR_BIT_LAYOUT result is handled inside finish: workspace reset, the whole call repeated
 * with hint_frames = 0, result->unhinted = 1.
 * ff_ctx_merge = begin + finish (outputs allocated up front).
 * Because finish waits on host memory it cannot be captured into a hipGraph (neither can anything that must learn
 * L_out before it continues); the capturable form of the same three launches is ff_merge_step / ff_merge_begin +
 * ff_merge_finish with stats_host_mapped = NULL (profiles/r02_hipgraph.txt: a replay costs more than the direct
 * launches). */
int ff_ctx_merge_begin(ff_ctx_t* ctx, const ff_merge_call_t* call);
int ff_ctx_merge_finish(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);
int ff_ctx_merge(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);

/* ff_ctx_merge without the wait in the middle (ABI v9): submit = begin + plan + K4, all enqueued (every output field of `call`
 * set on entry); collect = the wait for the result block and the context's bookkeeping (order swap, layout retry, mask gather) -
 * same `call`.  Nothing else may use the context in between.  For a host thread that keeps TWO samples in flight (two contexts,
 * two streams): submit(B) before collect(A), so that A's plan bubble and kernel ramps run under B's streaming pass - the
 * reference's way to load one GPU with two samples is two replicas on two threads (script/demo/llava_video_compare.py:217-223).
 * Results are those of ff_ctx_merge, bit for bit. */
int ff_ctx_merge_submit(ff_ctx_t* ctx, const ff_merge_call_t* call);
int ff_ctx_merge_collect(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);

/* The one-launch merge kernel (ABI v10, csrc/ff_resident.hip): when the whole activation fits into the chip's registers + LDS
 * (bf16, rows of at most 8 KiB, at most 56 visual tokens per CU: the LLaVA-Video-7B and Qwen2-VL-7B prefills the reference ships
 * for - 90-96 MB - do) and the by-patch order is known to the host (layout hint, or the order a previous call of the prefill left),
 * ff_ctx_merge_submit enqueues ONE kernel that reads every row ONCE - similarities, one grid barrier, the plan, the fold from the
 * resident rows - instead of K1, plan and merge kernel.  Same results, bit for bit.  Such a call may be submitted with
 * hidden_out = NULL ("plan only") or with buffers of fewer than l_out rows (outputs sized for the top-k branch's length,
 * main.py:122, while the plan took the threshold branch): ff_ctx_merge_collect then reports result->applied = 0 and
 * ff_ctx_merge_apply - with outputs of l_out rows - finishes the call with the merge kernel alone.
 * Returns 1 if ff_ctx_merge_submit(ctx, call) would take that path now (only the input half of `call` is looked at), else 0.
 * Two such kernels never run side by side on a device without one of them timing out (each needs every CU until its barrier):
 * a host that keeps two samples in flight on two streams sets ctx->res_off to a large number on both contexts. */
int ff_ctx_merge_one_launch(const ff_ctx_t* ctx, const ff_merge_call_t* call);

/* Outputs by mail (ABI v10): a one-launch call needs its output buffers only when its plan is done, ~35 us after the launch.
 * ff_ctx_merge_submit with call->late_outputs != 0 enqueues the kernel at once; the host allocates the outputs while the rows are
 * being read, sets hidden_out / L_cap / aux[].dst in `call` and hands them over with ff_ctx_merge_mail (plain stores into the pinned
 * block behind ctx->stats_host, words FF_MAIL_WORD..; no HIP call), then calls ff_ctx_merge_collect.  The kernel picks the mail up
 * behind its plan (it waits for it if it must: the host writes it unconditionally, before it starts waiting for the result - no
 * cycle).  hidden_out = NULL in the mail = no outputs: the launch stops behind its plan (applied = 0). */
enum { FF_MAIL_WORD = 16, FF_MAIL_WORDS = 8 };      /* {seq, hidden_out, L_cap, n_aux, aux[0..3].dst} */
int ff_ctx_merge_mail(ff_ctx_t* ctx, const ff_merge_call_t* call);

/* The merge call for EXACTLY SIZED outputs (ABI v9) - what the reference returns: hidden_states[token_mask, :], the position
 * embeddings and the attention mask gathered with the same mask (framefusion/main.py:132-138, 161-178).  The host learns l_out
 * before the merge kernel is enqueued and sizes hidden_out / aux[].dst / mask_out to it (L_cap = l_out) - no input-length
 * buffers, no copy out of them.
 *   ff_ctx_merge_begin  K1 (as above)
 *   ff_ctx_merge_plan   the plan kernel, enqueued behind it; no output field of `call` is looked at; nothing is waited for
 *   ff_ctx_merge_wait   waits for the result block (a wrong layout hint is repeated through K0 here): result->l_out
 *   ff_ctx_merge_apply  the merge kernel with the call's (now set) output fields, L_cap >= l_out; mask gather; order swap.
 *                       When l_out == L nothing is written: any non-NULL hidden_out will do, n_aux = 0 (the launch still
 *                       clears the select tables).
 * Between plan and merge kernel the GPU idles for the host's reaction time (~10-20 us); ff_ctx_merge_finish / _submit avoid that
 * by enqueueing the merge kernel blind, into buffers of L rows.  What FrameFusion.forward does by default (compact_outputs),
 * with the outputs of the top-k branch's length (L - int(sub * ftn), main.py:122: host arithmetic) allocated under K1, so that
 * the gap shrinks to the crossing itself when the plan decides that way. */
int ff_ctx_merge_plan(ff_ctx_t* ctx, const ff_merge_call_t* call);
int ff_ctx_merge_wait(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);
int ff_ctx_merge_apply(ff_ctx_t* ctx, const ff_merge_call_t* call, const ff_merge_result_t* result);

/* The attention mask of the merge call that just finished on this context, gathered with its keep set (main.py:137-138):
 * out[r, c] = mask[src[r], src[c]] for the l_out kept positions, row stride L_cap (>= l_out) elements.  FF_ERR_STATE unless
 * a merge call over exactly L tokens finished last on this context, folded something (l_out < L) and nothing was begun since.  For a host that
 * sizes mask_out AFTER it knows l_out - pass call->mask = NULL to ff_ctx_merge_finish and call this behind it (what
 * FrameFusion.forward does: an [L, L] capacity buffer for a 37 k-token call would be 2.7 GB).  ff_ctx_merge_finish with
 * call->mask set gathers into the caller's [L_cap, L_cap] buffer itself - after the result is known, and only when the
 * call folded something. */
int ff_ctx_gather_mask(ff_ctx_t* ctx, const void* mask, void* mask_out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                       ff_stream_t stream);

/* One prune call (main.py:61-101) on the context's scratch: arguments as ff_prune_step.  Nothing is
 * waited for (the output length S - n_img + k is the caller's arithmetic). */
typedef struct ff_prune_call {
    const void* hidden;
    const void* addend;
    void* hidden_out;            /* [L_cap, d]                                                          */
    const void* attn_w;          /* [H, num, S] of w_dtype, or the [S] importance when H * num == 1     */
    int64_t dtype, S, d, L_cap, w_dtype, H, num;
    int64_t tables_ready;        /* 0 / 1; 1: the importance's producer (ff_ctx_last_query_importance) filled the
                                    select tables in ctx->ws                                                */
    int64_t start, n_img, k;
    ff_stream_t stream;
    int64_t n_aux;
    ff_aux_t aux[FF_MAX_AUX];
    const void* mask;
    void* mask_out;
    int64_t mask_elem_bytes;
} ff_prune_call_t;
int ff_ctx_prune(ff_ctx_t* ctx, const ff_prune_call_t* call);

/* The attention hook of a context whose prune call comes next (main.py:61-101 fed by utils.py:27-57): the importance of
 * the last `num` queries as ff_last_query_attention computes it AND the select tables of importance[start, start + n_img)
 * in the context's workspace; the prune call that follows passes attn_w = importance, H = num = 1, tables_ready = 1.
 * The context must be clean (ff_ctx_reset first).  `ws`: ff_last_query_workspace_bytes(). */
int ff_ctx_last_query_importance(ff_ctx_t* ctx, const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv, int64_t num,
                                 int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride, double scale, int causal,
                                 const void* bias, void* importance, int64_t start, int64_t n_img, int64_t k_keep,
                                 void* ws, size_t ws_bytes, ff_stream_t stream);

/* Hook + prune in ONE host call (ABI v9): what the two calls above do back to back - importance of the last `num` queries
 * into the context's `sim` scratch, its select tables, the prune's plan, the gather - for a host that still holds q_last and K
 * when the prune is due (K sits in the layer's KV cache; modeling_qwen2.py:166-178 computes the weights inside attention and
 * main.py:61-101 consumes them after the residual add: nothing in between touches either).  Three launches (four for head
 * sizes off the tiled path), no read-back.  `call`: as ff_ctx_prune; its attn_w / w_dtype / H / num / tables_ready are ignored
 * (the importance has the dtype of q / K).  All members 8 bytes wide. */
typedef struct ff_lq_args {
    const void* q_last;          /* [H, num, dh] contiguous                                              */
    const void* k;               /* keys, as ff_last_query_attention                                     */
    int64_t dtype, H, H_kv, num, dh, k_head_stride, k_key_stride;
    double scale;
    int64_t causal;
    const void* bias;            /* optional [num, S]                                                    */
    void* ws;                    /* ff_last_query_workspace_bytes(dtype, H, num, S, dh) bytes            */
    size_t ws_bytes;
} ff_lq_args_t;
int ff_ctx_prune_from_qk(ff_ctx_t* ctx, const ff_prune_call_t* call, const ff_lq_args_t* lq);

/* The caller replaced patch_type / starts a new sample: forget the order (and reset the workspace
 * if a call died half-way).  Enqueues at most two memsets on `stream`. */
int ff_ctx_reset(ff_ctx_t* ctx, ff_stream_t stream);

/* Marks the workspace as holding select tables a producer outside the context filled (or may have
 * filled) - ff_last_query_attention with sel_ws = ctx->ws: if the matching ff_ctx_prune never comes,
 * the next call resets the workspace. */
void ff_ctx_expect_tables(ff_ctx_t* ctx);

/* Host memory the device and the host both see COHERENTLY while a kernel runs (hipHostMalloc: coherent | mapped) - what
 * ctx->stats_host must be when outputs go by mail (the kernel polls words the host writes mid-launch; the default pinned
 * allocation of most frameworks is only guaranteed visible at kernel boundaries: measured ~600 us late).  The two calls of
 * the ABI that allocate; nothing else does. */
void* ff_host_alloc(size_t bytes);
void ff_host_free(void* p);

/* sizeof() of the structures above as THIS library was compiled (0: ff_ctx_t, 1: ff_merge_call_t,
 * 2: ff_merge_result_t, 3: ff_prune_call_t, 4: ff_aux_t, 5: ff_lq_args_t), so a binding can verify its own layout. */
size_t ff_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* FRAMEFUSION_HIP_H */
