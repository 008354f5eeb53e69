/*
 * framefusion_hip.h - C ABI of libframefusion_hip.so, the MI355X (gfx950) implementation of the
 * FrameFusion token-reduction hot path.
 *
 * The reference (thu-nics/FrameFusion) is pure Python/torch and has no FFI; each entry point below
 * replaces the torch-op sequence of one reference function (file:line relative to the reference
 * tree) and is what a binding for that function would call.  The Python host in framefusion_amd/
 * binds them with ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; raw sizes, no torch types;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t; 0 = the null stream): no allocation, no
 *     synchronisation, outputs caller-allocated.  Exceptions, all named: ff_ctx_merge_finish / _wait / _collect
 *     wait on pinned HOST memory for the 256-byte result block; ff_host_alloc / ff_host_free allocate;
 *   - return value: 0 = OK, <0 = FF_ERR_* (nothing enqueued), >0 = hipError_t;
 *   - dtype codes FF_F32 / FF_BF16 / FF_F16 = the activation dtype T the reference computes in; index
 *     outputs are int32 (sequences < 2^31); patch types int64 as in the reference's `patch_type`;
 *   - sequence positions are "i" (0..L-1); by-patch SLOTS are "j" (0..Nv-1): the visual tokens sorted by
 *     (patch type, position) - compute_similarity_and_token_index_by_patch, framefusion/main.py:208-214.
 */
#ifndef FRAMEFUSION_HIP_H
#define FRAMEFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FF_ABI_VERSION 11

enum { FF_F32 = 0, FF_BF16 = 1, FF_F16 = 2 };

enum {
    FF_OK = 0,
    FF_ERR_ARG = -1,          /* null pointer / negative size / unknown dtype                            */
    FF_ERR_ALIGN = -2,        /* pointer or row size not 16-byte aligned                                 */
    FF_ERR_UNSUPPORTED = -3,  /* size outside what the kernels are built for                             */
    FF_ERR_WORKSPACE = -4,    /* workspace smaller than ff_workspace_bytes()                             */
    FF_ERR_DEVICE = -5,       /* a device-side check failed, or the result block was never published     */
    FF_ERR_STATE = -6         /* context call out of order (see the state diagram below); nothing changed */
};

/* Result block: written by the device, read back once per merge call by the host.  int64 each. */
enum {
    FF_STAT_NV = 0,        /* visual tokens with 0 <= type < patch_num (length of the by-patch order)     */
    FF_STAT_FTN = 1,       /* tokens with type != TEXT_TOKEN            (main.py:112)                     */
    FF_STAT_COUNT = 2,     /* #{j : sim[j] >= T(threshold)}             (main.py:113)                     */
    FF_STAT_BRANCH = 3,    /* 0: threshold set (main.py:116-120); 1: top-k (main.py:121-127)              */
    FF_STAT_K = 4,         /* top-k size int(sub * ftn) when BRANCH == 1                                  */
    FF_STAT_MERGED = 5,    /* tokens folded away (= L_in - L_out)                                         */
    FF_STAT_LOUT = 6,      /* output sequence length                                                      */
    FF_STAT_BELOW_LB = 7,  /* 1 if BRANCH == 0 and count/ftn < ratio_lower_bound                          */
    FF_STAT_KTH_KEY = 8,   /* debug: order-preserving key of the k-th largest similarity                  */
    FF_STAT_TIES_TAKEN = 9,/* debug: entries equal to the k-th value that were selected                   */
    FF_STAT_SEQ = 10,      /* sequence number of the call, written LAST (what the host polls)             */
    FF_STAT_ERROR = 11,    /* FF_ERR_BIT_* of the device-side checks; cleared once published              */
    FF_STAT_APPLIED = 12,  /* one-launch call: 1 = outputs written (or nothing folds), 0 = the plan only, */
                           /* 2 = the kernel waits for outputs of LOUT rows by mail                       */
    FF_STAT_ACK = 13,      /* pinned block only: 4 * seq + the mail slot the kernel took (3: none in time) */
    FF_STAT_T_ORDER = 16,  /* device block: 8 diagnostic words.  PINNED block: words 16.. are the host's   */
    FF_STAT_T_PLAN = 24,   /* 7 diagnostic words (phase stamps of the plan / the one-launch kernel)       */
    FF_STAT_WORDS = 32
};
enum {
    FF_ERR_BIT_BARRIER = 1,  /* a plan workgroup never saw a predecessor's total (bounded look-back)       */
    FF_ERR_BIT_LAYOUT = 2,   /* the frame-major layout hint does not describe patch_type: the library      */
                             /* repeats the call through the order kernels (result->unhinted)              */
    FF_ERR_BIT_RESIDENT = 4  /* the one-launch kernel gave up (grid barrier timed out behind another        */
                             /* barrier kernel / sequence not as described): repeated as three launches    */
};

typedef void* ff_stream_t; /* hipStream_t */

int ff_abi_version(void);
/* first 16 hex digits of the SHA-256 over the library's sources as they were when it was compiled */
const char* ff_source_hash(void);
const char* ff_error_string(int code);

/* Scratch bytes for a sequence of L tokens.  Workspace protocol: allocate it ZERO-INITIALISED, pass the SAME
 * (ws, ws_bytes) to every call on one sample.  It holds the select tables (level-0 histogram at the front,
 * per-slice level-1 histograms down from the end) and the one-launch kernel's barrier words: producers fill the
 * tables, the plan consumes them, the merge kernel clears them - every entry point leaves it as it found it.
 * After a failed call zero it again. */
size_t ff_workspace_bytes(int64_t L, int64_t patch_num);

/* ---- stage entry points (one kernel family each; what tests, tools and bench.py time) -------------------------
 * K0 ff_build_order: replaces torch.where(patch_type == arange(P)[:, None]) (main.py:208-210).  order[0..Nv) =
 *   visual positions stable-sorted by type, order[Nv..L) = the rest in sequence order; inv (optional) = inverse;
 *   stats[NV], stats[FTN].  patch_num <= 32768.
 * K1 ff_pair_similarity: the two [Nv-1, d] gathers + cosine_similarity + boundary fill (main.py:216-238, 345-349):
 *   sim[j] = T(T(sum T(a*b)) / T(T(|a|) * T(|b|))) for a = hidden[order[j-1]], b = hidden[order[j]]; -2 at chain starts.
 * ff_plan_merge: main.py:112-127 + the index algebra of merge_tokens_and_get_mask / find_contigious_latter_index
 *   (main.py:269-301, 351-380).  `threshold` already rounded to T.  Decision on device in double like python:
 *   ratio = count/ftn; ratio < sub ? threshold set : top-k with k = (int64)(sub*ftn), ties at the k-th value in
 *   ascending j.  member[L] (by slot: 1 = folded into its run's anchor), keep[L] / dst[L] (by position: kept? / output
 *   row or -1), stats.  L < 983 040.  Stand-alone form: builds tables and inverse itself.
 * ff_plan_from_index: the same outputs for an EXPLICIT merge set (static merge_tokens_and_get_mask, main.py:243-319).
 * K4 ff_merge_compact: index_add_ + divide (main.py:304-317) and the keep-mask gathers (main.py:132-138, 161-178):
 *   out[dst[i]] = T((..(T(h[i] + h[order[t+1]]) + ..) / T(n+1)) for every non-member slot t with n members behind
 *   it (FF_FOLD_MEAN: T(fp32 sum / (n+1)), the baseline's .mean(); FF_FOLD_DROP: members dropped); each aux tensor
 *   [outer, L, row_bytes] gathered likewise into [outer, L_cap, row_bytes]. */
int ff_build_order(const int64_t* patch_type, int64_t L, int64_t patch_num,
                   int32_t* order, int32_t* inv, int64_t* stats, void* ws, size_t ws_bytes, ff_stream_t stream);
int ff_pair_similarity(const void* hidden, int dtype, int64_t L, int64_t d,
                       const int64_t* patch_type, const int32_t* order, const int64_t* stats,
                       void* sim, ff_stream_t stream);
int ff_plan_merge(const void* sim, int dtype, const int32_t* order, int64_t L,
                  double threshold, double sub, double ratio_lb,
                  uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                  void* ws, size_t ws_bytes, ff_stream_t stream);
int ff_plan_from_index(const int64_t* merge_index, int64_t n_merge, const int32_t* order, int64_t L,
                       uint8_t* member, int32_t* dst, uint8_t* keep, int64_t* stats,
                       void* ws, size_t ws_bytes, ff_stream_t stream);

enum { FF_FOLD_DROP = 0, FF_FOLD_SEQUENTIAL = 1, FF_FOLD_MEAN = 2 };
typedef struct {
    const void* src;     /* [outer, L, row_bytes]                         */
    void* dst;           /* [outer, L_cap, row_bytes]                     */
    int64_t row_bytes;   /* bytes per token (>= 1)                        */
    int64_t outer;       /* leading dims folded together (>= 1)           */
    int64_t src_outer_bytes;  /* bytes between outer slices of src; 0 = L * row_bytes (dense).  Lets the [3, 1, L_out, dh]
                                 VIEW a merge call returned for an M-RoPE table go into the next call as it is */
} ff_aux_t;
#define FF_MAX_AUX 4
int ff_merge_compact(const void* hidden, void* hidden_out, int dtype, int64_t L, int64_t d,
                     int64_t L_cap, const int32_t* order, const uint8_t* member, int fold,
                     const int32_t* dst, const uint8_t* keep, const ff_aux_t* aux_host, int n_aux,
                     ff_stream_t stream);

/* Token gathers of the reference's public position handlers (main.py:142-178); aux as above.
 *   by index: output row r = input row index[r] (negative indices count from the end), outputs [outer, n, row_bytes];
 *   by mask:  keep = L bytes 0/1; dst[L] and stats receive the scan (FF_STAT_LOUT = number kept). */
int ff_gather_tokens_by_index(const int64_t* index, int64_t n, int64_t L, const ff_aux_t* aux_host, int n_aux,
                              ff_stream_t stream);
int ff_gather_tokens_by_mask(const uint8_t* keep, int64_t L, int64_t L_cap, int32_t* dst, int64_t* stats,
                             const ff_aux_t* aux_host, int n_aux, ff_stream_t stream);

/* ---- importance (framefusion/utils.py:27-57 + main.py:69-70) -----------------------------------
 * ff_head_mean: attn_w [H, num, S] (T) -> importance [S] = T(mean over H*num in fp32).
 * ff_last_query_attention: p = T(softmax_fp32(T(T(q K^T) * scale) + bias)) for the last `num` queries.  q_last
 *   [H, num, dh]; key (hk, s) at element offset hk * k_head_stride + s * k_key_stride (both 0: contiguous [H_kv, S, dh]);
 *   GQA folded in (head h reads kv head h / (H/H_kv), modeling_qwen2.py:147).  `causal` != 0: the causal bias of
 *   utils.py:34-38, else `bias` (optional [num, S] of T).  weights [H, num, S] / importance [S] optional.  sel_ws
 *   (optional): the workspace of the prune call that will consume the importance - its select tables are accumulated
 *   on the way.  ws: ff_last_query_workspace_bytes() bytes. */
int ff_head_mean(const void* attn_w, int dtype, int64_t H, int64_t num, int64_t S,
                 void* importance, ff_stream_t stream);
size_t ff_last_query_workspace_bytes(int dtype, int64_t H, int64_t num, int64_t S, int64_t dh);
int ff_last_query_attention(const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv,
                            int64_t num, int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride,
                            double scale, int causal, const void* bias,
                            void* weights, void* importance,
                            int64_t sel_lo, int64_t sel_hi, void* sel_ws, size_t sel_ws_bytes,
                            void* ws, size_t ws_bytes, ff_stream_t stream);

/* ---- token layout (the patch_type builders of the reference's packers: llava_video/modeling_llava_video.py:332-336,
 * qwenvl/modeling_qwen2_vl.py:123-127, internvl/modeling_internvl_chat.py:29,59-74, minicpmv:92-98, nvila:51,86-88).
 * ff_token_span: span[0..2] = first / last index with ids[i] == token (-1 if none), number of matches.
 * ff_fill_patch_type: positions begin..begin+count-1 get type (first + offset) % period, everything else -1.
 * ff_patch_type_from_mask: runs of `patch_num` nonzero bytes -> offsets inside the run; span[3] = runs, span[4] =
 *   runs of another length (the caller rejects the layout if nonzero). */
typedef struct { int32_t begin, count, first, period; } ff_segment_t;
int ff_token_span(const int64_t* ids, int64_t n, int64_t token, int64_t* span, ff_stream_t stream);
int ff_fill_patch_type(int64_t* patch_type, int64_t L, const ff_segment_t* segments_host,
                       int64_t n_segments, ff_stream_t stream);
int ff_patch_type_from_mask(const uint8_t* mask, int64_t n, int64_t patch_num, int64_t* patch_type,
                            int64_t* span, ff_stream_t stream);

/* ---- one merge call without a context: the CAPTURABLE form --------------------------------------------------
 * FrameFusion.forward's merge call (main.py:104-138) as three launches enqueued by one host call: K0 (skipped when
 * order_valid != 0 or the layout hint applies) -> K1 -> plan -> K4.  Nothing waits: with stats_host_mapped = NULL it
 * can be captured into a hipGraph (tools/kbench_graph.py captures and replays it; profiles/r02_hipgraph.txt: a
 * replay costs more than the direct launches).  stats_host_mapped (optional): pinned host block that receives the
 * result block, FF_STAT_SEQ = `seq` last.  `addend` (optional [L, d]): rows are T(hidden + addend), the decoder's
 * residual add (models/qwen2/modeling_qwen2.py:64-67) formed in registers.  Layout hint (hint_frames > 0, order_valid
 * == 0): hint_frames frames of patch_num visual tokens typed 0..patch_num-1 behind hint_pre other tokens, TEXT
 * elsewhere (llava_video/modeling_llava_video.py:335) - K1 derives, writes and VERIFIES the order; a mismatch sets
 * FF_ERR_BIT_LAYOUT.  Identity calls (stats[MERGED] == 0): nothing is written.  order_next / inv_next (both or
 * neither): the by-patch order of the COMPACTED sequence for the next call (order_valid = 1). */
int ff_merge_step(const void* hidden, const void* addend, void* hidden_out, int dtype, int64_t L, int64_t d, int64_t L_cap,
                  const int64_t* patch_type, int64_t patch_num, int order_valid,
                  double threshold, double sub, double ratio_lb,
                  int32_t* order, int32_t* inv, void* sim, uint8_t* member, int32_t* dst, uint8_t* keep,
                  int64_t* stats, int64_t* stats_host_mapped, int64_t seq,
                  const ff_aux_t* aux_host, int n_aux,
                  int64_t hint_pre, int64_t hint_frames, int32_t* order_next, int32_t* inv_next,
                  void* ws, size_t ws_bytes, ff_stream_t stream);

/* ---- call context: one FrameFusion.forward call (main.py:40-140) per host call --------------------------------
 * For hosts that keep one `FrameFusion` per sample.  The context owns nothing: it NAMES the per-sample scratch the
 * caller allocated once and carries the state between the calls of a prefill.  Plain host memory: zero it, fill the
 * owner fields; never share it between threads without a lock (the reference's instance is not thread-safe either).
 * All members 8 bytes wide (any FFI can fill them by offset). */
typedef struct ff_ctx {
    /* ---- owner fields: device scratch for sequences of up to `cap` tokens (cap a multiple of 4) ---- */
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
    int64_t* stats_host;   /* [FF_HOST_WORDS] pinned host memory the DEVICE can write; for outputs by mail it must be
                              COHERENT while a kernel runs (ff_host_alloc): words FF_MAIL_WORD.. are the host's  */
    void* ws;              /* ff_workspace_bytes(cap, .) bytes, zero-initialised                        */
    size_t ws_bytes;
    /* ---- library state (zero-initialise; read-only for the owner, except res_off) ---- */
    int64_t seq;           /* number of the last call (echoed by the device)                            */
    int64_t order_len;     /* length of the sequence `order`/`inv` describe; 0 = none                   */
    int64_t dirty;         /* a call died half-way: workspace + stats are reset by the next call        */
    int64_t in_flight;     /* the state of the diagram below: 0 clean, 1 begun, 3 result known, 4 submitted */
    int64_t swaps;         /* number of order <-> order_next exchanges so far (owner mirrors its views) */
    int64_t last_L;        /* input / output length of the merge call whose keep set is still in the     */
    int64_t last_l_out;    /* scratch (0: none): what ff_ctx_gather_mask may be asked for                */
    int64_t cur_nv;        /* visual / non-text tokens of the sequence `order` describes (0: unknown to  */
    int64_t cur_ftn;       /* the host): makes a later merge call eligible for the one-launch kernel     */
    int64_t res_active;    /* the submitted call went out as 1: the one-launch kernel, 2: three launches with a
                              blind, guarded merge kernel; 3: state 3* (a one-launch kernel waits for mail) */
    int64_t res_off;       /* > 0: merge calls left for which the one-launch kernel is not tried (it gave up recently;
                              an owner that keeps two samples in flight on two streams keeps this > 1)   */
} ff_ctx_t;

/* Inputs of one merge call (main.py:104-138).  The output fields (hidden_out, L_cap, aux[].dst, mask_out) are only
 * read by the entry points that enqueue the merge kernel (see the diagram). */
typedef struct ff_merge_call {
    const void* hidden;          /* [L, d] T                                                            */
    const void* addend;          /* optional [L, d] T: rows are T(hidden + addend)                      */
    void* hidden_out;            /* [L_cap, d] T                                                        */
    const int64_t* patch_type;   /* [L]                                                                 */
    int64_t dtype, L, d, L_cap, patch_num;
    int64_t order_valid;         /* != 0: patch_type is what the context's `order` was built for        */
    double threshold, sub, ratio_lb;   /* as ff_plan_merge                                              */
    int64_t force_k;             /* < 0: the threshold / budget policy; >= 0: top-k with this k (the fixed-sparsity
                                    baseline, modeling_qwen2_baseline.py:918-1012)                      */
    int64_t fold;                /* FF_FOLD_SEQUENTIAL (main.py) or FF_FOLD_MEAN (baseline)             */
    int64_t hint_pre, hint_frames;     /* frame-major layout hint, as ff_merge_step                     */
    ff_stream_t stream;
    int64_t n_aux;
    ff_aux_t aux[FF_MAX_AUX];
    const void* mask;            /* optional [L, L] attention mask of mask_elem_bytes per element       */
    void* mask_out;              /* [L_cap, L_cap]: written ONLY when the call folded something         */
    int64_t mask_elem_bytes;
    int64_t late_outputs;        /* ff_ctx_merge_submit of a one-launch call: != 0 = the outputs AND the auxiliary
                                    tensors follow by ff_ctx_merge_mail (n_aux / aux[] are not looked at by submit) */
} ff_merge_call_t;

typedef struct ff_merge_result {
    int64_t nv, ftn, count, branch, k, l_out;
    int64_t error;       /* FF_ERR_BIT_* of the device-side checks (0 on success)                       */
    int64_t unhinted;    /* 1: the layout hint was wrong; the call was repeated through K0 (stop hinting) */
    int64_t wait_ns;     /* time spent polling for the result block (diagnostics)                       */
    int64_t applied;     /* 1: the merge kernel of the call is enqueued (or ran); 0: the plan only - state 3,
                            ff_ctx_merge_apply must follow; 2: a one-launch kernel waits for outputs of l_out rows
                            by mail - state 3*, ff_ctx_merge_apply hands them over / confirms them (no launch) */
} ff_merge_result_t;

/* STATE DIAGRAM of a context (ctx->in_flight).  Every other transition returns FF_ERR_STATE and changes nothing;
 * ff_ctx_reset leads to 0 from anywhere.  tests/test_abi.py tries every (state, entry point) pair.
 *
 *   0 clean ----begin----> 1 begun ----finish (plan + K4, waits)-----------------------------> 0
 *      |                      |------wait (plan, waits)----> 3 result known ----apply (K4)----> 0
 *      |                      '------begin (restart: the workspace is reset)--> 1
 *      '--------submit (K1 + plan + K4, or ONE kernel)----> 4 submitted --[mail]--> 4
 *                                                              '--collect (waits)--> 0   (result->applied = 1)
 *                                                                              '---> 3   (result->applied = 0) --apply (K4)--> 0
 *                                                                              '---> 3*  (result->applied = 2) --[mail]--> 3*
 *                                                                                        --apply (mail if none holds l_out rows, then the
 *                                                                                          kernel's acknowledgement; no launch)--> 0
 *   prune / prune_from_qk / last_query_importance / gather_mask: state 0 only, stay in 0.
 *
 * begin:   (reset if dirty) + K0 unless order_valid / hinted + K1.  Enqueues only; no output field is looked at.
 * finish:  plan + K4 blind into buffers of L rows (L_cap >= L), then WAITS for the result block (published before K4
 *          starts): spin on pinned memory without any HIP call - FFI layers should release their interpreter lock -
 *          stream queried every ~50 us after 1 ms; a wrong layout hint is repeated through K0 inside.  The view form.
 * wait:    the plan alone, then waits: result->l_out.  apply: K4 into outputs of >= l_out rows (l_out == L: nothing is
 *          written, any non-NULL hidden_out, n_aux = 0), mask gather, order swap.  The form for EXACTLY sized outputs
 *          (hidden_states[token_mask, :], main.py:132-138) when no length can be guessed: the GPU idles for the host's
 *          reaction time between plan and K4.
 * submit:  the whole call enqueued by ONE crossing, every output field set (or late, see mail).  Buffers of L rows, or of
 *          a GUESSED length (the top-k branch's L - int(sub * ftn), main.py:122): the merge kernel is guarded on the
 *          device and writes nothing unless the plan's l_out is exactly L_cap - collect then reports applied = 0 and apply
 *          repeats it into buffers of l_out rows.  When the activation fits into the chip's registers + LDS
 *          (ff_ctx_merge_one_launch: bf16, rows <= 8 KiB, <= 56 visual tokens per CU, by-patch order known to the host from
 *          the hint or the previous call, an `addend` only with the previous call's order - the LLaVA-Video-7B / Qwen2-VL-7B
 *          prefills of 90-96 MB) "the whole call" is ONE kernel that reads every row ONCE (csrc/ff_resident.hip); same results
 *          bit for bit; hidden_out = NULL = plan only.
 *          Two such kernels side by side wait for each other's CUs until one gives up (~2 ms, repeated as three launches):
 *          an owner with two samples in flight sets ctx->res_off > 1 on both contexts.
 * mail:    outputs of a late_outputs submit: plain stores of {4 seq + slot, hidden_out, L_cap, n_aux, aux[]} into the pinned
 *          block (two slots of FF_MAIL_WORDS words from FF_MAIL_WORD on; each written once per call), no HIP call; one wave of
 *          the kernel relays them into device memory.  The auxiliary tensors travel whole (sources too): the kernel needs them
 *          only behind its plan, so the host describes them AFTER the launch.  Slot 1 = outputs that exist before the result
 *          does (L rows, or a guessed length): mailed between submit and collect, taken if they hold exactly l_out rows or a
 *          whole input.  Slot 2 = outputs sized to the result (state 3*, ctx->res_active == 3).
 * collect: the wait + bookkeeping of a submitted call (layout retry, give-up retry, mask gather, order swap).  The one-launch
 *          kernel publishes the result block right behind its grid barrier (the barrier carries the two counts that decide
 *          the branch and l_out), ~12 us before its plan is through, and then WAITS for outputs by mail, rows in hand:
 *          applied = 2.  The caller sizes its outputs to l_out (unless slot 1 holds them already) and calls apply, which mails
 *          what is not mailed yet and reads the kernel's acknowledgement (FF_STAT_ACK; two PCIe round trips) - the call block
 *          must name the buffers of the mail that was taken.  A kernel that waited ~4 ms in vain leaves with the plan only; apply
 *          sees that (ACK = 4 seq + 3) and enqueues the merge kernel as in state 3. */
int ff_ctx_merge_begin(ff_ctx_t* ctx, const ff_merge_call_t* call);
int ff_ctx_merge_finish(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);
int ff_ctx_merge_wait(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);
int ff_ctx_merge_apply(ff_ctx_t* ctx, const ff_merge_call_t* call, const ff_merge_result_t* result);
int ff_ctx_merge_submit(ff_ctx_t* ctx, const ff_merge_call_t* call);
int ff_ctx_merge_collect(ff_ctx_t* ctx, const ff_merge_call_t* call, ff_merge_result_t* result);
/* pinned block (ctx->stats_host, FF_HOST_WORDS words): the first FF_MAIL_WORD words are the device's (result block,
 * FF_STAT_ACK), then two mail slots of FF_MAIL_WORDS words the host writes:
 * {4 seq + slot, hidden_out, L_cap, n_aux, aux[FF_MAX_AUX] as ff_aux_t} */
enum { FF_MAIL_WORD = 16, FF_MAIL_WORDS = 24, FF_MAIL_SLOTS = 2, FF_HOST_WORDS = 64 };
int ff_ctx_merge_mail(ff_ctx_t* ctx, const ff_merge_call_t* call);
/* 1 if ff_ctx_merge_submit(ctx, call) would go out as the one-launch kernel now (input half of `call` only) */
int ff_ctx_merge_one_launch(const ff_ctx_t* ctx, const ff_merge_call_t* call);

/* The attention mask of the merge call that just finished on this context, gathered with its keep set (main.py:137-138):
 * out[r, c] = mask[src[r], src[c]], row stride L_cap >= l_out.  For a host that sizes mask_out after it knows l_out. */
int ff_ctx_gather_mask(ff_ctx_t* ctx, const void* mask, void* mask_out, int64_t elem_bytes, int64_t L, int64_t L_cap,
                       ff_stream_t stream);

/* One prune call (main.py:61-101): head mean of the weights [H, num, S] in THEIR dtype (+ select tables), top-k plan over
 * [start, start + n_img), gather by output rows.  H * num == 1: attn_w IS the importance [S]; tables_ready = 1: its producer
 * (ff_ctx_last_query_importance) filled the select tables.  Nothing is read back (the output length S - n_img + k is the
 * caller's arithmetic; L_cap >= that). */
typedef struct ff_prune_call {
    const void* hidden;
    const void* addend;
    void* hidden_out;            /* [L_cap, d]                                                          */
    const void* attn_w;
    int64_t dtype, S, d, L_cap, w_dtype, H, num;
    int64_t tables_ready;
    int64_t start, n_img, k;
    ff_stream_t stream;
    int64_t n_aux;
    ff_aux_t aux[FF_MAX_AUX];
    const void* mask;
    void* mask_out;
    int64_t mask_elem_bytes;
} ff_prune_call_t;
int ff_ctx_prune(ff_ctx_t* ctx, const ff_prune_call_t* call);

/* The attention hook of a context whose prune call comes next (utils.py:27-57 feeding main.py:61-101): the importance of the
 * last `num` queries AND the select tables of importance[start, start + n_img) in the context's workspace. */
int ff_ctx_last_query_importance(ff_ctx_t* ctx, const void* q_last, const void* k, int dtype, int64_t H, int64_t H_kv, int64_t num,
                                 int64_t S, int64_t dh, int64_t k_head_stride, int64_t k_key_stride, double scale, int causal,
                                 const void* bias, void* importance, int64_t start, int64_t n_img, int64_t k_keep,
                                 void* ws, size_t ws_bytes, ff_stream_t stream);

/* Hook + prune in ONE host call: importance into the context's `sim` scratch, its tables, plan, gather.  `call`: as
 * ff_ctx_prune (attn_w / w_dtype / H / num / tables_ready ignored). */
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

/* The caller replaced patch_type / starts a new sample / gave up on a call: forget the order, state 0 (and reset the
 * workspace if a call died half-way).  Enqueues at most two memsets on `stream`. */
int ff_ctx_reset(ff_ctx_t* ctx, ff_stream_t stream);

/* Pinned host memory that host and device see COHERENTLY while a kernel runs (hipHostMalloc coherent | mapped), zeroed: what
 * ctx->stats_host must be for outputs by mail.  The two calls of the ABI that allocate. */
void* ff_host_alloc(size_t bytes);
void ff_host_free(void* p);

/* sizeof() of the structures above as THIS library was compiled (0: ff_ctx_t, 1: ff_merge_call_t, 2: ff_merge_result_t,
 * 3: ff_prune_call_t, 4: ff_aux_t, 5: ff_lq_args_t), so a binding can verify its own layout. */
size_t ff_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* FRAMEFUSION_HIP_H */
