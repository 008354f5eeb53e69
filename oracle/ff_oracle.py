"""CPU oracle for the FrameFusion token-reduction hot path.

TEST INFRASTRUCTURE ONLY.  This file is a torch-CPU restatement of the algorithm in the
reference's ``framefusion/main.py`` (and the importance-weights helper of
``framefusion/utils.py:27-57``).  It is the checker the HIP path is compared against; it is
never imported by the product package ``framefusion_amd``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.

Pinning: ``oracle/make_golden.py`` imports the real reference from ``/root/reference`` (build
container only), runs both on the same seeded inputs, asserts bit-equality of every output and
writes the fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py`` re-checks this file
against those fixtures wherever the repo travels.

Every function builds its temporaries on the device of its inputs, so the same restatement also
runs as "PyTorch eager on the GPU" (tests/bench_eager_gpu.py times it there as a second baseline); the
parity tests always run it on CPU.

The arithmetic is deliberately expressed with the same torch primitives the reference uses
(activation-dtype ``mul``/``sum``/``norm``/``div``) because the result is defined by their staged
rounding; the *structure* (ordering, run detection, merging, selection)
is written independently as plain index arithmetic.

One documented deviation: ``torch.topk`` on CPU breaks ties at the k-th value arbitrarily
(``main.py:122``); the oracle (and the HIP path) take ties in ascending by-patch index, which is
what the reference's own GPU ``topk`` (radix select + ordered gather) does.  Fixtures whose cut is
unique are compared index-for-index; tie fixtures are compared as value multisets.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

TEXT_TOKEN = -1      # main.py:5
IGNORE_TOKEN = -2    # main.py:6


# --------------------------------------------------------------------------------------
# a2: compute budget (main.py:321-343).  Pure python floats; operation order preserved.
# --------------------------------------------------------------------------------------
def budget(sparsity_list: Sequence[float], cost: float, num_layers: int = 28) -> float:
    kept = 1
    spent = 0
    for sp in sparsity_list:
        kept *= (1 - sp)
        spent += kept
    remaining = num_layers * cost - spent
    if remaining < 0:
        raise ValueError("The cost is too small")
    share = remaining / ((num_layers - len(sparsity_list)) * kept)
    if share > 1:
        return 0
    return 1 - share


# --------------------------------------------------------------------------------------
# a3: by-patch order + staged-rounding cosine similarity (main.py:180-241, 345-349)
# --------------------------------------------------------------------------------------
def by_patch_order(patch_type: torch.Tensor, patch_num: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """order[j] = sequence index of the j-th visual token when sorted by (patch type, position).

    Equivalent to the row-major nonzero of the [P, L] one-hot compare at main.py:208-210:
    only types in [0, patch_num) take part; the sort is stable in position.
    """
    pt = patch_type.reshape(-1)
    pos = torch.nonzero((pt >= 0) & (pt < patch_num)).reshape(-1)
    rank = torch.argsort(pt[pos], stable=True)
    order = pos[rank]
    return order, pt[order]


def staged_cosine(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Cosine similarity in the activation dtype with the reference's staged roundings
    (main.py:345-349): T(sum T(a*b)) / T(T(|a|) * T(|b|))."""
    dot = torch.sum(a * b, dim=-1)
    return dot / (torch.norm(a, dim=-1) * torch.norm(b, dim=-1))


def pair_similarity(hidden: torch.Tensor, patch_type: torch.Tensor, patch_num: int):
    """Returns (sim [1, Nv] act dtype, order [1, Nv] int64).  sim[0] and every entry whose
    predecessor has another patch type hold IGNORE_TOKEN (main.py:225-238)."""
    assert hidden.shape[0] == 1, "Only support batch size 1"          # main.py:203
    order, ptype_sorted = by_patch_order(patch_type, patch_num)
    rows = hidden[0]
    sim = staged_cosine(rows[order[:-1]], rows[order[1:]])
    sim[ptype_sorted[:-1] != ptype_sorted[1:]] = IGNORE_TOKEN
    head = torch.full((1,), IGNORE_TOKEN, dtype=hidden.dtype, device=hidden.device)
    sim = torch.cat((head, sim))
    assert sim.shape[0] == order.shape[0]                               # main.py:240
    return sim[None, :], order[None, :]


# --------------------------------------------------------------------------------------
# a5: run-length encoding (main.py:351-380)
# --------------------------------------------------------------------------------------
def run_lengths(flags: torch.Tensor) -> torch.Tensor:
    """[B, n] 0/1 tensor -> same dtype, the length of every run of ones stored at the run's
    last element, zero elsewhere.  KAT (main.py:361-363): 0 1 1 1 0 0 1 1 -> 0 0 0 3 0 0 0 2."""
    out = torch.zeros_like(flags)
    for b in range(flags.shape[0]):
        row = flags[b]
        ones = (row == 1)
        pad = torch.zeros(1, dtype=torch.bool, device=flags.device)
        rises = torch.nonzero(ones & ~torch.cat((pad, ones[:-1]))).reshape(-1)
        falls = torch.nonzero(ones & ~torch.cat((ones[1:], pad))).reshape(-1)
        out[b, falls] = (falls - rises + 1).to(flags.dtype)
    return out


# --------------------------------------------------------------------------------------
# a4: selection (main.py:112-127).  Tie rule: ascending by-patch index (see module docstring).
# --------------------------------------------------------------------------------------
def topk_lowest_index(values: torch.Tensor, k: int) -> torch.Tensor:
    """Indices (ascending) of the k largest entries of a 1-D tensor; NaN ranks highest (as in
    torch.topk); ties at the cut go to the lowest index."""
    if k <= 0:
        return torch.empty(0, dtype=torch.long, device=values.device)
    ranked = torch.sort(values, descending=True, stable=True).indices[:k]
    return torch.sort(ranked).values


# --------------------------------------------------------------------------------------
# a6: run merge (main.py:243-319).  Functional: returns a new tensor.
# --------------------------------------------------------------------------------------
def merge_rows(hidden: torch.Tensor, order: torch.Tensor, merge_idx: torch.Tensor):
    """hidden [1, L, d]; order [1, Nv]; merge_idx ascending by-patch positions to fold into
    their run's anchor (the element just before the run).  Every anchor row becomes
    ((((a + m1) + m2) + ...) / (n+1)) with a rounding to the activation dtype after every add
    and after the divide - the order index_add_ applies on CPU (main.py:304-317).
    Returns (hidden_after [1, L, d], keep [1, L] bool)."""
    L = hidden.shape[1]
    keep = torch.ones(1, L, dtype=torch.bool, device=hidden.device)
    out = hidden.clone()
    if merge_idx.numel() == 0:                                          # main.py:264-266
        return out, keep
    order = order.reshape(-1)
    nv = order.shape[0]
    flags = torch.zeros(nv, dtype=torch.long, device=hidden.device)
    flags[merge_idx] = 1
    keep[0, order[merge_idx]] = False                                   # main.py:278-279
    # run lengths are stored in the activation dtype by the reference (main.py:269-276): exact
    # up to 256 (bf16) / 2048 (fp16); the oracle mirrors that storage.
    lens = run_lengths(flags[None, :].to(hidden.dtype))[0].to(torch.long)
    ends = torch.nonzero(lens).reshape(-1)
    n = lens[ends]
    anchors = ends - n            # may be -1 when a run starts at 0: wraps like python indexing
    rows = out[0]
    step = 1
    live = torch.ones_like(n, dtype=torch.bool)
    while True:
        live = n >= step
        if not bool(live.any()):
            break
        a_seq = order[anchors[live]]
        m_seq = order[anchors[live] + step]
        rows[a_seq] = rows[a_seq] + rows[m_seq]
        step += 1
    a_seq = order[anchors]
    rows[a_seq] = rows[a_seq] / (n[:, None] + 1)
    return out, keep


# --------------------------------------------------------------------------------------
# position-embedding containers (main.py:142-178)
# --------------------------------------------------------------------------------------
def gather_position_embeddings(position_embeddings, index: torch.Tensor):
    """index: int64 positions or a bool mask over the token axis."""
    if type(position_embeddings) == list:
        assert len(position_embeddings) == 2
        for i in range(2):
            t = position_embeddings[i]
            position_embeddings[i] = t[:, :, index, :] if t.ndim == 4 else t[:, index, :]
        return position_embeddings
    if type(position_embeddings) == torch.Tensor:
        if position_embeddings.ndim != 2:
            raise NotImplementedError("Only support 2D position embeddings")
        return position_embeddings[:, index]
    raise NotImplementedError("Only support list or tensor for position embeddings")


def _as_int(x) -> int:
    return int(x.item()) if isinstance(x, torch.Tensor) else int(x)


# --------------------------------------------------------------------------------------
# a0, P, a1, a7, a8: the state machine (main.py:8-140)
# --------------------------------------------------------------------------------------
class OracleFrameFusion:
    def __init__(self, cost=0.3, similarity_lower_bound=0.6, ratio_lower_bound=0.1):
        self.cost = cost
        self.similarity_lower_bound = similarity_lower_bound
        self.ratio_lower_bound = ratio_lower_bound

    def prepare(self, patch_type, patch_num, image_token_start_index, image_token_end_index,
                image_token_length, original_length, finish_merging=False, finish_pruning=False,
                sparsity_list: Optional[List[float]] = None):
        self.patch_type = patch_type
        self.patch_num = patch_num
        self.image_token_start_index = image_token_start_index
        self.image_token_end_index = image_token_end_index
        self.image_token_length = image_token_length
        self.original_length = original_length
        self.finish_merging = finish_merging
        self.finish_pruning = finish_pruning
        self.sparsity_list = [] if sparsity_list is None else sparsity_list

    # -- prune once (main.py:61-101) -----------------------------------------------------
    def _prune(self, hidden, position_embeddings, attention_mask, attn_w):
        q_len = hidden.shape[1]
        start = _as_int(self.image_token_start_index)
        n_img = _as_int(self.image_token_length - (self.original_length - q_len))
        importance = torch.mean(attn_w, dim=(1, 2))[0]
        ratio = budget(self.sparsity_list, self.cost)
        k = round(n_img * (1 - ratio))
        top = topk_lowest_index(importance[start:start + n_img], k) + start
        dev = hidden.device
        keep = torch.cat((torch.arange(start, device=dev), top, torch.arange(start + n_img, q_len, device=dev)))
        hidden = hidden[:, keep, :]
        position_embeddings = gather_position_embeddings(position_embeddings, keep)
        if attention_mask is not None:
            attention_mask = attention_mask[:, :, keep, :][:, :, :, keep]
        self.finish_pruning = True
        self.last_keep = keep
        return hidden, position_embeddings, attention_mask

    # -- merge (main.py:104-138) -----------------------------------------------------------
    def _merge(self, hidden, position_embeddings, attention_mask):
        upper = budget(self.sparsity_list, self.cost)
        sim, order = pair_similarity(hidden, self.patch_type, self.patch_num)
        frame_tokens = int((self.patch_type != TEXT_TOKEN).sum())
        merge_idx = torch.nonzero(sim[0] >= self.similarity_lower_bound).reshape(-1)
        ratio = merge_idx.shape[0] / frame_tokens
        if ratio < upper:
            self.sparsity_list.append(ratio)
            if ratio < self.ratio_lower_bound:
                self.finish_merging = True
        else:
            merge_idx = topk_lowest_index(sim[0], int(upper * frame_tokens))
            self.finish_merging = True
            self.finish_pruning = True
        merged, keep = merge_rows(hidden, order, merge_idx)
        sel = keep[0]
        self.patch_type = self.patch_type.reshape(1, -1)[:, sel]
        hidden = merged[:, sel, :]
        position_embeddings = gather_position_embeddings(position_embeddings, sel)
        if attention_mask is not None:
            attention_mask = attention_mask[:, :, sel, :][:, :, :, sel]
        self.last_keep = torch.nonzero(sel).reshape(-1)
        self.last_sim, self.last_order, self.last_merge_idx = sim, order, merge_idx
        return hidden, position_embeddings, attention_mask

    def forward(self, hidden_states, position_embeddings, attention_mask, self_attn_weights=None):
        q_len = hidden_states.shape[1]
        do_prune = q_len > 1 and self.finish_merging == True and self.finish_pruning == False
        do_merge = q_len > 1 and (not self.finish_merging)
        if do_prune:
            hidden_states, position_embeddings, attention_mask = self._prune(
                hidden_states, position_embeddings, attention_mask, self_attn_weights)
        if do_merge:
            hidden_states, position_embeddings, attention_mask = self._merge(
                hidden_states, position_embeddings, attention_mask)
        return hidden_states, position_embeddings, attention_mask

    __call__ = forward


# --------------------------------------------------------------------------------------
# a9: attention probabilities of the last `num` queries (utils.py:27-57)
# --------------------------------------------------------------------------------------
def last_query_attention(query, key, num=1, is_causal=False, scale=None, enable_gqa=False, attn_mask=None):
    """query [1, H, L, dh], key [1, Hk, S, dh] -> probabilities [1, H, num, S] in the activation
    dtype: softmax(T(T(q K^T) * scale) + bias); bias = causal (utils.py:34-38) or the caller's attn_mask
    (utils.py:40-44: -inf where a boolean mask is False, else the mask added into the T-typed bias)."""
    q = query[:, :, -num:, :]
    n_q, n_k = q.shape[-2], key.shape[-2]
    factor = 1 / math.sqrt(q.shape[-1]) if scale is None else scale
    bias = torch.zeros(n_q, n_k, dtype=q.dtype)
    if is_causal:
        assert attn_mask is None                                                    # utils.py:35
        future = torch.ones(n_q, n_k, dtype=torch.bool).triu(diagonal=n_k - n_q + 1)
        bias.masked_fill_(future, float("-inf"))
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            bias.masked_fill_(attn_mask.logical_not(), float("-inf"))
        else:
            bias += attn_mask
    if enable_gqa:
        key = key.repeat_interleave(q.shape[-3] // key.shape[-3], -3)
    w = q @ key.transpose(-2, -1) * factor
    w += bias
    return torch.softmax(w, dim=-1)


# --------------------------------------------------------------------------------------
# §8(f) rank 4: the fixed-sparsity merging baseline
# (framefusion/models/qwen2/modeling_qwen2_baseline.py:26-43, 905-1053, 1081-1085, 1180-1185)
# --------------------------------------------------------------------------------------
def density_overhead(sparsity_list: Sequence[float]) -> Tuple[float, float]:
    """compute_density_overhead (:26-43): (mean cumulative density, final density)."""
    cost = 0.0
    remaining = 1.0
    for s in sparsity_list:
        remaining *= 1 - s
        cost += remaining
    return cost / len(sparsity_list), remaining


def mean_merge_rows(hidden: torch.Tensor, order: torch.Tensor, merge_idx: torch.Tensor):
    """Every run's anchor row becomes ``rows[anchor .. anchor+n].mean(dim=0)`` in the activation
    dtype (:1034-1048: gather [R, n+1, d], ``.mean(dim=1)`` = fp32 sum / (n+1), one rounding),
    processed per distinct run length in ascending order like the reference (:1021-1048; runs are
    disjoint, so the order does not matter).  Returns (hidden_after [1, L, d], keep [1, L])."""
    L = hidden.shape[1]
    keep = torch.ones(1, L, dtype=torch.bool, device=hidden.device)
    out = hidden.clone()
    order = order.reshape(-1)
    flags = torch.zeros(order.shape[0], dtype=torch.long, device=hidden.device)
    flags[merge_idx] = 1
    keep[0, order[merge_idx]] = False                                   # :1011-1012
    lens = run_lengths(flags[None, :].to(hidden.dtype))[0].to(torch.long)   # :1014 (act dtype storage)
    rows = out[0]
    for n in torch.unique(lens).tolist():                               # :1016-1018
        if n <= 0:
            continue
        ends = torch.nonzero(lens == n).reshape(-1)
        span = (ends - n)[:, None] + torch.arange(n + 1, device=ends.device)[None, :]       # :1024-1032
        rows[order[ends - n]] = rows[order[span]].mean(dim=1)           # :1034-1048
    return out, keep


def fixed_sparsity_merge(hidden: torch.Tensor, patch_type: torch.Tensor, patch_num: int, sparsity: float,
                         position_embeddings=None, residual: Optional[torch.Tensor] = None):
    """One layer of the merging baseline (:916-1052 inside the attention forward, + the residual
    and rotary gathers of :1180-1185 and :1081-1085).  hidden [1, L, d] (the normed activations),
    patch_type [1, L].  Returns dict(hidden, patch_type, token_mask or None, position_embeddings,
    residual, prune_num, sim, order)."""
    out = dict(hidden=hidden, patch_type=patch_type, token_mask=None, position_embeddings=position_embeddings,
               residual=residual, prune_num=0, sim=None, order=None)
    if hidden.shape[1] <= 1:                                            # :916 (prefill only)
        return out
    ftn = int((patch_type != TEXT_TOKEN).sum())                         # :919
    prune_num = math.floor(sparsity * ftn)                              # :920
    out["prune_num"] = prune_num
    if prune_num <= 0:                                                  # :922
        return out
    pt = patch_type.reshape(-1)
    for p in range(patch_num):                                          # :982-983
        if not bool((pt == p).any()):
            raise ValueError("No token in this patch")
    sim, order = pair_similarity(hidden, patch_type, patch_num)        # :941-991 (same cosine, same order)
    merge_idx = topk_lowest_index(sim[0], prune_num)                    # :1001 (tie rule: module docstring)
    merged, keep = mean_merge_rows(hidden, order, merge_idx)
    out.update(hidden=merged[:, keep[0]], patch_type=patch_type.reshape(1, -1)[:, keep[0]], token_mask=keep,
               sim=sim, order=order)
    if position_embeddings is not None:                                 # :1081-1085 (3-D cos/sin lists)
        out["position_embeddings"] = [t[:, keep[0], :] for t in position_embeddings]
    if residual is not None:                                            # :1180-1185
        out["residual"] = residual[:, keep[0], :]
    return out
