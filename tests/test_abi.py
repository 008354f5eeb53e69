"""CPU: the C-ABI library builds/loads and exports exactly what include/framefusion_hip.h declares;
argument validation happens before any HIP call, so it can be exercised without a GPU."""
import ctypes as C
import os
import re

import pytest

from framefusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "framefusion_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ff_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert "ff_merge_step" in names and "ff_pair_similarity" in names
    assert sorted(_lib.PROTOTYPES) == names


def test_header_is_plain_c(tmp_path):
    """include/framefusion_hip.h is the boundary a C host binds: it must compile as C99 (no C++ in the signatures), and its
    structure sizes there are the ones the library and the ctypes mirrors agree on."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include "framefusion_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(ff_ctx_t), '
                   'sizeof(ff_merge_call_t), sizeof(ff_merge_result_t), sizeof(ff_prune_call_t), sizeof(ff_aux_t)); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(c) for c in (_lib.FFCtx, _lib.FFMergeCall, _lib.FFMergeResult, _lib.FFPruneCall, _lib.FFAux)]


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert _lib.load().ff_abi_version() == _lib.ABI_VERSION == 11


def test_stat_enum_matches_binding():
    text = open(HEADER).read()
    vals = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"FF_STAT_([A-Z_]+)\s*=\s*(\d+)", text))
    assert vals["WORDS"] == _lib.STAT_WORDS
    for k in ("NV", "FTN", "COUNT", "BRANCH", "K", "MERGED", "LOUT", "BELOW_LB", "KTH_KEY", "TIES_TAKEN", "SEQ"):
        assert vals[k] == getattr(_lib, "STAT_" + k), k
    dt = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"FF_(F32|BF16|F16)\s*=\s*(\d+)", text))
    assert (dt["F32"], dt["BF16"], dt["F16"]) == (_lib.FF_F32, _lib.FF_BF16, _lib.FF_F16)


def test_error_strings_and_workspace():
    lib = _lib.load()
    assert lib.ff_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert len(lib.ff_error_string(code)) > 4
    assert lib.ff_workspace_bytes(36864, 576) >= 48
    assert lib.ff_workspace_bytes(-1, 1) == 0
    # the importance workspace is sized for the path the launcher takes: 2-byte scores + statistics on the tiled path, two
    # fp32 [H * num, S] arrays on the general one - which is also where more than 4096 query rows go, whatever the head size
    bf16 = _lib.FF_BF16
    assert lib.ff_last_query_workspace_bytes(bf16, 28, 4, 1000, 128) < 2 * 28 * 4 * 1000 * 4
    assert lib.ff_last_query_workspace_bytes(bf16, 6, 4, 1000, 24) == 2 * 6 * 4 * 1000 * 4           # odd head size
    assert lib.ff_last_query_workspace_bytes(bf16, 28, 256, 1000, 128) == 2 * 28 * 256 * 1000 * 4    # 7168 rows
    # ABI v9: the in-grid hand-over paths (one launch for plan + merge, plan inside the importance kernel) are gone
    assert not hasattr(lib, "ff_set_fused_launch") and not hasattr(lib, "ff_set_fused_prune_plan")


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers / bad sizes are rejected before anything is enqueued
    assert lib.ff_build_order(None, 10, 4, None, None, None, None, 0, None) == -1
    assert lib.ff_pair_similarity(None, 1, 10, 64, None, None, None, None, None) == -1
    assert lib.ff_pair_similarity(16, 7, 10, 64, 16, 16, 16, 16, None) == -1          # unknown dtype
    assert lib.ff_pair_similarity(16, 1, 10, 3, 16, 16, 16, 16, None) == -2           # 6-byte rows: alignment
    assert lib.ff_pair_similarity(24, 1, 10, 64, 16, 16, 16, 16, None) == -2          # base not 16-byte aligned
    assert lib.ff_plan_merge(None, 1, None, 10, 0.6, 0.7, 0.1, None, None, None, None, None, 0, None) == -1
    assert lib.ff_merge_compact(None, None, 1, 10, 64, 10, None, None, 1, None, None, None, 0, None) == -1
    assert lib.ff_head_mean(None, 1, 4, 1, 10, None, None) == -1
    assert lib.ff_last_query_attention(16, 16, 1, 6, 4, 1, 10, 64, 0, 0, 0.1, 1, None, 16, None, 0, 0, None, 0, 16, 1 << 20, None) == -1  # H % H_kv
    # empty problems are a no-op
    assert lib.ff_build_order(16, 0, 4, 16, None, 16, 16, 1 << 24, None) == 0
    assert lib.ff_pair_similarity(16, 1, 0, 64, 16, 16, 16, 16, None) == 0


def test_context_structures_match_the_header():
    """ff_ctx_t / ff_merge_call_t / ff_prune_call_t / ff_merge_result_t: the ctypes mirrors have the library's
    sizes, every member is 8 bytes wide and in the header's order, and the packed-struct formats the host fills
    them with land on the right offsets."""
    lib = _lib.load()
    for which, cls in enumerate((_lib.FFCtx, _lib.FFMergeCall, _lib.FFMergeResult, _lib.FFPruneCall, _lib.FFAux, _lib.FFLqArgs)):
        assert lib.ff_abi_sizeof(which) == C.sizeof(cls), cls.__name__
    assert lib.ff_abi_sizeof(99) == 0
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for cname, cls in (("ff_ctx", _lib.FFCtx), ("ff_merge_call", _lib.FFMergeCall), ("ff_merge_result", _lib.FFMergeResult),
                       ("ff_prune_call", _lib.FFPruneCall), ("ff_lq_args", _lib.FFLqArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s_t;" % (cname, cname), text, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # "int64_t dtype, L, d" / "int32_t* order" / "ff_aux_t aux[FF_MAX_AUX]" / "const void* hidden"
            first, *rest = decl.split(",")
            names.append(re.findall(r"([A-Za-z_0-9]+)(?:\[[A-Z_]+\])?$", first.strip())[0])
            names += [r.strip().lstrip("*") for r in rest]
        assert names == [f[0] for f in cls._fields_], cname
        for fname, ftype in cls._fields_:
            assert C.sizeof(ftype) % 8 == 0, (cname, fname)
    assert _lib.MERGE_CALL_HEAD.size == _lib.FFMergeCall.aux.offset == _lib.FFMergeCall.n_aux.offset + 8
    assert _lib.FFMergeCall.hidden_out.offset == 16
    assert _lib.MERGE_CALL_MASK_OFFSET == _lib.FFMergeCall.mask.offset
    assert _lib.PRUNE_CALL_HEAD.size == _lib.FFPruneCall.aux.offset
    assert _lib.PRUNE_CALL_MASK_OFFSET == _lib.FFPruneCall.mask.offset
    assert _lib.MERGE_RESULT.size == C.sizeof(_lib.FFMergeResult)
    assert _lib.AUX_ENTRY.size == C.sizeof(_lib.FFAux) == 40
    assert _lib.LQ_ARGS.size == C.sizeof(_lib.FFLqArgs) == 112


def test_context_calls_validate_before_any_hip_call():
    lib = _lib.load()
    ctx, call, res = _lib.FFCtx(), _lib.FFMergeCall(), _lib.FFMergeResult()
    a = C.addressof
    assert lib.ff_ctx_merge_begin(None, a(call)) == -1
    assert lib.ff_ctx_merge_begin(a(ctx), None) == -1
    assert lib.ff_ctx_merge_begin(a(ctx), a(call)) == -1           # a zeroed context names no scratch
    assert lib.ff_ctx_merge_finish(a(ctx), a(call), None) == -1
    assert lib.ff_ctx_gather_mask(None, 16, 16, 2, 10, 10, None) == -1
    assert lib.ff_ctx_last_query_importance(None, 16, 16, 1, 4, 4, 1, 10, 64, 0, 0, 0.1, 1, None, 16, 0, 10, 5, 16, 1 << 20, None) == -1
    assert lib.ff_ctx_last_query_importance(a(ctx), 16, 16, 1, 4, 4, 1, 10, 64, 0, 0, 0.1, 1, None, 16, 0, 10, 5, 16, 1 << 20, None) == -1
    assert lib.ff_ctx_gather_mask(a(ctx), 16, 16, 2, 10, 10, None) == -1      # a zeroed context names no scratch
    for f in ("order", "order_next", "inv", "inv_next", "sim", "member", "dst", "keep", "stats", "stats_host", "ws"):
        setattr(ctx, f, 4096)
    ctx.cap = 1024
    call.L = 2048
    assert lib.ff_ctx_merge_begin(a(ctx), a(call)) == -1           # longer than the scratch
    call.L = 512
    assert lib.ff_ctx_merge_begin(a(ctx), a(call)) == -4           # workspace smaller than ff_workspace_bytes(cap)
    ctx.ws_bytes = lib.ff_workspace_bytes(1024, 1)
    assert lib.ff_ctx_merge_finish(a(ctx), a(call), a(res)) == _lib.ERR_STATE     # finish without begin
    # the exact-output flow is a state machine too: plan needs a begun call, wait a planned one, apply a known result
    assert lib.ff_ctx_merge_wait(a(ctx), a(call), a(res)) == _lib.ERR_STATE
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == _lib.ERR_STATE
    assert lib.ff_ctx_merge_collect(a(ctx), a(call), a(res)) == _lib.ERR_STATE
    assert lib.ff_ctx_merge_wait(a(ctx), None, a(res)) == -1 and lib.ff_ctx_merge_wait(a(ctx), a(call), None) == -1
    pc = _lib.FFPruneCall()
    pc.S = 4096
    assert lib.ff_ctx_prune(a(ctx), a(pc)) == -1
    assert lib.ff_ctx_prune(a(ctx), None) == -1
    lq = _lib.FFLqArgs()
    assert lib.ff_ctx_prune_from_qk(a(ctx), a(pc), None) == -1 and lib.ff_ctx_prune_from_qk(a(ctx), None, a(lq)) == -1
    assert lib.ff_ctx_prune_from_qk(a(ctx), a(pc), a(lq)) == -1            # S beyond the scratch
    pc.S, pc.start, pc.n_img, pc.k = 512, 4, 600, 5
    assert lib.ff_ctx_prune_from_qk(a(ctx), a(pc), a(lq)) == -1            # start + n_img > S
    pc.n_img, lq.dtype = 100, 7
    assert lib.ff_ctx_prune_from_qk(a(ctx), a(pc), a(lq)) == -1            # unknown dtype: before any HIP call
    assert lib.ff_ctx_reset(None, None) == -1
    assert len(lib.ff_error_string(_lib.ERR_DEVICE)) > 4 and len(lib.ff_error_string(_lib.ERR_STATE)) > 4


def test_library_carries_the_hash_of_its_sources(monkeypatch, tmp_path):
    """ff_source_hash(): the binary names the sources it was compiled from; load() refuses (rebuilds) a stale one."""
    import shutil
    lib = _lib.load()
    assert lib.ff_source_hash().decode() == _lib.source_hash()
    # a copy of the tree whose sources were edited after the build: the stamp no longer matches
    inc = tmp_path / "include"
    inc.mkdir()
    shutil.copy(HEADER, inc / "framefusion_hip.h")
    pkg = tmp_path / "pkg"
    shutil.copytree(_lib.CSRC, pkg / "csrc", ignore=shutil.ignore_patterns("*.o"))
    with open(pkg / "csrc" / "ff_common.h", "a") as f:
        f.write("\n// edited\n")
    monkeypatch.setattr(_lib, "CSRC", str(pkg / "csrc"))
    assert _lib.source_hash() != lib.ff_source_hash().decode()
    assert _lib.file_stamp() == lib.ff_source_hash().decode()      # the stamp is readable from the file, without loading it
    assert _lib.file_stamp(str(tmp_path / "nope.so")) is None


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "CSRC", str(tmp_path / "no_sources"))       # nothing to build from either
    with pytest.raises(_lib.FrameFusionHipError, match="no CPU/eager fallback"):
        _lib.load()


# the state diagram of include/framefusion_hip.h as a table: entry point -> the states (ctx->in_flight) it is legal in
_MERGE_ENTRIES = {
    "ff_ctx_merge_begin": (0, 1),
    "ff_ctx_merge_finish": (1,),
    "ff_ctx_merge_wait": (1,),
    "ff_ctx_merge_apply": (3,),
    "ff_ctx_merge_submit": (0, 1),
    "ff_ctx_merge_collect": (4,),
    "ff_ctx_merge_mail": (4,),
    "ff_ctx_prune": (0,),
    "ff_ctx_prune_from_qk": (0,),
    "ff_ctx_last_query_importance": (0,),
    "ff_ctx_gather_mask": (0,),
}


def test_every_illegal_transition_is_refused_and_changes_nothing():
    """Every (state, entry point) pair the diagram does not allow returns FF_ERR_STATE before any HIP call and leaves the
    context exactly as it was (so it is still usable: ff_ctx_reset, or the legal call, may follow).  The legal pairs are
    exercised on the GPU (tests/test_gpu_parity.py, tests/test_gpu_abi_host.py); here a legal call may fail for any other
    reason - there is no device - but never with FF_ERR_STATE."""
    lib = _lib.load()
    a = C.addressof
    states = (0, 1, 3, 4)
    assert set(s for legal in _MERGE_ENTRIES.values() for s in legal) <= set(states)
    assert sorted(n for n in _lib.PROTOTYPES if n.startswith("ff_ctx_") and n not in ("ff_ctx_reset", "ff_ctx_merge_one_launch")) == sorted(_MERGE_ENTRIES)

    def fresh(state):
        ctx, call, res, pc, lq = _lib.FFCtx(), _lib.FFMergeCall(), _lib.FFMergeResult(), _lib.FFPruneCall(), _lib.FFLqArgs()
        for f in ("order", "order_next", "inv", "inv_next", "sim", "member", "dst", "keep", "stats", "ws"):
            setattr(ctx, f, 4096)
        host = (C.c_int64 * _lib.HOST_WORDS)()
        ctx.stats_host = C.addressof(host)
        ctx.cap = 1024
        ctx.ws_bytes = lib.ff_workspace_bytes(1024, 1)
        ctx.in_flight = state
        ctx.res_active = 1 if state == 4 else 0
        ctx.seq = 7
        call.L, call.d, call.dtype, call.patch_num, call.fold, call.force_k = 512, 64, _lib.FF_BF16, 8, _lib.FOLD_SEQUENTIAL, -1
        call.hidden = call.patch_type = 4096
        res.l_out = 100
        pc.S, pc.d, pc.n_img, pc.k, pc.H, pc.num, pc.dtype, pc.w_dtype = 512, 64, 100, 5, 1, 1, _lib.FF_BF16, _lib.FF_BF16
        pc.hidden = pc.hidden_out = pc.attn_w = 4096
        pc.L_cap = 512
        lq.dtype, lq.H, lq.H_kv, lq.num, lq.dh = _lib.FF_BF16, 4, 4, 1, 64
        return ctx, call, res, pc, lq, host

    def invoke(name, ctx, call, res, pc, lq):
        fn = getattr(lib, name)
        if name in ("ff_ctx_merge_begin", "ff_ctx_merge_submit", "ff_ctx_merge_mail"):
            return fn(a(ctx), a(call))
        if name in ("ff_ctx_merge_finish", "ff_ctx_merge_wait", "ff_ctx_merge_apply", "ff_ctx_merge_collect"):
            return fn(a(ctx), a(call), a(res))
        if name == "ff_ctx_prune":
            return fn(a(ctx), a(pc))
        if name == "ff_ctx_prune_from_qk":
            return fn(a(ctx), a(pc), a(lq))
        if name == "ff_ctx_last_query_importance":
            return fn(a(ctx), 4096, 4096, _lib.FF_BF16, 4, 4, 1, 512, 64, 0, 0, 0.125, 1, None, 4096, 0, 100, 5, 4096, 1 << 20, None)
        return fn(a(ctx), 4096, 4096, 2, 512, 512, None)          # ff_ctx_gather_mask

    snapshot = lambda ctx: bytes(C.string_at(a(ctx), C.sizeof(ctx)))
    refused = 0
    for name, legal in _MERGE_ENTRIES.items():
        for state in states:
            if state in legal:
                continue
            ctx, call, res, pc, lq, host = fresh(state)
            before = snapshot(ctx)
            assert invoke(name, ctx, call, res, pc, lq) == _lib.ERR_STATE, (name, state)
            assert snapshot(ctx) == before, (name, state)
            refused += 1
            assert lib.ff_ctx_reset(a(ctx), None) == 0 and ctx.in_flight == 0      # ... and the way out works from anywhere
    assert refused == sum(len(states) - len(v) for v in _MERGE_ENTRIES.values())
    # state 3 has one sub-state: a one-launch kernel that published its result and waits for outputs by mail (collect returned
    # applied = 2; ctx->res_active = 3).  There - and only there - a mail is legal in state 3: both slots once, then no more.
    ctx, call, res, pc, lq, host = fresh(3)
    ctx.res_active = 3
    call.hidden_out, call.L_cap = 4096, 100
    assert lib.ff_ctx_merge_mail(a(ctx), a(call)) == 0 and host[_lib.MAIL_WORD] == 7 * 4 + 1 and host[_lib.MAIL_WORD + 2] == 100
    assert lib.ff_ctx_merge_mail(a(ctx), a(call)) == 0 and host[_lib.MAIL_WORD + _lib.MAIL_WORDS] == 7 * 4 + 2
    assert lib.ff_ctx_merge_mail(a(ctx), a(call)) == _lib.ERR_STATE


def test_apply_in_the_waiting_kernel_state_mails_and_confirms_without_a_device():
    """State 3* (collect returned applied = 2: a one-launch kernel waits for outputs by mail).  ff_ctx_merge_apply is pure host work
    until the kernel's acknowledgement is in: which slot it writes, which mail counts as sent (slot 1 only if it holds EXACTLY l_out
    rows or a whole input), that the call block must name the mailed buffers - checked here with the acknowledgement pre-written
    into the pinned block, so no HIP call is ever made."""
    lib = _lib.load()
    a = C.addressof
    L, l_out, seq = 512, 100, 7

    def fresh():
        ctx, call, res = _lib.FFCtx(), _lib.FFMergeCall(), _lib.FFMergeResult()
        for f in ("order", "order_next", "inv", "inv_next", "sim", "member", "dst", "keep", "stats", "ws"):
            setattr(ctx, f, 4096 + 64 * len(f))
        host = (C.c_int64 * _lib.HOST_WORDS)()
        ctx.stats_host = C.addressof(host)
        ctx.cap, ctx.ws_bytes = 1024, lib.ff_workspace_bytes(1024, 1)
        ctx.in_flight, ctx.res_active, ctx.seq, ctx.dirty = 3, 3, seq, 1
        host[_lib.STAT_LOUT] = l_out
        call.L, call.d, call.dtype, call.patch_num, call.fold, call.force_k = L, 64, _lib.FF_BF16, 8, _lib.FOLD_SEQUENTIAL, -1
        call.hidden = call.patch_type = 4096
        res.l_out, res.nv, res.ftn = l_out, 400, 400
        return ctx, call, res, host

    M, W = _lib.MAIL_WORD, _lib.MAIL_WORDS
    # nothing mailed yet: apply writes slot 1 itself and takes the kernel's word for it
    ctx, call, res, host = fresh()
    call.hidden_out, call.L_cap = 0x10000, l_out
    host[_lib.STAT_ACK] = seq * 4 + 1
    order, order_next = ctx.order, ctx.order_next
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == 0
    assert host[M] == seq * 4 + 1 and host[M + 1] == 0x10000 and host[M + 2] == l_out
    assert (ctx.in_flight, ctx.res_active, ctx.dirty, ctx.order_len) == (0, 0, 0, l_out)
    assert (ctx.order, ctx.order_next) == (order_next, order)               # the order swap of a call that folded
    assert (ctx.cur_nv, ctx.cur_ftn) == (400 - (L - l_out), 400 - (L - l_out))
    # a guess of another length sits in slot 1: it does not count, the exact outputs go into slot 2
    ctx, call, res, host = fresh()
    call.hidden_out, call.L_cap = 0x20000, l_out + 30
    assert lib.ff_ctx_merge_mail(a(ctx), a(call)) == 0 and host[M] == seq * 4 + 1
    call.hidden_out, call.L_cap = 0x30000, l_out
    host[_lib.STAT_ACK] = seq * 4 + 2
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == 0
    assert host[M + W] == seq * 4 + 2 and host[M + W + 1] == 0x30000 and host[M + W + 2] == l_out and ctx.in_flight == 0
    # the guess came true (slot 1 holds exactly l_out rows): nothing more is mailed, but the call block must name THOSE buffers
    ctx, call, res, host = fresh()
    call.hidden_out, call.L_cap = 0x20000, l_out
    assert lib.ff_ctx_merge_mail(a(ctx), a(call)) == 0
    call.hidden_out = 0x50000
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == _lib.ERR_ARG and ctx.in_flight == 3 and host[M + W] == 0
    call.hidden_out = 0x20000
    host[_lib.STAT_ACK] = seq * 4 + 1
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == 0 and host[M + W] == 0 and ctx.in_flight == 0
    # input-length buffers in slot 1 (the view form) hold any result
    ctx, call, res, host = fresh()
    call.hidden_out, call.L_cap = 0x20000, L
    assert lib.ff_ctx_merge_mail(a(ctx), a(call)) == 0
    host[_lib.STAT_ACK] = seq * 4 + 1
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == 0 and host[M + W] == 0
    # outputs that do not hold the result, or a result that is not this call's: refused, nothing changes
    ctx, call, res, host = fresh()
    call.hidden_out, call.L_cap = 0x20000, l_out - 1
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == _lib.ERR_ARG and ctx.in_flight == 3 and host[M] == 0
    call.L_cap, res.l_out = l_out, l_out + 1
    assert lib.ff_ctx_merge_apply(a(ctx), a(call), a(res)) == _lib.ERR_ARG and ctx.in_flight == 3
