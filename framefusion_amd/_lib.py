"""ctypes binding of libframefusion_hip.so (the C ABI declared in include/framefusion_hip.h).

There is no fallback: if the library is missing or fails to load, every product entry point
raises.  ``import torch`` happens first on purpose - the HIP runtime PyTorch-ROCm has already
loaded (SONAME libamdhip64.so.7) is then the one the kernels run on, so ``tensor.data_ptr()`` and
``torch.cuda.current_stream().cuda_stream`` are directly usable across the boundary.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libframefusion_hip.so")
CSRC = os.path.join(_HERE, "csrc")

FF_F32, FF_BF16, FF_F16 = 0, 1, 2
DTYPE_CODE = {torch.float32: FF_F32, torch.bfloat16: FF_BF16, torch.float16: FF_F16}

(STAT_NV, STAT_FTN, STAT_COUNT, STAT_BRANCH, STAT_K, STAT_MERGED, STAT_LOUT, STAT_BELOW_LB,
 STAT_KTH_KEY, STAT_TIES_TAKEN, STAT_SEQ) = range(11)
STAT_T_ORDER, STAT_T_PLAN = 16, 24
FOLD_DROP, FOLD_SEQUENTIAL, FOLD_MEAN = 0, 1, 2
STAT_ERROR = 11
ERR_BIT_BARRIER, ERR_BIT_LAYOUT, ERR_BIT_RESIDENT = 1, 2, 4
STAT_APPLIED = 12
STAT_WORDS = 32
STAT_ACK = 13               # pinned block: 4 * seq + the mail slot the one-launch kernel took (3: none in time)
MAIL_WORD, MAIL_WORDS, MAIL_SLOTS, HOST_WORDS = 16, 24, 2, 64       # pinned words 16..63: the two mail slots of a late_outputs call
MAX_AUX = 4
ABI_VERSION = 11
ERR_ARG = -1
ERR_DEVICE, ERR_STATE = -5, -6


class FFAux(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64), ("outer", C.c_int64),
                ("src_outer_bytes", C.c_int64)]


class FFCtx(C.Structure):
    """ff_ctx_t: names the per-sample scratch + the state the library keeps between the calls of a prefill."""
    _fields_ = [("cap", C.c_int64), ("order", C.c_void_p), ("order_next", C.c_void_p), ("inv", C.c_void_p),
                ("inv_next", C.c_void_p), ("sim", C.c_void_p), ("member", C.c_void_p), ("dst", C.c_void_p),
                ("keep", C.c_void_p), ("stats", C.c_void_p), ("stats_host", C.c_void_p), ("ws", C.c_void_p),
                ("ws_bytes", C.c_size_t),
                ("seq", C.c_int64), ("order_len", C.c_int64), ("dirty", C.c_int64), ("in_flight", C.c_int64),
                ("swaps", C.c_int64), ("last_L", C.c_int64), ("last_l_out", C.c_int64),
                ("cur_nv", C.c_int64), ("cur_ftn", C.c_int64), ("res_active", C.c_int64), ("res_off", C.c_int64)]


class FFMergeCall(C.Structure):
    """ff_merge_call_t (every member 8 bytes wide: filled with ONE struct.pack_into per call, MERGE_CALL_HEAD)."""
    _fields_ = [("hidden", C.c_void_p), ("addend", C.c_void_p), ("hidden_out", C.c_void_p), ("patch_type", C.c_void_p),
                ("dtype", C.c_int64), ("L", C.c_int64), ("d", C.c_int64), ("L_cap", C.c_int64), ("patch_num", C.c_int64),
                ("order_valid", C.c_int64),
                ("threshold", C.c_double), ("sub", C.c_double), ("ratio_lb", C.c_double),
                ("force_k", C.c_int64), ("fold", C.c_int64),
                ("hint_pre", C.c_int64), ("hint_frames", C.c_int64), ("stream", C.c_void_p), ("n_aux", C.c_int64),
                ("aux", FFAux * MAX_AUX),
                ("mask", C.c_void_p), ("mask_out", C.c_void_p), ("mask_elem_bytes", C.c_int64), ("late_outputs", C.c_int64)]


# head of ff_merge_call_t up to and including n_aux; MAX_AUX aux entries (AUX_ENTRY: 5 x 8 bytes each) and the mask triple follow
MERGE_CALL_HEAD = struct.Struct("=4Q6q3d4qQq")
MERGE_CALL_AUX_OFFSET = MERGE_CALL_HEAD.size
AUX_ENTRY = struct.Struct("=2Q3q")
MERGE_CALL_MASK_OFFSET = MERGE_CALL_AUX_OFFSET + MAX_AUX * AUX_ENTRY.size
MASK_TRIPLE = struct.Struct("=2Qq")
MERGE_CALL_LATE_OFFSET = MERGE_CALL_MASK_OFFSET + MASK_TRIPLE.size


class FFMergeResult(C.Structure):
    _fields_ = [("nv", C.c_int64), ("ftn", C.c_int64), ("count", C.c_int64), ("branch", C.c_int64), ("k", C.c_int64),
                ("l_out", C.c_int64), ("error", C.c_int64), ("unhinted", C.c_int64), ("wait_ns", C.c_int64),
                ("applied", C.c_int64)]


MERGE_RESULT = struct.Struct("=10q")


class FFPruneCall(C.Structure):
    """ff_prune_call_t."""
    _fields_ = [("hidden", C.c_void_p), ("addend", C.c_void_p), ("hidden_out", C.c_void_p), ("attn_w", C.c_void_p),
                ("dtype", C.c_int64), ("S", C.c_int64), ("d", C.c_int64), ("L_cap", C.c_int64), ("w_dtype", C.c_int64),
                ("H", C.c_int64), ("num", C.c_int64), ("tables_ready", C.c_int64),
                ("start", C.c_int64), ("n_img", C.c_int64), ("k", C.c_int64), ("stream", C.c_void_p), ("n_aux", C.c_int64),
                ("aux", FFAux * MAX_AUX),
                ("mask", C.c_void_p), ("mask_out", C.c_void_p), ("mask_elem_bytes", C.c_int64)]


PRUNE_CALL_HEAD = struct.Struct("=4Q11qQq")
PRUNE_CALL_AUX_OFFSET = PRUNE_CALL_HEAD.size
PRUNE_CALL_MASK_OFFSET = PRUNE_CALL_AUX_OFFSET + MAX_AUX * AUX_ENTRY.size


class FFLqArgs(C.Structure):
    """ff_lq_args_t (every member 8 bytes wide: LQ_ARGS.pack_into)."""
    _fields_ = [("q_last", C.c_void_p), ("k", C.c_void_p), ("dtype", C.c_int64), ("H", C.c_int64), ("H_kv", C.c_int64),
                ("num", C.c_int64), ("dh", C.c_int64), ("k_head_stride", C.c_int64), ("k_key_stride", C.c_int64),
                ("scale", C.c_double), ("causal", C.c_int64), ("bias", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


LQ_ARGS = struct.Struct("=2Q7qdq3Q")


class FFSegment(C.Structure):
    _fields_ = [("begin", C.c_int32), ("count", C.c_int32), ("first", C.c_int32), ("period", C.c_int32)]


class FrameFusionHipError(RuntimeError):
    pass


_vp, _i64, _i32, _f64, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/framefusion_hip.h one to one
PROTOTYPES = {
    "ff_abi_version": (C.c_int, []),
    "ff_source_hash": (C.c_char_p, []),
    "ff_error_string": (C.c_char_p, [_i32]),
    "ff_workspace_bytes": (_sz, [_i64, _i64]),
    "ff_build_order": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ff_pair_similarity": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ff_plan_merge": (_i32, [_vp, _i32, _vp, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ff_plan_from_index": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ff_merge_compact": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _vp, _i32, _vp, _vp, C.POINTER(FFAux), _i32,
                                _vp]),
    "ff_gather_tokens_by_index": (_i32, [_vp, _i64, _i64, C.POINTER(FFAux), _i32, _vp]),
    "ff_gather_tokens_by_mask": (_i32, [_vp, _i64, _i64, _vp, _vp, C.POINTER(FFAux), _i32, _vp]),
    "ff_head_mean": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp]),
    "ff_last_query_attention": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f64, _i32, _vp, _vp, _vp, _i64, _i64,
                                       _vp, _sz, _vp, _sz, _vp]),
    "ff_last_query_workspace_bytes": (_sz, [_i32, _i64, _i64, _i64, _i64]),
    "ff_token_span": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "ff_fill_patch_type": (_i32, [_vp, _i64, C.POINTER(FFSegment), _i64, _vp]),
    "ff_patch_type_from_mask": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "ff_ctx_merge_begin": (_i32, [_vp, _vp]),
    "ff_ctx_merge_finish": (_i32, [_vp, _vp, _vp]),
    "ff_ctx_merge_wait": (_i32, [_vp, _vp, _vp]),
    "ff_ctx_merge_apply": (_i32, [_vp, _vp, _vp]),
    "ff_ctx_merge_submit": (_i32, [_vp, _vp]),
    "ff_ctx_merge_collect": (_i32, [_vp, _vp, _vp]),
    "ff_ctx_merge_one_launch": (_i32, [_vp, _vp]),
    "ff_ctx_merge_mail": (_i32, [_vp, _vp]),
    "ff_ctx_prune": (_i32, [_vp, _vp]),
    "ff_ctx_prune_from_qk": (_i32, [_vp, _vp, _vp]),
    "ff_ctx_gather_mask": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "ff_ctx_reset": (_i32, [_vp, _vp]),
    "ff_abi_sizeof": (_sz, [_i32]),
    "ff_host_alloc": (_vp, [_sz]),
    "ff_host_free": (None, [_vp]),
    "ff_ctx_last_query_importance": (_i32, [_vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f64, _i32, _vp, _vp, _i64, _i64,
                                            _i64, _vp, _sz, _vp]),
    "ff_merge_step": (_i32, [_vp, _vp, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp,
                             _vp, _vp, _vp, _vp, _i64, C.POINTER(FFAux), _i32, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def hashed_sources():
    """The files csrc/Makefile hashes into the library (its HASHED list: SRCS + HDRS + the public header + itself), read
    from the Makefile so that the two lists cannot drift apart."""
    import re
    text = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    names = []
    for var in ("SRCS", "HDRS"):
        m = re.search(rf"^{var}\s*=\s*(.*)$", text, re.M)
        if m is None:
            raise FrameFusionHipError(f"csrc/Makefile has no {var} line")
        names += m.group(1).split()
    return tuple(names) + ("../../include/framefusion_hip.h", "Makefile")


def source_hash() -> str:
    """What csrc/Makefile bakes into the library as ff_source_hash(): the first 16 hex digits of the SHA-256 over
    the sources, concatenated in the order `$(sort ...)` gives them.  None when the sources are not there."""
    import hashlib
    h = hashlib.sha256()
    try:
        for name in sorted(hashed_sources()):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
    except (OSError, FrameFusionHipError):
        return None
    return h.hexdigest()[:16]


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile the HIP sources in csrc/ for gfx950 (hipcc cross-compiles without a GPU).  `force`: recompile every
    file (`make -B`) - what __graft_entry__.build() does, so that "it built" means "these sources compile"."""
    if not os.path.isdir(CSRC):
        raise FrameFusionHipError(f"{CSRC} not found")
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))] + (["-B"] if force else [])
    # one build at a time: several ranks of one job may find the same stale binary at the same moment
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode:
        raise FrameFusionHipError("building libframefusion_hip.so failed (see output above)")
    return LIB_PATH


def file_stamp(path: str = None):
    """The source hash baked into the library FILE (the bytes after the "FFSRCHASH:" marker), read without loading it."""
    try:
        with open(path or LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(b"FFSRCHASH:")
    return blob[at + 10:at + 26].decode("ascii", "replace") if at >= 0 else "(a library from before the stamp)"


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # not built yet (fresh checkout): compile the HIP sources once, in-tree; there is no other path
        try:
            build()
        except Exception as e:
            raise FrameFusionHipError(
                f"{LIB_PATH} is missing and could not be built ({e}): run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C {CSRC}`). framefusion_amd has no CPU/eager fallback.") from e
    # a stale binary (sources edited since it was built, or a copy whose timestamps hide that from make) is rebuilt,
    # never run: the stamp is read from the file BEFORE anything is loaded.  Rebuilding needs make + hipcc: a deployment
    # that ships sources next to a binary built from OTHER sources, without a compiler, gets an error that says so
    # (FF_AUTO_BUILD=0 turns the implicit build off altogether).
    want = source_hash()
    if want is not None and file_stamp() != want:
        import shutil
        if os.environ.get("FF_AUTO_BUILD", "1") == "0" or not (shutil.which("make") and shutil.which(os.environ.get("HIPCC", "hipcc"))):
            raise FrameFusionHipError(
                f"{LIB_PATH} was built from other sources (its stamp {file_stamp()}, the sources' {want}) and this process may "
                f"not or cannot rebuild it (FF_AUTO_BUILD=0, or make / hipcc not on PATH): run `make -C {CSRC}` where a ROCm "
                f"toolchain is installed, or ship the library without the csrc/ directory")
        build()
        if file_stamp() != want:
            build(force=True)
        if file_stamp() != want:
            raise FrameFusionHipError(f"{LIB_PATH} was built from other sources (its stamp {file_stamp()}, the sources' "
                                      f"{want}) and rebuilding did not change that")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise FrameFusionHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    got = lib.ff_abi_version()
    if got != ABI_VERSION:
        raise FrameFusionHipError(f"ABI mismatch: library {got}, binding {ABI_VERSION}")
    for which, cls in enumerate((FFCtx, FFMergeCall, FFMergeResult, FFPruneCall, FFAux, FFLqArgs)):
        if lib.ff_abi_sizeof(which) != C.sizeof(cls):
            raise FrameFusionHipError(f"ABI mismatch: sizeof({cls.__name__}) is {C.sizeof(cls)} here, "
                                      f"{lib.ff_abi_sizeof(which)} in the library")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().ff_error_string(rc)
        raise FrameFusionHipError(f"{what} failed: [{rc}] {msg.decode() if msg else '?'}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of PyTorch's current stream on the current device (the raw getters are ~10x cheaper
    than building a torch.cuda.Stream object; same value)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise FrameFusionHipError(
            f"{what}: framefusion_amd runs on MI355X only (got a {t.device.type} tensor); "
            "there is no CPU path in the product - the CPU oracle lives in oracle/ for tests.")
