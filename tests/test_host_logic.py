"""CPU: host-side logic of the product package that needs no GPU - the budget arithmetic, the
attach protocol of replace_framefusion_forward / the family registry, API helpers, and the fact
that the product refuses CPU tensors instead of falling back."""
from types import MethodType

import pytest
import torch
from torch import nn

import framefusion_amd as ffa
from framefusion_amd.synth import video_tokens, rotary_tables
from oracle import ff_oracle as orc


def test_budget_is_bit_identical_to_the_oracle(golden):
    g = golden("primitives")
    for lst, cost, want in zip(g["budget_lists"], g["budget_costs"], g["budget_vals"]):
        sl = [float(x) for x in str(lst).split(",") if x]
        got = ffa.FrameFusion._compute_pruning_ratio(sl, float(cost))
        assert float(got) == float(want) == float(orc.budget(sl, float(cost)))
    assert ffa.FrameFusion._compute_pruning_ratio([], 1.0) == 0
    with pytest.raises(ValueError, match="The cost is too small"):
        ffa.FrameFusion._compute_pruning_ratio([0] * 10, 0.3)


def test_find_contigious_latter_index_kat(golden):
    kat = torch.tensor([[0, 1, 1, 1, 0, 0, 1, 1]])
    assert ffa.find_contigious_latter_index(kat).tolist() == [[0, 0, 0, 3, 0, 0, 0, 2]]   # main.py:361-363
    g = golden("primitives")
    x = torch.from_numpy(g["runs_in"])
    assert torch.equal(ffa.find_contigious_latter_index(x), torch.from_numpy(g["runs_out"]))
    assert torch.equal(ffa.find_contigious_latter_index(x.to(torch.bfloat16)).long(), torch.from_numpy(g["runs_out"]))


def test_constants_and_surface():
    assert (ffa.TEXT_TOKEN, ffa.IGNORE_TOKEN) == (-1, -2)
    f = ffa.FrameFusion()
    assert (f.cost, f.similarity_lower_bound, f.ratio_lower_bound) == (0.3, 0.6, 0.1)
    assert list(f.parameters()) == []
    f.prepare(torch.tensor([[-1, 0, 1]]), 2, 1, 3, 2, 3, sparsity_list=[0.2])
    assert f.sparsity_list == [0.2] and not f.finish_merging and not f.finish_pruning
    for name in ("patch_type", "patch_num", "image_token_start_index", "image_token_end_index",
                 "image_token_length", "original_length"):
        assert hasattr(f, name)
    for name in ("compute_similarity_and_token_index_by_patch", "merge_tokens_and_get_mask", "_compute_pruning_ratio",
                 "position_embedding_handler_at_pruning", "position_embedding_handler_at_merging"):     # main.py:142,161
        assert callable(getattr(ffa.FrameFusion, name))
    # the handlers' container checks come before any device work (main.py:155-157, 175-177)
    with pytest.raises(NotImplementedError, match="Only support 2D"):
        f.position_embedding_handler_at_pruning(torch.zeros(1, 3, 2), torch.tensor([0]))
    with pytest.raises(NotImplementedError, match="list or tensor"):
        f.position_embedding_handler_at_merging((torch.zeros(1, 3, 2),) * 2, torch.ones(1, 3, dtype=torch.bool))
    with pytest.raises(AssertionError):
        f.position_embedding_handler_at_merging([torch.zeros(1, 3, 2)], torch.ones(1, 3, dtype=torch.bool))


def test_no_cpu_fallback():
    h, pt = video_tokens(4, 6, 32, pre=1, post=1)
    f = ffa.FrameFusion()
    f.prepare(pt, 6, 1, 25, 24, h.shape[1])
    with pytest.raises(ffa.FrameFusionHipError, match="MI355X only"):
        f(h, rotary_tables(h.shape[1], 8), None)
    with pytest.raises(ffa.FrameFusionHipError):
        ffa.FrameFusion.compute_similarity_and_token_index_by_patch(h, pt, 6)
    with pytest.raises(ffa.FrameFusionHipError):
        ffa.cosine_similarity(h[0], h[0])
    with pytest.raises(ffa.FrameFusionHipError):
        ffa.scaled_dot_product_attention(torch.zeros(1, 2, 4, 8), torch.zeros(1, 2, 4, 8), None)
    # decode steps and un-prepared instances behave like the reference without touching the device
    tok = torch.zeros(1, 1, 32)
    assert f(tok, "pos", "mask")[0] is tok
    with pytest.raises(AttributeError):
        ffa.FrameFusion()(h, None, None)


def test_get_attr_by_name():
    class Box:
        pass
    root = Box()
    root.llm = Box()
    root.llm.layers = [Box(), Box()]
    root.llm.layers[1].self_attn = "attn1"
    assert ffa.get_attr_by_name(root, "llm.layers.1.self_attn") == "attn1"


class _Attn(nn.Module):
    def forward(self, x):
        return x


class _Layer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = _Attn()

    def forward(self, x):
        return x


class _LLM(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(), _Layer(), _Layer()])


class _Wrapper(nn.Module):
    def __init__(self):
        super().__init__()
        self.model = _LLM()


def test_attach_protocol():
    """interface.py:169-214: ONE shared FrameFusion on the wrapper, the llm, every layer and every
    attention module; forwards re-bound with MethodType."""
    def llm_fwd(self, x): return ("llm", x)
    def dec_fwd(self, x): return ("dec", x)
    def att_fwd(self, x): return ("att", x)
    m = _Wrapper()
    ff = ffa.replace_framefusion_forward(m, 0.25, 0.7, 0.05, llm_fwd, dec_fwd, att_fwd)
    assert isinstance(ff, ffa.FrameFusion) and (ff.cost, ff.similarity_lower_bound, ff.ratio_lower_bound) == (0.25, 0.7, 0.05)
    assert m.framefusion is ff and m.model.framefusion is ff
    assert isinstance(m.model.forward, MethodType) and m.model.forward(1) == ("llm", 1)
    for layer in m.model.layers:
        assert layer.framefusion is ff and layer.self_attn.framefusion is ff
        assert layer.forward(2) == ("dec", 2) and layer.self_attn.forward(3) == ("att", 3)


def test_accelerate_hook_is_preserved():
    from accelerate.hooks import ModelHook, add_hook_to_module

    class Tag(ModelHook):
        def post_forward(self, module, output):
            return ("hooked", output)

    def dec_fwd(self, x): return ("dec", x)
    m = _Wrapper()
    add_hook_to_module(m.model.layers[0], Tag())
    ffa.replace_framefusion_forward(m, 0.3, 0.6, 0.1, lambda s, x: x, dec_fwd, lambda s, x: x)
    assert m.model.layers[0].forward(5) == ("hooked", ("dec", 5))       # interface.py:204-207
    assert m.model.layers[1].forward(5) == ("dec", 5)


def test_family_registry():
    m = _Wrapper()
    with pytest.raises(NotImplementedError):
        ffa.apply_framefusion(m, 0.3, 0.6, 0.1)
    calls = []

    def prepare_hook(self, *a):
        calls.append(a)
        return "prepared"
    fam = ffa.Family("toy", lambda mod: isinstance(mod, _Wrapper), lambda s, x: ("llm", x), lambda s, x: ("dec", x),
                     lambda s, x: ("att", x), prepare_hook=("build_inputs", prepare_hook))
    ffa.register_family(fam)
    try:
        ffa.apply_framefusion(m, 0.3, 0.6, 0.1)
        assert m.model.layers[2].forward(1) == ("dec", 1)
        assert m.build_inputs(7) == "prepared" and calls == [(7,)]
        m2 = _Wrapper()
        ffa.get_token_type(m2)
        assert m2.build_inputs(1) == "prepared" and not hasattr(m2, "framefusion")
    finally:
        from framefusion_amd import interface
        interface._FAMILIES[:] = [f for f in interface._FAMILIES if f.name != "toy"]


def test_synthetic_generator_is_deterministic_and_on_grid():
    a, pa = video_tokens(5, 7, 64, seed=3, pre=2, post=1, grid=0.125)
    b, pb = video_tokens(5, 7, 64, seed=3, pre=2, post=1, grid=0.125)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16)) and torch.equal(pa, pb)
    assert pa.tolist()[0][:3] == [-1, -1, 0] and pa.shape == (1, 2 + 35 + 1)
    x = a.float() * 8
    assert torch.equal(x, x.round()) and float(a.float().abs().max()) <= 4.0


def test_patch_type_assignment_invalidates_the_cached_order():
    """The by-patch order kept in the scratch is keyed on a generation counter that ANY assignment of
    patch_type bumps (prepare(), the compaction of a merge call, a caller writing the attribute)."""
    ff = ffa.FrameFusion()
    g0 = ff._ptype_gen
    ff.prepare(torch.zeros(1, 4, dtype=torch.long), 2, 0, 3, 4, 4)
    g1 = ff._ptype_gen
    ff.patch_type = ff.patch_type                         # even the same tensor object: a new generation
    assert g0 < g1 < ff._ptype_gen
    b = ffa.baseline.FixedSparsityMerging([0.1])
    h0 = b._ptype_gen
    b.prepare(torch.zeros(1, 4, dtype=torch.long), 2)
    assert b._ptype_gen > h0 and b.patch_type is not None


def test_copies_do_not_share_the_device_scratch():
    """copy.deepcopy / pickle of a used instance must not alias the pinned result block the device publishes into."""
    import copy
    import pickle
    ff = ffa.FrameFusion(0.25, 0.7, 0.2)
    ff.prepare(torch.zeros(1, 4, dtype=torch.long), 2, 0, 3, 4, 4)
    ff._scratch[("cuda", 0)] = object()                   # stands for a live _Scratch
    ff.last_call = dict(scratch=ff._scratch[("cuda", 0)])
    for clone in (copy.deepcopy(ff), pickle.loads(pickle.dumps(ff))):
        assert clone._scratch == {} and clone.last_call is None
        assert (clone.cost, clone.similarity_lower_bound, clone.ratio_lower_bound) == (0.25, 0.7, 0.2)
        assert clone.finish_merging is False and clone.sparsity_list == [] and torch.equal(clone.patch_type, ff.patch_type)
    assert ff._scratch and ff.last_call is not None       # the original keeps its own


def test_forward_residual_signature_and_errors():
    ff = ffa.FrameFusion()
    ff.prepare(torch.zeros(1, 4, dtype=torch.long), 2, 0, 3, 4, 4, finish_merging=True, finish_pruning=True)
    a, b = torch.ones(1, 4, 8), torch.full((1, 4, 8), 2.0)
    out, pe, m = ff.forward_residual(a, b, "pos", None)          # nothing due: a plain add, on any device
    assert torch.equal(out, a + b) and pe == "pos" and m is None
    with pytest.raises(ffa.FrameFusionHipError):
        ff.forward(b, "pos", None, residual=torch.ones(1, 3, 8))


def test_bf16_divide_equals_multiply_by_reciprocal_exhaustively():
    """What the merge kernel's bf16 flush relies on (csrc/ff_merge.hip): for EVERY bf16 value a and every run-length
    divisor T(k), k = 2..70 000, T(a / d) == T(a * RN(1 / d)) with fp32 arithmetic in between - the quotient of two 8-bit
    significands is never within 2^-17 of a bf16 rounding boundary.  fp16 (11-bit significands) fails the same check,
    which is why only the bf16 instantiation uses the reciprocal."""
    torch.set_flush_denormal(False)
    one = torch.tensor(1.0, dtype=torch.float32)

    def mismatches(dtype, int_view):
        a = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dtype).to(torch.float32)
        ds = torch.unique(torch.arange(2, 70001, dtype=torch.float32).to(dtype).to(torch.float32))
        bad = 0
        for d in ds:
            ref = (a / d).to(dtype)
            got = (a * (one / d)).to(dtype)
            same = (ref.view(int_view) == got.view(int_view)) | (torch.isnan(ref.float()) & torch.isnan(got.float()))
            bad += int((~same).sum())
        return bad, len(ds)
    bad, n = mismatches(torch.bfloat16, torch.int16)
    assert n == 1288 and bad == 0
    bad16, _ = mismatches(torch.float16, torch.int16)
    assert bad16 > 0                                   # the reason fp16 keeps the IEEE division


def test_outer_stride_of_token_dense_views():
    """main._outer_stride / _token_dense: which position tensors the merge kernel reads in place (ff_aux_t.src_outer_bytes)
    and which need a copy - no GPU involved."""
    from framefusion_amd.main import _outer_stride, _token_dense
    full = torch.zeros(3, 1, 10, 4)
    assert _outer_stride(full) == 40 and _token_dense(full) is full
    view = full.narrow(2, 0, 6)                                  # what a merge call returns for an M-RoPE table
    assert not view.is_contiguous() and _outer_stride(view) == 40
    wrapped = _token_dense(view)            # read in place: the tensor itself + its outer stride in bytes, found in THIS call
    assert wrapped.t is view and wrapped.outer_bytes == 40 * 4 and not hasattr(view, "_ff_outer_bytes")
    flat = torch.zeros(1, 10, 4).narrow(1, 0, 6)                 # [1, L_out, dh] of a [1, L_cap, dh] buffer: contiguous as it is
    assert flat.is_contiguous() and _outer_stride(flat) == 24
    two = torch.zeros(2, 3, 10, 4)                               # two leading dims, one uniform stride: collapsible
    assert _outer_stride(two) == 40 and _outer_stride(two.narrow(2, 2, 5)) == 40
    assert _outer_stride(two[:, ::2]) is None                    # leading dims 120 and 80 apart: not one stride
    assert _outer_stride(full.transpose(2, 3)) is None           # rows not dense
    assert _outer_stride(full[..., ::2]) is None
    gathered = _token_dense(full[..., ::2])
    assert gathered.is_contiguous() and gathered.shape == (3, 1, 10, 2)
    # a broadcast leading dim (cos.expand(3, 1, L, dh): strides (0, L*dh, dh, 1)) has ONE plane in memory: reading three planes
    # "in place" would run past its storage - it must be copied (the round-4 code passed it through with outer stride 0 = dense)
    one = torch.arange(40.).reshape(1, 1, 10, 4)
    exp = one.expand(3, 1, 10, 4)
    assert exp.stride(0) == 0 and _outer_stride(exp) is None
    dense = _token_dense(exp)
    assert dense is not exp and dense.is_contiguous() and torch.equal(dense, exp)
    lap = torch.zeros(100).as_strided((3, 1, 10, 4), (20, 40, 4, 1))          # overlapping planes: copied as well
    assert _outer_stride(lap) is None and _token_dense(lap).is_contiguous()


def test_bench_charges_every_call_kind_its_compulsory_bytes():
    """bench.call_bytes: what extra.configs[*].hbm_frac divides by.  A prune moves only the rows it keeps (rounds 1-4 charged it
    S + L_out rows, which put the 72B prune gather above the 8 TB/s peak); an identity merge writes nothing."""
    import bench
    d, elt, dh = 8192, 2, 128
    row, pe = d * elt, 2 * dh * elt
    S, L_out = 35072, 10205
    prune = bench.call_bytes("prune", S, L_out, S, d, elt, dh, kv_heads=8)
    assert prune == 8 * S * dh * elt + 2 * L_out * (row + pe) + 5 * S
    assert prune < bench.merge_bytes(S, L_out, d, elt, dh)                       # the old charge
    assert prune / 63.3e-6 < 8.0e12                                              # the measured 63.3 us stay below the peak
    assert bench.call_bytes("merge", 6404, 6404, 6377, 3584, elt, dh, pe_outer=3) == 6377 * 3584 * elt       # identity: K1's read only
    L, Lo = 36864, 11060
    assert bench.call_bytes("merge", L, Lo, L, 4096, elt, dh) == bench.merge_bytes(L, Lo, 4096, elt, dh) == 417513888
