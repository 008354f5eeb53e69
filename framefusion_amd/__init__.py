"""framefusion_amd - the FrameFusion token-reduction hot path on MI355X (gfx950).

Drop-in for the reference's ``framefusion.main`` / ``framefusion.interface`` surface; all tensor
work runs in hand-written HIP kernels behind the C ABI of ``include/framefusion_hip.h``.
"""
from .main import (FrameFusion, TEXT_TOKEN, IGNORE_TOKEN, cosine_similarity,  # noqa: F401
                   find_contigious_latter_index, call_b_with_residual)
from .interface import (apply_framefusion, get_token_type, replace_framefusion_forward,  # noqa: F401
                        register_family, Family)
from .utils import get_attr_by_name, scaled_dot_product_attention, last_query_importance  # noqa: F401
from ._lib import FrameFusionHipError, build, load  # noqa: F401
from .pair import FrameFusionPair  # noqa: F401
from . import layout, baseline  # noqa: F401

__all__ = ["FrameFusion", "apply_framefusion", "get_token_type", "replace_framefusion_forward",
           "register_family", "Family", "scaled_dot_product_attention", "last_query_importance",
           "get_attr_by_name", "cosine_similarity", "find_contigious_latter_index", "call_b_with_residual",
           "TEXT_TOKEN", "IGNORE_TOKEN", "FrameFusionPair", "FrameFusionHipError", "build", "load", "layout", "baseline"]
