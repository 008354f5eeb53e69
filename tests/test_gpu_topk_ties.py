"""What the reference REALLY executes at its two top-k sites on this hardware: torch.topk on the MI355X
(framefusion/main.py:122-124 on the [1, Nv] similarities, main.py:75 on the importance slice).  Only the
selected SET matters (both sites sort the indices afterwards).  bf16 similarities take a dozen distinct
values, so the cut of the headline workload always falls inside a tie class; the build takes ties in
ascending index order (oracle.topk_lowest_index, ff_plan.hip k_flags).  This test pins that rule to
torch.topk's behaviour on the GPU: dense ties, NaNs, 1-D and [1, N] inputs, k from 1 to N.
tests/probe_topk_ties.py prints the per-case outcome (profiles/r02_topk_ties.txt)."""
import pytest
import torch

from oracle import ff_oracle as orc
from tests.probe_topk_ties import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", [c[0] for c in cases()])
def test_rocm_topk_takes_ties_in_index_order(case):
    x = dict(cases())[case]
    n = x.numel()
    for k in sorted({1, 2, n // 7, int(0.3 * n), int(0.7 * n), n - 1, n}):
        want = orc.topk_lowest_index(x, k)
        for xd in (x.to(DEV), x.to(DEV)[None]):
            idx = torch.topk(xd, k).indices.reshape(-1)
            assert torch.equal(torch.sort(idx).values.cpu(), want), (case, k, tuple(xd.shape))
