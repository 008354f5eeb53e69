"""What does torch.topk ON THE MI355X do with ties at the k-th value?

The reference selects with torch.topk twice (framefusion/main.py:122-124 on the [1, Nv] similarity
vector, main.py:75 on the 1-D importance slice) and only the SET matters (both are sorted
afterwards).  bf16 similarities take a dozen distinct values, so the cut always falls inside a tie
class; the build takes ties in ascending index order (oracle.topk_lowest_index).  This probe runs
torch.topk on the GPU on such vectors and reports, per case, whether the selected set equals the
lowest-index rule, and if not, which tie members it took.  Output: one line per case + a summary;
tests/test_gpu_topk_ties.py asserts the same thing.
"""
import json
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ff_oracle as orc  # noqa: E402  (probe = test infrastructure)


def cases():
    g = torch.Generator().manual_seed(11)
    out = []
    # C2-like similarity vectors: few distinct bf16 values, -2 at chain starts
    for n, dtype in ((36864, torch.bfloat16), (36864, torch.float16), (13440, torch.bfloat16),
                     (73728, torch.bfloat16), (4608, torch.float32), (1000, torch.bfloat16), (300000, torch.bfloat16)):
        base = 0.8 + 0.02 * torch.randn(n, generator=g)
        x = base.to(dtype)
        if dtype == torch.float32:
            x = (x * 64).round() / 64          # ties in fp32 too
        x[::64] = -2.0
        out.append((f"sim_{n}_{str(dtype).split('.')[-1]}", x))
    # importance-like vectors: tiny positive bf16 values with many ties, NaNs
    imp = (torch.rand(18000, generator=g) * 1e-4).to(torch.bfloat16)
    out.append(("imp_18000_bf16", imp))
    xn = (0.5 + 0.01 * torch.randn(5000, generator=g)).to(torch.bfloat16)
    xn[torch.randint(0, 5000, (7,), generator=g)] = float("nan")
    out.append(("nan_5000_bf16", xn))
    out.append(("allequal_9000_bf16", torch.full((9000,), 0.75, dtype=torch.bfloat16)))
    return out


def main():  # pragma: no cover
    dev = "cuda"
    report, bad = [], 0
    for name, x in cases():
        n = x.numel()
        for k in sorted({1, 2, n // 7, int(0.3 * n), int(0.7 * n), n - 1, n}):
            want = orc.topk_lowest_index(x, k)
            for shape in ("1d", "2d"):
                xd = x.to(dev) if shape == "1d" else x.to(dev)[None]
                idx = torch.topk(xd, k).indices
                idx = idx if shape == "1d" else idx[0]
                got = torch.sort(idx).values.cpu()
                same = torch.equal(got, want)
                rec = dict(case=name, k=k, shape=shape, same_set=bool(same))
                if not same:
                    bad += 1
                    xs = x.float()
                    kth = torch.sort(xs.nan_to_num(nan=float("inf")), descending=True).values[k - 1]
                    tie = torch.nonzero(xs.nan_to_num(nan=float("inf")) == kth).reshape(-1)
                    tg = got[torch.isin(got, tie)]
                    tw = want[torch.isin(want, tie)]
                    rec.update(kth=float(kth), tie_size=int(tie.numel()), taken=int(tg.numel()),
                               got_first=tg[:8].tolist(), want_first=tw[:8].tolist(),
                               non_tie_equal=bool(torch.equal(got[~torch.isin(got, tie)], want[~torch.isin(want, tie)])))
                report.append(rec)
    for r in report:
        if not r["same_set"]:
            print(json.dumps(r))
    print(json.dumps(dict(summary=True, cases=len(report), mismatching=bad, torch=torch.__version__,
                          device=torch.cuda.get_device_name(0))))


if __name__ == "__main__":
    main()
