"""Test-only stand-in for the third-party `torch_scatter` package (pytorch-scatter 2.0.8,
pinned in the reference's env_gnn.yml:98), which is not installed in this image.

It exists ONLY so that tests/golden/make_golden.py can import the reference's
models/mpn.py unmodified (that file does `from torch_scatter import ...` at import time,
models/mpn.py:4).  It restates the published semantics of the three functions at the call
forms the reference uses (models/mpn.py:196,199,202):

  scatter_add (src, index, dim=0, dim_size=N) -> out[N, C], out[index[e]] += src[e]
  scatter_mean(src, index, dim=0, dim_size=N) -> sum / max(count, 1)
  scatter_max (src, index, dim=0, dim_size=N) -> (values, argmax); rows nobody writes stay 0

The semantics are pinned by tests/test_oracle.py::test_scatter_known_answers.
Never imported by the product package.
"""
import torch


def _expand(index, src, dim):
    if dim < 0:
        dim += src.dim()
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape).expand_as(src)
    return index, dim


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    index, dim = _expand(index, src, dim)
    if out is None:
        size = list(src.shape)
        size[dim] = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


scatter_sum = scatter_add


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    total = scatter_add(src, index, dim, out, dim_size)
    idx1d = index if index.dim() == 1 else index.select(1 - (dim % src.dim()), 0)
    count = torch.zeros(total.size(dim % src.dim()), dtype=src.dtype, device=src.device)
    count.scatter_add_(0, idx1d, torch.ones_like(idx1d, dtype=src.dtype))
    count.clamp_(min=1)
    shape = [1] * total.dim()
    shape[dim % src.dim()] = -1
    return total.div_(count.view(shape))


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    index_e, dim = _expand(index, src, dim)
    size = list(src.shape)
    size[dim] = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    vals = torch.zeros(size, dtype=src.dtype, device=src.device)
    vals = vals.scatter_reduce(dim, index_e, src, reduce="amax", include_self=False)
    # argmax: first source position attaining the max; dim_size of src for untouched rows
    n_src = src.size(dim)
    pos = torch.arange(n_src, device=src.device)
    shape = [1] * src.dim()
    shape[dim] = -1
    pos = pos.view(shape).expand_as(src)
    hit = src == vals.gather(dim, index_e)
    cand = torch.where(hit, pos, torch.full_like(pos, n_src))
    arg = torch.full(size, n_src, dtype=torch.long, device=src.device)
    arg = arg.scatter_reduce(dim, index_e, cand, reduce="amin", include_self=True)
    return vals, arg
