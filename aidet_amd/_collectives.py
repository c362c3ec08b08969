"""The three collectives of the tiled merge (dota_merge.py) behind one seam.

Production transport: RCCL (backend "nccl" on ROCm) on device tensors, one rank per GPU over xGMI -- the calls go
straight to ``torch.distributed``.  What replaces ``collect_results_gpu``'s pickle gathers (mmdet/apis/test.py:116-146).

Debug transport: a ``gloo`` group whose ranks hold DEVICE tensors -- several ranks may then share ONE GPU, which RCCL
refuses ("duplicate GPU").  gloo has no device all_to_all, so each collective is staged through host memory here: the
kernels on either side of the exchange are the real HIP ones, only the wire differs.  That is how the multi-rank
product path is exercised on a one-GPU box (tests/test_multiproc_gpu.py, ``bench.py --transport gloo``); it is never
the default and never timed as the product.
"""
import torch
import torch.distributed as dist


def _staged(t, group):
    """device tensor on a gloo group -> the collective runs on host copies"""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_gather_into_tensor(out, inp, group=None):
    if _staged(inp, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, inp.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)
    return out


def all_to_all_single(out, inp, out_splits, in_splits, group=None):
    if _staged(inp, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(h)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
    return out


def all_reduce(t, op=None, group=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t
