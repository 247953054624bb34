"""Ragged <-> padded conversions of per-sequence tables (structures/structutils.py:47-124 of the reference, which took
them from pytorch3d): same names, arguments, results and error messages.  Pure data movement on whatever device the
tensors live on (the surfel store itself keeps capacity-backed buffers and zero-copy views, structures/pointclouds.py;
these helpers are the stand-alone form the reference exports)."""
import torch

__all__ = ["list_to_padded", "padded_to_list"]


def list_to_padded(x, pad_size=None, pad_value=0.0, equisized=False):
    """list of B tensors (N_b, C_b) -> (B, pad_size[0], pad_size[1]) (default: the largest N_b and C_b among the
    non-empty items), filled with pad_value outside the items.  equisized=True: the items are known to have one shape
    and are simply stacked."""
    if equisized:
        return torch.stack(x, 0)
    if pad_size is not None:
        if len(pad_size) != 2:
            raise ValueError("Pad size must contain target size for 1st and 2nd dim")
        rows, cols = pad_size
    else:
        filled = [t for t in x if len(t) > 0]
        rows, cols = max(t.shape[0] for t in filled), max(t.shape[1] for t in filled)
    out = x[0].new_full((len(x), rows, cols), pad_value)
    for b, t in enumerate(x):
        if len(t) == 0:
            continue
        if t.ndim != 2:
            raise ValueError("Supports only 2-dimensional tensor items")
        out[b, : t.shape[0], : t.shape[1]] = t
    return out


def padded_to_list(x, split_size=None):
    """(B, N, C) -> list of B tensors; split_size[b] = N_b or (N_b, C_b) cuts item b down to its own size (views of x)."""
    if x.ndim != 3:
        raise ValueError("Supports only 3-dimensional input tensors")
    items = list(x.unbind(0))
    if split_size is None:
        return items
    if len(split_size) != x.shape[0]:
        raise ValueError("Split size must be of same length as inputs first dimension")
    for b, sz in enumerate(split_size):
        if isinstance(sz, int):
            items[b] = items[b][:sz]
        elif len(sz) == 2:
            items[b] = items[b][: sz[0], : sz[1]]
        else:
            raise ValueError("Support only for 2-dimensional unbinded tensor. " + " " * 20 +
                             "Split size for more dimensions provided")
    return items
