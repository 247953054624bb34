"""Test-infrastructure stand-in for chamferdist==1.0.0 `knn_points` (requirements.txt:2 of the
reference; source NOT under /root/reference). Semantics restated from pytorch3d's knn_points,
from which chamferdist 1.0.0 derives: K=1, squared-L2 `dists`, int64 `idx`, lowest index on
ties. Arithmetic is the oracle's (oracle/gs_oracle.c: gs_or_knn1): d = fma(dz,dz,fma(dy,dy,dx*dx)).
PARITY UNPINNED for tie-breaking and for the dist_thresh unit."""
from collections import namedtuple

import torch

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1,
               return_nn=False, return_sorted=True):
    assert K == 1
    from oracle import oracle as _o  # C brute force (OpenMP), same arithmetic as the HIP kernel
    dists, idxs = [], []
    with torch.no_grad():
        for b in range(p1.shape[0]):
            i, d = _o.knn1(p1[b].detach().contiguous().numpy(), p2[b].detach().contiguous().numpy())
            idxs.append(torch.from_numpy(i))
            dists.append(torch.from_numpy(d))
    return _KNN(torch.stack(dists).unsqueeze(-1), torch.stack(idxs).unsqueeze(-1), None)
