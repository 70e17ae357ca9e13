"""knn_module.KNN(k): brute-force nearest neighbours behind the reference's calling convention
(utils/knn_search.py:138-172):

    d, idx = knn_module.KNN(k)(target_F [1,f,n], source_F [1,f,m])
      k == 1 : d [1,1,m] float32, idx [1,1,m] int64   (CPU tensors, like the reference's .cpu() results)
      k >= 2 : idx [1,k,m] int64 ordered by increasing distance; d is returned as [1,k,1,m] exactly like the
               reference's deprecated 3-D `.T` produces (no caller reads it)

The distance formula / tie-break contract is the kernel's (roreg_nn_search, roreg_knn_search)."""
import torch

from .. import hip


class modified_knn_matcher():
    def __init__(self, k=1) -> None:
        self.k = k

    def __call__(self, target_F, source_F, nn_max_n=500, dist_type='L2'):
        if dist_type != 'L2':
            raise NotImplementedError('Not implemented')
        tgt = target_F.reshape(target_F.shape[-2], target_F.shape[-1]).t().to('cuda', torch.float32).contiguous()   # [n,f]
        src = source_F.reshape(source_F.shape[-2], source_F.shape[-1]).t().to('cuda', torch.float32).contiguous()   # [m,f]
        if self.k < 2:
            idx, d = hip.nn_search(src, tgt, want_dist=True)
            return d.cpu()[None, None], idx.cpu()[None, None]
        idx = hip.knn_search(src, tgt, self.k)                       # [m,k]
        d = torch.sqrt(((src[:, None, :] - tgt[idx]) ** 2).sum(-1) + 1e-7)     # [m,k] (unused by callers)
        return d.cpu().t()[:, None, :][None], idx.cpu().t()[None]


class knn_module_class():
    def KNN(self, k):
        return modified_knn_matcher(k)


knn_module = knn_module_class()
