"""knn_module.KNN(k): brute-force nearest neighbours behind the reference's calling convention
(utils/knn_search.py:138-172):

    d, idx = knn_module.KNN(k)(target_F [1,f,n], source_F [1,f,m])
      k == 1 : d [1,1,m] float32, idx [1,1,m] int64   (CPU tensors, like the reference's .cpu() results)
      k >= 2 : idx [1,k,m] int64 ordered by increasing distance; d is returned as [1,k,1,m] exactly like the
               reference's deprecated 3-D `.T` produces (no caller reads it)

and the class's other public methods with the reference's signatures, shapes and defaults: pdist (:17-24), find_nn_gpu (:26-66),
find_knn_gpu (:68-103), find_corr (:105-136).  dist_type 'L2' = sqrt(sum (a-b)^2 + 1e-7), 'SquareL2' = sum (a-b)^2; anything else raises
NotImplementedError('Not implemented') like the reference.  `nn_max_n` (the reference's chunk length against its N x M temporary) is
accepted and ignored: the kernels never materialise the matrix.  The distance formula / tie-break contract is the kernels'
(roreg_nn_search_ex, roreg_knn_search_ex, roreg_pdist): first minimum wins, k lists in increasing distance.

Any feature width and k <= 32 are served (the reference accepts anything): the tuned kernels are built for the two widths the pipeline
uses -- 3 (keypoint coordinates) and 32 (FCGF / invariant descriptors) -- with k <= 8; other widths and longer lists go through a plain
one-thread-per-source kernel with the same distance formula and tie-break (csrc/nn_search.hip, knn_generic_kernel).  k > 32 raises
NotImplementedError before a kernel is launched.  A one-row input keeps its row axis (the reference's `.squeeze()` would drop it)."""
import numpy as np
import torch

from .. import hip


def _squared(dist_type):
    if dist_type == 'L2':
        return False
    if dist_type == 'SquareL2':
        return True
    raise NotImplementedError('Not implemented')


TUNED_WIDTHS, MAX_K = (3, 32), 32


def _check_shape(f, k=1):
    if f < 1:
        raise ValueError(f'knn_module: feature width {f}')
    if k > MAX_K:
        raise NotImplementedError(f'knn_module: k = {k} is not supported by the HIP search kernels (k <= {MAX_K})')


def _rows(F, k=1):
    """the reference's `.squeeze()`d [rows, f] matrix, on the device in float32 (a single row keeps its row axis)"""
    F = F if torch.is_tensor(F) else torch.as_tensor(np.asarray(F))
    if F.dim() > 2:
        F = F.reshape(-1, F.shape[-1]) if F.numel() == F.shape[-2] * F.shape[-1] else F.squeeze()
    if F.dim() == 1:
        F = F[None]
    _check_shape(int(F.shape[-1]), k)
    return F.to('cuda', torch.float32).contiguous()


class modified_knn_matcher():
    def __init__(self, k=1) -> None:
        self.k = k

    def pdist(self, A, B, dist_type='L2'):
        """[m,f], [n,f] -> [m,n] on the inputs' device (utils/knn_search.py:17-24)."""
        sq = _squared(dist_type)
        A = A if torch.is_tensor(A) else torch.as_tensor(np.asarray(A))
        out = hip.pdist(A.to('cuda', torch.float32).contiguous(), (B if torch.is_tensor(B) else torch.as_tensor(np.asarray(B))).to('cuda', torch.float32).contiguous(), squared=sq)
        return out.to(A.device)

    def find_nn_gpu(self, source_F, target_F, nn_max_n=1000, return_distance=True, dist_type='SquareL2'):
        """-> (dists [m] f32, inds [m] i64) CPU tensors, or inds alone (utils/knn_search.py:26-66)."""
        sq = _squared(dist_type)
        idx, d = hip.nn_search(_rows(source_F), _rows(target_F), want_dist=True, squared=sq)
        dists, inds = d.cpu().squeeze(), idx.cpu().squeeze()
        return (dists, inds) if return_distance else inds

    def find_knn_gpu(self, source_F, target_F, nn_max_n=1000, return_distance=True, dist_type='SquareL2'):
        """-> (dists [m,1,k] f32, inds [m,k] i64) CPU tensors, or inds alone (utils/knn_search.py:68-103)."""
        sq = _squared(dist_type)
        idx, d = hip.knn_search(_rows(source_F, self.k), _rows(target_F, self.k), self.k, want_dist=True, squared=sq)
        dists, inds = d.cpu()[:, None, :], idx.cpu()
        return (dists, inds) if return_distance else inds

    def find_corr(self, F0, F1, subsample_size=-1, mutual=True, nn_max_n=500):
        """(inds0, inds1) of the (mutual) nearest neighbours in 'SquareL2', with the reference's np.random.choice subsampling
        (utils/knn_search.py:105-136)."""
        inds0, inds1 = np.arange(F0.shape[0]), np.arange(F1.shape[0])
        if subsample_size > 0:
            N0 = min(len(F0), subsample_size)
            N1 = min(len(F1), subsample_size)
            inds0 = np.random.choice(len(F0), N0, replace=False)
            inds1 = np.random.choice(len(F1), N1, replace=False)
            F0 = F0[inds0]
            F1 = F1[inds1]
        a, b = _rows(F0), _rows(F1)
        nn01 = hip.nn_search(a, b, squared=True)
        if not mutual:
            return inds0, inds1[nn01.cpu().numpy()]
        nn10 = hip.nn_search(b, a, squared=True)
        m, cnt = hip.mutual_matches(nn01, nn10)
        matches = m[:int(cnt.item())].cpu().numpy().astype(np.int32)
        return inds0[matches[:, 0]], inds1[matches[:, 1]]

    def __call__(self, target_F, source_F, nn_max_n=500, dist_type='L2'):
        sq = _squared(dist_type)
        _check_shape(int(target_F.shape[-2]), self.k)
        tgt = target_F.reshape(target_F.shape[-2], target_F.shape[-1]).t().to('cuda', torch.float32).contiguous()   # [n,f]
        src = source_F.reshape(source_F.shape[-2], source_F.shape[-1]).t().to('cuda', torch.float32).contiguous()   # [m,f]
        if self.k < 2:
            idx, d = hip.nn_search(src, tgt, want_dist=True, squared=sq)
            return d.cpu()[None, None], idx.cpu()[None, None]
        idx, d = hip.knn_search(src, tgt, self.k, want_dist=True, squared=sq)       # [m,k]
        return d.cpu().t()[:, None, :][None], idx.cpu().t()[None]


class knn_module_class():
    def KNN(self, k):
        return modified_knn_matcher(k)


knn_module = knn_module_class()
