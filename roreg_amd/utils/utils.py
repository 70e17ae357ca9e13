"""Host helpers of the stage classes, written from their contracts (the reference keeps its versions in utils/utils.py):

  make_non_exists_dir(path)        create the directory unless it is there                        (utils/utils.py:9-11)
  transform_points(pts, T)         rigid / homogeneous map of an [n,3] point array by a 3x3, 3x4   (utils/utils.py:38-46)
                                   or 4x4 matrix; any other shape is NotImplementedError
  to_cuda(batch)                   a list or dict of tensors (or of lists of tensors) moved to the (utils/utils.py:104-132)
                                   current device, same container type; anything else is NotImplementedError
  load_checkpoint(path)            the reference's .pth dicts, loaded on the host

The float64 arithmetic of transform_points is kept operation for operation (one BLAS product per call, the translation added
afterwards; the 4x4 case multiplies homogeneous rows and divides by the last column), because the estimator's goldens compare its
results to 1e-9 and the inlier decisions built on them bit for bit."""
import os

import numpy as np
import torch


def make_non_exists_dir(fn):
    os.makedirs(fn, exist_ok=True)


def _linear(pts, T):
    return pts @ T.T


def _affine(pts, T):
    return pts @ T[:, :3].T + T[:, 3:].T


def _projective(pts, T):
    rows = np.concatenate([pts, np.ones([pts.shape[0], 1])], 1) @ T.T
    return rows[:, :-1] / rows[:, -1:]


_TRANSFORM_BY_SHAPE = {(3, 3): _linear, (3, 4): _affine, (4, 4): _projective}


def transform_points(pts, transform):
    apply = _TRANSFORM_BY_SHAPE.get(tuple(transform.shape))
    if apply is None:
        raise NotImplementedError(f'transform of shape {tuple(transform.shape)}')
    return apply(pts, transform)


def _to_device(value):
    if torch.is_tensor(value):
        return value.cuda()
    if isinstance(value, list):
        return [t.cuda() for t in value]
    raise NotImplementedError(f'to_cuda: {type(value).__name__}')


def to_cuda(data):
    if isinstance(data, dict):
        return {name: _to_device(v) for name, v in data.items()}
    if isinstance(data, list):
        return [_to_device(v) for v in data]
    raise NotImplementedError(f'to_cuda: {type(data).__name__}')


def load_checkpoint(path):
    """The reference's checkpoints were saved from CUDA tensors together with optimizer state
    (train/trainer.py:77-84); load them on the host."""
    return torch.load(path, map_location='cpu', weights_only=False)
