"""Helpers the stages use (mirror of the test-time subset of utils/utils.py: make_non_exists_dir :9-11,
transform_points :38-46, to_cuda :104-132)."""
import os

import numpy as np
import torch


def make_non_exists_dir(fn):
    if not os.path.exists(fn):
        os.makedirs(fn)


def points_to_hpoints(points):
    n, _ = points.shape
    return np.concatenate([points, np.ones([n, 1])], 1)


def hpoints_to_points(hpoints):
    return hpoints[:, :-1] / hpoints[:, -1:]


def transform_points(pts, transform):
    h, w = transform.shape
    if h == 3 and w == 3:
        return pts @ transform.T
    if h == 3 and w == 4:
        return pts @ transform[:, :3].T + transform[:, 3:].T
    elif h == 4 and w == 4:
        return hpoints_to_points(points_to_hpoints(pts) @ transform.T)
    else:
        raise NotImplementedError


def to_cuda(data):
    if type(data) == list:
        results = []
        for item in data:
            if type(item).__name__ == 'Tensor':
                results.append(item.cuda())
            elif type(item).__name__ == 'list':
                results.append([t.cuda() for t in item])
            else:
                raise NotImplementedError
        return results
    elif type(data) == dict:
        results = {}
        for k, v in data.items():
            if type(v).__name__ == 'Tensor':
                results[k] = v.cuda()
            elif type(v).__name__ == 'list':
                results[k] = [t.cuda() for t in v]
            else:
                raise NotImplementedError
        return results
    else:
        raise NotImplementedError


def load_checkpoint(path):
    """The reference's checkpoints were saved from CUDA tensors together with optimizer state
    (train/trainer.py:77-84); load them on the host."""
    return torch.load(path, map_location='cpu', weights_only=False)
