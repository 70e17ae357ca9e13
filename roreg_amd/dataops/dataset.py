"""Evaluation datasets without open3d (mirror of dataops/dataset.py:41-225, test-time registries only).

ThrDMatchPartDataset exposes what the stages use: .name .pair_ids .pc_ids .gt_dir .get_kps(id) .get_transform(id0,id1)
.get_pc(id).  Point clouds are read with a small PLY reader (ascii / binary little-endian vertex x,y,z); keypoints are
indices into the cloud (Keypoints/cloud_bin_{k}Keypoints.txt) and are cached as Keypoints_PC/*.npy exactly like the
reference writes them -- but an existing cache is reused instead of re-reading the PLY on every call."""
import os

import numpy as np

from ..utils.utils import make_non_exists_dir

_PLY_TYPES = {'char': 'i1', 'uchar': 'u1', 'short': 'i2', 'ushort': 'u2', 'int': 'i4', 'uint': 'u4', 'float': 'f4', 'double': 'f8',
              'int8': 'i1', 'uint8': 'u1', 'int16': 'i2', 'uint16': 'u2', 'int32': 'i4', 'uint32': 'u4', 'float32': 'f4', 'float64': 'f8'}


def read_ply_points(path):
    """[N,3] float64 vertex positions of a PLY file (what o3d.io.read_point_cloud(...).points returns)."""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'{path}: not a PLY file')
        fmt = None
        n_vert = 0
        props = []
        in_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f'{path}: truncated header')
            tok = line.decode('ascii', 'replace').split()
            if not tok:
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError(f'{path}: list property on vertex element')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        names = [p[0] for p in props]
        if fmt == 'ascii':
            data = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            return np.stack([data[:, names.index(c)] for c in 'xyz'], 1).astype(np.float64)
        endian = '<' if fmt == 'binary_little_endian' else '>'
        dt = np.dtype([(n, endian + t) for n, t in props])
        data = np.frombuffer(f.read(n_vert * dt.itemsize), dtype=dt, count=n_vert)
        return np.stack([data[c] for c in 'xyz'], 1).astype(np.float64)


class ThrDMatchPartDataset:
    def __init__(self, root_dir, stationnum, gt_dir=None):
        self.root = root_dir
        self.gt_dir = f'{self.root}/PointCloud/gt.log' if gt_dir is None else gt_dir
        self.kps_pc_fn = [f'{self.root}/Keypoints_PC/cloud_bin_{k}Keypoints.npy' for k in range(stationnum)]
        self.kps_fn = [f'{self.root}/Keypoints/cloud_bin_{k}Keypoints.txt' for k in range(stationnum)]
        self.pc_ply_paths = [f'{self.root}/PointCloud/cloud_bin_{k}.ply' for k in range(stationnum)]
        self.pc_txt_paths = [f'{self.root}/PointCloud/cloud_bin_{k}.txt' for k in range(stationnum)]
        self.pair_id2transform = self.parse_gt_fn(self.gt_dir)
        self.pair_ids = [tuple(v.split('-')) for v in self.pair_id2transform.keys()]
        self.pc_ids = [str(k) for k in range(stationnum)]
        self.pair_num = len(self.pair_ids)
        self.name = '3dmatch/kitchen'

    @staticmethod
    def parse_gt_fn(fn):
        """gt.log: 5 lines per pair -- 'id0 id1 n' then a 4x4; the first 3 rows are kept as float32 (dataset.py:60-75)."""
        with open(fn, 'r') as f:
            lines = f.readlines()
        out = {}
        for k in range(len(lines) // 5):
            id0, id1 = [int(float(v)) for v in lines[k * 5].split()[0:2]]
            rows = [np.array(lines[k * 5 + r].split(), dtype=np.float32) for r in (1, 2, 3)]
            out['-'.join((str(id0), str(id1)))] = np.stack(rows, 0)
        return out

    def get_pair_ids(self):
        return self.pair_ids

    def get_pair_nums(self):
        return len(self.pair_ids)

    def get_cloud_ids(self):
        return self.pc_ids

    def get_pc_dir(self, cloud_id):
        return self.pc_ply_paths[int(cloud_id)]

    def get_key_dir(self, cloud_id):
        return self.kps_fn[int(cloud_id)]

    def get_name(self):
        return self.name

    def get_pc(self, pc_id):
        if os.path.exists(self.pc_ply_paths[int(pc_id)]):
            return read_ply_points(self.pc_ply_paths[int(pc_id)])
        return np.loadtxt(self.pc_txt_paths[int(pc_id)], delimiter=',')

    def get_transform(self, id0, id1):
        return self.pair_id2transform['-'.join((id0, id1))]

    def get_kps(self, cloud_id):
        k = int(cloud_id)
        if os.path.exists(self.kps_pc_fn[k]) and os.path.exists(self.kps_fn[k]) and \
                os.path.getmtime(self.kps_pc_fn[k]) >= os.path.getmtime(self.kps_fn[k]):
            return np.load(self.kps_pc_fn[k])
        pc = self.get_pc(cloud_id)
        if os.path.exists(self.kps_fn[k]):
            key_idxs = np.loadtxt(self.kps_fn[k]).astype(int)
        else:                                               # random sample 5000 (dataset.py:118-129)
            key_idxs = np.arange(pc.shape[0])
            np.random.shuffle(key_idxs)
            key_idxs = key_idxs[0:5000]
            make_non_exists_dir(f'{self.root}/Keypoints')
            np.savetxt(self.kps_fn[k], key_idxs)
        keys = pc[key_idxs]
        make_non_exists_dir(f'{self.root}/Keypoints_PC')
        np.save(self.kps_pc_fn[k], keys)
        return keys


_3DMATCH_SCENES = ["kitchen", "sun3d-home_at-home_at_scan1_2013_jan_1", "sun3d-home_md-home_md_scan9_2012_sep_30", "sun3d-hotel_uc-scan3",
                   "sun3d-hotel_umd-maryland_hotel1", "sun3d-hotel_umd-maryland_hotel3", "sun3d-mit_76_studyroom-76-1studyroom2",
                   "sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika"]
_3DMATCH_STATIONS = [60, 60, 60, 55, 57, 37, 66, 38]
_REGISTRY = {
    'demo': (['kitchen'], [2]),
    '3dmatch': (_3DMATCH_SCENES, _3DMATCH_STATIONS),
    '3dLomatch': (_3DMATCH_SCENES, _3DMATCH_STATIONS),
    'ETH': (['gazebo_summer', 'gazebo_winter', 'wood_autumn', 'wood_summer'], [32, 31, 32, 37]),
    'WHU-TLS': (['Park', 'Mountain', 'Campus', 'RiverBank', 'UndergroundExcavation', 'Tunnel'], [32, 6, 10, 7, 12, 7]),
}


def get_dataset_name(dataset_name, origin_data_dir):
    """{'wholesetname': name, scene: ThrDMatchPartDataset, ...} (dataset.py:132-196; the training sets are out of scope)."""
    if dataset_name not in _REGISTRY:
        raise NotImplementedError(dataset_name)
    scenes, stations = _REGISTRY[dataset_name]
    datasets = {'wholesetname': f'{dataset_name}'}
    for scene, n in zip(scenes, stations):
        if dataset_name == '3dLomatch':                      # same clouds as 3dmatch, low-overlap pair list
            root_dir = f'{origin_data_dir}/3dmatch/{scene}'
            ds = ThrDMatchPartDataset(root_dir, n, f'{root_dir}/PointCloud/gtLo.log')
        else:
            ds = ThrDMatchPartDataset(f'{origin_data_dir}/{dataset_name}/{scene}', n)
        ds.name = f'{dataset_name}/{scene}'
        datasets[scene] = ds
    return datasets
