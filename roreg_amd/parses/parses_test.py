"""Test-time configuration with the reference's flag names, defaults and `get_config()` contract (parses/parses_test.py:24-59):
one flat argparse namespace that every stage and network constructor receives as `cfg`.  The flags are declared as a table."""
import argparse

base_dir = './data'
backbone = 'FCGF'

# (group, flag, default, type or None for a store_true switch, help)
_FLAGS = [
    ('Dirs', 'base_dir', base_dir, str, 'root of the data tree; results.log is appended here'),
    ('Dirs', 'origin_data_dir', f'{base_dir}/origin_data', str, 'benchmark scenes (clouds, keypoint lists, gt.log / gt.info)'),
    ('Dirs', 'backbone', backbone, str, 'name of the dense descriptor backbone the group features were built from'),
    ('Dirs', 'output_cache_fn', f'{base_dir}/YOHO_{backbone}/Testset', str, 'inter-stage files of the evaluation'),
    ('Dirs', 'model_fn', f'./checkpoints/{backbone}', str, 'checkpoints: {GF,RD,RM,ET}/model_best.pth below this directory'),
    ('Dirs', 'SO3_related_files', './utils/group_related', str, 'icosahedral group tables (.npy); validated against the built-in copy'),
    ('Test_Args', 'GF', 'yoho_des', str, 'group-feature extractor'),
    ('Test_Args', 'RD', False, None, 'rotation-guided detector + NMS keypoint sampling'),
    ('Test_Args', 'RM', False, None, 'rotation-coherence matcher instead of mutual nearest neighbours'),
    ('Test_Args', 'ET', 'yohoc', str, 'estimator: yohoc (rotation-bin 3-point RANSAC) or yohoo (one-shot RANSAC)'),
    ('Test_Args', 'testset', '3dmatch', str, '3dmatch | 3dLomatch | ETH | demo'),
    ('Test_Args', 'keynum', 5000, int, 'keypoints per cloud'),
    ('Test_Args', 'max_iter', 1000, int, 'RANSAC hypotheses'),
    ('Test_Args', 'ransac_ird', 0.1, float, 'inlier distance of RANSAC [m]'),
    ('Test_Args', 'tau_1', 0.05, float, 'inlier-ratio threshold of the feature matching recall'),
    ('Test_Args', 'tau_2', 0.1, float, 'inlier distance of the feature matching recall [m]'),
    ('Test_Args', 'tau_3', 0.2, float, 'RMSE threshold of the registration recall [m]'),
    ('Test_Args', 'match_n', 0.5, float, 'share of the best-scored correspondences handed to the estimator (>= 1: a count; 0.99: all)'),
    ('Test_Args', 'bs_GF', 1250, int, 'keypoints per extractor batch'),
    ('Test_Args', 'bs_ET', 1000, int, 'correspondences per local-transform batch'),
    # not a reference flag: storage precision of the group features on the device
    ('Test_Args', 'dtype', 'fp32', str, 'fp32 | bf16: keep the group features (FCGF input and YOHO output) in bfloat16 on the device; arithmetic stays float32'),
]


def build_parser():
    parser = argparse.ArgumentParser()
    groups = {}
    for group, name, default, kind, text in _FLAGS:
        g = groups.setdefault(group, parser.add_argument_group(group))
        if kind is None:
            g.add_argument(f'--{name}', action='store_true', help=text)
        else:
            g.add_argument(f'--{name}', default=default, type=kind, help=text)
    return parser


parser = build_parser()


def get_config():
    """(namespace, unparsed arguments) -- Test.py's two parsers read the same sys.argv, hence parse_known_args."""
    return parser.parse_known_args()


def default_config(**overrides):
    """The defaults as a namespace without touching sys.argv (tests, bench)."""
    cfg = parser.parse_args([])
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def print_usage():
    parser.print_usage()
