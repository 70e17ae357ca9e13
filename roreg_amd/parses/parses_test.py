"""Test-time configuration: the same flags, defaults and `get_config()` contract as the reference's
parses/parses_test.py:24-59 (flat argparse namespace consumed as `cfg` by every stage/network ctor).
Additions (do not rename anything existing): --devices, --resident."""
import argparse

base_dir = './data'
backbone = 'FCGF'


def build_parser():
    parser = argparse.ArgumentParser()
    dirs = parser.add_argument_group('Dirs')
    dirs.add_argument('--base_dir', type=str, default=base_dir, help='base dir containing the whole project')
    dirs.add_argument('--origin_data_dir', type=str, default=f'{base_dir}/origin_data', help='the dir containing whole datas')
    dirs.add_argument('--backbone', type=str, default=backbone, help='name of backbone')
    dirs.add_argument('--output_cache_fn', type=str, default=f'{base_dir}/YOHO_{backbone}/Testset', help='eval cache dir')
    dirs.add_argument('--model_fn', type=str, default=f'./checkpoints/{backbone}', help='well trained model path')
    dirs.add_argument('--SO3_related_files', type=str, default='./utils/group_related', help='SO3 related files path')
    t = parser.add_argument_group('Test_Args')
    t.add_argument('--GF', default='yoho_des', type=str)
    t.add_argument('--RD', action='store_true')
    t.add_argument('--RM', action='store_true')
    t.add_argument('--ET', default='yohoc', type=str, help='ransac/yohoc/yohoo')
    t.add_argument('--testset', default='3dmatch', type=str, help='testset name')
    t.add_argument('--keynum', default=5000, type=int, help='number of keypoints')
    t.add_argument('--max_iter', default=1000, type=int, help='ransac iterations')
    t.add_argument('--ransac_ird', default=0.1, type=float, help='inliner threshold of ransac')
    t.add_argument('--tau_1', default=0.05, type=float, help='tau 1 for FMR, 5%%')
    t.add_argument('--tau_2', default=0.1, type=float, help='tau 2 for FMR, 0.1m')
    t.add_argument('--tau_3', default=0.2, type=float, help='tau 3 for RR, 0.2m')
    t.add_argument('--match_n', default=0.5, type=float,
                   help='use how many correspondences predicted for transformation estimation, 0.99 to use all, if n>=1, use top-n')
    t.add_argument('--bs_GF', default=1250, type=int, help='test batch size for group feature extraction')
    t.add_argument('--bs_ET', default=1000, type=int, help='test batch size for local transformation estimation')
    return parser


parser = build_parser()


def get_config():
    config, unparsed = parser.parse_known_args()
    return config, unparsed


def default_config(**overrides):
    """The defaults as a namespace without touching sys.argv (tests, bench)."""
    cfg = parser.parse_args([])
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def print_usage():
    parser.print_usage()
