"""Run the reference's Test.py unchanged on the MI355X path.

    python -m roreg_amd.dropin /path/to/RoReg/Test.py --RD --ET yohoo --keynum 5000 --testset 3dmatch

install() registers this package's mirrors under the reference's top-level module names (network, test, utils.*,
parses.parses_test, dataops.dataset), so `from test.evaluator import yoho_evaluator` etc. resolve here.  INTEGRATION.md
shows the equivalent one-line edits for maintainers who prefer to keep the reference tree importable."""
import importlib
import runpy
import sys

_ALIASES = {
    'network': 'roreg_amd.network', 'network.ops': 'roreg_amd.network.ops', 'network.group_feat': 'roreg_amd.network.group_feat',
    'network.rot_detect': 'roreg_amd.network.rot_detect', 'network.eqv_trans': 'roreg_amd.network.eqv_trans',
    'network.rot_coh_match': 'roreg_amd.network.rot_coh_match',
    'test': 'roreg_amd.test', 'test.extractor': 'roreg_amd.test.extractor', 'test.detector': 'roreg_amd.test.detector',
    'test.matcher': 'roreg_amd.test.matcher', 'test.estimator': 'roreg_amd.test.estimator', 'test.evaluator': 'roreg_amd.test.evaluator',
    'utils': 'roreg_amd.utils', 'utils.utils': 'roreg_amd.utils.utils', 'utils.r_eval': 'roreg_amd.utils.r_eval',
    'utils.knn_search': 'roreg_amd.utils.knn_search', 'utils.RR_cal': 'roreg_amd.utils.RR_cal',
    'parses': 'roreg_amd.parses', 'parses.parses_test': 'roreg_amd.parses.parses_test',
    'dataops': 'roreg_amd.dataops', 'dataops.dataset': 'roreg_amd.dataops.dataset',
}


def install():
    for alias, target in _ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        print(__doc__)
        return 2
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
