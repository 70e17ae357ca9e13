"""Mirror of network/__init__.py:7-20 (test-time entries; the *_train classes are out of scope)."""
from .group_feat import GF_test
from .rot_detect import detector_eqv_test
from .eqv_trans import ET_test
from .rot_coh_match import Match_ot

name2network = {
    'GF_test': GF_test,          # group feature extractor
    'RD_test': detector_eqv_test,  # rotation guided detector
    'RM_test': Match_ot,         # rotation coherence matcher
    'ET_test': ET_test,          # transformation estimation
}
