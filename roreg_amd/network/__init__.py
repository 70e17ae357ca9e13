"""Network registry under the reference's names (network/__init__.py:7-20).  Only the test-time classes exist here; training is out
of scope."""
from .eqv_trans import ET_test
from .group_feat import GF_test
from .rot_coh_match import Match_ot
from .rot_detect import detector_eqv_test

name2network = dict(
    GF_test=GF_test,              # 60-rotation group features (YOHO descriptor, part I)
    RD_test=detector_eqv_test,    # saliency from rotation coherence of the group features
    RM_test=Match_ot,             # optimal-transport matcher over rotation-coherent neighbourhoods
    ET_test=ET_test,              # residual rotation of a correspondence (quaternion)
)
