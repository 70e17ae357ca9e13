"""The reference's stage registries (test/__init__.py:6-22): stage classes are looked up by the names the command line uses."""
from . import detector, estimator, extractor, matcher
from .detector import yoho_det
from .estimator import (R_pre_log, extractor_dr_index, extractor_localtrans, refiner, yohoc, yohoc_ransac, yohoo, yohoo_ransac)
from .extractor import yoho_des
from .matcher import NMS_sample, mutual, yoho_mat

name2extractor = dict(yoho_des=yoho_des)
name2detector = dict(yoho_det=yoho_det)
name2matcher = dict(matmul=mutual, yoho_mat=yoho_mat)
name2estimator = dict(yohoc=yohoc, yohoo=yohoo)
