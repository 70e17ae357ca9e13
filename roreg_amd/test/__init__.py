"""Stage registry (mirror of test/__init__.py:6-22)."""
from .extractor import yoho_des
from .detector import yoho_det
from .matcher import NMS_sample, mutual, yoho_mat
from .estimator import (R_pre_log, refiner, extractor_dr_index, yohoc_ransac, yohoc, extractor_localtrans, yohoo_ransac,
                        yohoo)

name2extractor = {'yoho_des': yoho_des}
name2detector = {'yoho_det': yoho_det}
name2matcher = {'matmul': mutual, 'yoho_mat': yoho_mat}
name2estimator = {'yohoc': yohoc, 'yohoo': yohoo}
