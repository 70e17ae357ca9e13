"""roreg_amd -- MI355X-native implementation of RoReg's per-pair registration hot path.

Layout (DESIGN.md has the full map):
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/roreg_hip.h) -> libroreg_hip.so
  hip.py       ctypes binding of the C-ABI (raw device pointers + hipStream_t; no torch types cross it)
  network/     mirrors of network/{group_feat,rot_detect,eqv_trans,rot_coh_match}.py (same class names,
               same state_dict keys) whose forward() runs the HIP kernels
  test/        mirrors of test/{extractor,detector,matcher,estimator,evaluator}.py (file-coupled stages)
  utils/       mirrors of utils/{knn_search,r_eval,utils,RR_cal}.py
  parses/      parses_test (same flags)
  engine.py    device-resident scene pipeline used by bench.py (no per-stage file round trips)
  dropin.py    installs the mirrors under the reference's module names so Test.py runs unchanged
"""
__version__ = '0.1.0'
