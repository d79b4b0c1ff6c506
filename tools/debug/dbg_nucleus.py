import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_sampling as TS
for args in [(4881, 1.1854135049687542, 0.681, None, 1.08), (15946, 4.275605426803784, 0.973, 394, 0.87)]:
    try:
        TS.test_nucleus_distribution_matches_reference_semantics(*args)
        print(args, "ok")
    except AssertionError:
        tb = traceback.format_exc().strip().splitlines()
        print(args, tb[-3:], flush=True)
