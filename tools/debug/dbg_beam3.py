import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_beam as TB
for args in [(9, 191, 238), (9, 191, 191), (10, 191, 238), (9, 500, 500), (6, 191, 238)]:
    try:
        TB.test_beam_step_matches_oracle(*args)
        print(args, "ok")
    except AssertionError:
        tb = traceback.format_exc().strip().splitlines()
        print(args, tb[-3:], flush=True)
