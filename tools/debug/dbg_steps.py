import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_sampling as TS
from tests import test_gpu_beam as TB
for fn, args in [(TS.test_nucleus_distribution_matches_reference_semantics, (161, 6.89713224467348, 0.643, None, 0.5)),
                 (TS.test_nucleus_distribution_matches_reference_semantics, (5411, 6.3481825106439, 0.97, None, 0.52)),
                 (TB.test_beam_step_matches_oracle, (1, 46465, 46465)), (TB.test_beam_step_matches_oracle, (1, 46465, 46468)),
                 (TB.test_beam_step_matches_oracle, (2, 46465, 46465))]:
    try:
        fn(*args)
        print(args, "ok")
    except AssertionError:
        tb = traceback.format_exc().strip().splitlines()
        print(args, tb[-4:], flush=True)
