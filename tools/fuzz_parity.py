import sys, traceback
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import glimslib_amd._backend as backend
import test_gpu_parity as T
bad = 0
for seed in range(8, int(sys.argv[1]) if len(sys.argv) > 1 else 136):
    try:
        T.test_randomised_small_problems_match_the_oracle(backend, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", repr(e)[:300], flush=True)
print("done, failures:", bad)
