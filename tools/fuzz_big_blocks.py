"""Extended fuzz of the structured LLT family with blocks of up to 128 rows (round 4) against the oracle: the generator of
tests/test_gpu_structured.py (random block counts / sizes / batches, indefinite blocks, windowed L and L^T solves), four seeds x 40
cases with block sizes on both sides of 64.  Exits non-zero on a difference.   python tools/fuzz_big_blocks.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
load_package()
import jrl_qp_amd.capi as capi
from oracle_lib import get_oracle
from test_gpu_structured import random_structured_cases
o = get_oracle()
bad = 0
for seed, tops, mb in ((1, (65, 96, 128), 4), (2, (70, 100, 127), 6), (3, (64, 65, 66, 128), 3), (4, (8, 48, 128), 7)):
    b = random_structured_cases(capi, o, 40, 9000 + seed, verbose=True, tops=tops, max_blocks=mb)
    print("seed", seed, "tops", tops, "bad", b, flush=True)
    bad += b
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
