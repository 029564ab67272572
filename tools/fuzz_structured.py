"""Random-shape parity fuzz of the structured factorisations and solves (jrlqp_blocks_*) against the CPU oracle, bit for bit:
random block counts and sizes (1..64: register kernel up to 48 rows, barrier kernel beyond), batches, indefinite blocks at
random positions (column 0 of a block -- the failure the reference does not notice -- and later columns), and, after a
successful factorisation, L and L^T solves of random right-hand sides with random zero windows.
usage: fuzz_structured.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_package; load_package()
import jrl_qp_amd.capi as capi
from oracle_lib import get_oracle
from test_gpu_structured import spd, views, to_dev, from_dev, oracle_factor, rhs_dev
o = get_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
from test_gpu_structured import random_structured_cases
bad = random_structured_cases(capi, o, ncases, int(sys.argv[2]) if len(sys.argv) > 2 else 11, verbose=True)
print(f"{ncases} structured cases, {bad} mismatches")
sys.exit(1 if bad else 0)
