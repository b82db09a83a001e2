"""Fits per second of agp_fit_create_batch at N in {512, 1024, 2048, 4096} for B in {1, 8, 32} (device-resident inputs), with the
aggregate fraction of the fp64 MFMA peak; B = 1 is agp_fit_create."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from bench import fit_batch_rates

ctx = ab.Context(0)
batches = tuple(int(b) for b in os.environ.get('FIT_BATCHES', '1,8,32').split(','))
for row in fit_batch_rates(ab, ctx, sizes=[int(a) for a in sys.argv[1:]] or None, batches=batches):
    print(row, flush=True)
