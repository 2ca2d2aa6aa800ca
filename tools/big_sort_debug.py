import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, femto_amd
from femto_amd import textgen as tg
k = int(sys.argv[1])
t = tg.t_acgt(1 << k, 5)
t0 = time.time()
femto_amd.build_index(f"/tmp/big{k}", [t], device=0)
print("built", time.time() - t0)
