import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_ndt_amd import synth
from gtsam_ndt_amd.matcher import NdtBatch2D, NdtMatcher2D
def log(*a): print(*a, flush=True)
small = synth.make_pair(4, pair_index=3, n_tgt=20000, n_src=20000)
small2 = synth.make_pair(4, pair_index=4, n_tgt=20000, n_src=20000)
which = sys.argv[1]
with NdtBatch2D() as b:
    if which == "two_equal":
        r = b.align([(small["tx"], small["ty"]), (small2["tx"], small2["ty"])], [(small["sx"], small["sy"]), (small2["sx"], small2["sy"])], [small["init"], small2["init"]])
    elif which == "ragged_src":
        r = b.align([(small["tx"], small["ty"]), (small2["tx"], small2["ty"])], [(small["sx"][:777], small["sy"][:777]), (small2["sx"], small2["sy"])], [small["init"], small2["init"]])
    elif which == "ragged_tgt":
        r = b.align([(small["tx"][:5000], small["ty"][:5000]), (small2["tx"], small2["ty"])], [(small["sx"], small["sy"]), (small2["sx"], small2["sy"])], [small["init"], small2["init"]])
    elif which == "sparse":
        st = (np.array([0.0, 10.0], np.float32), np.array([0.0, 10.0], np.float32))
        r = b.align([st, (small2["tx"], small2["ty"])], [(small["sx"], small["sy"]), (small2["sx"], small2["sy"])], [small["init"], small2["init"]])
    elif which == "far":
        r = b.align([(small["tx"], small["ty"])], [(small["sx"] + 1000.0, small["sy"])], [small["init"]])
    elif which == "one":
        r = b.align([(small["tx"], small["ty"])], [(small["sx"], small["sy"])], [small["init"]])
    log(which, [(x.status, x.iterations, x.pose) for x in r])
