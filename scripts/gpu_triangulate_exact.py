import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import orc
from sfm_danpipeline_amd import _lib, triangulate, synth
ctx = _lib.default_context()
for m, seed in ((1, 1), (63, 2), (5000, 3), (20000, 4)):
    sc = synth.two_view_scene(m, seed=seed)
    for dist in (sc["dist"], np.array([0.05, -0.02, 0.001, -0.0005, 0.01])):
        X, err, keep = triangulate.triangulate_points(sc["P1"], sc["P2"], sc["K"], dist, sc["xy1"], sc["xy2"], ctx=ctx)
        Xo, erro, keepo = orc.triangulate(sc["P1"], sc["P2"], sc["K"], dist, sc["xy1"], sc["xy2"])
        print(m, bool(dist.any()), "X bit-equal:", np.array_equal(X.view(np.uint64), Xo.view(np.uint64)), "max|dX|", np.abs(X - Xo).max(), "err bit-equal:", np.array_equal(err.view(np.uint32), erro.view(np.uint32)), "keep equal:", np.array_equal(keep, keepo))
