"""Ad-hoc GPU check (run through gpurun): (S + D/r) z = g from the dense and from the dissected factorisation of the
reduced camera system against numpy, at a given size `n_cam,n_pt,obs_per_pt` (SFMHIP_BA_ND_CUTS / _DEBUG / _VERBOSE
are read by the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
os.environ["SFMHIP_BA_ND_VERBOSE"] = "1"
nc, npt, k = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "560,8000,8").split(",")]
pb = synth.ba_problem(nc, npt, k, seed=5)
for mode in ("0", "1"):
    os.environ["SFMHIP_BA_ND"] = mode
    prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    z, info = prob.reduced_step(1e4)
    S, g, cost = prob.reduced_system(1e4)
    zr = np.linalg.solve(S, g)
    e = np.abs(z - zr) / (np.abs(zr).max())
    cam_e = e[:-1].reshape(nc, 6).max(axis=1)
    bad = np.nonzero(cam_e > 1e-8)[0]
    print("mode", mode, prob.reduced_layout(), "info", info, "rel resid %.2e" % (np.linalg.norm(S @ z - g) / np.linalg.norm(g)),
          "bad cams", len(bad), "focal err %.2e" % e[-1])
    nanc = np.nonzero(np.isnan(z[:-1].reshape(nc, 6)).any(axis=1))[0]
    if len(nanc):
        runs = np.split(nanc, np.nonzero(np.diff(nanc) > 1)[0] + 1)
        print("    NaN cameras:", len(nanc), [(int(r[0]), int(r[-1])) for r in runs][:24], "focal nan", np.isnan(z[-1]))
    if len(bad):
        runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
        print("   ", [(int(r[0]), int(r[-1]), "%.1e" % cam_e[r].max()) for r in runs][:24])
    prob.close()
