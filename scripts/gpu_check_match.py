"""Ad-hoc GPU check (run through gpurun): matcher + triangulation parity against the oracle and
a first timing of the cfg2 sweep.  Not part of the product path."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import orc
from sfm_danpipeline_amd import synth, matcher, _lib

ctx = _lib.default_context()

ctx.set_timing(True)
ok_all = True

def cmp_pair(q, t, norm, tag):
    global ok_all
    s = matcher.ImageSet([q, t], norm=norm, ctx=ctx)
    s.prepare_async()
    pl = matcher.MatchPlan(s, [[0, 1]])
    pl.run_async(0.8)
    ki, kd = pl.fetch_knn(0)
    cnt, oq, ot, od = pl.fetch()
    r = orc.match_knn2(q, t, norm=(orc.NORM_HAMMING if norm == _lib.HAMMING else orc.NORM_L2), want_knn=True, threads=8)
    e_idx = np.array_equal(ki, r[3]); e_d = np.array_equal(kd.view(np.uint32), r[4].view(np.uint32))
    e_m = np.array_equal(oq, r[0]) and np.array_equal(ot, r[1]) and np.array_equal(od.view(np.uint32), r[2].view(np.uint32))
    print(f"{tag}: knn idx {e_idx} dist {e_d} matches {e_m} n={len(oq)}/{len(r[0])}", flush=True)
    if not (e_idx and e_d and e_m):
        ok_all = False
        bad = np.nonzero((ki != r[3]).any(axis=1))[0][:5]
        for b in bad: print("   q", b, "gpu", ki[b], kd[b], "orc", r[3][b], r[4][b])
    pl.close(); s.close()

imgs = synth.sift_image_set(3, 700, 128, bank=900)
cmp_pair(imgs[0], imgs[1], _lib.L2, "sift f32 700x700")
cmp_pair(imgs[0][:33], imgs[1][:257], _lib.L2, "sift f32 33x257")
cmp_pair(imgs[0].astype(np.uint8), imgs[1].astype(np.uint8), _lib.L2, "sift u8 700x700")
rng = np.random.default_rng(3)
qa = rng.integers(0, 2, (300, 128)).astype(np.float32) * 255  # extreme values: big distances, sqrt collisions
ta = rng.integers(0, 2, (400, 128)).astype(np.float32) * 255
cmp_pair(qa, ta, _lib.L2, "extremes 0/255 (fixup path)")
dup = np.repeat(imgs[1][:50], 3, axis=0)  # duplicate train rows: ties -> lower index
cmp_pair(imgs[0][:100], dup, _lib.L2, "duplicate trains (ties)")
nonint = imgs[0][:200] + 0.25
cmp_pair(nonint, imgs[1][:300], _lib.L2, "non-integer f32 (exact kernel)")
orbs = synth.orb_image_set(2, 600, bank=800)
cmp_pair(orbs[0], orbs[1], _lib.HAMMING, "orb hamming 600x600")
cmp_pair(orbs[0], orbs[1], _lib.L2, "orb L2_U8 (reference-literal)")
ak = rng.integers(0, 256, (300, 61), dtype=np.uint8); ak2 = rng.integers(0, 256, (280, 61), dtype=np.uint8)
cmp_pair(ak, ak2, _lib.L2, "akaze-like 61B L2")
cmp_pair(imgs[0][:40], imgs[1][:1], _lib.L2, "nt=1")
cmp_pair(imgs[0][:40], imgs[1][:2], _lib.L2, "nt=2")

# triangulation
from sfm_danpipeline_amd import triangulate as tri
sc = synth.two_view_scene(5000)
X, err, keep = tri.triangulate_points(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"], ctx=ctx)
Xo, erro, keepo = orc.triangulate(sc["P1"], sc["P2"], sc["K"], sc["dist"], sc["xy1"], sc["xy2"])
print("tri: X maxabs", np.abs(X - Xo).max(), "bit-equal X", np.array_equal(X, Xo), "keep eq", np.array_equal(keep, keepo), "err eq", np.array_equal(err, erro), flush=True)
ok_all &= np.array_equal(keep, keepo)

# cfg2 timing
t0 = time.time(); imgs = synth.sift_image_set(); print("gen cfg2", time.time() - t0, flush=True)
s = matcher.ImageSet(imgs, ctx=ctx)
pairs = synth.all_pairs(len(imgs))
pl = matcher.MatchPlan(s, pairs)
for it in range(3):
    ctx.synchronize(); t0 = time.time()
    s.prepare_async(); pl.run_async(0.8); ctx.synchronize()
    dt = time.time() - t0
    print(f"cfg2 sweep {it}: {dt*1e3:.3f} ms -> {len(pairs)/dt:.0f} pairs/s ; stages {pl.last_timing()}", flush=True)
cnt, oq, ot, od = pl.fetch()
print("cfg2 total matches", cnt.sum(), "checksum", matcher.match_checksum(cnt, oq, ot))
# spot-check 3 pairs against the oracle at full size
for p in (0, 611, 1224):
    r = orc.match_knn2(imgs[pairs[p, 0]], imgs[pairs[p, 1]], threads=8)
    a = pl.fetch_pair(p)
    e = all(np.array_equal(x, y) for x, y in zip(a, r))
    print("cfg2 pair", p, "eq oracle", e, len(a[0])); ok_all &= e
tops = 2.0 * 2000 * 2000 * 128 * len(pairs) / pl.last_timing()["knn_s"] / 1e12
print(f"knn kernel: {tops:.1f} TOP/s  ({tops/5000*100:.1f}% of i8 dense peak ~5 POP/s)")
print("ALL OK" if ok_all else "SOME MISMATCH")
