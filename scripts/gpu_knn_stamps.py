"""In-kernel timeline of the k-NN kernel (diagnostic build SFM_DBG=4, SFMHIP_SO=.../libsfmhip_dbg4.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, matcher, _lib
ctx = _lib.default_context()
imgs = synth.sift_image_set()
s = matcher.ImageSet(imgs, ctx=ctx)
pairs = synth.all_pairs(len(imgs))
pl = matcher.MatchPlan(s, pairs)
for it in range(3):
    s.prepare_async(); pl.run_async(0.8); ctx.synchronize()
buf = np.zeros(4096 * 8 * 16, dtype=np.uint64)
L = _lib.lib()
L.sfmhip_dbg_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.sfmhip_dbg_read_stamps(buf.ctypes.data, buf.nbytes) == 0
nwav = int(os.environ.get("SFMHIP_KNN_NW", "8"))
print('queries redone exactly by the compaction kernel (3 sweeps):', int(buf[-1]), ' of them overflow-flagged:', int(buf[-2]))
st = buf.reshape(4096, 8, 16).astype(np.int64)[:, :nwav, :]
st = st[st[:, 0, 5] > 0]          # (persistent workgroups: two per compute unit stamp, each its LAST work item)
print("workgroups that stamped:", len(st))
d = np.diff(st[:, :, :6], axis=2)
names = ["prologue(bq, first stage)", "main loop", "drain+resolve u0", "resolve u1", "emit"]
for i, n in enumerate(names):
    print(f"{n:28s} median {np.median(d[:, :, i]):9.0f}  p90 {np.percentile(d[:, :, i], 90):9.0f} cycles")
print("total median", np.median(st[:, :, 5] - st[:, :, 0]), " candidate-loop trips per wave: median", np.median(st[:, :, 6]), "p90", np.percentile(st[:, :, 6], 90), "max", st[:, :, 6].max())
wg = st[:, :, 5].max(axis=1) - st[:, :, 0].min(axis=1)
print("workgroup span median", np.median(wg))

# phases of the resolve (both query tiles together)
print("drain+thr+mask+candidates (2->8)", np.median(st[:, :, 8] - st[:, :, 2]), " list pack (8->9)", np.median(st[:, :, 9] - st[:, :, 8]),
      " rows+dots (9->10)", np.median(st[:, :, 10] - st[:, :, 9]), " read back + merge (10->3)", np.median(st[:, :, 3] - st[:, :, 10]),
      " [rows: issue (9->11)", np.median(st[:, :, 11] - st[:, :, 9]), "wait (11->12)", np.median(st[:, :, 12] - st[:, :, 11]), "dots (12->10)", np.median(st[:, :, 10] - st[:, :, 12]), "]",
      " rows in the list of tile 0: median", np.median(st[:, :, 13]), "max", st[:, :, 13].max())
c0, c1 = st[:, :, 13].ravel(), st[:, :, 7].ravel()
print("listed rows per query tile: tile 0 mean %.1f p10 %d p50 %d p90 %d; tile 1 mean %.1f; max of the two mean %.1f; trips mean %.2f; share of waves with max > 64: %.2f, > 128: %.2f" % (
    c0.mean(), np.percentile(c0, 10), np.percentile(c0, 50), np.percentile(c0, 90), c1.mean(), np.maximum(c0, c1).mean(), st[:, :, 6].mean(),
    (np.maximum(c0, c1) > 64).mean(), (np.maximum(c0, c1) > 128).mean()))

# co-residency: workgroups by CU (HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0])
hw = st[:, 0, 14]; xcc = st[:, 0, 15] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
start = st[:, :nwav, 1].min(axis=1); mid = st[:, :nwav, 2].max(axis=1); end = st[:, :nwav, 5].max(axis=1)
used = st[:, 0, 5] > 0
import collections
by = collections.defaultdict(list)
for i in np.nonzero(used)[0]: by[int(cu[i])].append(i)
print("CUs seen", len(by), "workgroups per CU: median", np.median([len(v) for v in by.values()]))
# fraction of main-loop time of a workgroup during which another workgroup of the same CU is also in its main loop
ov = []
for v in by.values():
    for a in v:
        o = 0
        for b in v:
            if a != b: o += max(0, min(mid[a], mid[b]) - max(start[a], start[b]))
        ov.append(o / max(1, mid[a] - start[a]))
print("main-loop overlap with a co-resident workgroup's main loop: median %.2f  mean %.2f" % (np.median(ov), np.mean(ov)))
