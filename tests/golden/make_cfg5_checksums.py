"""Generates tests/golden/cfg5_checksums.npz: the oracle's answer for EVERY pair of BASELINE cfg5.

cfg5 = 500 images x 5000 ORB-256-bit rows, all 124 750 pairs q < t, Hamming k-NN-2 + ratio 0.8
(the matcher semantics are those of the reference's getMatching, src/Sfm.cpp:590-608, with
cv::NORM_HAMMING in place of the hard-wired L2).  The C restatement (oracle/sfm_oracle_match.c, the row-by-row
insertion-list matcher -- not the blocked timing leg) runs once over all pairs in the build container, about
3.1e12 popcount distances: tens of minutes on 8 cores.  Per pair it leaves the match count and the [sum, xor] of
orc_match_mix(queryIdx, trainIdx, distance bits) over the pair's matches -- the same function
sfm_danpipeline_amd.synth.pair_checksums evaluates on the device's lists.

The GPU test (tests/test_gpu_match.py::test_cfg5_every_pair_against_the_oracle) and bench.py's cfg5 leg compare the
whole sweep with this file.  Usage:  python tests/golden/make_cfg5_checksums.py [--threads 8] [--chunk 1000]
A partial run is kept in tests/golden/_cfg5_partial.npz (git-ignored) and resumed.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import orc  # noqa: E402
from sfm_danpipeline_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--chunk", type=int, default=1000)
    ap.add_argument("--limit", type=int, default=0, help="stop after this many pairs (timing trials)")
    args = ap.parse_args()
    orc.use_native()  # (-march=native build of the same sources: integer arithmetic, the same lists)
    imgs = synth.orb_image_set()
    pairs = synth.all_pairs(len(imgs))
    n = len(pairs) if not args.limit else min(args.limit, len(pairs))
    part = os.path.join(HERE, "_cfg5_partial.npz")
    counts = np.zeros(len(pairs), np.int32)
    cs = np.zeros((len(pairs), 2), np.uint64)
    done = 0
    if os.path.exists(part):
        z = np.load(part)
        done = int(z["done"])
        counts[:done] = z["counts"][:done]
        cs[:done] = z["checksums"][:done]
        print(f"resuming at pair {done}", flush=True)
    t0 = time.time()
    while done < n:
        hi = min(done + args.chunk, n)
        c, s = orc.match_many_checksum(imgs, pairs[done:hi], norm=orc.NORM_HAMMING, threads=args.threads)
        counts[done:hi] = c
        cs[done:hi] = s
        done = hi
        np.savez(part, done=done, counts=counts, checksums=cs)
        el = time.time() - t0
        print(f"{done}/{n} pairs, {el:.0f} s", flush=True)
    if n == len(pairs):
        np.savez_compressed(os.path.join(HERE, "cfg5_checksums.npz"), counts=counts, checksums=cs,
                            n_images=np.int32(len(imgs)), n_feat=np.int32(len(imgs[0])), seed=np.int32(4321))
        os.remove(part)
        print("wrote cfg5_checksums.npz:", int(counts.sum()), "matches")


if __name__ == "__main__":
    main()
