"""CPU, world_size 2, gloo: the N>1 data path -- pair sharding with host-side merge, and the
per-iteration all-reduce of the reduced camera system -- with the oracle as the per-rank engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from sfm_danpipeline_amd import sharding, synth
    # --- matching: each rank matches its shard of the pair list
    imgs = synth.sift_image_set(5, 60, 128, bank=90, seed=2)
    pairs = synth.all_pairs(5)
    shards = sharding.shard_pairs(pairs, [len(i) for i in imgs], world)
    cnt, qs, ts, ds = [], [], [], []
    for p in shards[rank]:
        q, t, d = orc.match_knn2(imgs[pairs[p, 0]], imgs[pairs[p, 1]])
        cnt.append(len(q)); qs.append(q); ts.append(t); ds.append(d)
    mine = (np.array(cnt, np.int32), np.concatenate(qs), np.concatenate(ts), np.concatenate(ds))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)  # result gather only; no collective on the data path
    # --- BA: partial reduced systems summed by all-reduce
    pb = synth.ba_problem(6, 120, 4, seed=14)
    full = orc.ba_reduced_system(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    scale = full[3]
    loc = sharding.local_ba_problem(pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], pb["pts0"], rank, world)
    sc = np.concatenate([scale[:36], scale[36 + 3 * loc["lo"]:36 + 3 * loc["hi"]], scale[-1:]])
    Sr, gr, cr, _ = orc.ba_reduced_system(pb["cams0"], loc["pts"], pb["focal0"], loc["obs_cam"], loc["obs_pt"],
                                          loc["obs_xy"], scale=sc)
    Sr = Sr - np.diag(np.diag(Sr))
    buf = torch.from_numpy(np.concatenate([Sr.ravel(), gr, [cr]]))
    dist.all_reduce(buf)  # the C1 collective of SURVEY.md section 2a
    if rank == 0:
        counts, q, t, d = sharding.merge_pair_results(shards, gathered, len(pairs))
        ok_match = True
        off = 0
        for p in range(len(pairs)):
            rq, rt, rd = orc.match_knn2(imgs[pairs[p, 0]], imgs[pairs[p, 1]])
            n = counts[p]
            ok_match &= n == len(rq) and np.array_equal(q[off:off + n], rq) and np.array_equal(t[off:off + n], rt)
            off += n
        S = buf[:-38].numpy().reshape(37, 37)
        offdiag = full[0] - np.diag(np.diag(full[0]))
        ok_ba = (np.abs(S - offdiag).max() <= 1e-12 * np.abs(full[0]).max()
                 and np.abs(buf[-38:-1].numpy() - full[1]).max() <= 1e-12 * np.abs(full[1]).max()
                 and abs(float(buf[-1]) - full[2]) <= 1e-12 * full[2])
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write(f"{int(ok_match)} {int(ok_ba)}")
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_two_gloo(tmp_path):
    from oracle import orc
    orc.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result.txt").read() == "1 1"
