#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X:
    "image-pairs matched/sec + BA iterations/sec (200 cams, 100k pts, 1M obs)".

One *step* = one pass of the hot path over one batch of synthetic input:
  (a) matching:  prepare (f32 -> i8 tiles, norms) + all-pairs k=2 L2 matching + ratio test +
                 ordered compaction of BASELINE cfg2 (50 images x 2000 SIFT-128 rows, 1225 pairs)
                 with the f32 descriptors already resident in HBM;
  (b) BA:        one full LM iteration of BASELINE cfg4 (200 cams / 100k points / 1M obs):
                 linearise + Schur eliminate (+ all-reduce) + reduced solve + back-substitute +
                 candidate cost.
The two halves are timed as two regions of exactly K steps each, each bracketed by a barrier +
device synchronisation; `value` is the matching throughput (pairs/s over all ranks),
`ba_iterations_per_s` the BA rate, `ms_per_step` their sum per step.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL):
  * the headline stays N=1-compatible and weak-scaled: independent units -- every rank matches the 1225 pairs
    of its own 50-image set, no collective on the data path;
  * `cfg5_strong` (every N, 1 included): BASELINE cfg5 -- the 124 750 pairs of 500 ORB-256 images, Hamming --
    dealt over the ranks by sharding.shard_pairs, fixed total work, per-pair checksums merged on rank 0:
    the strong-scaling figure of the matcher, and its checksum does not depend on N;
  * BA is the one cfg4 problem, points block-partitioned over the ranks, the reduced camera
    system summed by one all-reduce per LM iteration (strong scaling; `ba_amdahl` splits the iteration
    into what shards, what is replicated and the exchange).

Extra fields: `sustained` (the same step looped for >= 2 s: past DVFS settling), `value_host_visible`
(every sweep followed by the packed device-to-host fetch of counts + match lists), the dominant kernel's own
duration behind `roofline` (`launch_ms`, HIP events on the library's stream around that one launch).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The ROCm runtime multiplexes a process's HIP streams onto 4 hardware queues unless told otherwise, and streams that share a queue
# run one after the other.  This process holds more than four at a time (two matching streams + their two copy streams + the four
# streams of the batch-mode BA leg): measured on one box, the four concurrent cfg4 problems of `ba_batch` make 7 150 LM iterations/s
# in total on 4 queues and 9 710 on 8 -- nothing else on the line moves.  A knob of the runtime, read when HIP initialises (hence
# before torch is imported); INTEGRATION.md says the same to a host program that drives several problems side by side.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

I8_DENSE_PEAK_TOPS = 5000.0   # 2x the ~2.5 PF dense bf16 MFMA peak (MI355X_MICROARCH.md, Matrix cores)
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TLANEOPS = 78.6        # 256 CU x 128 lanes x 2.4 GHz


def launcher_command(n_ranks, argv, port=None):
    """The command `bench.py --gpus N` starts when it is run without a launcher: one rank per GPU of this node under
    torch.distributed.run (static rendezvous on 127.0.0.1: the container's hostname may not resolve).  SFMHIP_BENCH_LAUNCHER
    replaces the `python -m torch.distributed.run` prefix (tests/test_bench_launcher.py drives the branch with a stub)."""
    import shlex
    import socket
    if port is None:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
    head = os.environ.get("SFMHIP_BENCH_LAUNCHER")
    head = shlex.split(head) if head else [sys.executable, "-m", "torch.distributed.run"]
    return head + ["--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                   os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_ranks, argv):
    """Starts the N ranks as a CHILD process (never exec: a process that may have initialised the GPU must not replace
    itself, and this one has not touched it), relays the child's stdout / stderr and returns its exit status."""
    import subprocess
    cmd = launcher_command(n_ranks, argv)
    print("[bench] --gpus %d without WORLD_SIZE: starting %s" % (n_ranks, " ".join(cmd)), file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")       # (one node by contract: RCCL's bootstrap on the loopback interface)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.Popen(cmd, env=env)           # inherits stdout / stderr: rank 0's JSON line goes straight through
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=256, help="pairs of cfg2 timed on the host cores")
    ap.add_argument("--cpu-ba-iters", type=int, default=8)
    ap.add_argument("--no-cfg5", action="store_true", help="skip the cfg5 strong-scaling leg")
    ap.add_argument("--ba-batch", type=int, default=4,
                    help="independent cfg4 problems in flight per GPU in the batch-mode BA leg (no collective; the BA rate "
                         "that scales with GPUs)")
    ap.add_argument("--no-ba-batch", action="store_true")
    ap.add_argument("--no-score", action="store_true", help="skip the findBestPair scoring leg (E-matrix RANSAC of 1225 pairs)")
    ap.add_argument("--lean", action="store_true",
                    help="timed regions and per-kernel samples only (no sustained / host-visible loops, no cfg5, no CPU "
                         "baseline): the command to run under rocprofv3, whose traces grow with every dispatch")
    ap.add_argument("--cfg5-sample", type=int, default=0,
                    help="with --lean: also sweep this many pairs of BASELINE cfg5 (evenly spaced over the 124 750, checksums asserted "
                         "against tests/golden/cfg5_checksums.npz) so that a profile of the command holds knn_keyed_kernel rows")
    ap.add_argument("--no-adjust-bundle", action="store_true", help="skip the whole-call leg (BundleAdjustment::adjustBundle through "
                                                                     "the C++ mirror, a subprocess)")
    ap.add_argument("--sustain-s", type=float, default=2.0, help="seconds of the sustained loops")
    ap.add_argument("--match-streams", type=int, default=2, choices=(1, 2),
                    help="2 (default): consecutive batches alternate between two resident buffers on two HIP streams -- the "
                         "small prepare / fix-up / compaction kernels of one batch overlap the sweep of the other (+10 %% "
                         "pairs/s); the roofline's launch_ms is sampled from single-stream steps either way.  1: one stream")
    args = ap.parse_args()
    if args.lean:
        args.no_cfg5 = args.no_cpu_baseline = args.no_score = args.no_adjust_bundle = True

    # `python bench.py --gpus N` on its own (no WORLD_SIZE in the environment): this process becomes the launcher -- before
    # anything here has touched a GPU, and without replacing itself -- of N ranks under torch.distributed.run, relays
    # their output (rank 0 prints the JSON line) and exits with the child's status
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (one rank per GPU; the modulo only matters for the functional check of the N>1 path on a
    # single-GPU box: SFMHIP_BENCH_BACKEND=gloo lets two ranks share device 0, which RCCL refuses)
    local_rank %= max(torch.cuda.device_count(), 1)
    backend = os.environ.get("SFMHIP_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")   # (N GPUs of ONE node: no interface to probe for; a launcher's own choice wins)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)
    if world != args.gpus:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")
    # how many ranks the collective backend really connects: the group's size AND an all-reduce of ones over it
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one[0])
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)

    from sfm_danpipeline_amd import _lib, bundle, matcher, sharding, synth

    # all library work rides torch's current stream so that RCCL collectives order with it
    stream = torch.cuda.current_stream(dev)
    ctx = _lib.Context(local_rank, stream=stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ workload (synthetic, seeded)
    n_img, n_feat, dim = 50, 2000, 128
    imgs = synth.sift_image_set(n_img, n_feat, dim, seed=1234 + 1000 * rank)   # rank r: its own image set
    pairs = synth.all_pairs(n_img)
    # --match-streams 2: consecutive batches alternate between two resident buffers on two HIP streams
    # (two contexts): the tail of one sweep and the small prepare / compaction kernels of the next
    # overlap.  Every step is one full prepare + knn + ratio/compaction pass over one batch of 1225 pairs.
    N_STREAMS = args.match_streams
    side_stream = torch.cuda.Stream(dev)
    ctxs = [ctx, _lib.Context(local_rank, stream=side_stream.cuda_stream)]
    isets, plans, d_keep = [], [], []
    for c in ctxs[:N_STREAMS]:
        d_imgs = [torch.from_numpy(a).to(dev) for a in imgs]                    # inputs resident in HBM
        d_keep.append(d_imgs)
        s_ = matcher.ImageSet(n_rows=[n_feat] * n_img, dim=dim, dtype=_lib.F32, norm=_lib.L2, ctx=c)
        for i, t in enumerate(d_imgs):
            s_.adopt_device(i, t.data_ptr(), keepalive=t)
        isets.append(s_)
        plans.append(matcher.MatchPlan(s_, pairs))
    iset, plan = isets[0], plans[0]
    step_no = [0]

    pb = synth.ba_problem(200, 100000, 10, seed=777)
    loc = sharding.local_ba_problem(pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], pb["pts0"], rank, world)
    ba = bundle.BaProblem(200, len(loc["pts"]), loc["obs_cam"], loc["obs_pt"], loc["obs_xy"], ctx=ctx)
    if world > 1:
        ba.set_allreduce(sharding.TorchAllReduce(device=dev), rank, world)
    ba.set_params(pb["cams0"], loc["pts"], pb["focal0"])

    def match_step(one_stream=False):
        j = 0 if one_stream else step_no[0] % N_STREAMS
        step_no[0] += 1
        isets[j].prepare_async()
        plans[j].run_async(0.8)

    # ------------------------------------------------------------------ warmup
    # BA first, the matching sweeps last: the timed matching region then starts from the load it measures.  (With
    # the LM iterations between the warm-up sweeps and the timed sweeps, the first sweeps after the switch from the
    # latency-bound BA kernels to the MFMA-bound sweep ran slow: a 20-step region measured 0.73 ms per sweep where
    # a 100-step one measured 0.686, scripts/gpu_sync_check.py: 0.667 either way without the switch.)
    ba.iterate(max(args.warmup, 1))
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.1:      # (and the device out of its idle power state)
        for _ in range(10):
            match_step()
        torch.cuda.synchronize(dev)
    for _ in range(max(args.warmup, 1)):
        match_step()
    barrier()
    # the measured ceiling of THIS box, right before the timed region: bare i8 MFMAs on random operands for 50 ms (the chip
    # lowers its clock under matrix load; the nominal 5 POP/s assumes 2.4 GHz), and the shader clock while the sweep itself
    # runs (a one-wave sampler on the second stream beside 30 single-stream sweeps)
    sustained_peak_ops, sustained_peak_ghz = ctx.probe_i8_mfma_peak(0.05)
    for _ in range(10):
        match_step(one_stream=True)
    torch.cuda.synchronize(dev)
    ctxs[1].probe_clock_start(0.012)
    for _ in range(30):
        match_step(one_stream=True)
    sweep_clock_ghz = ctxs[1].probe_clock_read()
    for _ in range(max(args.warmup, 1)):
        match_step()
    barrier()

    # ------------------------------------------------------------------ timed: matching, K steps
    knn_s = prep_s = comp_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        match_step()
    barrier()
    t_match = time.perf_counter() - t0
    # per-kernel device time of the dominant kernel: HIP events recorded by the library on the
    # stream the kernels run on, one extra step outside the timed region per sample
    # (stage timing is opt-in in the library: the events cost stream bubbles the timed region does not pay)
    samples = []
    ctx.set_timing(True)
    barrier()
    # (rounds of back-to-back steps, the events of the last step of a round read back: a step timed on
    # its own, with the GPU idle before it, holds a higher clock than the sustained loop does)
    for _ in range(4):
        for _ in range(min(args.steps, 5)):
            match_step(one_stream=True)
        tm = plan.last_timing()          # synchronises on the recorded events
        samples.append((tm["prepare_s"], tm["knn_kernel_s"], tm["compact_s"], tm["knn_s"]))
    ctx.set_timing(False)
    prep_s, knn_s, comp_s, knn_stage_s = [float(np.mean([s[i] for s in samples])) for i in range(4)]
    counts = plan.counts()

    # ------------------------------------------------------------------ sustained (>= 2 s) and host-visible rates
    def loop_for(seconds, fn, unit):
        barrier()
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(unit):
                fn()
            n += unit
            torch.cuda.synchronize(dev)
            if time.perf_counter() - t0 >= seconds:
                break
        barrier()
        return n, time.perf_counter() - t0
    n_sus, t_sus = (0, 1.0) if args.lean else loop_for(args.sustain_s, match_step, 50)
    sustained_pairs_s = n_sus * len(pairs) / t_sus

    # host-visible rate: every sweep's counts + {queryIdx, trainIdx, distance} lists in host memory -- what getMatching's
    # caller sees.  Pipelined (sfmhip_matchplan_pipeline): a second stream packs sweep n's lists into one of two pinned
    # buffers while sweep n + 1 runs; the host takes sweep n - 1's lists (fetch_wait) right after enqueueing sweep n + ...
    hv_seen = [0, 0]
    def match_and_fetch():
        # the lists of the run this plan did BEFORE its last (2 * N_STREAMS steps ago), taken right before the plan is given its
        # next run: that run's packing and copy are long done, the host does not block, and a sweep stays queued on each stream
        # behind the one in flight.  (Round 4 took the plan's LAST run, N_STREAMS steps ago: its packing launch only gets compute
        # units when the sweep in flight gives them back, i.e. near that sweep's end -- the host woke up with nothing queued behind,
        # and every step paid its wake-up and enqueue time: 0.86 of the device-only rate; scripts/gpu_hostvisible_ab2.py.)
        j = step_no[0] % N_STREAMS
        if hv_seen[0] >= 2 * N_STREAMS:
            c_, q_, t_, d_ = plans[j].fetch_wait(back=1)
            hv_seen[1] += int(c_.sum()) == len(q_) == len(t_) == len(d_)
        match_step()
        hv_seen[0] += 1
    if args.lean:
        n_hv, t_hv = 0, 1.0
    else:
        for pl_ in plans:
            pl_.pipeline()
        n_hv, t_hv = loop_for(1.0, match_and_fetch, 50)
        for pl_ in plans:                                     # the last two runs' lists of every plan
            for back_ in (1, 0):
                c_, q_, t_, d_ = pl_.fetch_wait(back=back_)
                assert int(c_.sum()) == len(q_)
                hv_seen[1] += 1
            assert np.array_equal(synth.pair_checksums(c_, q_, t_, d_), synth.pair_checksums(*pl_.fetch()))
        assert hv_seen[1] == hv_seen[0], (hv_seen, "every sweep's lists were seen by the host")
        for pl_ in plans:
            pl_.pipeline(-1)
    host_visible_pairs_s = n_hv * len(pairs) / t_hv
    # (and the stop-and-copy way, sfmhip_matchplan_fetch after every sweep: two synchronisations + two copies per sweep)
    def match_and_copy():
        match_step(one_stream=True)
        plan.fetch()
    n_hc, t_hc = (0, 1.0) if args.lean else loop_for(0.5, match_and_copy, 5)
    host_copy_pairs_s = n_hc * len(pairs) / t_hc

    # ------------------------------------------------------------------ timed: BA, K iterations
    # The timed iterations are iterations of a solve that is still MOVING: cfg4's synthetic start converges in about 40 LM
    # iterations, and sfmhip_ba_iterate (no stopping rule) past that point iterates on rejected steps with a radius that halves,
    # quarters, ... down to 0 -- infinities in the damping, kernels that run 3-4 % faster than on real numbers
    # (scripts/gpu_ba_radius_probe.py, round 5).  So every BA region of this file starts from the start: ba_restart() resets the
    # parameters and takes the first 15 iterations untimed, the K timed ones follow: iterations 16 .. 35 at the default K, independent of
    # --warmup (round 5's window moved with it: 24 .. 43 at --warmup 3, 26 .. 45 at the driver's 5, the last of them at / past
    # convergence); config.ba_timed_iterations on the line.
    BA_PRE = 15      # LM iterations of a fresh solve taken untimed: the K timed ones are iterations 16 .. 15 + K whatever --warmup is
    def ba_restart(p=None, start=None):
        p = ba if p is None else p
        c0_, p0_, f0_ = (pb["cams0"], loc["pts"], pb["focal0"]) if start is None else start
        p.set_params(c0_, p0_, f0_)
        return p.iterate(BA_PRE)
    ba_pre = ba_restart()
    barrier()
    t0 = time.perf_counter()
    ba_sum = ba.iterate(args.steps)
    barrier()
    t_ba = time.perf_counter() - t0
    ba_accepted_timed = int(ba_sum.successful_steps - ba_pre.successful_steps)   # (on the line: the timed iterations moved the solve)
    # per-stage device time: the same iterations once more with stage timing on, outside the
    # timed region (the library accumulates since the run began, hence the difference)
    ba_restart()
    ctx.set_timing(True)
    ba_t0 = ba.last_timing()
    ba.iterate(args.steps)
    ba_t = {k: v - ba_t0[k] for k, v in ba.last_timing().items()}
    ba_layout = dict(ba.reduced_layout(), front_tree=ba.reduced_tree())
    ctx.set_timing(False)
    barrier()
    # sustained BA rate: every ba.iterate issues all-reduces, so the number of batches must be THE SAME on every rank --
    # it is derived from the timed region's duration agreed over the ranks (MAX), not from each rank's own clock
    t_ba_agreed = t_ba
    if world > 1:
        tt = torch.tensor([t_ba], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_ba_agreed = float(tt[0])
    # (a batch = restart + the K timed iterations; the restarts keep the device as busy as the timed part does, only the K
    # iterations behind each count)
    t_batch = (2 * args.steps + BA_PRE) * t_ba_agreed / args.steps
    n_batches = 0 if args.lean else max(1, int(np.ceil(args.sustain_s / max(t_batch, 1e-6))))
    t_sus_ba = 0.0
    for _ in range(n_batches):
        ba_restart()
        barrier()
        t0 = time.perf_counter()
        ba.iterate(args.steps)
        barrier()
        t_sus_ba += time.perf_counter() - t0
    sustained_ba_its = args.steps * n_batches / max(t_sus_ba, 1e-9)

    # ------------------------------------------------------------------ what of the iteration shards, measured at 1/2, 1/4, 1/8
    # of the points: rank 0's point block of a world of 2 / 4 / 8 as a problem of its own on this GPU (the same cameras, hence the
    # same reduced system's size and front tree): its eliminate + back-substitute stage times over the same iterations -- what
    # `sharded_ms / N` only extrapolates (does ba_eliminate_mfma still fill 256 CUs with an eighth of the pieces?)
    shard_ms = {}
    if world == 1 and not args.lean:
        for w_ in (2, 4, 8):
            loc_ = sharding.local_ba_problem(pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], pb["pts0"], 0, w_)
            p_ = bundle.BaProblem(200, len(loc_["pts"]), loc_["obs_cam"], loc_["obs_pt"], loc_["obs_xy"], ctx=ctx)
            p_.set_params(pb["cams0"], loc_["pts"], pb["focal0"])
            p_.iterate(BA_PRE)
            ctx.set_timing(True)
            t0_ = p_.last_timing()
            p_.iterate(args.steps)
            t1_ = p_.last_timing()
            ctx.set_timing(False)
            shard_ms[w_] = {"points": int(len(loc_["pts"])), "eliminate_ms": round(1e3 * (t1_["eliminate_s"] - t0_["eliminate_s"]) / args.steps, 4),
                            "backsub_ms": round(1e3 * (t1_["backsub_s"] - t0_["backsub_s"]) / args.steps, 4),
                            "reduced_solve_ms": round(1e3 * (t1_["solve_s"] - t0_["solve_s"]) / args.steps, 4)}
            p_.close()
        barrier()

    # ------------------------------------------------------------------ BA in batch mode: one problem per rank
    # The strong-scaled iteration above cannot scale (ba_amdahl: half of it is the replicated reduced solve).  What a
    # node of N GPUs does scale is N independent problems -- one whole cfg4 problem per rank, no collective: the rate a
    # reconstruction service sees.  Reported next to the strong-scaled figure, never instead of it.
    ba_batch = None
    if not args.no_ba_batch and not args.lean:
        import threading
        ba.close()
        # several problems in flight per GPU (a context + stream + host thread each): the reduced solve of an LM
        # iteration is latency-bound on a handful of CUs, so another problem's elimination fits beside it
        # (scripts/gpu_ba_two_problems.py: 1.5 x with two, 1.9 x with three, 2.0 x with four)
        n_conc = max(1, args.ba_batch)
        b_streams = [torch.cuda.Stream(dev) for _ in range(n_conc)]
        b_ctxs = [_lib.Context(local_rank, stream=s_.cuda_stream) for s_ in b_streams]
        b_probs, b_starts = [], []
        for k_, c_ in enumerate(b_ctxs):
            pb_r = synth.ba_problem(200, 100000, 10, seed=777 + 100 * rank + k_)        # every problem its own
            p_ = bundle.BaProblem(200, len(pb_r["pts0"]), pb_r["obs_cam"], pb_r["obs_pt"], pb_r["obs_xy"], ctx=c_)
            b_starts.append((pb_r["cams0"], pb_r["pts0"], pb_r["focal0"]))
            ba_restart(p_, b_starts[-1])
            b_probs.append(p_)
        barrier()
        # every problem solved from its start five times over, 2 K iterations each time (the first 40 at the defaults: a solve
        # that is still moving, see ba_restart above); setting the start again is part of the measured time
        b_rounds, b_per = 5, 2 * args.steps
        b_iters = b_rounds * b_per
        def b_work(q_, st_):
            for _ in range(b_rounds):
                q_.set_params(*st_)
                q_.iterate(b_per)
        ths = [threading.Thread(target=b_work, args=(p_, st_)) for p_, st_ in zip(b_probs, b_starts)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        barrier()
        t_bb = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([t_bb], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_bb = float(tt[0])
        ba_batch = {"workload": "independent cfg4 problems (200 cams / 100k pts / 1M obs each), no collective: "
                                f"{n_conc} in flight per GPU on streams of their own, each solved from its start {b_rounds} times, "
                                f"{b_per} LM iterations a time (the resets included in the time)",
                    "problems": world * n_conc, "problems_per_gpu": n_conc,
                    "iterations_per_s_total": round(world * n_conc * b_iters / t_bb, 2),
                    "iterations_per_s_per_problem": round(b_iters / t_bb, 2), "scaling": "weak"}
        for p_ in b_probs:
            p_.close()

    # ------------------------------------------------------------------ cfg5, strong scaling over the ranks
    cfg5 = None
    cfg5_sample = None
    if args.lean and args.cfg5_sample > 0:
        # the profiled sample of cfg5 (K2, knn_keyed_kernel): pairs evenly spaced over the 124 750, every one checked against the golden file
        o_imgs = synth.orb_image_set()
        o_all = synth.all_pairs(len(o_imgs))
        pick = np.unique(np.linspace(0, len(o_all) - 1, min(args.cfg5_sample, len(o_all))).astype(np.int64))
        o_dev = [torch.from_numpy(a).to(dev) for a in o_imgs]
        o_set = matcher.ImageSet(n_rows=[len(a) for a in o_imgs], dim=32, dtype=_lib.U8, norm=_lib.HAMMING, ctx=ctx)
        for i, t in enumerate(o_dev):
            o_set.adopt_device(i, t.data_ptr(), keepalive=t)
        o_plan = matcher.MatchPlan(o_set, o_all[pick])
        o_set.prepare_async()
        o_plan.run_async(0.8)
        barrier()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(3):
            o_plan.run_async(0.8)
        barrier()
        t_s5 = (time.perf_counter() - t0) / 3
        k2_kernel_s = o_plan.last_timing()["knn_kernel_s"]
        ctx.set_timing(False)
        o_cnt, o_q, o_t, o_d = o_plan.fetch()
        gold = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_checksums.npz"))
        ok5 = bool(np.array_equal(o_cnt, gold["counts"][pick]) and np.array_equal(synth.pair_checksums(o_cnt, o_q, o_t, o_d), gold["checksums"][pick]))
        assert ok5, "cfg5 sample: the device's match lists differ from the oracle's"
        ops5 = 2.0 * 5000 * 5000 * 256 * len(pick)
        cfg5_sample = {"pairs": int(len(pick)), "seconds_per_sweep": round(t_s5, 6), "kernel_ms": round(1e3 * k2_kernel_s, 4),
                       "achieved": round(ops5 / max(k2_kernel_s, 1e-12) / 1e12, 2), "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                       "frac": round(ops5 / max(k2_kernel_s, 1e-12) / 1e12 / I8_DENSE_PEAK_TOPS, 4), "oracle_match": ok5}
        o_plan.close()
        o_set.close()
        del o_dev
    if not args.no_cfg5:
        o_imgs = synth.orb_image_set()         # 500 x 5000 x 32 B, seeded: the same set on every rank
        o_pairs = synth.all_pairs(len(o_imgs))
        shards = sharding.shard_pairs(o_pairs, [len(a) for a in o_imgs], world)
        mine = o_pairs[shards[rank]]
        o_dev = [torch.from_numpy(a).to(dev) for a in o_imgs]
        o_set = matcher.ImageSet(n_rows=[len(a) for a in o_imgs], dim=32, dtype=_lib.U8, norm=_lib.HAMMING, ctx=ctx)
        for i, t in enumerate(o_dev):
            o_set.adopt_device(i, t.data_ptr(), keepalive=t)
        o_plan = matcher.MatchPlan(o_set, mine)
        o_set.prepare_async()
        o_plan.run_async(0.8)                  # warm-up
        barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            o_set.prepare_async()
            o_plan.run_async(0.8)
        barrier()
        t_cfg5 = (time.perf_counter() - t0) / 2
        ctx.set_timing(True)                   # (K2's own time: HIP events around the k-NN launches of one more sweep)
        o_plan.run_async(0.8)
        barrier()
        k2_kernel_s = o_plan.last_timing()["knn_kernel_s"]
        ctx.set_timing(False)
        o_cnt, o_q, o_t, o_d = o_plan.fetch()
        cs = synth.pair_checksums(o_cnt, o_q, o_t, o_d)
        with np.errstate(over="ignore"):
            part = np.array([np.sum(cs[:, 0], dtype=np.uint64), np.bitwise_xor.reduce(cs[:, 1]) if len(cs) else np.uint64(0),
                             np.uint64(int(o_cnt.sum()))], np.uint64)
        if world > 1:
            tt = torch.tensor([t_cfg5], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_cfg5 = float(tt[0])
            gathered = [None] * world
            dist.all_gather_object(gathered, part)       # result merge only: no collective on the data path
            with np.errstate(over="ignore"):
                part = np.array([np.sum([g[0] for g in gathered], dtype=np.uint64),
                                 np.bitwise_xor.reduce(np.array([g[1] for g in gathered], np.uint64)),
                                 np.sum([g[2] for g in gathered], dtype=np.uint64)], np.uint64)
        # every pair of this rank against the oracle's answer (tests/golden/cfg5_checksums.npz: the C restatement's match count
        # and [sum, xor] of the match mix per pair, generated once by tests/golden/make_cfg5_checksums.py -- data, not code)
        gold = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_checksums.npz"))
        idx = np.asarray(shards[rank])
        cfg5_ok = bool(np.array_equal(o_cnt, gold["counts"][idx]) and np.array_equal(cs, gold["checksums"][idx]))
        assert cfg5_ok, "cfg5: the device's match lists differ from the oracle's (per-pair counts / checksums)"
        # ... and the merged checksum must be the N = 1 value whatever N is (the golden file's own totals)
        with np.errstate(over="ignore"):
            n1 = [int(np.sum(gold["checksums"][:, 0], dtype=np.uint64)), int(np.bitwise_xor.reduce(gold["checksums"][:, 1])),
                  int(gold["counts"].sum())]
        assert [int(part[0]), int(part[1]), int(part[2])] == n1, f"cfg5: merged checksum {part} differs from the N = 1 value {n1}"
        cfg5 = {"workload": "cfg5: 500 img x 5000 ORB-256, all 124750 pairs, Hamming; pairs dealt over the ranks "
                            "(sharding.shard_pairs), descriptors resident on every rank",
                "pairs": int(len(o_pairs)), "pairs_this_rank": int(len(mine)), "seconds_per_sweep": round(t_cfg5, 5),
                "pairs_per_s": round(len(o_pairs) / t_cfg5, 1), "scaling": "strong", "matches": int(part[2]),
                "checksum": [int(part[0]), int(part[1])], "checksum_equals_n1": True,
                "oracle_checked_pairs": int(len(o_pairs)), "oracle_match": cfg5_ok,
                "oracle": "every pair's count and checksum equal the C restatement's (tests/golden/cfg5_checksums.npz); "
                          "on N > 1 every rank asserts its own share",
                "knn_kernel_ms_this_rank": round(1e3 * k2_kernel_s, 3)}
        o_plan.close()
        o_set.close()
        del o_dev

    # ------------------------------------------------------------------ findBestPair's scoring half (rank 0)
    score_leg = None
    if rank == 0 and not args.no_score:
        from sfm_danpipeline_amd import scoring
        from oracle import sfm_oracle_score as _score_ck     # the checker (C restatement of OpenCV's route), timed on a sample
        Kc = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])
        srng = np.random.default_rng(4321)
        sp = []
        for p_ in range(len(pairs)):                          # one two-view scene per pair of the cfg2 sweep
            sc_ = synth.two_view_scene(m=int(srng.integers(150, 900)), seed=5000 + p_, K=Kc, noise_px=0.4,
                                       outlier_frac=float(srng.uniform(0.1, 0.5)))
            sp.append((sc_["xy1"], sc_["xy2"]))
        scoring.score_essential(sp[:8], Kc, ctx=ctx)          # warm-up
        t0 = time.perf_counter()
        s_inl, _, s_its = scoring.score_essential(sp, Kc, ctx=ctx)
        t_score = time.perf_counter() - t0
        n_ck = 96
        t0 = time.perf_counter()
        ck = [_score_ck.find_essential_mat_ransac(a_, b_, Kc) for a_, b_ in sp[:n_ck]]
        t_ck = (time.perf_counter() - t0) / n_ck
        score_bad = [i for i in range(n_ck) if (int(s_inl[i]), int(s_its[i])) != (ck[i][0], ck[i][3])]
        assert not score_bad, f"scoring differs from its restatement on sampled pairs {score_bad}"
        score_leg = {"workload": "E-matrix RANSAC score (cv::findEssentialMat(RANSAC, 0.999, 1.0) inlier count) of 1225 pairs, "
                                 "150-900 matches each, 10-50 % wrong matches; host buffers in, counts out (whole call)",
                     "pairs": len(sp), "matches": int(sum(len(a_) for a_, _ in sp)), "seconds": round(t_score, 5),
                     "pairs_per_s": round(len(sp) / t_score, 1), "ransac_iterations_mean": round(float(s_its.mean()), 1),
                     "cpu_restatement_pairs_per_s": round(1.0 / t_ck, 2), "cpu_sample": f"{n_ck} pairs, C restatement of OpenCV "
                     "3.4.1's route (Durand-Kerner solvePoly), 1 thread; counts and iteration numbers of the sample equal the "
                     "device's", "flags": scoring.last_flags(ctx), "parity": "unpinned (no OpenCV in the image)"}

    # ------------------------------------------------------------------ SIFT front end, batched (rank 0)
    sift_leg = None
    if rank == 0 and not args.no_score:
        from sfm_danpipeline_amd import features
        frng = np.random.default_rng(99)
        yy, xx = np.mgrid[0:480, 0:640]
        frames = []
        for _ in range(4):                                      # textured 640 x 480 frames: ~1.5 k keypoints each
            img = np.zeros((480, 640))
            for _b in range(900):
                cx, cy, sg, am = frng.uniform(0, 640), frng.uniform(0, 480), frng.uniform(1.2, 5), frng.uniform(30, 160)
                x0, x1, y0, y1 = int(max(cx - 4 * sg, 0)), int(min(cx + 4 * sg + 1, 640)), int(max(cy - 4 * sg, 0)), int(min(cy + 4 * sg + 1, 480))
                img[y0:y1, x0:x1] += am * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * sg * sg))
            frames.append(np.clip(img + frng.normal(0, 2.0, img.shape), 0, 255).astype(np.uint8))
        frames = [frames[i % 4] for i in range(32)]
        features.sift_batch(frames, ctx=ctx)                    # warm-up: worker contexts, scratch blocks
        t0 = time.perf_counter()
        outb = features.sift_batch(frames, ctx=ctx)
        t_batch = time.perf_counter() - t0
        t0 = time.perf_counter()
        for fr in frames[:8]:
            features.sift_detect_and_compute(fr, ctx=ctx)
        t_one = (time.perf_counter() - t0) / 8
        sift_leg = {"workload": "SIFT(0, 3, 0.04, 10, 1.6) detectAndCompute of 32 frames of 640 x 480 (synthetic texture), "
                                "sfmhip_sift_batch: images in flight on 8 worker streams, descriptor rows left in HBM",
                    "images": len(frames), "keypoints_mean": round(float(np.mean([len(k) for k, _ in outb])), 1),
                    "images_per_s_batched": round(len(frames) / t_batch, 1), "images_per_s_one_at_a_time": round(1.0 / t_one, 1)}
        del outb

    # max over ranks
    if world > 1:
        tt = torch.tensor([t_match, t_ba], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_match, t_ba = float(tt[0]), float(tt[1])
        tot = torch.tensor([float(counts.sum())], device=dev, dtype=torch.float64)
        dist.all_reduce(tot)
        total_matches = int(tot[0])
    else:
        total_matches = int(counts.sum())

    pairs_per_s = world * len(pairs) * args.steps / t_match
    ba_it_per_s = args.steps / t_ba
    ms_match = 1e3 * t_match / args.steps
    ms_ba = 1e3 * t_ba / args.steps

    # ------------------------------------------------------------------ rooflines
    # HBM traffic per launch comes from rocprofv3 PMC passes of this same command (FETCH_SIZE /
    # WRITE_SIZE in separate passes, gfx950 correction 2x on FETCH_SIZE), condensed by
    # scripts/pmc_summary.py and committed under profiles/: bench.py cannot collect counters itself.
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            traffic = json.load(fh)
    except (OSError, ValueError):
        pass

    def hbm_bytes(key):
        v = traffic.get(key, {}).get("hbm_bytes_per_launch")
        return int(v) if v is not None else None
    # (said on the line: the counters are the builder's box's, read from the repository -- this run measured times, not bytes)
    traffic_source = ("profiles/traffic.json: rocprofv3 PMC passes of `bench.py --lean` on the builder's box (" +
                      str(traffic.get("_source", "round-tagged files under profiles/")) + "), not measured in this run") if traffic else None

    ops_per_pair = 2.0 * n_feat * n_feat * dim                     # 1.024 GOP per cfg2 pair
    knn_tops = ops_per_pair * len(pairs) / knn_s / 1e12
    # secondary limiter: the epilogue, 19 VALU lane-ops per 16 distances
    # (v_max3_i32 slot maxima over tile pairs + a max3 tree per tile) against 256 CU x 128 lanes x 2.4 GHz
    valu_tlops = (19.0 / 16.0) * n_feat * n_feat * len(pairs) / knn_s / 1e12
    roofline = {"kernel": "knn_kernel<KS=4,L2,NU=2,SR=256,NW=4> (i8 MFMA 32x32x32, train norm through the C operand, "
                          "value-only slot / tile-maximum epilogue, exact resolve of the candidates)", "bound": "mfma",
                "achieved": round(knn_tops, 2), "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                "frac": round(knn_tops / I8_DENSE_PEAK_TOPS, 4), "traffic": hbm_bytes("knn_kernel"), "traffic_source": traffic_source,
                "sustained_peak": round(sustained_peak_ops / 1e12, 1), "sustained_peak_clock_ghz": round(sustained_peak_ghz, 3),
                "frac_of_sustained": round(knn_tops / max(sustained_peak_ops / 1e12, 1e-9), 4),
                "sweep_clock_ghz": round(sweep_clock_ghz, 3),
                "sustained_note": "sustained_peak = what this box held for bare v_mfma_i32_32x32x32_i8 on random operands, every "
                                  "SIMD busy, 50 ms, measured right before the timed region (csrc/probe.hip); sweep_clock_ghz = the "
                                  "shader clock while the sweep ran (s_memtime / s_memrealtime of a one-wave sampler on the "
                                  "second stream); frac stays against the nominal 5 POP/s",
                "f32_equivalent_frac": round(knn_tops / F32_MFMA_PEAK_TFLOPS, 3),
                "valu_epilogue": {"achieved": round(valu_tlops, 2), "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                                  "frac": round(valu_tlops / VALU_PEAK_TLANEOPS, 4),
                                  "note": "19 VALU ops per 32x32 tile = 1.19 per distance (v_max3_i32 / v_med3_i32 issue "
                                          "at ~4 cycles per wave64 op on gfx950: scripts/ubench/epi_mix.hip); the MFMA "
                                          "pipe bounds the sweep, the per-sweep resolve the rest (DESIGN.md K1)"},
                "launch_ms": round(knn_s * 1e3, 4), "knn_stage_ms": round(knn_stage_s * 1e3, 4),
                "prepare_ms": round(prep_s * 1e3, 4), "compact_ms": round(comp_s * 1e3, 4),
                "note": "launch_ms = that one kernel (hipEvents on the library's stream around its launch, mean of the last "
                        "step of 4 rounds of back-to-back steps); knn_stage_ms adds the exact / fix-up kernels behind it"}
    n_obs_l, n_pt_l, n_cam = len(loc["obs_cam"]), len(loc["pts"]), 200
    red_dim = 6 * n_cam + 1
    ba_bytes = 3 * n_obs_l * 24 + 2 * n_pt_l * 24 + 2 * red_dim * red_dim * 8 + n_cam * 48   # section 8d
    ba_stream_s = (ba_t["eliminate_s"] + ba_t["backsub_s"]) / max(args.steps, 1)
    ba_gbs = ba_bytes / ba_stream_s / 1e9
    tr_ba = [hbm_bytes("ba_eliminate_mfma"), hbm_bytes("ba_backsub_runs") if hbm_bytes("ba_backsub_runs") is not None else hbm_bytes("ba_backsub")]
    if hbm_bytes("ba_gather_rows") is not None:      # (the default slab epilogue's second kernel; absent from older profiles)
        tr_ba.append(hbm_bytes("ba_gather_rows"))
    roofline_ba = {"kernels": "ba_eliminate_mfma + ba_gather_rows (one linearisation: F^T F + Schur correction, per-workgroup slabs "
                              "summed in a fixed order: S, g and the cost are bitwise reproducible) + ba_backsub_runs (per LM iteration, "
                              "this rank's shard)",
                   "bound": "hbm", "achieved": round(ba_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(ba_gbs / HBM_PEAK_GBS, 4),
                   "traffic": sum(tr_ba) if all(t is not None for t in tr_ba) else None, "traffic_source": traffic_source,
                   "eliminate_ms": round(1e3 * ba_t["eliminate_s"] / args.steps, 4),
                   "allreduce_ms": round(1e3 * ba_t["allreduce_s"] / args.steps, 4),
                   "reduced_solve_ms": round(1e3 * ba_t["solve_s"] / args.steps, 4),
                   "backsub_cost_ms": round(1e3 * ba_t["backsub_s"] / args.steps, 4),
                   "launches_per_iter": round(ba_t["launches"] / max(args.steps, 1), 1),
                   "reduced_layout": ba_layout,
                   "note": "instruction-issue / latency-bound in the per-point linearisation; the reduced system is factored "
                           "as a tree of fronts of the recursively dissected camera graph, one workgroup per front "
                           "(reduced_layout.front_tree: the dependency chain is chain_tiles tile steps of 32 columns instead "
                           "of dense_tiles), or, where no such tree exists, as chains + separator / dense; DESIGN.md section 3"}

    # K2 (knn_keyed_kernel, cfg5): times from this run (the full sweep of this rank, or the profiled sample), the counters from the
    # committed PMC passes of `bench.py --lean --cfg5-sample N`
    roofline_k2 = None
    k2_src = cfg5_sample if cfg5_sample is not None else None
    if cfg5 is not None:
        ops_k2 = 2.0 * 5000 * 5000 * 256 * cfg5["pairs_this_rank"]
        k2_src = {"pairs": cfg5["pairs_this_rank"], "kernel_ms": cfg5["knn_kernel_ms_this_rank"],
                  "achieved": round(ops_k2 / max(cfg5["knn_kernel_ms_this_rank"] * 1e-3, 1e-12) / 1e12, 2)}
    if k2_src is not None:
        pmc_k2 = traffic.get("knn_keyed_kernel", {})
        roofline_k2 = {"kernel": "knn_keyed_kernel<KS=8,SR=128> (Hamming: i8 MFMA 32x32x32 on +-8 rows, the accumulator IS the key "
                                 "1024 * distance + row-in-chunk (the query scaled to +-64); per-lane top-2 by v_med3_u32 + v_min_u32)", "bound": "mfma",
                       "achieved": k2_src["achieved"], "peak": I8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                       "frac": round(k2_src["achieved"] / I8_DENSE_PEAK_TOPS, 4), "pairs": k2_src["pairs"], "kernel_ms": k2_src["kernel_ms"],
                       "work": "2 * 5000 * 5000 * 256 OP per cfg5 pair",
                       "traffic": pmc_k2.get("hbm_bytes_per_launch"), "mfma_util_pct": pmc_k2.get("mfma_util_pct"),
                       "valu_per_mfma": pmc_k2.get("valu_per_mfma"),
                       "traffic_source": traffic_source}

    # ------------------------------------------------------------------ the whole drop-in call (rank 0, N = 1)
    # BundleAdjustment::adjustBundle (reference include/BundleAdjustment.h:19-20) is a static one-shot function the reference means to
    # call once per added view (src/Sfm.cpp:883-888, :996) and that builds its problem from the containers every time
    # (src/BundleAdjustment.cpp:50-110).  Every BA number above is per iteration of a problem that exists; this is what the caller
    # of the reference's signature sees: the C++ mirror's call in the reference's containers (csrc/host/ba_selftest.cpp, a
    # subprocess), four calls on the same structure -- the first builds the plan, the others find it kept -- and four on structures never seen.
    adjust_call = None
    if rank == 0 and world == 1 and not args.no_adjust_bundle:
        import subprocess
        import tempfile
        from sfm_danpipeline_amd import build as _build
        exe = _build.build_ba_demo()
        adjust_call = {}
        for tag_, (nc_, np_, k_) in (("cfg3", (50, 20000, 10)), ("cfg4", (200, 100000, 10))):
            pb_ = pb if tag_ == "cfg4" else synth.ba_problem(nc_, np_, k_, seed=777)
            with tempfile.TemporaryDirectory() as d_:
                synth.write_ba_containers(os.path.join(d_, "in.bin"), pb_, 960.0, 540.0)
                r_ = subprocess.run([exe, os.path.join(d_, "in.bin"), os.path.join(d_, "out.bin")], capture_output=True, text=True,
                                    env=dict(os.environ, SFM_BA_SELFTEST_CALLS="4", SFM_BA_SELFTEST_NEW_STRUCTURE="4"))
            assert r_.returncode == 0 and "failed" not in r_.stderr, r_.stderr[-2000:]
            recs = [json.loads(l_) for l_ in r_.stdout.splitlines() if l_.startswith("{")]
            its_ = [int(l_.split("iterations")[1].split(",")[0]) for l_ in r_.stdout.splitlines() if l_.startswith("Bundle adjustment:")]
            keys = ("pack_ms", "create_ms", "set_params_ms", "run_ms", "get_params_ms", "keep_ms", "writeback_ms", "total_ms", "plan_reused",
                    "front_plan_reused")
            # (the median call by total of the last three of each kind: the host's share moves by milliseconds from call to call)
            rep_ = sorted(recs[1:4], key=lambda r__: r__["total_ms"])[1]
            new_ = sorted(recs[-3:], key=lambda r__: r__["total_ms"])[1]
            adjust_call[tag_] = {"lm_iterations": its_[0], "first_call_of_the_process": {k: recs[0][k] for k in keys},
                                 "repeated_call": {k: rep_[k] for k in keys},
                                 "new_structure_call": {k: new_[k] for k in keys},
                                 "solve_ms": rep_["run_ms"],
                                 "total_over_solve_new_structure": round(new_["total_ms"] / max(new_["run_ms"], 1e-9), 2),
                                 "total_over_solve_repeated": round(rep_["total_ms"] / max(rep_["run_ms"], 1e-9), 2)}
        adjust_call["note"] = ("ms of host wall clock per stage of ONE BundleAdjustment::adjustBundle call through the C++ mirror in the "
                               "reference's containers: pack (std::map tracks -> flat arrays), create (sfmhip_ba_create: grouping, signature "
                               "sort, chunking, gather lists, allocations, uploads -- or, plan_reused, the comparison of the structure with the "
                               "kept problem's + the new measurements), run (the LM loop to CONVERGENCE; the first call also plans the front "
                               "tree), write-back; first_call_of_the_process = everything cold (HIP start-up excluded: the context exists; the code "
                               "objects load, the arena and the pinned block are made, every page is touched for the first time), repeated_call = the "
                               "median by total of calls two to four on the same structure (the kept plan), new_structure_call = a call on a structure never seen, in a warm "
                               "process, the median by total of the last three of four (the reference's per-view pattern: a point lost a view -- the runs, pieces and gather lists are rebuilt; the memory "
                               "and, the camera graph being the same, the front tree are not: front_plan_reused)")

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc   # the checker, timed as the reported host-CPU baseline ("port")
        orc.build()
        native = orc.use_native()               # -O3 -march=native, compiled on this host
        # threads = physical cores (SMT siblings share the FMA pipes), capped by what the box's cgroup grants: more
        # runnable threads than the CPU quota only get throttled
        cores, host_cores, quota = orc.usable_cores(), orc.physical_cores(), orc.cpu_quota()
        # matcher: the cache-blocked, SIMD, atomics-free organisation of cv::batchDistance under parallel_for_
        # (orc_match_many_blocked); a bounded sample sized to ~4 s, three repetitions (min / max on the line); the
        # per-pair checksums of (queryIdx, trainIdx, distance bits) must equal those of the GPU lists
        g_cnt, g_q, g_t, g_d = plan.fetch()
        g_cs = synth.pair_checksums(g_cnt, g_q, g_t, g_d)
        assert np.array_equal(g_cs, orc.pair_checksums(g_cnt, g_q, g_t, g_d))
        npairs, cpu_match_s = min(args.cpu_pairs, len(pairs)), 0.0
        while True:
            t0 = time.perf_counter()
            cpu_counts, cpu_cs = orc.match_many_blocked(imgs, pairs[:npairs], threads=cores)
            cpu_match_s = time.perf_counter() - t0
            if cpu_match_s >= 2.5 or npairs == len(pairs):
                break
            npairs = min(len(pairs), max(npairs * 2, int(npairs * 4.0 / max(cpu_match_s, 1e-3))))
        assert np.array_equal(cpu_counts, g_cnt[:npairs]), "CPU baseline and GPU match counts differ"
        assert np.array_equal(cpu_cs, g_cs[:npairs]), "CPU baseline and GPU match lists differ (per-pair checksums)"
        reps = [cpu_match_s]
        for _ in range(2):
            t0 = time.perf_counter()
            orc.match_many_blocked(imgs, pairs[:npairs], threads=cores)
            reps.append(time.perf_counter() - t0)
        cpu_match_s = float(np.median(reps))
        # matcher, one thread: the same routine on a few pairs
        n1 = 8
        t0 = time.perf_counter()
        c1, cs1 = orc.match_many_blocked(imgs, pairs[:n1], threads=1)
        cpu_match1_s = time.perf_counter() - t0
        assert np.array_equal(cs1, g_cs[:n1])
        # (the row-by-row restatement that the parity tests use as checker, for the record: one pair, one thread)
        t0 = time.perf_counter()
        _, csr = orc.match_many_checksum(imgs, pairs[:1], threads=1)
        cpu_rowwise1_s = time.perf_counter() - t0
        assert np.array_equal(csr, g_cs[:1])
        # the library the reference links, where the box has it (it does not in the build image nor on the GPU boxes: SURVEY 8d's
        # optional "library CPU path"): the reference's own call, src/Sfm.cpp:590-608, on the head of the same sample
        library = None
        try:
            import cv2
        except Exception:
            cv2 = None
        if cv2 is not None:
            bf, nlib, lib_counts = cv2.BFMatcher(cv2.NORM_L2), min(npairs, 64), []
            f32 = {i: np.asarray(imgs[i], dtype=np.float32) for ij in pairs[:nlib] for i in ij}
            t0 = time.perf_counter()
            for (i, j) in pairs[:nlib]:
                knn = bf.knnMatch(f32[i], f32[j], k=2)
                lib_counts.append(sum(1 for m in knn if len(m) == 2 and m[0].distance <= 0.8 * m[1].distance))
            lib_s = time.perf_counter() - t0
            library = {"what": "cv2.BFMatcher(NORM_L2).knnMatch(k=2) + 0.8 ratio test per pair (src/Sfm.cpp:590-608)",
                       "pairs": nlib, "pairs_per_s": round(nlib / lib_s, 3), "threads": int(cv2.getNumThreads()),
                       "opencv": cv2.__version__, "match_counts_equal_gpu": bool(lib_counts == [int(c) for c in g_cnt[:nlib]])}
        # BA: one thread (Ceres' default num_threads, nothing at src/BundleAdjustment.cpp:115-121 overrides it) and all cores
        ba_args = (pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
        # The reduced solve of the TIMED leg is a blocked, vectorised right-looking Cholesky (64-column panels, an 8 x 16 register
        # tile of AVX-512 FMAs in the trailing update) -- what Eigen's LLT inside Ceres 1.13 is.  The checker's row-by-row
        # factorisation (kept for the parity tests, and timed once for the record) understates a CPU several times over.
        cpu_ba_rowwise_s, _ = orc.ba_time_iterations(*ba_args, 2)
        orc.ba_set_blocked_cholesky(True)
        orc.ba_cholesky_stats(reset=True)
        cpu_ba_s, _ = orc.ba_time_iterations(*ba_args, args.cpu_ba_iters)
        chol_flops, chol_s = orc.ba_cholesky_stats(reset=True)
        ba_threads = min(cores, 16)            # (more only adds private copies of the 11.5 MB reduced system to sum up)
        orc.ba_set_threads(ba_threads)
        cpu_ba_all_s, _ = orc.ba_time_iterations(*ba_args, args.cpu_ba_iters)
        orc.ba_set_threads(1)
        orc.ba_set_blocked_cholesky(False)
        cpu_pairs_s = npairs / cpu_match_s
        cpu_ba_its = args.cpu_ba_iters / cpu_ba_s
        if adjust_call is not None:
            # (beside the whole call: what the CPU restatement's solve of cfg4 would take at that rate -- its set-up is a few ms)
            adjust_call["cfg4"]["cpu_oracle_solve_ms_extrapolated"] = round(1e3 * (adjust_call["cfg4"]["lm_iterations"] + 1) / cpu_ba_its, 1)
        cpu_step_ms = 1e3 * (len(pairs) / cpu_pairs_s + 1.0 / cpu_ba_its)
        cpu_baseline = {"value": round(cpu_pairs_s, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": f"first {npairs} of the 1225 cfg2 pairs, query-block-parallel / train-tiled / SIMD matcher on "
                                  f"{cores} threads (host: {host_cores} physical cores, cgroup CPU quota {quota}; median of 3 runs, {cpu_match_s:.1f} s each), lists "
                                  f"checksum-equal to the GPU's; {args.cpu_ba_iters} LM iterations of cfg4 on 1 thread like "
                                  f"Ceres' default ({cpu_ba_s:.1f} s)",
                        "march_native": bool(native), "logical_cpus": os.cpu_count(), "physical_cores": host_cores, "cgroup_cpu_quota": quota,
                        "matcher_runs_pairs_per_s": {"min": round(npairs / max(reps), 3), "max": round(npairs / min(reps), 3)},
                        "matcher_1_thread_pairs_per_s": round(n1 / cpu_match1_s, 4),
                        "matcher_speedup_over_1_thread": round((npairs / cpu_match_s) / (n1 / cpu_match1_s), 1),
                        "matcher_rowwise_checker_1_thread_pairs_per_s": round(1.0 / cpu_rowwise1_s, 4),
                        "ba_iterations_per_s": round(cpu_ba_its, 4), "ba_cores": 1,
                        "ba_cholesky_gflops": round(chol_flops / max(chol_s, 1e-9) / 1e9, 2),
                        "ba_cholesky": "blocked right-looking LLT, 64-column panels, AVX-512 FMA trailing update, 1 thread (Eigen's LLT "
                                       "inside Ceres 1.13's DENSE_SCHUR, reference src/BundleAdjustment.cpp:116)",
                        "ba_rowwise_checker_iterations_per_s": round(2 / cpu_ba_rowwise_s, 4),
                        "ba_threaded_iterations_per_s": round(args.cpu_ba_iters / cpu_ba_all_s, 4), "ba_threads": ba_threads,
                        "ms_per_step_extrapolated": round(cpu_step_ms, 1),
                        "gpu_over_cpu_step": round(cpu_step_ms / (ms_match + ms_ba), 1),
                        "library": library,   # (None: no OpenCV on this box)
                        "note": "a reported baseline, not the target: the matcher leg is organised as cv::batchDistance under "
                                "parallel_for_ (query rows in parallel, train tiles in cache, AVX FMA sum of squared "
                                "differences, no atomics); the BA leg restates Ceres' DENSE_SCHUR iteration with a blocked, vectorised "
                                "dense Cholesky (ba_cholesky_gflops), serial as Eigen's LLT is; the roofline fractions say what the GPU "
                                "kernels are worth"}

    if rank == 0:
        out = {
            "metric": "image-pairs matched/sec + BA iterations/sec (200 cams, 100k pts, 1M obs)",
            "value": round(pairs_per_s, 1), "unit": "pairs/s",
            "ba_iterations_per_s": round(ba_it_per_s, 2),
            "n_gpus": world, "rccl_ranks": ranks_seen,
            "collective": {"backend": ("nccl (RCCL)" if backend == "nccl" else backend) if world > 1 else None,
                           "world_size": dist.get_world_size() if world > 1 else 1, "allreduce_of_ones": ranks_seen,
                           "launched_by": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "direct"},
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_match + ms_ba, 4), "ms_match_sweep": round(ms_match, 4),
            "ms_ba_iteration": round(ms_ba, 4),
            "methodology": "r6: BA regions = LM iterations 16 .. 15 + K of a fresh solve, independent of --warmup (r5: W + 21 .. W + 20 + K); "
                           "ba_batch includes set_params in its time; GPU_MAX_HW_QUEUES=8 for this process (ba_batch only); not comparable "
                           "with BENCH_r01-r04's BA rates, which ran on one long solve",
            "higher_is_better": True, "scaling": "weak", "ba_scaling": "strong",
            "vs_baseline": None, "dtype": "i8", "ba_dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2 all-pairs L2 knn-2 + ratio (50 img x 2000 SIFT-128, 1225 pairs/GPU) "
                                   "+ cfg4 BA LM iteration (200 cams / 100k pts / 1M obs)",
                       "pairs_per_gpu": int(len(pairs)), "matches_found": total_matches,
                       "ba_points_per_gpu": int(n_pt_l), "ba_obs_per_gpu": int(n_obs_l),
                       "ba_cost": [ba_sum.initial_cost, ba_sum.final_cost],
                       # (hand-offs of the front tree that ran out of their spin budget in the timed iterations and were repeated
                       # level by level -- a loaded device; 0 in every run measured: a line with another value timed something else)
                       "ba_spin_timeouts": int(ba_sum.spin_timeouts),
                       "ba_timed_iterations": [int(ba_pre.iterations) + 1, int(ba_sum.iterations)],
                       "ba_steps_accepted_in_timed_iterations": ba_accepted_timed,
                       "match_streams": N_STREAMS, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "parallelism": f"pairs x{world} (weak" + (f"; consecutive batches alternate between {N_STREAMS} HIP "
                                      "streams per GPU" if N_STREAMS > 1 else "") + f"), BA points/{world} + all-reduce"},
            "sustained": {"seconds": args.sustain_s, "pairs_per_s": round(sustained_pairs_s, 1),
                          "ba_iterations_per_s": round(sustained_ba_its, 2)},
            "value_host_visible": {"pairs_per_s": round(host_visible_pairs_s, 1),
                                   "frac_of_value": round(host_visible_pairs_s / max(pairs_per_s, 1e-9), 3),
                                   "stop_and_copy_pairs_per_s": round(host_copy_pairs_s, 1),
                                   "note": "every sweep's counts + {queryIdx, trainIdx, distance} lists in host memory: a "
                                           "second stream packs sweep n into a device buffer and copies it (DMA) into one of two "
                                           "pinned buffers while the next sweeps run (sfmhip_matchplan_pipeline / _fetch_wait); the host "
                                           "takes a plan's lists two of its runs behind the one it enqueues; stop_and_copy = "
                                           "sfmhip_matchplan_fetch after every sweep"},
            "cfg5_strong": cfg5, "find_best_pair_scoring": score_leg, "sift_front_end": sift_leg, "ba_batch": ba_batch,
            "adjust_bundle_call": adjust_call, "roofline_k2": roofline_k2, "cfg5_sample": cfg5_sample,
            "ba_amdahl": (lambda sh, rep, ar: {
                "sharded_ms": round(sh, 4), "replicated_ms": round(rep, 4), "allreduce_ms": round(ar, 4),
                # measured, not extrapolated: rank 0's point block of a world of 2 / 4 / 8 as a problem of its own on this GPU
                "shard_measured": shard_ms or None,
                "sharded_ms_at_8_measured": round(shard_ms[8]["eliminate_ms"] + shard_ms[8]["backsub_ms"], 4) if shard_ms else None,
                "bound_8gpu_speedup_measured": round((sh + rep) / (shard_ms[8]["eliminate_ms"] + shard_ms[8]["backsub_ms"] + rep + ar), 2) if shard_ms else None,
                # what sharding the reduced solve over ranks could reach: nothing below the dependency chain.  The front tree
                # already runs every front at once on one GPU (31 workgroups at cfg4); its time IS the leaf-to-root chain
                # (reduced_layout.front_tree.chain_tiles tile steps), which every subtree-per-rank split leaves whole and
                # lengthens by an exchange of the cut level's contribution tiles over xGMI.
                "replicated_ms_sharded": round(rep, 4),
                "bound_8gpu_speedup": round((sh + rep) / (sh / 8 + rep + ar), 2),
                "note": "linearise + eliminate + back-substitute shard with the points; the reduced solve runs on every "
                        "rank: as a front tree it is bound by its dependency chain, not by work, so splitting its subtrees over "
                        "ranks cannot shorten it (replicated_ms_sharded = replicated_ms: an estimate, unmeasured on hardware); "
                        "bound_8gpu_speedup = (sharded + replicated) / (sharded / 8 + replicated + allreduce): BA iterations/s "
                        "do not scale with N, independent BA problems do (ba_batch; DESIGN.md section 5)"})(
                1e3 * (ba_t["eliminate_s"] + ba_t["backsub_s"]) / args.steps, 1e3 * ba_t["solve_s"] / args.steps,
                1e3 * ba_t["allreduce_s"] / args.steps),
            "roofline": roofline, "roofline_ba": roofline_ba, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
