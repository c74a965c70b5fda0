#!/usr/bin/env python3
"""bench.py -- DDPG gradient updates/sec/GPU on synthetic Pendulum-shaped replay (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--agents A] [--updates-per-step U]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload: obs 3, act 1, shared/actor/critic layers 200/200/200,
batch 100, actor_lr 1e-3, critic_lr 1e-2, tau 0.01; every agent owns a device-resident replay of 1e6
synthetic Pendulum transitions (gamma_i = 0.99).  One GPU holds A independent agents (the reference's
INDEX axis: seeds/settings); one "step" = ONE launch of the fused kernel = U updates of every agent,
each update = device Philox sample of 100 distinct indices + gather + the full update_network
(target actor/critic, TD target, critic step, actor step, Polyak).  Inputs are resident in HBM before
the timed region.  value = (N * A * U * K) / max-over-ranks wall time.

Multi-GPU: ranks are independent (different seeds, no data-path collective) -> "scaling": "weak";
the only collective is an all-gather of one small per-rank result vector after the timed region
(the episodic-return all-gather of the 8-GPU INDEX sweep).

The JSON line also carries
  roofline      dominant kernel vs the fp32 matrix-core peak (the binding roof: 27.8 FLOP/B algorithmic
                intensity > 19.7 ridge), achieved = 73.2 MFLOP x updates per launch / HIP-event launch time
  roofline_hbm  the same launch priced in algorithmic HBM bytes (2,634,064 B per update) vs 8 TB/s
  cpu_baseline  the reference-structured CPU port (oracle/) timed on this host, rank 0 at N=1 only
  ddpg_separate (N=1 only) the `network: separate` variant of DDPG on the MFMA kernel
  sac, naf      (N=1 only) the same measurement for BASELINE configs[2] / [3]: SoftActorCritic (S=3 A=1 L=128) and
                NAF (S=8 A=2 L=200) fused update kernels, 256 agents x 1e6-record replays, each with its own
                flop_per_update and roofline fraction; the DDPG headline stays `value`
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

S, A_DIM, H, B = 3, 1, 200, 100
REPLAY_N = 10 ** 6
FLOP_PER_UPDATE = 73.2e6          # 366,000 MAC/sample * 2 * 100
BYTES_PER_UPDATE = 2634064.0      # params+Adam+target read+write, + 3,600 B gather
PEAK_FP32_MATRIX = 157.3e12       # MI355X_MICROARCH.md: v_mfma_f32_* dense peak = fp32 vector peak
PEAK_HBM = 8.0e12                 # MI355X_MICROARCH.md: HBM3E spec peak
# MAC counts per sample with every contraction done once (the Q hidden contraction
# of SAC is shared by Q(s,a) and Q(s,pi)); params x {theta, target, Adam m, v} read + written, + the gather
SAC_SHAPE = dict(S=3, A=1, L=128)
SAC_FLOP_PER_UPDATE = 168832 * 2 * 100.0
SAC_BYTES_PER_UPDATE = 51716 * 4 * 4 * 2 + 100 * (2 * 3 + 1 + 2) * 4.0
# ReverseKL (jsonfiles/agent/reverse_kl.json): batch 32 (the reference's default), N_param 64 -> K = 62 nodes.
# multiply-adds x 2: pi, V, V' forward + Q(s,a), Q(s,a_new) forward + Q at the B*K (state, node) pairs + the input and
# weight gradients of the three trained networks
KL_SHAPE = dict(S=3, A=1, L=200, B=32, N_PARAM=64)
def _kl_flop(B=32, K=62, S=3, L=200):
    fwd = lambda rows, k_in: 2 * rows * (k_in * L + L * L + L)
    return float(fwd(B, S) * 3 + fwd(B, S + 1) * 2 + fwd(B * K, S + 1) + 3 * 2 * B * (L * L + L)
                 + 3 * 2 * B * (L * L + (S + 1) * L + 2 * L))
KL_FLOP_PER_UPDATE = _kl_flop()
KL_BYTES_PER_UPDATE = 124004 * 4 * 4 * 2 + 32 * (2 * 3 + 1 + 2) * 4.0
NAF_SHAPE = dict(S=8, A=2, L=200)
NAF_FLOP_PER_UPDATE = 288600 * 2 * 100.0
NAF_BYTES_PER_UPDATE = 83406 * 4 * 4 * 2 + 100 * (2 * 8 + 2 + 2) * 4.0


def synthetic_pendulum_replay(n, seed=0):
    rng = np.random.RandomState(seed)
    th = rng.uniform(-np.pi, np.pi, n)
    thd = rng.uniform(-8.0, 8.0, n)
    a = rng.uniform(-2.0, 2.0, n)
    s = np.stack([np.cos(th), np.sin(th), thd], 1)
    r = -(th ** 2 + 0.1 * thd ** 2 + 0.001 * a ** 2)
    thd2 = thd + (-3 * 10.0 / 2 * np.sin(th + np.pi) + 3.0 * a) * 0.05
    th2 = th + thd2 * 0.05
    thd2 = np.clip(thd2, -8.0, 8.0)
    s2 = np.stack([np.cos(th2), np.sin(th2), thd2], 1)
    return (s.astype(np.float32), a[:, None].astype(np.float32), r.astype(np.float64), s2.astype(np.float32),
            np.full(n, 0.99, np.float64))


def synthetic_uniform_replay(n, sdim, adim, seed=0):
    """U-distributed replay of the same N for the SAC / NAF shapes"""
    rng = np.random.RandomState(seed)
    return (rng.uniform(-1, 1, (n, sdim)).astype(np.float32), rng.uniform(-1, 1, (n, adim)).astype(np.float32),
            rng.uniform(-16, 0, n).astype(np.float64), rng.uniform(-1, 1, (n, sdim)).astype(np.float32),
            np.full(n, 0.99, np.float64))


def _cpu_port_run(seconds, records):
    """Reference-structured CPU port on a bounded sample (one core): list-of-records replay + sample_n_k +
    5 np.array conversions + float64 TD glue + the C restatement of the 7-step update.  Returns (updates, s)."""
    from oracle.cpu_baseline import ListReplay, Transition
    from oracle.ddpg import DDPGOracle, Dims, init_params
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    s, a, r, s2, g = synthetic_pendulum_replay(records, 0)
    rep = ListReplay(records, 0)
    for i in range(records):
        rep.append(Transition(s[i].astype(np.float64), a[i].astype(np.float64), float(r[i]),
                              s2[i].astype(np.float64), 0.99))
    d = Dims(S, A_DIM, H, H, H)
    net = DDPGOracle(d, init_params(d, 0), 1e-3, 1e-2, 0.01, [-1, -1, -8], [1, 1, 8], [2.0])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        (bs, ba, br, bs2, bg), _ = rep.sample_batch(B)
        net.update(bs, ba, bs2, br, bg)
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(seconds=15.0, records=REPLAY_N, all_cores=True):
    """The CPU baseline beside the GPU number: (i) one core = the reference's own deployment unit
    (one INDEX per process), the primary figure; (ii) one independent process per host core (one INDEX each).
    Same workload as the GPU leg (a 1e6-record replay per process).  Runs BEFORE anything touches the GPU: the
    workers of (ii) are child processes."""
    n, dt = _cpu_port_run(seconds, records)
    out = {"value": n / dt, "unit": "gradient updates/s", "cores": 1, "kind": "port",
           "sample": "%d updates (%.1f s) on a %d-record list replay; oracle/ddpg_oracle.c + reference-"
                     "structured host loop; TF-1.15 itself cannot run here" % (n, dt, records)}
    if all_cores:
        import subprocess
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))     # a one-GPU box's CPU share is 16 cores
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--cpu-seconds", str(seconds),
               "--cpu-records", str(records)]
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
        rates = []
        deadline = time.time() + seconds + 180.0
        for pr in procs:
            try:
                o, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
                k, t = o.split()[-2:]
                rates.append(float(k) / float(t))
            except Exception:           # a worker that failed or overran is killed; the one-core figure stands
                pr.kill()
        out["all_cores"] = {"value": sum(rates) if rates else None, "cores": len(rates),
                            "sample": "%d independent processes (one INDEX each), %.0f s each" % (len(rates), seconds)}
    return out


PMC_FILES = ("r03_ddpg_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")


def pmc_traffic(n_agents, updates_per_launch, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, scripts/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes --
    profiles/r03_peaks.json calibrates the factor 2 for the k-loop's dword stream as well, so this is the figure, not
    an upper bound).  Counters cannot be read
    inside a timed run, so this is the measurement of the same command line taken when the kernel last changed
    (scaled per update when the launch length differs); returns (bytes per launch | None, source file | None)."""
    for name in PMC_FILES:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                m = json.load(f)
        except (OSError, ValueError):
            continue
        if kernel != "mfma" or m.get("agents") != n_agents:
            continue
        per_update = m.get("traffic_bytes_per_update_upper")
        if per_update is None and m.get("traffic_bytes_per_launch_upper") is not None:
            per_update = m["traffic_bytes_per_launch_upper"] / float(m["agents"] * m["updates_per_launch"])
        if per_update is None:
            continue
        return per_update * n_agents * updates_per_launch, "profiles/" + name
    return None, None


# ------------------------------------------------------------------------------------------------------------
# the timed region and its bookkeeping (shared by every workload; tests/test_distributed_cpu.py drives it with a
# stub population under gloo)
# ------------------------------------------------------------------------------------------------------------
def measure(pop, updates_per_step, steps, warmup, dist=None, device_sync=None):
    """`warmup` untimed steps, then EXACTLY `steps` timed steps bracketed by (population sync + device sync +
    barrier) on both sides.  Returns (this rank's wall seconds, HIP-event milliseconds on the population's stream)."""
    def sync_all():
        pop.sync()
        if device_sync is not None:
            device_sync()
        if dist is not None:
            dist.barrier()

    for _ in range(warmup):
        pop.update(updates_per_step)
    sync_all()
    t0 = time.perf_counter()
    pop.timer_begin()
    for _ in range(steps):
        pop.update(updates_per_step)
    ev_ms = pop.timer_end()
    sync_all()
    return time.perf_counter() - t0, ev_ms


def reduce_over_ranks(dt, result, dist=None, world=1, device="cpu"):
    """MAX of the wall time over ranks + the one collective of the path: an all-gather of each rank's small result
    vector (after the timed region).  Returns (dt_max, [result of rank 0, 1, ...])."""
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    res = torch.tensor([float(result)], dtype=torch.float64, device=device)
    if dist is None:
        return float(t.item()), [float(res.item())]
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [torch.zeros_like(res) for _ in range(world)]
    dist.all_gather(gathered, res)
    return float(t.item()), [float(g.item()) for g in gathered]


def aggregate_value(world, agents, updates_per_step, steps, dt_max):
    """whole-job updates/s: every rank ran agents * updates_per_step * steps updates in at most dt_max seconds"""
    return world * agents * updates_per_step * steps / dt_max


def measured_peaks():
    """roofs measured on a box of this pool by scripts/micro/peaks.hip (profiles/r03_peaks.json): quoted beside the spec
    peaks the fractions are priced against; None when the file is absent"""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_peaks.json")) as f:
            m = json.load(f)
        return {"fp32_matrix_tflops": m["mfma_f32_16x16x4_tflops"], "hbm_read_gbs": m["read_f4_gbs"],
                "hbm_copy_gbs": m["copy_f4_gbs"], "source": "profiles/r03_peaks.json"}
    except (OSError, ValueError, KeyError):
        return None


def roofline_record(flop_per_update, bytes_per_update, updates_per_launch, launch_s):
    ach_flops = flop_per_update * updates_per_launch / launch_s
    ach_bytes = bytes_per_update * updates_per_launch / launch_s
    roof = {"bound": "mfma", "achieved": ach_flops / 1e12, "peak": PEAK_FP32_MATRIX / 1e12, "unit": "TFLOP/s",
            "frac": ach_flops / PEAK_FP32_MATRIX, "kernel_ms_per_launch": launch_s * 1e3,
            "flop_per_update": flop_per_update}
    hbm = {"bound": "hbm", "achieved": ach_bytes / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
           "frac": ach_bytes / PEAK_HBM, "bytes_per_update": bytes_per_update}
    mp = measured_peaks()
    if mp:
        roof["peak_measured"] = mp["fp32_matrix_tflops"]
        roof["frac_of_measured"] = ach_flops / 1e12 / mp["fp32_matrix_tflops"]
        hbm["peak_measured"] = mp["hbm_read_gbs"]
        hbm["frac_of_measured"] = ach_bytes / 1e9 / mp["hbm_read_gbs"]
        roof["peak_measured_source"] = hbm["peak_measured_source"] = mp["source"]
    return roof, hbm


def _fill_from_host(pop, host, torch):
    dev = [torch.from_numpy(x).cuda() for x in host]
    torch.cuda.synchronize()
    pop.replay_fill_all_dev(host[0].shape[0], dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                            dev[3].data_ptr(), dev[4].data_ptr())
    pop.sync()
    del dev


def side_record(algo, NA, U, steps, warmup, torch, device):
    """BASELINE configs[2] / [3] (SAC-v1 / NAF) and the ReverseKL agent of SURVEY section 8(f) item 4: their fused update
    kernels on 256 agents x 1e6-record replays."""
    seeds = np.arange(NA, dtype=np.uint64) + 1
    if algo == "sac":
        from rlcontrol_amd.hip_sac import SACPopulation, init_params
        L = SAC_SHAPE["L"]
        pop = SACPopulation(NA, SAC_SHAPE["S"], SAC_SHAPE["A"], L, L, L, L, B, REPLAY_N, 0.01, -8.0, 8.0, 2.0, 1e-3, 1e-3,
                            0.2, seeds=seeds, device=device)
        for i in range(NA):
            pop.set_params(i, init_params(SAC_SHAPE["S"], SAC_SHAPE["A"], L, L, L, L, int(seeds[i])))
        host = synthetic_uniform_replay(REPLAY_N, SAC_SHAPE["S"], SAC_SHAPE["A"])
        flop, byts = SAC_FLOP_PER_UPDATE, SAC_BYTES_PER_UPDATE
        wl = "SoftActorCritic (SAC-v1) on synthetic replay (1e6 transitions/agent), obs=3 act=1 l1=l2=128 batch=100"
    elif algo == "ddpg_separate":
        # actor_network.py / critic_network.py as two networks (json key `network: separate`), on the MFMA
        # kernel since round 3; priced with the hydra network's 73.2 MFLOP (the second first layer adds < 1 %)
        from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params
        pop = DDPGPopulation(NA, S, A_DIM, H, H, H, B, REPLAY_N, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0], 1e-3, 1e-2,
                             seeds=seeds, device=device, separate_networks=True)
        for i in range(NA):
            pop.set_params(i, init_params(S, A_DIM, H, H, H, int(seeds[i]), "input_norm", True))
        host = synthetic_pendulum_replay(REPLAY_N, 0)
        flop, byts = FLOP_PER_UPDATE, BYTES_PER_UPDATE + 2 * (S * H + H) * 4 * 4 * 2.0
        wl = "DDPG with separate actor / critic networks on synthetic Pendulum-shaped replay, obs=3 act=1 200/200/200 batch=100"
    elif algo == "kl":
        from rlcontrol_amd.hip_kl import KLPopulation, init_params
        L = KL_SHAPE["L"]
        pop = KLPopulation("reverse", NA, KL_SHAPE["S"], KL_SHAPE["A"], L, L, L, L, KL_SHAPE["B"], REPLAY_N, 0.01, 2.0, 1e-3,
                           1e-3, 0.1, seeds=seeds, n_param=KL_SHAPE["N_PARAM"], device=device)
        for i in range(NA):
            pop.set_params(i, init_params(KL_SHAPE["S"], KL_SHAPE["A"], L, L, L, L, int(seeds[i])))
        host = synthetic_uniform_replay(REPLAY_N, KL_SHAPE["S"], KL_SHAPE["A"])
        flop, byts = KL_FLOP_PER_UPDATE, KL_BYTES_PER_UPDATE
        wl = ("ReverseKL (optim_type intg, N_param 64) on synthetic replay (1e6 transitions/agent), obs=3 act=1 "
              "four 200-wide layers batch=32: 1984 Q evaluations per update")
    else:
        from rlcontrol_amd.hip_naf import NAFPopulation, init_params
        L = NAF_SHAPE["L"]
        pop = NAFPopulation(NA, NAF_SHAPE["S"], NAF_SHAPE["A"], L, L, B, REPLAY_N, 0.01, -np.ones(8) * 10, np.ones(8) * 10,
                            np.ones(2), 1e-3, seeds=seeds, device=device)
        for i in range(NA):
            pop.set_params(i, init_params(NAF_SHAPE["S"], NAF_SHAPE["A"], L, L, int(seeds[i])))
        host = synthetic_uniform_replay(REPLAY_N, NAF_SHAPE["S"], NAF_SHAPE["A"])
        flop, byts = NAF_FLOP_PER_UPDATE, NAF_BYTES_PER_UPDATE
        wl = "NAF on synthetic replay (1e6 transitions/agent), obs=8 act=2 l1=l2=200 batch=100"
    _fill_from_host(pop, host, torch)
    dt, ev_ms = measure(pop, U, steps, warmup, None, torch.cuda.synchronize)
    roof, roof_hbm = roofline_record(flop, byts, NA * U, ev_ms * 1e-3 / steps)
    rec = {"value": aggregate_value(1, NA, U, steps, dt), "unit": "gradient updates/s", "workload": wl,
           "agents_per_gpu": NA, "updates_per_step": U, "steps": steps, "kernel": pop.kernel_in_use(),
           "ms_per_step": dt * 1e3 / steps, "updates_timed_per_agent": U * steps, "roofline": roof, "roofline_hbm": roof_hbm}
    pop.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--agents", type=int, default=256, help="independent agents per GPU (one per CU)")
    ap.add_argument("--updates-per-step", type=int, default=320,
                    help="updates of every agent per launch (320 x 20 steps = 6,400 timed updates per agent, > 2 s)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "mfma"])
    ap.add_argument("--split", type=int, default=1,
                    help="latency mode: every agent's minibatch over this many CUs (rlc_ddpg_set_split; use with --agents 1..32)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 run (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path where fewer GPUs than ranks exist: rank r uses GPU r %% n_gpus "
                         "(needs --backend gloo; the value is then NOT a scaling measurement and says so)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group at world size 1 too (one rank under torch.distributed.run): barrier, "
                         "MAX all-reduce and all-gather then go through RCCL -- the one-GPU rehearsal of --gpus 8")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-records", action="store_true", help="skip the SAC / NAF sub-records")
    ap.add_argument("--side-only", default="", choices=["", "sac", "naf", "kl", "ddpg_separate"],
                    help="profiling helper: run ONLY that sub-record (no DDPG headline) and print its JSON")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-records", type=int, default=REPLAY_N)
    ap.add_argument("--cpu-worker", action="store_true", help="internal: one process of the all-cores CPU baseline")
    args = ap.parse_args()
    if args.cpu_worker:
        n, dt = _cpu_port_run(args.cpu_seconds, args.cpu_records)
        print(n, dt)
        return

    rank = int(os.environ.get("RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_base = None
    if rank == 0 and world_env == 1 and not args.no_cpu_baseline:
        from __graft_entry__ import build_oracle
        build_oracle()
        cpu_base = cpu_baseline(args.cpu_seconds, args.cpu_records)   # before the GPU is initialised (child processes)

    import torch
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if args.share_gpu:
        if args.backend == "nccl":
            raise SystemExit("--share-gpu puts several ranks on one GPU: RCCL cannot do that, use --backend gloo")
        local_rank = local_rank % max(1, torch.cuda.device_count())      # device_count() does not initialise the GPU
    torch.cuda.set_device(local_rank)          # before any other GPU call; no re-exec anywhere in this file
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    import __graft_entry__ as entry
    if rank == 0 or world == 1:
        entry.build()
    if dist is not None:
        dist.barrier()
    if args.side_only:
        print(json.dumps(side_record(args.side_only, args.agents, args.updates_per_step, args.steps, args.warmup, torch,
                                     local_rank)))
        return
    from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params

    NA, U = args.agents, args.updates_per_step
    # rank r holds INDEX values r*NA .. r*NA+NA-1 of a seeds-only sweep: seed = index
    seeds = np.arange(rank * NA, (rank + 1) * NA, dtype=np.uint64) + 1
    pop = DDPGPopulation(NA, S, A_DIM, H, H, H, B, REPLAY_N, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0],
                         1e-3, 1e-2, seeds=seeds, device=local_rank)
    if args.kernel != "auto":
        pop.set_kernel(args.kernel)
    for i in range(NA):
        pop.set_params(i, init_params(S, A_DIM, H, H, H, int(seeds[i])))
    # synthetic replay -> HBM (torch is only the allocator/copy engine here), then into every agent's ring
    _fill_from_host(pop, synthetic_pendulum_replay(REPLAY_N, 0), torch)

    kernel = pop.kernel_in_use()
    if args.split > 1:
        pop.set_split(args.split)          # latency mode is part of what is timed
        kernel += "+split%d" % args.split
    dt, ev_ms = measure(pop, U, args.steps, args.warmup, dist, torch.cuda.synchronize)
    red_dev = "cuda" if (dist is None or args.backend == "nccl") else "cpu"
    dt_max, gathered = reduce_over_ranks(dt, float(np.mean(pop.last_tap(0, "q"))), dist, world, red_dev)
    pop.close()

    if rank == 0:
        traffic, traffic_source = pmc_traffic(NA, U, kernel)
        roof, roof_hbm = roofline_record(FLOP_PER_UPDATE, BYTES_PER_UPDATE, NA * U, ev_ms * 1e-3 / args.steps)
        roof["traffic"] = traffic
        roof["traffic_source"] = traffic_source      # a committed earlier profile of this command, not an in-run counter
        roof_hbm["traffic"] = traffic
        roof_hbm["traffic_source"] = traffic_source
        out = {
            "metric": "gradient updates/sec/GPU (batch=100), Pendulum-shaped DDPG",
            "value": aggregate_value(world, NA, U, args.steps, dt_max),
            "unit": "gradient updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DDPG on synthetic Pendulum-shaped replay (1e6 transitions/agent), obs=3 act=1 "
                                   "l1=l2=200 batch=100, fused HIP replay-sample+gather+update kernel",
                       "agents_per_gpu": NA, "updates_per_step": U, "kernel": kernel,
                       # batch 100 = six 16-row tiles + 4 rows: the MFMA kernels run their tail-of-four form
                       "kernel_tail_of_four": bool(kernel == "mfma" and os.environ.get("RLC_NO_TAIL4", "0") != "1"),
                       "updates_timed_per_agent": U * args.steps, "updates_warmup_per_agent": U * args.warmup,
                       "per_gpu_value": NA * U * args.steps / dt_max,
                       "parallelism": ("independent seeds x%d per rank, x%d ranks SHARING the GPUs present (rehearsal, not a "
                                       "scaling measurement)" if args.share_gpu else
                                       "independent seeds x%d per GPU, x%d GPUs") % (NA, world),
                       "per_rank_result": gathered},
            "roofline": roof,
            "roofline_hbm": roof_hbm,
        }
        if dist is not None:
            out["config"]["collectives"] = "%s: barrier x2, all_reduce(MAX) x1, all_gather x1 over %d rank(s)" % (
                dist.get_backend(), world)
        out["cpu_baseline"] = cpu_base
        if world == 1 and not args.no_side_records:
            # BASELINE configs[2], [3] on the same box, same agent count and replay size (extra keys; `value` is DDPG)
            # the same launch length as the headline: U x steps = 6,400 timed updates per agent by default
            for algo in ("sac", "naf", "kl", "ddpg_separate"):
                out[algo] = side_record(algo, NA, U, args.steps, args.warmup, torch, local_rank)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
