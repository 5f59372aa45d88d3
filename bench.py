#!/usr/bin/env python3
"""MPN forward throughput on MI355X (BASELINE.json metric: MPN forward edges/s on the 10M-edge mesh batch).

One "step" = one SupervisedMPN forward (K0-K10: embeddings, 10 message passing steps, decoder + readout) over
one synthetic disjoint batch of 100 triangulated 120x120 grids (N = 1,440,000, E = 9,984,200, L = 128), inputs
already resident in HBM, CSR built once per mesh before the timed region (reported separately).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Graph-wise data parallelism, no data-path collective.  With N > 1 ranks the default is what BASELINE config 5 asks for,
STRONG scaling: the ONE 100-graph batch is sharded across the ranks at graph boundaries (shards resident before the
timed region); the same run then also times the weak form (every rank its own 10M-edge batch) and reports it under
`weak_scaling`.  --scaling weak makes the weak form the headline instead.
--workload airfoil1 / cap500k: the regime AMBER's refinement loop runs in (one mesh, or one size-capped training
batch, per forward; latent 64, 20 MP steps): reports host wall-clock per model(batch) next to device time.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# torch and the library are imported by _load() AFTER the self-launch decision: the parent of a `--gpus N` run must not
# touch the GPU (it only starts the ranks, waits for them and passes their exit status on).
torch = csr_mod = _lib = engine = sharding = syn = SupervisedMPN = default_architecture = None


def _load():
    global torch, csr_mod, _lib, engine, sharding, syn, SupervisedMPN, default_architecture
    import torch as _torch

    from amber_amd import _lib as lib_, csr as csr_, engine as engine_, sharding as sharding_, synthetic as syn_
    from amber_amd.architecture.supervised_mpn import SupervisedMPN as model_
    from amber_amd.config import default_architecture as arch_

    torch, csr_mod, _lib, engine, sharding, syn = _torch, csr_, lib_, engine_, sharding_, syn_
    SupervisedMPN, default_architecture = model_, arch_


HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (same guide)
F32_MFMA_PEAK_TFLOPS = 157.3   # fp32-input MFMA = the fp32 vector rate (same guide)
RENDEZVOUS_TIMEOUT_S = 120.0   # a rank that cannot join the process group / pass the first barrier in this time exits non-zero


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graphs", type=int, default=None, help="graphs per batch (default 100 = BASELINE config 5; --train: 3)")
    ap.add_argument("--nx", type=int, default=None, help="grid vertices per side (default 120; --train: 144)")
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--latent", type=int, default=None, help="default 128 (--train: 64)")
    ap.add_argument("--mp-steps", type=int, default=None, help="default 10 (--train: 20)")
    ap.add_argument("--variant", choices=["grid", "hetero", "brick"], default="grid",
                    help="grid = BASELINE config 5 (the headline); hetero = grid sides ~ U{80..160} per graph; brick = '3D-like' "
                         "tetrahedralised 19^3 bricks, in-degree ~15 (SURVEY.md section 8d's variants; secondary lines)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: strong (default) = the one batch sharded over the ranks, weak = one batch per rank")
    ap.add_argument("--workload", choices=["config5", "airfoil1", "cap500k"], default="config5",
                    help="config5 = BASELINE config 5 (the headline); airfoil1 = ONE airfoil-sized mesh graph (144x144 grid: "
                         "N = 20,736, E = 144,002), latent 64, 20 MP steps (configs 2 / 4: one mesh per forward); cap500k = three "
                         "of them, the reference batcher's N + E <= 500k cap (config 3)")
    ap.add_argument("--precision", choices=["fp32", "f16x2", "f16"], default=None,
                    help="MLP contraction precision (default f16x2: fp32-grade, 1e-5; f16: reduced, 1e-2; fp32: exact fp32 MFMA)")
    ap.add_argument("--train", action="store_true",
                    help="secondary benchmark: one training step (forward with tape + backward + gradient all-reduce) on a "
                         "BASELINE config 3 shaped batch (airfoil: 3 graphs of ~20.7k nodes / 144k edges, latent 64, 20 MP steps) "
                         "unless --graphs/--nx/--ny/--latent/--mp-steps are given")
    ap.add_argument("--activation", choices=["leakyrelu", "relu", "tanh"], default="leakyrelu",
                    help="stack.mlp.activation_function (secondary lines: the headline is leakyrelu, the reference's default)")
    ap.add_argument("--residual", choices=["inner", "outer", "none"], default="inner", help="stack.residual_connections")
    ap.add_argument("--layer-norm", choices=["inner", "outer", "none"], default="inner", help="stack.layer_norm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-baseline", action="store_true",
                    help="also time the reference's formulation as PyTorch-ROCm eager ops on this GPU (the oracle's op sequence "
                         "on cuda tensors, full workload) and report it as `eager_gpu_baseline`; a baseline leg like cpu_baseline")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run the N > 1 code path with every rank on cuda:0 and the gloo backend (a one-GPU "
                         "box cannot host RCCL ranks); the printed value is NOT a scaling measurement")
    ap.add_argument("--init-dist", action="store_true",
                    help="initialise the process group (RCCL) even with one rank: exercises on a one-GPU box the RCCL start-up, "
                         "barriers and gathers that the N > 1 runs use (launch under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--cpu-graphs", type=int, default=10, help="graphs in the CPU-baseline sample (10 = 1M edges)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1 headline run: skip the `secondary` object (one-mesh forward, size-capped batch, config-3 training step)")
    ap.add_argument("--dry-run-rendezvous", action="store_true",
                    help="test hook for the self-launch path: the ranks only join a gloo group on the CPU, pass a barrier and an "
                         "all_gather, and rank 0 prints what it saw; no GPU call is made")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 and args.workload == "config5" and not args.train else "weak"
    small = {"airfoil1": 1, "cap500k": 3}.get(args.workload)
    defaults = dict(graphs=small or 3, nx=144, ny=144, latent=64, mp_steps=20) if (args.train or small) else \
        dict(graphs=100, nx=120, ny=120, latent=128, mp_steps=10)
    for k, v in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def host_cores():
    """(logical CPUs the OS shows, physical cores, CPUs this process may run on)."""
    logical = os.cpu_count() or 1
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or logical
    except Exception:  # noqa: BLE001
        physical = logical
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    return logical, physical, usable


def cpu_baseline(cfg, state_dict, args):
    """The reference CPU path (oracle = torch-CPU restatement of the reference forward) on the host cores,
    on a bounded sample of the same workload: `cpu_graphs` of the 120x120 graphs.  One warm-up forward (thread pool,
    first-touch allocation), then the median of three."""
    from oracle import mpn_oracle as orc

    logical, physical, usable = host_cores()
    threads = min(16, usable)  # the box's CPU share for one GPU is 16
    torch.set_num_threads(threads)
    g = syn.grid_batch(num_graphs=args.cpu_graphs, nx=args.nx, ny=args.ny)
    ocfg = dict(cfg, in_node_features=4, in_edge_features=2, embedding=None)
    e = int(g.edge_index.shape[1])
    times = []
    with torch.no_grad():
        for _ in range(4):
            t0 = time.perf_counter()
            orc.supervised_mpn_forward(state_dict, ocfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
            times.append(time.perf_counter() - t0)
    runs = sorted(times[1:])
    dt = runs[1]
    return {"value": e / dt, "unit": "edges/s", "cores": threads, "kind": "port",
            "host": {"os_cpu_count": logical, "physical_cores": physical, "usable_cpus": usable},
            "sample": f"{args.cpu_graphs} of the {args.nx}x{args.ny} graphs (E={e}), median of 3 forwards after 1 warm-up "
                      f"({', '.join(f'{t:.2f}' for t in times)} s), torch CPU fp32, {threads} threads"}


def eager_gpu_baseline(cfg, state_dict, g, dev):
    """What the reference's own formulation costs on this GPU: the oracle's op-for-op restatement (cat -> Linear ->
    activation -> residual -> LayerNorm -> scatter_add mean ...) as PyTorch-ROCm eager kernels on the full workload.
    A baseline leg beside cpu_baseline (SURVEY.md section 8d asks for it); never the thing shipped."""
    from oracle import mpn_oracle as orc

    sd = {k: v.to(dev) for k, v in state_dict.items()}
    ocfg = dict(cfg, in_node_features=4, in_edge_features=2, embedding=None)
    x, ea, ei, mask = g.x.to(dev), g.edge_attr.to(dev), g.edge_index.to(dev), g.mask_output.to(dev)
    e = int(ei.shape[1])
    times = []
    with torch.no_grad():
        for _ in range(3):  # first pass = warm-up
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            out = orc.supervised_mpn_forward(sd, ocfg, x, ea, ei, mask)
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    peak = torch.cuda.max_memory_allocated(dev) / 2**30
    del out, sd, x, ea, ei, mask
    torch.cuda.empty_cache()
    return {"value": e / dt, "unit": "edges/s", "kind": "port", "ms_per_forward": dt * 1e3,
            "sample": f"full workload (E={e}), best of 2 after 1 warm-up, torch {torch.__version__} eager fp32 on cuda:0, "
                      f"peak allocated {peak:.0f} GiB"}


def measured_traffic(latent: int, edges: int):
    """(bytes per edge-kernel launch from the PMC passes | None, note).  The counters were collected on the headline
    workload (config 5: latent 128, E = 9,984,200); any other workload reports None."""
    if latent != 128 or edges != 9984200:
        return None, "PMC traffic is measured on the headline workload only (tools/profile_round.sh)"
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, "no profiles/traffic.json"
    try:
        t = json.load(open(tpath))
    except Exception as exc:  # noqa: BLE001
        return None, f"profiles/traffic.json unreadable: {exc}"
    if t.get("csrc_hash") != _lib.csrc_hash():
        return None, f"profiles/traffic.json was taken on kernel sources {t.get('csrc_hash')}, this build is {_lib.csrc_hash()}: re-run tools/profile_round.sh"
    return t.get("edge_kernel_hbm_bytes_per_launch"), t.get("note")


def needs_self_launch(args) -> bool:
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment)."""
    return args.gpus > 1 and "WORLD_SIZE" not in os.environ


def self_launch(args) -> int:
    """Start one fresh child process per rank (this same script, same arguments, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    as torch.distributed.run would set them), wait for all of them and return non-zero if any failed.  The parent has made
    no GPU call and imported neither torch nor the library; the children are ordinary new processes (no re-exec of a process
    that touched the GPU).  Rank 0's JSON line reaches stdout directly (the children inherit it)."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AMBER_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, first_failure = 0, None
    alive = list(range(len(procs)))
    while alive:
        time.sleep(0.2)
        for k in list(alive):
            r = procs[k].poll()
            if r is None:
                continue
            alive.remove(k)
            if r != 0 and rc == 0:
                rc, first_failure = (r if r > 0 else 1), time.monotonic()
                print(f"bench.py: rank {k} exited with status {r}", file=sys.stderr, flush=True)
        # a failed rank leaves the others waiting in a collective: give them a moment to notice, then end them (exact PIDs)
        if first_failure is not None and alive and time.monotonic() - first_failure > 20.0:
            for k in alive:
                procs[k].terminate()
            deadline = time.monotonic() + 10.0
            for k in alive:
                try:
                    procs[k].wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[k].kill()
            alive = []
    return rc


class _Watchdog:
    """Exit non-zero (with every thread's stack on stderr) instead of hanging when a start-up phase - joining the process
    group, the first barrier - takes longer than RENDEZVOUS_TIMEOUT_S."""

    def __init__(self, what: str, seconds: float = RENDEZVOUS_TIMEOUT_S):
        self.what, self.seconds = what, seconds

    def __enter__(self):
        import faulthandler

        print(f"bench.py[rank {os.environ.get('RANK', 0)}]: {self.what} (watchdog {self.seconds:.0f} s)", file=sys.stderr, flush=True)
        faulthandler.dump_traceback_later(self.seconds, exit=True)
        return self

    def __exit__(self, *exc):
        import faulthandler

        faulthandler.cancel_dump_traceback_later()
        return False


def init_ranks(args):
    """(rank, world, device, dist-or-None): one process per GPU, RCCL (backend "nccl") over xGMI."""
    import datetime

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    dev = None
    if not args.dry_run_rendezvous:
        if not args.rehearse_on_one_gpu and local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
        dev = torch.device("cuda", 0 if args.rehearse_on_one_gpu else local_rank)
        torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.init_dist:
        import torch.distributed as dist

        limit = datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S)
        with _Watchdog("joining the process group"):
            if args.rehearse_on_one_gpu or args.dry_run_rendezvous:
                dist.init_process_group("gloo", timeout=limit)
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))
        with _Watchdog("first barrier"):
            dist.barrier()
            if dev is not None:
                torch.cuda.synchronize(dev)
    return rank, world, dev, dist


def dry_run_rendezvous(args):
    """The self-launch path without a GPU: every rank joins, passes a barrier and contributes its rank to an all_gather."""
    rank, world, _, dist = init_ranks(args)
    seen = [0]
    if dist is not None:
        outs = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(outs, torch.tensor([rank], dtype=torch.int64))
        seen = [int(o.item()) for o in outs]
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run_rendezvous": True, "world_size": world, "ranks_seen": seen,
                          "backend": dist.get_backend() if dist is not None else None,
                          "self_launched": bool(os.environ.get("AMBER_BENCH_SELF_LAUNCHED"))}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def gather_floats(vals, dist, dev):
    """[[vals of rank 0], [vals of rank 1], ...] on every rank.  Host tensors under gloo (one-GPU rehearsals): gloo's own
    staging of device tensors is what hung a rehearsal in round 1; RCCL takes device tensors."""
    if dist is None:
        return [list(vals)]
    comm = torch.device("cpu") if dist.get_backend() == "gloo" else dev
    mine = torch.tensor(list(vals), dtype=torch.float64, device=comm)
    outs = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, mine)
    return [o.cpu().tolist() for o in outs]


def timed_steps(step, args, dist, dev):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize pairs.  Returns (seconds, last output)."""
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    t_local = time.perf_counter() - t0  # this rank's own K steps (before waiting for the others)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, t_local, out


def train_steps(dev, dist, rank, *, graphs, nx, ny, latent, mp_steps, steps, warmup):
    """`steps` timed training steps (forward with tape, backward incl. the gradient all-reduce when there are ranks, Adam)
    after `warmup` untimed ones.  Returns (seconds, N, E, architecture config)."""
    torch.manual_seed(0)
    cfg = default_architecture(latent_dimension=latent, num_steps=mp_steps)
    model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).to(dev).train()
    g = syn.grid_batch(num_graphs=graphs, nx=nx, ny=ny, seed=1234 + rank).to(dev)
    n, e = int(g.x.shape[0]), int(g.edge_index.shape[1])
    g.amber_csr = csr_mod.build_csr(g.edge_index, n)
    target = torch.randn(n, 1, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(model(g), target)
        loss.backward()  # gradient all-reduce (RCCL, one flat buffer) happens inside when world > 1
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(loss)
    return elapsed, n, e, cfg


def train_bench(args):
    """Secondary metric (not BASELINE's headline): training-step throughput, edges/s of forward + backward."""
    rank, world, dev, dist = init_ranks(args)
    elapsed, n, e, cfg = train_steps(dev, dist, rank, graphs=args.graphs, nx=args.nx, ny=args.ny, latent=args.latent,
                                     mp_steps=args.mp_steps, steps=args.steps, warmup=args.warmup)
    per_rank = gather_floats([float(e) * args.steps, elapsed], dist, dev)
    if rank == 0:
        t = max(r[1] for r in per_rank)
        print(json.dumps({
            "metric": "mpn_train_step_edges_per_sec", "value": sum(r[0] for r in per_rank) / t, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": t / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "precision": engine.default_precision, "data": "synthetic",
            "config": {"workload": f"training step (forward with tape, backward, Adam) on {args.graphs} triangulated {args.nx}x{args.ny} "
                                   f"grids per GPU, N={n} E={e}, latent {args.latent}, {args.mp_steps} MP steps, edge_dropout {cfg['edge_dropout']} "
                                   f"with a fresh CSR per step for the kept edges (BASELINE config 3 shape by default)",
                       "parallelism": f"graph-wise dp{world}, one gradient all-reduce per step",
                       "backend": dist.get_backend() if dist is not None else None,
                       "backward": ("exact fp32 MFMA" if engine.default_precision == "fp32" or args.latent == 256
                                    or os.environ.get("AMBER_MPN_EXACT_BACKWARD") else
                                    "bf16x3 operands" + (", chain and weight gradients of a module fused"
                                                         if args.latent <= 64 and not os.environ.get("AMBER_MPN_NO_FUSED_BACKWARD")
                                                         else ""))}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def small_forward(dev, graphs, calls=50, warmup=10):
    """The regime AMBER's refinement loop runs in: `graphs` airfoil-sized mesh graphs (144x144 grid each: N = 20,736,
    E = 144,002) per forward at latent 64 / 20 MP steps (1 = one mesh per _predict, BASELINE configs 2 / 4; 3 = the
    reference batcher's N + E <= 500k cap, config 3).  Back-to-back calls, CSR cached per mesh."""
    torch.manual_seed(0)
    cfg = default_architecture(latent_dimension=64, num_steps=20)
    model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).eval().to(dev)
    g = syn.grid_batch(num_graphs=graphs, nx=144, ny=144).to(dev)
    n, e = int(g.x.shape[0]), int(g.edge_index.shape[1])
    g.amber_csr = csr_mod.build_csr(g.edge_index, n)
    with torch.no_grad():
        for _ in range(warmup):
            out = model(g)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(calls):
            out = model(g)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / calls * 1e3
        lat = []
        for _ in range(15):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            model(g)
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t0) * 1e3)
    assert torch.isfinite(out).all()
    desc = engine.make_desc(has_head=True, decoder_layers=2, decoder_activation="leakyrelu", decoder_norm=None, **model.mpn.desc_kwargs())
    pieces, _ = model.decoder_mlp.param_list(single_row=False)
    flat, packed = engine.PackedParams().get(desc, model.mpn.param_list() + pieces + [model.readout.weight, model.readout.bias])
    *_, kms, kcnt = engine.forward(desc, flat, packed, g.x, g.edge_attr, g.amber_csr, want_node=False, want_edge=False,
                                   out_index=None, n_out=n, timed=True)
    return {"N": n, "E": e, "latent": 64, "mp_steps": 20, "ms_per_forward_pipelined": ms, "edges_per_sec": e / (ms * 1e-3),
            "ms_per_forward_waited_median": sorted(lat)[len(lat) // 2], "launches_per_forward": int(sum(kcnt[:4])),
            "calls_timed": calls}


def secondary_figures(dev):
    """What the N = 1 headline run appends AFTER its timed region (a second or so of extra run time): the small-graph
    forwards and the config-3 training step, so the driver's record carries them too."""
    steps = 20
    elapsed, n, e, _ = train_steps(dev, None, 0, graphs=3, nx=144, ny=144, latent=64, mp_steps=20, steps=steps, warmup=5)
    return {"airfoil1": small_forward(dev, 1), "cap500k": small_forward(dev, 3),
            "train_config3": {"N": n, "E": e, "latent": 64, "mp_steps": 20, "ms_per_step": elapsed / steps * 1e3,
                              "edges_per_sec": e * steps / elapsed, "steps_timed": steps,
                              "what": "forward with tape + backward + Adam, edge_dropout 0.1 with a fresh CSR per step"}}


def main():
    args = parse()
    if needs_self_launch(args):
        sys.exit(self_launch(args))  # before torch / the library are imported: the parent never touches the GPU
    _load()
    if args.dry_run_rendezvous:
        return dry_run_rendezvous(args)
    if os.environ.get("AMBER_BENCH_WATCHDOG"):  # development: dump every thread's stack and exit if the run takes longer
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["AMBER_BENCH_WATCHDOG"]), exit=True)
    if args.precision:
        engine.default_precision = args.precision
    if args.train:
        return train_bench(args)
    rank, world, dev, dist = init_ranks(args)

    # ---- network: BASELINE config 5 (latent 128, 10 MP steps, inner residual + inner LN, leakyrelu) ----
    torch.manual_seed(0)
    cfg = default_architecture(latent_dimension=args.latent, num_steps=args.mp_steps)
    default_modes = (args.activation, args.residual, args.layer_norm) == ("leakyrelu", "inner", "inner")
    if not default_modes:  # secondary lines: the other modes mpn.yaml can express
        cfg["stack"]["mlp"]["activation_function"] = args.activation
        cfg["stack"]["residual_connections"] = None if args.residual == "none" else args.residual
        cfg["stack"]["layer_norm"] = None if args.layer_norm == "none" else args.layer_norm
    example = syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))
    model = SupervisedMPN(cfg, example).eval()
    state_dict = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)

    # ---- data: synthetic mesh-graph batch, resident in HBM ----
    def make_graph(seed):
        if args.variant == "brick":
            return syn.brick_batch(args.graphs, 19, 19, 19, seed=seed)
        return syn.grid_batch(num_graphs=args.graphs, nx=args.nx, ny=args.ny, seed=seed, hetero=args.variant == "hetero")

    def resident(g):
        gd = g.to(dev)
        nn = int(gd.x.shape[0])
        csr_mod.build_csr(gd.edge_index, nn)  # the process's first build carries one-off costs (code objects, allocator): not timed
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        gd.amber_csr = csr_mod.build_csr(gd.edge_index, nn)  # once per mesh, outside the timed region
        torch.cuda.synchronize(dev)
        return gd, nn, int(gd.edge_index.shape[1]), (time.perf_counter() - t0) * 1e3

    strong = args.scaling == "strong" and world > 1
    shard_info = None
    if strong:
        full = make_graph(1234)  # the ONE batch of BASELINE config 5, the same on every rank
        shard_info = sharding.shard_stats(full, world)
        g = sharding.shard_batch(full, rank, world)
        del full
    else:
        g = make_graph(1234 + rank)
    gd, n, e, csr_ms = resident(g)

    def step():
        with torch.no_grad():
            return model(gd)

    elapsed, t_local, out = timed_steps(step, args, dist, dev)
    assert torch.isfinite(out).all()
    per_rank = gather_floats([float(e) * args.steps, elapsed, t_local, float(n), float(e)], dist, dev)
    total_edges, t = sum(r[0] for r in per_rank), max(r[1] for r in per_rank)

    # N > 1: the other scaling form in the same run, as a secondary figure
    secondary = None
    if world > 1 and args.workload == "config5":
        if strong:
            del gd
            torch.cuda.empty_cache()
            gd, n2, e2, _ = resident(make_graph(1234 + rank))
        else:
            full = make_graph(1234)
            shard_info = sharding.shard_stats(full, world)
            del gd
            torch.cuda.empty_cache()
            gd, n2, e2, _ = resident(sharding.shard_batch(full, rank, world))
            del full
        el2, tl2, out2 = timed_steps(step, args, dist, dev)
        assert torch.isfinite(out2).all()
        pr2 = gather_floats([float(e2) * args.steps, el2, tl2], dist, dev)
        t2 = max(r[1] for r in pr2)
        secondary = {"scaling": "weak" if strong else "strong", "value": sum(r[0] for r in pr2) / t2, "unit": "edges/s",
                     "ms_per_step": t2 / args.steps * 1e3, "global_edges_per_step": sum(r[0] for r in pr2) / args.steps,
                     "rank_ms_per_step_min": min(r[2] for r in pr2) / args.steps * 1e3,
                     "rank_ms_per_step_max": max(r[2] for r in pr2) / args.steps * 1e3}
        if strong:  # back to the headline's shard for the per-kernel figures below
            del gd
            torch.cuda.empty_cache()
            gd, n, e, _ = resident(g)

    # ---- per-kernel device time, measured live with HIP events on the launch stream (rank 0) ----
    roofline = roofline_hbm = kernels = cpu = eager = extra = None
    if rank == 0:
        L, S = args.latent, args.mp_steps
        desc = engine.make_desc(has_head=True, decoder_layers=2, decoder_activation="leakyrelu", decoder_norm=None,
                                **model.mpn.desc_kwargs())
        pieces, _ = model.decoder_mlp.param_list(single_row=False)
        flat, packed = engine.PackedParams().get(desc, model.mpn.param_list() + pieces + [model.readout.weight, model.readout.bias])
        ws = engine.workspace(desc, n, e, dev)
        reps = max(1, min(3, args.steps))
        ms = [0.0] * _lib.NUM_KERNEL_CLASSES
        cnt = [0] * _lib.NUM_KERNEL_CLASSES
        for _ in range(reps):
            _, _, _, m, c = engine.forward(desc, flat, packed, gd.x, gd.edge_attr, gd.amber_csr, want_node=False, want_edge=False,
                                           out_index=None, n_out=n, ws=ws, timed=True)
            ms = [a + b for a, b in zip(ms, m)]
            cnt = [a + b for a, b in zip(cnt, c)]
        edge_ms = ms[1] / max(cnt[1], 1)   # average duration of one edge-kernel launch
        node_ms = ms[2] / max(cnt[2], 1)
        # classes 4 / 5 repeat the first edge launch (embedding fused: reads no e) and the last node launch (head fused)
        steady_edge_ms = (ms[1] - ms[4]) / max(cnt[1] - cnt[4], 1) if cnt[1] > cnt[4] else edge_ms
        steady_node_ms = (ms[2] - ms[5]) / max(cnt[2] - cnt[5], 1) if cnt[2] > cnt[5] else node_ms
        kernels = {"embed_ms_per_forward": ms[0] / reps, "edge_ms_per_launch": edge_ms, "node_ms_per_launch": node_ms,
                   "edge_ms_per_launch_steady_state": steady_edge_ms, "edge_ms_first_launch": ms[4] / max(cnt[4], 1),
                   "node_ms_per_launch_steady_state": steady_node_ms, "node_ms_last_launch": ms[5] / max(cnt[5], 1),
                   "edge_launches_per_forward": cnt[1] // reps, "node_launches_per_forward": cnt[2] // reps,
                   "device_ms_per_forward": sum(ms[:4]) / reps, "csr_build_ms_once_per_mesh": csr_ms}
        # Dominant kernel = edge_kernel_f2<L> (split-precision fp16x2 MFMA chain, fp32-grade).  Per launch
        # (stated in DESIGN.md):
        #   algorithmic bytes: E * (16 L + 8): read e, gather two node rows, write e', two int32 indices
        #   (the first launch computes e from the raw edge features and the last writes no e': 4 L E fewer bytes each)
        #   matrix work: E * 4 L^2 useful flops in the weight-split form (direct form E * 8 L^2); every product is
        #   carried out as three fp16 MFMAs, so the matrix pipe executes 3x that
        # With fp16 matrix rates the MFMA floor of a launch (~0.8 ms) lies below its HBM floor (2.5 ms): HBM binds.
        # `achieved` divides by the STEADY-STATE variant's launch time (the launches that read e, gather two node rows and write
        # e'): the first launch (no e read) is cheaper and would flatter the figure if averaged in.
        bytes_per_launch = e * (16.0 * L + 8.0)
        flops_useful = e * 4.0 * L * L
        ach_gbs = bytes_per_launch / (steady_edge_ms * 1e-3) / 1e9
        ach_tf_exec = 3.0 * flops_useful / (steady_edge_ms * 1e-3) / 1e12
        traffic, traffic_note = measured_traffic(L, e)
        split = engine.default_precision != "fp32" and L <= 128  # the split-precision kernels ran (else: exact fp32 MFMA)
        if split:
            roofline = {"kernel": f"edge_kernel_f2<{L}>", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                        "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": steady_edge_ms,
                        "avg_launch_ms_all_variants": edge_ms,
                        "traffic_gbs": (traffic / (steady_edge_ms * 1e-3) / 1e9) if traffic else None,
                        "traffic_frac_of_peak": (traffic / (steady_edge_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}
            roofline_hbm = {"kernel": f"edge_kernel_f2<{L}>", "bound": "mfma", "achieved": ach_tf_exec, "peak": F16_MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": ach_tf_exec / F16_MFMA_PEAK_TFLOPS,
                            "note": "executed fp16 MFMA flops = 3 x useful E*4L^2 (hi*hi, hi*lo, lo*hi); not the binding roof"}
        elif engine.default_precision != "fp32" and L == 256:
            # latent 256 outside the exact mode: the row-per-lane kernels on bf16x3 operands (six bf16 MFMAs a product,
            # matrices streamed once per workgroup through an LDS window); neither roof is close - the larger fraction is named
            ach_b3 = 6.0 * flops_useful / (steady_edge_ms * 1e-3) / 1e12
            roofline = {"kernel": f"edge_kernel<{L}, bf16x3>", "bound": "mfma", "achieved": ach_b3, "peak": F16_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach_b3 / F16_MFMA_PEAK_TFLOPS, "traffic": None,
                        "traffic_note": "executed bf16 MFMA flops = 6 x useful E*4L^2 (three bf16 pieces a float, six piece products)",
                        "avg_launch_ms": steady_edge_ms}
            roofline_hbm = {"kernel": f"edge_kernel<{L}, bf16x3>", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach_gbs / HBM_PEAK_GBS, "note": "algorithmic bytes E*(16L+8)"}
        else:
            # exact fp32 MFMA (v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate): the matrix pipe binds, not HBM
            ach_tf = flops_useful / (steady_edge_ms * 1e-3) / 1e12
            roofline = {"kernel": f"edge_kernel<{L}>", "bound": "mfma", "achieved": ach_tf, "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach_tf / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                        "traffic_note": "exact fp32 path: E*4L^2 fp32 MFMA flops per launch",
                        "avg_launch_ms": steady_edge_ms}
            roofline_hbm = {"kernel": f"edge_kernel<{L}>", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach_gbs / HBM_PEAK_GBS, "note": "algorithmic bytes E*(16L+8); not the binding roof of the exact path"}
        # host overhead: wall-clock of one model(batch) call that is waited for (what a refinement loop sees per mesh)
        lat = []
        for _ in range(max(5, min(20, args.steps))):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t0) * 1e3)
        lat.sort()
        kernels["wall_ms_per_call_waited_median"] = lat[len(lat) // 2]
        kernels["wall_ms_per_call_pipelined"] = per_rank[0][2] / args.steps * 1e3
        # what AMBER's refinement loop sees: every iteration brings a NEW mesh, so the destination sort is part of the call
        # (a fresh edge_index tensor per call: the per-mesh CSR cache cannot hit); warmed, unlike csr_build_ms_once_per_mesh,
        # which is the first CSR build of the process
        fresh = []
        for _ in range(max(5, min(20, args.steps))):
            g2 = type(gd)(x=gd.x, edge_attr=gd.edge_attr, edge_index=gd.edge_index.clone(), batch=gd.batch, mask_output=gd.mask_output)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            with torch.no_grad():
                model(g2)
            torch.cuda.synchronize(dev)
            fresh.append((time.perf_counter() - t0) * 1e3)
        fresh.sort()
        kernels["wall_ms_per_call_new_mesh_median"] = fresh[len(fresh) // 2]
        kernels["launches_per_forward"] = sum(cnt[:4]) // reps
        if world == 1 and not args.no_cpu_baseline and args.workload == "config5" and default_modes:
            cpu = cpu_baseline(cfg, state_dict, args)
        if world == 1 and args.eager_baseline:
            eager = eager_gpu_baseline(cfg, state_dict, g, dev)
        if world == 1 and args.workload == "config5" and args.variant == "grid" and default_modes and not args.no_secondary:
            del ws
            torch.cuda.empty_cache()
            extra = secondary_figures(dev)

    if rank == 0:
        value = total_edges / t
        bytes_fwd = e * (16 * args.latent + 8) * args.mp_steps + n * 8 * args.latent * args.mp_steps \
            + e * (4 * 2 + 4 * args.latent) + n * (4 * 4 + 4 * args.latent) + n * (4 * args.latent + 4)
        small = {"airfoil1": "configs 2 / 4 shape: ONE airfoil-sized mesh graph per forward (", "cap500k": "config 3 shape: one size-capped "
                 "batch (N + E <= 500k) per forward ("}.get(args.workload)
        what = {"grid": (small + f"{args.graphs} triangulated {args.nx}x{args.ny} grid(s))") if small else
                f"BASELINE config 5: disjoint batch of {args.graphs} triangulated {args.nx}x{args.ny} grids",
                "hetero": f"config 5 'hetero' variant: {args.graphs} triangulated grids with sides ~ U{{80..160}}",
                "brick": f"config 5 '3D-like' variant: {args.graphs} tetrahedralised 19x19x19 bricks (in-degree ~15)"}[args.variant]
        line = {
            "metric": "mpn_forward_edges_per_sec", "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": t / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "precision": engine.default_precision, "data": "synthetic",
            "config": {"workload": f"{what} per "
                                   f"{'GPU' if args.scaling == 'weak' else 'job'}, N={n} E={e} on rank 0, F_n=4 F_e=2, latent {args.latent}, "
                                   f"{args.mp_steps} MP steps, {args.residual} residual + {args.layer_norm} LayerNorm, {args.activation}, decoder + readout",
                       "global_edges_per_step": total_edges / args.steps, "parallelism": f"graph-wise dp{world}, no collective",
                       "backend": dist.get_backend() if dist is not None else None,
                       "world_size": dist.get_world_size() if dist is not None else 1},
            "edge_steps_per_sec": value * args.mp_steps,
            "hbm_roofline_frac_whole_forward": (bytes_fwd / (t / args.steps)) / (HBM_PEAK_GBS * 1e9) if world == 1 else None,
            "roofline": roofline, "roofline_other": roofline_hbm, "kernels": kernels, "cpu_baseline": cpu,
        }
        if world > 1:
            line["ranks"] = {"nodes": [int(r[3]) for r in per_rank], "edges": [int(r[4]) for r in per_rank],
                             "ms_per_step_min": min(r[2] for r in per_rank) / args.steps * 1e3,
                             "ms_per_step_max": max(r[2] for r in per_rank) / args.steps * 1e3}
            if shard_info is not None:
                line["ranks"]["partition_of_the_one_batch"] = shard_info  # graphs / nodes / edges per rank, imbalance %, ideal speed-up
            if secondary is not None:
                line[secondary["scaling"] + "_scaling"] = secondary
        if eager is not None:
            line["eager_gpu_baseline"] = eager
        if extra is not None:
            line["secondary"] = extra
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
