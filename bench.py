#!/usr/bin/env python3
"""Headline benchmark: training clips/sec of the RGB+Depth token-fusion step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one full training step of r3d_amd's FUTR on one batch of synthetic clips: forward + 3 losses + backward +
AdamW (+ gradient all-reduce over RCCL when N>1), dropout active (the reference's first-epoch state), inputs already
resident in HBM, fp32 end to end.  Workload at every N: BASELINE.json configs[1] (DARai RGB+Depth
futr_safuser_tokenfusion, batch 8 PER GPU, 16-frame clips, hidden 128) -> weak scaling.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(B=8, S=16, H=128, K=17, D=2048, P=224 * 224, Q=8, heads=8, n_dec=1, n_enc=2, lr=1e-3, wd=5e-3)
# BASELINE.json's other configurations at their per-GPU shapes: parity-test cases (tests/test_engine_gpu.py), selectable
# here with --config for profiling only -- the headline line is always cfg2
OTHER = dict(cfg3=dict(S=32), cfg4=dict(S=64, H=512), cfg5=dict(S=16, H=1024))
# algorithmic work of ONE per-GPU step ("Totals"): bytes and FLOPs
STEP_WORK = dict(cfg2=dict(bytes=346e6, flops=3.95e9), cfg3=dict(bytes=407e6, flops=7.84e9),
                 cfg4=dict(bytes=1.88e9, flops=84.1e9))
NAMES = dict(cfg2="DARai RGB+Depth futr_safuser_tokenfusion, batch=8 per GPU, 16-frame clips, hidden=128, n_class=17, "
                  "depth 224x224 (BASELINE.json configs[1])")


def make_inputs(c, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    pad = c["K"] + 1
    feats = torch.randn(c["B"], c["S"], c["D"], generator=g)
    depth = torch.rand(c["B"], c["S"], 1, 224, 224, generator=g)
    lab = torch.randint(0, c["K"] - 1, (c["B"], c["S"]), generator=g)
    lab[1::2, c["S"] - max(c["S"] // 8, 1):] = pad
    tgt = torch.randint(0, c["K"] - 1, (c["B"], c["Q"]), generator=g)
    dur = torch.rand(c["B"], c["Q"], generator=g) + 0.05
    dur = dur / dur.sum(1, keepdim=True)
    return [t.to(device) for t in (feats, depth, lab, dur, tgt)]


def build_model(c, device, variant="tokenfusion"):
    if variant == "bn":               # the BN-blend fuser (futr_safuser_batchnormalization), same workload shape
        from r3d_amd.model.futr_safuser_batchnormalization import FUTR
    else:
        from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    args = argparse.Namespace(input_dim=c["D"], seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    torch.manual_seed(1)
    m = FUTR(c["K"], c["H"], c["K"] + 1, device, args, n_query=c["Q"], n_head=c["heads"], num_encoder_layers=c["n_enc"],
             num_decoder_layers=c["n_dec"], depth_pixels=c["P"]).to(device)
    return m.train()


def time_kernel(fn, iters=20, warm=5, reps=5):
    """Average launch duration of fn's kernel: `iters` launches captured into a hipGraph (no host launch gaps between
    them), replayed `reps` times with HIP events around each replay on the launch stream; the median replay is reported
    (single replays scatter by +-10 % with the memory clock state right after the timed region)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # on torch's current stream,
        e0.record()                                                                           # where ops.* launch
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / iters * 1e-3)
    return sorted(times)[len(times) // 2]              # seconds per launch


def kernel_rooflines(eng, c):
    """Live HIP-event timing of the three heavy kernels with the step's real operands (after the timed region).  The labels
    name the kernels that run in the step (rocprofv3 names in profiles/r03_*_kernel_stats.csv)."""
    from r3d_amd import ops
    from r3d_amd._lib import GEMM_NT, GEMM_TN
    from r3d_amd.engine import DROP_P
    st, a, w = eng.last, eng.arena, eng.last["w"]
    N, H, P = c["B"] * c["S"], c["H"], c["P"]
    bf3 = eng.depth_prec == 1
    out = {}
    t = time_kernel(lambda: ops.gemm(GEMM_TN, w.d_dep_pre, st["x_dep"], a.g("depth_projection.weight"), ws=eng.ws,
                                     prec=eng.depth_prec))
    out["depth_projection_wgrad (" + ("wgrad_panel_bf3_kernel / gemm_bf3_tn_kernel: bf16x3 split, fp32 accumulate"
                                      if bf3 else "gemm_f32 TN") + ")"] = dict(
        seconds=t, flops=2.0 * N * P * H, bytes=4.0 * (N * P + N * H + H * P))

    def fwd():
        d = ops.gemm(GEMM_NT, st["x_dep"], a.p("depth_projection.weight"), w.dep_pre, ws=eng.ws, defer_reduce=True,
                     prec=eng.depth_prec)
        return d
    t = time_kernel(fwd)
    out["depth_projection_fwd (" + ("gemm_bf3_nt_kernel split-K: bf16x3 split, fp32 accumulate; timed alone -- in the step the "
                                    "same launch also carries the RGB embedding's 12 workgroups" if bf3
                                    else "gemm_f32 NT split-K") + ")"] = dict(
        seconds=t, flops=2.0 * N * P * H, bytes=4.0 * (N * P + H * P + N * H))
    lr_t, step_t = eng.lr_t, eng.step_t
    scratch = [a.params[:a.n_live].clone(), a.exp_avg.clone(), a.exp_avg_sq.clone()]
    if st["drop"]:
        # the variant the training step runs: AdamW + the next step's dropout masks in one launch
        pool, off = w.drop_pool.clone(), eng.drop_offset.clone()
        t = time_kernel(lambda: ops.adamw_flat_dropout(scratch[0], a.grads, scratch[1], scratch[2], lr_t, step_t, pool, DROP_P,
                                                       eng.drop_seed, off, weight_decay=c["wd"]))
        label = "adamw (adamw_dropout kernel: AdamW over the flat arenas + the next step's dropout masks)"
    else:
        t = time_kernel(lambda: ops.adamw_flat(scratch[0], a.grads, scratch[1], scratch[2], lr_t, step_t, weight_decay=c["wd"]))
        label = "adamw (adamw_kernel over the flat arenas)"
    out[label] = dict(seconds=t, flops=0.0, bytes=28.0 * a.n_live)
    return out


def erank_field(eng):
    """Effective rank of the last step's fused token matrix [N, H] (the quantity the metric's "effective-rank match"
    names): the HIP one-sided Jacobi (r3d_erank_jacobi / r3d_erank_blocked, timed with HIP events on the launch stream,
    median of 5) against torch.linalg.svdvals (LAPACK) on the CPU copy of the same matrix.  Outside the timed region.
    The reference has no SVD: svdvals on the same tokens IS the checker BASELINE.json's tolerance
    (+-0.5) refers to."""
    from r3d_amd import ops
    x = eng.last["w"].fused
    N, H = x.shape
    dev = x.device
    lds = ops.erank_fits(N, H)
    if lds:
        sigma, stats = torch.empty(1, H, device=dev), torch.empty(1, 4, device=dev)
        af = torch.empty(1, H, N, device=dev)
        run = lambda: ops.erank_jacobi(x, sigma, stats, af_t=af)             # noqa: E731
        kernel = "erank_jacobi_kernel (columns resident in one CU's LDS)"
    else:
        xx = x.t().contiguous() if N < H else x
        state = {}

        def run():
            state["r"] = ops.erank_blocked(xx)
        kernel = "erank_blk_round_kernel (two-level block Jacobi, columns in HBM/L2)"
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    st = (stats[0] if lds else state["r"][1]).cpu()
    sv = torch.linalg.svdvals(x.detach().cpu().double())
    pr = sv / sv.sum()
    pr = pr[pr > 0]
    want = float(torch.exp(-(pr * pr.log()).sum()))
    return dict(hip=float(st[0]), svdvals=want, abs_diff=abs(float(st[0]) - want), tolerance=0.5, sweeps=float(st[3]),
                us=sorted(ts)[2], matrix=[N, H], kernel=kernel, in_step=False,
                note="measured after the timed region on the last step's fused tokens; the headline step runs with "
                     "erank_weight = 0 (the reference's loss, SURVEY.md F1)")


PMC_TABLE = os.path.join("profiles", "r03_pmc_hbm.json")


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    in separate runs, gfx950 correction applied: tools/pmc_summary.py -> profiles/r03_pmc_hbm.json).  A profiler cannot
    run inside the timed process, so this is the figure of the committed profile of the same command (the line says so in
    roofline.traffic_source), or None."""
    path = os.path.join(ROOT, PMC_TABLE)
    key = "adamw" if kernel_label.startswith("adamw") else ("wgrad_panel_bf3" if "wgrad" in kernel_label else "gemm_bf3_nt")
    try:
        tab = json.load(open(path))
    except OSError:
        return None
    for k, v in tab.items():
        if key in k:
            return v["hbm_bytes"]
    return None


def cpu_baseline(c, budget_s=8.0):
    """The oracle (CPU restatement, 'port') timed on this box's host cores on a bounded sample of the same workload, at 16
    threads (one GPU's share of the host), 64 and every core available to the process; the best is `value`, all are stated."""
    from oracle import futr_oracle as O, synth
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    import argparse as ap
    from r3d_amd.model.futr_safuser_tokenfusion import FUTR
    args = ap.Namespace(input_dim=c["D"], seg=True, anticipate=True, max_pos_len=2000, input_type="i3d_transcript")
    m = FUTR(c["K"], c["H"], c["K"] + 1, torch.device("cpu"), args, n_query=c["Q"], n_head=c["heads"],
             num_encoder_layers=c["n_enc"], num_decoder_layers=c["n_dec"])
    params = {n: p.detach().clone() for n, p in m.named_parameters()}
    batch = [torch.from_numpy(x) for x in synth.make_batch(c["B"], c["S"], c["K"], c["K"] + 1, 1)]
    runs = []
    # every core of a 256-thread host is not a meaningful arm for a 4 GFLOP step: measured in the round-3 profile refresh,
    # 256 threads ran ONE step in 38 s (0.2 clips/s, oversubscribed intra-op pools) -- "all cores" is timed up to 128
    sizes = {max(1, min(16, avail)), max(1, min(64, avail))}
    if avail <= 128:
        sizes.add(avail)
    for cores in sorted(sizes):
        torch.set_num_threads(cores)
        tr = O.CpuTrainer({k: v.clone() for k, v in params.items()}, c["K"] + 1, c["heads"], c["n_dec"], c["lr"], c["wd"])
        tr.step(batch)                                                    # warm-up
        t0, n = time.perf_counter(), 0
        while True:
            tr.step(batch)
            n += 1
            if time.perf_counter() - t0 > budget_s or n >= 400:
                break
        dt = time.perf_counter() - t0
        runs.append(dict(cores=cores, value=c["B"] * n / dt, steps=n, seconds=dt))
    best = max(runs, key=lambda r: r["value"])
    return dict(value=best["value"], unit="clips/s", cores=best["cores"], kind="port", host_cores=os.cpu_count(),
                host_cores_available=avail, by_threads={str(r["cores"]): r["value"] for r in runs},
                all_cores_note=(None if avail <= 128 else f"{avail} threads not timed: 256 threads ran one step in 38 s "
                                                          "(0.2 clips/s) in profiles/r03_bench_cfg2.json's refresh"),
                sample="; ".join(f"{r['steps']} full CPU training steps (fwd+3 losses+autograd bwd+AdamW) of the same "
                                 f"B={c['B']},S={c['S']},H={c['H']} workload in {r['seconds']:.1f}s on {r['cores']} threads"
                                 for r in runs) + f"; torch {torch.__version__} CPU; box: {os.cpu_count()} cores, {avail} "
                                                  f"available to this process")


PROBE = dict(ran=False, rc=None)


def rccl_probe_child(a, timeout_s=300):
    """Multi-GPU only, before this process touches the GPU: run the RCCL-on-the-launch-stream step (own communicators,
    collectives captured into the step's hipGraph) for a few steps in a CHILD process per rank.  A crash or hang there --
    which no try/except in this process could survive -- costs the child, and the bench falls back to the
    torch.distributed exchanges.  Returns True if the child finished cleanly."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
    port = int(env.get("MASTER_PORT", "29533"))
    env["MASTER_PORT"] = str(20000 + (port + 7919) % 40000)      # the children rendezvous among themselves
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, os.path.abspath(__file__), "--probe-child", "--gpus", str(a.gpus), "--steps", "10", "--warmup",
           "2", "--no-cpu-baseline", "--config", a.config, "--variant", a.variant]
    for flag, on in (("--force-dist", a.force_dist), ("--replicated-depth", a.replicated_depth),
                     ("--frame-major-input", a.frame_major_input), ("--separate-tail", a.separate_tail),
                     ("--fused-adamw", a.fused_adamw), ("--eval-dropout-off", a.eval_dropout_off)):
        if on:
            cmd.append(flag)
    try:
        p = subprocess.Popen(cmd, env=env, stdout=sys.stderr, stderr=sys.stderr)
    except OSError as e:
        print(f"[bench] probe child could not start: {e}", file=sys.stderr, flush=True)
        return False
    PROBE["ran"] = True
    try:
        rc = p.wait(timeout=timeout_s)
        PROBE["rc"] = rc
        if rc != 0:
            print(f"[bench] probe child exited with {rc}; using torch.distributed exchanges", file=sys.stderr, flush=True)
        return rc == 0
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
        PROBE["rc"] = "timeout"
        print("[bench] probe child timed out; using torch.distributed exchanges", file=sys.stderr, flush=True)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step from Python instead of replaying hipGraphs")
    ap.add_argument("--steps-per-graph", type=int, default=8,
                    help="one GPU: consecutive training steps captured into one hipGraph (consecutive graph launches leave "
                         "the GPU idle for ~8 us; a loader that stages this many batches ahead amortises it).  The timed "
                         "region still runs EXACTLY --steps steps (a remainder replays a one-step graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-dropout-off", action="store_true", help="bench the eval()-state step (dropout off)")
    ap.add_argument("--side-stream", action="store_true", help="run the query self-attention branch on a second HIP stream")
    ap.add_argument("--no-auto-side-stream", action="store_true",
                    help="hidden >= 512: keep the query self-attention branch on the launch stream (by default it is forked "
                         "there on one rank; A/B)")
    ap.add_argument("--fused-decoder", action="store_true", help="run the decoder layer as decoder.hip (one workgroup per "
                                                                 "clip) instead of composed GEMM / attention / LN launches")
    ap.add_argument("--fused-adamw", action="store_true",
                    help="apply AdamW to depth_projection.weight in its weight-gradient GEMM's epilogue instead of writing "
                         "the gradient and updating it in the flat AdamW launch (measured neutral: 376 vs 374 us/step)")
    ap.add_argument("--replicated-depth", action="store_true",
                    help="N>1: keep depth_projection replicated and all-reduce its 25.7 MB gradient (plain data parallel) "
                         "instead of sharding it over pixels")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 = the headline workload; the others (per-GPU shapes of BASELINE.json configs[2..4]) are "
                         "for profiling")
    ap.add_argument("--variant", default="tokenfusion", choices=["tokenfusion", "bn"],
                    help="tokenfusion = the headline model (BASELINE.json); bn = the BN-blend fuser variant, profiling only")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank control flow on one GPU)")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="multi-GPU: exchanges through torch.distributed (ProcessGroupNCCL's stream + event joins, several "
                         "graphs per step) instead of RCCL enqueued on the launch stream")
    ap.add_argument("--no-gemm-ln", action="store_true",
                    help="LayerNorm sites as GEMM + LayerNorm launches instead of the row-complete gemm_ln kernel (A/B)")
    ap.add_argument("--no-fold-rowsums", action="store_true",
                    help="bias / broadcast / LayerNorm parameter sums as their own launch instead of problems of the grouped "
                         "weight-gradient launch (A/B)")
    ap.add_argument("--no-ride-attention", action="store_true",
                    help="the query self-attention cores as launches of their own instead of riders (A/B)")
    ap.add_argument("--ride-attention-bwd", action="store_true",
                    help="also the backward query self-attention core as a rider (measured slower; off by default)")
    ap.add_argument("--split-k4h", action="store_true",
                    help="the K = 4H input-gradient products as two half-K problems summed by the LayerNorm backward "
                         "(measured neutral; off by default)")
    ap.add_argument("--no-fuser-chain", action="store_true",
                    help="the fuser block's row-local chain as grouped GEMM / gemm_ln launches instead of the one-launch "
                         "chain kernels (A/B)")
    ap.add_argument("--chain-fp32", action="store_true",
                    help="the chain kernels' products on the exact-fp32 MFMA instead of the bf16 matrix cores (bf16x3 split, "
                         "pre-split weight planes) (A/B)")
    ap.add_argument("--no-decoder-chain", action="store_true",
                    help="the decoder layer's query side as separate attention / GEMM / LayerNorm / loss launches instead of "
                         "the one-launch decoder chain kernel (A/B)")
    ap.add_argument("--no-fused-adamw", action="store_true",
                    help="never update depth_projection.weight inside its weight-gradient kernel (by default it is, where the "
                         "product runs on the tiled bf16x3 kernel: the wide / long per-GPU shapes, not the headline one; A/B)")
    ap.add_argument("--no-ride-planes", action="store_true",
                    help="re-split the chain weights in a launch of its own instead of as rider workgroups of the embedding seam (A/B)")
    ap.add_argument("--no-pair-embeddings", action="store_true",
                    help="the RGB embedding as a launch of its own (fp32 MFMA) instead of as a second product of the depth "
                         "projection's launch (A/B)")
    ap.add_argument("--overlap-planes", action="store_true",
                    help="re-split the chain weights on a parallel branch of the graph beside the input projections instead "
                         "of in stream order in front of the fuser chain (measured slower: the join costs more; A/B)")
    ap.add_argument("--overlap-param-tail", action="store_true",
                    help="the grouped weight-gradient launch and the small bucket's AdamW on a parallel branch beside the "
                         "depth weight gradient and its AdamW instead of in stream order (measured slower; A/B)")
    ap.add_argument("--erank-main-stream", action="store_true",
                    help="with --erank-weight: the Jacobi forward in stream order instead of on the side stream (A/B)")
    ap.add_argument("--no-paired", action="store_true",
                    help="every GEMM of the fuser / query-branch chains through the planner on its own (no shared launches)")
    ap.add_argument("--no-defer-loss", action="store_true",
                    help="reduce the loss statistics inside the loss kernel (last-arriving workgroup) instead of in one extra "
                         "workgroup of the AdamW launch (A/B)")
    ap.add_argument("--separate-tail", action="store_true",
                    help="decoder tail forward, losses and tail backward as three launches instead of one")
    ap.add_argument("--frame-major-input", action="store_true",
                    help="multi-GPU, sharded: keep the resident depth input [N, P] and re-lay it out every step instead of "
                         "holding it pixel-block-major")
    ap.add_argument("--erank-weight", type=float, default=0.0,
                    help="profiling only: run the step WITH the build-side rank-enhancing penalty (Jacobi forward on the "
                         "fused tokens + two-GEMM backward inside the step); the headline line is always 0 = the "
                         "reference's loss")
    ap.add_argument("--erank-warm", action="store_true",
                    help="with --erank-weight: warm-start the Jacobi sweep from the previous step's singular basis")
    ap.add_argument("--probe-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-probe", action="store_true",
                    help="multi-GPU: skip the child-process rehearsal of the RCCL step (needed under a profiler)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the multi-rank flow (process group, exchanges, graphs around them) with however "
                         "many ranks there are, even one")
    a = ap.parse_args()
    if a.probe_child and os.environ.get("R3D_PROBE_FAIL") == "1":      # (test hook: a failing rehearsal)
        sys.exit(3)
    # libraries write to fd 1 (RCCL prints a version banner there): stdout is kept for the one JSON line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist_on = world > 1 or a.force_dist
    probe_ok = True
    if dist_on and a.backend == "nccl" and not a.torch_collectives and not a.probe_child and not a.no_probe:
        probe_ok = rccl_probe_child(a)       # BEFORE anything here initialises the GPU
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if a.force_dist:
        os.environ["R3D_REHEARSE_DIST"] = "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(a.backend)
        t_ok = torch.tensor([1.0 if probe_ok else 0.0], device=device)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if float(t_ok.item()) < 1.0:            # some rank's rehearsal failed: every rank takes the torch.distributed path
            a.torch_collectives = True
    c = dict(CFG, **OTHER.get(a.config, {}))
    model = build_model(c, device, a.variant)
    if a.eval_dropout_off:
        model.eval()
    eng = model.engine()
    eng.erank_weight = a.erank_weight
    eng.erank_warm_start = a.erank_warm
    eng.use_side_stream = a.side_stream
    eng.auto_side_stream = not a.no_auto_side_stream
    eng.use_fused_decoder = a.fused_decoder
    eng.use_gemm_ln = not a.no_gemm_ln
    eng.use_fuser_chain = not a.no_fuser_chain
    eng.use_decoder_chain = not a.no_decoder_chain
    eng.chain_bf3 = not a.chain_fp32
    eng.overlap_planes = a.overlap_planes
    eng.ride_planes = not a.no_ride_planes
    eng.pair_embeddings = not a.no_pair_embeddings
    eng.overlap_param_tail = a.overlap_param_tail
    eng.erank_side_stream = not a.erank_main_stream
    if a.no_paired:
        eng.use_paired_launches = False
    eng.fold_rowsums = not a.no_fold_rowsums
    eng.ride_attention = not a.no_ride_attention
    eng.ride_attention_bwd = a.ride_attention_bwd
    eng.split_k4h = a.split_k4h
    eng.defer_tail = not a.separate_tail      # forward -> losses -> backward run back to back: one tail/loss launch
    # single-GPU flow: the loss / counter statistics (read by the host after the run) are reduced by one extra workgroup
    # of the AdamW launch instead of the loss kernel's last-arriving workgroup (the multi-GPU flows keep the latter)
    eng.defer_loss_reduce = not dist_on and not a.no_defer_loss
    from r3d_amd.parallel import DataParallelStep
    feats, depth, lab, dur, tgt = make_inputs(c, device, seed=1 + rank)
    x_dep2d = depth.reshape(c["B"] * c["S"], -1)
    x_stage = x_dep2d
    training = model.training
    dp = tp = None
    slot = [0]
    fuse_adam = a.fused_adamw

    def step_eager():
        if dp is not None:
            dp.prepare_duration_denominator(dur, c["K"] + 1)
        eng.forward_begin(feats, depth, lab, "train", training)
        if tp is not None:                  # the NEXT step's depth input travels under this step
            slot[0] ^= 1
            tp.prefetch(x_dep2d, slot[0])
            tp.exchange_forward(eng._fw["w"])
        eng.forward_finish()
        eng.losses(lab, tgt, dur, tick=True)
        fuse = (fuse_adam and (dp is None or tp is not None)) or (      # that gradient needs no exchange
            dp is None and not a.no_fused_adamw and eng.depth_adamw_fusable())   # ... and the engine finds it pays (tiles 10 / 12)
        eng.backward(fused_adamw=dict(lr=c["lr"], weight_decay=c["wd"], grad_scale=gscale) if fuse else None,
                     adamw_next=dp is None)
        if dp is not None:
            dp.wait_grads()
        eng.adamw(c["lr"], c["wd"], grad_scale=gscale, ticked=True, skip_depth=fuse, prefill_dropout=True)

    mode = "replicated"
    rs = None                               # RcclStep: exchanges on the launch stream, one hipGraph per step
    pad = c["K"] + 1
    if dist_on:
        want_tp = not a.replicated_depth and c["P"] % (4 * world) == 0
        pg_in = dist.new_group(backend=a.backend) if want_tp else None

        def agreed(fn):
            """Run fn on every rank; True only if it succeeded everywhere."""
            ok = torch.ones(1, device=device)
            try:
                fn()
                torch.cuda.synchronize()
            except Exception as e:          # noqa: BLE001
                print(f"[rank {rank}] {fn.__name__} failed ({type(e).__name__}: {e}); falling back", flush=True)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            return float(ok.item()) >= 1.0

        def rebuild(pixel_shard):
            nonlocal model, eng, dp, tp, gscale
            model = build_model(c, device, a.variant)
            if a.eval_dropout_off:
                model.eval()
            eng = model.engine()
            eng.use_side_stream = a.side_stream
            eng.use_gemm_ln = not a.no_gemm_ln
            eng.defer_tail = not a.separate_tail
            dp = DataParallelStep(eng, pixel_shard=pixel_shard, input_group=pg_in if pixel_shard else None)
            dp.broadcast_parameters()
            tp, gscale = dp.tp, dp.grad_scale

        rebuild(want_tp)
        if a.backend == "nccl" and not a.torch_collectives:
            def rccl_trial():
                nonlocal rs, x_stage
                from r3d_amd.parallel import RcclStep
                from r3d_amd.rccl import RcclComm
                # AdamW of the owned columns inside the weight-gradient GEMM's epilogue: measured on the per-rank shapes
                # (tools/psdp_shapes.py) 8 / 5 us faster than GEMM + adamw_2d at 2 / 4 ranks, neutral at 8
                rs = RcclStep(dp, RcclComm(), RcclComm(), c["lr"], c["wd"],
                              fuse_adam or (tp is not None and world in (2, 4) and c["H"] <= 128))
                if tp is not None and not a.frame_major_input:
                    # the resident depth input in the layout the sharded projection sends: [W, N, P/W] pixel-block-major
                    # (what a loader writes at host-to-device time); saves the 25.7 MB re-layout pass per step
                    x_stage = x_dep2d.view(x_dep2d.shape[0], world, -1).transpose(0, 1).contiguous()
                rs.stage(x_stage, dur, pad, 0)
                for s in (0, 1):
                    rs.stage(x_stage, dur, pad, s ^ 1)
                    rs.run(feats, depth, lab, dur, tgt, pad, training, slot=s)
            if not agreed(rccl_trial):
                rs = None
                rebuild(want_tp)
        if rs is None and tp is not None and not agreed(step_eager):     # torch.distributed form of the sharded step
            rebuild(False)
        if tp is not None:
            mode = "pixel-sharded depth_projection"
    else:
        gscale = 1.0
    if rs is not None:
        def step_eager():                   # noqa: F811  (same step, exchanges enqueued by RCCL on this stream)
            rs.stage(x_stage, dur, pad, slot[0] ^ 1)            # the next step's inputs
            rs.run(feats, depth, lab, dur, tgt, pad, training, slot=slot[0])
            slot[0] ^= 1
    for _ in range(3):
        step_eager()
    torch.cuda.synchronize()
    launch = "eager"
    run_step = step_eager
    run_many = None
    if not a.no_graph:
        try:
            if dp is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step_eager()
                run_step, launch = g.replay, "hipGraph (1 graph/step)"
                spg = max(1, a.steps_per_graph)
                if spg > 1:
                    # graphs of spg, spg/2, .., 2 steps: any step count is replayed with the fewest launches (a remainder of
                    # single-step graphs would pay the ~8 us between graph launches on every one of its steps)
                    run_many, k = [], spg
                    while k > 1:
                        gk = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gk):
                            for _ in range(k):
                                step_eager()
                        run_many.append((gk, k))
                        k //= 2
                    launch = f"hipGraph ({spg} steps/graph)"
            elif rs is not None:
                G, S = {}, {}
                side = torch.cuda.Stream(device)
                side.wait_stream(torch.cuda.current_stream())
                for s_ in (0, 1):
                    S[s_] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(S[s_], stream=side):                        # next-step inputs: their own graph
                        rs.stage(x_stage, dur, pad, s_)
                    eng._drop_ready = eng.last["w"] if training else None             # masks come from the AdamW launch
                    G[s_] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(G[s_]):
                        rs.run(feats, depth, lab, dur, tgt, pad, training, slot=s_)
                eng._drop_ready = None
                ev_side = [torch.cuda.Event(), torch.cuda.Event()]
                evs = [torch.cuda.Event() for _ in range(4)]
                nstep = [0]
                for e in ev_side + evs:
                    e.record()

                # Ordering between the two streams.  step -> needs its staged inputs: a stream-level wait on the side
                # graph's event (long complete by then: ~4 us).  side graph for step t+1 -> must not overwrite the slot
                # step t-1 still reads: measured, a stream-level dependency FROM the busy launch stream costs ~50 us per
                # step on this runtime (in-graph fork or event alike), so this one is kept on the host instead: the host
                # launches step t, then waits for step t-1's event before it launches the side graph.  Step t is already
                # queued behind step t-1, so the GPU never idles while the host waits.
                def run_step():
                    s_, cur = slot[0], torch.cuda.current_stream()
                    t = nstep[0]
                    cur.wait_event(ev_side[s_])
                    G[s_].replay()
                    evs[t % 4].record(cur)
                    evs[(t - 1) % 4].synchronize()
                    with torch.cuda.stream(side):
                        S[s_ ^ 1].replay()
                        ev_side[s_ ^ 1].record(side)
                    nstep[0] = t + 1
                    slot[0] ^= 1
                launch = "hipGraph (1 graph/step, RCCL exchanges captured on the launch stream; next inputs staged by a side graph)"
            elif tp is not None:
                hook = eng.grad_hook
                eng.grad_hook = None
                ptr = x_dep2d.data_ptr()
                gA, gC = {}, {}
                gB, gD = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                tp.ready.clear()
                for s in (0, 1):
                    tp.prefetch(x_dep2d, s)
                    sh = tp.ready[ptr]
                    if sh["work"] is not None:
                        sh["work"].wait()
                        sh["work"] = None
                    torch.cuda.synchronize()
                    gA[s] = torch.cuda.CUDAGraph()
                    eng._drop_ready = eng.last["w"] if training else None             # masks come from gD's AdamW launch
                    with torch.cuda.graph(gA[s]):
                        eng.forward_begin(feats, depth, lab, "train", training)       # consumes the slot-s shard
                    w_ = eng._fw["w"]
                    if s == 0:
                        with torch.cuda.graph(gB):
                            eng.forward_finish()
                            eng.losses(lab, tgt, dur, tick=True)
                            eng.backward_main()
                    gC[s] = torch.cuda.CUDAGraph()
                    eng.prepare_fused_adamw(dict(lr=c["lr"], weight_decay=c["wd"], grad_scale=gscale) if fuse_adam else None)
                    with torch.cuda.graph(gC[s]):
                        tp.wgrad(w_, eng.ws, eng._adam)
                with torch.cuda.graph(gD):
                    eng.adamw(c["lr"], c["wd"], grad_scale=gscale, ticked=True, skip_depth=fuse_adam, prefill_dropout=True)
                eng._drop_ready = None
                eng.grad_hook = hook
                slot[0] = 0
                tp.prefetch(x_dep2d, 0)

                def run_step():
                    cur = slot[0]
                    sh = tp.ready.pop(ptr)
                    if sh["work"] is not None:
                        sh["work"].wait()                # stream-level join with the prefetched all-to-all
                    slot[0] = cur ^ 1
                    dp.prepare_duration_denominator(dur, c["K"] + 1, async_group=pg_in)   # joins before the losses
                    tp.prefetch(x_dep2d, slot[0])        # next step's input, on its own communicator
                    gA[cur].replay()
                    tp.exchange_forward(w_)
                    dp.wait_duration_denominator()
                    gB.replay()
                    tp.exchange_backward(w_)
                    dp._on_stage("small_ready")          # all-reduce of the replicated parameters' gradients under gC
                    gC[cur].replay()
                    dp.wait_grads()
                    gD.replay()
                launch = "hipGraph (4 graphs/step around the RCCL exchanges)"
            else:
                hook = eng.grad_hook
                eng.grad_hook = None
                g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                eng._drop_ready = eng.last["w"] if training else None                 # masks come from g3's AdamW launch
                with torch.cuda.graph(g1):
                    eng.forward(feats, depth, lab, "train", training)
                    eng.losses(lab, tgt, dur, tick=True)
                    eng.backward_main()
                with torch.cuda.graph(g2):
                    eng.backward_depth_wgrad()
                with torch.cuda.graph(g3):
                    eng.adamw(c["lr"], c["wd"], grad_scale=gscale, ticked=True, prefill_dropout=True)
                eng._drop_ready = None
                eng.grad_hook = hook

                def run_step():
                    dp.prepare_duration_denominator(dur, c["K"] + 1)
                    g1.replay()
                    dp._on_stage("small_ready")          # RCCL all-reduce of the small bucket overlaps g2
                    g2.replay()
                    dp._on_stage("big_ready")
                    dp.wait_grads()
                    g3.replay()
                launch = "hipGraph (3 graphs/step around 2 RCCL all-reduces)"
        except Exception as e:                            # capture unsupported -> keep the eager path, say so
            launch = f"eager (graph capture failed: {type(e).__name__})"
            run_step = step_eager
            run_many = None
            torch.cuda.synchronize()
            if tp is not None:
                tp.ready.clear()
    def run_steps(n):
        """exactly n training steps"""
        for gk, k in (run_many or []):
            for _ in range(n // k):
                gk.replay()
            n = n % k
        for _ in range(n):
            run_step()

    run_steps(a.warmup)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(a.steps)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    single = None
    if run_many and not dist_on and not a.probe_child:
        # A/B beside the headline: the same K steps replayed as ONE step per hipGraph launch -- what train()'s _GraphedSteps
        # does per batch (it copies a fresh batch into the static buffers before every replay); the headline replays
        # --steps-per-graph consecutive steps per launch over the resident batch
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            run_step()
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        single = dict(ms_per_step=d1 / a.steps * 1e3, value=c["B"] * a.steps / d1, launch="hipGraph (1 step/graph)")
    w = eng.last["w"]
    loss_now = [float(x) for x in w.loss.cpu()]
    if a.probe_child:                       # the rehearsal: success = the one-graph RCCL step ran and stayed finite
        good = rs is not None and launch.startswith("hipGraph (1 graph") and all(x == x and abs(x) < 1e30 for x in loss_now)
        print(f"[bench probe child rank {rank}] {'ok' if good else 'NOT ok'}: {launch}; {dt / a.steps * 1e3:.3f} ms/step",
              file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(0 if good else 3)          # no teardown: nothing of RCCL / HIP / graphs can hang the exit
    if rank == 0:
        kr = kernel_rooflines(eng, c)
        name, dom = max(kr.items(), key=lambda kv: kv[1]["seconds"])
        ai = dom["flops"] / dom["bytes"] if dom["bytes"] else 0.0
        if ai > 157.3e12 / 8.0e12:
            roof = dict(bound="mfma", achieved=dom["flops"] / dom["seconds"] / 1e12, peak=157.3, unit="TFLOP/s")
        else:
            roof = dict(bound="hbm", achieved=dom["bytes"] / dom["seconds"] / 1e9, peak=8000.0, unit="GB/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["traffic"] = pmc_traffic(name) if a.config == "cfg2" else None     # measured HBM bytes per launch (PMC)
        roof["traffic_source"] = (PMC_TABLE + " (rocprofv3 --pmc passes of this command, committed; not measured in this run)"
                                  if roof["traffic"] is not None else None)
        roof["algorithmic_bytes"] = dom["bytes"]
        roof["kernel"] = name
        roof["us_per_launch"] = dom["seconds"] * 1e6
        out = dict(metric="training clips/sec (RGB+Depth fusion, DARai) at 1/2/4/8 GPUs; effective-rank match",
                   value=world * c["B"] * a.steps / dt, unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32" + (" (depth products: bf16x3 split on the bf16 matrix cores, fp32 accumulate)"
                                  if eng.depth_prec == 1 else ""),
                   data="synthetic",
                   config=dict(workload=NAMES.get(a.config, f"{a.config} per-GPU shape B={c['B']} S={c['S']} H={c['H']} "
                                                                  "(profiling only, not the headline workload)"),
                               global_batch=world * c["B"], clip_frames=c["S"], hidden=c["H"],
                               parallelism=f"dp{world}" + (f" ({mode})" if dist_on else ""),
                               launch=launch, dropout="on" if training else "off"),
                   roofline=roof,
                   kernels={k: dict(us=v["seconds"] * 1e6, GBps=v["bytes"] / v["seconds"] / 1e9,
                                    TFLOPs=v["flops"] / v["seconds"] / 1e12) for k, v in kr.items()},
                   final_losses=loss_now)
        # the whole step against ITS bound (algorithmic bytes and FLOPs of one step of this configuration)
        if a.config in STEP_WORK:
            wk = STEP_WORK[a.config]
            t_step = dt / a.steps
            hbm_t, mfma_t = wk["bytes"] / 8.0e12, wk["flops"] / 157.3e12
            out["roofline_step"] = dict(bound="hbm" if hbm_t >= mfma_t else "mfma", algorithmic_bytes=wk["bytes"],
                                        algorithmic_flops=wk["flops"], achieved_GBps=wk["bytes"] / t_step / 1e9,
                                        achieved_TFLOPs=wk["flops"] / t_step / 1e12,
                                        frac=max(hbm_t, mfma_t) / t_step, per="GPU step",
                                        note="SURVEY.md 8(d) totals / ms_per_step against 8 TB/s and 157.3 TFLOP/s (fp32 MFMA)")
        if single is not None:
            out["single_step_graph"] = single
        if dist_on:
            from r3d_amd import rccl as _rccl
            flow = ("rccl-graph" if rs is not None else ("torch-collectives" if launch.startswith("hipGraph") else "eager"))
            out["multi_gpu"] = dict(world=world, rccl_ranks=(rs.comm.nranks() if rs is not None else None),
                                    torch_distributed_ranks=dist.get_world_size() if dist.is_initialized() else 1,
                                    flow=flow, mode=mode, probe_child_ran=PROBE["ran"], probe_child_rc=PROBE["rc"],
                                    torch_collectives_forced=bool(a.torch_collectives))
        try:
            out["erank"] = erank_field(eng)
        except Exception as e:                                # noqa: BLE001  (the headline number must still print)
            out["erank"] = dict(error=f"{type(e).__name__}: {e}")
        if a.variant != "tokenfusion":
            out["config"]["workload"] += f" [{a.variant} fuser variant: profiling only, not the headline model]"
        if a.erank_weight != 0.0:
            out["config"]["workload"] += (f" [with the build-side effective-rank penalty, weight {a.erank_weight}, "
                                          f"{'warm-started' if a.erank_warm else 'cold'} Jacobi in the step: profiling only]")
            out["erank_in_step"] = dict(sweeps=float(eng.last["w"].er_stats[0, 3]), erank=float(eng.last["w"].er_stats[0, 0]))
        if world == 1 and not a.no_cpu_baseline and a.config == "cfg2" and a.variant == "tokenfusion":
            out["cpu_baseline"] = cpu_baseline(c)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist_on:
        # every rank is past its last collective (the barrier after the timed loop); leave without the interpreter's
        # teardown of communicators, captured graphs and HIP state, so nothing can hang after the result is out
        sys.stdout.flush()
        sys.stderr.flush()
        dist.barrier()
        os._exit(0)


if __name__ == "__main__":
    main()
