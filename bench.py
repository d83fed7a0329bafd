#!/usr/bin/env python
"""bench.py -- train poses/sec of the ManiPose lifting hot path on MI355X (BASELINE.json metric).

One "step" = forward + WTA multi-hypothesis loss + backward + gradient all-reduce + Adam on a synthetic batch of
B windows per GPU of H36M shape (T=243, J=17, K=5, C=512, depth 8), inputs resident in HBM.
    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank; without it the
process spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself (before anything touches a GPU) and
relays its output.
Prints ONE JSON line on rank 0 with
  `roofline`      dominant kernel = gemm_bf16_persist_kernel, the forward + dgrad Linear GEMMs; HIP events inside the engine on the
                  stream each launch goes to;
  `parity`        MPJPE of the TIMED precision against the fp32 CPU oracle at full size (T=243 K=5, the bench's own weights), measured in
                  this run AT THE TIMED BATCH: the parity windows are embedded in a full B-window eval forward, so the kernels that are
                  timed (the persistent split-precision GEMMs incl. the recomputed-LayerNorm residual epilogue) are the ones checked;
                  the same windows as a stand-alone small batch are reported next to it (`mpjpe_m_small_batch`); bound 1e-4 m
                  (north star); the oracle forward runs in a CPU child process; at N>1 every rank checks its own replica;
  `rccl`          (N>1) backend name and the sum of a ones tensor all-reduced over the process group = the ranks RCCL really joined;
  `other_precisions`  (N=1) poses/s and the same parity figure of the other two precisions, a few steps each;
  `other_configs`  (N=1) the other single-GPU BASELINE configurations in the timed precision, a few steps each with the same CPU-child parity:
                  T27_K1_manifold_train (configs[1]), T81_K5_train and T81_K5_eval_tta (configs[4]'s shape: evaluation with flip-TTA, MPJPE / PCK@150 / AUC
                  against the oracle's composition of the same procedure);  `small_batch`: the headline workload at the reference's batch sizes 3 and 25;
  `cpu_baseline`  oracle/manipose_ref.py timed on the host cores (rank 0, N=1 only), B=1 and the reference's default B=3; `cores_limited_by` says which
                  limit (cgroup quota / affinity mask / MANIPOSE_CPU_THREADS) set the thread count.
Default precision: "bf16x3" (split operands, fp32 accumulate; bf16 backward) - the fastest precision whose drift stays inside the 1e-4 m bound;
"bf16" (BASELINE config #3's wording) is faster but drifts ~3-4.6 mm.  Operand form of the split precision (mp_model_config::f16f8, named in
`config.split_forms`): since round 6 the default is `--f16f8 3` - all four Linear layers of a block of the rotations net as ONE fp16 + ONE block-scaled
fp8 matrix-core product per 64 reduction indices (hand-scheduled k-steps; the attention products stay three bf16 products; bf16 backward): same-box
A/B +3.4 % poses/s (+4.3 % on the final round-6 tree) over the three-product form at 1.6e-5 m instead of 0.94e-5 m.  `--f16f8 0` = every product as three
bf16 products (rounds 2-5; timed for a few steps in every default run as `other_precisions.bf16x3_three_bf16_products`); 1 / 2 = the older partial
forms (qkv / fc1 [/ fc2 with --f16-backward]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_POSE = {243: 3.705, 81: 3.562, 27: 3.513}      # SURVEY.md 8d (3 x forward GEMM+attention FLOPs)
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}                  # MI355X_MICROARCH.md: dense matrix peaks
PEAK_HBM_GBPS = 8000.0                                         # HBM3E spec peak (measured copy peak on this pool: ~5.5-6.3 TB/s)


def pmc_traffic_per_launch(precision, batch, forms=""):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (2 x FETCH_SIZE per the gfx950 correction of
    MI355X_MICROARCH.md section HBM, + WRITE_SIZE): profiles/pmc_traffic.json, written by tools/refresh_profiles.sh together with the
    <tag>_pmc_hbm_traffic.csv whose last row it repeats (one source, so the two cannot diverge).  It is NOT measured in this run (counters
    need rocprofv3 around the process): the record says where it comes from.  None when that configuration was never profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(f"{precision}{forms}:{batch}")
        if not e or e.get("bytes_per_launch") is None:
            return None
        return {"bytes_per_launch": e["bytes_per_launch"],
                "source": f"{e.get('source', 'profiles/pmc_traffic.json')} (offline rocprofv3 --pmc passes of this command, not a measurement of this run)"}
    except (OSError, ValueError):
        return None



def host_cores(why=False):
    """CPU cores this job may actually use: cgroup quota (cpu.max) if set, else the affinity mask, else cpu_count; capped by
    MANIPOSE_CPU_THREADS (default 64).  why=True: (n, which of those limits decided)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n, reason = aff, f"affinity mask ({aff} of os.cpu_count()={os.cpu_count()})"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(float(quota) / float(period)))
            if q < n:
                n, reason = q, f"cgroup cpu.max quota {quota}/{period} = {q} cores (affinity mask {aff}, os.cpu_count()={os.cpu_count()})"
    except Exception:
        pass
    cap = int(os.environ.get("MANIPOSE_CPU_THREADS", "64"))
    if cap < n:
        n, reason = cap, f"MANIPOSE_CPU_THREADS cap {cap} (the job could use {n})"
    n = max(1, n)
    return (n, reason) if why else n


def cpu_baseline(T, K, steps=10):
    """Oracle (CPU restatement of the reference, fp32, torch autograd + torch.optim.Adam) on a bounded sample:
    B=1 window of the same workload, 3 warm-up + `steps` timed training steps; then the reference's default batch (B=3,
    hpe/conf/config.yaml:26) for 1 warm-up + 3 timed steps."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import manipose_ref as orc
    ncores, why = host_cores(why=True)
    torch.set_num_threads(ncores)
    cfg = dict(orc.FULL_CFG, T=T, n_hyp=K)
    st = {k: v.requires_grad_(True) for k, v in orc.make_state(cfg, seed=0).items()}
    opt = torch.optim.Adam(list(st.values()), lr=4e-5, weight_decay=1e-6)

    def timed(B, warm, n):
        X, y = orc.synthetic_batch(B, T, seed=42)
        times = []
        for i in range(n + warm):
            t0 = time.perf_counter()
            poses, scores = orc.rmcl_manifold_forward(X, st, orc.oracle_cfg(cfg))
            total, _ = orc.rmcl_training_loss(poses, scores, y)
            opt.zero_grad()
            total.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        return sorted(times[warm:])[len(times[warm:]) // 2]
    warm = 3
    dt = timed(1, warm, steps)
    dt3 = timed(3, 1, 3)
    model_name = "?"
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "?")
    except OSError:
        pass
    return {"value": T / dt, "unit": "poses/s", "cores": torch.get_num_threads(), "kind": "port",
            "cores_limited_by": why, "value_batch3": 3 * T / dt3,
            "sample": f"B=1 window T={T} K={K}, eval-mode DropPath off, {warm} warm-up + {steps} timed steps (median), "
                      f"fwd+loss+bwd+Adam, torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads of os.cpu_count()="
                      f"{os.cpu_count()} ({model_name}); value_batch3 = the same at the reference's default batch of 3 windows "
                      f"(1 warm-up + 3 timed steps)"}


def parity_oracle(io_dir):
    """(CPU child) fp32 oracle forwards on the weights / inputs the parent saved (parity_in.pt: a list of jobs, one per model configuration);
    writes the poses and scores next to them.  A job with `y` also gets the oracle's composition of the evaluation procedure (flip test-time
    augmentation, weighted-average aggregation, MPJPE / 3DPCK@150 mm / AUC: eval_utils.py:16-223, pck.py:92-199) on its windows."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import manipose_ref as orc
    torch.set_num_threads(host_cores())
    jobs = torch.load(os.path.join(io_dir, "parity_in.pt"))
    out = {}
    with torch.no_grad():
        for job in jobs:
            cfg = dict(orc.FULL_CFG, T=job["T"], n_hyp=job["K"] if job["arch"] == "rmcl" else 0)
            ocfg = orc.oracle_cfg(cfg)
            if job["arch"] == "rmcl":
                poses, scores = orc.rmcl_manifold_forward(job["X"], job["state"], ocfg)
                res = {"poses": poses, "scores": scores}
                if job.get("y") is not None:
                    y = job["y"]
                    p1, s1 = orc.rmcl_manifold_forward(orc.flip_pose(job["X"]), job["state"], ocfg)
                    agg = (orc.aggregate(poses, scores, "weighted_ave") + orc.aggregate(orc.flip_pose(p1), s1, "weighted_ave")) / 2
                    pck, auc = orc.keypoint_3d_pck_auc(1000 * agg.reshape(-1, 17, 3), 1000 * y.reshape(-1, 17, 3))
                    res["eval"] = {"mpjpe_mm": 1000 * orc.mpjpe_error(agg, y).item(), "pck150": pck.item(), "auc": auc.item()}
            else:
                res = {"poses": orc.manifold_forward(job["X"], job["state"], ocfg)}
            out[job["name"]] = res
    torch.save(out, os.path.join(io_dir, "parity_out.pt"))


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


PARITY_BOUND_M = 1e-4          # BASELINE.json north_star: outputs within 1e-4 on MPJPE
PARITY_WINDOWS = 3             # parity windows per run (oracle forward on the host: ~1 s each), spread over the timed batch
EXTRA_BATCH = {"fp32": 16, "bf16": 79, "bf16x3": 79}


def _power_pass(step, seconds=3.0):
    """socket power (W) and shader clock (MHz) sampled through rocm-smi every ~0.2 s while `step` runs back to back; None if rocm-smi is not usable"""
    import re, shutil, subprocess, threading
    import torch
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run([smi, "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
                row = [l for l in out.splitlines() if l.startswith("card0")]
                if row:
                    f = row[0].split(",")
                    samples.append((int(re.sub(r"\D", "", f[5])), float(f[9])))
            except Exception:
                return
            stop.wait(0.2)

    cap = None
    try:
        out = subprocess.run([smi, "--showmaxpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
        row = [l for l in out.splitlines() if l.startswith("card0")]
        nums = re.findall(r"\d+(?:\.\d+)?", row[0].split(",", 1)[1]) if row else []
        cap = float(nums[0]) if nums else None
    except Exception:
        pass
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    try:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            step()
            torch.cuda.synchronize()
    finally:
        stop.set()
        th.join(timeout=10)
    busy = samples[2:] if len(samples) > 4 else samples      # (the first samples still see the ramp)
    if not busy:
        return None
    return {"mean_w": sum(w for _, w in busy) / len(busy), "max_w": max(w for _, w in busy), "cap_w": cap,
            "mean_sclk_mhz": sum(c for c, _ in busy) / len(busy), "part_max_sclk_mhz": 2400, "samples": len(busy),
            "note": "rocm-smi (socket graphics package power, sclk) sampled every ~0.2 s over ~3 s of the timed configuration's steps, behind the timed region"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("MANIPOSE_BENCH_BATCH", "158")),
                    help="windows per GPU (158: ceil(158*4131/256) = 2550 row panels, so the 2 / 4 / 6 column tiles of the four Linear shapes "
                         "give 19.9 / 39.8 / 59.8 rounds of 256 persistent workgroups - no nearly empty last round; 169 GiB of workspace.  "
                         "Rounds 1-4 ran 79 (124 GiB then, 85 GiB now): 158 amortises the per-step fixed work, +1.6 %% poses/s)")
    ap.add_argument("--frames", type=int, default=243)
    ap.add_argument("--hyp", type=int, default=5)
    ap.add_argument("--precision", default=os.environ.get("MANIPOSE_PRECISION", "bf16x3"), choices=["bf16", "bf16x3", "fp32"],
                    help="bf16x3 (default) = split bf16 hi/lo operands, 3 matrix-core products per product: inside the 1e-4 m parity bound; "
                         "bf16 = plain bf16 matrix cores (fp32 accumulate/residual/softmax), ~3 mm drift; fp32 = fp32 matrix cores")
    ap.add_argument("--f16f8", type=int, default=int(os.environ.get("MANIPOSE_F16F8", "3")), choices=[0, 1, 2, 3],
                    help="bf16x3 only (mp_model_config::f16f8): 1 = qkv / fc1 as one fp16 + one block-scaled fp8 product, 2 = fc2 as well (needs --f16-backward), "
                         "3 (default since round 6) = all four Linear layers of a block, bf16 backward; 0 = every product as three bf16 products (rounds 2-5)")
    ap.add_argument("--f16-backward", action="store_true", help="with --f16f8 >= 1: backward GEMMs of those layers on saturating scaled-fp16 operands")
    ap.add_argument("--single-queue", action="store_true",
                    help="every kernel on the caller's stream (mp_model_config::streams = 3): isolated kernel durations under rocprofv3")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="mp_set_option test / tuning hook (e.g. attn_two_phase=0, gemm_persist_wgs=192); A/B timing only")
    ap.add_argument("--grad-buckets", action="store_true",
                    help="N > 1: overlap the gradient exchange with the backward (one all-reduce per layer of the rotations net on a communication "
                         "stream) instead of one all-reduce of the flat buffer behind it; off by default (never measured on a multi-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the one-queue pass behind the timed region (profiling runs: a kernel-stats file should hold the timed configuration's launches only)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock samples behind the timed region (N=1 only)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity measurement (profiling passes)")
    ap.add_argument("--no-extra", action="store_true", help="skip the few-step runs of the other precisions (N=1 only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the few-step runs of the other single-GPU BASELINE configurations (T=27 K=1 manifold train, T=81 K=5 train, T=81 K=5 "
                         "evaluation with flip-TTA) and of the reference's own batch sizes (3, 25) behind the timed region (N=1 only)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) CPU child: time the CPU oracle and print its JSON")
    ap.add_argument("--parity-io", default=None, help="(internal) CPU child: directory with parity_in.pt; the oracle forward is written next to it")
    args = ap.parse_args()
    if args.cpu_baseline_only or args.parity_io:
        if args.parity_io:
            parity_oracle(args.parity_io)
        print(json.dumps(cpu_baseline(args.frames, args.hyp) if args.cpu_baseline_only else {}), flush=True)
        return
    import subprocess
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start one rank per GPU as fresh child processes (this process never touches a GPU) and relay
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import tempfile
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # the model is built on the CPU first: its initial weights and one full-size window go to the CPU child, which returns the fp32
    # oracle's outputs for the parity block (and, at N=1, the CPU baseline timing).  Nothing here has touched the GPU yet.
    from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
    from manipose_amd.training import LiftingTrainer

    def build_model(precision, batch, arch="rmcl", frames=None, f16f8=None):
        from manipose_amd import ManifoldMixSTE
        torch.manual_seed(42)
        frames = frames or args.frames
        mdl = (RMCLManifoldMixSTE(h36m_skeleton(), num_frame=frames, n_hyp=args.hyp, drop_path_rate=0.1) if arch == "rmcl"
               else ManifoldMixSTE(h36m_skeleton(), num_frame=frames, drop_path_rate=0.1))
        with torch.no_grad():                      # exercise the (zero-initialised) positional tables too
            for n, p in mdl.named_parameters():
                if n.endswith("pos_embed"):
                    p.normal_(0.0, 0.02)
        mdl.precision = precision
        if precision == "bf16x3":
            mdl.f16f8, mdl.f16_backward = (args.f16f8 if f16f8 is None else f16f8), (args.f16_backward if f16f8 is None else False)
        mdl.side_stream = mdl.wgrad_stream = not args.single_queue
        mdl.max_batch_hint = batch
        return mdl

    for opt in args.option:                             # process-wide kernel selectors (A/B timing): include/manipose_hip.h, mp_set_option
        from manipose_amd import _lib as _l
        name, _, val = opt.partition("=")
        _l.check(_l.load().mp_set_option(name.encode(), int(val)), f"mp_set_option({name})")
    model = build_model(args.precision, args.batch)
    cpu_json, oracle_out, oracle_all = None, None, {}
    # parity windows: the same on every rank (own generator), placed at the start, middle and end of the timed batch
    gp = torch.Generator().manual_seed(4242)
    npar = min(PARITY_WINDOWS, args.batch)
    X_par = (0.3 * torch.randn(npar, args.frames, 17, 2, generator=gp)).clamp(-1, 1)

    def spread(nb):          # rows of an nb-window batch that carry the parity windows: first, middle, last
        return [min(nb - 1, (i * (nb - 1)) // max(1, npar - 1)) for i in range(npar)]
    want_cpu = world == 1 and args.gpus == 1 and not args.no_cpu_baseline
    # N=1: the other single-GPU BASELINE configurations, a few steps each behind the timed region (BASELINE.json configs[1] and [4]; their
    # token counts = 79 windows of 243 frames).  The models are built here, on the CPU, so that the CPU child can run the oracle on their weights.
    OTHER = {"T27_K1_manifold_train": dict(arch="manifold", T=27, B=711), "T81_K5_train": dict(arch="rmcl", T=81, B=237)}
    other_models, other_par = {}, {}
    want_other = world == 1 and args.gpus == 1 and not args.no_other_configs and not args.no_parity and args.frames == 243 and args.hyp == 5
    if want_other:
        for name, oc in OTHER.items():
            other_models[name] = build_model(args.precision, oc["B"], oc["arch"], oc["T"])
            gq = torch.Generator().manual_seed(4243 + oc["T"])
            Xq = (0.3 * torch.randn(PARITY_WINDOWS, oc["T"], 17, 2, generator=gq)).clamp(-1, 1)
            yq = 0.075 * torch.randn(PARITY_WINDOWS, oc["T"], 17, 3, generator=gq)      # errors around the 150 mm threshold: PCK / AUC informative
            yq[:, :, 0] = 0
            other_par[name] = (Xq, yq)
    if rank == 0 and (want_cpu or not args.no_parity):
        io_dir = tempfile.mkdtemp(prefix="manipose_bench_")
        child = [sys.executable, os.path.abspath(__file__), "--frames", str(args.frames), "--hyp", str(args.hyp)]
        if not args.no_parity:
            jobs = [{"name": "main", "arch": "rmcl", "state": {k: v.detach().clone() for k, v in model.state_dict().items()}, "X": X_par, "T": args.frames, "K": args.hyp}]
            for name, mdl in other_models.items():
                jobs.append({"name": name, "arch": OTHER[name]["arch"], "state": {k: v.detach().clone() for k, v in mdl.state_dict().items()},
                             "X": other_par[name][0], "y": other_par[name][1] if OTHER[name]["arch"] == "rmcl" else None, "T": OTHER[name]["T"], "K": args.hyp})
            torch.save(jobs, os.path.join(io_dir, "parity_in.pt"))
            del jobs
            child += ["--parity-io", io_dir]
        if want_cpu:
            child += ["--cpu-baseline-only"]
        try:
            r = subprocess.run(child, capture_output=True, text=True, timeout=420)
            js = json.loads(r.stdout.strip().splitlines()[-1])
            if want_cpu:
                cpu_json = js
            if not args.no_parity:
                oracle_all = torch.load(os.path.join(io_dir, "parity_out.pt"))
                oracle_out = {k: oracle_all["main"][k] for k in ("poses", "scores")}
        except Exception as e:       # noqa: BLE001
            if want_cpu:
                cpu_json = {"value": None, "unit": "poses/s", "cores": host_cores(), "kind": "port",
                            "sample": f"CPU oracle did not finish inside its 420 s bound ({type(e).__name__})"}
        import shutil
        shutil.rmtree(io_dir, ignore_errors=True)

    # rehearsal hooks (a one-GPU box cannot host two RCCL ranks): MANIPOSE_BENCH_DEVICE pins every rank to one device and
    # MANIPOSE_BENCH_BACKEND=gloo carries the collectives through the host; the driver's multi-GPU runs use neither
    if os.environ.get("MANIPOSE_BENCH_DEVICE"):
        local = int(os.environ["MANIPOSE_BENCH_DEVICE"])
    backend = os.environ.get("MANIPOSE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    # every rank starts from rank 0's weights (one broadcast of the flat buffer) and - for the parity block - holds rank 0's oracle output
    rccl = None
    have_oracle = torch.tensor([1 if oracle_out is not None else 0], device="cuda", dtype=torch.int32)
    model = model.cuda()
    if world > 1:
        from manipose_amd.distributed import broadcast_parameters
        model._ensure_engine(args.batch, torch.device("cuda", local))
        broadcast_parameters(model._flat)
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)                  # every rank contributes 1: the sum is the number of ranks the backend really joined
        rccl = {"backend": dist.get_backend(), "ranks": int(round(ones.item())), "world_size": world,
                "device": torch.cuda.get_device_name(local)}
        dist.broadcast(have_oracle, src=0)
        if have_oracle.item():
            K = args.hyp
            op = oracle_out["poses"].cuda() if rank == 0 else torch.empty(npar, K, args.frames, 17, 3, device="cuda")
            osc = oracle_out["scores"].cuda() if rank == 0 else torch.empty(npar, K, args.frames, 1, device="cuda")
            dist.broadcast(op, src=0)
            dist.broadcast(osc, src=0)
            oracle_out = {"poses": op, "scores": osc}
    elif oracle_out is not None:
        oracle_out = {k: v.cuda() for k, v in oracle_out.items()}

    B, T = args.batch, args.frames
    g = torch.Generator(device="cuda").manual_seed(42 + rank)
    X = (0.3 * torch.randn(B, T, 17, 2, device="cuda", generator=g)).clamp(-1, 1)
    y = 0.3 * torch.randn(B, T, 17, 3, device="cuda", generator=g)
    y[:, :, 0] = 0

    def parity_of(mdl, Xfull, ora=None, Xp=None):
        """MPJPE (m) of mdl's eval-mode forward against the oracle's poses on the parity windows, measured twice: with the windows embedded
        in the full batch Xfull at its first, middle and last row (the shapes - and therefore the kernels - of the timed steps) and as a
        small stand-alone batch.  None without the oracle output.  ora / Xp: another configuration's oracle output and parity windows."""
        ora = oracle_out if ora is None else ora
        Xp = X_par if Xp is None else Xp
        if ora is None or Xfull.shape[0] < npar:
            return None
        idx = spread(Xfull.shape[0])
        Xe = Xfull.clone()
        Xe[idx] = Xp.cuda()
        with torch.no_grad():
            of, o1 = mdl.eval()(Xe), mdl(Xp.cuda())
        if "scores" not in ora:                     # single-hypothesis model: poses only
            d = (of[idx] - ora["poses"].cuda()).norm(dim=-1)
            d1 = (o1 - ora["poses"].cuda()).norm(dim=-1)
            return {"mpjpe_m": d.mean().item(), "mpjpe_m_small_batch": d1.mean().item(), "batch": int(Xfull.shape[0]), "windows_checked": npar,
                    "rows_in_batch": idx, "max_joint_err_m": d.max().item()}
        pf, sf = of[0][idx].clone(), of[1][idx].clone()
        p1 = o1[0]
        d = (pf - ora["poses"].cuda()).norm(dim=-1)
        d1 = (p1 - ora["poses"].cuda()).norm(dim=-1)
        # mean = MPJPE, the north-star metric; the tail is a handful of joints behind near-degenerate 6-D frames (two nearly colinear
        # 3-vectors: the Gram-Schmidt step divides by |a x b|), where ANY operand rounding is amplified
        return {"mpjpe_m": d.mean().item(), "mpjpe_m_small_batch": d1.mean().item(), "batch": int(Xfull.shape[0]), "windows_checked": npar, "rows_in_batch": idx,
                "p999_joint_err_m": torch.quantile(d.flatten().float(), 0.999).item(), "max_joint_err_m": d.max().item(),
                "score_max_abs_diff": (sf - ora["scores"].cuda()).abs().max().item()}

    parity = parity_of(model, X)               # on the initial weights, before any optimisation step; every rank checks its own replica
    if world > 1 and parity is not None:
        worst = torch.tensor([parity["mpjpe_m"]], device="cuda", dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        parity["mpjpe_m_worst_rank"] = worst.item()
        parity["ranks_checked"] = world
    model = model.train()
    trainer = LiftingTrainer(model, lr=4e-5, weight_decay=1e-6, seed=42, grad_buckets=args.grad_buckets)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model built, B={B} T={T} precision={args.precision}" + (f", parity at B={B}: {parity['mpjpe_m']:.2e} m" if parity else ""))
    for i in range(args.warmup):
        terms = trainer.train_step(X, y)
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, workspace {model._engine.workspace_bytes / 2**30:.1f} GiB")
    barrier()
    log("warmup done")
    eng = model._engine
    if not args.no_prof:
        eng.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = trainer.train_step(X, y)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f} s")
    prof = eng.prof_collect() if not args.no_prof else None
    kinds = eng.prof_kinds() if not args.no_prof else None
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss = float(terms.sum().item())

    # N > 1: the open question of DESIGN section 6, answered by the run itself - a few steps in each exchange mode with device events around
    # the backward and the exchange: does the backward get longer with RCCL kernels resident (bucketed), and how much of the single
    # collective is exposed behind it?
    exchange = None
    if world > 1:
        exchange = {}
        nx = max(2, min(5, args.steps))
        for mode in ("single", "bucketed"):
            trainer.grad_buckets = mode == "bucketed"
            trainer.train_step(X, y)                      # one untimed step in the mode (communication stream / buffers)
            trainer.time_exchange = True
            barrier()
            tx0 = time.perf_counter()
            for _ in range(nx):
                trainer.train_step(X, y)
            barrier()
            dtx = torch.tensor([time.perf_counter() - tx0], device="cuda", dtype=torch.float64)
            dist.all_reduce(dtx, op=dist.ReduceOp.MAX)
            tm = trainer.exchange_times()
            trainer.time_exchange = False
            tt = torch.tensor([tm["backward_ms"], tm["exposed_exchange_ms"]], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            exchange[mode] = {"ms_per_step": 1e3 * dtx.item() / nx, "backward_ms": tt[0].item(), "exposed_exchange_ms": tt[1].item(), "steps": nx,
                              "timed_mode": mode == ("bucketed" if args.grad_buckets else "single")}
        trainer.grad_buckets = args.grad_buckets

    # N=1: the kernel classes again with every kernel on ONE queue (mp_model_set_streams on the same model, a few steps behind the timed region).
    # Under the three queues of the timed steps a class's HIP-event time includes the time its kernels spent time-slicing with the other
    # queues (the classes then sum to far more than the step); on one queue a kernel's event time is its own duration and the classes sum to
    # the step.
    isolated = None
    workspace_gib = model._engine.workspace_bytes / 2**30
    if rank == 0 and world == 1 and prof is not None and not args.single_queue and not args.no_isolated:
        ni = max(2, min(5, args.steps))
        eng.set_streams(side_stream=False, wgrad_stream=False)
        trainer.train_step(X, y)
        torch.cuda.synchronize()
        eng.prof_collect()                                # (reset)
        ti = time.perf_counter()
        for _ in range(ni):
            trainer.train_step(X, y)
        torch.cuda.synchronize()
        dti = (time.perf_counter() - ti) / ni
        pi = eng.prof_collect()
        pi.pop("gemm_persist")
        eng.set_streams(side_stream=True, wgrad_stream=True)
        isolated = {"steps": ni, "ms_per_step": 1e3 * dti, "classes_ms_per_step": {n: v["ms"] / ni for n, v in pi.items()},
                    "classes_sum_ms_per_step": sum(v["ms"] for v in pi.values()) / ni}
        log(f"one-queue pass: {isolated['ms_per_step']:.1f} ms per step, kernel classes sum {isolated['classes_sum_ms_per_step']:.1f} ms")

    # N=1: socket power and shader clock while the same steps run (behind the timed region; rocm-smi child processes, read-only).  Round 5 found the
    # GEMMs and the step as a whole at the board's power cap ("power"): the kernel times then follow energy, not the schedule.
    power = None
    if rank == 0 and world == 1 and not args.no_power:
        power = _power_pass(lambda: trainer.train_step(X, y), seconds=3.0)
        if power:
            log(f"power pass: {power['mean_w']:.0f} W mean of a {power.get('cap_w')} W cap, shader clock {power['mean_sclk_mhz']:.0f} MHz, {power['samples']} samples")

    # N=1: the other precisions on the same workload, a few steps each (their own batch sizes), with the same parity measurement
    other = {}
    run_other = rank == 0 and world == 1 and bool(other_models) and all(k in oracle_all for k in other_models)
    if rank == 0 and world == 1 and (not args.no_extra or run_other):
        del trainer, terms, eng
        model._engine = None
        del model
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_extra:
        variants = [(p_, p_, None) for p_ in ("bf16x3", "bf16", "fp32") if p_ != args.precision]
        if args.precision == "bf16x3" and args.f16f8 != 0:      # the timed operand form's predecessor: every product as three bf16 products (rounds 2-5)
            variants.insert(0, ("bf16x3_three_bf16_products", "bf16x3", 0))
        for label, prec, f8 in variants:
            try:
                Bx = EXTRA_BATCH[prec]
                mx = build_model(prec, Bx, f16f8=f8).cuda()
                Xx = X[:Bx] if Bx <= B else X.repeat((Bx + B - 1) // B, 1, 1, 1)[:Bx]
                yx = y[:Bx] if Bx <= B else y.repeat((Bx + B - 1) // B, 1, 1, 1)[:Bx]
                par = parity_of(mx, Xx)
                tx = LiftingTrainer(mx.train(), lr=4e-5, weight_decay=1e-6, seed=42)
                nst = 5 if prec != "fp32" else 3
                tx.train_step(Xx, yx)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nst):
                    tx.train_step(Xx, yx)
                torch.cuda.synchronize()
                dtx = (time.perf_counter() - t1) / nst
                other[label] = {"poses_per_s": Bx * T / dtx, "ms_per_step": 1e3 * dtx, "windows_per_gpu": Bx, "steps": nst, "warmup": 1,
                               "parity_mpjpe_m": par["mpjpe_m"] if par else None,
                               "within_bound": (par["mpjpe_m"] <= PARITY_BOUND_M) if par else None}
                log(f"{label}: {other[label]['poses_per_s']:.0f} poses/s, parity {other[label]['parity_mpjpe_m']}")
                mx._engine = None
                del tx, mx
                torch.cuda.empty_cache()
            except Exception as e:       # noqa: BLE001
                other[label] = {"error": f"{type(e).__name__}: {e}"}
    # N=1: the other single-GPU BASELINE configurations and the reference's own batch sizes, a few steps each in the timed precision, with the same
    # parity measurement (CPU-child oracle on the same weights; parity windows embedded at the first / middle / last row of the batch)
    other_cfgs, small_batch = {}, {}
    if run_other:
        sys.path.insert(0, os.path.join(ROOT, "hpe"))
        from _entry import evaluate

        def time_train(mdl, Xb, yb, nst):
            tr = LiftingTrainer(mdl.train(), lr=4e-5, weight_decay=1e-6, seed=42)
            tr.train_step(Xb, yb)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nst):
                tr.train_step(Xb, yb)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / nst
        for name, oc in OTHER.items():
            try:
                Bo, To = oc["B"], oc["T"]
                mo = other_models.pop(name).cuda()
                go = torch.Generator(device="cuda").manual_seed(43)
                Xo = (0.3 * torch.randn(Bo, To, 17, 2, device="cuda", generator=go)).clamp(-1, 1)
                yo = 0.3 * torch.randn(Bo, To, 17, 3, device="cuda", generator=go)
                yo[:, :, 0] = 0
                ora = oracle_all[name]
                par = parity_of(mo, Xo, ora, other_par[name][0])
                rp = None
                if "eval" in ora:            # (the evaluation procedure against the oracle's, on the INITIAL weights - before the timed training steps move them)
                    Xp, yp = (t.cuda() for t in other_par[name])
                    rp = evaluate(mo, Xp, yp, batch=Xp.shape[0], tta=True, analytics=True)
                dto = time_train(mo, Xo, yo, 4)
                other_cfgs[name] = {"workload": f"H36M lifting T={To} J=17 K={args.hyp if oc['arch'] == 'rmcl' else 1} "
                                                + ("ManiPose (RMCLManifoldMixSTE)" if oc["arch"] == "rmcl" else "single-hypothesis ManifoldMixSTE")
                                                + " full width (C=512, depth 8), train step fwd+loss+bwd+Adam",
                                    "poses_per_s": Bo * To / dto, "ms_per_step": 1e3 * dto, "windows_per_gpu": Bo, "steps": 4, "warmup": 1, "precision": args.precision,
                                    "model_tflops": Bo * To / dto * TRAIN_GFLOP_PER_POSE[To] / 1e3,
                                    "parity_mpjpe_m": par["mpjpe_m"], "parity_mpjpe_m_small_batch": par["mpjpe_m_small_batch"],
                                    "within_bound": max(par["mpjpe_m"], par["mpjpe_m_small_batch"]) <= PARITY_BOUND_M}
                log(f"{name}: {other_cfgs[name]['poses_per_s']:.0f} poses/s, parity {par['mpjpe_m']:.2e} m")
                if "eval" in ora:
                    # BASELINE config #5's procedure on this model: evaluation with flip test-time augmentation batched into one forward, aggregation,
                    # MPJPE / 3DPCK@150 mm / AUC (hpe/_entry.py evaluate = eval_utils.py:16-223), (a) on the parity windows against the oracle's
                    # composition of the same procedure, (b) timed over 4 batches of Be windows (forward of 2 Be windows each)
                    Be = Bo // 2
                    Xe4, ye4 = Xo[:Be].repeat(4, 1, 1, 1), 0.25 * yo[:Be].repeat(4, 1, 1, 1)
                    evaluate(mo, Xe4[:Be], ye4[:Be], batch=Be, tta=True, analytics=True)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    evaluate(mo, Xe4, ye4, batch=Be, tta=True, analytics=True)
                    torch.cuda.synchronize()
                    dte = time.perf_counter() - t1
                    oe = ora["eval"]
                    other_cfgs[name.replace("_train", "_eval_tta")] = {
                        "workload": f"MPI-INF-3DHP-shaped evaluation T={To} K={args.hyp}: flip-TTA in one forward of 2 x {Be} windows, weighted-average / best-score / "
                                    f"oracle aggregation, MPJPE, 3DPCK@150 mm, AUC, P-MPJPE (analytics kernels), synthetic windows",
                        "poses_per_s": 4 * Be * To / dte, "ms_per_batch": 1e3 * dte / 4, "windows_per_batch": Be, "batches": 4, "precision": args.precision,
                        "mpjpe_mm": rp["mpjpe"], "mpjpe_mm_oracle": oe["mpjpe_mm"], "pck150": rp["analytics"]["pck"], "pck150_oracle": oe["pck150"],
                        "auc": rp["analytics"]["auc"], "auc_oracle": oe["auc"], "windows_checked": int(Xp.shape[0]),
                        "within_bound": abs(rp["mpjpe"] - oe["mpjpe_mm"]) <= 1e3 * PARITY_BOUND_M and abs(rp["analytics"]["pck"] - oe["pck150"]) <= 0.2
                                        and abs(rp["analytics"]["auc"] - oe["auc"]) <= 0.2}
                    log(f"{name} eval+TTA: {4 * Be * To / dte:.0f} poses/s, MPJPE {rp['mpjpe']:.3f} mm (oracle {oe['mpjpe_mm']:.3f})")
                mo._engine = None
                del mo, Xo, yo
                gc.collect()
                torch.cuda.empty_cache()
            except Exception as e:       # noqa: BLE001
                other_cfgs[name] = {"error": f"{type(e).__name__}: {e}"}
        # the reference's own batch sizes at T=243 (hpe/conf/config.yaml:26: 3; hpe/conf/train/mix_ste.yaml:4: 25): what a drop-in user who keeps the
        # recipe gets (the persistent GEMMs then have 49 / 404 row panels for 256 CUs)
        try:
            ms = build_model(args.precision, 25).cuda()
            for Bs in (3, 25):
                reps = (Bs + B - 1) // B
                dts = time_train(ms, X.repeat(reps, 1, 1, 1)[:Bs].contiguous(), y.repeat(reps, 1, 1, 1)[:Bs].contiguous(), 10)
                small_batch[str(Bs)] = {"poses_per_s": Bs * T / dts, "ms_per_step": 1e3 * dts, "steps": 10, "warmup": 1, "precision": args.precision}
                log(f"T={T} B={Bs}: {Bs * T / dts:.0f} poses/s")
            ms._engine = None
            del ms
            torch.cuda.empty_cache()
        except Exception as e:       # noqa: BLE001
            small_batch = {"error": f"{type(e).__name__}: {e}"}
    forms = "" if args.precision != "bf16x3" or args.f16f8 == 0 else f"+f16f8{args.f16f8}" + ("b" if args.f16_backward else "")
    if rank == 0:
        poses_per_s = world * B * T * args.steps / dt
        gf = TRAIN_GFLOP_PER_POSE.get(T, 3.705)
        out = {"metric": "train poses/sec (whole node), H36M T=243 J=17 K=5", "value": poses_per_s, "unit": "poses/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": {"workload": f"H36M lifting T={T} J=17 K={args.hyp} ManiPose full (C=512, depth 8), train step "
                                      f"fwd+WTA loss+bwd+allreduce+Adam", "windows_per_gpu": B, "global_batch": world * B,
                          "seq_len": T, "parallelism": f"dp{world}", "precision": args.precision, "drop_path_rate": 0.1,
                          **({"split_forms": (("qkv, proj, fc1, fc2: f16f8 (fp16 plane x fp16 plane + ONE block-scaled e4m3 product of 8-bit correction planes per 64 "
                                               "reduction indices; hand-scheduled k-steps); attention: bf16x3 (three bf16 products of bf16 hi/lo planes), its output "
                                               "written as f16f8 planes; bf16 backward (the fp16 planes rounded to bf16 where it reads them)") if args.f16f8 == 3 else
                                              ("qkv, fc1" + (", fc2" if args.f16f8 >= 2 else "") + ": f16f8 (fp16 hi plane x fp16 hi plane + ONE block-scaled e4m3 "
                                               "product of 8-bit correction planes per 64 reduction indices); " + ("proj" if args.f16f8 >= 2 else "proj, fc2")
                                               + ", attention: bf16x3 (three bf16 products of bf16 hi/lo planes)"
                                               + ("; backward of the f16f8 layers on saturating fp16 operands with a per-backward device-side power-of-two "
                                                  "gradient scale" if args.f16_backward else "; bf16 backward"))
                                              if args.f16f8 >= 1 else "all: bf16x3 (three bf16 products of bf16 hi/lo planes), bf16 backward")}
                             if args.precision == "bf16x3" else {}),
                          **({"single_queue": True} if args.single_queue else {}), **({"options": args.option} if args.option else {}),
                          **({"traffic_key_suffix": forms} if forms else {}),
                          "gradient_exchange": ("none" if world == 1 else ("8 layer buckets overlapped with the backward + remainder" if args.grad_buckets
                                                                          else "one all-reduce of the flat buffer (137.8 MB) behind the backward"))},
               "loss": loss, "model_tflops": poses_per_s * gf / 1e3, "workspace_gib": workspace_gib}
        if parity is not None:
            worst = max(parity["mpjpe_m"], parity["mpjpe_m_small_batch"], parity.get("mpjpe_m_worst_rank", 0.0))
            out["parity"] = dict(parity, precision=args.precision, bound_m=PARITY_BOUND_M, within_bound=worst <= PARITY_BOUND_M,
                                 sample=f"T={T} K={args.hyp}: {npar} parity windows embedded in an eval-mode forward of the TIMED batch "
                                        f"(B={B} windows: the persistent split-precision GEMMs and their recomputed-LayerNorm residual epilogue "
                                        f"run, as in the timed steps) on the bench model's initial weights (seed 42), and the same windows as a "
                                        f"stand-alone B={npar} batch (mpjpe_m_small_batch), vs the fp32 CPU oracle "
                                        f"(oracle/manipose_ref.py) run in a CPU child process of this job")
        if rccl is not None:
            out["rccl"] = rccl
        if exchange is not None:
            out["gradient_exchange"] = dict(exchange, note="device events on the caller's stream, max over ranks: backward_ms = mp_model_backward alone "
                                            "(bucketed: with the per-layer all-reduces running beside it on a communication stream), exposed_exchange_ms = the "
                                            "wait between the end of the backward and the optimizer step; `value` is timed in the mode with timed_mode = true")
        if other:
            out["other_precisions"] = other
        if other_cfgs:
            out["other_configs"] = other_cfgs
        if small_batch:
            out["small_batch"] = dict(small_batch, note="train poses/s of the headline workload at the reference's own batch sizes (conf/config.yaml:26, conf/train/mix_ste.yaml:4)") \
                if "error" not in small_batch else small_batch
        if prof is not None:
            # dominant kernel = the one with the largest share of the step: gemm_bf16_persist_kernel (forward + dgrad Linear GEMMs,
            # ~45 % of the kernel time) in the bf16 mode, the fp32 MFMA GEMM of the forward in the fp32 mode
            sub = prof.pop("gemm_persist")
            if args.precision != "fp32" and sub["launches"] > 0:
                k = sub
                kname = ("gemm_bf16_persist_kernel (forward + dgrad Linear GEMMs: persistent, direct-to-LDS 256x256x64 tiles, "
                         "v_mfma_f32_16x16x32_bf16; all instantiations" +
                         ("; the split-precision forward instantiations read two planes per operand and issue 3 bf16 products per k-tile"
                          + ((" - or, f16f8 = 3, ALL forward launches issue one fp16 + one double-depth fp8 product (v_mfma_f32_16x16x32_f16 + "
                              "v_mfma_scale_f32_16x16x128_f8f6f4)") if args.f16f8 == 3 else
                             (" (proj, fc2) or one fp16 + one double-depth fp8 product (qkv, fc1: v_mfma_f32_16x16x32_f16 + "
                              "v_mfma_scale_f32_16x16x128_f8f6f4)" if args.f16f8 >= 1 else ""))
                          + "; flops = 2 M N K, bytes = both planes)" if args.precision == "bf16x3" else ")"))
            else:
                k = prof["gemm_fwd"]
                kname = ("gemm_bf16_glds_kernel (forward Linear GEMMs)" if args.precision != "fp32"
                         else "gemm_f32_kernel<AL=0,BL=0,*> (forward Linear GEMMs, v_mfma_f32_32x32x2_f32)")
            # Roofline of the dominant kernel on ALGORITHMIC work: flops = 2 M N K of the mathematical products, bytes = every
            # operand read once + every output written once.  SURVEY 8d classes the Linear GEMMs as MFMA-bound (weights reused across all
            # tokens), so `achieved` / `frac` are model flops / time against the dense bf16 matrix peak; the split-precision forward ISSUES
            # three bf16 products per product, reported separately as `mfma_issue_*` (its effective roof for fp32-grade products is
            # peak / 3), and the bandwidth view of the same launches as `hbm_*`.
            sec = k["ms"] * 1e-3
            issued_tflops = k["flops"] / sec / 1e12 if sec > 0 else 0.0
            model_tflops = k["model_flops"] / sec / 1e12 if sec > 0 else 0.0
            gbps = k["bytes"] / sec / 1e9 if sec > 0 else 0.0
            peak_tf, peak_bw = PEAK_TFLOPS[args.precision], PEAK_HBM_GBPS
            nl = max(1, k["launches"])
            ridge = peak_tf * 1e12 / (peak_bw * 1e9)

            def roof(flops, nbytes, ms):
                """Roofline position of a set of launches on ALGORITHMIC work: bound by intensity against the ridge (not a literal), the
                fraction of the dense matrix peak and of the roof the model allows at that intensity, min(peak, intensity x 8 TB/s)."""
                sec_ = ms * 1e-3
                inten = flops / nbytes if nbytes > 0 else None
                tf = flops / sec_ / 1e12 if sec_ > 0 else 0.0
                attainable = min(peak_tf, inten * peak_bw / 1e3) if inten is not None else peak_tf
                return {"bound": "mfma" if (inten is None or inten >= ridge) else "hbm", "tflops": tf, "frac": tf / peak_tf,
                        "attainable_tflops": attainable, "frac_attainable": tf / attainable if attainable > 0 else None,
                        "intensity_flop_per_byte": inten, "hbm_gbps": nbytes / sec_ / 1e9 if sec_ > 0 else 0.0}
            r = roof(k["model_flops"], k["bytes"], k["ms"])
            out["roofline"] = {"bound": r["bound"], "kernel": kname,
                               "achieved": model_tflops, "peak": peak_tf, "unit": "TFLOP/s", "frac": model_tflops / peak_tf,
                               "attainable": r["attainable_tflops"], "frac_attainable": r["frac_attainable"],
                               "traffic": pmc_traffic_per_launch(args.precision, B, forms),
                               "avg_launch_ms": k["ms"] / nl, "launches": k["launches"],
                               "flops_per_launch": k["model_flops"] / nl, "bytes_per_launch": k["bytes"] / nl,
                               "intensity_flop_per_byte": r["intensity_flop_per_byte"],
                               "ridge_flop_per_byte": ridge,
                               "hbm_gbps": gbps, "hbm_frac": gbps / peak_bw,
                               "mfma_issue_tflops": issued_tflops, "mfma_issue_frac": issued_tflops / peak_tf,
                               # what register-only bf16 MFMA loops sustain on toggling operands under this board's 1400 W cap (tools/probes/mfma_power.hip,
                               # profiles/r05_probes/power_cap.log): the matrix roof that real data can reach; `peak` above stays the guide's dense figure
                               "mfma_sustained_tflops_under_power_cap": 2070.0, "mfma_issue_frac_of_sustained": issued_tflops / 2070.0,
                               "issued_flops_per_launch": k["flops"] / nl,
                               "note": ("`bound` = intensity (2 M N K / algorithmic bytes, SURVEY 8d) against the ridge 2500 TF / 8 TB/s; `frac` = achieved / "
                                        "dense bf16 matrix peak, `frac_attainable` = achieved / min(peak, intensity x 8 TB/s). All instantiations of the "
                                        "kernel are averaged here: `per_instantiation` splits them."
                                        + (" bf16x3: a forward launch issues 3 bf16 matrix-core products per product (6 M N K, mfma_issue_*; f16f8 "
                                           "launches: one fp16 + one double-rate fp8 product = 4 M N K of bf16-rate issue), so fp32-grade products have an "
                                           "effective matrix roof of peak / 3 (peak / 2); dgrad launches are plain bf16." if args.precision == "bf16x3" else ""))}
            if kinds:
                # the same accounting per instantiation (rotations net; the weight-gradient rows run gemm_bf16_glds_kernel and carry no byte count)
                tab = {}
                for name, v in kinds.items():
                    if not name.startswith("rot.") or v["launches"] == 0:
                        continue
                    nlk = v["launches"]
                    e = {"launches_per_step": nlk / args.steps, "persistent_kernel": v["persist_launches"] == nlk, "ms_per_launch": v["ms"] / nlk,
                         "ms_per_step": v["ms"] / args.steps, "flops_2mnk_per_launch": v["model_flops"] / nlk,
                         "issued_flops_per_launch": v["flops"] / nlk}
                    if v["bytes"] > 0:
                        rr = roof(v["model_flops"], v["bytes"], v["ms"])
                        e.update({"algorithmic_bytes_per_launch": v["bytes"] / nlk, "bound": rr["bound"], "tflops": rr["tflops"], "frac": rr["frac"],
                                  "frac_attainable": rr["frac_attainable"], "hbm_gbps": rr["hbm_gbps"]})
                    else:
                        tf = v["model_flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
                        e.update({"tflops": tf, "frac": tf / peak_tf})
                    tab[name[4:]] = e
                out["roofline"]["per_instantiation"] = tab
                out["roofline"]["per_instantiation_note"] = ("HIP-event times inside the timed three-queue steps: the *.wgrad rows run on the weight-gradient queue, which "
                                                             "time-slices with the main queue - their ms_per_launch is queue time, not kernel duration (one-queue profile: "
                                                             "profiles/*_single_queue_kernel_stats.csv; isolated: tools/wgrad_bench.py)")
            # kernel classes: `isolated_ms_per_step` / `share` from the one-queue pass (a kernel's own duration; they sum to that pass's step
            # time); `event_ms_three_queues` = HIP-event time inside the timed steps, where the engine's three queues time-slice (a class's
            # events then cover the other queues' kernels too: these sum to MORE than the step and rank the classes wrongly)
            kc = {}
            iso = isolated["classes_ms_per_step"] if isolated else None
            tot_iso = sum(iso.values()) if iso else 0.0
            for n, v in prof.items():
                e = {"event_ms_three_queues": v["ms"] / args.steps}
                if iso is not None:
                    e["isolated_ms_per_step"] = iso[n]
                    e["share"] = iso[n] / tot_iso if tot_iso else 0.0
                    e["tflops"] = (v["flops"] / args.steps / (iso[n] * 1e-3) / 1e12) if iso[n] > 0 else 0.0
                elif args.single_queue:
                    tot = sum(q["ms"] for q in prof.values())
                    e = {"isolated_ms_per_step": v["ms"] / args.steps, "share": v["ms"] / tot if tot else 0.0,
                         "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0}
                kc[n] = e
            out["kernel_classes"] = kc
            if power:
                out["power"] = power
            if isolated:
                out["kernel_classes_basis"] = {"isolated_sum_ms_per_step": isolated["classes_sum_ms_per_step"], "one_queue_ms_per_step": isolated["ms_per_step"],
                                               "one_queue_steps": isolated["steps"], "three_queue_ms_per_step": 1e3 * dt / args.steps,
                                               "note": "isolated_ms_per_step: every kernel on one queue (mp_model_set_streams on the timed model, steps run "
                                                       "behind the timed region); the classes sum to the one-queue step; the timed value uses three queues"}
        if cpu_json is not None:
            out["cpu_baseline"] = cpu_json
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
