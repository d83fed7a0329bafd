"""GPU tests that start child processes: the bench's self-spawned two-rank path, the bucketed gradient exchange, the single-rank RCCL
check, the entry point's sharded evaluation / two-rank training, the diagnostics-switch check.  The file name sorts it behind every other
test file on purpose: a harness problem of a child process (ports, pipes, a stuck rendezvous) must not keep ``pytest -x`` from the kernel
and model parity tests of test_gpu_parity.py.

Conventions for every test here: rendezvous ports are ephemeral (`free_port`), a rank reports through ONE file per rank under pytest's
tmp_path (never through words interleaved on the launcher's shared pipe), children never outlive their timeout.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MPJPE_TOL_M = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher(port, nproc=2):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_bench_two_rank_path_rehearsal_on_one_gpu(lib):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py spawns torch.distributed.run itself (before touching a GPU); the two
    ranks are pinned to this GPU with gloo carrying the collectives (a one-GPU box cannot host two RCCL ranks): init, barrier, gradient
    all-reduce, max-over-ranks timing, one JSON line from rank 0 with the in-run parity block."""
    root = ROOT
    env = clean_env(MANIPOSE_BENCH_DEVICE="0", MANIPOSE_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--frames", "27"], capture_output=True, text=True, timeout=400, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert "cpu_baseline" not in d and d["roofline"]["launches"] >= 0
    assert d["dtype"] == "bf16x3" and d["parity"]["within_bound"] and d["parity"]["mpjpe_m"] <= 1e-4, d.get("parity")
    assert "qkv, proj, fc1, fc2: f16f8" in d["config"]["split_forms"] and "bf16 backward" in d["config"]["split_forms"]      # the default operand form since round 6 (--f16f8 3)
    rf = d["roofline"]
    assert rf["bound"] == ("mfma" if rf["intensity_flop_per_byte"] is None or rf["intensity_flop_per_byte"] >= rf["ridge_flop_per_byte"] else "hbm")
    assert 0 < rf["frac"] <= rf["frac_attainable"] <= 1.0 and isinstance(rf["per_instantiation"], dict)
    assert rf["traffic"] is None or "source" in rf["traffic"]
    # N > 1: the run answers by itself what the two exchange modes cost (device events around the backward and the collective)
    ex = d["gradient_exchange"]
    assert set(ex) >= {"single", "bucketed"} and ex["single"]["timed_mode"] and not ex["bucketed"]["timed_mode"]
    assert all(ex[m]["backward_ms"] > 0 and ex[m]["exposed_exchange_ms"] >= 0 for m in ("single", "bucketed"))


def test_bucketed_gradient_exchange_equals_the_single_all_reduce(lib, tmp_path):
    """LiftingTrainer(grad_buckets=True): one all-reduce per layer of the rotations net on a communication stream, each behind a
    device-side wait for that layer's gradients of the running backward (mp_model_grad_bucket_wait), the rest behind the whole backward
    - against the single all-reduce of the flat buffer.  Two ranks on this GPU with gloo carrying the collectives (a one-GPU box cannot
    host two RCCL ranks): after three training steps on different per-rank batches every rank must hold the same parameter BITS in both
    modes, and the two ranks must agree with each other."""
    root = ROOT
    code = (
        "import json, os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton\n"
        "from manipose_amd.distributed import broadcast_parameters\n"
        "from manipose_amd.training import LiftingTrainer\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('gloo')\n"
        "rank = dist.get_rank()\n"
        "outs = []\n"
        "for buckets in (False, True):\n"
        "    torch.manual_seed(3)\n"
        "    m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=27, embed_dim_rot=128, depth_rot=3, num_heads_rot=8, embed_dim_seg=64, depth_seg=2,\n"
        "                           num_heads_seg=4, n_hyp=3, drop_path_rate=0.1)\n"
        "    m.precision = 'bf16x3'\n"
        "    m = m.cuda().train()\n"
        "    broadcast_parameters(m.flat_parameters())\n"
        "    tr = LiftingTrainer(m, lr=1e-3, weight_decay=1e-6, seed=5, grad_buckets=buckets)\n"
        "    g = torch.Generator(device='cuda').manual_seed(10 + rank)\n"
        "    X = (0.3 * torch.randn(4, 27, 17, 2, device='cuda', generator=g)).clamp(-1, 1)\n"
        "    y = 0.3 * torch.randn(4, 27, 17, 3, device='cuda', generator=g)\n"
        "    for _ in range(3):\n"
        "        tr.train_step(X, y)\n"
        "    torch.cuda.synchronize()\n"
        "    outs.append(m.flat_parameters().clone())\n"
        "same_modes = bool(torch.equal(outs[0], outs[1]))\n"
        "other = outs[1].clone()\n"
        "dist.broadcast(other, src=0)\n"
        "rep = dict(rank=rank, modes_equal=same_modes, ranks_equal=bool(torch.equal(other, outs[1])), finite=bool(torch.isfinite(outs[1]).all()))\n"
        "with open(os.path.join(os.environ['MANIPOSE_TEST_REPORT_DIR'], f'rank{rank}.json'), 'w') as f:\n"
        "    json.dump(rep, f)\n"
        "sys.stdout.write(json.dumps(rep) + '\\n'); sys.stdout.flush()\n"
        "dist.destroy_process_group()\n")
    script = str(tmp_path / "_bucket_ranks.py")            # pytest's own scratch directory: nothing is written into the repository
    with open(script, "w") as f:
        f.write(code)
    r = subprocess.run(launcher(free_port()) + [script], capture_output=True, text=True, timeout=400,
                       env=clean_env(MANIPOSE_TEST_REPORT_DIR=str(tmp_path)), cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    # one report FILE per rank: the ranks share the launcher's stdout pipe, and the words of two simultaneous print() calls interleave there
    # (GPUTEST_r04: 'rankrank  10  modes_equalmodes_equal ...' with every value True)
    for rank in (0, 1):
        with open(tmp_path / f"rank{rank}.json") as f:
            rep = json.load(f)
        assert rep == {"rank": rank, "modes_equal": True, "ranks_equal": True, "finite": True}, (rep, r.stdout[-1000:])


def test_rccl_backend_single_rank_collectives_on_the_flat_gradient_buffer(lib):
    """backend="nccl" (= RCCL) on this GPU with world size 1: init with device_id, broadcast of the flat parameter buffer, SUM
    all-reduce of a flat gradient buffer of the real size, barrier, destroy - the calls training.py / distributed.py make per step."""
    root = ROOT
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "from manipose_amd.distributed import broadcast_parameters\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "g = torch.full((34_450_000,), 2.0, device='cuda')\n"
        "dist.all_reduce(g, op=dist.ReduceOp.SUM)\n"
        "broadcast_parameters(g, 0)\n"
        "dist.broadcast(g, src=0)\n"
        "dist.barrier()\n"
        "torch.cuda.synchronize()\n"
        "assert float(g.sum().item()) == 2.0 * g.numel()\n"
        "dist.destroy_process_group()\n"
        "print('rccl ok', dist.is_nccl_available())\n")
    # an ephemeral rendezvous port (other tests of this session start torch.distributed.run children; a fixed port can meet a
    # listener that has not gone away yet); NCCL_DEBUG=INFO so that a failure carries RCCL's own account of where it stopped.
    # HSA_ENABLE_IPC_MODE_LEGACY=0 is the pool's contract for multi-process GPU work (the host driver only supports dmabuf IPC; without it
    # RCCL's hipIpcGetMemHandle fails): already exported on the box, repeated so the child has it whatever the caller's environment.
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO")
    # bounded wait with our own kill of the child's process group (a child stuck inside the RCCL bootstrap does not always die on
    # subprocess.run's timeout and would take the rest of the suite with it).  A child that does not finish is a FAILURE of this test,
    # reported with the tail of its output - the one RCCL check of the suite must not turn into a skip when it matters.
    import signal, tempfile, time
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        p = subprocess.Popen([sys.executable, "-c", code], stdout=out, stderr=err, text=True, env=env, cwd=root, start_new_session=True)
        t0 = time.time()
        while p.poll() is None and time.time() - t0 < 180:
            time.sleep(0.5)
        timed_out = p.poll() is None
        if timed_out:
            os.killpg(p.pid, signal.SIGKILL)
            t1 = time.time()
            while p.poll() is None and time.time() - t1 < 15:
                time.sleep(0.5)
        out.seek(0); err.seek(0)
        so, se = out.read(), err.read()
    log_dir = os.environ.get("MANIPOSE_TEST_LOG_DIR")      # (optional) keep RCCL's log of this run; the test writes nothing into the repo by itself
    if log_dir and os.path.isdir(log_dir):
        with open(os.path.join(log_dir, "rccl_single_rank.log"), "w") as f:
            f.write(so + "\n--- stderr ---\n" + se)
    assert not timed_out, "the single-rank RCCL process did not finish within 180 s (killed); its output:\n" + so[-500:] + "\n" + se[-3000:]
    assert p.returncode == 0 and "rccl ok True" in so, (so[-500:], se[-2000:])


def test_sharded_evaluation_two_ranks_equals_single_process(lib, raw_dataset_dir, tmp_path):
    """The entry point's test pass with the windows dealt over two ranks (both on this GPU, gloo carrying the sum-reduce of the error sums /
    frame counts and the gather of the shifted variance sums) prints the tables a single process prints."""
    import re
    root = ROOT
    d, _ = raw_dataset_dir
    args = ["run.train=false", "run.test=true", "train.batch_size_test=3", "data.seq_len=9", f"data.data_dir={d}", "data.keypoints=gt",
            "model.channels=64", "model.layers=2", "model.nheads=4", "model.channels_seg=32", "model.layers_seg=1", "model.nheads_seg=4",
            "multi_hyp.n_hyp=3", "model.precision=fp32", "train.flip_aug=false"]     # (random flips of the evaluation windows - which the
    # reference's loader also draws - swap left and right bones, so the per-bone consistency metric depends on the RNG stream)
    env = dict(os.environ, MANIPOSE_DEVICE="0", MANIPOSE_DIST_BACKEND="gloo")
    one = subprocess.run([sys.executable, os.path.join(root, "hpe", "main_h36m_lifting.py")] + args, capture_output=True, text=True,
                         timeout=300, cwd=str(tmp_path), env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run(launcher(free_port()) + [os.path.join(root, "hpe", "main_h36m_lifting.py")] + args, capture_output=True, text=True,
                         timeout=300, cwd=str(tmp_path), env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    num = lambda out: [[float(x) for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", l)] for l in out.splitlines() if l.startswith("test")]
    a, b = num(one.stdout), num(two.stdout)
    assert len(a) == len(b) >= 5 and all(len(x) == len(y) and len(x) > 0 for x, y in zip(a, b)), (one.stdout[-1500:], two.stdout[-1500:])
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, rtol=2e-4, atol=2e-3)
    # and one epoch of two-rank training on the same files: shuffled windows dealt over the ranks (padded to equal step counts), the
    # gradient all-reduce in every step, the summed validation loss, the sharded MPJPE evaluation
    targs = [a for a in args if not a.startswith("run.train")] + ["run.train=true", "train.epochs=1", "train.batch_size=4", "run.test=false",
                                                                  "train.mpjpe_epoch_interval=1", "run.experiment=two_rank"]
    tr = subprocess.run(launcher(free_port()) + [os.path.join(root, "hpe", "main_h36m_lifting.py")] + targs, capture_output=True, text=True,
                        timeout=300, cwd=str(tmp_path), env=env)
    assert tr.returncode == 0, tr.stderr[-2000:]
    assert "epoch 0:" in tr.stdout and "eval:" in tr.stdout and os.path.exists(os.path.join(str(tmp_path), "two_rank", "model_end.pth"))


def test_product_library_ignores_the_timing_ablation_switches(lib):
    """MANIPOSE_GEMM_DEBUG / MANIPOSE_GEMM_ABL / MANIPOSE_ATTN_DEBUG switch wrong-by-design timing ablations on in the DIAGNOSTICS build
    only (MP_DIAG=1 build.sh); with all of them set, a child process using the product library must still produce the oracle's numbers."""
    root = ROOT
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'oracle')!r})\n"
        "import manipose_ref as orc\n"
        "from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton\n"
        "cfg = dict(T=27, J=17, num_bones=16, C_rot=256, depth_rot=2, heads_rot=4, C_seg=128, depth_seg=1, heads_seg=8, n_hyp=2)\n"
        "st = orc.make_state(cfg, seed=2)\n"
        "for prec in ('bf16x3', 'bf16'):\n"
        "    m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=27, embed_dim_rot=256, depth_rot=2, num_heads_rot=4, embed_dim_seg=128, depth_seg=1,\n"
        "                           num_heads_seg=8, n_hyp=2, drop_path_rate=0.0)\n"
        "    m.load_state_dict(st, strict=True); m.precision = prec; m = m.cuda().eval()\n"
        "    X, y = orc.synthetic_batch(4, 27, seed=3)\n"
        "    p, s = m(X.cuda())\n"
        "    (p.square().sum() + s.square().sum()).backward()\n"
        "    g = torch.cat([q.grad.reshape(-1) for q in m.parameters()])\n"
        "    with torch.no_grad():\n"
        "        op, _ = orc.rmcl_manifold_forward(X, st, orc.oracle_cfg(cfg))\n"
        "    print(prec, float((p.detach().cpu() - op).norm(dim=-1).mean()), bool(torch.isfinite(g).all()), float(g.abs().sum()))\n")
    env = dict(os.environ, MANIPOSE_GEMM_DEBUG="7", MANIPOSE_GEMM_ABL="3", MANIPOSE_ATTN_DEBUG="3")
    env.pop("MANIPOSE_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l.startswith("bf16")}
    assert float(rows["bf16x3"][0]) <= MPJPE_TOL_M and float(rows["bf16"][0]) <= 8e-3, rows
    assert rows["bf16x3"][1] == "True" and rows["bf16"][1] == "True" and float(rows["bf16x3"][2]) > 0, rows
