"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the parameter layout and
initialisation match the reference, the product refuses to compute on CPU, and the 2-rank data-parallel exchange
(gloo) reproduces the single-process gradient."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import manipose_ref as orc
from helpers import fixture_state, load_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model(fx):
    from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
    c = fx["cfg"]
    m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=c["T"], embed_dim_rot=c["C_rot"], depth_rot=c["depth_rot"],
                           num_heads_rot=c["heads_rot"], embed_dim_seg=c["C_seg"], depth_seg=c["depth_seg"],
                           num_heads_seg=c["heads_seg"], n_hyp=c["n_hyp"], drop_path_rate=0.0)
    m.load_state_dict(fixture_state(fx), strict=True)
    return m


def test_library_exports_every_declared_symbol():
    from manipose_amd import _lib
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 30 and "mp_model_forward" in names and "mp_fk_decode_fwd" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/manipose_hip.h but not exported"
        assert n in _lib._SIGNATURES, f"{n} has no ctypes signature in manipose_amd/_lib.py"
    assert lib.mp_abi_version() == _lib.ABI_VERSION == 8


def test_state_dict_layout_matches_reference(golden_dir):
    from manipose_amd import ManifoldMixSTE, RMCLManifoldMixSTE, h36m_skeleton
    m = RMCLManifoldMixSTE(h36m_skeleton())
    got = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in m.state_dict().items())
    assert got == open(golden_dir + "/state_dict_keys_T243_K5.txt").read().split()
    z = np.load(golden_dir + "/param_counts.npz")
    assert sum(p.numel() for p in m.parameters()) == int(z["rmcl_T243_K5"]) == 34440062
    m27 = ManifoldMixSTE(h36m_skeleton(), num_frame=27)
    assert sum(p.numel() for p in m27.parameters()) == int(z["manifold_T27"])
    # engine layout (flat buffer) covers exactly the state-dict, 16-byte aligned, non-overlapping
    lay = m.flat_layout()
    assert {n for n, _, _ in lay} == set(m.state_dict())
    ends = 0
    for n, off, num in lay:
        assert off % 4 == 0 and off >= ends
        ends = off + num
    flat = m.flat_parameters()
    name, off, num = lay[5]
    assert torch.equal(flat[off:off + num], dict(m.named_parameters())[name].detach().reshape(-1))
    assert isinstance(m, RMCLManifoldMixSTE) and isinstance(m, ManifoldMixSTE)     # reference isinstance dispatch


def test_default_init_consumes_rng_like_reference(golden_dir):
    """Same seed -> same initial weights as the reference constructor (fixture made by oracle/gen_golden.py)."""
    from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
    z = np.load(golden_dir + "/init_seed42_small.npz")
    torch.manual_seed(42)
    m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=27, embed_dim_rot=64, depth_rot=2, embed_dim_seg=32, depth_seg=2,
                           n_hyp=5, drop_path_rate=0.1)
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), z[k], err_msg=k)


def test_no_cpu_fallback_and_argument_errors():
    from manipose_amd import _lib
    from manipose_amd.metrics import rmcl_training_loss
    fx = load_fixture("rmcl_tiny")
    model = _tiny_model(fx)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.from_numpy(fx["X"]))
    with pytest.raises(AssertionError):
        model(torch.zeros(2, 5, 17, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rmcl_training_loss(torch.from_numpy(fx["poses"]), torch.from_numpy(fx["scores"]), torch.from_numpy(fx["y"]))
    lib = _lib.load()
    rc = lib.mp_fk_decode_fwd(None, 6, 6, None, None, 1, 1, 1, None)       # argument validation happens before any launch
    assert rc == 1 and b"null" in lib.mp_last_error()
    cfg = _lib.ModelConfig(arch=7, num_frame=9, num_joints=17, num_bones=16, embed_dim_rot=32, depth_rot=1,
                           num_heads_rot=4, embed_dim_seg=16, depth_seg=1, num_heads_seg=4, n_hyp=3, max_batch=0)
    import ctypes as C
    h = C.c_void_p()
    assert lib.mp_model_create(C.byref(cfg), C.byref(h)) == 1 and b"arch" in lib.mp_last_error()


def test_skeleton_tables():
    from manipose_amd.data import h36m_skeleton
    sk = h36m_skeleton()
    assert sk.num_joints == 17 and sk.num_bones == 16
    assert list(sk.has_children.astype(int)) == [1, 1, 1, 0, 1, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0]
    assert sk.bones_left == (3, 4, 5, 10, 11, 12) and sk.bones_right == (0, 1, 2, 13, 14, 15)
    assert sk.bones[0] == (1, 0) and sk.children[8] == [9, 11, 14]


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    from manipose_amd.distributed import allreduce_gradients, broadcast_parameters, init_from_env, shard_windows
    r, w, _ = init_from_env("gloo")
    fx = load_fixture("rmcl_tiny")
    model = _tiny_model(fx)
    flat = model.flat_parameters()                       # CPU layout-only engine
    if r != 0:
        flat.zero_()                                     # prove the broadcast brings rank 0's weights
    broadcast_parameters(flat)
    X, y = orc.synthetic_batch(4, fx["cfg"]["T"], seed=5)
    idx = list(shard_windows(4, r, w))
    st = {k: p for k, p in model.named_parameters()}    # views of the flat buffer
    poses, scores = orc.rmcl_manifold_forward(X[idx], st, orc.oracle_cfg(fx["cfg"]))   # oracle = stand-in compute on CPU
    total, _ = orc.rmcl_training_loss(poses, scores, y[idx])
    total.backward()
    g = torch.zeros_like(flat)
    for (off, n), p in zip(model._slots, model._plist):
        g[off:off + n] = p.grad.reshape(-1)
    allreduce_gradients(g)
    if r == 0:
        torch.save({"g": g / w, "layout": model.flat_layout()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_full_batch(tmp_path):
    """world_size-2 gloo run of the product's exchange path (flat layout + broadcast + SUM all-reduce + 1/world):
    averaged shard gradients == gradient of the concatenated batch."""
    out = str(tmp_path / "dp.pt")
    port = 29600 + os.getpid() % 200
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    fx = load_fixture("rmcl_tiny")
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    X, y = orc.synthetic_batch(4, fx["cfg"]["T"], seed=5)
    poses, scores = orc.rmcl_manifold_forward(X, st, orc.oracle_cfg(fx["cfg"]))
    orc.rmcl_training_loss(poses, scores, y)[0].backward()
    for name, off, n in res["layout"]:
        np.testing.assert_allclose(res["g"][off:off + n].numpy(), st[name].grad.reshape(-1).numpy(), rtol=2e-4, atol=2e-6,
                                   err_msg=name)


def test_pose_flip_semantics():
    """Flip = negate x + swap left/right joints, in place, tuple in/out (reference augmentations/functional.py:7-28);
    an involution; aggregate-then-flip == flip-then-aggregate (what the batched flip-TTA relies on)."""
    from manipose_amd.augmentations import pose_flip
    from manipose_amd.data import h36m_skeleton
    sk = h36m_skeleton()
    g = torch.Generator().manual_seed(3)
    p = torch.randn(2, 5, 17, 3, generator=g)
    ref = p.clone()
    ref[..., 0] *= -1
    idx = list(range(17))
    for l, r in zip(sk.joints_left, sk.joints_right):
        idx[l], idx[r] = r, l
    ref = ref[..., idx, :]
    q = p.clone()
    (out,) = pose_flip((q,), sk)
    assert out is q and torch.equal(q, ref)
    pose_flip((q,), sk)
    assert torch.equal(q, p)
    hyp = torch.randn(2, 3, 5, 17, 3, generator=g)
    sc = torch.rand(2, 3, 5, 1, generator=g).softmax(1)
    a = orc.aggregate(pose_flip((hyp.clone(),), sk)[0], sc)
    b = pose_flip((orc.aggregate(hyp, sc),), sk)[0]
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-6)


def test_epoch_batches_deal_every_window_to_exactly_one_rank():
    """hpe/_entry.py::epoch_batches (DistributedSampler-style dealing of the shuffled window indices, drop_last=False)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hpe"))
    from _entry import epoch_batches

    class FakeGen:
        def __len__(self):
            return 37

        def batch(self, idx):
            return list(idx), None

    seen = []
    for rank in range(3):
        got = [b for b, _ in epoch_batches(FakeGen(), 4, shuffle=True, rank=rank, world=3, seed=5)]
        assert all(len(b) == 4 for b in got[:-1]) and 1 <= len(got[-1]) <= 4
        seen += [i for b in got for i in b]
    assert sorted(seen) == list(range(37))
    a = [b for b, _ in epoch_batches(FakeGen(), 4, shuffle=True, rank=1, world=3, seed=5)]
    b = [b for b, _ in epoch_batches(FakeGen(), 4, shuffle=True, rank=1, world=3, seed=6)]
    assert a != b                                                     # a new permutation every epoch
    assert [i for bb, _ in epoch_batches(FakeGen(), 5, shuffle=False) for i in bb] == list(range(37))
    # training: padded to a multiple of the world size, so every rank runs the same number of (collective-holding) steps
    shares = [[b for b, _ in epoch_batches(FakeGen(), 4, shuffle=True, rank=r, world=3, seed=5, pad=True)] for r in range(3)]
    assert len({len(sh) for sh in shares}) == 1 and len({tuple(map(len, sh)) for sh in shares}) == 1
    assert set(i for sh in shares for b in sh for i in b) == set(range(37)) and sum(len(b) for sh in shares for b in sh) == 39


def test_lr_schedulers_follow_torch():
    """CosineAnnealingLR / ReduceLROnPlateau restated for the fused optimizer (manipose_amd/optim.py) against torch's own, configured as
    the reference does (main_h36m_lifting.py:243-264)."""
    from manipose_amd.optim import make_lr_scheduler

    class Opt:
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    def torch_opt(lr):
        return torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=lr)

    mine, ref = Opt(4e-5), torch_opt(4e-5)
    a = make_lr_scheduler(mine, "cosine", epochs=20, n_annealing=2, lr_min=1e-6)
    b = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=10, eta_min=1e-6)
    ref.step()                                                  # torch wants an optimizer step before the first scheduler step
    for _ in range(25):
        a.step(); b.step()
        assert abs(mine.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) <= 1e-12
    mine, ref = Opt(4e-5), torch_opt(4e-5)
    a = make_lr_scheduler(mine, "plateau", epochs=0, lr_min=1e-6, lr_patience=2, lr_threshold=0.1)
    b = torch.optim.lr_scheduler.ReduceLROnPlateau(ref, mode="min", factor=0.5, min_lr=1e-6, patience=2, threshold=0.1)
    g = torch.Generator().manual_seed(0)
    best = 1e10
    for i in range(60):
        val = float(10.0 / (1 + 0.2 * i) + 0.3 * torch.rand(1, generator=g))
        best = min(best, val)
        a.step(best); b.step(best)
        assert abs(mine.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) <= 1e-15, i
    assert mine.param_groups[0]["lr"] < 4e-5
    st = a.state_dict()
    c = make_lr_scheduler(Opt(1.0), "plateau", epochs=0)
    c.load_state_dict(st)
    assert c.best == a.best and c.num_bad_epochs == a.num_bad_epochs
    with pytest.raises(ValueError):
        make_lr_scheduler(mine, "step", epochs=10)


# ---------------------------------------------------------------------------------------------------------------------------------
# boundary documents and entry-point grammar
# ---------------------------------------------------------------------------------------------------------------------------------
def _header_prototypes():
    import re
    text = open(os.path.join(ROOT, "include", "manipose_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for name, args in re.findall(r"\b(mp_\w+)\s*\(([^()]*)\)\s*;", text):
        a = args.strip()
        protos[name] = 0 if a in ("", "void") else a.count(",") + 1
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                fields += [re.sub(r"\[.*\]", "", f.strip().split()[-1].lstrip("*")) for f in decl.split(",")]
        structs[name] = fields
    version = int(re.search(r"#define MP_ABI_VERSION (\d+)", text).group(1))
    return protos, structs, version


def test_integration_doc_matches_the_header():
    """the generated ctypes declarations are up to date with manipose_amd/_lib.py, and every argtypes list, structure
    and `lib.mp_*(...)` call in the document's code blocks agrees with include/manipose_hip.h (argument counts, field lists, ABI)."""
    import ast, re, subprocess
    from manipose_amd import _lib
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration_stub.py"), "--check"]).returncode == 0, \
        "INTEGRATION.md is stale: run python tools/gen_integration_stub.py"
    protos, structs, version = _header_prototypes()
    assert version == _lib.ABI_VERSION
    for name, (_, args) in _lib._SIGNATURES.items():           # the binding itself against the header
        assert protos[name] == len(args), f"{name}: header has {protos[name]} parameters, _lib.py {len(args)}"
    assert [n for n, _ in _lib.ModelConfig._fields_] == structs["mp_model_config"]
    assert [n for n, _ in _lib.LossConfig._fields_] == structs["mp_loss_config"]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    n_argtypes = n_calls = n_structs = 0
    for code in blocks:
        tree = ast.parse(code)
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) and node.targets[0].attr == "argtypes":
                fn = node.targets[0].value.attr
                assert len(node.value.elts) == protos[fn], f"INTEGRATION.md: {fn}.argtypes has {len(node.value.elts)} entries, header {protos[fn]}"
                n_argtypes += 1
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
                    and node.func.value.id == "lib" and node.func.attr.startswith("mp_"):
                assert len(node.args) == protos[node.func.attr], f"INTEGRATION.md: call of {node.func.attr} with {len(node.args)} arguments, header {protos[node.func.attr]}"
                n_calls += 1
            if isinstance(node, ast.ClassDef) and node.name in ("ModelConfig", "LossConfig"):
                fl = [e.elts[0].value for e in node.body[0].value.elts]
                assert fl == structs["mp_model_config" if node.name == "ModelConfig" else "mp_loss_config"]
                n_structs += 1
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "ModelConfig":
                kws = {k.arg for k in node.keywords}                       # unnamed fields are zero = their defaults
                assert kws <= set(structs["mp_model_config"]) and {"arch", "num_frame", "max_batch", "precision", "rot_rep_dim"} <= kws
    assert n_argtypes >= 15 and n_calls >= 6 and n_structs == 2


def test_entry_point_override_grammar_and_readme_commands():
    """The reference's documented command lines (README.md:54-72,100-111) parse; config groups merge; Hydra-style scalars; typos fail."""
    sys.path.insert(0, os.path.join(ROOT, "hpe"))
    from _entry import load_config, parse_value
    c = load_config(["run.checkpoint_model=hpe/checkpoints/manipose_h36m.pth", "run.train=False", "run.test=True", "data.data_dir=/PATH/TO/H36M/DATA/"])
    assert c.run.train is False and c.run.test is True and c.data.data_dir == "/PATH/TO/H36M/DATA/" and c.data.seq_len == 243
    assert c.model.precision == "bf16x3"                               # the entry points default to the precision inside the parity bound
    c = load_config(["run.checkpoint_model=hpe/checkpoints/manipose_3dhp.pth", "+data=mpi_inf_3dhp", "train.batch_size=30", "train.batch_size_test=30",
                     "run.train=False", "run.test=True", "data.data_dir=/PATH/TO/MPI/DATA/"], {"data.dataset": "3dhp"})
    assert c.data.dataset == "3dhp" and c.data.seq_len == 27 and c.data.keypoints == "gt" and c.data.pad == 0 and c.data.out_all is True
    assert c.data.downsample == 1 and c.train.batch_size == 30 and c.data.data_dir == "/PATH/TO/MPI/DATA/"
    c = load_config(["model=small", "train=debug", "train.lr=4e-5", "train.lr_min=1e-6", "train.vel_loss=2.", "data.seq_len=81", "+data.stride=3"])
    assert c.model.channels == 64 and c.model.channels_seg == 64 and c.train.epochs == 1 and c.train.mpjpe_epoch_interval == 1
    assert isinstance(c.train.lr, float) and c.train.lr == 4e-5 and c.train.lr_min == 1e-6 and c.train.vel_loss == 2.0 and c.data.stride == 3
    c = load_config(["train=mix_ste"])
    assert c.train.epochs == 500 and c.train.batch_size == 25 and c.train.workers == 10 and c.train.lr == 0.00004
    assert parse_value("4e-5") == 4e-5 and parse_value("0.") == 0.0 and parse_value("True") is True and parse_value("null") is None
    assert parse_value("cpn_ft_h36m_dbb") == "cpn_ft_h36m_dbb" and parse_value("[1, 2]") == [1, 2] and parse_value("'27'") == "27"
    for bad in (["run.tran=False"], ["run.mlfow_on=True"], ["runs.train=False"], ["+data=no_such_dataset"], ["model=huge"], ["train"], ["+optim=adam"]):
        with pytest.raises(SystemExit):
            load_config(bad)


def test_fused_adam_state_dict_is_interchangeable_with_torch_adam():
    """params{tag}.pth (main_h36m_lifting.py:75-98): FusedAdam emits and accepts torch.optim.Adam's state_dict layout (per-parameter
    exp_avg / exp_avg_sq in model.parameters() order); the flat layout written by earlier versions still loads."""
    from manipose_amd.optim import FusedAdam
    fx = load_fixture("rmcl_tiny")
    m = _tiny_model(fx)
    ref = torch.optim.Adam(m.parameters(), lr=4e-5, weight_decay=1e-6)
    g = torch.Generator().manual_seed(0)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    ref.step(); ref.step()
    sd = ref.state_dict()
    opt = FusedAdam(m, lr=1.0)
    opt.load_state_dict(sd)
    assert opt.step_count == 2 and opt.param_groups[0]["lr"] == 4e-5
    lay = {n: (o, k) for n, o, k in m.flat_layout()}
    for i, (n, p) in enumerate(m.named_parameters()):
        o, k = lay[n]
        assert torch.equal(opt.exp_avg[o:o + k], sd["state"][i]["exp_avg"].reshape(-1)), n
        assert torch.equal(opt.exp_avg_sq[o:o + k], sd["state"][i]["exp_avg_sq"].reshape(-1)), n
    out = opt.state_dict()
    ref2 = torch.optim.Adam(m.parameters(), lr=1.0)
    ref2.load_state_dict(out)                                          # torch accepts what FusedAdam wrote
    assert ref2.state_dict()["param_groups"][0]["lr"] == 4e-5 and len(ref2.state_dict()["state"]) == len(list(m.parameters()))
    for i in range(len(sd["state"])):
        assert torch.equal(ref2.state_dict()["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    legacy = {"step": 7, "exp_avg": opt.exp_avg.clone(), "exp_avg_sq": opt.exp_avg_sq.clone(), "lr": 1e-3}
    opt.load_state_dict(legacy)
    assert opt.step_count == 7 and opt.param_groups[0]["lr"] == 1e-3


def test_fused_adam_loads_a_muadam_state_with_several_param_groups():
    """A params*.pth of a model.mup run of the reference is written by mup.optim.MuAdam: one param group per width multiplier of the
    matrix-like parameters (lr / width_mult, weight_decay * width_mult) followed by the vector-like group, state indices running through
    the groups.  The same grouping built here with torch.optim.Adam stands in for the absent package (parity unpinned for its part)."""
    from manipose_amd.optim import FusedAdam
    fx = load_fixture("mup_rmcl")
    m = _mup_model(fx)
    matrix, vector = {}, []
    for p in m.parameters():
        (matrix.setdefault(p.infshape.width_mult(), []) if p.infshape.ninf() == 2 else vector).append(p)
    assert len(matrix) == 2 and vector                                  # rot (x2) and seg (x4) width multipliers
    groups = [{"params": ps, "lr": 4e-5 / wm, "weight_decay": 1e-6 * wm} for wm, ps in matrix.items()] + [{"params": vector, "lr": 4e-5, "weight_decay": 1e-6}]
    ref = torch.optim.Adam(groups)
    g = torch.Generator().manual_seed(1)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    ref.step(); ref.step(); ref.step()
    sd = ref.state_dict()
    assert len(sd["param_groups"]) == 3
    opt = FusedAdam(m, lr=1.0)
    opt.load_state_dict(sd)
    assert opt.step_count == 3 and opt.param_groups[0]["lr"] == 4e-5
    lay = {n: (o, k) for n, o, k in m.flat_layout()}
    name_of = {id(p): n for n, p in m.named_parameters()}
    order = [p for grp in groups for p in grp["params"]]
    for i, p in enumerate(order):
        o, k = lay[name_of[id(p)]]
        assert torch.equal(opt.exp_avg[o:o + k], sd["state"][i]["exp_avg"].reshape(-1)), name_of[id(p)]
        assert torch.equal(opt.exp_avg_sq[o:o + k], sd["state"][i]["exp_avg_sq"].reshape(-1)), name_of[id(p)]
    from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
    plain = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=9, embed_dim_rot=32, depth_rot=1, num_heads_rot=4, embed_dim_seg=16, depth_seg=1, num_heads_seg=4, n_hyp=2)
    with pytest.raises(ValueError, match="base shapes"):
        FusedAdam(plain).load_state_dict(sd)


def test_mup_checkpoint_is_loaded_after_the_base_shapes_and_used_untouched(tmp_path, monkeypatch):
    """create_model / set_mup_base_shapes run BEFORE load_state_dict in the reference (main_h36m_lifting.py:673-708, :754-764): the
    rescaling of freshly initialised MuReadout weights by sqrt(width_mult) must never touch checkpoint weights, and mu_init_params only
    runs without a checkpoint.  The entry point's model-construction part is driven up to the first device call."""
    sys.path.insert(0, os.path.join(ROOT, "hpe"))
    import _entry
    import manipose_amd.mup_lite as ml
    overrides = ["model.mup=true", "model.channels=128", "model.channels_seg=64", "model.layers=1", "model.layers_seg=1", "data.seq_len=9",
                 "multi_hyp.n_hyp=2", "run.seed=3"]
    cfg = _entry.load_config(overrides, None)
    torch.manual_seed(3)
    src = _entry.instantiate_model(cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.uniform_(-0.5, 0.5)
    want = {k: v.clone() for k, v in src.state_dict().items()}
    ck = tmp_path / "mup_model.pth"
    torch.save({"model_pos": want}, ck)
    seen = {}

    class Stop(Exception):
        pass

    def fake_to(self, *a, **k):                 # first device call of run(): capture the model as constructed, then leave
        seen["state"] = {k_: v.detach().clone() for k_, v in self.state_dict().items()}
        seen["infshapes"] = all(getattr(p, "infshape", None) is not None for p in self.parameters())
        raise Stop()
    calls = []
    real_init = ml.mu_init_params
    monkeypatch.setattr(ml, "mu_init_params", lambda mdl: (calls.append(1), real_init(mdl)))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.nn.Module, "to", fake_to)
    with pytest.raises(Stop):
        _entry.run(overrides + [f"run.checkpoint_model={ck}"])
    assert seen["infshapes"] and not calls
    assert all(torch.equal(seen["state"][k], want[k]) for k in want), [k for k in want if not torch.equal(seen["state"][k], want[k])][:3]
    with pytest.raises(Stop):                   # without a checkpoint: muP re-initialisation runs (once)
        _entry.run(overrides)
    assert calls == [1]


def test_gradient_buckets_cover_the_layers_of_the_rotations_net():
    """mp_model_grad_bucket_info: bucket i is exactly the contiguous flat-buffer range of STEblocks.i and TTEblocks.i (24 tensors), the
    buckets are disjoint and everything else (embeddings, positional tables, shared norms, heads, segments net) lies outside them."""
    from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
    m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=27, embed_dim_rot=64, depth_rot=3, num_heads_rot=4, embed_dim_seg=32, depth_seg=2,
                           num_heads_seg=4, n_hyp=2)
    m.flat_parameters()
    buckets = m._engine.grad_buckets()
    assert len(buckets) == 3
    inside = [[] for _ in buckets]
    for name, off, n in m.flat_layout():
        hit = [i for i, (bo, bn) in enumerate(buckets) if bo <= off and off + n <= bo + bn]
        part = [i for i, (bo, bn) in enumerate(buckets) if off < bo + bn and off + n > bo]
        assert hit == part, (name, hit, part)                    # never straddles a bucket boundary
        if hit:
            inside[hit[0]].append(name)
    for i, names in enumerate(inside):
        want = {n for n, _, _ in m.flat_layout() if n.startswith(f"rotations_module.STEblocks.{i}.") or n.startswith(f"rotations_module.TTEblocks.{i}.")}
        assert set(names) == want and len(want) == 24, (i, sorted(set(names) ^ want)[:4])
    ends = [bo + bn for bo, bn in buckets]
    assert all(buckets[i + 1][0] >= ends[i] for i in range(len(buckets) - 1))


def test_generated_code_has_no_packed_op_reading_a_freshly_loaded_high_register():
    """Audit of the gfx950 assembly of every kernel source (csrc/common.h, "packed-fp32 guard"): no packed fp32 op at all, in particular none
    whose low lane takes the high register of a pair.  Round 3 saw run-to-run different results (2 mm on the segment lengths at the
    benchmark's batch, non-reproducible gradients) at two sites with that operand form; the form alone is not faulty (round-4 probe), the cause
    is unknown, and the library keeps the whole instruction class out.  tools/scan_pk_opsel.py compiles the sources with hipcc
    (cross-compilation, no GPU needed) and must find none."""
    import shutil, subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_pk_opsel.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 suspicious packed ops" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    import glob
    nsrc = len(glob.glob(os.path.join(ROOT, "manipose_amd", "csrc", "*.hip")))
    assert f"0 suspicious packed ops in {nsrc} files" in r.stdout, r.stdout[-500:]        # every source was compiled and scanned
    assert "0 packed fp32 ops" in r.stdout, r.stdout[-500:]                                # build.sh: -packed-fp32-ops (no v_pk_*_f32 at all)


def test_product_library_reads_no_environment_variable():
    """ABI v7: everything that changes a model's arithmetic or stream use is a field of mp_model_config, the kernel selectors are explicit
    mp_set_option calls, and the product library reads NO environment variable: it does not import getenv, contains no MANIPOSE_* name
    (the wrong-by-design timing ablations MANIPOSE_GEMM_DEBUG / _ABL / MANIPOSE_ATTN_DEBUG exist in the diagnostics build only, MP_DIAG=1
    build.sh), and every getenv in the sources sits inside an #ifdef MP_GEMM_DIAG block."""
    import glob, re, subprocess
    from manipose_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"MANIPOSE_" not in blob, re.findall(rb"MANIPOSE_[A-Z0-9_]+", blob)[:5]
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout
    for path in glob.glob(os.path.join(ROOT, "manipose_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "manipose_amd", "csrc", "*.h")):
        depth = 0                                   # nesting depth inside #ifdef MP_GEMM_DIAG
        stack = []
        for no, line in enumerate(open(path), 1):
            t = line.strip()
            if t.startswith("#if"):
                stack.append("MP_GEMM_DIAG" in t and not t.startswith("#ifndef"))
            elif t.startswith("#else") and stack:
                stack[-1] = False
            elif t.startswith("#endif") and stack:
                stack.pop()
            if "getenv" in line and not t.startswith("//"):
                assert any(stack), f"{path}:{no}: getenv outside the diagnostics build: {t}"


def test_model_config_carries_the_engine_options():
    """mp_model_config (ABI v7) <-> _lib.ModelConfig <-> the model attributes: f16f8 / f16_backward / streams reach the engine per model,
    default to 0 (three bf16 products everywhere, bf16 backward, both extra streams), and a layout-only handle accepts them without a GPU."""
    import re
    from manipose_amd import RMCLManifoldMixSTE, _lib, h36m_skeleton
    from manipose_amd.architectures.engine import LiftEngine
    hdr = open(_lib.HEADER_PATH).read()
    body = hdr[hdr.index("typedef struct mp_model_config {"):hdr.index("} mp_model_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [n for decl in re.findall(r"(?:int|float)\s+([^;]+);", body) for n in re.split(r"\s*,\s*", decl.strip())]
    assert fields == [n for n, _ in _lib.ModelConfig._fields_], (fields, [n for n, _ in _lib.ModelConfig._fields_])
    assert fields[-4:] == ["f16f8", "f16_backward", "streams", "debug"]
    base = dict(arch="rmcl_manifold", num_frame=27, num_joints=17, num_bones=16, embed_dim_rot=256, depth_rot=2, num_heads_rot=4, embed_dim_seg=32,
                depth_seg=1, num_heads_seg=4, n_hyp=2, drop_path_rate=0.0, max_batch=0, precision="bf16x3")
    e0 = LiftEngine(**base)
    assert (e0.cfg.f16f8, e0.cfg.f16_backward, e0.cfg.streams) == (0, 0, 0)
    e1 = LiftEngine(**base, f16f8=2, f16_backward=True, side_stream=False, wgrad_stream=False)
    assert (e1.cfg.f16f8, e1.cfg.f16_backward, e1.cfg.streams) == (2, 1, 3)
    assert e0.layout == e1.layout                               # the parameter layout does not depend on the operand form
    with pytest.raises(RuntimeError, match="f16_backward"):
        LiftEngine(**base, f16f8=2, f16_backward=False)
    m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=27, embed_dim_rot=256, depth_rot=2, num_heads_rot=4, embed_dim_seg=32, depth_seg=1,
                           num_heads_seg=4, n_hyp=2)
    assert m._engine_options() == dict(f16f8=0, f16_backward=False, side_stream=True, wgrad_stream=True, hazard_check=False)
    m.f16f8, m.f16_backward, m.wgrad_stream, m.hazard_check = 1, True, False, True
    assert m._engine_options() == dict(f16f8=1, f16_backward=True, side_stream=True, wgrad_stream=False, hazard_check=True)
    e2 = LiftEngine(**base, hazard_check=True)
    assert e2.cfg.debug == 1 and e0.cfg.debug == 0


def test_stream_hazard_tracker_finds_missing_events():
    """csrc/hazard.h through its own C entry points (no device): vector clocks per stream, events carry the recording stream's clock.  The
    scenarios are the engine's patterns: fork / join of a side stream, an operand handed to the weight-gradient stream and rewritten after
    its 'done' event, and each of them with the event left out."""
    import ctypes as C
    from manipose_amd import _lib
    lib = _lib.load()

    def run(ops):
        h = lib.mp_hazard_create()
        try:
            for op in ops:
                if op[0] == "L":
                    _, stream, name, acc = op
                    n = len(acc)
                    addr = (C.c_int64 * n)(*[a for a, _, _ in acc]); nb = (C.c_int64 * n)(*[b for _, b, _ in acc]); wr = (C.c_int * n)(*[w for _, _, w in acc])
                    assert lib.mp_hazard_launch(h, stream, name.encode(), n, addr, nb, wr) == 0
                elif op[0] == "R":
                    assert lib.mp_hazard_record(h, op[1], op[2]) == 0
                else:
                    assert lib.mp_hazard_wait(h, op[1], op[2]) == 0
            out = (C.c_int64 * 4)()
            buf = C.create_string_buffer(8192)
            assert lib.mp_hazard_report(h, out, buf, len(buf)) == 0
            return list(out), buf.value.decode()
        finally:
            lib.mp_hazard_destroy(h)

    X, G, TMP = 0x1000, 0x9000, 0x20000
    # fork / join: main writes X, side reads it behind the fork event, writes G; main reads G behind the join event
    ok = [("L", 0, "produce", [(X, 256, 1)]), ("R", 0, 0), ("W", 1, 0), ("L", 1, "side", [(X, 256, 0), (G, 64, 1)]), ("R", 1, 1), ("W", 0, 1),
          ("L", 0, "consume", [(G, 64, 0)])]
    out, msg = run(ok)
    assert out == [3, 2, 0, 2] and msg == ""                     # 3 launches, 2 conflicting cross-stream pairs, both ordered, 2 events
    out, msg = run([o for o in ok if o != ("W", 1, 0)])          # the side stream does not wait for the fork event
    assert out[2] == 1 and "RAW" in msg and "'side'" in msg and "'produce'" in msg
    out, msg = run([o for o in ok if o != ("W", 0, 1)])          # main does not wait for the join event
    assert out[2] == 1 and "RAW" in msg and "'consume'" in msg
    # the weight-gradient pattern: main writes TMP, the other stream reads it (E event), main rewrites TMP only behind the W event
    wg = [("L", 0, "dgrad", [(TMP, 1024, 1)]), ("R", 0, 0), ("W", 2, 0), ("L", 2, "wgrad", [(TMP, 1024, 0)]), ("R", 1, 2), ("W", 0, 1),
          ("L", 0, "next dgrad", [(TMP + 512, 1024, 1)])]
    out, msg = run(wg)
    assert out[2] == 0 and out[1] == 2
    out, msg = run([o for o in wg if o != ("W", 0, 1)])          # the rewrite does not wait for the reader: write after read
    assert out[2] == 1 and msg.startswith("WAR") and "next dgrad" in msg
    # transitivity: 0 -> 1 -> 2 orders 0 before 2; an event that was never recorded orders nothing; disjoint bytes never conflict
    out, msg = run([("L", 0, "a", [(X, 16, 1)]), ("R", 0, 0), ("W", 1, 0), ("L", 1, "b", [(G, 16, 1)]), ("R", 1, 1), ("W", 2, 1), ("L", 2, "c", [(X, 16, 1)])])
    assert out[2] == 0 and out[1] == 1
    out, msg = run([("L", 0, "a", [(X, 16, 1)]), ("W", 1, 5), ("L", 1, "c", [(X + 8, 16, 1)])])
    assert out[2] == 1 and msg.startswith("WAW")
    out, msg = run([("L", 0, "a", [(X, 16, 1)]), ("L", 1, "c", [(X + 16, 16, 1)]), ("L", 1, "d", [(X, 16, 0)]), ("L", 2, "e", [(X + 16, 4, 0)])])
    assert out[2] == 2                                           # d reads a's bytes, e reads c's bytes; a and c do not overlap


def test_generated_k_step_is_up_to_date():
    """manipose_amd/csrc/kloop_asm.inc is the output of tools/gen_kloop_asm.py (the hand-scheduled k-step of the persistent GEMM)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_kloop_asm.py"), "--check"])
    assert r.returncode == 0, "run python tools/gen_kloop_asm.py"


# ---------------------------------------------------------------------------------------------------------------------------------
# toy experiment (BASELINE config #1: 1-D -> 2-D circle-manifold MLPs on the CPU) against vectors produced by the reference's own
# toy_experiment package (oracle/gen_golden_toy.py -> tests/golden/toy.npz)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def toy():
    sys.path.insert(0, os.path.join(ROOT, "toy_experiment"))
    import circle_toy
    return circle_toy, np.load(os.path.join(ROOT, "tests", "golden", "toy.npz"))


def test_toy_scenarios_sample_bit_identically_to_the_reference(toy):
    ct, fx = toy
    for name in ("easy", "hard-1", "hard-2", "hard-4"):
        ds = ct.LiftingDataset(ct.scenario(name, 1.0, 42), 1000, 1000, 1000)
        assert np.array_equal(ds.X_train.numpy(), fx[f"data.{name}.X_train"]) and np.array_equal(ds.Y_train.numpy(), fx[f"data.{name}.Y_train"])
        assert np.array_equal(ds.X_val[:32].numpy(), fx[f"data.{name}.X_val_head"]) and np.array_equal(ds.Y_test[:32].numpy(), fx[f"data.{name}.Y_test_head"])
        sums = [t.double().sum().item() for t in (ds.X_train, ds.X_val, ds.X_test, ds.Y_train, ds.Y_val, ds.Y_test)]
        np.testing.assert_allclose(sums, fx[f"data.{name}.sums"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(ct.scenario(name, 1.0, 0).pdf(np.linspace(-np.pi, np.pi, 41)), fx[f"data.{name}.pdf"], rtol=1e-12)
        assert ds.X_train.shape == (1000, 1) and ds.Y_train.shape == (1000, 2)
        np.testing.assert_allclose(ds.Y_train.norm(dim=1).numpy(), 1.0, atol=1e-6)            # every target lies on the circle
    assert abs(float(ct.LiftingDataset(ct.HardBimodalDist(radius=1, random_state=42), 1000, 1000, 1000).X_train[0]) - 0.5287) < 1e-4
    with pytest.raises(ValueError):
        ct.scenario("torus", 1.0, 0)


def test_toy_models_match_the_reference_forward_and_state_dict(toy):
    ct, fx = toy
    X = torch.from_numpy(fx["data.hard-2.X_train"])[:64]
    for tag, act in (("tanh", torch.nn.Tanh), ("relu", torch.nn.ReLU), ("sqrelu", ct.SquaredReLU)):
        torch.manual_seed(42)
        m = ct.Mlp(in_features=1, hidden_features=32, out_features=2, n_layers=2, act_layer=act)
        want = {k[len(f"mlp.{tag}.w::"):]: v for k, v in fx.items() if k.startswith(f"mlp.{tag}.w::")}
        assert list(m.state_dict()) == list(want) or set(m.state_dict()) == set(want)
        for k, v in m.state_dict().items():
            assert np.array_equal(v.numpy(), want[k]), k                                            # same initialiser draws in the same order
        np.testing.assert_allclose(m.train()(X).detach().numpy(), fx[f"mlp.{tag}.train_out"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(m.eval()(X).detach().numpy(), fx[f"mlp.{tag}.eval_out"], rtol=1e-6, atol=1e-7)
    torch.manual_seed(42)
    c = ct.ConstrainedMlp(in_features=1, hidden_features=32, out_features=1, n_layers=2, act_layer=torch.nn.Tanh, radius=1.0)
    for k, v in c.state_dict().items():
        assert np.array_equal(v.numpy(), fx["constrained.w::" + k]), k
    out = c.train()(X).detach()
    np.testing.assert_allclose(out.numpy(), fx["constrained.train_out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out.norm(dim=1).numpy(), 1.0, atol=1e-6)                             # the manifold constraint holds by construction
    torch.manual_seed(42)
    r = ct.ConstrainedMlpRmcl(in_features=1, hidden_features=32, out_features=1, n_layers=2, act_layer=torch.nn.Tanh, radius=1.0, n_hyp=5, beta=0.1)
    for k, v in r.state_dict().items():
        assert np.array_equal(v.numpy(), fx["rmcl.w::" + k]), k
    hyp = r.train()(X).detach()
    np.testing.assert_allclose(hyp.numpy(), fx["rmcl.train_hyp"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r.aggregate(hyp).numpy(), fx["rmcl.weighted_ave"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(hyp[..., 2].sum(dim=1).numpy(), 1.0, atol=1e-6)
    # rMCL loss (the reference's 2-D helpers are shadowed at HEAD): closed form on a hand-made case
    h = torch.tensor([[[1.0, 0.0, 0.8], [0.0, 1.0, 0.2]]])
    y = torch.tensor([[0.0, 1.0]])
    want = 0.0 + 0.1 * (-(np.log(1 - 0.8) + np.log(0.2)) / 2)                                       # winner = head 1 (exact), BCE vs one-hot (0, 1)
    assert abs(r.wta_with_scoring_l2_loss(h, y).item() - want) < 1e-6
    assert torch.equal(r.aggregate(h, "best_score"), torch.tensor([[1.0, 0.0]])) and abs(ct.oracle_multihyp_mpjpe(h, y)) < 1e-7
    with pytest.raises(ValueError):
        r.aggregate(h, "median")


@pytest.mark.parametrize("tag,lr", [("constrained", 1e-2), ("mlp", 1e-3)])
def test_toy_trainer_reproduces_the_reference_training_curves(toy, tmp_path, tag, lr):
    """Three epochs of the reference's Trainer (Adam, ReduceLROnPlateau, MSE, shuffled batches from the torch RNG, best-validation
    checkpoint reloaded at the end), then its evaluation metrics."""
    import types
    ct, fx = toy
    ds = ct.LiftingDataset(ct.HardBimodalDist(radius=1.0, random_state=42), 1000, 1000, 1000)
    torch.manual_seed(42)
    model = (ct.ConstrainedMlp(in_features=1, hidden_features=32, out_features=1, n_layers=2, act_layer=torch.nn.Tanh, radius=1.0) if tag == "constrained"
             else ct.Mlp(in_features=1, hidden_features=32, out_features=2, n_layers=2, act_layer=torch.nn.Tanh))
    tr = ct.Trainer(model=model, optim_cls=torch.optim.Adam, sched_cls=torch.optim.lr_scheduler.ReduceLROnPlateau, checkpointing_dir=tmp_path, lr=lr,
                    config_train=types.SimpleNamespace(lr_min=0.0, lr_patience=10, lr_threshold=1e-4))
    tr.train(epochs=3, loader=ds.get_tr_loader(batch_size=100, num_workers=0), loss_func=torch.nn.functional.mse_loss, val_data=ds.validation_set, log=None)
    np.testing.assert_allclose(tr.loss_list, fx[f"train.{tag}.loss_list"], rtol=1e-5)
    np.testing.assert_allclose(tr.val_loss_list, fx[f"train.{tag}.val_loss_list"], rtol=1e-5)
    (val_mpjpe, test_mpjpe), (_, pred), _ = tr.eval((ds.validation_set, ds.test_set), ct.calc_mpjpe)
    (val_dtc, test_dtc), _, _ = tr.eval((ds.validation_set, ds.test_set), ct.distance_to_circle)
    np.testing.assert_allclose([val_mpjpe, test_mpjpe, val_dtc, test_dtc], fx[f"train.{tag}.metrics"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(pred[:32].numpy(), fx[f"train.{tag}.test_pred_head"], rtol=1e-4, atol=1e-5)
    assert (tmp_path / "model_best_val.pth").exists() and (tmp_path / "params_best_val.pth").exists()


def test_toy_entry_point_runs_the_readme_recipes(tmp_path, monkeypatch):
    """`cd toy_experiment; python main.py model.arch=constrained +train=constrained_easy` end to end, shortened."""
    sys.path.insert(0, os.path.join(ROOT, "toy_experiment"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("toy_main", os.path.join(ROOT, "toy_experiment", "main.py"))
    toy_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(toy_main)
    monkeypatch.chdir(tmp_path)
    v = toy_main.main(["model.arch=constrained", "+train=constrained_easy", "train.epochs=4", "train.workers=0", "run.experiment=t1"])
    assert 0 < v < 0.2                                                       # the easy scenario is unimodal: the constrained MLP fits it
    for f in ("model_best_val.pth", "train_loss.npy", "test_predictions.npy"):
        assert (tmp_path / "outputs" / "t1" / f).exists()
    pred = np.load(tmp_path / "outputs" / "t1" / "test_predictions.npy")
    np.testing.assert_allclose(np.linalg.norm(pred, axis=1), 1.0, atol=1e-5)         # on the circle
    v2 = toy_main.main(["model.arch=constrained_rmcl", "+train=rmcl_constrained_hard2", "data.scenario=hard-2", "train.epochs=3", "train.workers=0"])
    assert np.isfinite(v2)
    v3 = toy_main.main(["model.arch=mlp", "+train=mlp_hard2", "data.scenario=hard-2", "train.epochs=2", "train.workers=0", "model.act=sqrelu"])
    assert np.isfinite(v3)
    for bad in (["data.scenario=torus-2Dto3D"], ["+train=nope"], ["model.arc=mlp"]):
        with pytest.raises(SystemExit):
            toy_main.main(bad)
    with pytest.raises(ValueError):
        toy_main.main(["model.arch=transformer", "train.epochs=1"])


# ---------------------------------------------------------------------------------------------------------------------------------
# muP mode (model.mup=True): the reference's own muP code pinned by fixtures from the reference (oracle/gen_golden_mup.py); the third-party
# mup package's parts (MuReadout multiplier, base shapes, MuAdam) restated in manipose_amd/mup_lite.py (parity unpinned)
# ---------------------------------------------------------------------------------------------------------------------------------
def _mup_oracle_cfg(fx):
    c = fx["cfg"]
    wm_rot, wm_seg = (float(v) for v in fx["width_mult"])
    seg = orc.mup_scales(c["C_seg"], c["heads_seg"], c["depth_seg"], readout=1.0 / wm_seg)
    if c["n_hyp"]:
        rot = dict(readout=1.0 / wm_rot)                       # RMCLRotMixSTE: backbone without muP, MuReadout heads
    else:
        rot = orc.mup_scales(c["C_rot"], c["heads_rot"], c["depth_rot"], readout=1.0 / wm_rot)
    return dict(orc.oracle_cfg(c), mup={"rot": rot, "seg": seg})


@pytest.mark.parametrize("name", ["mup_manifold", "mup_rmcl"])
def test_oracle_mup_mode_vs_reference_fixture(name):
    fx = load_fixture(name)
    st = {k: v.requires_grad_(True) for k, v in fixture_state(fx).items()}
    X, y = torch.from_numpy(fx["X"]), torch.from_numpy(fx["y"])
    cfg = _mup_oracle_cfg(fx)
    if fx["cfg"]["n_hyp"]:
        poses, scores = orc.rmcl_manifold_forward(X, st, cfg)
        np.testing.assert_allclose(scores.detach().numpy(), fx["scores"], rtol=1e-5, atol=1e-6)
        total, _ = orc.rmcl_training_loss(poses, scores, y)
    else:
        poses = orc.manifold_forward(X, st, cfg)
        total, _ = orc.manifold_training_loss(poses, y)
    np.testing.assert_allclose(poses.detach().numpy(), fx["poses"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(total.item(), float(fx["loss_total"]), rtol=1e-5)
    total.backward()
    for k, v in st.items():
        want = fx["g::" + k]
        np.testing.assert_allclose(v.grad.numpy(), want, rtol=2e-3, atol=1e-6 + 1e-4 * np.abs(want).max(), err_msg=k)


def _mup_model(fx):
    """The product's model with mup=True and the fixture's width multipliers, obtained the way the package derives them: base shapes from a
    base and a delta model (widths rot 16 -> base, seg 4 -> base: multipliers 32/16 = 2 and 16/4 = 4)."""
    from manipose_amd import ManifoldMixSTE, RMCLManifoldMixSTE, h36m_skeleton
    from manipose_amd.mup_lite import make_base_shapes, set_base_shapes
    c = fx["cfg"]

    def build(C_rot, C_seg):
        kw = dict(num_frame=c["T"], embed_dim_rot=C_rot, depth_rot=c["depth_rot"], num_heads_rot=c["heads_rot"], embed_dim_seg=C_seg,
                  depth_seg=c["depth_seg"], num_heads_seg=c["heads_seg"], drop_path_rate=0.0, mup=True)
        return RMCLManifoldMixSTE(h36m_skeleton(), n_hyp=c["n_hyp"], **kw) if c["n_hyp"] else ManifoldMixSTE(h36m_skeleton(), **kw)
    model = build(c["C_rot"], c["C_seg"])
    set_base_shapes(model, make_base_shapes(build(16, 4), build(64, 8)), rescale_params=False)
    model.load_state_dict(fixture_state(fx), strict=True)
    return model


@pytest.mark.parametrize("name", ["mup_manifold", "mup_rmcl"])
def test_mup_model_construction_matches_the_reference(name):
    from manipose_amd.mup_lite import MuReadout, mup_lr_multipliers
    fx = load_fixture(name)
    model = _mup_model(fx)
    sc = model._scale_cfg()
    a_rot, a_seg = (float(v) for v in fx["attn_scales"])
    r_rot, _, r_seg = (float(v) for v in fx["resid_scales"])
    d_rot = fx["cfg"]["C_rot"] // fx["cfg"]["heads_rot"]
    assert abs((sc["qk_scale_rot"] or d_rot ** -0.5) - a_rot) < 1e-7 and abs(sc["qk_scale_seg"] - a_seg) < 1e-7       # mix_ste.py:243
    assert abs((sc["resid_scale_rot"] or 1.0) - r_rot) < 1e-7 and abs(sc["resid_scale_seg"] - r_seg) < 1e-7             # :330
    assert abs(sc["readout_mult_rot"] - 0.5) < 1e-7 and abs(sc["readout_mult_seg"] - 0.25) < 1e-7                       # 1 / width_mult
    got = sorted(n for n, m in model.named_modules() if isinstance(m, MuReadout))
    assert got == sorted(str(n) for n in fx["readout_modules"])                                                        # which layers are MuReadouts
    mult = mup_lr_multipliers(model)
    assert mult["rotations_module.STEblocks.0.attn.qkv.weight"] == (0.5, 2.0)          # two width dimensions: lr / width_mult, wd * width_mult
    assert mult["segments_module.TTEblocks.1.mlp.fc2.weight"] == (0.25, 4.0)
    assert mult["rotations_module.Spatial_patch_to_embedding.weight"] == (1.0, 1.0) and mult["rotations_module.Spatial_norm.weight"] == (1.0, 1.0)
    eng_cfg = model._engine_cfg
    assert eng_cfg["embed_dim_rot"] == fx["cfg"]["C_rot"]
