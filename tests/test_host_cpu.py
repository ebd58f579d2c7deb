"""CPU-side tests: the C-ABI library loads and exports every symbol of include/gpt_hip.h, refuses to
compute without a GPU, and the host logic (affine alignment, transport algebra, resampling,
quaternions, sharding + gloo collectives) matches the golden vectors / oracle."""
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT, assert_parity, load_golden


def header_functions():
    src = open(os.path.join(ROOT, "include", "gpt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from gaussian_process_transportation_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gpt_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert b"gfx950" in _lib.load().gpt_version()


def test_no_cpu_fallback():
    from gaussian_process_transportation_amd import _lib, GaussianProcess
    if _lib.load().gpt_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.GptError):
        _lib.Handle(0)
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    gp = GaussianProcess(kernel=C(1.0) * RBF(0.2) + WhiteKernel(1e-3), optimizer=None, verbose=False)
    with pytest.raises(_lib.GptError):
        gp.fit(np.random.rand(10, 2), np.random.rand(10, 2))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaussian_process_transportation_amd")
    for dirpath, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d != "build"]          # untracked build scratch (objects, A/B libraries)
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # only inside the CPU-baseline legs: every function body that imports the oracle is a *cpu_baseline function
    for m in re.finditer(r"from oracle|import oracle", bench):
        owner = re.findall(r"^def (\w+)\(", bench[:m.start()], flags=re.M)[-1]
        assert owner.endswith("cpu_baseline"), owner


def test_affine_transform_matches_golden():
    from gaussian_process_transportation_amd import AffineTransform
    g = load_golden("letterS_2d")
    a = AffineTransform(verbose=False).fit(g["source"], g["target"])
    assert_parity(a.rotation_matrix, g["rotation"], 1e-12, "R")
    assert_parity(a.S_centroid, g["S_centroid"], 1e-14, "S centroid")
    assert_parity(a.T_centroid, g["T_centroid"], 1e-14, "T centroid")
    assert_parity(a.predict(g["source"]), g["gp_X"], 1e-12, "aligned source")
    assert a.derivative(g["demo"]).shape == (400, 2, 2)
    s = AffineTransform(do_scale=True, verbose=False).fit(g["source"], g["target"])
    assert float(s.scale) == pytest.approx(float(g["scale2"]), rel=1e-12)
    assert_parity(s.derivative(g["demo"][:3]), np.repeat(g["rotation"][None], 3, 0), 1e-12, "derivative ignores scale")
    i = AffineTransform(do_rotation=False, verbose=False).fit(g["source"], g["target"])
    assert np.array_equal(i.rotation_matrix, np.eye(2))
    one = AffineTransform(verbose=False).fit(g["source"][:1], g["target"][:1])      # too few points -> identity
    assert np.array_equal(one.rotation_matrix, np.eye(2))
    # reflection branch: a mirrored target must still give det(R) = +1
    mirrored = g["source"] * np.array([1.0, -1.0])
    r = AffineTransform(verbose=False).fit(g["source"], mirrored)
    assert np.linalg.det(r.rotation_matrix) == pytest.approx(1.0, abs=1e-12)
    g3 = load_golden("surface_3d")
    a3 = AffineTransform(verbose=False).fit(g3["source"], g3["target"])
    assert_parity(a3.rotation_matrix, g3["rotation"], 1e-12, "R 3-D")


def test_resample_matches_golden():
    from gaussian_process_transportation_amd.utils import resample
    g = load_golden("letterS_2d")
    assert_parity(resample(g["demo_raw"], 400), g["demo"], 1e-12, "demo")
    assert_parity(resample(g["floor_raw"], 20), g["source"], 1e-12, "floor")
    assert_parity(resample(g["newfloor_raw"], 20), g["target"], 1e-12, "newfloor")


class _OracleDeltaMap:
    """Test double for the delta_map plugin slot: the CPU oracle behind the reference's method names."""

    def __init__(self, g):
        from oracle import gp_oracle as orc
        self.o = orc.GaussianProcessOracle(g["constant_value"], g["length_scale"], g["noise_level"])

    def fit(self, X, Y):
        self.o.fit(X, Y)

    def predict(self, x, return_std=False):
        return self.o.predict(x, return_std=return_std)

    def derivative(self, x, return_var=False):
        return self.o.derivative(x, return_var=return_var)


def test_policy_transportation_algebra_with_oracle_plugin():
    """The duck-typed plugin boundary: PolicyTransportation around a CPU delta_map reproduces the
    reference's transported trajectory / velocity / variance."""
    from gaussian_process_transportation_amd import PolicyTransportation
    g = load_golden("letterS_2d")
    pt = PolicyTransportation(_OracleDeltaMap(g), verbose=False)
    pt.fit(g["source"], g["target"])
    traj, std = pt.transport(g["demo"])
    vel, var_vel = pt.transport_velocity(g["demo"], g["delta"])
    assert_parity(traj, g["traj"], 1e-9, "traj")
    assert_parity(std, g["std"], 1e-7, "std")
    assert_parity(vel, g["vel"], 1e-7, "vel")
    assert_parity(var_vel, g["var_vel"], 1e-6, "var_vel")
    traj2, none = pt.transport(g["demo"], return_std=False)
    assert none is None and np.array_equal(traj2, traj)
    vel2, none = pt.transport_velocity(g["demo"], g["delta"], return_var=False)
    assert none is None and np.array_equal(vel2, vel)


def test_quaternion_helpers():
    from gaussian_process_transportation_amd.quaternion import (quaternion_from_nonorthogonal, quaternion_multiply,
                                                                rotation_matrix_from_quaternion)
    rng = np.random.default_rng(0)
    q = rng.standard_normal((50, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 0] < 0] *= -1
    R = rotation_matrix_from_quaternion(q)
    assert_parity(quaternion_from_nonorthogonal(R), q, 1e-12, "q(R(q))")
    # closest rotation of a perturbed matrix = polar factor
    A = R + 0.05 * rng.standard_normal(R.shape)
    U, _, Vt = np.linalg.svd(A)
    polar = U @ Vt
    qa = quaternion_from_nonorthogonal(A)
    assert np.max(np.abs(rotation_matrix_from_quaternion(qa) - polar)) < 0.02
    p = rng.standard_normal((50, 4)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    assert_parity(rotation_matrix_from_quaternion(quaternion_multiply(q, p)),
                  R @ rotation_matrix_from_quaternion(p), 1e-12, "R(q p) = R(q) R(p)")


def test_shard_range():
    from gaussian_process_transportation_amd.distributed import shard_range
    for M in (0, 1, 7, 500_000, 4_000_001):
        for world in (1, 2, 3, 8):
            edges = [shard_range(M, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == M
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    from gaussian_process_transportation_amd.distributed import broadcast_geometry, gather_rows, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        geom = broadcast_geometry((8192, 3, 3, 1, 0) if rank == 0 else None, src=0)
        M = 1001
        a, b = shard_range(M, rank, world)
        full = np.arange(M * 3, dtype=np.float64).reshape(M, 3)
        got = gather_rows(full[a:b] * 2.0, M)
        q.put((rank, geom, bool(np.array_equal(got, full * 2.0))))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    """world_size-2 rehearsal (CPU, gloo) of the multi-GPU host logic: geometry broadcast from the fitting
    rank, contiguous query shards, row gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, (8192, 3, 3, 1, 0), True), (1, (8192, 3, 3, 1, 0), True)]


def _run_bench(*argv, env_extra=None):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` (the driver's command shape, no torch.distributed environment) must start two
    ranks itself and relay ONE JSON line; --dry-run keeps the ranks on gloo / CPU."""
    import json
    r = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 2


def test_bench_launcher_starts_eight_ranks():
    """configs[3]'s command shape, `python bench.py --gpus 8`: eight ranks rendezvous (gloo / CPU rehearsal), every rank
    sees eight, one JSON line comes back."""
    import json
    r = _run_bench("--gpus", "8", "--steps", "2", "--warmup", "0", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8


def test_bench_launcher_keeps_a_device_restriction_and_refuses_too_few_devices():
    """HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES already in the environment stay in force (rank r = device r of the
    visible set); fewer visible devices than --gpus is refused by the launcher before a single rank is started."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    r = _run_bench("--gpus", "4", "--steps", "1", "--dry-run", env_extra={"HIP_VISIBLE_DEVICES": "2,3", "GPT_BENCH_DRY_CHECK_DEVICES": "1"})
    assert r.returncode != 0 and "HIP_VISIBLE_DEVICES" in r.stderr and "2 device(s) visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", env_extra={"ROCR_VISIBLE_DEVICES": "5,6", "HIP_VISIBLE_DEVICES": "0,1", "GPT_BENCH_DRY_CHECK_DEVICES": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    old = {k: os.environ.pop(k, None) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    try:
        assert bench.visible_device_shortfall(8) is None
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,2,3,4,5,6,7"
        assert bench.visible_device_shortfall(8) is None and bench.visible_device_shortfall(9) is not None
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1,-1,3"
        assert bench.visible_device_shortfall(2) is None and bench.visible_device_shortfall(3) is not None
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        assert bench.visible_device_shortfall(1) is not None
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    # the ranks themselves check the device count before the process group exists
    src = open(os.path.join(ROOT, "bench.py")).read()
    init = src[src.index("def init_ranks"):src.index("def broadcast_fitted")]
    assert init.index("torch.cuda.device_count()") < init.index("init_process_group")


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies must not leave the launcher (or the other rank) hanging, and the exit code is not 0."""
    r = _run_bench("--gpus", "2", "--steps", "1", "--dry-run", env_extra={"GPT_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_does_not_touch_the_gpu_stack():
    """The launcher branch must run before torch / the HIP library are imported (a process that has initialised
    the GPU must never start or become the ranks)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args") < main.index("import torch")
    launcher = src[src.index("def launch_ranks"):src.index("def dry_run")]
    assert "import torch" not in launcher and "_lib" not in launcher and "os.exec" not in src


VI_GEN, VI_FIRST, VI_ZERO = 1, 2, 4


def _diag_cost(qa, qb):
    """gpt_plan.h var_diag_cost: quarters [qa, qb) of a diagonal tile — row groups g and 7 - g share a SIMD, group g has 16 (g + 1) steps."""
    if (qa, qb) == (0, 4):
        return 72
    steps = lambda g: max(min(16 * (g + 1), 32 * qb) - 32 * qa, 0)
    return max(max((steps(g) + steps(7 - g)) // 2, (6 * max(steps(g), steps(7 - g)) + 9) // 10) for g in range(4))


def _check_var_plan(cols, nbi, nt, P, order, cut_diag=None):
    """Replays the item lists of csrc/gpt_plan.h the way k_var executes them and checks what the kernel relies on.
    cut_diag: None = the plan decides whether a diagonal tile may be divided (only where a workgroup's share is below 4 tiles)."""
    from collections import defaultdict
    from gaussian_process_transportation_amd import _lib
    pl = _lib.debug_var_plan(cols, nbi, nt, P, order)
    if cut_diag is not None:
        assert pl["cut_diag"] == cut_diag
    cut_diag = pl["cut_diag"]
    ncb, nfull = pl["ncb"], pl["nfull"]
    assert ncb == -(-cols // 64) and nfull == ncb // P * P
    ncb_t = ncb - nfull
    KQ = 4                                # item k ranges count quarter tiles (gpt_plan.h VAR_KQ)
    cover = np.zeros((max(ncb_t, 1), nt, nbi, KQ * nbi), dtype=np.int64)
    slot_sweeps = {}                      # slab slot -> [(cb, task, ib)] folded into it
    vslot_part = {}                       # vslab slot -> (cb, task, ib, k_lo, k_hi)
    cost = np.zeros(P)
    ib_ = pl["item_begin"]
    assert ib_[0] == 0 and ib_[-1] == pl["n_items"] and np.all(np.diff(ib_) >= 0)
    for p in range(P):
        cur_cb, generated, running = None, set(), None
        for cb, task, ib, k_lo, k_hi, flags, slot, vslot in pl["items"][ib_[p]:ib_[p + 1]]:
            assert nfull <= cb < ncb and 0 <= task < nt and 0 <= ib < nbi and 0 <= k_lo < k_hi <= KQ * (ib + 1)
            if not cut_diag:
                assert k_hi <= KQ * ib or (k_hi == KQ * (ib + 1) and k_lo <= KQ * ib), "the diagonal tile must not be divided here"
            if cb != cur_cb:
                assert flags & VI_FIRST, "a new column block must drop the previous scratch image"
                cur_cb, generated = cb, set()
            tiles = set(range(k_lo, k_hi))
            if flags & VI_GEN:
                generated |= tiles
            else:
                assert tiles <= generated, "reload of B fragments this workgroup never generated for this block"
            cover[cb - nfull, task, ib, k_lo:k_hi] += 1
            cost[p] += 32 * max(min(k_hi, KQ * ib) - k_lo, 0) + (_diag_cost(max(k_lo - KQ * ib, 0), k_hi - KQ * ib) if k_hi > KQ * ib else 0)
            if vslot >= 0:
                assert slot < 0 and vslot not in vslot_part
                vslot_part[vslot] = (cb, task, ib, k_lo, k_hi)
            else:
                assert (k_lo, k_hi) == (0, KQ * (ib + 1)), "only whole sweeps may be folded into column sums"
                if flags & VI_ZERO:
                    assert not running, "column sums dropped without a flush"
                    running = []
                assert running is not None, "accumulating onto sums that were already flushed"
                assert all((c, t) == (cb, task) for c, t, _ in running)
                running.append((cb, task, ib))
                if slot >= 0:
                    assert slot not in slot_sweeps
                    slot_sweeps[slot] = running
                    running = None
        assert not running, "workgroup ends with unflushed column sums"
    if ncb_t:
        want = np.repeat(np.tril(np.ones((nbi, nbi), dtype=np.int64)), KQ, axis=1)[None, None]      # [ib][quarter]: quarter < 4 (ib + 1)
        assert np.array_equal(cover, np.broadcast_to(want, cover.shape)), "a tile is missing or computed twice"
    assert sorted(vslot_part) == list(range(pl["n_vslots"]))
    for v0, v1, slot in pl["splits"]:
        parts = [vslot_part[v] for v in range(v0, v1)]
        assert len(parts) >= 2 and len({pp[:3] for pp in parts}) == 1
        assert parts[0][3] == 0 and parts[-1][4] == KQ * (parts[0][2] + 1)
        assert all(a[4] == b[3] for a, b in zip(parts, parts[1:])), "parts of a cut sweep must tile its k range in order"
        for g in range(8):                # one slab slot per row group (gpt_plan.h VAR_SPLIT_SLOTS), k_var_finalize adds them
            assert slot + g not in slot_sweeps
            slot_sweeps[slot + g] = [parts[0][:3]] if g == 0 else []
    assert sorted(slot_sweeps) == list(range(nfull * nt, pl["n_slots"]))
    assert len(pl["fin"]) == ncb_t * nt
    for e, (b, en) in enumerate(pl["fin"]):
        cbt, task = divmod(e, nt)
        got = sorted(s for sl in range(b, en) for s in slot_sweeps[sl])
        assert got == [(nfull + cbt, task, ib) for ib in range(nbi)], "finalize must see every i-block of a (block, task) once"
    return pl, cost


@pytest.mark.parametrize("cols,nbi,nt,P", [(64, 1, 1, 256), (460, 5, 1, 256), (1000, 16, 1, 256), (4 * 4096, 16, 1, 256),
                                           (500_000, 16, 1, 256), (4 * 500_000, 16, 1, 256), (4_000_000, 4, 3, 256),
                                           (3 * 700, 2, 3, 256), (300, 3, 2, 7), (16384, 16, 1, 256), (16384 + 64, 16, 1, 256),
                                           (10_000, 16, 1, 256), (4 * 10_000 // 3, 16, 1, 256), (133 * 64, 16, 1, 256), (250 * 64, 16, 1, 256),
                                           (128 * 64, 5, 1, 256), (200 * 64, 4, 3, 256), (5 * 64, 2, 1, 7), (500_000, 16, 1, 256)])
@pytest.mark.parametrize("order", [-1, 0, 1])
def test_variance_work_plan_is_a_partition(cols, nbi, nt, P, order):
    pl, cost = _check_var_plan(cols, nbi, nt, P, order)
    busy = cost[cost > 0]
    ncb_t = pl["ncb"] - pl["nfull"]
    assert pl["cohorts"] == (order <= 0 and 2 * ncb_t >= P and ncb_t < P and nbi >= 2 and pl["cohorts"])      # only where the plan may use them
    if len(busy) == P and pl["nfull"] == 0:               # every workgroup has work
        if pl["cohorts"]:
            # two cohorts (whole long sweeps | cut short sweeps), balanced by ONE boundary (s, f): used only when the longer
            # cohort's span is within 4 % of the ideal share (gpt_plan.h) — held here to 4 % + the cut's one-quarter-tile slack
            ideal = cost.sum() / P
            assert 1 <= pl["cohort_s"] < nbi and 0 <= pl["cohort_f"] < pl["cohort_s"]
            assert busy.max() <= 1.04 * ideal + 32 + 8 * nt, (busy.max(), ideal)
        elif order == 1:
            assert busy.max() <= 1.08 * busy.mean() or busy.max() - busy.min() <= 2 * 128 + 72     # (sweep-major diagnostic order: overheads of many cuts)
        else:
            # one list cut at quarter-tile granularity: shares within a tile of each other (a diagonal tile is not divided)
            assert busy.max() - busy.min() <= 2 * 128 + 72


def test_variance_work_plan_cohorts_keep_the_long_sweeps_whole():
    """128 .. 255 column blocks on 256 workgroups (the reference's 10^4-point grids at N = 8192; the tail of the benchmark's
    launch): workgroup b takes the long sweeps of block b whole and in the same order as every other workgroup of that
    cohort (they walk the inverse factor in step), the short sweeps are shared by the remaining workgroups."""
    from gaussian_process_transportation_amd import _lib
    for cols, nbi in ((10_000, 16), (133 * 64, 16), (250 * 64, 16), (200 * 64, 16)):
        pl = _lib.debug_var_plan(cols, nbi, 1, 256, -1)
        ncb = pl["ncb"]
        ib_ = pl["item_begin"]
        first = [pl["items"][ib_[p]:ib_[p + 1]] for p in range(ncb)]
        whole0 = [it for it in first[0] if int(it[7]) < 0]
        s = min(int(it[2]) for it in whole0)
        for b, items in enumerate(first):
            assert [int(it[0]) for it in items] == [b] * len(items)
            whole = [it for it in items if int(it[7]) < 0]
            part = [it for it in items if int(it[7]) >= 0]
            assert [int(it[2]) for it in whole] == list(range(nbi - 1, s - 1, -1))          # ib = nbi-1 .. s, whole
            assert all(int(it[3]) == 0 and int(it[4]) == 4 * (int(it[2]) + 1) for it in whole)
            # at most one partial product, the LAST tiles of the next shorter sweep (what balances the two cohorts), the
            # same for every block
            assert len(part) <= 1 and all(int(it[2]) == s - 1 and int(it[3]) >= 4 and int(it[4]) == 4 * s for it in part)  # ... up to the diagonal
            assert [tuple(int(v) for v in it[2:5]) for it in part] == [tuple(int(v) for v in it[2:5]) for it in first[0] if int(it[7]) >= 0]
        rest = pl["items"][ib_[ncb]:]
        assert len(rest) and max(int(it[2]) for it in rest) == s - 1
        _check_var_plan(cols, nbi, 1, 256, -1)


def test_variance_work_plan_divides_diagonal_tiles_only_for_small_shares(monkeypatch):
    """configs[1] (N = 1024, M = 50 000): three rounds + 14 blocks = 42 tiles, 28 of them diagonal, for 256 workgroups.  With the diagonal
    tile indivisible the tail took as long as one (78 units + its opening); cut at quarters no workgroup gets more than 46.  Lists whose
    shares are several tiles (every large model) keep their diagonal tiles whole — and their plans as they were measured."""
    pl, cost = _check_var_plan(50_000, 2, 1, 256, -1, cut_diag=True)
    assert cost.max() <= 40 and pl["n_splits"] == 28           # (cost: without the per-item overhead of 6)
    monkeypatch.setenv("GPT_VAR_CUT_DIAG", "0")
    pl0, cost0 = _check_var_plan(50_000, 2, 1, 256, -1, cut_diag=False)
    assert cost0.max() == 72 and pl0["n_splits"] == 14
    monkeypatch.delenv("GPT_VAR_CUT_DIAG")
    for cols, nbi in ((10_000, 16), (4096, 16), (500_000, 16), (10_000, 5), (4 * 500_000, 16)):
        _check_var_plan(cols, nbi, 1, 256, -1, cut_diag=False)
    for cols, nbi in ((4096, 2), (4096, 5), (50_000, 1), (460, 5), (64, 1), (16384 + 64, 16)):
        _check_var_plan(cols, nbi, 1, 256, -1, cut_diag=True)
    monkeypatch.setenv("GPT_VAR_CUT_DIAG", "1")                # forced: every shape must still be a partition the kernel can execute
    rng = np.random.default_rng(11)
    for _ in range(40):
        P = int(rng.choice([1, 3, 8, 64, 256]))
        _check_var_plan(int(rng.integers(1, 64 * 3 * P)), int(rng.integers(1, 20)), int(rng.choice([1, 1, 2, 3])), P, int(rng.integers(-1, 2)))


def test_variance_work_plan_random_shapes():
    rng = np.random.default_rng(5)
    for _ in range(60):
        P = int(rng.choice([1, 3, 8, 64, 256, 304]))
        nbi = int(rng.integers(1, 20))
        nt = int(rng.choice([1, 1, 2, 3, 5]))
        cols = int(rng.integers(1, 64 * 3 * P))
        _check_var_plan(cols, nbi, nt, P, int(rng.integers(-1, 2)))


def test_host_orchestration_under_sanitizers():
    """the host side of the C ABI is built with -fsanitize=address,undefined (g++, `make host-asan`, inert
    stand-ins for the HIP runtime and the kernels in csrc/host_stub/) and driven through its argument checks, state
    machine, staging / hand-off copies and variance work plans (tests/asan_driver.py).  GPU sanitizers are not available
    on the pool; this is the CPU-side half."""
    import shutil
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "gaussian_process_transportation_amd", "csrc")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    subprocess.run(["make", "-C", csrc, "host-asan"], check=True, capture_output=True)
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", GPT_HIP_LIB=os.path.join(csrc, "build", "libgpt_host_asan.so"),
               PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "ASAN_DRIVER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_hyperparameter_search_driver_sequential_and_concurrent(monkeypatch):
    """hyperopt.optimize_hyperparameters with the objective served by the CPU oracle instead of the GPU handle (a stand-in
    for `_lib.Handle`): the driver must (1) reproduce scikit-learn's own fit — same L-BFGS-B protocol, same restart
    points from the global RNG — and (2) give bit-identical results and leave the RNG in the same state whether the
    restart runs are driven one after the other or concurrently on a handle each."""
    import types
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    from gaussian_process_transportation_amd import hyperopt, _lib
    from oracle import gp_oracle as orc

    created = []

    class OracleHandle:                                     # the three members of _lib.Handle the driver uses
        def __init__(self, device=0):
            created.append(self)
            self.closed = False

        def lml_objective(self, X, Y, ls, c, noise, jitter, ktype=0):
            theta = np.log(np.concatenate([[c], np.atleast_1d(ls), [noise]]))
            val, grad = orc.log_marginal_likelihood(theta, X, Y, np.size(ls), alpha=jitter)
            if not np.isfinite(val):
                raise np.linalg.LinAlgError("not positive definite")
            return val, grad

        def close(self):
            self.closed = True

    monkeypatch.setattr(_lib, "Handle", OracleHandle)
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (60, 2))
    Y = np.column_stack([np.sin(4 * X[:, 0]), np.cos(3 * X[:, 1]) * X[:, 0]]) + 0.02 * rng.standard_normal((60, 2))
    for kernel in (C(1.0) * RBF([0.5, 0.5]) + WhiteKernel(0.01), C(1.0) * RBF(0.5) + WhiteKernel(0.01, "fixed")):
        out = {}
        for workers in ("1", "4"):
            monkeypatch.setenv("GPT_OPT_WORKERS", workers)
            gp = types.SimpleNamespace(_kernel_in=kernel, optimizer="fmin_l_bfgs_b", _handle=None, device=0, alpha=1e-10, X=X, Y=Y,
                                       _ktype=0, n_restarts_optimizer=3)
            p = kernel.get_params()
            np.random.seed(5)
            c, ls, noise, lml = hyperopt.optimize_hyperparameters(gp, p["k1__k1__constant_value"], np.atleast_1d(p["k1__k2__length_scale"]),
                                                                  p["k2__noise_level"])
            out[workers] = (c, ls, noise, lml, np.random.uniform())
        assert out["1"][0] == out["4"][0] and np.array_equal(out["1"][1], out["4"][1]) and out["1"][2:] == out["4"][2:]
        np.random.seed(5)
        ref = GaussianProcessRegressor(kernel=kernel, alpha=1e-10, n_restarts_optimizer=3).fit(X, Y)
        assert out["1"][3] == pytest.approx(ref.log_marginal_likelihood_value_, rel=1e-8)
        rp = ref.kernel_.get_params()
        assert out["1"][0] == pytest.approx(rp["k1__k1__constant_value"], rel=1e-4)
        assert_parity(out["1"][1], np.atleast_1d(rp["k1__k2__length_scale"]), 1e-4, "length-scales vs sklearn")
        assert out["1"][2] == pytest.approx(rp["k2__noise_level"], rel=1e-4)
    # every extra handle of a concurrent search is closed again; the owner's handle stays open
    extra_open = [h for h in created if not h.closed]
    assert len(extra_open) == 4          # one owner handle per optimize_hyperparameters call (2 kernels x 2 worker settings)


# ---------------------------------------------------------------------------------------------------------------
# Plan of the factor + inverse (csrc/gpt_fit_plan.h): host code, replayed here for every padded size and form
def _fit_op_accesses(f, NP):
    """(reads, writes) of one plan op as rectangles (matrix, row0, row1, col0, col1); scratch regions as ('S', a, b, 0, 1)."""
    kind = f["kind_name"]
    off, b, r = f["off"], f["n1"], f["n2"]
    diagK, diagW = ("K", off, off + b, off, off + b), ("W", off, off + b, off, off + b)
    S0, S1 = ("S", f["r0"], f["r0"] + f["r0_size"], 0, 1), ("S", f["r1"], f["r1"] + f["r1_size"], 0, 1)
    if kind == "POTRF":                  # right-looking: the columns and everything their trailing updates reach; L_kk parked in W
        tr = ("K", off, f["row_end"], off, f["row_end"])
        return [tr, diagW], [tr, diagW]
    if kind == "FINISH":
        return [diagK, diagW], [diagK, diagW]
    if kind == "TRINV":
        return [diagK, diagW, S0], [diagW, S0]
    if kind == "UPDATE":
        C = ("K", off, NP, off, off + b)
        return [("K", off, NP, f["k0"], f["k0"] + f["kw"]), C], [C]
    if kind == "TRSM":
        return [("K", off + b, NP, off, off + b), diagW], [S0]
    if kind == "COPY_L21":
        return [S0], [("K", off + b, NP, off, off + b)]
    if kind == "T":
        return [("K", off, off + b, 0, off), ("W", 0, off, 0, off)], [S1]
    if kind == "WFIN":
        return [diagW, S1], [("W", off, off + b, 0, off)]
    return [], []


def _overlap(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2] and a[3] < b[4] and b[3] < a[4]


def _trinv_extent(n):
    ext, sz = 0, 64                      # what trinv_levels really touches
    while sz < n:
        npairs = (n + 2 * sz - 1) // (2 * sz)
        m_last = n - (npairs - 1) * 2 * sz - sz
        nbp = npairs
        if m_last <= 0:
            nbp, m_last = npairs - 1, sz
        if nbp > 0:
            ext = max(ext, (nbp - 1) * sz * sz + min(m_last, sz) * sz)
        sz *= 2
    return ext


def _check_fit_plan(NP, form, panel, streams):
    from gaussian_process_transportation_amd import _lib
    pl = _lib.debug_fit_plan(NP, form, panel, streams)
    arena = pl["arena"]
    ops = []
    for row in pl["ops"]:
        f = dict(zip(_lib.FIT_OP_FIELDS, (int(v) for v in row)))
        f["kind_name"] = _lib.FIT_OP_KINDS[f["kind"]]
        ops.append(f)
    n = len(ops)
    if form < 0 and panel < 0 and streams < 0:
        assert pl["allocated"] >= arena          # what the fit workspace allocates for the plan the environment selects
    assert pl["allocated"] >= _lib.debug_fit_plan(NP, 0)["arena"]       # ... and for the one-leaf form it may fall back to
    # ---- regions inside the arena; what is factored, finished and inverted tiles the diagonal exactly
    cover = {"POTRF": [], "FINISH": [], "TRINV": []}
    upd = {}
    for f in ops:
        for reg in ("r0", "r1"):
            assert 0 <= f[reg] and f[reg] + f[reg + "_size"] <= arena, (NP, form, panel, f)
        off, b, r = f["off"], f["n1"], f["n2"]
        assert 0 <= off and off + b + r <= NP and b % 64 == 0
        if f["kind_name"] in cover:
            cover[f["kind_name"]].append((off, off + b))
        if f["kind_name"] == "POTRF":
            assert off + b <= f["row_end"] <= NP and f["row_end"] % 64 == 0
        if f["kind_name"] == "TRINV":
            assert f["r0_size"] >= _trinv_extent(b)
        if f["kind_name"] == "UPDATE":
            assert off + b + r == NP
            upd.setdefault(off, []).append((f["k0"], f["k0"] + f["kw"]))
        if f["kind_name"] == "TRSM":
            assert f["r0_size"] == b * r and off + b + r == NP
        if f["kind_name"] in ("T", "WFIN"):
            assert f["r1_size"] == b * off
    for k, cov in cover.items():
        cov.sort()
        if k == "TRINV" and pl["form"] == 1:
            continue                             # (the two halves; their off-diagonal block comes from T / WFIN)
        assert cov[0][0] == 0 and cov[-1][1] == NP and all(cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1)), (k, cov)
    if pl["form"] == 2:
        for off, _ in cover["POTRF"][1:]:
            ks = sorted(upd[off])
            assert ks[0][0] == 0 and ks[-1][1] == off and all(ks[i][1] == ks[i + 1][0] for i in range(len(ks) - 1)), "updates must cover [0, off) once"
        assert sum(f["kind_name"] == "WFIN" for f in ops) == len(cover["POTRF"]) - 1 == sum(f["kind_name"] == "T" for f in ops)
    if pl["form"] == 1:
        (h0, h1), (r0, r1) = cover["TRINV"]
        assert h0 == 0 and h1 == r0 and r1 == NP
        assert [(f["off"], f["n1"]) for f in ops if f["kind_name"] in ("T", "WFIN")] == [(h1, NP - h1)] * 2
    assert sum(f["kind_name"] == "FACTORED" for f in ops) == 1
    # ---- happens-before from (stream order + events) must order every pair of ops that touch overlapping memory
    hb = np.zeros((n, n), dtype=bool)
    last_on, recorded = {}, {}
    for i, f in enumerate(ops):
        if f["stream"] in last_on:
            hb[last_on[f["stream"]], i] = True
        last_on[f["stream"]] = i
        for w in ("wait0", "wait1", "wait2"):
            if f[w] >= 0:
                assert f[w] in recorded, "wait for an event nobody has recorded yet"
                hb[recorded[f[w]], i] = True
        if f["record"] >= 0:
            assert f["record"] not in recorded and f["record"] < pl["n_events"] <= 256
            recorded[f["record"]] = i
    for k in range(n):                           # transitive closure (ops are in issue order: edges go forward)
        hb[:, :] |= np.outer(hb[:, k], hb[k, :])
    acc = [_fit_op_accesses(f, NP) for f in ops]
    for i in range(n):
        for j in range(i + 1, n):
            if hb[i, j]:
                continue
            ri, wi = acc[i]
            rj, wj = acc[j]
            clash = any(_overlap(a, b) for a in wi for b in rj + wj) or any(_overlap(a, b) for a in ri for b in wj)
            assert not clash, (NP, form, panel, "unordered ops touch the same memory", ops[i], ops[j])
    # the caller continues in the main stream: its last op must come after everything
    main = [i for i, f in enumerate(ops) if f["stream"] == 0]
    assert all(i == main[-1] or hb[i, main[-1]] for i in range(n)), "work left unjoined when launch_factor_inverse returns"
    return pl, ops


def test_fit_plan_regions_and_ordering_for_every_size_and_form(monkeypatch):
    """Round 3 ended with a GPU memory fault from a scratch layout that assumed a half split.  Layout and
    cross-stream order of the factor + inverse are now DATA (csrc/gpt_fit_plan.h), and this replays them for every padded size up
    to 16384, the three forms, four panel widths and the group widths that move form 1's split off the half — every region
    inside the arena (and inside what the workspace allocates), the factored / finished / inverted blocks tiling the diagonal,
    form 2's updates covering the columns in front of each panel exactly once, and the happens-before relation of (stream order
    + events) ordering EVERY pair of operations that touch overlapping memory (K, W or scratch), with the CU-masked streams and
    in the serial order; nothing left unjoined at the end."""
    seen = {0: 0, 1: 0, 2: 0}
    for NP in range(512, 16384 + 1, 512):
        for form in (-1, 0, 1, 2):
            for panel in ((512, 1024, 1536, 2048) if form == 2 else (-1,)):
                for streams in (1, 0):
                    if form == 2 and NP > 8192 and panel == 512:
                        continue                 # (keeps the closure small)
                    pl, _ = _check_fit_plan(NP, form, panel, streams)
                    seen[pl["form"]] += 1
    assert min(seen.values()) > 40
    # the shipped choice: split where the second half is chain-bound, one leaf elsewhere
    assert [_lib_form(NP) for NP in (512, 4096, 4608, 8192, 12288, 12800, 16384)] == [0, 0, 1, 1, 1, 0, 0]
    pl, ops = _check_fit_plan(8192, -1, -1, -1)
    assert pl["form"] == 1 and pl["side_eighths"] == 5 and {f["stream"] for f in ops} == {0, 1, 2}
    assert [f["off"] for f in ops if f["kind_name"] == "T"] == [4096]
    assert _check_fit_plan(5632, -1, -1, -1)[0]["side_eighths"] == 4
    # group widths that move the split off the half (GPT_POTRF_GROUP = 4: NP = 5632 -> h = 2560, r = 3072 — the shape of round 3's
    # fault): the arena follows the split
    monkeypatch.setenv("GPT_POTRF_GROUP", "4")
    for NP in range(4608, 12288 + 1, 512):
        pl, ops = _check_fit_plan(NP, -1, -1, -1)
        if pl["form"] == 1:
            h = [f["off"] for f in ops if f["kind_name"] == "T"][0]
            assert h % (4 * 128) == 0 and pl["arena"] >= (NP - h) * h + _trinv_extent(max(h, NP - h))
    pl, ops = _check_fit_plan(5632, -1, -1, -1)
    assert [f["off"] for f in ops if f["kind_name"] == "T"] == [2560]
    monkeypatch.delenv("GPT_POTRF_GROUP")
    pl, ops = _check_fit_plan(8192, 2, 1024, 1)
    kinds = [f["kind_name"] for f in ops]
    assert kinds.count("POTRF") == 8 and kinds.count("UPDATE") == 7 + 6 and kinds.count("TRSM") == 7      # 8 panels; 7 last + 6 look-ahead updates
    assert {f["stream"] for f in ops} == {0, 1, 2} and {f["stream"] for f in _check_fit_plan(8192, 2, 1024, 0)[1]} == {0}


def _lib_form(NP):
    from gaussian_process_transportation_amd import _lib
    return _lib.debug_fit_plan(NP)["form"]


class _FakeHandle:
    """Stands in for _lib.Handle in the CPU test of the device group: records calls, predicts f(x) row by row."""
    made = []

    def __init__(self, device=0):
        self.device = device
        self.calls = []
        self.replica_of = None
        _FakeHandle.made.append(self)

    def fit(self, *a, **k):
        self.calls.append("fit")

    def factor_copy_from(self, src):
        self.replica_of = src

    def model_info(self):
        return 1, 0

    def predict_all(self, Xq, mean=False, var=False, J=False, Jvar=False, dvar=False):
        import threading
        self.calls.append(("predict", Xq.shape[0], threading.current_thread().name))
        if np.any(Xq[:, 0] < -1e6):
            raise ValueError("poisoned shard")
        M, D = Xq.shape
        return {"mean": np.stack([Xq.sum(1), Xq[:, 0]], 1) if mean else None, "var": (Xq ** 2).sum(1) if var else None,
                "J": np.repeat(Xq[:, None, :], 2, axis=1) if J else None, "Jvar": Xq * 3.0 if Jvar else None,
                "dvar": (Xq * 5.0).T.copy() if dvar else None}

    def close(self):
        self.calls.append("close")

    def export(self):
        return "exported-by-%d" % self.device


def test_device_group_shards_rows_over_handles_and_assembles(monkeypatch):
    """GaussianProcess(devices=[...]): host plumbing of the one-process multi-GPU form — fit on the
    first handle, one replica copy per further device, contiguous balanced row shards on a thread per device, outputs
    assembled in row order (dvar along its second axis), small batches left to the first device, errors propagated."""
    from gaussian_process_transportation_amd import _lib, device_group
    _FakeHandle.made = []
    monkeypatch.setattr(_lib, "Handle", _FakeHandle)
    g = device_group.DeviceGroup([0, 1, 2])
    h0, h1, h2 = g.handles
    with pytest.raises(_lib.GptError):
        g.predict_all(np.zeros((5000, 3)), mean=True)
    g.fit("X", "Y")
    assert h0.calls == ["fit"] and h1.replica_of is h0 and h2.replica_of is h0 and h1.calls == []
    assert g.export() == "exported-by-0"                      # everything but predictions: the first handle
    rng = np.random.default_rng(0)
    X = rng.standard_normal((5001, 3))
    out = g.predict_all(X, mean=True, var=True, J=True, Jvar=True, dvar=True)
    ref = _FakeHandle(9).predict_all(X, mean=True, var=True, J=True, Jvar=True, dvar=True)
    for k in ref:
        assert np.array_equal(out[k], ref[k]), k
    sizes = [c[1] for h in (h0, h1, h2) for c in h.calls if isinstance(c, tuple)]
    assert sizes == [1667, 1667, 1667]
    assert all(c[2].startswith("gpt-dev") for h in (h0, h1, h2) for c in h.calls if isinstance(c, tuple))     # the group's worker threads
    assert g.predict_all(X, var=True)["mean"] is None
    small = g.predict_all(X[:100], mean=True)                  # not worth a hand-off: first device alone
    assert np.array_equal(small["mean"], ref["mean"][:100]) and [c for c in h1.calls if isinstance(c, tuple)][-1][1] == 1667
    bad = X.copy(); bad[4000, 0] = -1e9
    with pytest.raises(ValueError, match="poisoned"):
        g.predict_all(bad, mean=True)
    g.close()
    assert h2.calls[-1] == "close"
    # the class surface: devices= reaches the regressor; one device keeps the plain handle
    from gaussian_process_transportation_amd import GaussianProcessTransportation
    t = GaussianProcessTransportation(optimizer=None, devices=[0, 1], verbose=False)
    assert t.method.delta_map.devices == [0, 1] and t.method.delta_map.device == 0
    with pytest.raises(ValueError):
        GaussianProcessTransportation(devices=[])
