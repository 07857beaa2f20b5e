"""bench.py's launcher contract, checked without a GPU: `--gpus N` either runs N ranks or refuses loudly."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_gpus_n_without_enough_devices_refuses_instead_of_running_one_rank():
    """VERDICT r01 / ADVICE: `python bench.py --gpus 2` used to run ONE rank and print n_gpus: 1."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing" in r.stderr
    assert r.stdout.strip() == ""            # no JSON line that could be mistaken for a 2-GPU result


def test_world_size_mismatch_refuses():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_byte_and_flop_formulas():
    """100,840 B / problem and 195,264 B / problem (KKT) at the headline shape; 17.3 / 256 kFLOP."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes_per_problem(12, 4, 50) == 100840
    assert bench.kkt_algorithmic_bytes_per_problem(12, 4, 50) == 195264
    assert abs(bench.algorithmic_flops_per_problem_step(12, 4) - 17290) < 30
    assert abs(bench.algorithmic_flops_per_problem_step(32, 8) - 255862) < 400


def test_parity_object_accepts_the_oracles_own_numbers_and_rejects_a_wrong_trajectory():
    """bench.py's in-run self-certification: `parity.ok` for results within rtol 1e-3 /
    atol 1e-4 of the oracle on the first 64 problems of the timed batch, not ok (and the run exits non-zero) otherwise."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import lqr_oracle as O
    rng = np.random.default_rng(3)
    T, B, ns, nc = 12, 70, 12, 4
    n = ns + nc
    A = rng.standard_normal((T, B, n, n)).astype(np.float32)
    C = np.einsum("tbji,tbjk->tbik", A, A)
    c = rng.standard_normal((T, B, n)).astype(np.float32)
    F = np.concatenate((np.eye(ns) + 0.2 * rng.standard_normal((T - 1, B, ns, ns)) / np.sqrt(ns),
                        rng.standard_normal((T - 1, B, ns, nc)) / np.sqrt(ns)), 3).astype(np.float32)
    f = (0.1 * rng.standard_normal((T - 1, B, ns))).astype(np.float32)
    x0 = rng.standard_normal((B, ns)).astype(np.float32)
    u = np.zeros((T, B, nc), np.float32)
    x, _ = O.traj_cost(x0.astype(np.float64), u.astype(np.float64), F.astype(np.float64), f.astype(np.float64))
    p = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in dict(x_init=x0, C=C, c=c, F=F, f=f, cur_x=x.astype(np.float32), cur_u=u).items()}
    o = O.lqr_step(*(p[k].numpy().astype(np.float64) for k in ("x_init", "C", "c", "F", "f", "cur_x", "cur_u")), None, None, lockstep=False)
    r = {k: torch.from_numpy(o[k].astype(np.float32)) for k in ("new_x", "new_u", "costs", "alphas")}
    good = bench.parity_check(p, r, False)
    assert good["ok"] and good["problems"] == 64 and good["max_err_over_tol_u"] < 0.1
    r["new_u"][3, 5, 1] += 0.01
    bad = bench.parity_check(p, r, False)
    assert not bad["ok"] and bad["max_err_over_tol_u"] > 1.0


def _small_headline_problem(B, T=12, bounded=False, seed=3):
    import numpy as np
    from oracle import lqr_oracle as O
    rng = np.random.default_rng(seed)
    ns, nc = 12, 4
    n = ns + nc
    A = rng.standard_normal((T, B, n, n)).astype(np.float32)
    C = np.einsum("tbji,tbjk->tbik", A, A)
    c = rng.standard_normal((T, B, n)).astype(np.float32)
    F = np.concatenate((np.eye(ns) + 0.2 * rng.standard_normal((T - 1, B, ns, ns)) / np.sqrt(ns),
                        rng.standard_normal((T - 1, B, ns, nc)) / np.sqrt(ns)), 3).astype(np.float32)
    f = (0.1 * rng.standard_normal((T - 1, B, ns))).astype(np.float32)
    x0 = rng.standard_normal((B, ns)).astype(np.float32)
    u = np.clip(0.3 * rng.standard_normal((T, B, nc)), -1, 1).astype(np.float32) if bounded else np.zeros((T, B, nc), np.float32)
    x, _ = O.traj_cost(x0.astype(np.float64), u.astype(np.float64), F.astype(np.float64), f.astype(np.float64))
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in dict(x_init=x0, C=C, c=c, F=F, f=f, cur_x=x.astype(np.float32), cur_u=u).items()}


def test_extra_rows_certify_themselves_box_constrained_step_and_backward():
    """Round 4: every `extra` row that times a step or a backward carries a `parity` object.  The two
    checkers, fed the oracle's own float32-rounded numbers (ok) and a spoiled copy (not ok): the box-constrained step with
    its tie accounting, and the five gradients of the backward."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import lqr_oracle as O
    p = _small_headline_problem(40, bounded=True)
    h = [p[k].numpy().astype(np.float64) for k in ("x_init", "C", "c", "F", "f", "cur_x", "cur_u")]
    o = O.lqr_step(*h, -1.0, 1.0, lockstep=False)
    r = {k: torch.from_numpy(o[k].astype(np.float32)) for k in ("new_x", "new_u", "costs", "alphas")}
    good = bench.parity_check(p, r, True, n=bench.PARITY_ROW)
    assert good["ok"] and good["problems"] == 32 and good["active_set_ties"] == 0 and good["line_search_ties"] == 0

    class SecondSolve:
        """stands in for the HIP library's second solve of the problems in question (gains written out)"""
        def __init__(self, flip):
            self.flip, self.calls = flip, 0

        def lqr_step(self, x_init, C, c, F, f, cur_x, cur_u, opts, want_gains=False, **kw):
            assert want_gains and opts.u_lower == -1.0 and opts.u_upper == 1.0
            self.calls += 1
            oo = O.lqr_step(*(t.numpy().astype(np.float64) for t in (x_init, C, c, F, f, cur_x, cur_u)), -1.0, 1.0, lockstep=False, return_gains=True)
            K = oo["K"].astype(np.float32)
            if self.flip:          # the kernel's float32 run clamped a control the float64 run left free (or the other way round)
                free = np.argwhere(~(K == 0).all(axis=-1))[0]
                K[free[0], :, free[2]] = 0.0
            return {"K": torch.from_numpy(K)}
    # (round 5) ONE problem off: excused as an active-set tie only when CONFIRMED -- a second solve of that problem
    # with the gains written out must clamp a different set of controls than the float64 run somewhere on the horizon
    r["new_u"][2, 4, 0] += 0.05
    assert not bench.parity_check(p, r, True, n=bench.PARITY_ROW)["ok"]                        # nobody to confirm it: a failure
    same_sets = SecondSolve(flip=False)
    one = bench.parity_check(p, r, True, n=bench.PARITY_ROW, be=same_sets)
    assert not one["ok"] and one["out_of_tolerance_unexplained"] == 1 and one["active_set_ties"] == 0 and same_sets.calls == 1
    other_sets = SecondSolve(flip=True)
    one = bench.parity_check(p, r, True, n=bench.PARITY_ROW, be=other_sets)
    assert one["ok"] and one["active_set_ties"] == 1 and one["out_of_tolerance_unexplained"] == 0
    # ... three confirmed ties are still a failure
    for b in (7, 9):
        r["new_u"][2, b, 0] += 0.05
    three = bench.parity_check(p, r, True, n=bench.PARITY_ROW, be=SecondSolve(flip=True))
    assert not three["ok"] and three["active_set_ties"] == 3
    # ... and a confirmed tie that is off AND worse than the nominal is no tie at all
    r = {k: torch.from_numpy(o[k].astype(np.float32)) for k in ("new_x", "new_u", "costs", "alphas")}
    r["new_u"][2, 4, 0] += 0.05
    r["costs"][4] = float(o["old_costs"][4]) + 10.0
    assert not bench.parity_check(p, r, True, n=bench.PARITY_ROW, be=SecondSolve(flip=True))["ok"]
    # slices of one batch: the worst of them decides
    r = {k: torch.from_numpy(o[k].astype(np.float32)) for k in ("new_x", "new_u", "costs", "alphas")}
    sl = bench.parity_slices(p, r, True, [(0, 8), (16, 8), (32, 8)])
    assert sl["ok"] and sl["problems"] == 24 and [d["first_problem"] for d in sl["slices"]] == [0, 16, 32]
    r["new_u"][1, 35, 2] += 0.05
    assert not bench.parity_slices(p, r, True, [(0, 8), (16, 8), (32, 8)])["ok"]
    # the backward
    nx, nu = torch.from_numpy(o["new_x"].astype(np.float32)), torch.from_numpy(o["new_u"].astype(np.float32))
    g = torch.Generator().manual_seed(1)
    gx, gu = torch.randn(nx.shape, generator=g), torch.randn(nu.shape, generator=g)
    ob = O.kkt_backward(h[1], h[2], h[3], h[4], nx.numpy().astype(np.float64), nu.numpy().astype(np.float64), gx.numpy().astype(np.float64),
                        gu.numpy().astype(np.float64), -1.0, 1.0, lockstep=False)
    grads = {k: torch.from_numpy(ob[k].astype(np.float32)) for k in ("dx_init", "dC", "dc", "dF", "df")}
    ok = bench.kkt_parity_check(p, nx, nu, gx, gu, grads, True)
    assert ok["ok"] and ok["problems"] == bench.PARITY_KKT and set(ok["worst_rel"]) == {"dx_init", "dC", "dc", "dF", "df"}
    grads["dF"][3, 2, 1, 1] += 1.0
    assert not bench.kkt_parity_check(p, nx, nu, gx, gu, grads, True)["ok"]


def test_cpu_baseline_times_the_unmodified_reference_where_it_is_on_the_box():
    """north_star: "the reference timed on the host cores of the same box (core count stated) in the same run".  bench.py tries:
    with $MPC_REFERENCE_DIR or /root/reference on the box, the unmodified LQRStep forward runs in a child interpreter on a chunk
    of the timed batch and its float32 results are held against the timed results (here: the oracle's numbers stand in for
    the kernel's).  The GPU box has no reference: the function then returns (None, None) and the port stands alone."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import lqr_oracle as O
    if bench.reference_dir() is None:
        assert bench.reference_cpu_baseline({}, False) == (None, None)
        pytest.skip("no reference on this box")
    p = _small_headline_problem(24, T=10)
    row, o = bench.reference_cpu_baseline(p, False, reps=1)
    assert o is not None, row
    assert row["kind"] == "reference" and row["cores"] >= 1 and row["value"] > 0 and "UNMODIFIED" in row["sample"]
    oo = O.lqr_step(*(p[k].numpy().astype(np.float64) for k in ("x_init", "C", "c", "F", "f", "cur_x", "cur_u")), None, None, lockstep=False)
    r = dict(new_x=torch.from_numpy(oo["new_x"]), new_u=torch.from_numpy(oo["new_u"]))
    par = bench.reference_parity(o, r, 24, False)
    assert par["ok"] and par["asserted"] and par["share_within_tol"] == 1.0
    r["new_u"][1, 2, 3] += 0.02
    assert not bench.reference_parity(o, r, 24, False)["ok"]


def test_contract_line_fits_the_driver_whatever_the_run_recorded():
    """the one stdout line had grown to 31.8 KB and the driver recorded `parsed: null`.  The line is now built by
    bench.contract_line: <= 4 KB, `roofline` and `cpu_baseline` inside, the secondary rows as a digest -- held here on the
    largest full record there is (round 5's), on the same record with every string and row count inflated, and on an N > 1
    record."""
    import copy
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line_final.json")))
    assert len(json.dumps(full)) > 20000
    s = bench.contract_line(full)
    assert "\n" not in s and len(s) < 4096
    d = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-4) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    assert d["roofline"]["achieved"] / d["roofline"]["peak"] == pytest.approx(d["roofline"]["frac"], rel=1e-3)
    assert d["roofline"]["traffic"] and d["roofline"]["same_set"]["frac"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and "reference_probe" not in d["cpu_baseline"]
    assert d["parity"]["ok"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert set(d["extra_digest"]) == set(full["extra"]) and d["extra_digest"]["lqr_step_bounded"][2] is True
    # a run that recorded far more: the line sheds its optional parts instead of growing
    big = copy.deepcopy(full)
    for i in range(400):
        big["extra"]["another_row_with_a_long_name_%03d" % i] = copy.deepcopy(full["extra"]["lqr_step_bounded"])
    big["config"]["workload"] = big["config"]["workload"]
    big["cpu_baseline"]["sample"] = "x" * 5000
    s2 = bench.contract_line(big)
    d2 = json.loads(s2)
    assert len(s2) <= 4096 and "roofline" in d2 and "cpu_baseline" in d2 and d2["value"] == d["value"]
    # N > 1: the rows of the distributed line are digested too
    multi = copy.deepcopy(full)
    multi["n_gpus"] = 8
    multi.pop("cpu_baseline")
    multi["extra"] = {"strong_scaling": {"ms_per_step": 0.05, "value": 4.0e9, "roofline": {"frac": 0.2}, "parity_all_ranks_ok": True,
                                         "collective": "y" * 300},
                      "cfg5": {"ms_per_step": 0.4, "value": 1.3e9, "roofline": {"frac": 0.3}, "parity_all_ranks_ok": True}}
    d3 = json.loads(bench.contract_line(multi))
    assert d3["extra_digest"]["strong_scaling"] == [0.05, 0.2, True, 4.0e9] and "cpu_baseline" not in d3


@pytest.mark.gpu
def test_bench_stdout_is_one_short_json_line():
    """The driver's own call, shortened (no secondary rows): stdout is exactly one line, < 4 KB, json.loads gives `roofline`
    and `cpu_baseline`; the full record is on stderr and in bench_full.json."""
    import json
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        r = _run(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extra"], {"MPC_BENCH_RECORD_DIR": td})
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        assert len(lines) == 1 and len(lines[0]) < 4096
        d = json.loads(lines[0])
        assert d["roofline"]["frac"] > 0.2 and d["cpu_baseline"]["value"] > 0 and d["parity"]["ok"]
        assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1
        full = json.load(open(os.path.join(td, "bench_full.json")))
        assert full["value"] == pytest.approx(d["value"], rel=1e-4) and "bench.py full record: " in r.stderr
