"""Comparison helpers shared by the parity tests, smoke() and bench.py.

Two kinds of comparison are needed because of a property of the REFERENCE ALGORITHM itself
(see DESIGN.md "Numerical reproducibility of the reference algorithm"):

* strict() - per voxel, for well-conditioned problems (polynomial, design-matrix and
  single-exponential models; one VB iteration of any model from an identical state). The
  tolerances are set by the reference's own numerical Jacobian: a central difference with step
  1e-5|theta| (floor 1e-10) amplifies last-bit differences of the model prediction by
  |f| / (2 delta), i.e. up to ~1e-6 relative in J and more in the covariance.

* population() - for the bi-exponential fit. Its first iterations start from two identical
  decay rates, J'J is numerically singular, and the iteration map amplifies rounding noise by
  ~1e11 within four iterations: two CPU builds of the same oracle source (with / without FMA
  contraction) already disagree by more than 1e-4 on a quarter of the voxels after 50
  iterations. Per-voxel agreement with "the" CPU result is therefore not a property any
  non-bit-identical implementation (or the reference itself on another compiler) can have; what
  can and must agree is the population: share of voxels that agree, share of failed voxels,
  distribution of the fitted noise precision / free energy.
"""
import hashlib
import os

import numpy as np

import oracle

TOL_MEAN = 1e-6   # |d mean| <= TOL_MEAN * max(|mean|, posterior sd)
TOL_COV = 2e-4    # |d cov_ij| <= TOL_COV * sd_i sd_j   (finite-difference noise, see above)
TOL_F = 1e-6      # |dF| <= TOL_F * max(1, |F|)
NORTH_STAR = 1e-4  # BASELINE.json: posterior means within 1e-4 relative
FLOOR_FACTOR = 3   # a comparison that opted in may use this many times the OBSERVED CPU-vs-CPU difference (round 2: 10)


def voxel_errors(holder, a, b, mask=None):
    """Per-voxel scaled errors between result dicts a (CPU) and b (GPU)."""
    cfg = holder.cfg
    P, n = cfg.n_params, cfg.n_params + holder.n_noise_outputs
    if mask is None:
        mask = np.ones(a["mvn"].shape[1], dtype=bool)
    cov_a, mean_a = oracle.unpack_mvn(a["mvn"][:, mask], n)
    cov_b, mean_b = oracle.unpack_mvn(b["mvn"][:, mask], n)
    sd = np.sqrt(np.abs(np.einsum("vii->vi", cov_a)))

    def ratio(num, den):
        # 0 / 0 (an entry with zero spread that both sides reproduce exactly) counts as no error; a
        # NaN or an infinity on ONE side only must not disappear in a nan-aware maximum: it is inf
        with np.errstate(divide="ignore", invalid="ignore"):
            q = num / den
        q = np.where((num == 0) & (den == 0), 0.0, q)
        return np.where(np.isnan(q), np.inf, q)

    same_nonfinite = (~np.isfinite(mean_a)) & ((mean_a == mean_b) | (np.isnan(mean_a) & np.isnan(mean_b)))
    e_mean = np.where(same_nonfinite, 0.0, ratio(np.abs(mean_a - mean_b), np.maximum(np.abs(mean_a), sd))).max(axis=1)
    same_cov = (~np.isfinite(cov_a)) & ((cov_a == cov_b) | (np.isnan(cov_a) & np.isnan(cov_b)))
    e_cov = np.where(same_cov, 0.0, ratio(np.abs(cov_a - cov_b), sd[:, :, None] * sd[:, None, :])).reshape(len(sd), -1).max(axis=1)
    rel = np.where(same_nonfinite[:, :P], 0.0,
                   ratio(np.abs(mean_a[:, :P] - mean_b[:, :P]), np.maximum(np.abs(mean_a[:, :P]), 1e-12))).max(axis=1)
    return e_mean, e_cov, rel


RAISED = []  # (what, dict of the tolerances applied) for every strict() call that ran at a raised bound


def strict(holder, a, b, what="", tol_mean=TOL_MEAN, tol_cov=TOL_COV, check_f=None, allow_iter_mismatch=0, cpu2=None,
           tol_f=TOL_F, allow_floor=False):
    """Per-voxel parity: status and iteration counts identical, every value finite where the CPU's is,
    values within tolerance. Returns the observed errors AND the tolerances that were applied.

    cpu2 (optional) = result of the second CPU build (oracle.run_fma). If two CPU builds of the
    same source already differ by more than the base tolerance on this problem (the
    finite-difference Jacobian amplifies rounding, see module docstring), the tolerance MAY be
    raised to 3x that measured floor - never beyond the north-star bound of 1e-4 on the means
    (1e-2 sd_i sd_j on covariances) - but only for a caller that opted in with allow_floor=True:
    a test that needs the raised bound without having asked for it fails, and every call that ran
    at a raised bound is recorded in RAISED (tests/conftest.py prints them at the end of the run;
    DESIGN.md section 5.2 lists the tests that opt in)."""
    cfg = holder.cfg
    base = dict(tol_mean=tol_mean, tol_cov=tol_cov, tol_f=tol_f)
    floor = dict(tol_mean=0.0, tol_cov=0.0, tol_f=0.0)
    # (cpu2 may be a list: several other CPU builds of the same source - FMA contraction, an exp of the device's
    # accuracy class (oracle.run_*_exp1ulp) -, the floor is the largest distance any of them has from `a`)
    for other in (cpu2 if isinstance(cpu2, (list, tuple)) else ([] if cpu2 is None else [cpu2])):
        okf = (a["status"] == 0) & (other["status"] == 0) & (a["iterations"] == other["iterations"])
        if okf.any():
            f_mean, f_cov, _ = voxel_errors(holder, a, other, okf)
            floor["tol_mean"] = max(floor["tol_mean"], min(FLOOR_FACTOR * float(f_mean.max()), NORTH_STAR))
            floor["tol_cov"] = max(floor["tol_cov"], min(FLOOR_FACTOR * float(f_cov.max()), 1e-2))
            if cfg.need_f:
                Fa, Fc = a["free_energy"][okf], other["free_energy"][okf]
                floor["tol_f"] = max(floor["tol_f"], min(FLOOR_FACTOR * float(np.max(np.abs(Fa - Fc) / np.maximum(1.0, np.abs(Fa)))), 1e-3))
    assert np.array_equal(a["status"], b["status"]), (what, "status", np.flatnonzero(a["status"] != b["status"])[:8])
    n_it = int(np.count_nonzero(a["iterations"] != b["iterations"]))
    assert n_it <= allow_iter_mismatch, (what, "iterations differ on %d voxels" % n_it)
    ok = (a["status"] == 0) & (a["iterations"] == b["iterations"])
    out = dict(err_means=0.0, err_cov=0.0, rel_means=0.0, err_f=0.0, raised=False, **base)
    if not ok.any():
        return out
    e_mean, e_cov, rel = voxel_errors(holder, a, b, ok)
    out.update(err_means=float(e_mean.max()), err_cov=float(e_cov.max()), rel_means=float(rel.max()))
    errs = dict(tol_mean=out["err_means"], tol_cov=out["err_cov"])
    if check_f if check_f is not None else bool(cfg.need_f):
        Fa, Fb = a["free_energy"][ok], b["free_energy"][ok]
        assert np.all(np.isfinite(Fb) | ~np.isfinite(Fa)), (what, "non-finite F")
        out["err_f"] = float(np.max(np.abs(Fa - Fb) / np.maximum(1.0, np.abs(Fa))))
        errs["tol_f"] = out["err_f"]
    label = dict(tol_mean="means", tol_cov="cov", tol_f="F")
    for k, err in errs.items():
        if err <= base[k]:
            continue
        # beyond the base tolerance: only acceptable below the measured CPU-vs-CPU floor, and only on request
        assert err <= floor[k], (what, label[k], err, "base tolerance %g, 3 x CPU-vs-CPU floor %g" % (base[k], floor[k]))
        assert allow_floor or os.environ.get("PARITY_RECORD_ONLY"), (
            what, label[k], err, "within 3 x the CPU-vs-CPU floor (%g) but beyond the base tolerance %g, and the "
            "test did not opt in with allow_floor=True" % (floor[k], base[k]))
        out[k] = floor[k]
        out["raised"] = True
    if out["raised"]:
        RAISED.append((what, {k: out[k] for k in base}, {k: errs.get(k, 0.0) for k in base}))
    return out


def population_stats(holder, a, b):
    """Agreement statistics between two runs of a chaotic problem."""
    cfg = holder.cfg
    n = cfg.n_params + holder.n_noise_outputs
    off = n * (n + 1) // 2
    both = (a["status"] == 0) & (b["status"] == 0)
    e_mean, e_cov, rel = voxel_errors(holder, a, b, both)
    phi_a = a["mvn"][off + cfg.n_params][a["status"] == 0]
    phi_b = b["mvn"][off + cfg.n_params][b["status"] == 0]
    q = [0.1, 0.5, 0.9]
    return dict(
        frac_within_1e4=float(np.mean(e_mean <= NORTH_STAR)),
        frac_within_1e4_rel=float(np.mean(rel <= NORTH_STAR)),  # |a - b| / max(|b|, 1e-12), model parameters
        frac_within_1e6=float(np.mean(e_mean <= 1e-6)),
        median_err=float(np.median(e_mean)),
        bad_a=float(np.mean(a["status"] != 0)), bad_b=float(np.mean(b["status"] != 0)),
        phi_quantiles_a=np.quantile(phi_a, q), phi_quantiles_b=np.quantile(phi_b, q),
        iters_a=float(np.mean(a["iterations"])), iters_b=float(np.mean(b["iterations"])),
    )


def population(holder, cpu, gpu, floor, what="", slack=None):
    """gpu-vs-cpu agreement must be as good as cpu-vs-cpu' agreement (floor = population_stats
    of the two CPU builds). Both are shares of a finite sample of voxels of a chaotic fit - which voxel lands on
    which fixed point is a coin that every build tosses anew - so the comparison allows the sampling noise of such
    a share and nothing else: 3 sigma of a binomial share p over n voxels (round 2: a flat 5 points). Where a
    binary128 ground truth exists the bar is no_worse_than_the_cpu_builds instead, without any allowance."""
    s = population_stats(holder, cpu, gpu)
    n = max(1, int(np.count_nonzero((cpu["status"] == 0) & (gpu["status"] == 0))))

    def allowance(p):
        return slack if slack is not None else 3.0 * float(np.sqrt(max(p * (1 - p), 1.0 / n) / n))
    assert s["frac_within_1e4"] >= floor["frac_within_1e4"] - allowance(floor["frac_within_1e4"]), (what, s, floor)
    assert s["frac_within_1e6"] >= floor["frac_within_1e6"] - allowance(floor["frac_within_1e6"]), (what, s, floor)
    assert s["median_err"] <= max(10 * floor["median_err"], 1e-7), (what, s, floor)
    assert s["bad_b"] <= max(2 * max(floor["bad_a"], floor["bad_b"]), 0.005), (what, s, floor)
    assert np.allclose(s["phi_quantiles_b"], s["phi_quantiles_a"], rtol=0.03), (what, s)
    assert abs(s["iters_b"] - s["iters_a"]) <= 0.05 * max(s["iters_a"], 1.0) + 0.5, (what, s)
    return s


# ---- binary128 ground truth for the bi-exponential fit (tests/golden/make_c3_truth.py) ---------------
def load_c3_truth():
    """The committed ground truth; checks that cases.exp_problem still generates the series it was
    computed from."""
    import sys
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if gdir not in sys.path:
        sys.path.insert(0, gdir)
    import make_c3_truth as mt
    f = np.load(os.path.join(gdir, "c3_truth_binary128.npz"))
    V = int(f["n_voxels"])
    _, y = mt.problem(V)
    assert hashlib.sha256(y.tobytes()).hexdigest() == str(f["data_sha256"]), "the seeded series changed"
    return dict(mvn=f["mvn"], status=f["status"], iterations=f["iterations"], its=[int(i) for i in f["its"]],
                trace_means=f["trace_means"], n_voxels=V, free_energy=f["free_energy"])


def truth_stats(holder, truth, r, with_f=False):
    """Distribution of the per-voxel error of result r against the ground truth: the scaled metric of
    voxel_errors and SURVEY 8d's pure relative one, over the voxels r finished. with_f: also the free
    energy's error |F - F_truth| / max(1, |F_truth|); truth["iterations"] present and r run with an F-driven
    detector: also the share of voxels that stopped after the truth's number of iterations."""
    ok = (r["status"] == 0) & (truth["status"] == 0)
    e, e_cov, rel = voxel_errors(holder, truth, r, ok)
    q = lambda x, p: float(np.quantile(x, p))
    out = dict(failed=float(np.mean(r["status"] != 0)), within_1e4=float(np.mean(e <= NORTH_STAR)),
               within_1e6=float(np.mean(e <= 1e-6)), within_1e4_rel=float(np.mean(rel <= NORTH_STAR)),
               median=q(e, 0.5), p75=q(e, 0.75), p90=q(e, 0.9), p99=q(e, 0.99),
               median_rel=q(rel, 0.5), p90_rel=q(rel, 0.9), median_cov=q(e_cov, 0.5))
    if with_f:
        # The free energy of a voxel that ended on another fixed point than the truth's is another number altogether, and
        # which voxels do is the coin every build tosses anew (the shares above). What an implementation OWES is F where
        # its posterior is the truth's: the error of F over the voxels whose posterior is within 1e-6 of the truth.
        Ft, Fr = truth["free_energy"][ok], r["free_energy"][ok]
        with np.errstate(invalid="ignore"):
            ef = np.abs(Fr - Ft) / np.maximum(1.0, np.abs(Ft))
        ef = np.where(np.isnan(ef), np.inf, ef)
        there = e <= 1e-6
        out["f_all_within_1e4"] = float(np.mean(ef <= 1e-4))  # (reported, not compared: it follows the coin)
        if there.any():
            efc = ef[there]
            out.update(f_within_1e6=float(np.mean(efc <= 1e-6)), f_within_1e8=float(np.mean(efc <= 1e-8)), f_median=q(efc, 0.5),
                       f_p75=q(efc, 0.75), f_p90=q(efc, 0.9), f_p99=q(efc, 0.99))
        else:
            out.update(f_within_1e6=1.0, f_within_1e8=1.0, f_median=0.0, f_p75=0.0, f_p90=0.0, f_p99=0.0)
        if "iterations" in truth:
            out["same_iterations"] = float(np.mean(r["iterations"][ok] == truth["iterations"][ok]))
    return out


def no_worse_than_the_cpu_builds(gpu, cpu1, cpu2, what="", with_f=False, median_factor=1.5, tail_factor=1.0):
    """The bar where a binary128 ground truth exists (tests/golden/make_c*_truth.py): the kernels' error against what
    the ALGORITHM computes must be no worse than that of the worse of two fp64 CPU builds of the oracle - NO slack -
    in the shares of voxels within 1e-4 / 1e-6 of the truth, in the 75th / 90th / 99th percentile of the error and in
    the share of failed voxels (and the same for F). One stated exception: the MEDIAN of the final error, ~5e-10 on a
    converged voxel, may be median_factor x the CPU's (the lane kernels' exponentials carry up to 7 extra roundings
    between two exact evaluations, vb_models.h: 6e-10 against 5e-10, five orders below the north star)."""
    shares = ["within_1e4", "within_1e6", "within_1e4_rel"] + (["f_within_1e6"] if with_f else [])
    # (a share that is a handful of voxels in every build says nothing: compared from 2 % of the sample on)
    shares = [k for k in shares if min(cpu1[k], cpu2[k]) >= 0.02]
    tails = ["p75", "p90", "p99", "failed"]
    if with_f and "same_iterations" in gpu:
        shares.append("same_iterations")
    for k in shares:
        assert gpu[k] >= min(cpu1[k], cpu2[k]), (what, k, gpu[k], cpu1[k], cpu2[k])
    for k in tails:
        # tail_factor (> 1 only where the caller says why): for the upper percentiles of a run that is stopped inside
        # the chaotic phase, where they are voxels on another trajectory altogether in every build
        f = tail_factor if k in ("p90", "p99") else 1.0
        assert gpu[k] <= f * max(cpu1[k], cpu2[k]), (what, k, gpu[k], cpu1[k], cpu2[k])
    assert gpu["median"] <= median_factor * max(cpu1["median"], cpu2["median"]), (what, gpu["median"], cpu1["median"], cpu2["median"])
    if with_f:
        # F where the posterior is the truth's (truth_stats). Its percentiles sit at the rounding level of a sum of
        # ~20 terms of size 1e2 (1e-12 relative: the kernels add the terms in another order and keep some of them
        # between evaluations): the same factor as the median above while the CPU's value is below 1e-8,
        # none above it.
        for k in ("f_median", "f_p75", "f_p90", "f_p99"):
            worst = max(cpu1[k], cpu2[k])
            allowed = max(median_factor * worst, 2e-12) if worst < 1e-8 else worst
            assert gpu[k] <= allowed, (what, k, gpu[k], cpu1[k], cpu2[k])


def load_c5_truth():
    """tests/golden/c5_truth_binary128.npz (make_c5_truth.py): the spatial bi-exponential fit on a 16 x 16 x 12 block"""
    import sys
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if gdir not in sys.path:
        sys.path.insert(0, gdir)
    import make_c5_truth as mt
    f = np.load(os.path.join(gdir, "c5_truth_binary128.npz"))
    h, sp, y = mt.problem(need_f=True)
    assert hashlib.sha256(y.tobytes()).hexdigest() == str(f["data_sha256"]), "the seeded series changed"
    return dict(mvn=f["mvn"], status=f["status"], free_energy=f["free_energy"], its=[int(i) for i in f["its"]],
                trace_means=f["trace_means"]), (h, sp, y)


def truth_trace_stats(holder, truth_means, r):
    """Relative error |m - truth| / max(|truth|, 1e-12) (max over the parameters) of the posterior means
    after a fixed number of iterations; truth_means [P][V]."""
    P = holder.cfg.n_params
    n = P + holder.n_noise_outputs
    off = n * (n + 1) // 2
    m = r["mvn"][off:off + P]
    ok = (r["status"] == 0) & np.isfinite(truth_means).all(axis=0)
    with np.errstate(invalid="ignore"):
        rel = np.max(np.abs(m[:, ok] - truth_means[:, ok]) / np.maximum(np.abs(truth_means[:, ok]), 1e-12), axis=0)
    rel = np.where(np.isnan(rel), np.inf, rel)
    return dict(median=float(np.median(rel)), p90=float(np.quantile(rel, 0.9)), p99=float(np.quantile(rel, 0.99)),
                within_1e4=float(np.mean(rel <= NORTH_STAR)), failed=float(np.mean(r["status"] != 0)))


def cubic_case_against_truth(name, run_gpu, base_tolerance=True):
    """tests/cases.py:cubic_cases()[name] on the device (run_gpu(holder, data)) and on both CPU builds, each measured
    against the binary128 ground truth (tests/golden/cubic_truth_binary128.npz): every voxel of the device's result within
    the base tolerances (1e-6 of max(|mean|, sd), 2e-4 sd_i sd_j, 1e-6 on F) of the truth; status and iterations identical."""
    import hashlib
    import os
    import cases
    import oracle
    h, y = cases.cubic_cases()[name]
    T = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cubic_truth_binary128.npz"))
    key = name.replace(" ", "_")
    assert str(T[key + "/data_sha256"]) == hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()
    truth = {k: T[key + "/" + k] for k in ("mvn", "free_energy", "status", "iterations")}
    cpu, fma, gpu = oracle.run(h, y), oracle.run_fma(h, y), run_gpu(h, y)
    assert np.all(truth["status"] == 0)
    ok = np.ones(h.cfg.n_voxels, dtype=bool)
    stats = {}
    for tag, r in (("cpu", cpu), ("fma", fma), ("gpu", gpu)):
        assert np.array_equal(r["status"], truth["status"]) and np.array_equal(r["iterations"], truth["iterations"]), tag
        e_mean, e_cov, _ = voxel_errors(h, truth, r, ok)
        f_err = (np.abs(truth["free_energy"] - r["free_energy"]) / np.maximum(1.0, np.abs(truth["free_energy"]))) if h.cfg.need_f else np.zeros(1)
        stats[tag] = dict(med=np.median(e_mean), p99=np.percentile(e_mean, 99), worst=e_mean.max(), cov=e_cov.max(), f=f_err.max())
    print(name, {t: {k: float("%.3g" % v) for k, v in st.items()} for t, st in stats.items()})
    # Measured (round 4): the kernels' worst voxel is 3e-9 - 2e-7 from the truth where the CPU builds' is 2e-7 - 4e-5 -
    # what the earlier rounds' raised bounds absorbed was the ORACLE's error (its LU inverse on a design whose columns
    # span four orders of magnitude; the kernels' symmetric sweep loses less). So the device is held to the BASE
    # tolerances against the truth, per voxel, and - as a guard on the comparison itself - the CPU builds to 1e-4.
    g = stats["gpu"]
    if base_tolerance:
        assert g["worst"] <= TOL_MEAN and g["cov"] <= TOL_COV and g["f"] <= TOL_F, (name, stats)
    else:
        # (the wave-per-voxel kernel inverts through LDS with one matrix entry per lane, another operation order: it is
        # as far from the truth as a CPU build - held to no worse than the worse of the two, in every statistic)
        # (the worst voxel of a few hundred is one draw of that noise: the two CPU builds' own worst voxels are up to 2 x
        # apart - FLOOR_FACTOR for it, as parity.strict allows where a floor is measured)
        worse = {k: max(stats["cpu"][k], stats["fma"][k]) for k in stats["cpu"]}
        slack = {"med": 1.0, "p99": 1.0, "worst": FLOOR_FACTOR, "cov": FLOOR_FACTOR, "f": FLOOR_FACTOR}
        for k in ("med", "p99", "worst", "cov", "f"):
            assert g[k] <= max(slack[k] * worse[k], {"med": 1e-9, "p99": TOL_MEAN, "worst": TOL_MEAN, "cov": TOL_COV, "f": TOL_F}[k]), (name, k, stats)
    for tag in ("cpu", "fma"):
        assert stats[tag]["worst"] <= NORTH_STAR, (name, tag, stats)
    return stats
