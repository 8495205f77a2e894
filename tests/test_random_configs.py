"""Randomised configurations: model, series length, noise model and pattern, masked timepoints,
prior types, convergence detector, free energy on/off, kernel mapping - HIP path against the
oracle with the strict per-voxel comparison (tests/parity.py). The fixed tests cover each feature
on its own; this sweep covers their combinations. Seeds are fixed: a failure reproduces."""
import numpy as np
import pytest

import hipengine
import oracle
import parity
from fabber_core_amd import hiplib, vbabi

AR = vbabi.NOISE_AR1


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.integers(40, 130))
    noise_kind = rng.choice(["white", "white", "white-pattern", "ar1", "ar2"])
    T = int(rng.integers(12, 60))
    if noise_kind == "ar2":
        T += T % 2
    t = np.arange(T, dtype=np.float64)
    model = rng.choice(["poly", "linear", "exp"])
    opts, names = {}, None
    if model == "poly":
        degree = int(rng.integers(0, 4))
        coef = rng.uniform(-2, 2, (degree + 1, V)) / np.array([10.0 ** n for n in range(degree + 1)])[:, None]
        y = sum(coef[n][None, :] * ((t[:, None] + 1) ** n) for n in range(degree + 1))
        opts.update(model=vbabi.MODEL_POLY, degree=degree)
        names = ["c%d" % n for n in range(degree + 1)]
    elif model == "linear":
        P = int(rng.integers(1, 8))      # 7 has no lane instantiation: wave kernel
        X = np.stack([np.cos(np.pi * (t + 0.5) * k / T) for k in range(P)], axis=1)
        y = X @ rng.normal(0, 3, (P, V))
        opts.update(model=vbabi.MODEL_LINEAR, design=X)
        names = ["Parameter_%d" % (k + 1) for k in range(P)]
    else:
        dt = 2.0 / T
        amp, rate = rng.uniform(0.5, 2.0, V), rng.uniform(0.5, 2.0, V)
        y = amp[None, :] * np.exp(-rate[None, :] * t[:, None] * dt)
        opts.update(model=vbabi.MODEL_EXP, num_exps=1, dt=dt)
        names = ["amp1", "r1"]
    y = y + rng.normal(0, 0.05 * max(1.0, float(np.abs(y).mean())), y.shape)
    kw = dict(max_iterations=int(rng.integers(2, 9)))
    # priors
    overrides, images = {}, {}
    for name in names:
        r = rng.random()
        if model == "exp":
            if r < 0.2:
                overrides[name] = dict(mean=float(rng.uniform(0.8, 1.5)), prec=float(10 ** rng.uniform(-3, 0)))
            continue
        if r < 0.15:
            overrides[name] = dict(type="A")
        elif r < 0.3:
            overrides[name] = dict(type="I", prec=float(10 ** rng.uniform(-4, 1)))
            images[name] = rng.normal(0, 1, V)
        elif r < 0.45:
            overrides[name] = dict(mean=float(rng.normal(0, 1)), prec=float(10 ** rng.uniform(-6, 0)))
    kw.update(param_overrides=overrides, image_priors=images)
    # noise
    variant = "auto"
    if noise_kind == "white-pattern":
        kw["noise_pattern"] = str(rng.choice(["12", "123", "1122"]))
    elif noise_kind == "ar1":
        kw["noise"] = AR
    elif noise_kind == "ar2":
        kw.update(noise=AR, num_echoes=2, ar_cross_terms=str(rng.choice(["none", "same", "dual"])))
    if noise_kind.startswith("white"):
        if rng.random() < 0.3:
            kw["masked_timepoints"] = tuple(sorted(set(int(x) for x in rng.integers(1, T + 1, 2))))
        if rng.random() < 0.2:
            kw["prior_noise_stddev"] = float(rng.uniform(0.05, 0.5))
        variant = str(rng.choice(["auto", "lane", "wave"]))
    conv = str(rng.choice(["maxits", "maxits", "pointzeroone", "freduce", "trialmode", "lm"]))
    if conv == "lm" and not noise_kind.startswith("white"):
        conv = "maxits"
    if conv != "maxits":
        kw.update(convergence=conv, min_fchange=0.01, max_iterations=int(rng.integers(6, 20)))
    elif rng.random() < 0.5:
        kw["need_f"] = True
    model_id = opts.pop("model")
    h = vbabi.build_config(model_id, V, T, **opts, **kw)
    h.spec = dict(model=str(model), opts=dict(opts), kw=dict(kw), names=names, noise_kind=str(noise_kind), conv=conv)
    desc = "seed %d: %s T=%d V=%d %s conv=%s variant=%s priors=%s" % (seed, model, T, V, noise_kind, conv, variant,
                                                                  {k: v.get("type", "N") for k, v in overrides.items()})
    return h, y.astype(np.float32), variant, desc


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(96))
def test_random_configuration(seed):
    h, y, variant, desc = random_case(seed)
    cpu = oracle.run(h, y)
    cpu2 = oracle.run_fma(h, y)
    hiplib.set_variant(variant)
    try:
        got = hipengine.run(h, y)
    finally:
        hiplib.set_variant("auto")
    # a voxel whose |dF| sits on a detector's threshold may stop one iteration apart
    uses_f = h.cfg.convergence != vbabi.CONV_MAXITS
    # The exponential model's rate starts at a Fabber-space mean of exactly 0, where the reference's
    # finite-difference step is its 1e-10 floor: the first Jacobian carries ~1e-6 of rounding noise
    # (a last-bit difference between the device's exp and the host's is enough), and after the few
    # iterations run here the free energy still shows it at the 1e-6 level (means stay within
    # their 1e-6 bound). Both CPU builds call the same libm, so the measured floor cannot see it.
    tol_f = 2e-5 if h.cfg.model == vbabi.MODEL_EXP else parity.TOL_F
    # (random problems include ill-conditioned ones: every configuration may use the measured CPU floor,
    # and the ones that did are listed at the end of the run)
    parity.strict(h, cpu, got, what=desc, cpu2=cpu2, allow_iter_mismatch=max(1, h.cfg.n_voxels // 50) if uses_f else 0,
                  tol_f=tol_f, allow_floor=True)


def test_random_cases_are_valid_for_the_oracle():
    """(CPU) every generated configuration runs in the oracle without failed voxels."""
    for seed in range(96):
        h, y, _, desc = random_case(seed)
        res = oracle.run(h, y)
        assert np.mean(res["status"] != 0) < 0.05, desc


# ---- spatial VB -----------------------------------------------------------------------------------
def random_spatial_case(seed):
    rng = np.random.default_rng(5000 + seed)
    shape = tuple(int(x) for x in rng.integers(5, 12, 3))
    mask = rng.random(shape) < rng.uniform(0.7, 1.0)
    mask[0, 0, 0] = True
    coords = vbabi.grid_coords(shape, mask)
    V = coords.shape[1]
    T = int(rng.integers(10, 40))
    t = np.arange(T, dtype=np.float64)
    field = 1.0 + 0.5 * np.sin(coords[0] / 2.0) * np.cos(coords[1] / 3.0) + 0.1 * coords[2]
    if rng.random() < 0.5:
        degree = int(rng.integers(0, 3))
        y = field[None, :] + sum(rng.normal(0, 0.3) * (t[:, None] / T) ** n for n in range(1, degree + 1))
        opts = dict(degree=degree)
        model, names = vbabi.MODEL_POLY, ["c%d" % n for n in range(degree + 1)]
    else:
        dt = 2.0 / T
        y = field[None, :] * np.exp(-rng.uniform(0.7, 1.5) * t[:, None] * dt)
        opts = dict(num_exps=1, dt=dt)
        model, names = vbabi.MODEL_EXP, ["amp1", "r1"]
    y = y + rng.normal(0, 0.05, (T, V))
    overrides, images = {}, {}
    spatial_types = ["M", "m", "P", "p"]
    overrides[names[0]] = dict(type=str(rng.choice(spatial_types)))
    for name in names[1:]:
        r = rng.random()
        if r < 0.3:
            overrides[name] = dict(type=str(rng.choice(spatial_types)))
        elif r < 0.45 and model == vbabi.MODEL_POLY:
            overrides[name] = dict(type="A")
        elif r < 0.6 and model == vbabi.MODEL_POLY:
            overrides[name] = dict(type="I", prec=float(10 ** rng.uniform(-3, 0)))
            images[name] = rng.normal(0, 0.3, V)
    h = vbabi.build_config(model, V, T, max_iterations=int(rng.integers(3, 8)), need_f=bool(rng.random() < 0.4),
                           param_overrides=overrides, image_priors=images, **opts)
    sp = vbabi.SpatialHolder(coords, spatial_dims=int(rng.choice([3, 3, 2, 1])),
                             spatial_speed=float(rng.choice([-1.0, -1.0, 1.5, 3.0])), q1=float(rng.choice([10.0, 5.0])),
                             q2=float(rng.choice([1.0, 2.0])), update_first_iter=bool(rng.random() < 0.3))
    desc = "spatial seed %d: shape %s V=%d T=%d model %d priors %s dims %d speed %g first %d F %d" % (
        seed, shape, V, T, model, {k: v["type"] for k, v in overrides.items()}, sp.sp.spatial_dims, sp.sp.spatial_speed,
        sp.sp.update_first_iter, h.cfg.need_f)
    return h, sp, y.astype(np.float32), desc


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(32))
def test_random_spatial_configuration(seed):
    h, sp, y, desc = random_spatial_case(seed)
    cpu = oracle.run_spatial(h, sp, y)
    cpu2 = oracle.run_spatial_fma(h, sp, y)
    got = hiplib.run_spatial_host(h, sp, y)
    for r in (cpu, cpu2):
        r.setdefault("f_history_len", np.zeros(h.cfg.n_voxels, dtype=np.int32))
    tol_f = 2e-5 if h.cfg.model == vbabi.MODEL_EXP else parity.TOL_F
    others = [cpu2]
    if h.cfg.model == vbabi.MODEL_EXP:
        # (a few iterations from rates of exactly 0: the step of the first central differences is 1e-10, the last bit
        # of exp shows 1e6-fold in J, and the device's exp is a 1-ulp exp where glibc's is 0.51: one more CPU build)
        others.append(oracle.run_spatial_exp1ulp(h, sp, y))
        others[-1].setdefault("f_history_len", np.zeros(h.cfg.n_voxels, dtype=np.int32))
    parity.strict(h, cpu, got, what=desc, cpu2=others, tol_f=tol_f, allow_floor=True)


def test_random_spatial_cases_are_valid_for_the_oracle():
    for seed in range(32):
        h, sp, y, desc = random_spatial_case(seed)
        res = oracle.run_spatial(h, sp, y)
        assert np.mean(res["status"] != 0) < 0.2, desc    # (failed voxels are part of what is compared)


# ---- the same configurations through the reference's API ------------------------------------------
def capi_options(h, tmp_path):
    """The fabber options that describe configuration h (built by random_case) + extra data sets."""
    sp, kw = h.spec, h.spec["kw"]
    o = {"method": "vb", "max-iterations": kw["max_iterations"], "save-mvn": True, "save-mean": True, "allow-bad-voxels": True}
    extra = {}
    if sp["model"] == "poly":
        o.update(model="poly", degree=sp["opts"]["degree"])
    elif sp["model"] == "linear":
        design = tmp_path / "design.mat"
        np.savetxt(design, sp["opts"]["design"], fmt="%.17g")
        o.update(model="linear", basis=str(design))
    else:
        o.update(model="exp", **{"num-exps": 1, "dt": repr(sp["opts"]["dt"])})
    for n, (name, ov) in enumerate(kw["param_overrides"].items(), start=1):
        o["PSP_byname%d" % n] = name
        if "type" in ov:
            o["PSP_byname%d_type" % n] = ov["type"]
        if "mean" in ov:
            o["PSP_byname%d_mean" % n] = repr(ov["mean"])
        if "prec" in ov:
            o["PSP_byname%d_prec" % n] = repr(ov["prec"])
        if ov.get("type") == "I":
            o["PSP_byname%d_image" % n] = "img_" + name
            extra["img_" + name] = kw["image_priors"][name]
    kind = sp["noise_kind"]
    o["noise"] = "white" if kind.startswith("white") else "ar"
    if "noise_pattern" in kw:
        o["noise-pattern"] = kw["noise_pattern"]
    if kind == "ar2":
        o.update({"num-echoes": 2, "ar1-cross-terms": kw["ar_cross_terms"]})
    for i, mt in enumerate(kw.get("masked_timepoints", ()), start=1):
        o["mt%d" % i] = mt
    if "prior_noise_stddev" in kw:
        o["prior-noise-stddev"] = repr(kw["prior_noise_stddev"])
    if sp["conv"] != "maxits":
        o.update({"convergence": sp["conv"], "min-fchange": kw["min_fchange"]})
        if sp["conv"] == "lm":
            o["max-fchange"] = kw["min_fchange"]
    if kw.get("need_f"):
        o["save-free-energy"] = True
    return o, extra


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 96, 3))
def test_random_configuration_through_the_c_abi(seed, tmp_path):
    """Two independent routes to the engine's problem block: the C++ host layer resolving the
    reference's options (FwdModel::GetParameters, noise model / detector options) and
    vbabi.build_config. Same data, same results."""
    from fabber_core_amd import fabber
    h, y, _, desc = random_case(seed)
    V = h.cfg.n_voxels
    options, extra = capi_options(h, tmp_path)
    vol = y.T.reshape(V, 1, 1, -1)
    extra = {k: np.asarray(v, dtype=np.float32).reshape(V, 1, 1) for k, v in extra.items()}
    # (image priors travel as float32 through the C ABI: give the direct route the same values)
    for name, img in h.spec["kw"]["image_priors"].items():
        img[:] = img.astype(np.float32)
    out = fabber.run(vol, options, extra_data=extra)
    eng = hipengine.run(h, y)
    got = out["finalMVN"].reshape(V, -1).T
    scale = np.maximum(np.abs(eng["mvn"]), 1e-30)
    ok = eng["status"] == 0
    err = (np.abs(got - eng["mvn"]) / scale)[:, ok]
    assert err.max() < 5e-6, (desc, options, float(err.max()), np.unravel_index(err.argmax(), err.shape))
