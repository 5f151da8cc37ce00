"""Generate tests/golden/ground_truth.npz: fp64 trajectories from scipy DOP853 (rtol=atol=1e-12).

The reference's JAX stack is not importable in the build container, so no
diffrax output can be captured; these vectors pin the oracle and the HIP path against an
INDEPENDENT integrator (scipy 1.15.3 DOP853) driving an independent vectorised RHS
(tests/helpers.py:rhs_numpy).  Case inputs are the literals of the reference examples
(cited per case) plus three draws of the cfg-2 / cfg-3 / cfg-5 synthetic generators.

Run:  python tests/golden/make_golden.py      (a few seconds; output ~1 MB)
"""

from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
from dynode_amd import ModelDesc, synthetic  # noqa: E402


def cases():
    out = []
    # examples/sir.py:34,43,90-91 (1 bin, y0=(0.9,0.1,0), r0=2, T_inf=7)
    wl = synthetic.sir_literal()
    out.append(("sir_literal", wl.model, wl.y0, wl.params[0], wl.contact, 150.0))
    # tests/test_sir_dynamics/test_sir.py:9-16 initial condition (0.99, 0.01, 0), 300 days
    out.append(("sir_final_size", wl.model, np.array([0.99, 0.01, 0.0]), wl.params[0], wl.contact, 300.0))
    # tests/test_simulation/test_odes.py:17-42: beta*s*i without /N, y0=(99,1,0)
    out.append(("sir_unnormalised", ModelDesc(n_age=1, normalize=False), np.array([99.0, 1.0, 0.0]),
                np.array([2.0 / 7, 1.0 / 7]), np.array([[1.0]]), 100.0))
    # examples/sir_age_stratified.py:46-66,70,81-85
    wl = synthetic.sir_two_age_literal(t1=100.0)
    out.append(("sir_two_age", wl.model, wl.y0, wl.params[0], wl.contact, 100.0))
    # BASELINE.json cfg 1 as worded: the same 2-age-group SIR, 1 parameter set, 365 days
    out.append(("sir_two_age_365", wl.model, wl.y0, wl.params[0], wl.contact, 365.0))
    # examples/seirs.py:32-33,37,47,58 (r0=2, T_inf=7, latent 3, waning 60)
    seirs = ModelDesc(n_age=1, has_e=True, has_wane=True)
    out.append(("seirs", seirs, np.array([0.99, 0.0, 0.01, 0.0]),
                np.array([2.0 / 7, 1.0 / 7, 1.0 / 3, 1.0 / 60]), np.array([[1.0]]), 365.0))
    # examples/seirs_seasonal_forcing.py:61-63 (amp 0.2, phase 0, period 365)
    seas = ModelDesc(n_age=1, has_e=True, has_wane=True, seasonal=True)
    out.append(("seirs_seasonal", seas, np.array([0.99, 0.0, 0.01, 0.0]),
                np.array([2.0 / 7, 1.0 / 7, 1.0 / 3, 1.0 / 60, 0.2, 0.0, 365.0]), np.array([[1.0]]), 365.0))
    # examples/seirs_multi_strain_age_stratified.py:46-49,95,146-172 (2 age x 3 strain literal)
    ms = ModelDesc(n_age=2, n_strain=3, has_e=True, has_wane=True, has_c=True)
    r0 = np.array([2.0, 2.5, 1.8]); ti = np.array([7.0, 6.0, 8.0]); tl = np.array([3.0, 2.5, 4.0])
    tw = np.array([60.0, 80.0, 50.0])
    demo = np.array([0.75, 0.25])
    y0 = np.zeros(ms.state_dim)
    y0[:2] = 1000 * 0.99 * demo
    y0[2 + 6:2 + 12] = (1000 * 0.01 * demo[:, None] * (r0 / r0.sum())[None, :]).ravel()
    out.append(("multi_strain_2x3", ms, y0, np.concatenate([r0 / ti, 1 / ti, 1 / tl, 1 / tw]),
                np.array([[0.7, 0.3], [0.3, 0.7]]), 365.0))
    # synthetic BASELINE configs, three draws each
    for tag, wl in (("cfg2", synthetic.sir_age_stratified(3, seed=0)),
                    ("cfg3", synthetic.seirs_multi_strain(3, seed=1)),
                    ("cfg5", synthetic.seirs_multi_strain(3, seed=5, seasonal=True)),
                    ("cfg3w8", synthetic.seirs_multi_strain(2, seed=1, W=8))):
        for b in range(wl.B):
            y0 = wl.y0[b] if wl.y0.ndim == 2 else wl.y0
            out.append((f"{tag}_{b}", wl.model, y0, wl.params[b], wl.contact, 365.0))
    return out


SEIP_FIELDS = ("n_age", "n_strain", "has_e", "has_wane", "has_c", "n_wane", "normalize", "seasonal", "has_intro", "n_vax_tiers",
               "n_vax_knots", "family", "seasonal_vax")


def seip_cases():
    """Two members of the SEIP family (ode_model.md; no code in the reference): the generator's draws, two trajectories each."""
    out = []
    for tag, shape in (("seip_4x2x3x4", dict(A=4, L=2, K1=3, M1=4, n_knots=2, seasonal_vax=True)),            # D = 480
                       ("seip_3x3x2x3_intro", dict(A=3, L=3, K1=2, M1=3, n_knots=1, seasonal=True, intro=True))):   # D = 432
        wl = synthetic.seip(B=2, seed=41, t1=150.0, **shape)
        out.append((tag, wl))
    return out


def main_seip():
    """tests/golden/ground_truth_seip.npz: SciPy DOP853 (rtol 1e-11) on the independent NumPy statement of the SEIP equations
    (tests/helpers.py:rhs_seip_numpy) -- the SEIP oracle is pinned by these arrays, not by a probe run."""
    blob, names = {}, []
    for name, wl in seip_cases():
        m = wl.model
        ts = np.linspace(0.0, 150.0, 16)
        blob[f"{name}/model"] = np.array([int(getattr(m, f)) for f in SEIP_FIELDS], dtype=np.int64)
        blob[f"{name}/intro_age_mask"] = np.array([int(v) for v in m.intro_age_mask], dtype=np.int64)
        blob[f"{name}/y0"] = np.asarray(wl.y0, float)
        blob[f"{name}/params"] = np.asarray(wl.params, float)
        blob[f"{name}/contact"] = np.asarray(wl.contact, float)
        blob[f"{name}/ts"] = ts
        blob[f"{name}/ys"] = np.stack([H.ground_truth_seip(m, wl.y0[b], wl.params[b], wl.contact, 150.0, ts) for b in range(wl.B)])
        names.append(name)
        print(f"{name:22s} D={m.state_dim:4d} n_save={ts.size} |y|max={np.abs(blob[f'{name}/ys']).max():.4g}")
    blob["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ground_truth_seip.npz"), **blob)


def main():
    blob = {}
    names = []
    for name, m, y0, p, C, t1 in cases():
        ts = np.linspace(0.0, t1, int(t1 // 5) + 1)  # every ~5 days
        ys = H.ground_truth(m, y0, p, C, t1, ts)
        names.append(name)
        blob[f"{name}/model"] = np.array([m.n_age, m.n_strain, m.has_e, m.has_wane, m.has_c, m.n_wane,
                                          m.normalize, m.seasonal], dtype=np.int32)
        blob[f"{name}/y0"] = np.asarray(y0, float)
        blob[f"{name}/params"] = np.asarray(p, float)
        blob[f"{name}/contact"] = np.asarray(C, float)
        blob[f"{name}/t1"] = np.array(t1)
        blob[f"{name}/ts"] = ts
        blob[f"{name}/ys"] = ys
        print(f"{name:22s} D={m.state_dim:4d} n_save={ts.size} |y|max={np.abs(ys).max():.4g}")
    blob["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ground_truth.npz"), **blob)


if __name__ == "__main__":
    main()
    main_seip()
