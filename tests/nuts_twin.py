"""Test infrastructure: a NumPy restatement of ONE launch of the sampler kernel (`dyn_nuts_advance`; per-chain adaptation, or pooled windows with K['pooled']) --
everything a chain does between two potential evaluations, as csrc/nuts_device.hpp `advance_chain` and csrc/nuts_kernel.hip
`nuts_advance_lanes` do it: second momentum half step and energy error, multinomial choice inside the subtree, checkpointed
U-turn test, biased progressive merge, dual averaging (t0 = 10, kappa = 0.75, gamma = 0.05), Welford window with the
regularised dense matrix and chol(inv(.)) at its end, the recorded draw, fresh momentum, direction, first half of the next
leapfrog.  The algorithm is numpyro's iterative NUTS (reference call site src/dynode/infer/inference.py:149-163; numpyro 0.15
itself is third party and not in the tree); the counter-based Philox4x32-10 stream (Salmon et al. 2011) is restated here in
Python integers.  Plain loops over chains, float64 throughout: it checks the kernels launch by launch from THEIR state
(tests/test_gpu_infer.py), so rounding differences cannot accumulate.  Never imported by the package."""
import math

import numpy as np

M32 = 0xFFFFFFFF
POOL_SCALE = 1073741824.0     # 2^30 fixed point of the pooled window sums (csrc/nuts_device.hpp)


def philox4x32_10(c, k):
    c0, c1, c2, c3 = c
    k0, k1 = k
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


class Stream:
    def __init__(self, seed, ctr, chain):
        self.key, self.ctr, self.chain = (seed & M32, (seed >> 32) & M32), int(ctr), int(chain)

    def uniform(self):
        o = philox4x32_10((self.ctr & M32, (self.ctr >> 32) & M32, self.chain, 0), self.key)
        self.ctr += 1
        return ((((o[0] << 20) ^ (o[1] >> 12)) + 0.5) * (1.0 / 4503599627370496.0))

    def normal(self):
        u1, u2 = self.uniform(), self.uniform()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586476925286766559 * u2)


def logaddexp(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    return max(a, b) + math.log1p(math.exp(-abs(a - b)))


def is_turning(imm, rl, rr, rsum):
    rs = rsum - 0.5 * (rl + rr)
    return float((imm @ rl) @ rs) <= 0.0 or float((imm @ rr) @ rs) <= 0.0


def mass_sqrt(imm):
    inv = np.linalg.inv(imm)
    return np.linalg.cholesky(0.5 * (inv + inv.T))


def advance(S, K, u_new, g_new):
    """One launch for every chain.  S: dict of numpy arrays laid out as `dyn_nuts_state` (include/dynode_hip.h), changed in
    place; K: dict of the struct's scalars (seed, num_warmup, num_samples, max_depth, target_accept, max_delta_energy,
    windows = [(start, end), ...]); (u_new [C], g_new [C, D]): the potential at S['z_eval']."""
    C, D = S["z"].shape
    Dm, total = K["max_depth"], K["num_warmup"] + K["num_samples"]
    with np.errstate(invalid="ignore"):      # (inf - inf behind a wall of non-finite potentials: NaN compares false, as in the kernel)
        for c in range(C):
            if S["it"][c] < total:
                _advance_chain(S, K, c, D, Dm, total, u_new, g_new)
    if K.get("pooled"):
        S["pool_ro"][...] = S["pool"]      # (readers of the next launch see the pool as it stands after this one)


def _advance_chain(S, K, c, D, Dm, total, u_new, g_new):
    """Chain c, if it has transitions left."""
    it = int(S["it"][c])
    imm = S["imm"][c]
    rng = Stream(K["seed"], S["rng_ctr"][c], c)
    eps_signed = S["eps"][c] * S["sgn"][c]
    un, gn, zn = float(u_new[c]), g_new[c].copy(), S["z_eval"][c].copy()
    bad = not (math.isfinite(un) and np.isfinite(gn).all())
    if bad:
        gn = np.zeros(D)
    rn = S["r_half"][c] - 0.5 * eps_signed * gn
    de = (math.inf if bad else un) + 0.5 * float(rn @ (imm @ rn)) - S["e0"][c]
    if math.isnan(de):
        de = math.inf
    lw, div = -de, de > K["max_delta_energy"]
    acc = math.exp(min(-de, 0.0))
    # grow the subtree by this leaf
    new_w = logaddexp(S["s_weight"][c], lw)
    if rng.uniform() < math.exp(lw - new_w):
        S["s_zp"][c], S["s_gp"][c], S["s_up"][c] = zn, gn, (math.inf if bad else un)
    s_weight = new_w
    S["s_rsum"][c] += rn
    s_div = bool(S["s_div"][c]) or div
    s_acc, s_n = S["s_acc"][c] + acc, int(S["s_n"][c]) + 1
    S["zc"][c], S["rc"][c], S["gc"][c] = zn, rn, gn
    # checkpointed U-turn test
    leaf = int(S["leaf"][c])
    idx_max = bin(leaf >> 1).count("1")
    trailing = 0
    while (leaf >> trailing) & 1:
        trailing += 1
    idx_min = idx_max - trailing + 1
    s_turn = bool(S["s_turn"][c])
    if leaf & 1 == 0:
        S["r_ck"][c, idx_max], S["rs_ck"][c, idx_max] = rn, S["s_rsum"][c]
    else:
        for k in range(idx_max, idx_min - 1, -1):
            s_turn = s_turn or is_turning(imm, S["r_ck"][c, k], rn, S["s_rsum"][c] - S["rs_ck"][c, k] + S["r_ck"][c, k])
    leaf += 1
    # subtree complete -> merge into the trajectory
    depth, right = int(S["depth"][c]), bool(S["right"][c])
    sub_done = s_turn or s_div or leaf >= (1 << depth)
    stop = False
    if sub_done:
        if (not s_turn and not s_div) and rng.uniform() < math.exp(min(s_weight - S["weight"][c], 0.0)):
            S["zp"][c], S["gp"][c], S["up"][c] = S["s_zp"][c], S["s_gp"][c], S["s_up"][c]
        e = ("zr", "rr", "gr") if right else ("zl", "rl", "gl")
        S[e[0]][c], S[e[1]][c], S[e[2]][c] = S["zc"][c], S["rc"][c], S["gc"][c]
        S["weight"][c] = logaddexp(S["weight"][c], s_weight)
        S["r_sum"][c] += S["s_rsum"][c]
        S["sum_acc"][c] += s_acc
        S["n_prop"][c] += s_n
        depth += 1
        stop = s_turn or s_div or is_turning(imm, S["rl"][c], S["rr"][c], S["r_sum"][c]) or depth >= Dm
    eps = float(S["eps"][c])
    if stop:
        warm = it < K["num_warmup"]
        n_prop = int(S["n_prop"][c])
        a_prob = S["sum_acc"][c] / float(n_prop if n_prop > 0 else 1)
        S["z"][c], S["g"][c], S["u"][c] = S["zp"][c], S["gp"][c], S["up"][c]
        z = S["z"][c]
        if warm:
            t1 = S["da_t"][c] + 1.0
            w = 1.0 / (t1 + 10.0)
            gbar = (1.0 - w) * S["da_gbar"][c] + w * (K["target_accept"] - a_prob)
            x = S["da_mu"][c] - math.sqrt(t1) / 0.05 * gbar
            wx = t1 ** -0.75
            xbar = (1.0 - wx) * S["da_xbar"][c] + wx * x
            S["da_t"][c], S["da_gbar"][c], S["da_xbar"][c] = t1, gbar, xbar
            eps = math.exp(x)
            S["eps_avg"][c] = math.exp(xbar)
            if K.get("pooled") and S["pend"][c] > 0:
                # the pooled statistics of every chain that had closed this window when the previous launch ended
                pw = S["pool_ro"][int(S["pend"][c]) - 1].astype(np.float64)
                N = pw[0]
                nn = max(N, 2.0)
                with np.errstate(divide="ignore", invalid="ignore"):
                    mu = pw[1:1 + D] / POOL_SCALE / N
                    cov = (pw[1 + D:].reshape(D, D) / POOL_SCALE - N * np.outer(mu, mu)) / (nn - 1.0)
                cand = (nn / (nn + 5.0)) * cov + 1e-3 * (5.0 / (nn + 5.0)) * np.eye(D)
                try:
                    chol = mass_sqrt(cand)
                except np.linalg.LinAlgError:
                    chol = np.full((D, D), np.nan)
                if N >= 2.0 and np.isfinite(cand).all() and np.isfinite(chol).all() and (np.diag(chol) > 0).all() and (np.diag(cand) > 0).all():
                    S["imm"][c], S["mm_sqrt"][c] = cand, chol
                    imm = S["imm"][c]
                    eps = float(S["eps_avg"][c])
                    S["da_mu"][c] = math.log(10.0 * eps)
                    S["da_t"][c] = S["da_gbar"][c] = S["da_xbar"][c] = 0.0
                S["pend"][c] = 0
            wi = int(S["wi"][c])
            if wi < len(K["windows"]) and K["windows"][wi][0] <= it < K["windows"][wi][1]:
                n1 = S["wf_n"][c] + 1.0
                d0 = z - S["wf_mean"][c]
                S["wf_mean"][c] += d0 / n1
                S["wf_m2"][c] += np.outer(d0, z - S["wf_mean"][c])
                S["wf_n"][c] = n1
                if it + 1 == K["windows"][wi][1] and K.get("pooled"):
                    # this chain's window into the pool (fixed point: the sums do not depend on the order); applied at its NEXT transition end
                    mean, fix = S["wf_mean"][c], lambda v: np.rint(np.asarray(v) * POOL_SCALE).astype(np.int64)
                    S["pool"][wi, 0] += int(n1)
                    S["pool"][wi, 1:1 + D] += fix(n1 * mean)
                    S["pool"][wi, 1 + D:] += fix(S["wf_m2"][c] + n1 * np.outer(mean, mean)).ravel()
                    S["pend"][c] = wi + 1
                elif it + 1 == K["windows"][wi][1]:
                    nn = max(n1, 2.0)
                    S["imm"][c] = (nn / (nn + 5.0)) * S["wf_m2"][c] / (nn - 1.0) + 1e-3 * (5.0 / (nn + 5.0)) * np.eye(D)
                    S["mm_sqrt"][c] = mass_sqrt(S["imm"][c])
                    imm = S["imm"][c]
                    eps = float(S["eps_avg"][c])
                    S["da_mu"][c] = math.log(10.0 * eps)
                    S["da_t"][c] = S["da_gbar"][c] = S["da_xbar"][c] = 0.0
                if it + 1 == K["windows"][wi][1]:
                    S["wf_n"][c], S["wf_mean"][c], S["wf_m2"][c] = 0.0, 0.0, 0.0
                    S["wi"][c] = wi + 1
            if it + 1 == K["num_warmup"]:
                eps = float(S["eps_avg"][c])
                if K.get("pooled"):
                    # final step size: geometric mean over the chains that had finished warm-up when the previous launch ended, and this one
                    nw, le = len(K["windows"]), math.log(S["eps_avg"][c])
                    pr = S["pool_ro"][nw]
                    S["pool"][nw, 0] += 1
                    S["pool"][nw, 1] += int(np.rint(le * POOL_SCALE))
                    eps = math.exp((float(pr[1]) / POOL_SCALE + le) / (float(pr[0]) + 1.0))
        else:
            j = it - K["num_warmup"]
            S["out_z"][c, j], S["out_acc"][c, j], S["out_n"][c, j], S["out_div"][c, j] = z, a_prob, n_prop, int(s_div)
        S["eps"][c] = eps
        it += 1
        S["it"][c] = it
        r0 = S["mm_sqrt"][c] @ np.array([rng.normal() for _ in range(D)])
        S["e0"][c] = S["u"][c] + 0.5 * float(r0 @ (imm @ r0))
        for k in ("zl", "zr", "zp"):
            S[k][c] = z
        for k in ("rl", "rr", "r_sum"):
            S[k][c] = r0
        for k in ("gl", "gr", "gp"):
            S[k][c] = S["g"][c]
        S["up"][c] = S["u"][c]
        S["weight"][c], S["sum_acc"][c], S["n_prop"][c] = 0.0, 0.0, 0
        depth = 0
    S["depth"][c] = depth
    go_right = right
    if sub_done:
        go_right = rng.uniform() < 0.5
        S["right"][c], S["sgn"][c] = int(go_right), (1.0 if go_right else -1.0)
        e = ("zr", "rr", "gr") if go_right else ("zl", "rl", "gl")
        S["zc"][c], S["rc"][c], S["gc"][c] = S[e[0]][c], S[e[1]][c], S[e[2]][c]
        S["s_zp"][c], S["s_gp"][c], S["s_rsum"][c], S["s_up"][c] = S["zp"][c], S["gp"][c], 0.0, S["up"][c]
        s_weight, s_turn, s_div, leaf = -math.inf, False, False, 0
        S["s_acc"][c], S["s_n"][c] = 0.0, 0
        S["r_ck"][c], S["rs_ck"][c] = 0.0, 0.0
    else:
        S["s_acc"][c], S["s_n"][c] = s_acc, s_n
    S["s_weight"][c], S["s_turn"][c], S["s_div"][c], S["leaf"][c] = s_weight, int(s_turn), int(s_div), leaf
    es = eps * (1.0 if go_right else -1.0)
    rh = S["rc"][c] - 0.5 * es * S["gc"][c]
    S["r_half"][c] = rh
    S["z_eval"][c] = S["z"][c] if it >= total else S["zc"][c] + es * (imm @ rh)
    S["rng_ctr"][c] = rng.ctr
