import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cases import golden_problem
from hyperion_amd.run import run_problem
from test_oracle_golden import killed_counts
tau = sys.argv[2] if len(sys.argv) > 2 else "1000000"
prob, z = golden_problem("pinte_seds.tau=%s.npz" % tau)
gold = z["golden/seds"]; w = prob.density * prob.volumes
for rep in range(int(sys.argv[1])):
    K = 32
    S, e_last, se_last, kint = [], [], [], []
    for k in range(K):
        prob.config.seed = -(900 + k)
        r = run_problem(prob)
        S.append(r.peeled[0]["seds"]); e_last.append((r.iterations[-1].specific_energy * w).sum()); se_last.append(r.iterations[-1].specific_energy[0])
        kint.append([it.killed_int for it in r.iterations])
    S = np.array(S)
    I, sg = S.mean(axis=0)[0, 0, :, 0, :], S.std(axis=0, ddof=1)[0, 0, :, 0, :]
    g = gold[0, 0, :, 0, :]
    sel = (sg > 0) & (I > 1e-3 * I.max())
    zs = (g - I)[sel] / sg[sel]
    well = sg[sel] < 0.3 * I[sel]
    se = np.array(se_last); m, sd = se.mean(axis=0), se.std(axis=0, ddof=1)
    gold_se = z["golden/specific_energy_last"][0]
    okc = (m > 0) & (gold_se > 0) & (sd < 0.3 * m) & (w[0] > 0)
    le = np.log10(np.array(e_last)); e_gold = (z["golden/specific_energy_last"] * w).sum()
    zt = (np.log10(e_gold) - le.mean()) / (le.std(ddof=1) * np.sqrt(1 + 1 / K))
    n = min(len(k) for k in kint); c = np.array([k[:n] for k in kint], dtype=float)
    gk = np.array([x[1] for x in killed_counts("test_pinte_seds.tau=%s" % tau)["iterations"]][:n], dtype=float)
    mm, var = c.mean(axis=0), np.maximum(c.var(axis=0, ddof=1), c.mean(axis=0))
    never = c.max(axis=0) == 0
    zk = (gk - mm)[~never] / np.sqrt(var[~never] * (1 + 1 / K) + 1)
    tot_sd = np.sqrt(max(c.sum(axis=1).var(ddof=1), mm.sum()) * (1 + 1 / K) + 1)
    print("well %d maxz %.2f mz2 %.2f meanz %.2f | nonwell min %.2f frac>6 %.2f | okc %d medlog %.3f | ztot %.2f | killed maxz %.2f tot %.2f | n_it min %d" % (
        well.sum(), np.abs(zs[well]).max(), (zs[well] ** 2).mean(), zs[well].mean(), zs[~well].min() if (~well).any() else 0, (zs[~well] > 6).mean() if (~well).any() else 0,
        okc.sum(), np.median(np.log10(gold_se[okc] / m[okc])), zt, np.abs(zk).max(), (gk.sum() - mm.sum()) / tot_sd, n), flush=True)
