"""Oracle (test infrastructure, PARITY UNPINNED like the rest of oracle/): the MPC-step NLP of
`VehicleFollower.setup_controller` (reference `confrez/control/vehicle_follower.py:146-368`) in the reference's OWN
decision variables -- x, y, psi, v, delta, a, w, the OBCA duals l, m of every static obstacle and lambda_ij, lambda_ji,
s of every neighbour -- as plain functions for a general-purpose NLP solver (scipy).  Nothing of the engine's
formulation (certificate elimination, working sets, slacks) is used here; it exists so that the oracle's answers can be
checked against an independent solver on the formulation IPOPT sees.

Variable vector, stage-major: per stage [x y psi v delta a w | l (4 n_obs) | m (4 n_obs) | per neighbour lam_ij (4),
lam_ji (4), s (2)]."""
import numpy as np

from .dynamics import bicycle_rk4
from .mpc_nlp import MpcSpec, rot


class ReferenceNlp:
    def __init__(self, spec: MpcSpec, x0, ref, nbr):
        self.spec, self.x0, self.ref, self.nbr = spec, np.asarray(x0, float), np.asarray(ref, float), np.asarray(nbr, float)
        self.no, self.nn, self.N = spec.n_obs, spec.n_nbr, spec.N
        self.ns = 7 + 8 * self.no + 10 * self.nn
        self.n = self.N * self.ns

    # ---- packing -----------------------------------------------------------------------------------
    def pack(self, sol):
        X = np.zeros((self.N, self.ns))
        for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w")):
            X[:, c] = sol[k]
        X[:, 7 : 7 + 4 * self.no] = sol["l"]
        X[:, 7 + 4 * self.no : 7 + 8 * self.no] = sol["m"]
        for o in range(self.nn):
            b = 7 + 8 * self.no + 10 * o
            X[:, b : b + 4], X[:, b + 4 : b + 8], X[:, b + 8 : b + 10] = sol["lam_ij"][o], sol["lam_ji"][o], sol["s"][o]
        return X.ravel()

    def unpack(self, X):
        X = np.asarray(X).reshape(self.N, self.ns)
        sol = {k: X[:, c].copy() for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}
        sol["l"], sol["m"] = X[:, 7 : 7 + 4 * self.no].copy(), X[:, 7 + 4 * self.no : 7 + 8 * self.no].copy()
        b = 7 + 8 * self.no
        sol["lam_ij"] = np.stack([X[:, b + 10 * o : b + 10 * o + 4] for o in range(self.nn)]) if self.nn else np.zeros((0, self.N, 4))
        sol["lam_ji"] = np.stack([X[:, b + 10 * o + 4 : b + 10 * o + 8] for o in range(self.nn)]) if self.nn else np.zeros((0, self.N, 4))
        sol["s"] = np.stack([X[:, b + 10 * o + 8 : b + 10 * o + 10] for o in range(self.nn)]) if self.nn else np.zeros((0, self.N, 2))
        return sol

    def bounds(self):
        lo, hi = np.full((self.N, self.ns), -np.inf), np.full((self.N, self.ns), np.inf)
        bd = self.spec.bounds
        for c, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):  # :204-240
            lo[:, c], hi[:, c] = bd[2 * j], bd[2 * j + 1]
        lo[:, 7 : 7 + 8 * self.no] = 0.0  # l, m >= 0 :201-202
        for o in range(self.nn):
            b = 7 + 8 * self.no + 10 * o
            lo[:, b : b + 8] = 0.0  # lambda_ij, lambda_ji >= 0 :315-316
        return lo.ravel(), hi.ravel()

    # ---- objective and constraints --------------------------------------------------------------------
    def cost(self, X):  # :263-272
        s, wt, r = self.unpack(X), self.spec.weights, self.ref
        return float(np.sum(wt[0] * (s["x"] - r[0]) ** 2 + wt[1] * (s["y"] - r[1]) ** 2 + wt[2] * (s["psi"] - r[2]) ** 2
                            + wt[3] * s["a"] ** 2 + wt[4] * s["v"] ** 2 * s["w"] ** 2 + wt[5] * s["delta"] ** 2))

    def constraints(self, X):
        """(eq, ineq): eq == 0, ineq >= 0, rows in the order of the reference's subject_to calls."""
        sp_, s = self.spec, self.unpack(X)
        G, g = sp_.G, sp_.g
        z = np.stack([s["x"], s["y"], s["psi"], s["v"], s["delta"]], -1)
        eq, ineq = [z[0] - self.x0], []  # :194-199
        for i in range(self.N):
            if i < self.N - 1:  # :243-260
                eq.append(z[i + 1] - bicycle_rk4(z[i], np.array([s["a"][i], s["w"][i]]), sp_.dt, sp_.wb, sp_.rk_substeps))
            t, R = z[i, :2], rot(z[i, 2])
            for j in range(self.no):  # :280-290
                A, b = sp_.A_obs[j], sp_.b_obs[j]
                lj, mj = s["l"][i, 4 * j : 4 * j + 4], s["m"][i, 4 * j : 4 * j + 4]
                ineq.append([np.dot(-g, mj) + np.dot(A @ t - b, lj) - sp_.dmin])
                eq.append(G.T @ mj + R.T @ A.T @ lj)
                eq.append([np.dot(A.T @ lj, A.T @ lj) - 1.0])
            for o in range(self.nn):  # :322-352
                lik, mik, sik = s["lam_ij"][o][i], s["lam_ji"][o][i], s["s"][o][i]
                tR = rot(-z[i, 2]); tA = G @ tR; tb = G @ tR @ t + g
                ot = self.nbr[o, :2, i]; oR = rot(-self.nbr[o, 2, i]); oA = G @ oR; ob = G @ oR @ ot + g
                ineq.append([-np.dot(tb, lik) - np.dot(ob, mik) - sp_.dmin])
                eq.append(tA.T @ lik + sik)
                eq.append(oA.T @ mik - sik)
                ineq.append([1.0 - np.dot(sik, sik)])
        return np.concatenate([np.ravel(e) for e in eq]), (np.concatenate([np.ravel(e) for e in ineq]) if ineq else np.zeros(0))
