"""CPU test double of eqf_vio_amd.tiled.HipBackend (TEST INFRASTRUCTURE -- never imported by the package): the same backend protocol on
torch CPU tensors, so that the exchange schedule of eqf_vio_amd/tiled.py (TiledFilter._chain, the SUMMA-restricted broadcasts, the
closed loop) can run with gloo on machines without a GPU.  The O(N) state and the per-landmark blocks come from the numpy oracle
(oracle/eqf_numpy.py, the restatement of VIOFilter.cpp / EqFMatrices.cpp / VIOGroup.cpp); the dense tile operations are torch.linalg.
What the double computes per call is written to mirror csrc/eqf_tiled.hpp one to one:

    propagate      <-> k_tl_build + k_tl_base + k_tl_riccati     (VIOFilter.cpp:146-209)
    add_landmarks  <-> k_tl_append + k_tl_init_local             (VIOFilter.cpp:345-391 on an empty state)
    edit_landmarks <-> k_tl_edit_state + k_tl_edit_local         (VIOFilter.cpp:345-443 on landmark SLOTS: a removed landmark leaves an
                                                                  inactive slot -- zero rows / columns, unit diagonal, D = I, L = 0, C = 0)
    update_prep    <-> k_tl_prep + k_tl_eprep + k_tl_form_s + k_tl_form_e
    update_finish  <-> k_tl_finish (updateFinishBody)            (EqFMatrices.cpp:173-275, VIOFilter.cpp:295-296)
    potrf / trsm_left / gemm_tn <-> eqf_tile_potrf / eqf_tile_trsm / eqf_tile_gemm_tn
"""
import numpy as np
import torch

from oracle import eqf_numpy as O

NARROW_S, NARROW_E = 18, 11


def numpy_settings(d):
    dd = dict(d)
    cx, cq = dd.pop("cameraOffset_x"), dd.pop("cameraOffset_q")
    s = O.Settings(**dd)
    s.cameraOffset = O.SE3(np.asarray(cq, dtype=float), np.asarray(cx, dtype=float))
    return s


class NumpyBackend:
    DREC = 64 * 64 + 4 * 16 * 16

    def __init__(self, settings_dict, capacity):
        self.f = O.VIOFilter(numpy_settings(settings_dict))
        self.cap = capacity
        self.device = torch.device("cpu")
        S = self.f.Sigma
        self.Sbb = S[:11, :11].copy()        # replicated base block
        self.Sb = np.zeros((11, 0))          # replicated base rows of the landmark columns
        self._info = 0
        self.last = {}
        self.active = np.zeros(0, dtype=bool)  # per slot: holds a landmark

    # ---- plumbing
    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def empty(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def set_geometry(self, geo):
        self.geo = geo
        self.rows = (np.repeat(3 * geo.rowMap.astype(np.int64), 3) + np.tile(np.arange(3), geo.nlr))  # landmark coordinates 0 .. 3N-1
        self.cols = (np.repeat(3 * geo.colMap.astype(np.int64), 3) + np.tile(np.arange(3), geo.nlc))

    # ---- VIOFilter::processIMUData / integrateUpToTime without a dense Sigma
    def propagate(self, stamp, omega, accel, is_imu, Sll):
        f, s = self.f, self.f.settings
        unbiased = None
        if is_imu:
            unbiased = O.IMUVelocity(stamp, np.asarray(omega, dtype=float), np.asarray(accel, dtype=float)) - f.inputBias
            if not f.initialisedFlag:
                f.initialiseFromIMUData(unbiased)
        status = self._integrate(stamp, (not s.fastRiccati) if is_imu else True, Sll)
        if is_imu:
            f.currentVelocity = unbiased
            f.currentTime = stamp
            return status
        if status == 0 and not f.initialisedFlag:
            return 3
        return status

    BURST_MAX = 16

    def propagate_burst(self, records, vision_stamp, Sll):
        """eqf_tiled_propagate_burst's contract: the state after the calls, call for call (the device does it in one pass over Sll)"""
        status = [self.propagate(st, w, a, True, Sll) for (st, w, a) in records]
        if vision_stamp is not None:
            status.append(self.propagate(vision_stamp, None, None, False, Sll))
        return status

    def _integrate(self, newTime, doRiccati, Sll):
        f, s = self.f, self.f.settings
        if f.currentTime < 0:
            return 1
        dt = newTime - f.currentTime
        if dt <= 0:
            return 2
        f.accumulatedTime += dt
        f.accumulatedVelocity = f.accumulatedVelocity + f.currentVelocity * dt
        N = len(f.xi0.ids)
        currentState = f.stateEstimate()
        if doRiccati:
            T = f.accumulatedTime
            n = 11 + 3 * N
            A0t = O.eqf_state_matrix_A(f.X, f.xi0, f.accumulatedVelocity * (1.0 / T))
            Bt = O.eqf_input_matrix_B(f.X, f.xi0)
            Ab = np.zeros((n, n))
            Ab[6:, 6:] = A0t
            Ab[6:, 0:6] = -Bt
            F = np.eye(n) + Ab * T
            Bn = np.zeros((n, 6))
            Bn[6:] = Bt
            for i in np.nonzero(~self.active)[0]:  # a hole steps with the identity (k_tl_build): D = I, L = 0, no input noise
                F[11 + 3 * i: 14 + 3 * i, :] = 0.0
                F[11 + 3 * i: 14 + 3 * i, 11 + 3 * i: 14 + 3 * i] = np.eye(3)
                Bn[11 + 3 * i: 14 + 3 * i] = 0.0
            R = np.array([s.velOmegaVariance] * 3 + [s.velAccelVariance] * 3)
            P = np.array([s.biasOmegaProcessVariance] * 3 + [s.biasAccelProcessVariance] * 3 + [s.gravityProcessVariance] * 2
                         + [s.velocityProcessVariance] * 3 + [s.pointProcessVariance] * (3 * N))
            Q = T * (np.diag(P) + (Bn * R) @ Bn.T)
            Fbb, L, D = F[:11, :11], F[11:, :11], F[11:, 11:]  # F = [[Fbb, 0], [L, D]], D block diagonal
            Sbb, Sb = self.Sbb, self.Sb
            if N and Sll is not None and Sll.numel():
                r, c = self.rows, self.cols
                Sl = Sll.numpy()
                G_I = L[r] @ Sbb + D[np.ix_(r, r)] @ Sb[:, r].T
                Sl[...] = (D[np.ix_(r, r)] @ Sl + L[r] @ Sb[:, c]) @ D[np.ix_(c, c)].T + G_I @ L[c].T + Q[np.ix_(11 + r, 11 + c)]
            self.Sb = Fbb @ (Sbb @ L.T + Sb @ D.T) + Q[:11, 11:]
            self.Sbb = Fbb @ Sbb @ Fbb.T + Q[:11, :11]
            f.accumulatedVelocity = O.IMUVelocity()
            f.accumulatedTime = 0.0
        if s.useDiscreteVelocityLift:
            f.X = f.X * O.lift_velocity_discrete(currentState, f.currentVelocity, dt)
        else:
            f.X = f.X * O.vio_exp(dt * O.lift_velocity(currentState, f.currentVelocity))
        for i in np.nonzero(~self.active)[0]:
            f.X.Q[i] = O.SOT3()
        f.currentTime = newTime
        return 0

    def add_landmarks(self, bearings, Sll):
        f = self.f
        assert len(f.xi0.ids) == 0
        y = np.asarray(bearings, dtype=float).reshape(-1, 3)
        N = len(y)
        ids = np.arange(N, dtype=np.int64)
        f.xi0.p = y * f.settings.initialSceneDepth
        f.xi0.ids = ids.copy()
        f.X.ids = ids.copy()
        f.X.Q = [O.SOT3() for _ in range(N)]
        self.Sb = np.zeros((11, 3 * N))
        self.active = np.ones(N, dtype=bool)
        if Sll is not None and Sll.numel():
            Sl = Sll.numpy()
            Sl[...] = 0.0
            eq = self.rows[:, None] == self.cols[None, :]
            Sl[eq] = f.settings.initialPointVariance

    def initial_scene_depth(self):
        return float(self.f.settings.initialSceneDepth)

    def edit_landmarks(self, remove_slots, add_slots, add_bearings, depth, new_num_slots, Sll):
        f = self.f
        oldN = len(f.xi0.ids)
        n = max(oldN, int(new_num_slots))
        assert self.geo.N == n, "the geometry in force must cover max(old, new) slots"
        hole = np.array([1.0, 0.0, 0.0])
        if n > oldN:
            f.xi0.p = np.vstack([f.xi0.p.reshape(-1, 3), np.tile(hole, (n - oldN, 1))])
            f.X.Q = list(f.X.Q) + [O.SOT3() for _ in range(n - oldN)]
            self.Sb = np.hstack([self.Sb, np.zeros((11, 3 * (n - oldN)))])
            self.active = np.concatenate([self.active, np.zeros(n - oldN, dtype=bool)])
        mark = np.zeros(n, dtype=int)
        for s_ in np.asarray(remove_slots, dtype=int):
            assert self.active[s_] and mark[s_] == 0
            mark[s_] = 1
            f.xi0.p[s_] = hole
        y = np.asarray(add_bearings, dtype=float).reshape(-1, 3)
        for k, s_ in enumerate(np.asarray(add_slots, dtype=int)):
            assert (not self.active[s_] or mark[s_] == 1) and mark[s_] != 2 and s_ < new_num_slots
            mark[s_] = 2
            f.xi0.p[s_] = y[k] * depth
        for s_ in np.nonzero(mark)[0]:
            f.X.Q[s_] = O.SOT3()
            self.Sb[:, 3 * s_: 3 * s_ + 3] = 0.0
            self.active[s_] = mark[s_] == 2
        if Sll is not None and Sll.numel():
            Sl = Sll.numpy()
            mr, mc = mark[self.rows // 3], mark[self.cols // 3]
            Sl[mr > 0, :] = 0.0
            Sl[:, mc > 0] = 0.0
            eq = self.rows[:, None] == self.cols[None, :]
            Sl[eq & (mr[:, None] == 1)] = 1.0
            Sl[eq & (mr[:, None] == 2)] = f.settings.initialPointVariance
        assert not self.active[new_num_slots:].any()
        n = int(new_num_slots)
        f.xi0.p = f.xi0.p[:n].copy()
        f.X.Q = list(f.X.Q)[:n]
        self.Sb = self.Sb[:, : 3 * n].copy()
        self.active = self.active[:n].copy()
        f.xi0.ids = np.arange(n, dtype=np.int64)
        f.X.ids = np.arange(n, dtype=np.int64)

    # ---- first half of the update
    def _lift_rows(self):
        """Z_i = Qhat_i R_C^T pHatMat_i Ad(P0)  (3 x 6 per landmark; EqFMatrices.cpp:221-235)"""
        f = self.f
        xiHat = O.state_group_action(f.X, f.xi0)
        R_CT = O.quat_to_matrix(O.quat_inverse(O.quat_mul(xiHat.pose.q, xiHat.cameraOffset.q)))
        AdP0 = f.xi0.pose.adjoint()
        PC = xiHat.pose * xiHat.cameraOffset
        N = len(f.xi0.ids)
        Z = np.zeros((3 * N, 6))
        for i in range(N):
            pHat = PC.apply(xiHat.p[i])
            pHatMat = np.zeros((3, 6))
            pHatMat[:, 0:3] = -O.skew(pHat)
            pHatMat[:, 3:6] = np.eye(3)
            Z[3 * i:3 * i + 3] = f.X.Q[i].as_matrix3() @ R_CT @ pHatMat @ AdP0
        return Z

    def update_prep(self, bearings, Sll, M, E, G11):
        f, s, geo = self.f, self.f.settings, self.geo
        y = np.asarray(bearings, dtype=float).reshape(-1, 3)
        N = len(f.xi0.ids)
        y0 = O.measure_system_state(f.xi0)
        delta = O.output_coordinate_chart(O.output_group_action(f.X.inverse(), y), y0)
        C0 = O.eqf_output_matrix_C(f.xi0)[:, 5:]        # (2N x 3N), block diagonal
        Z = self._lift_rows()
        for i in np.nonzero(~self.active)[0]:  # a hole measures nothing (k_tl_prep, lmc = 0)
            delta[2 * i: 2 * i + 2] = 0.0
            C0[2 * i: 2 * i + 2, :] = 0.0
            Z[3 * i: 3 * i + 3] = 0.0
        V = C0 @ Z
        self.last = {"delta": delta}
        r, c = self.rows, self.cols
        r2 = (np.repeat(2 * geo.rowMap.astype(np.int64), 2) + np.tile(np.arange(2), geo.nlr))
        c2 = (np.repeat(2 * geo.colMap.astype(np.int64), 2) + np.tile(np.arange(2), geo.nlc))
        Sl = Sll.numpy()
        Cr, Cc = C0[np.ix_(r2, r)], C0[np.ix_(c2, c)]
        Wm = Cr @ Sl
        Am = Wm @ Cc.T + s.measurementVariance * (r2[:, None] == c2[None, :])
        Mn = M.numpy()
        Mn[:, : 2 * geo.nlc] = Am
        Mn[:, 2 * geo.nlc: 5 * geo.nlc] = Wm
        Mn[:, 5 * geo.nlc: 5 * geo.nlc + 11] = Cr @ self.Sb[:, r].T
        Mn[:, 5 * geo.nlc + 11] = delta[r2]
        Mn[:, 5 * geo.nlc + 12: 5 * geo.nlc + 18] = V[r2]
        # Sigma_e = Sigma[6:, 6:]: its five base coordinates eliminated, the Schur complement of the landmark part goes to the chain
        Lg = np.linalg.cholesky(self.Sbb[6:, 6:])
        Lgi = np.linalg.inv(Lg)
        Pg = Lgi @ self.Sb[6:, :]
        En = E.numpy()
        En[:, : 3 * geo.nlc] = Sl - Pg[:, r].T @ Pg[:, c]
        En[:, 3 * geo.nlc: 3 * geo.nlc + 6] = Z[r]
        En[:, 3 * geo.nlc + 6: 3 * geo.nlc + 11] = -Pg[:, r].T @ Lgi
        g = np.zeros((11, 11))
        g[6:, 6:] = Lgi.T @ Lgi
        G11.copy_(torch.from_numpy(g))

    # ---- second half: gamma, bundleLift through the Gram matrices, Delta, X <- Delta X, bias, base panel
    def update_finish(self, acc, Gnn, G11):
        f, s = self.f, self.f.settings
        acc, Gnn, G11 = acc.numpy(), Gnn.numpy(), G11.numpy()
        N = len(f.xi0.ids)
        gamma = np.concatenate([Gnn[:11, 11], acc[11, : 3 * N]])
        hV = Gnn[12:18, 11]
        gammaE = gamma[6:]
        xi0m = O.project_to_manifold(f.xi0)
        eta0 = xi0m.gravityDir / np.linalg.norm(xi0m.gravityDir)
        Gamma = None
        if s.useInnovationLift:
            dU0 = np.zeros(6)
            dU0[0:3] = -O.skew(eta0) @ O.stereo_sphere_chart_inv_diff(np.zeros(2), eta0) @ gammaE[0:2]
            KPara = np.zeros((6, 4))
            KPara[0:3, 0] = eta0
            KPara[3:6, 1:4] = np.eye(3)
            KPerp = np.zeros((6, 6))
            KPerp[0:3, 0:3] = np.eye(3) - np.outer(eta0, eta0)
            dUf = KPerp @ dU0
            G6, T65 = G11[:6, :6], G11[:6, 6:]
            rhs6 = -(hV - T65 @ gammaE[0:5]) - G6 @ dUf
            sol = np.linalg.solve(KPara.T @ G6 @ KPara, KPara.T @ rhs6)
            Gamma = np.zeros(9 + 3 * N)
            Gamma[0:6] = dUf + KPara @ sol
            Gamma[6:] = gammaE[2:]
            if s.useDiscreteInnovationLift:
                Delta = O.lift_total_space_innovation_discrete(Gamma, f.xi0)
            else:
                Delta = O.vio_exp(O.lift_total_space_innovation(Gamma, f.xi0))
        else:
            Delta = O.vio_exp(O.lift_innovation(gammaE, f.xi0))
        self.last.update(gamma=gamma, Gamma=Gamma)
        f.inputBias = f.inputBias + gamma[0:6]
        f.X = Delta * f.X
        self.Sb = self.Sb - acc[:11, : 3 * N]
        self.Sbb = self.Sbb - Gnn[:11, :11]

    # ---- dense tile operations
    # (no second stream on the CPU: the look-ahead of TiledFilter._chain runs in program order)
    def side(self):
        import contextlib

        return contextlib.nullcontext()

    main = aux = aux_side = side

    def record(self):
        return True

    def wait(self, ev):
        pass

    def potrf(self, Akk, drec=None):
        A = Akk.numpy()
        try:
            L = np.linalg.cholesky(np.tril(A) + np.tril(A, -1).T)
        except np.linalg.LinAlgError:
            self._info = 1
            L = np.eye(len(A))
        A[...] = np.tril(L) + np.triu(A, 1)  # (lower triangle <- L; the stale upper triangle stays, as on the device)
        return torch.zeros(((len(A) + 63) // 64) * self.DREC, dtype=torch.float64)

    def trsm_left(self, L, drec, Bm):
        Bm.copy_(torch.linalg.solve_triangular(torch.tril(L), Bm, upper=False))

    def gemm_tn(self, Cm, A, B, alpha, mask=None):
        if Cm.numel() == 0 or A.shape[0] == 0:
            return
        P = A.T @ B
        if mask is not None:
            rb, cb, rblk0, Pr, pr, cblk0, Pc, pc = mask
            I = (rblk0 + torch.arange(Cm.shape[0]) // rb) * Pr + pr
            J = (cblk0 + torch.arange(Cm.shape[1]) // cb) * Pc + pc
            P = torch.where(I[:, None] <= J[None, :], P, torch.zeros_like(P))  # blocks below the diagonal: untouched
        Cm.add_(P, alpha=alpha)

    def mirror_lower(self, Cm, rb):
        n = Cm.shape[0]
        I = torch.arange(n) // rb
        low = I[:, None] > I[None, :]
        Cm.copy_(torch.where(low, Cm.T, Cm))

    def factor_info(self):
        v, self._info = self._info, 0
        return v

    # ---- getters
    def num_landmarks(self):
        return len(self.f.xi0.ids)

    def time(self):
        return self.f.currentTime

    def device_error(self):
        return 0

    def outlier_threshold(self):
        return float(self.f.settings.outlierThreshold)

    def state_estimate(self):
        e = self.f.stateEstimate()
        return {"q": e.pose.q, "x": e.pose.x, "v": e.velocity, "p": e.p}

    def bias(self):
        return self.f.inputBias

    def last_update(self):
        return self.last

    def base_rows(self):
        return np.hstack([self.Sbb, self.Sb])
