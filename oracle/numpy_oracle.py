"""Second, independent restatement of the reference filters, on the LAPACK drivers Armadillo itself calls.

TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see kfpos_oracle.h). Its job is to cross-check the C++ oracle: same
algorithm, different author-time, and -- instead of the C++ oracle's hand-written LU / Jacobi-SVD -- the LAPACK routines
behind arma::inv / pinv / solve, reached through scipy.linalg.lapack (the same reference LAPACK interface; the BLAS
underneath is whatever scipy ships, as it would be whatever the reference's host ships). One tag per object,
pure-Python loops: small cases only.

Which driver Armadillo calls depends on its version, and the reference pins none (CMakeLists.txt:29,
find_package(Armadillo REQUIRED)). FLAVOUR selects one of three generations of the dispatch, so that the tests can
state the spread over all of them instead of picking one (KFPOS_ARMA_FLAVOUR in the environment, or set_flavour()):

  "lapack" (default) -- the general dense path every version ends in:
      inv(A)                         dgetrf + dgetri                     (auxlib::inv)
      pinv(A)                        dgesdd, cut at max(m,n) s_max eps   (op_pinv via svd_econ "dc")
      solve(A, b, equilibrate)       dgesvx fact='E'; rcond < eps or failure -> dgelsd minimum-norm solution
                                     (auxlib::solve_square_refine / solve_approx_svd)
      solve(A, b)                    dgetrf + dgetrs + dgecon, same fallback (auxlib::solve_square_rcond)
  "old"    -- Armadillo <= 8 as shipped with the ROS releases of the reference's time: as "lapack", but inv() of
      matrices up to 4x4 in closed form (cofactors / determinant; the tiny-matrix path -- its exact expression order
      is not restated, only its kind) and plain solve() through dgesv without a condition estimate
  "new"    -- Armadillo >= 9/10: structure detection in front of the general path: inv() of a diagonal matrix by
      reciprocals, of a matrix that looks symmetric positive definite by dpotrf + dpotri; pinv() of a symmetric matrix
      through dsyevd; solve(.., equilibrate) of a matrix that looks sympd by dposvx fact='E'

Reference map: MLLocation.cpp:24-37,153-278; KalmanFilterTOA.cpp:70-156,185-338,362-391,438-473;
KalmanFilterTOAIMU.cpp:100-195,242-340,392-473 (with the 3-token repair of SURVEY.md 0.2).
"""
from __future__ import annotations

import os

import numpy as np
from scipy.linalg import lapack as _la

EPS = np.finfo(np.float64).eps
FLAVOURS = ("lapack", "old", "new")
FLAVOUR = os.environ.get("KFPOS_ARMA_FLAVOUR", "lapack")


def set_flavour(name: str):
    global FLAVOUR
    if name not in FLAVOURS:
        raise ValueError(name)
    FLAVOUR = name


class LinAlgThrow(Exception):
    """Stands for the std::runtime_error Armadillo throws from inv / pinv / solve."""


def _stdmax(a, b):
    return b if a < b else a


def _fortran(A):
    return np.asfortranarray(A, dtype=np.float64)


def _looks_sympd(A):
    """Armadillo's cheap guess (sympd_helper::guess_sympd in kind): symmetric to a relative tolerance, positive
    diagonal, no off-diagonal entry dominating the diagonal."""
    n = A.shape[0]
    if n < 2 or A.shape[0] != A.shape[1]:
        return False
    d = np.diag(A)
    if np.any(d <= 0):
        return False
    tol = 100 * EPS
    if np.any(np.abs(A - A.T) > tol * np.maximum(np.abs(A), np.abs(A.T)) + 0 * tol):
        return False
    off = np.abs(A - np.diag(d))
    return bool(np.all(off.max(initial=0.0) < d.max())) and bool(np.all(2 * off <= d[:, None] + d[None, :]))


def _inv_cofactor(A):
    """closed-form inverse of a 1x1 .. 4x4 matrix: adjugate / determinant (FLAVOUR "old": the tiny-matrix path)"""
    n = A.shape[0]
    if n == 1:
        if A[0, 0] == 0:
            raise LinAlgThrow("inv")
        return np.array([[1.0 / A[0, 0]]])
    adj = np.empty((n, n))
    for i in range(n):
        for j in range(n):
            minor = np.delete(np.delete(A, i, 0), j, 1)
            adj[j, i] = (-1.0) ** (i + j) * _det_small(minor)
    det = sum(A[0, j] * adj[j, 0] for j in range(n))
    if not abs(det) >= EPS:   # the tiny path hands anything this small (or NaN) to LAPACK instead
        return None
    return adj / det


def _det_small(M):
    n = M.shape[0]
    if n == 1:
        return M[0, 0]
    if n == 2:
        return M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    return sum((-1.0) ** j * M[0, j] * _det_small(np.delete(np.delete(M, 0, 0), j, 1)) for j in range(n))


def arma_inv(A):
    """arma::inv(A) for a square dense A. Throws (LinAlgThrow) where Armadillo's inv() returns false: an exactly
    zero pivot."""
    if A.size == 0:
        return A.copy()
    if not np.all(np.isfinite(A)):
        return np.full_like(A, np.nan)  # NaN in, NaN out (dgetrf / dgetri do not reject it)
    n = A.shape[0]
    if FLAVOUR == "old" and n <= 4:
        out = _inv_cofactor(A)
        if out is not None:
            return out
    if FLAVOUR == "new":
        if n > 1 and not np.any(A - np.diag(np.diag(A))):  # is_diagmat(): reciprocals of the diagonal
            d = np.diag(A)
            if np.any(d == 0):
                raise LinAlgThrow("inv")
            return np.diag(1.0 / d)
        if _looks_sympd(A):
            c, info = _la.dpotrf(_fortran(A), lower=1)
            if info == 0:
                ci, info = _la.dpotri(c, lower=1)
                if info == 0:
                    low = np.tril(ci)
                    return low + np.tril(ci, -1).T
            # not positive definite after all: the general path
    lu, piv, info = _la.dgetrf(_fortran(A))
    if info != 0:
        raise LinAlgThrow("inv")
    out, info = _la.dgetri(lu, piv)
    if info != 0:
        raise LinAlgThrow("inv")
    return np.ascontiguousarray(out)


def arma_pinv(A):
    """arma::pinv(A): SVD by divide and conquer, singular values <= max(m,n) s_max eps dropped."""
    if A.size == 0:
        return A.copy()
    if not np.all(np.isfinite(A)):
        raise LinAlgThrow("pinv")
    if FLAVOUR == "new" and A.shape[0] == A.shape[1] and np.array_equal(A, A.T):
        w, v, info = _la.dsyevd(_fortran(A), compute_v=1, lower=1)   # op_pinv::apply_sym
        if info != 0:
            raise LinAlgThrow("pinv")
        aw = np.abs(w)
        tol = A.shape[0] * aw.max() * EPS
        keep = aw > tol
        if not keep.any():
            return np.zeros_like(A)
        return (v[:, keep] / w[keep]) @ v[:, keep].T
    u, s, vt, info = _la.dgesdd(_fortran(A), compute_uv=1, full_matrices=0)
    if info != 0:
        raise LinAlgThrow("pinv")
    tol = max(A.shape) * s[0] * EPS
    keep = s > tol
    if not keep.any():
        return np.zeros((A.shape[1], A.shape[0]))
    return (vt[keep].T / s[keep]) @ u[:, keep].T


def _approx_svd(A, b):
    """auxlib::solve_approx_svd: the minimum-norm least-squares solution by dgelsd at machine-precision rank cut"""
    m, n = A.shape
    nrhs = 1
    work, iwork, info = _la.dgelsd_lwork(m, n, nrhs, -1.0)
    x, _, _, info = _la.dgelsd(_fortran(A), _fortran(b.reshape(-1, 1)), int(work), int(iwork), cond=-1.0)
    if info != 0:
        raise LinAlgThrow("solve")
    return np.ascontiguousarray(x[:n, 0])


def arma_solve_equilibrate(A, b):
    """arma::solve(A, b, solve_opts::equilibrate), square A (MLLocation.cpp:210): the expert driver with
    equilibration and iterative refinement; a singular or (rcond < eps) near-singular system gets the approximate
    solution instead (Armadillo warns and carries on)."""
    if not (np.all(np.isfinite(A)) and np.all(np.isfinite(b))):
        return np.full_like(b, np.nan)  # the reference tests the result for NaN (KalmanFilterTOA.cpp:270)
    n = A.shape[0]
    if FLAVOUR == "new" and _looks_sympd(A):
        out = _la.dposvx(_fortran(A), _fortran(b.reshape(-1, 1)), fact="E", lower=1)
        x, rcond, info = out[5], out[6], out[-1]   # a_s, lu, equed, s, b_s, x, rcond, ferr, berr, info
        if (info == 0 or info == n + 1) and rcond >= EPS:
            return np.ascontiguousarray(x[:, 0])
        if info == 0 or info == n + 1:
            return _approx_svd(A, b)
        # not positive definite: the general expert driver
    out = _la.dgesvx(_fortran(A), _fortran(b.reshape(-1, 1)), fact="E")
    x, rcond, info = out[7], out[8], out[-1]
    if (info == 0 or info == n + 1) and rcond >= EPS:
        return np.ascontiguousarray(x[:, 0])
    return _approx_svd(A, b)


def distances(p, meas):
    return [np.sqrt((bx - p[0]) * (bx - p[0]) + (by - p[1]) * (by - p[1]) + (bz - p[2]) * (bz - p[2]))
            for (_, _, bx, by, bz) in meas]


def ml_error(meas, p):
    if not meas:
        return -1.0
    d = distances(p, meas)
    return float(sum((d[i] - meas[i][0]) ** 2 for i in range(len(meas))))


def ml_estimate(meas, seed):
    """meas: list of (range, errEst, bx, by, bz). Returns (pos, cov or None, iters)."""
    p = np.array(seed, dtype=np.float64)
    n = len(meas)
    if n < 4:
        return p, None, 0
    cost, new_cost, it = np.float64(1e20), np.float64(1.0), 0
    with np.errstate(all="ignore"):
        while abs(cost - new_cost) / cost > 1e-3 and it < 10000:
            it += 1
            cost = new_cost
            d = distances(p, meas)
            g, Hs = np.zeros(3), np.zeros((3, 3))
            for i, (r, e, bx, by, bz) in enumerate(meas):
                v = np.array([bx - p[0], by - p[1], bz - p[2]])
                g += (r - d[i]) * v / (d[i] * e)
                d3 = d[i] ** 3
                for k in range(3):
                    Hs[k, k] += (1 - r / d[i] + r * v[k] * v[k] / d3) / e
                for (k, l) in ((0, 1), (0, 2), (1, 2)):
                    t = r * v[k] * v[l] / (d3 * e)
                    Hs[k, l] += t
                    Hs[l, k] += t
            p = arma_solve_equilibrate(Hs, Hs @ p - g)
            d = distances(p, meas)
            new_cost = np.float64(sum((meas[i][0] - d[i]) ** 2 / meas[i][1] for i in range(n)))
        d = distances(p, meas)
        rng_err = ml_error(meas, p)
        J = np.array([[(p[0] - m[2]) / d[i], (p[1] - m[3]) / d[i], (p[2] - m[4]) / d[i]]
                      for i, m in enumerate(meas)])
        obs = np.array([_stdmax(m[1], rng_err) for m in meas])
        C = arma_inv(J.T @ arma_inv(np.diag(obs)) @ J)
    return p, C, it


def pred_F(n, t):
    F = np.eye(n)
    for k in range(3):
        F[k, k + 3] = t
        if n == 9:
            F[k, k + 6] = t * t / 2
            F[k + 3, k + 6] = t
    return F


def pred_Q(n, t, accel_noise, jolt):
    Q = np.zeros((n, n))
    if n == 6:
        t2, a2 = t ** 2 / 2, accel_noise * accel_noise
        for k in range(3):
            Q[k, k] = a2 * t2 * t2
            Q[k, k + 3] = Q[k + 3, k] = a2 * t2 * t
            Q[k + 3, k + 3] = a2 * t * t
    else:
        u = [t ** 3 / 6, t ** 2 / 2, t]
        for k in range(3):
            for a in range(3):
                for b in range(3):
                    Q[k + 3 * a, k + 3 * b] = jolt * u[min(a, b)] * u[max(a, b)]
    return Q


def iekf_step(n, pred, Pm, has_ranging, meas_all, ignored, has_imu, imu_acc, imu_cov, max_steps, tol):
    """Shared body of kalmanStep3DIgnoreAnchor / kalmanStep3D. Returns (state, P, cost, gain_iters)."""
    rm = [m for i, m in enumerate(meas_all) if i != ignored] if has_ranging else []
    x = np.array(pred, dtype=np.float64)
    nr = len(rm)
    m = nr + (3 if has_imu else 0)
    R, z = np.eye(m), np.zeros(m)
    with np.errstate(all="ignore"):
        if has_ranging:
            ml, _, _ = ml_estimate(rm, x[:3])
            if n == 6 and np.any(np.isnan(ml)):
                ml = x[:3].copy()
            e_ml = ml_error(rm, ml)
            for i, mm in enumerate(rm):
                z[i] = mm[0]
                R[i, i] = _stdmax(e_ml, mm[1])
        if has_imu:
            z[nr:] = imu_acc
            R[nr:, nr:] = np.asarray(imu_cov).reshape(3, 3)
        Ri, Pp = arma_inv(R), arma_pinv(Pm)
        H, K = np.zeros((m, n)), np.zeros((n, m))
        cost, gains = np.float64(1e20), 0
        for _ in range(max_steps):
            h = np.zeros(m)
            d = distances(x[:3], rm)
            h[:nr] = d
            if has_imu:
                h[nr:] = x[6:9]
            y, dl = z - h, pred - x
            new_cost = np.float64(y @ Ri @ y + dl @ Pp @ dl)
            if abs(cost - new_cost) / cost < tol:
                break
            cost = new_cost
            H[:] = 0.0
            for i, mm in enumerate(rm):
                H[i, 0:3] = [(x[0] - mm[2]) / d[i], (x[1] - mm[3]) / d[i], (x[2] - mm[4]) / d[i]]
            if has_imu:
                for k in range(3):
                    H[nr + k, 6 + k] = x[6 + k]
            K = Pm @ H.T @ arma_inv(H @ Pm @ H.T + R)
            x = x + (dl + K @ (y - H @ dl))
            gains += 1
        Pn = (np.eye(n) - K @ H) @ Pm
    return x, Pn, cost, gains


class NumpyFilter:
    """One reference filter; dt is passed in instead of read from the wall clock."""

    def __init__(self, model, anchors, accel_noise=0.5, jolt=0.5, ignore_worst=False,
                 cost_threshold=0.5, init_pos=None):
        self.n = 9 if model == 1 else 6
        self.anchors = np.asarray(anchors, dtype=np.float64)
        self.accel_noise, self.jolt = accel_noise, jolt
        self.ignore_worst, self.cost_threshold = ignore_worst, cost_threshold
        self.fixed = init_pos is not None
        self.pos = np.array(init_pos, dtype=np.float64) if self.fixed else np.full(3, np.nan)
        self.vel = np.zeros(3)
        self.P = np.zeros((self.n, self.n))
        self.imu = None

    def _meas(self, range_mm, err_est):
        return [(float(mm) / 1000, float(e), *self.anchors[a]) for a, (mm, e) in
                enumerate(zip(range_mm, err_est)) if mm > 0]

    def step_toa(self, range_mm, err_est, dt):
        meas = self._meas(range_mm, err_est)
        if self.n == 6:
            return self._toa6(meas, dt)
        return self._imu9(True, meas, self.imu is not None, dt)

    def step_imu(self, accel, cov, dt):
        if self.n == 9:
            self.imu = (np.array(accel, dtype=np.float64), np.array(cov, dtype=np.float64))
            self._imu9(False, [], True, dt)

    def _toa6(self, meas, dt):
        if not self.fixed and np.any(np.isnan(self.pos)):
            if len(meas) < 4:
                return
            p, C, _ = ml_estimate(meas, [1.0, 1.0, 4.0])
            self.pos = p
            for i in range(3):
                self.P[i, 0], self.P[i, 1], self.P[i, 2] = C[i, 0], C[i, 1], C[i, 1]
            return
        F, Q = pred_F(6, dt), pred_Q(6, dt, self.accel_noise, self.jolt)
        pred = F @ np.concatenate([self.pos, np.zeros(3)])
        self.P = F @ self.P @ F.T + Q
        try:
            if len(meas) > 4 and self.ignore_worst:
                x_all, P_all, c_all, _ = iekf_step(6, pred, self.P, True, meas, -1, False, None, None, 10, 1e-3)
                best, max_d, worst_cost = None, 0.0, 0.0
                for i, mm in enumerate(meas):
                    xi, Pi, ci, _ = iekf_step(6, pred, self.P, True, meas, i, False, None, None, 10, 1e-3)
                    dd = mm[0] - np.sqrt((mm[2] - xi[0]) ** 2 + (mm[3] - xi[1]) ** 2 + (mm[4] - xi[2]) ** 2)
                    if i == 0 or dd > max_d:
                        max_d, worst_cost, best = dd, ci, (xi, Pi)
                x, Pn = x_all, P_all
                if max_d > 0 and (c_all - worst_cost) > self.cost_threshold:
                    x, Pn = best
            else:
                x, Pn, _, _ = iekf_step(6, pred, self.P, True, meas, -1, False, None, None, 10, 1e-3)
        except LinAlgThrow:
            return
        self.P, self.pos = Pn, x[:3].copy()

    def _imu9(self, has_ranging, meas, has_imu, dt):
        if not self.fixed and (np.isnan(self.pos[0]) or np.isnan(self.pos[1])):
            if has_ranging and len(meas) >= 4:
                p, C, _ = ml_estimate(meas, [1.0, 1.0, 4.0])
                self.pos = p
                self.P[0, 0], self.P[1, 0], self.P[0, 1], self.P[1, 1] = C[0, 0], C[1, 0], C[0, 1], C[1, 1]
            return
        F, Q = pred_F(9, dt), pred_Q(9, dt, self.accel_noise, self.jolt)
        pred = F @ np.concatenate([self.pos, self.vel, np.zeros(3)])
        self.P = F @ self.P @ F.T + Q
        acc, cov = self.imu if has_imu else (None, None)
        try:
            x, Pn, _, _ = iekf_step(9, pred, self.P, has_ranging, meas, -1, has_imu, acc, cov, 20, 1e-4)
        except LinAlgThrow:
            return
        self.P, self.pos, self.vel = Pn, x[:3].copy(), x[3:6].copy()

    def get_pose(self, dt_ahead):
        F, Q = pred_F(self.n, dt_ahead), pred_Q(self.n, dt_ahead, self.accel_noise, self.jolt)
        st = np.concatenate([self.pos, self.vel if self.n == 9 else np.zeros(3), np.zeros(self.n - 6)])
        Pp = F @ self.P @ F.T + Q
        return (F @ st)[:3], Pp[:3, :3]


# ---------------------------------------------------------------------------------------------------
# 8-state planar filter (KalmanFilter, ALGORITHM_KF): KalmanFilter.cpp:66-229, 224-321, 365-501, 558-745;
# MLLocation::estimatePosition2D, MLLocation.cpp:48-143 (with the tentative-z repair of DESIGN.md).
# ---------------------------------------------------------------------------------------------------
def arma_solve(A, b):
    """arma::solve(A, b) with default options, square A (MLLocation.cpp:101): LU with a reciprocal condition
    estimate, and the approximate (dgelsd) solution when rcond < eps; FLAVOUR "old": dgesv, approximate solution only
    when the factorisation hits an exactly zero pivot."""
    if not (np.all(np.isfinite(A)) and np.all(np.isfinite(b))):
        return np.full_like(b, np.nan)
    if FLAVOUR == "old":
        _, _, x, info = _la.dgesv(_fortran(A), _fortran(b.reshape(-1, 1)))
        return np.ascontiguousarray(x[:, 0]) if info == 0 else _approx_svd(A, b)
    anorm = _la.dlange("1", _fortran(A))
    lu, piv, info = _la.dgetrf(_fortran(A))
    if info != 0:
        return _approx_svd(A, b)
    rcond, info2 = _la.dgecon(lu, anorm, norm="1")
    if info2 != 0 or rcond < EPS:
        return _approx_svd(A, b)
    x, info = _la.dgetrs(lu, piv, _fortran(b.reshape(-1, 1)))
    return np.ascontiguousarray(x[:, 0])


def ml_estimate_2d(meas, seed):
    p = np.array(seed, dtype=np.float64)
    n = len(meas)
    if n < 3:
        return p, None, 0
    cost, new_cost, step, it = np.float64(1e20), np.float64(ml_error(meas, p)), 1.0, 0
    with np.errstate(all="ignore"):
        while abs(cost - new_cost) / cost > 1e-3 and it < 10000:
            it += 1
            cost = new_cost
            d = distances(p, meas)
            g, Hs = np.zeros(2), np.zeros((2, 2))
            for i, (r, e, bx, by, _) in enumerate(meas):
                v = np.array([bx - p[0], by - p[1]])
                g += (r - d[i]) * v / (d[i] * e)
                d3 = d[i] * d[i] * d[i]
                Hs[0, 0] += (1 - r / d[i] + r * v[0] * v[0] / d3) / e
                Hs[1, 1] += (1 - r / d[i] + r * v[1] * v[1] / d3) / e
                t = r * v[0] * v[1] / (d3 * e)
                Hs[0, 1] += t
                Hs[1, 0] += t
            q = arma_solve(Hs, Hs @ p[:2] - g * step)
            tent = np.array([q[0], q[1], p[2]])
            tent_cost = np.float64(ml_error(meas, tent))
            if tent_cost > cost:
                step /= 2
            else:
                new_cost, step, p = tent_cost, 1.0, tent
        d = distances(p, meas)
        rng_err = ml_error(meas, p)
        J = np.array([[(p[0] - m[2]) / d[i], (p[1] - m[3]) / d[i]] for i, m in enumerate(meas)])
        obs = np.array([_stdmax(m[1], rng_err) for m in meas])
        C = arma_inv(J.T @ arma_inv(np.diag(obs)) @ J)
    return p, C, it


def planar_F(t):
    F = np.eye(8)
    F[0, 2] = F[1, 3] = F[2, 4] = F[3, 5] = F[6, 7] = t
    F[0, 4] = F[1, 5] = t * t / 2
    return F


def planar_Q(t, accel_noise, jolt):
    u = [t ** 3 / 6, t ** 2 / 2, t]
    Q = np.zeros((8, 8))
    for k in range(2):
        for a in range(3):
            for b in range(3):
                Q[k + 2 * a, k + 2 * b] = jolt * u[min(a, b)] * u[max(a, b)]
    Q[6, 6], Q[6, 7], Q[7, 6], Q[7, 7] = accel_noise * u[1] * u[1], accel_noise * u[1] * t, accel_noise * u[1] * t, accel_noise * t * t
    return Q


def normalize_angle(a):
    if a > np.pi:
        return a - 2 * np.pi
    if a <= -np.pi:
        return a + 2 * np.pi
    return a


class NumpyPlanarFilter:
    """One KalmanFilter (8 states). cfg keys as oracle_py.PlanarConfig."""

    def __init__(self, anchors, accel_noise=0.5, jolt=0.5, init_pos=None, **cfg):
        self.anchors = np.asarray(anchors, dtype=np.float64)
        self.accel_noise, self.jolt = accel_noise, jolt
        self.c = dict(use_fixed_height=0, fixed_height=0.0, init_angle=0.0, px4_height=0.0, px4_arm_p1=0.0,
                      px4_arm_p2=0.0, px4_cov_velocity=0.0, px4_cov_gyro_z=0.0, imu_use_fixed_cov_acc=0,
                      imu_cov_acc=0.0, imu_use_fixed_cov_ang_vel_z=0, imu_cov_ang_vel_z=0.0, mag_angle_offset=0.0,
                      mag_cov=0.0)
        self.c.update(cfg)
        self.fixed = init_pos is not None
        self.xy = np.array(init_pos[:2], dtype=np.float64) if self.fixed else np.full(2, np.nan)
        self.z = float(self.c["fixed_height"])
        self.vel = np.zeros(2)
        self.ang, self.ang_speed = float(self.c["init_angle"]), 0.0
        self.P = np.zeros((8, 8))
        self.px4 = self.imu = self.mag = None

    def _meas(self, range_mm, err_est):
        return [(float(mm) / 1000, float(e), *self.anchors[a]) for a, (mm, e) in
                enumerate(zip(range_mm, err_est)) if mm > 0]

    def step_toa(self, range_mm, err_est, dt):
        self._estimate(self._meas(range_mm, err_est), self.px4, self.imu, self.mag, dt, True)

    def step_px4flow(self, ix, iy, irz, itime_us, quality, dt):
        c, quality = self.c, int(quality)
        with np.errstate(all="ignore"):
            tsec = np.float64(itime_us) / 1000000.0
            vx, vy, gz = ix / tsec * c["px4_height"], iy / tsec * c["px4_height"], irz / tsec
            if quality == 0:
                return
            cv = c["px4_cov_velocity"] / tsec * c["px4_height"] / quality if itime_us > 0 else c["px4_cov_velocity"] * quality
        self.px4 = (vx, vy, gz, cv, c["px4_cov_gyro_z"])
        self._estimate(None, self.px4, None, None, dt, False)

    def step_imu(self, ang_vel, cov_ang_vel, lin_acc, cov_acc, dt):
        c, ca = self.c, np.asarray(cov_acc, dtype=np.float64).ravel()
        cxy = np.array([[c["imu_cov_acc"] if c["imu_use_fixed_cov_acc"] else ca[0], ca[1]],
                        [ca[3], c["imu_cov_acc"] if c["imu_use_fixed_cov_acc"] else ca[4]]])
        cw = c["imu_cov_ang_vel_z"] if c["imu_use_fixed_cov_ang_vel_z"] else np.asarray(cov_ang_vel).ravel()[8]
        self.imu = (lin_acc[0], lin_acc[1], ang_vel[2], cxy, cw)
        self._estimate(None, None, self.imu, None, dt, False)

    def step_mag(self, mag_xyz, dt):
        self.mag = (np.arctan2(mag_xyz[1], mag_xyz[0]) - self.c["mag_angle_offset"], self.c["mag_cov"])
        self._estimate(None, None, None, self.mag, dt, False)

    def step_compass(self, compass, dt):
        self.mag = (normalize_angle(compass), self.c["mag_cov"])
        self._estimate(None, self.px4, self.imu, self.mag, dt, False)

    def _estimate(self, meas, px4, imu, mag, dt, has_r):
        if not self.fixed and np.any(np.isnan(self.xy)):
            if has_r:
                if self.c["use_fixed_height"]:
                    if len(meas) < 3:
                        return
                    p, C, _ = ml_estimate_2d(meas, [1.0, 1.0, self.z])
                else:
                    if len(meas) < 4:
                        return
                    p, C, _ = ml_estimate(meas, [1.0, 1.0, 4.0])
                    self.z = p[2]
                self.xy = p[:2].copy()
                self.P[:2, :2] = C[:2, :2]
            return
        F, Q = planar_F(dt), planar_Q(dt, self.accel_noise, self.jolt)
        pred = F @ np.array([self.xy[0], self.xy[1], self.vel[0], self.vel[1], 0.0, 0.0, self.ang, self.ang_speed])
        self.P = F @ self.P @ F.T + Q
        pred[6] = normalize_angle(pred[6])
        try:
            x, Pn = self._kalman_step(pred, self.P, meas if has_r else [], px4, imu, mag, dt)
        except LinAlgThrow:
            return
        self.P, self.xy, self.vel, self.ang, self.ang_speed = Pn, x[:2].copy(), x[2:4].copy(), x[6], x[7]

    def _kalman_step(self, pred, Pm, rm, px4, imu, mag, t):
        nr = len(rm)
        i_px4 = nr
        i_imu = i_px4 + (3 if px4 else 0)
        i_mag = i_imu + (3 if imu else 0)
        m = i_mag + (1 if mag else 0)
        R, zz = np.eye(m), np.zeros(m)
        x = np.array(pred, dtype=np.float64)
        p1, p2 = self.c["px4_arm_p1"], self.c["px4_arm_p2"]
        with np.errstate(all="ignore"):
            if nr > 0:
                ml, _, _ = ml_estimate_2d(rm, [x[0], x[1], self.z])
                e_ml = ml_error(rm, ml)
                for i, mm in enumerate(rm):
                    zz[i], R[i, i] = mm[0], _stdmax(e_ml, mm[1])
            if px4:
                zz[i_px4:i_px4 + 3] = px4[:3]
                R[i_px4, i_px4] = R[i_px4 + 1, i_px4 + 1] = px4[3]
                R[i_px4 + 2, i_px4 + 2] = px4[4]
            if imu:
                zz[i_imu:i_imu + 3] = imu[:3]
                R[i_imu:i_imu + 2, i_imu:i_imu + 2] = imu[3]
                R[i_imu + 2, i_imu + 2] = imu[4]
            if mag:
                zz[i_mag], R[i_mag, i_mag] = mag
            Ri, Pp = arma_inv(R), arma_pinv(Pm)
            H, K = np.zeros((m, 8)), np.zeros((8, m))
            cost = np.float64(1e20)
            for _ in range(20):
                vx, vy, ax, ay, th, om = x[2:8]
                h = np.zeros(m)
                d = distances([x[0], x[1], self.z], rm)
                h[:nr] = d
                c_, s_ = np.cos(th), np.sin(th)
                if px4:
                    h[i_px4] = c_ * vx + s_ * vy + 1 / t * ((1 - np.cos(om * t)) * p1 - np.sin(om * t) * p2)
                    h[i_px4 + 1] = -s_ * vx + c_ * vy + 1 / t * (np.sin(om * t) * p1 + (1 - np.cos(om * t)) * p2)
                    h[i_px4 + 2] = om
                if imu:
                    h[i_imu:i_imu + 3] = [c_ * ax + s_ * ay, -s_ * ax + c_ * ay, om]
                if mag:
                    h[i_mag] = th
                y, dl = zz - h, pred - x
                if mag:
                    y[i_mag] = normalize_angle(y[i_mag])
                new_cost = np.float64(y @ Ri @ y + dl @ Pp @ dl)
                if abs(cost - new_cost) / cost < 1e-4:
                    break
                cost = new_cost
                H[:] = 0.0
                for i, mm in enumerate(rm):
                    H[i, 0], H[i, 1] = (x[0] - mm[2]) / d[i], (x[1] - mm[3]) / d[i]
                if px4:
                    H[i_px4, [2, 3, 6, 7]] = [c_, s_, -s_ * vx + c_ * vy, p1 * np.sin(om * t) - p2 * np.cos(om * t)]
                    H[i_px4 + 1, [2, 3, 6, 7]] = [-s_, c_, -c_ * vx - s_ * vy, p1 * np.cos(om * t) + p2 * np.sin(om * t)]
                    H[i_px4 + 2, 7] = 1
                if imu:
                    H[i_imu, [4, 5, 6]] = [c_, s_, -s_ * ax + c_ * ay]
                    H[i_imu + 1, [4, 5, 6]] = [-s_, c_, -c_ * ax - s_ * ay]
                    H[i_imu + 2, 7] = 1
                if mag:
                    H[i_mag, 6] = 1
                K = Pm @ H.T @ arma_inv(H @ Pm @ H.T + R)
                x = x + (dl + K @ (y - H @ dl))
            Pn = (np.eye(8) - K @ H) @ Pm
        return x, Pn

    def state(self):
        return np.array([self.xy[0], self.xy[1], self.vel[0], self.vel[1], 0.0, 0.0, self.ang, self.ang_speed])
