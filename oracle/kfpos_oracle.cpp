/*
 * kfpos_oracle.cpp -- CPU restatement of roskfpos's EKF predict/update hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see kfpos_oracle.h).  PARITY UNPINNED: no reference
 * golden vector exists and the reference cannot be built in this image.
 *
 * The code below follows the reference line by line, dense and in the same
 * operation order, with every Armadillo call replaced by the mathematical
 * contract Armadillo documents for it:
 *   inv(A)            LU with partial pivoting; failure ("std::runtime_error")
 *                     on an exactly zero pivot; NaN/inf input propagates (LAPACK
 *                     getrf/getri do not reject it)
 *   pinv(A)           SVD, singular values below max(m,n)*s_max*eps dropped;
 *                     failure on non-finite input (the SVD does not converge)
 *   solve(A,b,equilibrate)
 *                     row/column equilibration, LU, one step of iterative
 *                     refinement; rcond < eps falls back to pinv(A)*b; NaN/inf
 *                     input gives a NaN solution, not a failure -- the reference
 *                     itself tests the ML position for NaN and falls back
 *                     (KalmanFilterTOA.cpp:270-272), which only makes sense if
 *                     solve() hands NaN through
 *   std::max(a,b)     (a < b) ? b : a   -- matters for NaN
 * Armadillo itself is an unpinned third-party dependency of the reference
 * (CMakeLists.txt:29, find_package(Armadillo REQUIRED), no version).
 *
 * Reference map (paths relative to /root/reference/src/kfpos/algorithms):
 *   ml_estimate            MLLocation.cpp:153-257
 *   ml_error               MLLocation.cpp:263-278
 *   distances              MLLocation.cpp:24-37
 *   topn_keep              MLLocation.cpp:284-300, 325-339 (composition)
 *   toa6_* / Toa6          KalmanFilterTOA.cpp:43-61,70-156,185-238,242-338,341-433,438-473
 *   imu9_* / Imu9          KalmanFilterTOAIMU.cpp:49-92,100-195,242-340,345-473,476-510
 *                          with the documented 3-token repair:
 *                          countValid+=9 -> +=3 (:261,:356), jacobian(.,9) -> (.,8) (:451)
 */
#include "kfpos_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

const double EPS = std::numeric_limits<double>::epsilon();

inline double stdmax(double a, double b) { return (a < b) ? b : a; } /* std::max semantics */

/* ------------------------------------------------------------------ dense helpers */
struct Mat {
    int r = 0, c = 0;
    std::vector<double> a; /* row-major */
    Mat() {}
    Mat(int r_, int c_, double v = 0.0) : r(r_), c(c_), a((size_t)r_ * c_, v) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

Mat eye(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1.0;
    return m;
}
Mat mul(const Mat &A, const Mat &B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; ++i)
        for (int j = 0; j < B.c; ++j) {
            double s = 0.0;
            for (int k = 0; k < A.c; ++k) s += A(i, k) * B(k, j);
            C(i, j) = s;
        }
    return C;
}
Mat tr(const Mat &A) {
    Mat T(A.c, A.r);
    for (int i = 0; i < A.r; ++i)
        for (int j = 0; j < A.c; ++j) T(j, i) = A(i, j);
    return T;
}
Mat add(const Mat &A, const Mat &B) {
    Mat C(A.r, A.c);
    for (size_t i = 0; i < C.a.size(); ++i) C.a[i] = A.a[i] + B.a[i];
    return C;
}
Mat sub(const Mat &A, const Mat &B) {
    Mat C(A.r, A.c);
    for (size_t i = 0; i < C.a.size(); ++i) C.a[i] = A.a[i] - B.a[i];
    return C;
}
bool all_finite(const Mat &A) {
    for (double v : A.a)
        if (!std::isfinite(v)) return false;
    return true;
}

/* LU with partial pivoting, in place; piv[k] = row swapped into k. false on a zero pivot. */
bool lu_factor(Mat &A, std::vector<int> &piv) {
    const int n = A.r;
    piv.assign(n, 0);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); p = i; }
        piv[k] = p;
        if (best == 0.0) return false;
        if (p != k)
            for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j));
        for (int i = k + 1; i < n; ++i) {
            A(i, k) /= A(k, k);
            const double l = A(i, k);
            for (int j = k + 1; j < n; ++j) A(i, j) -= l * A(k, j);
        }
    }
    return true;
}
void lu_solve(const Mat &LU, const std::vector<int> &piv, std::vector<double> &b) {
    const int n = LU.r;
    for (int k = 0; k < n; ++k)
        if (piv[k] != k) std::swap(b[k], b[piv[k]]);
    for (int i = 1; i < n; ++i)
        for (int j = 0; j < i; ++j) b[i] -= LU(i, j) * b[j];
    for (int i = n - 1; i >= 0; --i) {
        for (int j = i + 1; j < n; ++j) b[i] -= LU(i, j) * b[j];
        b[i] /= LU(i, i);
    }
}

/* arma::inv contract */
bool inv_lu(const Mat &A, Mat &out) {
    const int n = A.r;
    out = Mat(n, n);
    if (n == 0) return true;
    if (!all_finite(A)) { /* NaN in, NaN out */
        for (double &v : out.a) v = NAN;
        return true;
    }
    Mat LU = A;
    std::vector<int> piv;
    if (!lu_factor(LU, piv)) return false;
    std::vector<double> col(n);
    for (int j = 0; j < n; ++j) {
        std::fill(col.begin(), col.end(), 0.0);
        col[j] = 1.0;
        lu_solve(LU, piv, col);
        for (int i = 0; i < n; ++i) out(i, j) = col[i];
    }
    return true;
}

/* one-sided Jacobi SVD of a square matrix: A = U diag(s) V^T */
bool jacobi_svd(const Mat &A, Mat &U, std::vector<double> &s, Mat &V) {
    const int n = A.r;
    if (!all_finite(A)) return false;
    U = A;
    V = eye(n);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < n; ++i) {
                    alpha += U(i, p) * U(i, p);
                    beta += U(i, q) * U(i, q);
                    gamma += U(i, p) * U(i, q);
                }
                if (gamma == 0.0) continue;
                const double lim = EPS * std::sqrt(alpha * beta);
                if (std::fabs(gamma) <= lim) continue;
                off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; ++i) {
                    const double up = U(i, p), uq = U(i, q);
                    U(i, p) = cs * up - sn * uq;
                    U(i, q) = sn * up + cs * uq;
                    const double vp = V(i, p), vq = V(i, q);
                    V(i, p) = cs * vp - sn * vq;
                    V(i, q) = sn * vp + cs * vq;
                }
            }
        if (off == 0.0) break;
    }
    s.assign(n, 0.0);
    for (int j = 0; j < n; ++j) {
        double nrm = 0;
        for (int i = 0; i < n; ++i) nrm += U(i, j) * U(i, j);
        nrm = std::sqrt(nrm);
        s[j] = nrm;
        if (nrm > 0)
            for (int i = 0; i < n; ++i) U(i, j) /= nrm;
    }
    return true;
}

/* arma::pinv contract (default tolerance) */
bool pinv_svd(const Mat &A, Mat &out) {
    const int n = A.r;
    out = Mat(n, n);
    if (n == 0) return true;
    Mat U, V;
    std::vector<double> s;
    if (!jacobi_svd(A, U, s, V)) return false;
    double smax = 0;
    for (double v : s) smax = std::max(smax, v);
    const double tol = (double)n * smax * EPS;
    for (int k = 0; k < n; ++k) {
        if (!(s[k] > tol)) continue; /* also drops an all-zero matrix */
        const double is = 1.0 / s[k];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) out(i, j) += V(i, k) * is * U(j, k);
    }
    return true;
}

/* arma::solve(A, b, solve_opts::equilibrate) contract for a square system, one rhs */
bool solve_equil(const Mat &A, const std::vector<double> &b, std::vector<double> &x) {
    const int n = A.r;
    x.assign(n, 0.0);
    if (n == 0) return true;
    bool finite = all_finite(A);
    for (double v : b) finite = finite && std::isfinite(v);
    if (!finite) { /* NaN in, NaN out */
        x.assign(n, NAN);
        return true;
    }
    /* dgeequ / dlaqge */
    const double smlnum = std::numeric_limits<double>::min() / EPS, bignum = 1.0 / smlnum;
    std::vector<double> R(n, 1.0), C(n, 1.0);
    double rmin = bignum, rmax = 0, amax = 0;
    bool ok = true;
    for (int i = 0; i < n; ++i) {
        double m = 0;
        for (int j = 0; j < n; ++j) m = std::max(m, std::fabs(A(i, j)));
        R[i] = m;
        rmax = std::max(rmax, m);
        rmin = std::min(rmin, m);
        amax = std::max(amax, m);
    }
    if (rmin == 0.0) ok = false; /* a zero row: singular */
    Mat As = A;
    std::vector<double> bs = b;
    bool rowsc = false, colsc = false;
    if (ok) {
        for (int i = 0; i < n; ++i) R[i] = 1.0 / std::min(std::max(R[i], smlnum), bignum);
        const double rowcnd = std::max(rmin, smlnum) / std::min(rmax, bignum);
        double cmin = bignum, cmax = 0;
        for (int j = 0; j < n; ++j) {
            double m = 0;
            for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(A(i, j)) * R[i]);
            C[j] = m;
            cmax = std::max(cmax, m);
            cmin = std::min(cmin, m);
        }
        if (cmin == 0.0) ok = false;
        if (ok) {
            for (int j = 0; j < n; ++j) C[j] = 1.0 / std::min(std::max(C[j], smlnum), bignum);
            const double colcnd = std::max(cmin, smlnum) / std::min(cmax, bignum);
            rowsc = (rowcnd < 0.1) || amax < smlnum || amax > bignum;
            colsc = (colcnd < 0.1);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    As(i, j) = A(i, j) * (rowsc ? R[i] : 1.0) * (colsc ? C[j] : 1.0);
            if (rowsc)
                for (int i = 0; i < n; ++i) bs[i] = b[i] * R[i];
        }
    }
    double rcond = 0.0;
    Mat LU = As;
    std::vector<int> piv;
    if (ok && lu_factor(LU, piv)) {
        /* exact 1-norm condition number of the equilibrated matrix */
        Mat Ai;
        inv_lu(As, Ai);
        double na = 0, ni = 0;
        for (int j = 0; j < n; ++j) {
            double sa = 0, si = 0;
            for (int i = 0; i < n; ++i) { sa += std::fabs(As(i, j)); si += std::fabs(Ai(i, j)); }
            na = std::max(na, sa);
            ni = std::max(ni, si);
        }
        rcond = 1.0 / (na * ni);
        if (!(rcond >= EPS)) ok = false;
    } else {
        ok = false;
    }
    if (ok) {
        std::vector<double> y = bs;
        lu_solve(LU, piv, y);
        /* one refinement step (dgerfs) */
        std::vector<double> res(n);
        for (int i = 0; i < n; ++i) {
            double s = bs[i];
            for (int j = 0; j < n; ++j) s -= As(i, j) * y[j];
            res[i] = s;
        }
        lu_solve(LU, piv, res);
        for (int i = 0; i < n; ++i) y[i] += res[i];
        for (int i = 0; i < n; ++i) x[i] = colsc ? y[i] * C[i] : y[i];
        return true;
    }
    /* approximate (minimum-norm) solution */
    Mat Ap;
    if (!pinv_svd(A, Ap)) return false;
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) s += Ap(i, j) * b[j];
        x[i] = s;
    }
    return true;
}

/* ------------------------------------------------------------------ MLLocation */
struct Meas { /* RangingMeasurement, sensor_types.h:27-32 */
    double ranging, errorEstimation, bx, by, bz;
};
struct Pos3 {
    double x, y, z;
    Mat cov; /* Vector3::covarianceMatrix; empty unless estimatePosition filled it */
};

/* MLLocation.cpp:24-37 */
std::vector<double> distances(const Pos3 &p, const std::vector<Meas> &m) {
    std::vector<double> d;
    d.reserve(m.size());
    for (const Meas &r : m)
        d.push_back(std::sqrt((r.bx - p.x) * (r.bx - p.x) + (r.by - p.y) * (r.by - p.y) +
                              (r.bz - p.z) * (r.bz - p.z)));
    return d;
}

/* MLLocation.cpp:263-278 */
double ml_error(const std::vector<Meas> &m, const Pos3 &p) {
    if (m.empty()) return -1;
    std::vector<double> d = distances(p, m);
    double e = 0.0;
    for (size_t i = 0; i < m.size(); ++i) e += (d[i] - m[i].ranging) * (d[i] - m[i].ranging);
    return e;
}

/* MLLocation.cpp:153-257. false = an Armadillo call would have thrown std::runtime_error. */
bool ml_estimate(const std::vector<Meas> &m, const Pos3 &seed, Pos3 &out, int *iters_out) {
    Pos3 position = seed;
    const int n = (int)m.size();
    if (iters_out) *iters_out = 0;
    if (n < 4) { out = position; return true; } /* :158-161 */

    double cost = 1e20, newCost = 1;
    const int maxIters = 10000;
    int iter = 0;
    while ((std::fabs(cost - newCost) / cost > 1e-3) && (iter < maxIters)) { /* :168 */
        iter += 1;
        cost = newCost;
        std::vector<double> d = distances(position, m);
        double g[3] = {0, 0, 0};
        Mat Hs(3, 3);
        for (int i = 0; i < n; ++i) {
            const Meas &r = m[i];
            const double dx = r.bx - position.x, dy = r.by - position.y, dz = r.bz - position.z;
            g[0] += (r.ranging - d[i]) * dx / (d[i] * r.errorEstimation);
            g[1] += (r.ranging - d[i]) * dy / (d[i] * r.errorEstimation);
            g[2] += (r.ranging - d[i]) * dz / (d[i] * r.errorEstimation);
            const double d3 = d[i] * d[i] * d[i];
            Hs(0, 0) += (1 - r.ranging / d[i] + r.ranging * dx * dx / d3) / r.errorEstimation;
            Hs(1, 1) += (1 - r.ranging / d[i] + r.ranging * dy * dy / d3) / r.errorEstimation;
            Hs(2, 2) += (1 - r.ranging / d[i] + r.ranging * dz * dz / d3) / r.errorEstimation;
            const double dxy = r.ranging * dx * dy / (d3 * r.errorEstimation);
            const double dxz = r.ranging * dx * dz / (d3 * r.errorEstimation);
            const double dyz = r.ranging * dy * dz / (d3 * r.errorEstimation);
            Hs(0, 1) += dxy; Hs(0, 2) += dxz; Hs(1, 2) += dyz;
            Hs(1, 0) += dxy; Hs(2, 0) += dxz; Hs(2, 1) += dyz;
        }
        const double pos[3] = {position.x, position.y, position.z};
        std::vector<double> rhs(3), np;
        for (int i = 0; i < 3; ++i)
            rhs[i] = Hs(i, 0) * pos[0] + Hs(i, 1) * pos[1] + Hs(i, 2) * pos[2] - g[i]; /* :209 */
        if (!solve_equil(Hs, rhs, np)) return false; /* :210 */
        position.x = np[0]; position.y = np[1]; position.z = np[2];
        d = distances(position, m);
        newCost = 0.0;
        for (int i = 0; i < n; ++i)
            newCost += (m[i].ranging - d[i]) * (m[i].ranging - d[i]) / m[i].errorEstimation;
    }
    if (iters_out) *iters_out = iter;

    std::vector<double> d = distances(position, m);
    Mat J(n, 3);
    std::vector<double> obs(n);
    const double rangingError = ml_error(m, position); /* :237 */
    for (int i = 0; i < n; ++i) {
        J(i, 0) = (position.x - m[i].bx) / d[i];
        J(i, 1) = (position.y - m[i].by) / d[i];
        J(i, 2) = (position.z - m[i].bz) / d[i];
        obs[i] = stdmax(m[i].errorEstimation, rangingError); /* :248 */
    }
    Mat D(n, n), Di;
    for (int i = 0; i < n; ++i) D(i, i) = obs[i];
    if (!inv_lu(D, Di)) return false;
    Mat JtWJ = mul(mul(tr(J), Di), J), C;
    if (!inv_lu(JtWJ, C)) return false; /* :252 */
    position.cov = C;
    out = position;
    return true;
}

/* Top-N composition (BASELINE config 5): residual^2 at the ML position of all
 * ranges (bestRangingsByDistance, MLLocation.cpp:284-300), ascending sort, drop the last
 * min(n-4, N) (estimatePositionIgnoreN, MLLocation.cpp:325-339). Kept ranges stay in
 * their original order. */
void topn_keep(const std::vector<Meas> &m, const Pos3 &seed, int topN, std::vector<int> &keep) {
    const int n = (int)m.size();
    keep.assign(n, 1);
    Pos3 p;
    if (!ml_estimate(m, seed, p, nullptr)) return;
    std::vector<double> d = distances(p, m);
    std::vector<std::pair<double, int>> q(n);
    for (int i = 0; i < n; ++i) q[i] = {(d[i] - m[i].ranging) * (d[i] - m[i].ranging), i};
    std::sort(q.begin(), q.end(),
              [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
    const int drop = std::min(n - 4, topN);
    for (int k = 0; k < drop; ++k) keep[q[n - 1 - k].second] = 0;
}

/* ------------------------------------------------------------------ per-tag filter members */
/* sensor_types.h:32-60: the latched samples of the other three sensors */
struct Px4Meas { double vx, vy, gyroz, integrationTime, covVel, covGyroZ; };
struct PlanarImuMeas { double angVelZ, covAngVelZ, ax, ay, covXY[4]; };
struct MagMeas { double angle, cov; };
struct Tag {
    double pos[3], vel[3];          /* mPosition.{x,y,z}, mVelocity (mAcceleration is always 0) */
    double P[81];                   /* estimationCovariance, n*n row-major */
    bool started;                   /* mLastKFTimestamp != time_point::min() */
    bool hasImu;                    /* mHasImuMeasurement */
    double imuAcc[3], imuCov[9];    /* lastImuMeasurement */
    double ang, angSpeed;           /* KalmanFilter: mAngle, mAngularSpeed; pos[2] doubles as mUWBtagZ */
    bool hasPx4, hasPImu, hasMag;   /* mHasPX4FlowMeasurement, mHasImuMeasurement, mHasMagMeasurement */
    Px4Meas px4;                    /* lastPX4FlowMeasurement */
    PlanarImuMeas pimu;             /* lastImuMeasurement */
    MagMeas mag;                    /* lastMagMeasurement */
};

struct Params {
    int model, n, topN;
    int mlVariant = 0; /* ALGORITHM_ML: ML_VARIANT_NORMAL 0 / IGNORE_N 1 / BEST 2 (MLLocation.h:5-7) */
    double accelNoise, jolt, costThreshold;
    bool ignoreWorst, useFixedInit;
    /* 8-state planar filter (KalmanFilter): <uwb useFixedHeight fixedHeight/> of config_uwb.xml, initAngle */
    bool useFixedHeight = false;
    double fixedHeight = 0.0, initAngle = 0.0;
    /* config_px4flow.xml / config_imu.xml / config_mag.xml, KalmanFilter.cpp:765-842 */
    double px4Height = 0, px4ArmP1 = 0, px4ArmP2 = 0, px4CovVel = 0, px4CovGyroZ = 0;
    bool imuFixedCovAcc = false, imuFixedCovAngVelZ = false;
    double imuCovAcc = 0, imuCovAngVelZ = 0, magAngleOffset = 0, magCov = 0;
};

/* predictionMatrix: KalmanFilterTOA.cpp:362-369 / KalmanFilterTOAIMU.cpp:392-402 */
Mat pred_F(int n, double t) {
    Mat F = eye(n);
    for (int k = 0; k < 3; ++k) {
        F(k, k + 3) = t;
        if (n == 9) { F(k, k + 6) = t * t / 2; F(k + 3, k + 6) = t; }
    }
    return F;
}
/* predictionErrorCovariance: KalmanFilterTOA.cpp:371-391 / KalmanFilterTOAIMU.cpp:405-421 */
Mat pred_Q(const Params &pr, double timeLag) {
    Mat Q(pr.n, pr.n);
    if (pr.n == 6) {
        const double t2 = std::pow(timeLag, 2) / 2, t = timeLag;
        const double a2 = pr.accelNoise * pr.accelNoise;
        for (int k = 0; k < 3; ++k) {
            Q(k, k) = a2 * t2 * t2;
            Q(k, k + 3) = a2 * t2 * t;
            Q(k + 3, k) = a2 * t2 * t;
            Q(k + 3, k + 3) = a2 * t * t;
        }
    } else {
        const double t3 = std::pow(timeLag, 3) / 6, t2 = std::pow(timeLag, 2) / 2, t = timeLag;
        const double j = pr.jolt;
        const double u[3] = {t3, t2, t};
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    /* the reference writes j*t3*t3, j*t3*t2, ... : (j*u_small)*u_other, left to right */
                    const double ua = u[std::min(a, b)], ub = u[std::max(a, b)];
                    Q(k + 3 * a, k + 3 * b) = j * ua * ub;
                }
    }
    return Q;
}

Mat P_of(const Tag &tg, int n) {
    Mat P(n, n);
    std::memcpy(P.a.data(), tg.P, sizeof(double) * n * n);
    return P;
}
void P_to(Tag &tg, const Mat &P) { std::memcpy(tg.P, P.a.data(), sizeof(double) * P.r * P.c); }

struct StepOut {
    std::vector<double> state;
    Mat P;
    double cost;
    int gainIters, mlIters;
    unsigned flags;
};

/* kalmanStep3DIgnoreAnchor (KalmanFilterTOA.cpp:242-338) and kalmanStep3D
 * (KalmanFilterTOAIMU.cpp:242-340, repaired) share this body.
 * Pm is the member covariance, already predicted. false = std::runtime_error. */
bool iekf_step(const Params &pr, const std::vector<double> &predictedState, const Mat &Pm,
               bool hasRanging, const std::vector<Meas> &all, int indexIgnoredAnchor,
               bool hasImu, const double *imuAcc, const double *imuCov,
               int maxSteps, double minRelativeError, StepOut &out) {
    const int n = pr.n;
    out.flags = 0;
    out.gainIters = 0;
    out.mlIters = 0;
    std::vector<Meas> rm;
    if (hasRanging)
        for (int i = 0; i < (int)all.size(); ++i)
            if (i != indexIgnoredAnchor) rm.push_back(all[i]);

    std::vector<double> newState(predictedState);
    int countValid = hasRanging ? (int)rm.size() : 0;
    int indexImu = 0;
    if (hasImu) { indexImu = countValid; countValid += 3; /* repaired: was += 9 */ }

    Mat R = eye(countValid);
    std::vector<double> z(countValid, 0.0);

    if (hasRanging) {
        Pos3 seed{newState[0], newState[1], newState[2], Mat()};
        Pos3 ml;
        if (!ml_estimate(rm, seed, ml, &out.mlIters)) return false;
        if ((int)rm.size() < 4) out.flags |= KFO_ST_FEW_RANGES;
        if (n == 6 && (std::isnan(ml.x) || std::isnan(ml.y) || std::isnan(ml.z))) { /* TOA.cpp:270-272 */
            ml = seed;
            out.flags |= KFO_ST_ML_FALLBACK;
        }
        const double mlRangingError = ml_error(rm, ml);
        for (int i = 0; i < (int)rm.size(); ++i) {
            z[i] = rm[i].ranging;
            R(i, i) = stdmax(mlRangingError, rm[i].errorEstimation); /* TOA.cpp:281 */
        }
    }
    if (hasImu) { /* TOAIMU.cpp:278-294 */
        for (int k = 0; k < 3; ++k) {
            z[indexImu + k] = imuAcc[k];
            for (int l = 0; l < 3; ++l) R(indexImu + k, indexImu + l) = imuCov[3 * k + l];
        }
    }

    Mat H(countValid, n), K(n, countValid);
    Mat Ri, Pp;
    if (!inv_lu(R, Ri)) return false;   /* TOA.cpp:289 */
    if (!pinv_svd(Pm, Pp)) return false; /* TOA.cpp:290 */

    double cost = 1e20;
    for (int iter = 0; iter < maxSteps; ++iter) {
        Pos3 cur{newState[0], newState[1], newState[2], Mat()};
        std::vector<double> h(countValid, 0.0);
        if (hasRanging) {
            std::vector<double> d = distances(cur, rm);
            for (size_t i = 0; i < rm.size(); ++i) h[i] = d[i];
        }
        if (hasImu)
            for (int k = 0; k < 3; ++k) h[indexImu + k] = newState[6 + k];
        Mat y(countValid, 1), dlt(n, 1);
        for (int i = 0; i < countValid; ++i) y(i, 0) = z[i] - h[i];
        for (int i = 0; i < n; ++i) dlt(i, 0) = predictedState[i] - newState[i];
        const double newCost = add(mul(mul(tr(y), Ri), y), mul(mul(tr(dlt), Pp), dlt))(0, 0); /* :302-305 */
        if (std::fabs(cost - newCost) / cost < minRelativeError) break;                      /* :307 */
        cost = newCost;

        if (hasRanging) { /* jacobianRangings, TOA.cpp:421-433 */
            std::vector<double> d = distances(cur, rm);
            for (size_t i = 0; i < rm.size(); ++i) {
                H((int)i, 0) = (cur.x - rm[i].bx) / d[i];
                H((int)i, 1) = (cur.y - rm[i].by) / d[i];
                H((int)i, 2) = (cur.z - rm[i].bz) / d[i];
                for (int c = 3; c < n; ++c) H((int)i, c) = 0;
            }
        }
        if (hasImu) { /* jacobianImu, TOAIMU.cpp:441-473 (column 9 -> 8 repaired) */
            for (int k = 0; k < 3; ++k) {
                for (int c = 0; c < n; ++c) H(indexImu + k, c) = 0;
                H(indexImu + k, 6 + k) = newState[6 + k]; /* sic: the state value, not 1 */
            }
        }
        Mat S = add(mul(mul(H, Pm), tr(H)), R), Si;
        if (!inv_lu(S, Si)) return false;
        K = mul(mul(Pm, tr(H)), Si); /* :316-317 */
        Mat dir = add(dlt, mul(K, sub(y, mul(H, dlt)))); /* :319 */
        for (int i = 0; i < n; ++i) newState[i] = newState[i] + dir(i, 0);
        out.gainIters++;
    }
    out.P = mul(sub(eye(n), mul(K, H)), Pm); /* :326 */
    out.state = newState;
    out.cost = cost;
    return true;
}

/* kalmanStep3DCanIgnoreAnAnchor, KalmanFilterTOA.cpp:185-238 */
bool toa6_can_ignore(const Params &pr, const std::vector<double> &predictedState, const Mat &Pm,
                     const std::vector<Meas> &all, double costThreshold, StepOut &result, int *ignored) {
    double worstIgnoredCost = 0, maxDistance = 0;
    StepOut best, withAll;
    int ignoredAnchorIndex = -1;
    if (!iekf_step(pr, predictedState, Pm, true, all, -1, false, nullptr, nullptr, 10, 1e-3, withAll))
        return false;
    for (int i = 0; i < (int)all.size(); ++i) {
        StepOut s;
        if (!iekf_step(pr, predictedState, Pm, true, all, i, false, nullptr, nullptr, 10, 1e-3, s))
            return false;
        const double x = s.state[0], y = s.state[1], zz = s.state[2];
        const double dist = std::sqrt(std::pow(all[i].bx - x, 2) + std::pow(all[i].by - y, 2) +
                                      std::pow(all[i].bz - zz, 2));
        const double diff = all[i].ranging - dist;
        if (i == 0 || diff > maxDistance) {
            maxDistance = diff;
            worstIgnoredCost = s.cost;
            best = s;
            ignoredAnchorIndex = i;
        }
    }
    result = withAll;
    *ignored = -1;
    if (maxDistance > 0) {
        const double costDiff = withAll.cost - worstIgnoredCost;
        if (costDiff > costThreshold) {
            result.state = best.state;
            result.P = best.P;
            result.gainIters = best.gainIters;
            result.mlIters = best.mlIters;
            *ignored = ignoredAnchorIndex;
        }
    }
    return true;
}

inline unsigned pack_status(unsigned flags, int gainIters, int mlIters, int ignored) {
    return flags | ((unsigned)std::min(gainIters, 255) << 8) | ((unsigned)std::min(mlIters, 255) << 16) |
           ((unsigned)(ignored + 1) << 24);
}

/* KalmanFilterTOA::estimatePositionKF, KalmanFilterTOA.cpp:70-156 */
unsigned toa6_estimate(const Params &pr, Tag &tg, const std::vector<Meas> &allIn, double timeLag) {
    std::vector<Meas> all = allIn;
    tg.started = true;
    if (!pr.useFixedInit) {
        if (std::isnan(tg.pos[0]) || std::isnan(tg.pos[1]) || std::isnan(tg.pos[2])) {
            /* :93-106. With <4 ranges the reference indexes an empty covariance matrix
             * (Armadillo bounds error -> abort); here the tag simply stays uninitialised. */
            if ((int)all.size() < 4) return KFO_ST_FEW_RANGES;
            Pos3 seed{1.0, 1.0, 4.0, Mat()}, ml;
            int it = 0;
            if (!ml_estimate(all, seed, ml, &it)) return KFO_ST_UPDATE_SKIPPED;
            tg.pos[0] = ml.x; tg.pos[1] = ml.y; tg.pos[2] = ml.z;
            const Mat &C = ml.cov;
            double *P = tg.P; /* 6x6 row-major */
            P[0 * 6 + 0] = C(0, 0); P[1 * 6 + 0] = C(1, 0); P[2 * 6 + 0] = C(2, 0);
            P[0 * 6 + 1] = C(0, 1); P[1 * 6 + 1] = C(1, 1); P[2 * 6 + 1] = C(2, 1);
            P[0 * 6 + 2] = C(0, 1); P[1 * 6 + 2] = C(1, 1); P[2 * 6 + 2] = C(2, 1); /* sic: column 1 again */
            return pack_status(KFO_ST_ML_INIT, 0, it, -1);
        }
    }
    /* top-N composition happens on the caller side of newTOAMeasurement */
    if (pr.topN > 0 && !(std::isnan(tg.pos[0]) || std::isnan(tg.pos[1]) || std::isnan(tg.pos[2]))) {
        std::vector<int> keep;
        Pos3 seed{tg.pos[0], tg.pos[1], tg.pos[2], Mat()};
        topn_keep(all, seed, pr.topN, keep);
        std::vector<Meas> kept;
        for (size_t i = 0; i < all.size(); ++i)
            if (keep[i]) kept.push_back(all[i]);
        all.swap(kept);
    }
    const int n = 6;
    std::vector<double> cur = {tg.pos[0], tg.pos[1], tg.pos[2], 0.0, 0.0, 0.0}; /* mVelocity is never written */
    const Mat F = pred_F(n, timeLag), Q = pred_Q(pr, timeLag);
    std::vector<double> pred(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) pred[i] += F(i, j) * cur[j];
    Mat Pm = add(mul(mul(F, P_of(tg, n)), tr(F)), Q); /* :122 */
    P_to(tg, Pm);

    StepOut ns;
    int ignored = -1;
    bool ok;
    if ((int)all.size() > 4 && pr.ignoreWorst) {
        ok = toa6_can_ignore(pr, pred, Pm, all, pr.costThreshold, ns, &ignored);
    } else {
        ok = iekf_step(pr, pred, Pm, true, all, -1, false, nullptr, nullptr, 10, 1e-3, ns);
    }
    if (!ok) return KFO_ST_UPDATE_SKIPPED; /* :151-153, predicted P kept */
    P_to(tg, ns.P);
    tg.pos[0] = ns.state[0]; tg.pos[1] = ns.state[1]; tg.pos[2] = ns.state[2];
    return pack_status(ns.flags, ns.gainIters, ns.mlIters, ignored);
}

/* MLLocation as the estimator (ALGORITHM_ML): newTOAMeasurement keeps the epoch (MLLocation.cpp:472-486),
 * getPose solves it from the never-updated _previousEstimation (:421-469): variant NORMAL 3-D
 * (estimatePosition) or IGNORE_N (estimatePositionIgnoreN, :307-347, which re-solves on the ranges SORTED by
 * residual minus the worst min(n-4, N)). The solve is deterministic, so it is done here once per epoch and
 * kept in the tag: pos = estimate, P (3x3) = its covariance. The 2-D variant is not restated: getPose indexes (0,2)
 * of the 2x2 covariance of estimatePosition2D. BEST (estimatePositionBestGroup, :348-414) is restated where the
 * reference defines it -- 4 or 5 ranges: its erase loop (:377-381) removes by an index into the vector it is shrinking,
 * so with two or more ranges to drop (6 ranges and more) the second erase hits end() or beyond -- undefined behaviour
 * from the very first group (v = 1111 0..0 ends in a false at index n-1). With 4 ranges there is one group (all of
 * them), with 5 there are five, each missing exactly one range, in std::prev_permutation order: without range 4, 3, 2,
 * 1, 0. The group with the smallest trace of its covariance wins; `<=` (:407) lets a later group win a tie, and a NaN
 * trace never wins unless it is the first. */
unsigned ml_estimator(const Params &pr, Tag &tg, const std::vector<Meas> &all, const double seed3[3]) {
    tg.started = true;
    Pos3 seed{seed3[0], seed3[1], seed3[2], Mat()}, est;
    int it = 0;
    if ((int)all.size() < 4) { /* the seed comes back with an empty covariance: getPose would index it (abort) */
        for (int k = 0; k < 3; ++k) tg.pos[k] = seed3[k];
        for (int k = 0; k < 9; ++k) tg.P[k] = NAN;
        return KFO_ST_FEW_RANGES;
    }
    if (!ml_estimate(all, seed, est, &it)) return KFO_ST_UPDATE_SKIPPED; /* (BEST: :357-366, then thrown away) */
    if (pr.mlVariant == 2) {
        const int n = (int)all.size();
        if (n > 5) return KFO_ST_UPDATE_SKIPPED; /* undefined in the reference (erase(end()), :379-381) */
        std::vector<Pos3> groups;
        std::vector<int> left_out, passes;
        std::vector<bool> v(n);
        std::fill(v.begin(), v.begin() + 4, true);
        do {
            std::vector<Meas> nr(all);
            int out = -1;
            for (int i = 0; i < n; ++i) {
                if (!v[i]) {
                    nr.erase(nr.begin() + i); /* at most one erase for n <= 5: the index is still the original one */
                    out = i;
                }
            }
            Pos3 g;
            int git = 0;
            if (!ml_estimate(nr, seed, g, &git)) return KFO_ST_UPDATE_SKIPPED; /* inv() throws out of getPose */
            groups.push_back(g);
            left_out.push_back(out);
            passes.push_back(git);
        } while (std::prev_permutation(v.begin(), v.end()));
        double minError = 0;
        int minIndex = -1;
        for (int i = 0; i < (int)groups.size(); ++i) { /* :398-411 */
            const double currentError = groups[i].cov(0, 0) + groups[i].cov(1, 1) + groups[i].cov(2, 2);
            if (minIndex == -1) { minIndex = i; minError = currentError; }
            if (currentError <= minError) { minIndex = i; minError = currentError; }
        }
        const Pos3 &b = groups[minIndex];
        tg.pos[0] = b.x; tg.pos[1] = b.y; tg.pos[2] = b.z;
        for (int k = 0; k < 9; ++k) tg.P[k] = b.cov.a[k];
        int most = 0;
        for (int q : passes) most = std::max(most, q);
        /* gain-iteration byte: the most Gauss-Newton passes any group took; ML byte: the winner's; ignored: the range
         * (index among this epoch's > 0 ranges) the winning group did without */
        return pack_status(0, most, passes[minIndex], left_out[minIndex]);
    }
    int itRank = 0; /* Gauss-Newton passes of the ranking solve, reported in the gain-iteration byte */
    if (pr.topN > 0) {
        itRank = (int)all.size() > 4 ? it : 0; /* with exactly 4 ranges nothing is ranked away (the re-solve on the
                                                  re-ordered ranges below still runs, as in the reference) */
        std::vector<double> d = distances(est, all);
        std::vector<std::pair<double, int>> q(all.size());
        for (size_t i = 0; i < all.size(); ++i) q[i] = {(d[i] - all[i].ranging) * (d[i] - all[i].ranging), (int)i};
        std::sort(q.begin(), q.end(),
                  [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
        std::vector<Meas> kept;
        for (auto &e : q) kept.push_back(all[e.second]); /* reordered, as the reference does (:327-331) */
        const int drop = std::min((int)all.size() - 4, pr.topN);
        for (int k = 0; k < drop; ++k) kept.pop_back();
        if (!ml_estimate(kept, seed, est, &it)) return KFO_ST_UPDATE_SKIPPED;
    }
    tg.pos[0] = est.x; tg.pos[1] = est.y; tg.pos[2] = est.z;
    for (int k = 0; k < 9; ++k) tg.P[k] = est.cov.a[k];
    return pack_status(0, itRank, it, -1);
}

/* KalmanFilterTOAIMU::estimatePositionKF, KalmanFilterTOAIMU.cpp:100-195 */
unsigned imu9_estimate(const Params &pr, Tag &tg, bool hasRanging, const std::vector<Meas> &all,
                       bool hasImu, double timeLag) {
    tg.started = true;
    const int n = 9;
    if (!pr.useFixedInit) {
        if (std::isnan(tg.pos[0]) || std::isnan(tg.pos[1])) { /* :121-122, z is not tested */
            if (hasRanging) {
                if ((int)all.size() < 4) return KFO_ST_FEW_RANGES; /* reference: out-of-bounds abort */
                Pos3 seed{1.0, 1.0, 4.0, Mat()}, ml;
                int it = 0;
                if (!ml_estimate(all, seed, ml, &it)) return KFO_ST_UPDATE_SKIPPED;
                tg.pos[0] = ml.x; tg.pos[1] = ml.y; tg.pos[2] = ml.z;
                const Mat &C = ml.cov;
                tg.P[0 * 9 + 0] = C(0, 0); tg.P[1 * 9 + 0] = C(1, 0);
                tg.P[0 * 9 + 1] = C(0, 1); tg.P[1 * 9 + 1] = C(1, 1); /* :134-137, xy block only */
                return pack_status(KFO_ST_ML_INIT, 0, it, -1);
            }
            return 0;
        }
    }
    std::vector<double> st = {tg.pos[0], tg.pos[1], tg.pos[2], tg.vel[0], tg.vel[1], tg.vel[2], 0.0, 0.0, 0.0};
    const Mat F = pred_F(n, timeLag), Q = pred_Q(pr, timeLag);
    std::vector<double> pred(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) pred[i] += F(i, j) * st[j];
    Mat Pm = add(mul(mul(F, P_of(tg, n)), tr(F)), Q); /* :179 */
    P_to(tg, Pm);

    StepOut ns;
    /* no try/catch in the reference: a throwing inverse aborts the node. Here: update skipped. */
    if (!iekf_step(pr, pred, Pm, hasRanging, all, -1, hasImu, tg.imuAcc, tg.imuCov, 20, 1e-4, ns))
        return KFO_ST_UPDATE_SKIPPED;
    P_to(tg, ns.P); /* :338 */
    for (int k = 0; k < 3; ++k) { tg.vel[k] = ns.state[3 + k]; tg.pos[k] = ns.state[k]; } /* :189-194 */
    return pack_status(ns.flags, ns.gainIters, ns.mlIters, -1);
}

/* ------------------------------------------------------------------ 8-state planar filter, ranging rows */
/* MLLocation::estimatePosition2D, MLLocation.cpp:48-143. REPAIR: `Vector3 tentativePos;` (:64) is
 * never given a z before its distances are taken (:105-106) -- undefined behaviour in the reference; the
 * evident intent, z = the fixed height of `position`, is what is restated. */
bool ml_estimate_2d(const std::vector<Meas> &m, const Pos3 &seed, Pos3 &out, int *iters_out) {
    Pos3 position = seed;
    const int n = (int)m.size();
    if (iters_out) *iters_out = 0;
    if (n < 3) { out = position; return true; } /* :54-58 */
    double cost = 1e20, newCost = 1, step = 1;
    newCost = ml_error(m, position); /* :65 */
    int iter = 0;
    while ((std::fabs(cost - newCost) / cost > 1e-3) && (iter < 10000)) {
        iter += 1;
        cost = newCost;
        std::vector<double> d = distances(position, m);
        double g[2] = {0, 0};
        Mat H2(2, 2);
        for (int i = 0; i < n; ++i) {
            const Meas &r = m[i];
            const double dx = r.bx - position.x, dy = r.by - position.y;
            g[0] += (r.ranging - d[i]) * dx / (d[i] * r.errorEstimation);
            g[1] += (r.ranging - d[i]) * dy / (d[i] * r.errorEstimation);
            const double d3 = d[i] * d[i] * d[i];
            H2(0, 0) += (1 - r.ranging / d[i] + r.ranging * dx * dx / d3) / r.errorEstimation;
            H2(1, 1) += (1 - r.ranging / d[i] + r.ranging * dy * dy / d3) / r.errorEstimation;
            const double dxy = r.ranging * dx * dy / (d3 * r.errorEstimation);
            H2(0, 1) += dxy;
            H2(1, 0) += dxy;
        }
        std::vector<double> rhs(2), np;
        rhs[0] = H2(0, 0) * position.x + H2(0, 1) * position.y - g[0] * step; /* :99 */
        rhs[1] = H2(1, 0) * position.x + H2(1, 1) * position.y - g[1] * step;
        if (!solve_equil(H2, rhs, np)) return false; /* arma::solve(A, b), :100 */
        Pos3 tentative = position; /* REPAIR: z */
        tentative.x = np[0];
        tentative.y = np[1];
        const double tentativeCost = ml_error(m, tentative);
        if (tentativeCost > cost) {
            step /= 2;
        } else {
            newCost = tentativeCost;
            step = 1;
            position.x = np[0];
            position.y = np[1];
        }
    }
    if (iters_out) *iters_out = iter;
    std::vector<double> d = distances(position, m);
    Mat J(n, 2), D(n, n), Di;
    const double rangingError = ml_error(m, position);
    for (int i = 0; i < n; ++i) {
        J(i, 0) = (position.x - m[i].bx) / d[i];
        J(i, 1) = (position.y - m[i].by) / d[i];
        D(i, i) = stdmax(m[i].errorEstimation, rangingError); /* :136 */
    }
    if (!inv_lu(D, Di)) return false;
    Mat C;
    if (!inv_lu(mul(mul(tr(J), Di), J), C)) return false; /* :139 */
    position.cov = C;
    out = position;
    return true;
}

/* predictionMatrix / predictionErrorCovariance, KalmanFilter.cpp:583-609 */
Mat planar_F(double t) {
    Mat F = eye(8);
    F(0, 2) = t; F(0, 4) = t * t / 2;
    F(1, 3) = t; F(1, 5) = t * t / 2;
    F(2, 4) = t; F(3, 5) = t;
    F(6, 7) = t;
    return F;
}
Mat planar_Q(const Params &pr, double timeLag) {
    const double t3 = std::pow(timeLag, 3) / 6, t2 = std::pow(timeLag, 2) / 2, t = timeLag;
    const double a = pr.accelNoise, j = pr.jolt; /* sic: accelNoise is NOT squared here (:598, :606-607) */
    const double u[3] = {t3, t2, t};
    Mat Q(8, 8);
    for (int k = 0; k < 2; ++k)
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) Q(k + 2 * p, k + 2 * q) = j * u[std::min(p, q)] * u[std::max(p, q)];
    Q(6, 6) = a * t2 * t2;
    Q(6, 7) = a * t2 * t;
    Q(7, 6) = a * t2 * t;
    Q(7, 7) = a * t * t;
    return Q;
}
double normalize_angle(double angle) { /* :699-706 */
    if (angle > M_PI) return angle - 2 * M_PI;
    if (angle <= -M_PI) return angle + 2 * M_PI;
    return angle;
}

/* PX4FLOWOutput / ImuOutput and the four Jacobian blocks, KalmanFilter.cpp:558-581, 611-697 */
struct PlanarRows { int m, nr, iPx4, iImu, iMag; };
PlanarRows planar_rows(bool hasR, int nr, bool hasPx4, bool hasImu, bool hasMag) { /* :375-399 */
    PlanarRows r{0, hasR ? nr : 0, 0, 0, 0};
    r.m = r.nr;
    if (hasPx4) { r.iPx4 = r.m; r.m += 3; }
    if (hasImu) { r.iImu = r.m; r.m += 3; }
    if (hasMag) { r.iMag = r.m; r.m += 1; }
    return r;
}

/* KalmanFilter::estimatePositionKF, KalmanFilter.cpp:224-321, and kalmanStep3D, :365-501 */
unsigned planar_estimate(const Params &pr, Tag &tg, bool hasR, const std::vector<Meas> &all, bool hasPx4,
                         const Px4Meas &px4, bool hasImu, const PlanarImuMeas &imu, bool hasMag, const MagMeas &mag,
                         double timeLag) {
    tg.started = true;
    if (!pr.useFixedInit && (std::isnan(tg.pos[0]) || std::isnan(tg.pos[1]))) { /* :243-246 */
        if (!hasR) return 0; /* :249: only a ranging epoch can initialise */
        Pos3 ml;
        int it = 0;
        if (pr.useFixedHeight) {
            if ((int)all.size() < 3) return KFO_ST_FEW_RANGES; /* reference: empty covariance indexed -> abort */
            if (!ml_estimate_2d(all, Pos3{1.0, 1.0, tg.pos[2], Mat()}, ml, &it)) return KFO_ST_UPDATE_SKIPPED;
        } else {
            if ((int)all.size() < 4) return KFO_ST_FEW_RANGES;
            if (!ml_estimate(all, Pos3{1.0, 1.0, 4.0, Mat()}, ml, &it)) return KFO_ST_UPDATE_SKIPPED;
            tg.pos[2] = ml.z; /* mUWBtagZ = mPosition.z, :257 */
        }
        tg.pos[0] = ml.x; tg.pos[1] = ml.y;
        tg.P[0 * 8 + 0] = ml.cov(0, 0); tg.P[1 * 8 + 0] = ml.cov(1, 0);
        tg.P[0 * 8 + 1] = ml.cov(0, 1); tg.P[1 * 8 + 1] = ml.cov(1, 1);
        return pack_status(KFO_ST_ML_INIT, 0, it, -1);
    }
    const int n = 8;
    std::vector<double> st = {tg.pos[0], tg.pos[1], tg.vel[0], tg.vel[1], 0.0, 0.0, tg.ang, tg.angSpeed};
    const Mat F = planar_F(timeLag), Q = planar_Q(pr, timeLag);
    std::vector<double> pred(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) pred[i] += F(i, j) * st[j];
    Mat Pm = add(mul(mul(F, P_of(tg, n)), tr(F)), Q);
    P_to(tg, Pm);
    pred[6] = normalize_angle(pred[6]); /* :301 */

    const PlanarRows rw = planar_rows(hasR, (int)all.size(), hasPx4, hasImu, hasMag);
    const int m = rw.m;
    unsigned flags = 0;
    int mlIters = 0, gains = 0;
    std::vector<double> x(pred);
    Mat R = eye(m);
    std::vector<double> z(m, 0.0);
    if (hasR) { /* :403-411 */
        Pos3 ml;
        if (!ml_estimate_2d(all, Pos3{x[0], x[1], tg.pos[2], Mat()}, ml, &mlIters)) return KFO_ST_UPDATE_SKIPPED;
        if (rw.nr < 3) flags |= KFO_ST_FEW_RANGES;
        const double e_ml = ml_error(all, ml);
        for (int i = 0; i < rw.nr; ++i) {
            z[i] = all[i].ranging;
            R(i, i) = stdmax(e_ml, all[i].errorEstimation);
        }
    }
    if (hasPx4) { /* :413-424 */
        z[rw.iPx4] = px4.vx; z[rw.iPx4 + 1] = px4.vy; z[rw.iPx4 + 2] = px4.gyroz;
        R(rw.iPx4, rw.iPx4) = px4.covVel;
        R(rw.iPx4 + 1, rw.iPx4 + 1) = px4.covVel;
        R(rw.iPx4 + 2, rw.iPx4 + 2) = px4.covGyroZ;
    }
    if (hasImu) { /* :426-436 */
        z[rw.iImu] = imu.ax; z[rw.iImu + 1] = imu.ay; z[rw.iImu + 2] = imu.angVelZ;
        R(rw.iImu, rw.iImu) = imu.covXY[0];
        R(rw.iImu, rw.iImu + 1) = imu.covXY[1];
        R(rw.iImu + 1, rw.iImu) = imu.covXY[2];
        R(rw.iImu + 1, rw.iImu + 1) = imu.covXY[3];
        R(rw.iImu + 2, rw.iImu + 2) = imu.covAngVelZ;
    }
    if (hasMag) { /* :438-442 */
        z[rw.iMag] = mag.angle;
        R(rw.iMag, rw.iMag) = mag.cov;
    }
    Mat H(m, n), K(n, m), Ri, Pp;
    if (!inv_lu(R, Ri)) return KFO_ST_UPDATE_SKIPPED;
    if (!pinv_svd(Pm, Pp)) return KFO_ST_UPDATE_SKIPPED;
    double cost = 1e20;
    for (int iter = 0; iter < 20; ++iter) {
        Pos3 cur{x[0], x[1], tg.pos[2], Mat()};
        const double vx = x[2], vy = x[3], ax = x[4], ay = x[5], th = x[6], om = x[7];
        std::vector<double> d = hasR ? distances(cur, all) : std::vector<double>();
        Mat y(m, 1), dlt(n, 1);
        for (int i = 0; i < rw.nr; ++i) y(i, 0) = z[i] - d[i];
        if (hasPx4) { /* px4flowOutput, :558-566 */
            const double p1 = pr.px4ArmP1, p2 = pr.px4ArmP2;
            const double ox = std::cos(th) * vx + std::sin(th) * vy +
                              1 / timeLag * ((1 - std::cos(om * timeLag)) * p1 - std::sin(om * timeLag) * p2);
            const double oy = -std::sin(th) * vx + std::cos(th) * vy +
                              1 / timeLag * (std::sin(om * timeLag) * p1 + (1 - std::cos(om * timeLag)) * p2);
            y(rw.iPx4, 0) = z[rw.iPx4] - ox;
            y(rw.iPx4 + 1, 0) = z[rw.iPx4 + 1] - oy;
            y(rw.iPx4 + 2, 0) = z[rw.iPx4 + 2] - om;
        }
        if (hasImu) { /* imuOutput, :568-576 */
            y(rw.iImu, 0) = z[rw.iImu] - (std::cos(th) * ax + std::sin(th) * ay);
            y(rw.iImu + 1, 0) = z[rw.iImu + 1] - (-std::sin(th) * ax + std::cos(th) * ay);
            y(rw.iImu + 2, 0) = z[rw.iImu + 2] - om;
        }
        if (hasMag) y(rw.iMag, 0) = normalize_angle(z[rw.iMag] - th); /* :460-462 */
        for (int i = 0; i < n; ++i) dlt(i, 0) = pred[i] - x[i];
        const double newCost = add(mul(mul(tr(y), Ri), y), mul(mul(tr(dlt), Pp), dlt))(0, 0);
        if (std::fabs(cost - newCost) / cost < 1e-4) break;
        cost = newCost;
        /* `jacobian` is allocated uninitialised once (:444) and every block writes all 8 columns of its rows */
        for (int i = 0; i < rw.nr; ++i) { /* jacobianRangings, :611-625 */
            for (int c = 0; c < n; ++c) H(i, c) = 0;
            H(i, 0) = (cur.x - all[i].bx) / d[i];
            H(i, 1) = (cur.y - all[i].by) / d[i];
        }
        if (hasPx4) { /* jacobianPx4flow, :627-655 */
            const int r0 = rw.iPx4;
            for (int k = 0; k < 3; ++k)
                for (int c = 0; c < n; ++c) H(r0 + k, c) = 0;
            H(r0, 2) = std::cos(th); H(r0, 3) = std::sin(th);
            H(r0, 6) = -std::sin(th) * vx + std::cos(th) * vy;
            H(r0, 7) = pr.px4ArmP1 * std::sin(om * timeLag) - pr.px4ArmP2 * std::cos(om * timeLag);
            H(r0 + 1, 2) = -std::sin(th); H(r0 + 1, 3) = std::cos(th);
            H(r0 + 1, 6) = -std::cos(th) * vx - std::sin(th) * vy;
            H(r0 + 1, 7) = pr.px4ArmP1 * std::cos(om * timeLag) + pr.px4ArmP2 * std::sin(om * timeLag);
            H(r0 + 2, 7) = 1;
        }
        if (hasImu) { /* jacobianImu, :657-686 */
            const int r0 = rw.iImu;
            for (int k = 0; k < 3; ++k)
                for (int c = 0; c < n; ++c) H(r0 + k, c) = 0;
            H(r0, 4) = std::cos(th); H(r0, 5) = std::sin(th);
            H(r0, 6) = -std::sin(th) * ax + std::cos(th) * ay;
            H(r0 + 1, 4) = -std::sin(th); H(r0 + 1, 5) = std::cos(th);
            H(r0 + 1, 6) = -std::cos(th) * ax - std::sin(th) * ay;
            H(r0 + 2, 7) = 1;
        }
        if (hasMag) { /* jacobianMag, :688-697 */
            for (int c = 0; c < n; ++c) H(rw.iMag, c) = 0;
            H(rw.iMag, 6) = 1;
        }
        Mat S = add(mul(mul(H, Pm), tr(H)), R), Si;
        if (!inv_lu(S, Si)) return KFO_ST_UPDATE_SKIPPED;
        K = mul(mul(Pm, tr(H)), Si);
        Mat dir = add(dlt, mul(K, sub(y, mul(H, dlt))));
        for (int i = 0; i < n; ++i) x[i] += dir(i, 0);
        gains++;
    }
    P_to(tg, mul(sub(eye(n), mul(K, H)), Pm));
    tg.pos[0] = x[0]; tg.pos[1] = x[1];
    tg.vel[0] = x[2]; tg.vel[1] = x[3];
    tg.ang = x[6]; tg.angSpeed = x[7]; /* :316-319; the acceleration state is not kept */
    return pack_status(flags, gains, mlIters, -1);
}

} // namespace

/* ------------------------------------------------------------------ C interface */
struct kfo_filter_bank {
    Params pr;
    int T, A;
    std::vector<double> anchors;
    std::vector<Tag> tags;
    std::vector<double> ml_seed; /* ALGORITHM_ML: _previousEstimation per tag */
};

namespace {
template <class Fn>
void parallel_tags(int T, int nThreads, Fn fn) {
    if (nThreads <= 1 || T < 2 * nThreads) { fn(0, T); return; }
    std::vector<std::thread> th;
    const int chunk = (T + nThreads - 1) / nThreads;
    for (int k = 0; k < nThreads; ++k) {
        const int lo = k * chunk, hi = std::min(T, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

/* newTOAMeasurement's zip of ranges > 0 (KalmanFilterTOA.cpp:48-57), fed the way
 * calculateTagLocationWithRangings does: (double) mm / 1000 (Posgenerator.cpp:483-487). */
std::vector<Meas> gather(const kfo_filter_bank *o, const int32_t *mm, const double *err) {
    std::vector<Meas> m;
    for (int a = 0; a < o->A; ++a)
        if (mm[a] > 0) {
            const double r = (double)mm[a] / 1000;
            if (r > 0) m.push_back({r, err[a], o->anchors[3 * a], o->anchors[3 * a + 1], o->anchors[3 * a + 2]});
        }
    return m;
}
unsigned finite_flag(const Tag &tg, int n) {
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(tg.pos[k]) || !std::isfinite(tg.vel[k])) return KFO_ST_NONFINITE;
    for (int i = 0; i < n * n; ++i)
        if (!std::isfinite(tg.P[i])) return KFO_ST_NONFINITE;
    return 0;
}
} // namespace


namespace {
template <class Fn> void planar_each(kfo_filter_bank *o, const double *dt, int dt_len, uint32_t *status, int n_threads, Fn fn) {
    parallel_tags(o->T, n_threads, [=](int lo, int hi) {
        for (int t = lo; t < hi; ++t) {
            Tag &tg = o->tags[t];
            const double lag = dt[dt_len > 1 ? t : 0];
            if (dt_len > 1 && lag < 0) {
                if (status) status[t] = KFO_ST_SKIPPED;
                continue;
            }
            unsigned st = fn(t, tg, lag);
            const bool uninit = !o->pr.useFixedInit && std::isnan(tg.pos[0]);
            if (!uninit && !(st & KFO_ST_SKIPPED)) st |= finite_flag(tg, 8);
            if (status) status[t] = st;
        }
    });
}
} // namespace

extern "C" {

void kfo_set_ml_variant(kfo_filter_bank *o, int variant) { o->pr.mlVariant = variant; }

kfo_filter_bank *kfo_create(int model, int n_tags, int max_anchors, double accel_noise, double jolt,
                            int ignore_worst, double cost_threshold, int top_n, int use_init_pos,
                            const double *init_pos) {
    kfo_filter_bank *o = new kfo_filter_bank();
    o->pr.model = model;
    o->pr.n = (model == KFO_MODEL_TOA_IMU) ? 9 : (model == KFO_MODEL_ML ? 3 : (model == KFO_MODEL_PLANAR ? 8 : 6));
    o->pr.topN = top_n;
    o->pr.accelNoise = accel_noise;
    o->pr.jolt = jolt;
    o->pr.costThreshold = cost_threshold;
    o->pr.ignoreWorst = ignore_worst != 0;
    o->pr.useFixedInit = use_init_pos != 0;
    o->T = n_tags;
    o->A = max_anchors;
    o->anchors.assign((size_t)3 * max_anchors, 0.0);
    o->tags.resize(n_tags);
    o->ml_seed.assign((size_t)3 * n_tags, 0.0);
    for (int t = 0; t < n_tags; ++t)
        for (int k = 0; k < 3; ++k) o->ml_seed[3 * (size_t)t + k] = (use_init_pos && init_pos) ? init_pos[3 * t + k] : 0.0;
    for (int t = 0; t < n_tags; ++t) {
        Tag &tg = o->tags[t];
        std::memset(&tg, 0, sizeof(Tag));
        for (int k = 0; k < 3; ++k)
            tg.pos[k] = (use_init_pos && init_pos) ? init_pos[3 * t + k] : (use_init_pos ? 0.0 : NAN);
    }
    return o;
}
void kfo_destroy(kfo_filter_bank *o) { delete o; }
void kfo_set_planar(kfo_filter_bank *o, const kfo_planar_config *c) {
    Params &p = o->pr;
    p.useFixedHeight = c->use_fixed_height != 0;
    p.fixedHeight = c->fixed_height;
    p.initAngle = c->init_angle;
    p.px4Height = c->px4_height; p.px4ArmP1 = c->px4_arm_p1; p.px4ArmP2 = c->px4_arm_p2;
    p.px4CovVel = c->px4_cov_velocity; p.px4CovGyroZ = c->px4_cov_gyro_z;
    p.imuFixedCovAcc = c->imu_use_fixed_cov_acc != 0; p.imuCovAcc = c->imu_cov_acc;
    p.imuFixedCovAngVelZ = c->imu_use_fixed_cov_ang_vel_z != 0; p.imuCovAngVelZ = c->imu_cov_ang_vel_z;
    p.magAngleOffset = c->mag_angle_offset; p.magCov = c->mag_cov;
    for (Tag &tg : o->tags) {
        tg.pos[2] = c->fixed_height; /* mUWBtagZ, KalmanFilter.cpp:797; the z of an initial position is not used */
        tg.ang = c->init_angle;
        tg.angSpeed = 0.0;
    }
}
void kfo_get_height(const kfo_filter_bank *o, double *z) {
    for (int t = 0; t < o->T; ++t) z[t] = o->tags[t].pos[2];
}
int kfo_state_dim(const kfo_filter_bank *o) { return o->pr.n; }
void kfo_set_anchors(kfo_filter_bank *o, const double *xyz, int n_anchors) {
    o->A = n_anchors;
    o->anchors.assign(xyz, xyz + (size_t)3 * n_anchors);
}

void kfo_step_toa(kfo_filter_bank *o, const int32_t *range_mm, const double *err_est, const double *dt,
                  int dt_len, uint32_t *status, int n_threads) {
    parallel_tags(o->T, n_threads, [=](int lo, int hi) {
        for (int t = lo; t < hi; ++t) {
            Tag &tg = o->tags[t];
            std::vector<Meas> m = gather(o, range_mm + (size_t)t * o->A, err_est + (size_t)t * o->A);
            const double lag = dt[dt_len > 1 ? t : 0];
            if (dt_len > 1 && lag < 0) { /* no epoch for this tag: the reference makes no call at all */
                if (status) status[t] = KFO_ST_SKIPPED;
                continue;
            }
            if (o->pr.model == KFO_MODEL_ML) {
                const double one14[3] = {1.0, 1.0, 4.0};
                const unsigned st = ml_estimator(o->pr, tg, m, o->pr.useFixedInit ? &o->ml_seed[3 * (size_t)t] : one14);
                if (status) status[t] = st;
                continue;
            }
            if (o->pr.model == KFO_MODEL_PLANAR) {
                /* newTOAMeasurement, KalmanFilter.cpp:66-99: the three latched samples ride along */
                unsigned st = planar_estimate(o->pr, tg, true, m, tg.hasPx4, tg.px4, tg.hasPImu, tg.pimu, tg.hasMag, tg.mag, lag);
                const bool uninit = !o->pr.useFixedInit && std::isnan(tg.pos[0]);
                if (!uninit) st |= finite_flag(tg, 8);
                if (status) status[t] = st;
                continue;
            }
            unsigned st = (o->pr.n == 6) ? toa6_estimate(o->pr, tg, m, lag)
                                         : imu9_estimate(o->pr, tg, true, m, tg.hasImu, lag);
            const bool uninit = !o->pr.useFixedInit && std::isnan(tg.pos[0]); /* still waiting for ML init */
            if (!uninit) st |= finite_flag(tg, o->pr.n);
            if (status) status[t] = st;
        }
    });
}

void kfo_step_imu(kfo_filter_bank *o, const double *accel, const double *cov, const double *dt, int dt_len,
                  uint32_t *status, int n_threads) {
    parallel_tags(o->T, n_threads, [=](int lo, int hi) {
        for (int t = lo; t < hi; ++t) {
            Tag &tg = o->tags[t];
            unsigned st = 0;
            if (dt_len > 1 && dt[t] < 0) {
                if (status) status[t] = KFO_ST_SKIPPED;
                continue;
            }
            if (o->pr.n == 9) { /* KalmanFilterTOA::newIMUMeasurement is a no-op (KalmanFilterTOA.cpp:64) */
                std::memcpy(tg.imuAcc, accel + 3 * (size_t)t, 3 * sizeof(double));
                std::memcpy(tg.imuCov, cov + 9 * (size_t)t, 9 * sizeof(double));
                tg.hasImu = true;
                st = imu9_estimate(o->pr, tg, false, std::vector<Meas>(), true, dt[dt_len > 1 ? t : 0]);
                if (o->pr.useFixedInit || !std::isnan(tg.pos[0])) st |= finite_flag(tg, 9);
            }
            if (status) status[t] = st;
        }
    });
}

/* KalmanFilter::newPX4FlowMeasurement, KalmanFilter.cpp:102-135. flow: T x 5 = integrationX, integrationY,
 * integrationRotationZ, integrationTime [us], quality. */
void kfo_planar_px4flow(kfo_filter_bank *o, const double *flow, const double *dt, int dt_len, uint32_t *status, int n_threads) {
    planar_each(o, dt, dt_len, status, n_threads, [=](int t, Tag &tg, double lag) -> unsigned {
        const double *f = flow + 5 * (size_t)t;
        const Params &pr = o->pr;
        const int quality = (int)f[4];
        Px4Meas m;
        m.vy = f[1] / (f[3] / 1000000.0) * pr.px4Height;
        m.vx = f[0] / (f[3] / 1000000.0) * pr.px4Height;
        m.gyroz = f[2] / (f[3] / 1000000.0);
        m.integrationTime = f[3] / 1000000.0;
        if (quality == 0) return KFO_ST_SKIPPED; /* :113-115: nothing latched, no estimate, no timestamp */
        if (f[3] > 0) m.covVel = pr.px4CovVel / m.integrationTime * pr.px4Height / quality;
        else m.covVel = pr.px4CovVel * quality;
        m.covGyroZ = pr.px4CovGyroZ;
        tg.px4 = m;
        tg.hasPx4 = true;
        return planar_estimate(pr, tg, false, std::vector<Meas>(), true, m, false, PlanarImuMeas(), false, MagMeas(), lag);
    });
}
/* KalmanFilter::newIMUMeasurement, KalmanFilter.cpp:139-182 */
void kfo_planar_imu(kfo_filter_bank *o, const double *ang_vel, const double *cov_ang_vel, const double *lin_acc,
                    const double *cov_acc, const double *dt, int dt_len, uint32_t *status, int n_threads) {
    planar_each(o, dt, dt_len, status, n_threads, [=](int t, Tag &tg, double lag) -> unsigned {
        const Params &pr = o->pr;
        const double *ca = cov_acc + 9 * (size_t)t;
        PlanarImuMeas m;
        m.covXY[0] = pr.imuFixedCovAcc ? pr.imuCovAcc : ca[0];
        m.covXY[1] = ca[1];
        m.covXY[2] = ca[3];
        m.covXY[3] = pr.imuFixedCovAcc ? pr.imuCovAcc : ca[4];
        m.covAngVelZ = pr.imuFixedCovAngVelZ ? pr.imuCovAngVelZ : cov_ang_vel[9 * (size_t)t + 8];
        m.angVelZ = ang_vel[3 * (size_t)t + 2];
        m.ax = lin_acc[3 * (size_t)t];
        m.ay = lin_acc[3 * (size_t)t + 1];
        tg.pimu = m;
        tg.hasPImu = true;
        return planar_estimate(pr, tg, false, std::vector<Meas>(), false, Px4Meas(), true, m, false, MagMeas(), lag);
    });
}
/* KalmanFilter::newMAGMeasurement, KalmanFilter.cpp:185-199 (the covarianceMag argument is ignored there) */
void kfo_planar_mag(kfo_filter_bank *o, const double *mag_xyz, const double *dt, int dt_len, uint32_t *status, int n_threads) {
    planar_each(o, dt, dt_len, status, n_threads, [=](int t, Tag &tg, double lag) -> unsigned {
        MagMeas m;
        m.angle = std::atan2(mag_xyz[3 * (size_t)t + 1], mag_xyz[3 * (size_t)t]) - o->pr.magAngleOffset;
        m.cov = o->pr.magCov;
        tg.mag = m;
        tg.hasMag = true;
        return planar_estimate(o->pr, tg, false, std::vector<Meas>(), false, Px4Meas(), false, PlanarImuMeas(), true, m, lag);
    });
}
/* KalmanFilter::newCompassMeasurement, KalmanFilter.cpp:201-229: the latched PX4Flow and IMU samples ride along */
void kfo_planar_compass(kfo_filter_bank *o, const double *compass, const double *dt, int dt_len, uint32_t *status, int n_threads) {
    planar_each(o, dt, dt_len, status, n_threads, [=](int t, Tag &tg, double lag) -> unsigned {
        MagMeas m;
        m.angle = normalize_angle(compass[t]);
        m.cov = o->pr.magCov;
        tg.mag = m;
        tg.hasMag = true;
        return planar_estimate(o->pr, tg, false, std::vector<Meas>(), tg.hasPx4, tg.px4, tg.hasPImu, tg.pimu, true, m, lag);
    });
}

void kfo_get_pose(const kfo_filter_bank *o, double dt_ahead, double *pos, double *cov3x3, double *vel,
                  uint32_t *status) {
    const int n = o->pr.n;
    const bool is_ml = o->pr.model == KFO_MODEL_ML;
    const bool is_planar = o->pr.model == KFO_MODEL_PLANAR;
    const Mat F = is_ml ? Mat() : (is_planar ? planar_F(dt_ahead) : pred_F(n, dt_ahead)),
              Q = is_ml ? Mat() : (is_planar ? planar_Q(o->pr, dt_ahead) : pred_Q(o->pr, dt_ahead));
    for (int t = 0; t < o->T; ++t) {
        const Tag &tg = o->tags[t];
        if (!tg.started) { /* getPose returns false: pose untouched = NaN (Posgenerator.cpp:542) */
            for (int k = 0; k < 3; ++k) { pos[3 * t + k] = NAN; if (vel) vel[3 * t + k] = NAN; }
            for (int k = 0; k < 9; ++k) cov3x3[9 * t + k] = NAN;
            if (status) status[t] = KFO_ST_NOT_STARTED;
            continue;
        }
        if (o->pr.model == KFO_MODEL_ML) { /* MLLocation::getPose: the estimate and its 3x3 covariance, no motion model */
            for (int k = 0; k < 3; ++k) { pos[3 * t + k] = tg.pos[k]; if (vel) vel[3 * t + k] = 0.0; }
            for (int k = 0; k < 9; ++k) cov3x3[9 * t + k] = tg.P[k];
            if (status) status[t] = 0;
            continue;
        }
        std::vector<double> st(n, 0.0), pred(n, 0.0);
        if (is_planar) st = {tg.pos[0], tg.pos[1], tg.vel[0], tg.vel[1], 0.0, 0.0, tg.ang, tg.angSpeed};
        else
            for (int k = 0; k < 3; ++k) { st[k] = tg.pos[k]; if (n == 9) st[3 + k] = tg.vel[k]; }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) pred[i] += F(i, j) * st[j];
        Mat Pp = add(mul(mul(F, P_of(tg, n)), tr(F)), Q);
        if (is_planar) { /* KalmanFilter::getPose + stateToPose, KalmanFilter.cpp:709-745, 324-363: position block of the 6x6 */
            pos[3 * t] = pred[0]; pos[3 * t + 1] = pred[1]; pos[3 * t + 2] = tg.pos[2];
            if (vel) { vel[3 * t] = pred[2]; vel[3 * t + 1] = pred[3]; vel[3 * t + 2] = 0.0; }
            const double c[9] = {Pp(0, 0), Pp(0, 1), 0, Pp(1, 0), Pp(1, 1), 0, 0, 0, 0.01};
            std::memcpy(cov3x3 + 9 * t, c, sizeof(c));
            if (status) status[t] = 0;
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            pos[3 * t + k] = pred[k];
            if (vel) vel[3 * t + k] = pred[3 + k];
            for (int l = 0; l < 3; ++l) cov3x3[9 * t + 3 * k + l] = Pp(k, l);
        }
        if (status) status[t] = 0;
    }
}

void kfo_get_state(const kfo_filter_bank *o, double *x, double *P) {
    const int n = o->pr.n;
    for (int t = 0; t < o->T; ++t) {
        const Tag &tg = o->tags[t];
        for (int k = 0; k < n; ++k) x[(size_t)t * n + k] = 0.0;
        if (o->pr.model == KFO_MODEL_PLANAR) {
            const double s8[8] = {tg.pos[0], tg.pos[1], tg.vel[0], tg.vel[1], 0.0, 0.0, tg.ang, tg.angSpeed};
            std::memcpy(x + (size_t)t * 8, s8, sizeof(s8));
        } else
            for (int k = 0; k < 3; ++k) { x[(size_t)t * n + k] = tg.pos[k]; if (n == 9) x[(size_t)t * n + 3 + k] = tg.vel[k]; }
        std::memcpy(P + (size_t)t * n * n, tg.P, sizeof(double) * n * n);
    }
}
void kfo_set_state(kfo_filter_bank *o, const double *x, const double *P, int started) {
    const int n = o->pr.n;
    for (int t = 0; t < o->T; ++t) {
        Tag &tg = o->tags[t];
        if (o->pr.model == KFO_MODEL_PLANAR) { /* the height (pos[2]) is kept */
            const double *s8 = x + (size_t)t * 8;
            tg.pos[0] = s8[0]; tg.pos[1] = s8[1]; tg.vel[0] = s8[2]; tg.vel[1] = s8[3]; tg.ang = s8[6]; tg.angSpeed = s8[7];
        } else
            for (int k = 0; k < 3; ++k) { tg.pos[k] = x[(size_t)t * n + k]; tg.vel[k] = (n == 9) ? x[(size_t)t * n + 3 + k] : 0.0; }
        std::memcpy(tg.P, P + (size_t)t * n * n, sizeof(double) * n * n);
        tg.started = started != 0;
    }
}

int kfo_ml_estimate(int n, const double *anchors_xyz, const double *ranges, const double *err_est,
                    const double *seed, double *pos, double *cov3x3) {
    std::vector<Meas> m;
    for (int i = 0; i < n; ++i)
        m.push_back({ranges[i], err_est[i], anchors_xyz[3 * i], anchors_xyz[3 * i + 1], anchors_xyz[3 * i + 2]});
    Pos3 s{seed[0], seed[1], seed[2], Mat()}, out;
    int it = 0;
    if (!ml_estimate(m, s, out, &it)) return -1;
    pos[0] = out.x; pos[1] = out.y; pos[2] = out.z;
    if (cov3x3)
        for (int i = 0; i < 9; ++i) cov3x3[i] = (out.cov.r == 3) ? out.cov.a[i] : NAN;
    return it;
}
void kfo_predict_matrices(int model, double dt, double accel_noise, double jolt, double *F, double *Q) {
    Params pr{};
    pr.model = model;
    pr.n = (model == KFO_MODEL_TOA_IMU) ? 9 : 6;
    pr.accelNoise = accel_noise;
    pr.jolt = jolt;
    const Mat f = pred_F(pr.n, dt), q = pred_Q(pr, dt);
    std::memcpy(F, f.a.data(), sizeof(double) * f.a.size());
    std::memcpy(Q, q.a.data(), sizeof(double) * q.a.size());
}
static Mat mat_from(int n, const double *A) {
    Mat m(n, n);
    std::memcpy(m.a.data(), A, sizeof(double) * n * n);
    return m;
}
int kfo_inv(int n, const double *A, double *out) {
    Mat r;
    if (!inv_lu(mat_from(n, A), r)) return 1;
    std::memcpy(out, r.a.data(), sizeof(double) * n * n);
    return 0;
}
int kfo_pinv(int n, const double *A, double *out) {
    Mat r;
    if (!pinv_svd(mat_from(n, A), r)) return 1;
    std::memcpy(out, r.a.data(), sizeof(double) * n * n);
    return 0;
}
int kfo_solve_equilibrate(int n, const double *A, const double *b, double *x) {
    std::vector<double> bb(b, b + n), xx;
    if (!solve_equil(mat_from(n, A), bb, xx)) return 1;
    std::memcpy(x, xx.data(), sizeof(double) * n);
    return 0;
}
void kfo_topn_keep(int n, const double *anchors_xyz, const double *ranges, const double *err_est,
                   const double *seed, int top_n, int *keep) {
    std::vector<Meas> m;
    for (int i = 0; i < n; ++i)
        m.push_back({ranges[i], err_est[i], anchors_xyz[3 * i], anchors_xyz[3 * i + 1], anchors_xyz[3 * i + 2]});
    std::vector<int> k;
    topn_keep(m, Pos3{seed[0], seed[1], seed[2], Mat()}, top_n, k);
    for (int i = 0; i < n; ++i) keep[i] = k[i];
}

} // extern "C"
