/*
 * vb_oracle.cc - CPU restatement of fabber_core's voxelwise VB loop (white noise).
 *
 * TEST INFRASTRUCTURE ONLY - see vb_oracle.h. Every function cites the reference file:line it
 * follows (paths relative to the fabber_core source tree). The code deliberately keeps the
 * reference's data flow - an MVN class with lazily inverted precision/covariance, a stored
 * Jacobian, one full model evaluation per finite-difference point - so that it is an
 * independent statement of the algorithm and not a re-run of the GPU formulation (which works
 * from streamed moments and never stores J).
 */
#include "vb_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
// ---------------------------------------------------------------------------------------------
// Exceptions mirroring the two families the voxel loop catches (inference_vb.cc:529-544)
// ---------------------------------------------------------------------------------------------
struct InternalError : std::runtime_error
{
    int code;
    InternalError(int c, const char *m)
        : std::runtime_error(m)
        , code(c)
    {
    }
};
struct SingularError : std::runtime_error
{
    SingularError()
        : std::runtime_error("matrix is singular")
    {
    }
};


// The arithmetic type of the restated algorithm. liboracle.so / liboracle_fma.so: double, as the
// reference. liboracle_quad.so (-DORACLE_QUAD, make liboracle_quad.so): IEEE binary128 through
// libquadmath, the SAME statements evaluated with 113-bit significands - the ground truth against
// which the rounding error of an fp64 build (CPU or GPU) of this algorithm is measured
// (tests/test_reference_chaos.py). Inputs, constants and outputs stay double.
#ifdef ORACLE_QUAD
#include <quadmath.h>
typedef __float128 real;
static inline real r_exp(real x) { return expq(x); }
static inline real r_log(real x) { return logq(x); }
static inline real r_sqrt(real x) { return sqrtq(x); }
static inline real r_fabs(real x) { return fabsq(x); }
static inline real r_pow(real x, real y) { return powq(x, y); }
#else
typedef double real;
#ifdef ORACLE_EXP_1ULP
// experiment (tools/measure): an exp that is good to 1 ulp instead of glibc's ~0.5 - the accuracy class of the device's
// exp - by adding a pseudo-random half ulp to the correctly rounded value
static inline real r_exp(real x)
{
    const double e = std::exp(x);
    unsigned long long b;
    memcpy(&b, &x, 8);
    b ^= b >> 29;
    b *= 0x9E3779B97F4A7C15ull;
    const int k = (int)(b >> 62); // 0..3
    return k == 0 ? std::nextafter(e, INFINITY) : (k == 1 ? std::nextafter(e, -INFINITY) : e);
}
#else
static inline real r_exp(real x) { return std::exp(x); }
#endif
static inline real r_log(real x) { return std::log(x); }
static inline real r_sqrt(real x) { return std::sqrt(x); }
static inline real r_fabs(real x) { return std::fabs(x); }
static inline real r_pow(real x, real y) { return std::pow(x, y); }
#endif

typedef std::vector<real> vec;

// Dense n x n row-major matrix
struct Mat
{
    int n;
    vec a;
    Mat(int n_ = 0)
        : n(n_)
        , a((size_t)n_ * n_, 0.0)
    {
    }
    real &operator()(int r, int c)
    {
        return a[(size_t)r * n + c];
    }
    real operator()(int r, int c) const
    {
        return a[(size_t)r * n + c];
    }
    static Mat identity(int n)
    {
        Mat m(n);
        for (int i = 0; i < n; i++)
            m(i, i) = 1.0;
        return m;
    }
};

// NEWMAT .i() as provided by armawrap -> Armadillo inv() -> LAPACK getrf/getri: LU with partial
// pivoting; an exactly zero pivot is "singular" (dist_mvn.cc:211-224 catches that case).
// Returns LU determinant info through logabs/sign when requested.
static bool lu_decompose(Mat &lu, std::vector<int> &piv, int &sign)
{
    const int n = lu.n;
    piv.resize(n);
    sign = 1;
    for (int k = 0; k < n; k++)
    {
        int p = k;
        real best = r_fabs(lu(k, k));
        for (int r = k + 1; r < n; r++)
        {
            real val = r_fabs(lu(r, k));
            if (val > best)
            {
                best = val;
                p = r;
            }
        }
        piv[k] = p;
        if (p != k)
        {
            for (int c = 0; c < n; c++)
                std::swap(lu(k, c), lu(p, c));
            sign = -sign;
        }
        if (lu(k, k) == 0.0)
            return false;
        for (int r = k + 1; r < n; r++)
        {
            lu(r, k) /= lu(k, k);
            const real f = lu(r, k);
            for (int c = k + 1; c < n; c++)
                lu(r, c) -= f * lu(k, c);
        }
    }
    return true;
}

static int g_inverse_mode = -1; // -1: not decided yet (environment), 0: LU, 1: symmetric sweep

static Mat inverse(const Mat &m)
{
    const int n = m.n;
    // Which algorithm stands in for NEWMAT's .i(). Default (0): LU with partial pivoting, the LAPACK
    // getrf / getri pair behind Armadillo's inv() that armawrap forwards a general matrix to - the
    // documented behaviour of the third-party code; it is what every parity tolerance is measured
    // against. 1 (oracle_set_inverse(1) or ORACLE_SWEEP_INVERSE in the environment): the unpivoted
    // symmetric sweep the kernels use (vb_math.h). On well-conditioned matrices the two agree to
    // rounding; on the numerically singular precisions a few bi-exponential voxels reach, WHICH
    // voxels end in a non-finite prediction follows the algorithm (tests/test_reference_chaos.py),
    // so the per-voxel status of the kernels is compared with this variant
    // (tests/test_hip_parity.py::test_c3_status_matches_the_oracle_with_the_same_inverse).
    if (g_inverse_mode < 0)
        g_inverse_mode = getenv("ORACLE_SWEEP_INVERSE") != nullptr ? 1 : 0;
    if (g_inverse_mode == 1)
    {
        Mat w = m;
        for (int k = 0; k < n; k++)
        {
            const real d = w(k, k);
            if (d == 0.0)
                throw SingularError();
            const real rd = 1.0 / d;
            for (int i = 0; i < n; i++)
            {
                if (i == k)
                    continue;
                const real cik = w(i, k) * rd;
                for (int j = 0; j <= i; j++)
                {
                    if (j == k)
                        continue;
                    w(i, j) -= cik * w(std::max(j, k), std::min(j, k));
                    w(j, i) = w(i, j);
                }
            }
            for (int i = 0; i < n; i++)
                if (i != k)
                {
                    w(std::max(i, k), std::min(i, k)) *= rd;
                    w(std::min(i, k), std::max(i, k)) = w(std::max(i, k), std::min(i, k));
                }
            w(k, k) = -rd;
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                w(i, j) = -w(i, j);
        return w;
    }
    Mat lu = m;
    std::vector<int> piv;
    int sign;
    if (!lu_decompose(lu, piv, sign))
        throw SingularError();
    Mat inv(n);
    vec col(n);
    for (int j = 0; j < n; j++)
    {
        for (int i = 0; i < n; i++)
            col[i] = (i == j) ? 1.0 : 0.0;
        for (int k = 0; k < n; k++)
            if (piv[k] != k)
                std::swap(col[k], col[piv[k]]);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < i; k++)
                col[i] -= lu(i, k) * col[k];
        for (int i = n - 1; i >= 0; i--)
        {
            for (int k = i + 1; k < n; k++)
                col[i] -= lu(i, k) * col[k];
            col[i] /= lu(i, i);
        }
        for (int i = 0; i < n; i++)
            inv(i, j) = col[i];
    }
    // NEWMAT SymmetricMatrix assignment keeps the lower triangle (lossy '<<'); symmetrise so
    // that both triangles are identical, as a SymmetricMatrix's storage guarantees.
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            inv(j, i) = inv(i, j);
    return inv;
}

// NEWMAT LogDeterminant(): log|det| and sign
static real logdet(const Mat &m, int &sign)
{
    Mat lu = m;
    std::vector<int> piv;
    if (!lu_decompose(lu, piv, sign))
    {
        sign = 0;
        return -INFINITY;
    }
    real l = 0;
    for (int i = 0; i < m.n; i++)
    {
        real d = lu(i, i);
        if (d < 0)
        {
            sign = -sign;
            d = -d;
        }
        l += r_log(d);
    }
    return l;
}

// ---------------------------------------------------------------------------------------------
// MVNDist (dist_mvn.cc). Lazy precision <-> covariance with 1e-10 ridge retry.
// ---------------------------------------------------------------------------------------------
struct Mvn
{
    int n;
    vec means;
    mutable Mat prec, cov;
    mutable bool precValid, covValid;

    Mvn()
        : n(-1)
        , precValid(false)
        , covValid(false)
    {
    }
    explicit Mvn(int dim)
        : n(-1)
        , precValid(false)
        , covValid(false)
    {
        SetSize(dim);
    }
    // dist_mvn.cc:174-195
    void SetSize(int dim)
    {
        if (dim <= 0)
            throw InternalError(FVB_BAD_RESULT, "MVNDist::SetSize dim<=0");
        if (n != dim)
        {
            n = dim;
            means.assign(dim, 0.0);
            prec = Mat::identity(dim);
            cov = Mat::identity(dim);
        }
        precValid = true;
        covValid = true;
    }
    // dist_mvn.cc:197-230
    const Mat &GetPrecisions() const
    {
        if (n == -1)
            throw InternalError(FVB_BAD_RESULT, "MVNDist::GetPrecisions size = -1");
        if (!precValid)
        {
            try
            {
                prec = inverse(cov);
            }
            catch (SingularError &)
            {
                Mat tmp = cov;
                for (int i = 0; i < n; i++)
                    tmp(i, i) += 1e-10;
                prec = inverse(tmp);
            }
            precValid = true;
        }
        return prec;
    }
    // dist_mvn.cc:232-265
    const Mat &GetCovariance() const
    {
        if (n == -1)
            throw InternalError(FVB_BAD_RESULT, "MVNDist::GetCovariance size = -1");
        if (!covValid)
        {
            try
            {
                cov = inverse(prec);
            }
            catch (SingularError &)
            {
                Mat tmp = prec;
                for (int i = 0; i < n; i++)
                    tmp(i, i) += 1e-10;
                cov = inverse(tmp);
            }
            covValid = true;
        }
        return cov;
    }
    // dist_mvn.cc:267-285
    void SetPrecisions(const Mat &from)
    {
        prec = from;
        precValid = true;
        covValid = false;
    }
    void SetCovariance(const Mat &from)
    {
        cov = from;
        covValid = true;
        precValid = false;
    }
};

// ---------------------------------------------------------------------------------------------
// Scalar special functions
// ---------------------------------------------------------------------------------------------
// tools.cc:87-98 (6-term Lanczos)
static real gammaln(real x)
{
    static const real series[7] = { 2.5066282746310005, 76.18009172947146, -86.50532032941677,
        24.01409824083091, -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5 };
    real total = 1.000000000190015;
    for (int i = 2; i <= 7; i++)
        total += series[i - 1] / (x + i - 1);
    return r_log(series[0] * total / x) + (x + 0.5) * r_log(x + 5.5) - x - 5.5;
}

// MISCMATHS::digamma is third-party (FSL miscmaths, not in the reference tree). Restated from
// the function's definition in fp64: psi(x) = psi(x+1) - 1/x until x >= 10, then the
// asymptotic expansion ln x - 1/2x - sum B_2k / (2k x^2k).
static real digamma(real x)
{
    real r = 0;
    while (x < 10.0)
    {
        r -= 1.0 / x;
        x += 1.0;
    }
    const real f = 1.0 / (x * x);
    const real t = f
        * (-1.0 / 12.0
              + f * (1.0 / 120.0
                        + f * (-1.0 / 252.0
                                  + f * (1.0 / 240.0
                                            + f * (-1.0 / 132.0
                                                      + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + r_log(x) - 0.5 / x + t;
}

// ---------------------------------------------------------------------------------------------
// Transforms (transforms.h:114-242, transforms.cc:17-25)
// ---------------------------------------------------------------------------------------------
static real to_model(int tr, real val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_LOG:
        return r_exp(val);
    case FVB_TRANSFORM_SOFTPLUS:
        return (val < 10) ? r_log(1 + r_exp(val)) : val;
    case FVB_TRANSFORM_FRACTIONAL:
        return 1 / (1 + r_exp(val));
    case FVB_TRANSFORM_ABS:
        return r_fabs(val);
    default:
        return val;
    }
}
static real to_fabber(int tr, real val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_LOG:
        return r_log(val);
    case FVB_TRANSFORM_SOFTPLUS:
        return (val < 10) ? r_log(r_exp(val) - 1) : val;
    case FVB_TRANSFORM_FRACTIONAL:
        return r_log(1 / val - 1);
    default:
        return val;
    }
}
static real to_model_var(int tr, real val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_IDENTITY:
    case FVB_TRANSFORM_FRACTIONAL:
        return val;
    case FVB_TRANSFORM_LOG:
        return r_exp(val);
    default: // transforms.cc:17-20
        return r_pow(to_model(tr, r_sqrt(val)) - to_model(tr, 0), 2);
    }
}
static real to_fabber_var(int tr, real val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_IDENTITY:
    case FVB_TRANSFORM_FRACTIONAL:
        return val;
    case FVB_TRANSFORM_LOG:
        return r_log(val);
    default: // transforms.cc:22-25
        return r_pow(to_fabber(tr, to_model(tr, 0) + r_sqrt(val)), 2);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward models
// ---------------------------------------------------------------------------------------------
struct Model
{
    const fvb_config *cfg;
    int T, P;
    vec data; // current voxel data (FwdModel::PassData, fwdmodel.cc:198-208)

    // EvaluateModel in MODEL space
    void EvaluateModel(const vec &params, vec &result) const
    {
        result.assign(T, 0.0);
        switch (cfg->model)
        {
        case FVB_MODEL_POLY: // fwdmodel_poly.cc:62-80
        {
            const int degree = cfg->model_iopt[0];
            for (int i = 1; i <= T; i++)
            {
                real res = 0;
                int p = 1;
                for (int n = 0; n <= degree; n++)
                {
                    res += params[n] * p;
                    p *= i;
                }
                result[i - 1] = res;
            }
            break;
        }
        case FVB_MODEL_LINEAR: // fwdmodel_linear.cc:92-96, centre = offset = 0 (:66-69)
        {
            for (int t = 0; t < T; t++)
            {
                real s = 0;
                for (int j = 0; j < P; j++)
                    s += cfg->design[(size_t)t * P + j] * (params[j] - 0.0);
                result[t] = s + 0.0;
            }
            break;
        }
        case FVB_MODEL_EXP: // examples/fwdmodel_exp.cc:65-82
        {
            const int num = cfg->model_iopt[0];
            const real dt = cfg->model_dopt[0];
            for (int i = 0; i < num; i++)
            {
                real amp = params[2 * i];
                real r = params[2 * i + 1];
                for (int k = 0; k < T; k++)
                {
                    real t = real(k) * dt;
                    real val = amp * r_exp(-r * t);
                    result[k] += val;
                }
            }
            break;
        }
        default:
            throw InternalError(FVB_BAD_RESULT, "oracle: model has no CPU body");
        }
    }

    // fwdmodel.cc:365-382
    void EvaluateFabber(const vec &params, vec &result) const
    {
        vec tparams(P);
        for (int i = 0; i < P; i++)
            tparams[i] = to_model(FVB_PARAM(cfg, transform, i), params[i]);
        EvaluateModel(tparams, result);
    }

    // Model hook InitVoxelPosterior, model space
    void InitVoxelPosterior(Mvn &post) const
    {
        if (cfg->model == FVB_MODEL_EXP) // examples/fwdmodel_exp.cc:84-91
        {
            const int num = cfg->model_iopt[0];
            real data_max = data[0];
            for (int t = 1; t < T; t++)
                if (data[t] > data_max)
                    data_max = data[t];
            for (int i = 0; i < num; i++)
                post.means[2 * i] = data_max / (num + i);
        }
    }

    // fwdmodel.cc:284-313
    void GetInitialPosterior(Mvn &post, int v) const
    {
        post.SetSize(P);
        Mat cov = post.GetCovariance();
        for (int p = 0; p < P; p++)
        {
            if (FVB_PARAM(cfg, prior_type, p) == FVB_PRIOR_IMAGE)
                post.means[p] = FVB_PARAM(cfg, image_prior, p)[v];
            else
                post.means[p] = FVB_PARAM(cfg, post_mean, p);
            cov(p, p) = FVB_PARAM(cfg, post_var, p);
        }
        post.SetCovariance(cov);
        InitVoxelPosterior(post);
        // ToFabber, fwdmodel.cc:315-324
        Mat c2 = post.GetCovariance();
        for (int p = 0; p < P; p++)
        {
            post.means[p] = to_fabber(FVB_PARAM(cfg, transform, p), post.means[p]);
            c2(p, p) = to_fabber_var(FVB_PARAM(cfg, transform, p), c2(p, p));
        }
        post.SetCovariance(c2);
    }
};

// ---------------------------------------------------------------------------------------------
// LinearizedFwdModel (fwdmodel_linear.cc:126-182)
// ---------------------------------------------------------------------------------------------
struct Linearized
{
    const Model *model;
    vec centre, offset;
    vec J; // T x P row-major
    void ReCentre(const vec &about)
    {
        const int T = model->T, P = model->P;
        centre = about;
        model->EvaluateFabber(centre, offset);
        for (int t = 0; t < T; t++)
            if (!(0 * offset[t] == 0 * offset[t])) // :134
                throw InternalError(FVB_BAD_OFFSET, "ReCentre: Non-finite values found in offset");
        J.assign((size_t)T * P, 0.0);
        vec centre2, centre3, offset2, offset3;
        for (int i = 0; i < P; i++)
        {
            real delta = centre[i] * 1e-5; // :157-161
            if (delta < 0)
                delta = -delta;
            if (delta < 1e-10)
                delta = 1e-10;
            centre3 = centre;
            centre2 = centre;
            centre2[i] += delta;
            centre3[i] -= delta;
            model->EvaluateFabber(centre2, offset2);
            model->EvaluateFabber(centre3, offset3);
            const real denom = centre2[i] - centre3[i];
            for (int t = 0; t < T; t++)
                J[(size_t)t * P + i] = (offset2[t] - offset3[t]) / denom; // :170
        }
        // Experiment switch (tests/measure/structured_j.py): the same difference quotient for the
        // exponential model without the cancellation of the unperturbed terms. Never set in tests.
        static const bool structured = getenv("ORACLE_STRUCTURED_J") != nullptr;
        if (structured && model->cfg->model == FVB_MODEL_EXP)
        {
            const real dt = model->cfg->model_dopt[0];
            for (int i = 0; i < P; i++)
            {
                real delta = centre[i] * 1e-5;
                if (delta < 0)
                    delta = -delta;
                if (delta < 1e-10)
                    delta = 1e-10;
                const real c2 = centre[i] + delta, c3 = centre[i] - delta;
                const int tr = FVB_PARAM(model->cfg, transform, i);
                const real p2 = to_model(tr, c2), p3 = to_model(tr, c3);
                const int e = i / 2;
                const real amp = to_model(FVB_PARAM(model->cfg, transform, 2 * e), centre[2 * e]);
                const real rate = to_model(FVB_PARAM(model->cfg, transform, 2 * e + 1), centre[2 * e + 1]);
                for (int t = 0; t < T; t++)
                {
                    const real tt = real(t) * dt;
                    if (i % 2 == 0)
                        J[(size_t)t * P + i] = ((p2 - p3) / (c2 - c3)) * r_exp(-rate * tt);
                    else
                        J[(size_t)t * P + i] = (r_exp(-p2 * tt) - r_exp(-p3 * tt)) * (amp / (c2 - c3));
                }
            }
        }
        for (size_t k = 0; k < J.size(); k++)
            if (!(0 * J[k] == 0 * J[k])) // :174
                throw InternalError(FVB_BAD_JACOBIAN, "ReCentre: Non-finite values found in jacobian");
    }
};

// ---------------------------------------------------------------------------------------------
// White noise model (noisemodel_white.cc)
// ---------------------------------------------------------------------------------------------
struct Gamma
{
    real b, c;
};
typedef std::vector<Gamma> NoiseParams;

struct WhiteNoise
{
    const fvb_config *cfg;
    int T, nPhis;
    std::vector<vec> Qis; // diagonal of each Q_i (noisemodel_white.cc:166-226)
    int nMasked;

    void init(const fvb_config *c)
    {
        cfg = c;
        T = c->n_times;
        nPhis = c->n_phis;
        Qis.assign(nPhis, vec(T, 0.0));
        nMasked = 0;
        for (int t = 0; t < T; t++)
        {
            int idx = cfg->phi_index ? cfg->phi_index[t] : 0;
            if (idx == 255)
                nMasked++;
            else
                Qis[idx][t] = 1.0;
        }
    }

    // noisemodel_white.cc:228-273
    void UpdateNoise(NoiseParams &post, const NoiseParams &prior, const Mvn &theta, const Linearized &lin,
        const vec &data) const
    {
        const int P = theta.n;
        vec k(T);
        for (int t = 0; t < T; t++)
        {
            real s = 0;
            for (int j = 0; j < P; j++)
                s += lin.J[(size_t)t * P + j] * (lin.centre[j] - theta.means[j]);
            k[t] = data[t] - lin.offset[t] + s;
        }
        const Mat &Sigma = theta.GetCovariance();
        for (int i = 0; i < nPhis; i++)
        {
            const vec &Qi = Qis[i];
            real kQk = 0;
            for (int t = 0; t < T; t++)
                kQk += k[t] * Qi[t] * k[t];
            // (Sigma * J' * Qi * J).Trace()
            real tr = 0;
            for (int a = 0; a < P; a++)
                for (int b2 = 0; b2 < P; b2++)
                {
                    real jqj = 0;
                    for (int t = 0; t < T; t++)
                        jqj += lin.J[(size_t)t * P + b2] * Qi[t] * lin.J[(size_t)t * P + a];
                    tr += Sigma(a, b2) * jqj;
                }
            real tmp = kQk + tr;
            post[i].b = 1 / (tmp * 0.5 + 1 / prior[i].b); // :255
            real nTimes = 0;
            for (int t = 0; t < T; t++)
                nTimes += Qi[t];
            post[i].c = (nTimes - 1) * 0.5 + prior[i].c; // :263
            if (cfg->locked_noise_stdev > 0)             // :265-271
                post[i].b = 1 / post[i].c / cfg->locked_noise_stdev / cfg->locked_noise_stdev;
        }
    }

    // noisemodel_white.cc:275-363
    void UpdateTheta(const NoiseParams &noise, Mvn &theta, const Mvn &thetaPrior, const Linearized &lin,
        const vec &data, float LMalpha) const
    {
        const int P = theta.n;
        const vec &ml = lin.centre;
        const vec &gml = lin.offset;
        const vec &J = lin.J;
        vec X(T, 0.0);
        for (int i = 0; i < nPhis; i++)
            for (int t = 0; t < T; t++)
                X[t] += Qis[i][t] * (noise[i].b * noise[i].c);
        Mat Ltmp(P);
        for (int a = 0; a < P; a++)
            for (int b2 = 0; b2 <= a; b2++)
            {
                real s = 0;
                for (int t = 0; t < T; t++)
                    s += J[(size_t)t * P + a] * X[t] * J[(size_t)t * P + b2];
                Ltmp(a, b2) = Ltmp(b2, a) = s;
            }
        const Mat &L0 = thetaPrior.GetPrecisions();
        Mat L(P);
        for (int a = 0; a < P; a++)
            for (int b2 = 0; b2 < P; b2++)
                L(a, b2) = L0(a, b2) + Ltmp(a, b2);
        theta.SetPrecisions(L); // :305
        // :308-316 sign check only logs
        vec mTmp(P, 0.0);
        if (LMalpha <= 0.0)
        {
            // :321 J' X (data - gml + J ml)
            vec w(T);
            for (int t = 0; t < T; t++)
            {
                real s = 0;
                for (int j = 0; j < P; j++)
                    s += J[(size_t)t * P + j] * ml[j];
                w[t] = X[t] * (data[t] - gml[t] + s);
            }
            for (int a = 0; a < P; a++)
            {
                real s = 0;
                for (int t = 0; t < T; t++)
                    s += J[(size_t)t * P + a] * w[t];
                mTmp[a] = s;
            }
            vec rhs(P);
            for (int a = 0; a < P; a++)
            {
                real s = 0;
                for (int b2 = 0; b2 < P; b2++)
                    s += L0(a, b2) * thetaPrior.means[b2];
                rhs[a] = mTmp[a] + s;
            }
            const Mat &Sigma = theta.GetCovariance();
            for (int a = 0; a < P; a++)
            {
                real s = 0;
                for (int b2 = 0; b2 < P; b2++)
                    s += Sigma(a, b2) * rhs[b2];
                theta.means[a] = s; // :327-328
            }
        }
        else
        {
            // :330-350 Levenberg-Marquardt form
            const Mat &prec = theta.GetPrecisions();
            vec Delta(P);
            for (int a = 0; a < P; a++)
            {
                real s = 0;
                for (int t = 0; t < T; t++)
                    s += J[(size_t)t * P + a] * X[t] * (data[t] - gml[t]);
                real p1 = 0, p2 = 0;
                for (int b2 = 0; b2 < P; b2++)
                {
                    p1 += L0(a, b2) * thetaPrior.means[b2];
                    p2 += L0(a, b2) * ml[b2];
                }
                Delta[a] = s + p1 - p2;
            }
            Mat M(P);
            for (int a = 0; a < P; a++)
                for (int b2 = 0; b2 < P; b2++)
                    M(a, b2) = prec(a, b2) + ((a == b2) ? (real)LMalpha * prec(a, a) : 0.0);
            try
            {
                Mat Mi = inverse(M);
                for (int a = 0; a < P; a++)
                {
                    real s = 0;
                    for (int b2 = 0; b2 < P; b2++)
                        s += Mi(a, b2) * Delta[b2];
                    theta.means[a] = ml[a] + s;
                }
            }
            catch (SingularError &)
            {
                // :347-350 warn, keep means
            }
        }
    }

    // noisemodel_white.cc:365-454
    real CalcFreeEnergy(const NoiseParams &noise, const NoiseParams &noisePrior, const Mvn &theta,
        const Mvn &thetaPrior, const Linearized &lin, const vec &data) const
    {
        const int P = theta.n;
        const vec &J = lin.J;
        vec k(T);
        for (int t = 0; t < T; t++)
        {
            real s = 0;
            for (int j = 0; j < P; j++)
                s += J[(size_t)t * P + j] * (lin.centre[j] - theta.means[j]);
            k[t] = data[t] - lin.offset[t] + s;
        }
        const Mat &Linv = theta.GetCovariance();
        int nTimes = T - nMasked;
        int nTheta = P;
        int sgn;
        real expectedLogThetaDist
            = +0.5 * logdet(theta.GetPrecisions(), sgn) - 0.5 * nTheta * (r_log(2 * M_PI) + 1);
        real expectedLogPhiDist = 0;
        real parts[10];
        for (int i = 0; i < 10; i++)
            parts[i] = 0;
        for (int i = 0; i < nPhis; i++)
        {
            real si = noise[i].b, ci = noise[i].c;
            real siPrior = noisePrior[i].b, ciPrior = noisePrior[i].c;
            expectedLogPhiDist
                += -gammaln(ci) - ci * r_log(si) - ci + (ci - 1) * (digamma(ci) + r_log(si));
            real trQ = 0;
            for (int t = 0; t < T; t++)
                trQ += Qis[i][t];
            parts[0] += (digamma(ci) + r_log(si)) * (trQ * 0.5 + ciPrior - 1);
            parts[9] += -gammaln(ciPrior) - ciPrior * r_log(siPrior) - si * ci / siPrior;
            real kk = 0;
            for (int t = 0; t < T; t++)
                kk += (Qis[i][t] * k[t]) * (Qis[i][t] * k[t]);
            real tr = 0; // (Ji' Ji Linv).Trace()
            for (int a = 0; a < P; a++)
                for (int b2 = 0; b2 < P; b2++)
                {
                    real jj = 0;
                    for (int t = 0; t < T; t++)
                        jj += (Qis[i][t] * J[(size_t)t * P + a]) * (Qis[i][t] * J[(size_t)t * P + b2]);
                    tr += jj * Linv(b2, a);
                }
            parts[2] += -0.5 * si * ci * kk - 0.5 * tr; // :416-417
        }
        parts[3] = +0.5 * logdet(thetaPrior.GetPrecisions(), sgn) - 0.5 * nTimes * r_log(2 * M_PI)
            - 0.5 * nTheta * r_log(2 * M_PI);
        const Mat &L0 = thetaPrior.GetPrecisions();
        real q = 0;
        for (int a = 0; a < P; a++)
            for (int b2 = 0; b2 < P; b2++)
                q += (theta.means[a] - thetaPrior.means[a]) * L0(a, b2) * (theta.means[b2] - thetaPrior.means[b2]);
        parts[4] = -0.5 * q;
        real tr2 = 0;
        for (int a = 0; a < P; a++)
            for (int b2 = 0; b2 < P; b2++)
                tr2 += Linv(a, b2) * L0(b2, a);
        parts[5] = -0.5 * tr2;
        real F = -expectedLogThetaDist - expectedLogPhiDist;
        for (int i = 0; i < 10; i++)
            F += parts[i];
        if (!(F - F == 0)) // :445
            throw InternalError(FVB_BAD_FREE_ENERGY, "WhiteNoiseModel::Non-finite free energy!");
        return F;
    }
};

// ---------------------------------------------------------------------------------------------
// Convergence detectors (convergence.cc, convergence.h)
// ---------------------------------------------------------------------------------------------
struct Conv
{
    virtual ~Conv()
    {
    }
    virtual bool Test(real F) = 0;
    virtual void Reset(real F = -99e99) = 0;
    virtual bool UseF() const
    {
        return false;
    }
    virtual bool NeedSave()
    {
        return false;
    }
    virtual bool NeedRevert()
    {
        return false;
    }
    virtual float LMalpha()
    {
        return 0.0;
    }
};
struct CountingConv : Conv // convergence.cc:34-67
{
    int m_its, m_max_its;
    explicit CountingConv(int max_its)
        : m_its(0)
        , m_max_its(max_its)
    {
    }
    bool Test(real)
    {
        ++m_its;
        return m_its >= m_max_its;
    }
    void Reset(real = -99e99)
    {
        m_its = 0;
    }
};
struct FchangeConv : CountingConv // convergence.cc:69-110
{
    real m_prev_f, m_min_fchange;
    bool m_revert, m_save;
    FchangeConv(int max_its, real min_fchange)
        : CountingConv(max_its)
        , m_min_fchange(min_fchange)
    {
        Reset();
    }
    void Reset(real F = -99e99)
    {
        CountingConv::Reset();
        m_prev_f = F;
        m_save = false;
        m_revert = false;
    }
    bool Test(real F)
    {
        real diff = F - m_prev_f;
        m_prev_f = F;
        diff = diff > 0 ? diff : -diff;
        if (diff < m_min_fchange)
            return true;
        return CountingConv::Test(F);
    }
    bool UseF() const
    {
        return true;
    }
    bool NeedSave()
    {
        return m_save;
    }
    bool NeedRevert()
    {
        return m_revert;
    }
};
struct FreduceConv : FchangeConv // convergence.cc:111-139
{
    FreduceConv(int max_its, real min_fchange)
        : FchangeConv(max_its, min_fchange)
    {
    }
    bool Test(real F)
    {
        real diff = F - m_prev_f;
        if (diff < 0)
        {
            m_revert = true;
            return true;
        }
        return FchangeConv::Test(F);
    }
};
struct TrialModeConv : FchangeConv // convergence.cc:140-251
{
    int m_trials, m_max_trials;
    bool m_trialmode;
    TrialModeConv(int max_its, real min_fchange, int max_trials)
        : FchangeConv(max_its, min_fchange)
        , m_max_trials(max_trials)
    {
        m_max_its += 1; // :145
        Reset();
    }
    void Reset(real = -99e99)
    {
        FchangeConv::Reset();
        m_trials = 0;
        m_save = true;
        m_trialmode = false;
    }
    bool Test(real F)
    {
        real diff = F - m_prev_f;
        if (!m_trialmode)
        {
            if (diff < 0)
            {
                m_its = 1;
                m_trials = 1;
                m_trialmode = true;
                m_revert = true;
                m_save = false;
                return false;
            }
            real absdiff = diff > 0 ? diff : -diff;
            if (absdiff < m_min_fchange)
            {
                m_revert = false;
                m_save = false;
                return true;
            }
            m_save = true;
            m_revert = false;
            m_prev_f = F;
            ++m_its;
            return (m_its >= m_max_its);
        }
        ++m_trials;
        if (diff > 0)
        {
            real absdiff = diff > 0 ? diff : -diff;
            if (absdiff < m_min_fchange)
            {
                m_revert = false;
                m_save = false;
                return true;
            }
            m_trialmode = false;
            m_trials = 0;
            m_save = true;
            m_revert = false;
            m_prev_f = F;
            return false;
        }
        if (m_trials >= m_max_trials)
        {
            m_save = false;
            m_revert = true;
            return true;
        }
        m_save = false;
        m_revert = false;
        return false;
    }
};
struct LMConv : Conv // convergence.cc:252-385
{
    int m_its, m_max_its;
    real m_prev, m_max_fchange;
    bool m_save, m_revert, m_LM;
    real m_alpha, m_alphastart, m_alphamax;
    LMConv(int max_its, real max_fchange)
        : m_max_its(max_its)
        , m_max_fchange(max_fchange)
    {
        Reset();
    }
    void Reset(real F = -99e99)
    {
        m_its = 0;
        m_prev = F;
        m_save = true;
        m_revert = false;
        m_alphastart = 1e-6;
        m_alpha = 0.0;
        m_alphamax = 1e6;
        m_LM = false;
    }
    bool UseF() const
    {
        return true;
    }
    bool NeedSave()
    {
        return m_save;
    }
    bool NeedRevert()
    {
        return m_revert;
    }
    float LMalpha()
    {
        return (float)m_alpha;
    }
    bool Test(real F)
    {
        real diff = F - m_prev;
        real absdiff = diff < 0 ? -diff : diff;
        if (!m_LM)
        {
            if (diff < 0)
            {
                m_LM = true;
                m_revert = true;
                m_alpha = m_alphastart;
                return false;
            }
            else if (absdiff < m_max_fchange)
            {
                m_revert = false;
                return true;
            }
            else if (m_its >= m_max_its)
            {
                m_revert = false;
                return true;
            }
            m_prev = F;
            ++m_its;
            return false;
        }
        if (diff > 0)
        {
            if (m_alpha == m_alphastart)
                m_LM = false;
            else
            {
                m_alpha /= 10;
                m_LM = true;
            }
            m_revert = false;
            m_prev = F;
            ++m_its;
            return false;
        }
        else if (m_alpha >= m_alphamax)
        {
            m_revert = true;
            return true;
        }
        else if (m_its >= m_max_its)
        {
            m_revert = false;
            return true;
        }
        m_alpha *= 10;
        m_revert = true;
        return false;
    }
};

static Conv *make_conv(int conv, int max_its, int max_trials, real min_fchange)
{
    switch (conv)
    {
    case FVB_CONV_MAXITS:
        return new CountingConv(max_its);
    case FVB_CONV_FCHANGE:
        return new FchangeConv(max_its, min_fchange);
    case FVB_CONV_FREDUCE:
        return new FreduceConv(max_its, min_fchange);
    case FVB_CONV_TRIALMODE:
        return new TrialModeConv(max_its, min_fchange, max_trials);
    case FVB_CONV_LM:
        return new LMConv(max_its, min_fchange);
    }
    throw std::runtime_error("oracle: unknown convergence detector");
}

// ---------------------------------------------------------------------------------------------
// Priors (priors.cc:108-181). Returns the free-energy contribution.
// ---------------------------------------------------------------------------------------------
static real apply_prior(const fvb_config *cfg, int k, Mvn *prior, const Mvn &fwd_post, int v, int it)
{
    switch (FVB_PARAM(cfg, prior_type, k))
    {
    case FVB_PRIOR_NORMAL: // DefaultPrior::ApplyToMVN :108-117
    {
        prior->means[k] = FVB_PARAM(cfg, prior_mean, k);
        Mat prec = prior->GetPrecisions();
        prec(k, k) = FVB_PARAM(cfg, prior_prec, k);
        prior->SetPrecisions(prec);
        return 0;
    }
    case FVB_PRIOR_IMAGE: // ImagePrior::ApplyToMVN :133-142
    {
        prior->means[k] = FVB_PARAM(cfg, image_prior, k)[v];
        Mat prec = prior->GetPrecisions();
        prec(k, k) = FVB_PARAM(cfg, prior_prec, k);
        prior->SetPrecisions(prec);
        return 0;
    }
    case FVB_PRIOR_ARD: // ARDPrior::ApplyToMVN :150-181
    {
        Mat cov = prior->GetCovariance();
        real post_mean = fwd_post.means[k];
        real post_cov = fwd_post.GetCovariance()(k, k);
        real new_cov = post_mean * post_mean + post_cov;
        if (it == 0)
        {
            cov(k, k) = FVB_PARAM(cfg, prior_var, k);
            prior->means[k] = FVB_PARAM(cfg, prior_mean, k);
        }
        else
        {
            cov(k, k) = new_cov;
        }
        prior->SetCovariance(cov);
        real b = 2 / new_cov;
        return -1.5 * (r_log(b) + digamma(0.5)) - 0.5 - gammaln(0.5) - 0.5 * r_log(b);
    }
    default:
        throw std::runtime_error("oracle: spatial priors are handled by the spatial loop");
    }
}

// WhiteParams::OutputAsMVN (noisemodel_white.cc:55-68) + MVNDist concat ctor (dist_mvn.cc:57-100)
// + MVNDist::Save packing (dist_mvn.cc:410-429)
static void write_result(const fvb_config *cfg, const fvb_outputs *out, int v, const Mvn &fwd_post,
    const NoiseParams &noise)
{
    const int P = cfg->n_params, N = cfg->n_phis, n = P + N;
    const size_t V = cfg->n_voxels;
    Mat cov(n);
    for (int r = 0; r < P; r++)
        for (int c = 0; c < P; c++)
        {
            try
            {
                cov(r, c) = fwd_post.GetCovariance()(r, c);
            }
            catch (std::exception &)
            {
                cov(r, c) = 0;
            }
        }
    vec means(n);
    for (int p = 0; p < P; p++)
        means[p] = fwd_post.means[p];
    for (int i = 0; i < N; i++)
    {
        means[P + i] = noise[i].b * noise[i].c;               // dist_gamma.cc:21-24
        cov(P + i, P + i) = noise[i].b * noise[i].b * noise[i].c; // dist_gamma.cc:25-28
    }
    size_t row = 0;
    for (int r = 0; r < n; r++)
        for (int c = 0; c <= r; c++)
            out->mvn[(row++) * V + v] = cov(r, c);
    for (int p = 0; p < n; p++)
        out->mvn[(row++) * V + v] = means[p];
    out->mvn[(row++) * V + v] = 1.0;
}

// zero +- identity (inference_vb.cc:556-570)
static void write_fallback(const fvb_config *cfg, const fvb_outputs *out, int v)
{
    const int n = cfg->n_params + cfg->n_phis;
    const size_t V = cfg->n_voxels;
    size_t row = 0;
    for (int r = 0; r < n; r++)
        for (int c = 0; c <= r; c++)
            out->mvn[(row++) * V + v] = (r == c) ? 1.0 : 0.0;
    for (int p = 0; p < n; p++)
        out->mvn[(row++) * V + v] = 0.0;
    out->mvn[(row++) * V + v] = 1.0;
}

// MVNDist::Load (dist_mvn.cc:324-375) for one voxel + GetSubmatrix (dist_mvn.cc:136-166) +
// WhiteParams::InputFromMVN (noisemodel_white.cc:70-79)
static void load_from_mvn(const fvb_config *cfg, int v, Mvn &fwd_post, NoiseParams &noise)
{
    const int P = cfg->n_params, N = cfg->n_phis, n = P + N;
    const size_t V = cfg->n_voxels;
    Mat cov(n);
    size_t row = 0;
    for (int r = 0; r < n; r++)
        for (int c = 0; c <= r; c++)
        {
            real val = cfg->init_mvn[(row++) * V + v];
            cov(r, c) = cov(c, r) = val;
        }
    vec means(n);
    for (int p = 0; p < n; p++)
        means[p] = cfg->init_mvn[(row++) * V + v];
    if (cfg->init_mvn[row * V + v] != 1)
        throw std::runtime_error("MVNDist::Load - last value != 1");
    fwd_post.SetSize(P);
    Mat c1(P);
    for (int r = 0; r < P; r++)
        for (int c = 0; c < P; c++)
            c1(r, c) = cov(r, c);
    fwd_post.SetCovariance(c1);
    for (int p = 0; p < P; p++)
        fwd_post.means[p] = means[p];
    for (int i = 0; i < N; i++)
    {
        real m = means[P + i], var = cov(P + i, P + i);
        noise[i].b = var / m; // dist_gamma.cc:29-33
        noise[i].c = m / noise[i].b;
    }
}

static inline real load_data(const fvb_config *cfg, const void *data, size_t idx)
{
    return cfg->data_f64 ? ((const double *)data)[idx] : (real)((const float *)data)[idx];
}

#ifndef ORACLE_QUAD // the binary128 build restates the white-noise loops only (voxelwise and spatial)
#include "vb_oracle_ar.inc"
#include "vb_oracle_arn.inc"
#endif
#include "vb_oracle_spatial.inc"
#ifndef ORACLE_QUAD
#include "vb_oracle_nlls.inc"
#endif

} // namespace

// =============================================================================================
// Public entry points
// =============================================================================================
extern "C" {

int32_t fabber_vb_mvn_rows_oracle(int32_t n)
{
    return n * (n + 1) / 2 + n + 1;
}

// Vb::DoCalculations -> SetupPerVoxelDists -> DoCalculationsVoxelwise
// (inference_vb.cc:144-248, 360-413, 415-576)
int32_t oracle_vb_run(const fvb_config *cfg, const void *data, const fvb_outputs *out, int32_t v_begin,
    int32_t v_end, int32_t halt_bad_voxel, const oracle_trace *trace)
{
    if (cfg->abi_version != FVB_ABI_VERSION)
        return -1;
    if (cfg->noise != FVB_NOISE_WHITE && cfg->noise != FVB_NOISE_AR1)
        return -2;
    const int T = cfg->n_times, P = cfg->n_params, N = cfg->n_phis;
    const size_t V = cfg->n_voxels;
    Model model;
    model.cfg = cfg;
    model.T = T;
    model.P = P;
    model.data.assign(T, 0.0);
#ifdef ORACLE_QUAD
    if (cfg->noise == FVB_NOISE_AR1)
        return -2;
#else
    if (cfg->noise == FVB_NOISE_AR1)
    {
        if (cfg->phi_index)
            for (int t = 0; t < T; t++)
                if (cfg->phi_index[t] == 255)
                    return -3; // masked timepoints are rejected for AR noise (noisemodel_ar.cc:351-355)
        // noisemodel_ar.cc:334-349: one or two echoes, cross terms only with two
        if (cfg->n_phis < 1 || cfg->n_phis > 2 || cfg->ar_cross_terms < 0 || cfg->ar_cross_terms > 2
            || (cfg->n_phis == 1 && cfg->ar_cross_terms != 0) || T % cfg->n_phis != 0)
            return -4;
        int32_t first = 0;
        for (int v = v_begin; v < v_end; v++)
        {
            vec y(T);
            for (int t = 0; t < T; t++)
                y[t] = load_data(cfg, data, (size_t)t * V + v);
            model.data = y;
            // one echo without cross terms: the stencil restatement (vb_oracle_ar.inc) unless the
            // general one (vb_oracle_arn.inc) is asked for, which covers every configuration
            const bool general = cfg->n_phis != 1 || cfg->ar_cross_terms != 0 || getenv("ORACLE_AR_GENERAL") != nullptr;
            int st = general ? run_voxel_arn(cfg, out, v, model, y, cfg->need_f != 0)
                             : run_voxel_ar(cfg, out, v, model, y, cfg->need_f != 0);
            if (st != FVB_OK && halt_bad_voxel && first == 0)
            {
                first = v + 1;
                break;
            }
        }
        return first;
    }
#endif
    WhiteNoise noise_model;
    noise_model.init(cfg);

    // Initial noise distributions (noisemodel_white.cc:127-164 resolved by the host into cfg)
    NoiseParams initialNoisePrior(N), initialNoisePosterior(N);
    for (int i = 0; i < N; i++)
    {
        initialNoisePrior[i].b = cfg->noise_prior_b[i];
        initialNoisePrior[i].c = cfg->noise_prior_c[i];
        initialNoisePosterior[i].b = cfg->noise_post_b[i];
        initialNoisePosterior[i].c = cfg->noise_post_c[i];
    }
    const bool needF = cfg->need_f != 0;
    int32_t first_bad = 0;

    for (int v = v_begin; v < v_end; v++)
    {
        // PassModelData (inference_vb.cc:250-264): float image -> real column
        vec y(T);
        for (int t = 0; t < T; t++)
            y[t] = load_data(cfg, data, (size_t)t * V + v);
        model.data = y;

        // ---- SetupPerVoxelDists, per-voxel part (:207-247) ----
        Mvn fwd_post;
        NoiseParams noise_post = initialNoisePosterior;
        NoiseParams noise_prior = initialNoisePrior;
        Linearized lin;
        lin.model = &model;
        Mvn fwd_prior(P); // :159 mean 0, precision I
        real F = 1234.5678;
        real Fprior = 0;
        int it = 0;
        int status = FVB_OK;
        int hist_len = 0;
        Conv *conv = make_conv(cfg->convergence, cfg->max_iterations, cfg->max_trials, cfg->min_fchange);
        try
        {
            if (cfg->init_mvn)
                load_from_mvn(cfg, v, fwd_post, noise_post);
            else
                model.GetInitialPosterior(fwd_post, v);
            lin.ReCentre(fwd_post.means); // :235

            // ---- DoCalculationsVoxelwise (:423-571) ----
            NoiseParams noisePosteriorSave = noise_post;
            Mvn fwdPosteriorSave = fwd_post;
            Mvn fwdPriorSave = fwd_prior;
            lin.ReCentre(fwd_post.means); // :443
            conv->Reset();
            do
            {
                if (conv->NeedSave())
                {
                    noisePosteriorSave = noise_post;
                    fwdPosteriorSave = fwd_post;
                    fwdPriorSave = fwd_prior;
                }
                for (int k = 0; k < P; k++)
                    Fprior = apply_prior(cfg, k, &fwd_prior, fwd_post, v, it); // :460-463 ('=' !)
                if (needF)
                    F = noise_model.CalcFreeEnergy(noise_post, noise_prior, fwd_post, fwd_prior, lin, y) + Fprior;
                noise_model.UpdateTheta(noise_post, fwd_post, fwd_prior, lin, y, conv->LMalpha());
                if (needF)
                    F = noise_model.CalcFreeEnergy(noise_post, noise_prior, fwd_post, fwd_prior, lin, y) + Fprior;
                noise_model.UpdateNoise(noise_post, noise_prior, fwd_post, lin, y);
                if (needF)
                    F = noise_model.CalcFreeEnergy(noise_post, noise_prior, fwd_post, fwd_prior, lin, y) + Fprior;
                lin.ReCentre(fwd_post.means); // :490
                if (needF)
                    F = noise_model.CalcFreeEnergy(noise_post, noise_prior, fwd_post, fwd_prior, lin, y) + Fprior;
                if (out->f_history && hist_len < cfg->f_history_rows)
                    out->f_history[(size_t)hist_len * V + v] = F;
                hist_len++;
                if (trace && it < trace->max_rows)
                {
                    for (int p = 0; p < P; p++)
                        trace->means[((size_t)it * P + p) * V + v] = fwd_post.means[p];
                    if (trace->noise_b)
                        for (int i = 0; i < N; i++)
                            trace->noise_b[((size_t)it * N + i) * V + v] = noise_post[i].b;
                }
                ++it;
            } while (!conv->Test(F));

            if (conv->NeedSave()) // :506-513
            {
                noisePosteriorSave = noise_post;
                fwdPosteriorSave = fwd_post;
                fwdPriorSave = fwd_prior;
            }
            if (conv->NeedRevert()) // :516-525
            {
                noise_post = noisePosteriorSave;
                fwd_post = fwdPosteriorSave;
                fwd_prior = fwdPriorSave;
                lin.ReCentre(fwd_post.means);
                if (needF)
                    F = noise_model.CalcFreeEnergy(noise_post, noise_prior, fwd_post, fwd_prior, lin, y) + Fprior;
            }
        }
        catch (InternalError &e)
        {
            status = e.code;
        }
        catch (SingularError &)
        {
            status = FVB_BAD_RESULT;
        }
        delete conv;
        if (status != FVB_OK && halt_bad_voxel && first_bad == 0)
            first_bad = v + 1;

        // ---- result assembly (:546-570) ----
        try
        {
            write_result(cfg, out, v, fwd_post, noise_post);
        }
        catch (std::exception &)
        {
            write_fallback(cfg, out, v);
            if (status == FVB_OK)
                status = FVB_BAD_RESULT;
        }
        if (out->f_history && hist_len < cfg->f_history_rows)
            out->f_history[(size_t)hist_len * V + v] = F; // :553-554
        hist_len++;
        if (out->f_history_len)
            out->f_history_len[v] = hist_len;
        if (out->free_energy)
            out->free_energy[v] = F;
        if (out->status)
            out->status[v] = status;
        if (out->iterations)
            out->iterations[v] = it;
        if (first_bad)
            break; // the reference rethrows here
    }
    return first_bad;
}

void oracle_set_inverse(int32_t mode)
{
    g_inverse_mode = mode;
}

// Vb::DoCalculationsSpatial. Returns 0, or < 0 with the exception text in oracle_last_error().
static std::string g_oracle_error;
int32_t oracle_vb_run_spatial(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out)
{
    if (cfg->abi_version != FVB_ABI_VERSION)
        return -1;
    try
    {
        if (cfg->noise == FVB_NOISE_WHITE)
            return run_spatial<SpatialWhite>(cfg, sp, data, out);
#ifdef ORACLE_QUAD
        return -2;
#else
        if (cfg->noise != FVB_NOISE_AR1)
            return -1;
        if (cfg->phi_index)
            for (int t = 0; t < cfg->n_times; t++)
                if (cfg->phi_index[t] == 255)
                    return -3; // masked timepoints are rejected for AR noise (noisemodel_ar.cc:351-355)
        if (cfg->n_phis < 1 || cfg->n_phis > 2 || cfg->ar_cross_terms < 0 || cfg->ar_cross_terms > 2
            || (cfg->n_phis == 1 && cfg->ar_cross_terms != 0) || cfg->n_times % cfg->n_phis != 0)
            return -4;
        return run_spatial<SpatialAr>(cfg, sp, data, out);
#endif
    }
    catch (std::exception &e)
    {
        g_oracle_error = e.what();
        return -10;
    }
}
const char *oracle_last_error(void)
{
    return g_oracle_error.c_str();
}

#ifndef ORACLE_QUAD
// InferenceTechnique::SaveResults (inference.cc:112-281), Vb::SaveResults (inference_vb.cc:966-995)
int32_t oracle_vb_postproc(const fvb_config *cfg, const void *data, const double *mvn, const fvb_postproc *pp)
{
    const int T = cfg->n_times, P = cfg->n_params;
    const int N = cfg->noise == FVB_NOISE_AR1 ? 2 + cfg->ar_cross_terms + cfg->n_phis : cfg->n_phis, n = P + N;
    const size_t V = cfg->n_voxels;
    const int nCov = n * (n + 1) / 2;
    Model model;
    model.cfg = cfg;
    model.T = T;
    model.P = P;
    for (size_t v = 0; v < V; v++)
    {
        vec means(n), var(n);
        for (int i = 0; i < n; i++)
        {
            means[i] = mvn[(size_t)(nCov + i) * V + v];
            var[i] = mvn[(size_t)(i * (i + 1) / 2 + i) * V + v];
        }
        for (int p = 0; p < P; p++)
        {
            // FwdModel::ToModel (fwdmodel.cc:326-337)
            double mm = to_model(FVB_PARAM(cfg, transform, p), means[p]);
            double mv = to_model_var(FVB_PARAM(cfg, transform, p), var[p]);
            double sd = std::sqrt(mv);
            if (pp->mean)
                pp->mean[p * V + v] = mm;
            if (pp->var)
                pp->var[p * V + v] = mv;
            if (pp->std)
                pp->std[p * V + v] = sd;
            if (pp->zstat)
                pp->zstat[p * V + v] = mm / sd;
        }
        for (int i = 0; i < N; i++)
        {
            if (pp->noise_mean)
                pp->noise_mean[i * V + v] = means[P + i];
            if (pp->noise_std)
                pp->noise_std[i * V + v] = std::sqrt(var[P + i]);
        }
        if (pp->modelfit || pp->residuals)
        {
            model.data.assign(T, 0.0);
            for (int t = 0; t < T; t++)
                model.data[t] = load_data(cfg, data, (size_t)t * V + v);
            vec params(means.begin(), means.begin() + P), fit;
            model.EvaluateFabber(params, fit);
            for (int t = 0; t < T; t++)
            {
                if (pp->modelfit)
                    pp->modelfit[(size_t)t * V + v] = fit[t];
                if (pp->residuals)
                    pp->residuals[(size_t)t * V + v] = model.data[t] - fit[t];
            }
        }
    }
    return 0;
}

int32_t oracle_nlls_run(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    int32_t v_begin, int32_t v_end, int32_t halt_bad_voxel)
{
    if (cfg->abi_version != FVB_ABI_VERSION)
        return -1;
    return run_nlls(cfg, nl, data, out, v_begin, v_end, halt_bad_voxel);
}


// Neighbour lists as Vb::CalcNeighbours builds them: for voxel v (0-based) up to 6 first
// neighbours into nn[v*6..] (1-based ids, 0 = none) and the number of second neighbours (with
// duplicates) into n2count[v]; second neighbours themselves into nn2[v*30..].
int32_t oracle_calc_neighbours(const int32_t *coords, int32_t n_voxels, int32_t spatial_dims, int32_t *nn,
    int32_t *nn2, int32_t *n2count)
{
    SpatialCtx ctx;
    try
    {
        calc_neighbours(coords, n_voxels, spatial_dims, ctx);
    }
    catch (std::exception &e)
    {
        g_oracle_error = e.what();
        return -10;
    }
    for (int v = 0; v < n_voxels; v++)
    {
        for (int i = 0; i < 6; i++)
            nn[v * 6 + i] = i < (int)ctx.neighbours[v].size() ? ctx.neighbours[v][i] : 0;
        n2count[v] = (int)ctx.neighbours2[v].size();
        for (int i = 0; i < 30; i++)
            nn2[v * 30 + i] = i < (int)ctx.neighbours2[v].size() ? ctx.neighbours2[v][i] : 0;
    }
    return 0;
}

int32_t oracle_spatial_prior_apply(const int32_t *coords, int32_t n_voxels, int32_t spatial_dims, int32_t type, double mean0,
    double prec0, double aK, const double *means, double *prior_mean, double *prior_prec)
{
#ifdef ORACLE_QUAD
    return -1;
#else
    SpatialCtx ctx;
    try
    {
        calc_neighbours(coords, n_voxels, spatial_dims, ctx);
        ctx.nvoxels = n_voxels;
        ctx.it = 0;
        ctx.fwd_post.assign(n_voxels, Mvn(1));
        ctx.fwd_prior.assign(n_voxels, Mvn(1));
        for (int v = 0; v < n_voxels; v++)
            ctx.fwd_post[v].means[0] = means[v];
        SpatialPriorK s;
        s.idx = 0;
        s.type = type;
        s.mean0 = mean0;
        s.prec0 = prec0;
        s.aK = aK;
        s.dims = spatial_dims;
        s.speed = -1;
        s.q1 = 10;
        s.q2 = 1;
        s.update_first_iter = false; // (aK stays what the caller gave)
        for (int v = 1; v <= n_voxels; v++)
        {
            ctx.v = v;
            s.ApplyToMVN(&ctx.fwd_prior[v - 1], ctx);
            prior_mean[v - 1] = (double)ctx.fwd_prior[v - 1].means[0];
            prior_prec[v - 1] = (double)ctx.fwd_prior[v - 1].GetPrecisions()(0, 0);
        }
    }
    catch (std::exception &e)
    {
        g_oracle_error = e.what();
        return -10;
    }
    return 0;
#endif
}

double oracle_gammaln(double x)
{
    return gammaln(x);
}
double oracle_digamma(double x)
{
    return digamma(x);
}
double oracle_transform_to_model(int32_t tr, double x)
{
    return to_model(tr, x);
}
double oracle_transform_to_fabber(int32_t tr, double x)
{
    return to_fabber(tr, x);
}
double oracle_transform_to_model_var(int32_t tr, double x)
{
    return to_model_var(tr, x);
}
double oracle_transform_to_fabber_var(int32_t tr, double x)
{
    return to_fabber_var(tr, x);
}

int32_t oracle_evaluate_fabber(const fvb_config *cfg, const double *params, double *result)
{
    Model model;
    model.cfg = cfg;
    model.T = cfg->n_times;
    model.P = cfg->n_params;
    model.data.assign(model.T, 0.0);
    vec p(params, params + model.P), r;
    try
    {
        model.EvaluateFabber(p, r);
    }
    catch (std::exception &)
    {
        return -1;
    }
    for (int t = 0; t < model.T; t++)
        result[t] = r[t];
    return 0;
}

int32_t oracle_convergence_trace(int32_t conv, int32_t max_iterations, int32_t max_trials, double min_fchange,
    const double *F, int32_t nF, int32_t *done, int32_t *save, int32_t *revert, double *alpha,
    int32_t stop_at_done)
{
    Conv *c = make_conv(conv, max_iterations, max_trials, min_fchange);
    c->Reset();
    int n = 0;
    for (int i = 0; i < nF; i++)
    {
        bool d = c->Test(F[i]);
        done[i] = d;
        save[i] = c->NeedSave();
        revert[i] = c->NeedRevert();
        alpha[i] = c->LMalpha();
        n++;
        if (d && stop_at_done)
            break;
    }
    delete c;
    return n;
}

int32_t oracle_inverse(int32_t n, const double *a, double *inv)
{
    Mat m(n);
    m.a.assign(a, a + (size_t)n * n);
    try
    {
        Mat r = inverse(m);
        std::memcpy(inv, r.a.data(), sizeof(double) * n * n);
        return 0;
    }
    catch (SingularError &)
    {
        return 1;
    }
}

double oracle_logdet(int32_t n, const double *a, int32_t *sign)
{
    Mat m(n);
    m.a.assign(a, a + (size_t)n * n);
    int s;
    double l = logdet(m, s);
    if (sign)
        *sign = s;
    return l;
}
#endif // !ORACLE_QUAD

} // extern "C"
