// noisemodel_ar.cc - host description of the AR(1) noise model (options, initial
// distributions, MVN round trip). Reference: noisemodel_ar.cc:245-403.
#include "noisemodel_ar.h"

#include "../../../include/fabber_vb.h"

using namespace std;
using NEWMAT::IdentityMatrix;
using NEWMAT::SymmetricMatrix;

Ar1cParams::Ar1cParams(int nAlpha, int nPhi)
    : alpha(nAlpha)
    , phis(nPhi)
{
}
Ar1cParams::Ar1cParams(const Ar1cParams &from)
    : alpha(from.alpha)
    , phis(from.phis)
{
}
Ar1cParams *Ar1cParams::Clone() const
{
    return new Ar1cParams(*this);
}
const Ar1cParams &Ar1cParams::operator=(const NoiseParams &in)
{
    const Ar1cParams &from = dynamic_cast<const Ar1cParams &>(in);
    alpha = from.alpha;
    phis = from.phis;
    return *this;
}
const MVNDist Ar1cParams::OutputAsMVN() const
{
    MVNDist phiMVN(phis.size());
    SymmetricMatrix vars(phis.size());
    vars = 0;
    for (size_t i = 0; i < phis.size(); i++)
    {
        phiMVN.means(i + 1) = phis[i].CalcMean();
        vars(i + 1, i + 1) = phis[i].CalcVariance();
    }
    phiMVN.SetCovariance(vars);
    return MVNDist(alpha, phiMVN);
}
void Ar1cParams::InputFromMVN(const MVNDist &mvn)
{
    const int nAlpha = alpha.means.Nrows();
    if (nAlpha + (int)phis.size() != mvn.GetSize())
        throw FabberRunDataError("Ar1cParams::InputFromMVN - MVN has the wrong size");
    alpha.CopyFromSubmatrix(mvn, 1, nAlpha, true);
    const SymmetricMatrix &cov = mvn.GetCovariance();
    for (size_t i = 1; i <= phis.size(); i++)
    {
        phis[i - 1].SetMeanVariance(mvn.means(nAlpha + i), cov(nAlpha + i, nAlpha + i));
        for (size_t j = i + 1; j <= phis.size(); j++)
            if (cov(nAlpha + i, nAlpha + j) != 0.0)
                throw FabberRunDataError("Phis should have zero covariance!");
    }
}
void Ar1cParams::Dump(std::ostream &os) const
{
    os << "Alpha:" << endl;
    alpha.Dump(os);
    for (size_t i = 0; i < phis.size(); i++)
    {
        os << "Phi_" << i + 1 << ": ";
        phis[i].Dump(os);
    }
}

NoiseModel *Ar1cNoiseModel::NewInstance()
{
    return new Ar1cNoiseModel();
}

void Ar1cNoiseModel::Initialize(FabberRunData &args)
{
    NoiseModel::Initialize(args);
    const string echoes = args.GetStringDefault("num-echoes", "(default)");
    if (echoes == "(default)")
    {
        nPhis = 1;
        WARN_ONCE("Defaulting to --num-echoes=1");
    }
    else
    {
        nPhis = convertTo<int>(echoes);
    }
    ar1Type = args.GetStringDefault("ar1-cross-terms", "none");
    NumAlphas(); // validates ar1Type
    if (nPhis == 1)
    {
        if (ar1Type != "none")
            throw InvalidOptionValue("ar1-cross-terms", ar1Type, "You must use ar1-cross-terms=none with num-echoes=1");
    }
    else if (nPhis != 2)
    {
        throw InvalidOptionValue("num-echoes", stringify(nPhis), "Must be 1 or 2");
    }
    if (args.HaveKey("mt1"))
        throw InvalidOptionValue("mt1", "", "Masked time points are not supported for the AR noise model");
}

Ar1cParams *Ar1cNoiseModel::NewParams() const
{
    return new Ar1cParams(NumAlphas(), nPhis);
}
int Ar1cNoiseModel::NumParams()
{
    return nPhis;
}
int Ar1cNoiseModel::NumOutputParams()
{
    return NumAlphas() + nPhis;
}
int Ar1cNoiseModel::NumAlphas() const
{
    if (ar1Type == "same")
        return 3;
    if (ar1Type == "dual")
        return 4;
    if (ar1Type == "none")
        return 2;
    throw InvalidOptionValue("ar1-cross-terms", ar1Type, "Must be none, same or dual");
}

void Ar1cNoiseModel::HardcodedInitialDists(NoiseParams &priorIn, NoiseParams &posteriorIn) const
{
    Ar1cParams &prior = dynamic_cast<Ar1cParams &>(priorIn);
    Ar1cParams &posterior = dynamic_cast<Ar1cParams &>(posteriorIn);
    const int nAlphas = NumAlphas();
    prior.alpha.means = 0;
    posterior.alpha.means = 0;
    SymmetricMatrix weak(IdentityMatrix(nAlphas) * 1e-4);
    prior.alpha.SetPrecisions(weak);
    posterior.alpha.SetPrecisions(weak);
    for (size_t i = 0; i < prior.phis.size(); i++)
    {
        prior.phis[i].b = 1e6;
        prior.phis[i].c = 1e-6;
        posterior.phis[i].b = 1e-8;
        posterior.phis[i].c = 1e-6;
    }
}

void Ar1cNoiseModel::ConfigureEngine(fvb_config &cfg, int n_times, std::vector<unsigned char> &phi_index) const
{
    cfg.noise = FVB_NOISE_AR1;
    cfg.n_phis = nPhis;
    cfg.ar_cross_terms = NumAlphas() - 2; // none / same / dual
    if (n_times % nPhis != 0)
        throw InvalidOptionValue("num-echoes", stringify(nPhis), "The number of timepoints (" + stringify(n_times)
            + ") is not a multiple of the number of echoes");
    Ar1cParams prior(NumAlphas(), nPhis), post(NumAlphas(), nPhis);
    HardcodedInitialDists(prior, post);
    for (int i = 0; i < nPhis; i++)
    {
        cfg.noise_prior_b[i] = prior.phis[i].b;
        cfg.noise_prior_c[i] = prior.phis[i].c;
        cfg.noise_post_b[i] = post.phis[i].b;
        cfg.noise_post_c[i] = post.phis[i].c;
    }
    cfg.locked_noise_stdev = -1;
    phi_index.assign(n_times, 0);
    if (!m_masked_tpoints.empty())
        throw InvalidOptionValue("mt1", "", "Masked time points are not supported for the AR noise model");
}
