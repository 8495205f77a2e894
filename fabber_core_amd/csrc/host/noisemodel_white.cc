// noisemodel_white.cc - host description of the white noise model: option parsing, initial
// Gamma distributions, noise pattern -> per-timepoint index table handed to the HIP engine.
// Reference: noisemodel_white.cc:31-226.
#include "noisemodel_white.h"

#include "../../../include/fabber_vb.h"

#include <algorithm>

using namespace std;
using NEWMAT::SymmetricMatrix;

WhiteParams::WhiteParams(int N)
    : nPhis(N)
    , phis(N)
{
}
WhiteParams::WhiteParams(const WhiteParams &from)
    : nPhis(from.nPhis)
    , phis(from.phis)
{
}
WhiteParams *WhiteParams::Clone() const
{
    return new WhiteParams(*this);
}
const WhiteParams &WhiteParams::operator=(const NoiseParams &in)
{
    const WhiteParams &from = dynamic_cast<const WhiteParams &>(in);
    if (from.nPhis != nPhis)
        throw FabberInternalError("WhiteParams: size mismatch in assignment");
    phis = from.phis;
    return *this;
}
const MVNDist WhiteParams::OutputAsMVN() const
{
    MVNDist mvn(phis.size());
    SymmetricMatrix vars(phis.size());
    vars = 0;
    for (size_t i = 0; i < phis.size(); i++)
    {
        mvn.means(i + 1) = phis[i].CalcMean();
        vars(i + 1, i + 1) = phis[i].CalcVariance();
    }
    mvn.SetCovariance(vars);
    return mvn;
}
void WhiteParams::InputFromMVN(const MVNDist &mvn)
{
    const SymmetricMatrix &cov = mvn.GetCovariance();
    for (size_t i = 1; i <= phis.size(); i++)
    {
        phis[i - 1].SetMeanVariance(mvn.means(i), cov(i, i));
        for (int j = i + 1; j <= mvn.means.Nrows(); j++)
            if (cov(i, j) != 0.0)
                throw FabberRunDataError("Phis should have zero covariance!");
    }
}
void WhiteParams::Dump(ostream &os) const
{
    for (size_t i = 0; i < phis.size(); i++)
    {
        os << "WhiteNoiseModel::Phi_" << i + 1 << ": ";
        phis[i].Dump(os);
    }
}

NoiseModel *WhiteNoiseModel::NewInstance()
{
    return new WhiteNoiseModel();
}

static int pattern_symbol(char c)
{
    if (c >= '1' && c <= '9')
        return c - '0';
    if (c >= 'A' && c <= 'Z')
        return c - 'A' + 10;
    if (c >= 'a' && c <= 'z')
        return c - 'a' + 10;
    throw InvalidOptionValue("noise-pattern", stringify(c), "Invalid character");
}

void WhiteNoiseModel::Initialize(FabberRunData &args)
{
    NoiseModel::Initialize(args);
    // e.g. "123123..." : each distinct symbol is one noise precision, repeated along the data
    phiPattern = args.GetStringDefault("noise-pattern", "1");
    if (phiPattern.empty())
        throw InvalidOptionValue("noise-pattern", "", "Must not be empty");
    m_num_phis = 0;
    for (size_t i = 0; i < phiPattern.size(); i++)
        m_num_phis = std::max(m_num_phis, pattern_symbol(phiPattern[i]));
    if (m_num_phis > FVB_MAX_PHIS)
        throw InvalidOptionValue("noise-pattern", phiPattern, "At most " + stringify(FVB_MAX_PHIS) + " noise parameters are supported");

    lockedNoiseStdev = convertTo<double>(args.GetStringDefault("locked-noise-stdev", "-1"));
    if (!(lockedNoiseStdev == -1 || lockedNoiseStdev > 0))
        throw InvalidOptionValue("locked-noise-stdev", stringify(lockedNoiseStdev), "Must be > 0");
    phiprior = convertTo<double>(args.GetStringDefault("prior-noise-stddev", "-1"));
    if (phiprior < 0 && phiprior != -1)
        throw InvalidOptionValue("prior-noise-stddev", stringify(phiprior), "Must be > 0");
}

int WhiteNoiseModel::NumParams()
{
    return m_num_phis;
}
WhiteParams *WhiteNoiseModel::NewParams() const
{
    return new WhiteParams(m_num_phis);
}

void WhiteNoiseModel::HardcodedInitialDists(NoiseParams &priorIn, NoiseParams &posteriorIn) const
{
    WhiteParams &prior = dynamic_cast<WhiteParams &>(priorIn);
    WhiteParams &posterior = dynamic_cast<WhiteParams &>(posteriorIn);
    for (int i = 0; i < m_num_phis; i++)
    {
        if (phiprior == -1)
        {
            // non-informative prior; a tiny initial precision for the posterior
            prior.phis[i].b = 1e6;
            prior.phis[i].c = 1e-6;
            posterior.phis[i].b = 1e-8;
            posterior.phis[i].c = 50;
        }
        else
        {
            // a given noise std dev counts as one measurement (c = 0.5)
            prior.phis[i].c = posterior.phis[i].c = 0.5;
            prior.phis[i].b = posterior.phis[i].b = 1 / (phiprior * phiprior * prior.phis[i].c);
        }
    }
}

std::vector<int> WhiteNoiseModel::ExpandPattern(int n_times) const
{
    if ((int)phiPattern.size() > n_times)
        throw InvalidOptionValue("noise-pattern", phiPattern, "Pattern length exceeds data length");
    std::vector<int> idx(n_times);
    for (int t = 0; t < n_times; t++)
        idx[t] = pattern_symbol(phiPattern[t % phiPattern.size()]) - 1;
    return idx;
}

void WhiteNoiseModel::ConfigureEngine(fvb_config &cfg, int n_times, std::vector<unsigned char> &phi_index) const
{
    cfg.noise = FVB_NOISE_WHITE;
    cfg.n_phis = m_num_phis;
    WhiteParams prior(m_num_phis), post(m_num_phis);
    HardcodedInitialDists(prior, post);
    for (int i = 0; i < m_num_phis; i++)
    {
        cfg.noise_prior_b[i] = prior.phis[i].b;
        cfg.noise_prior_c[i] = prior.phis[i].c;
        cfg.noise_post_b[i] = post.phis[i].b;
        cfg.noise_post_c[i] = post.phis[i].c;
    }
    cfg.locked_noise_stdev = lockedNoiseStdev;
    std::vector<int> idx = ExpandPattern(n_times);
    phi_index.resize(n_times);
    for (int t = 0; t < n_times; t++)
        phi_index[t] = (unsigned char)idx[t];
    for (size_t k = 0; k < m_masked_tpoints.size(); k++)
    {
        const int mt = m_masked_tpoints[k];
        if (mt < 1 || mt > n_times)
            throw InvalidOptionValue("mt" + stringify(k + 1), stringify(mt), "Masked timepoint outside the data");
        phi_index[mt - 1] = 255;
    }
}
