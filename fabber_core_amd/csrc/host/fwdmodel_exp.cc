// fwdmodel_exp.cc - multi-exponential decay; device body = fvb::ExpModel (../vb_models.h)
#include "fwdmodel_exp.h"

#include "priors.h"

#include "../../../include/fabber_vb.h"

#include <math.h>

using namespace std;

FwdModel *ExpFwdModel::NewInstance()
{
    return new ExpFwdModel();
}
string ExpFwdModel::ModelVersion() const
{
    return "1.0";
}
string ExpFwdModel::GetDescription() const
{
    return "Example model of a sum of exponentials";
}

void ExpFwdModel::GetOptions(vector<OptionSpec> &opts) const
{
    OptionSpec dt = { "dt", OPT_FLOAT, "Time separation between samples", OPT_REQ, "" };
    OptionSpec num = { "num-exps", OPT_INT, "Number of independent decay rates", OPT_NONREQ, "1" };
    opts.push_back(dt);
    opts.push_back(num);
}

void ExpFwdModel::Initialize(FabberRunData &rundata)
{
    FwdModel::Initialize(rundata);
    m_dt = rundata.GetDouble("dt");
    m_num = rundata.GetIntDefault("num-exps", 1);
    if (m_num < 1)
        throw InvalidOptionValue("num-exps", stringify(m_num), "Must be >= 1");
}

void ExpFwdModel::GetParameterDefaults(vector<Parameter> &params) const
{
    // amplitude and rate of each exponential are inferred in log space so they stay positive
    params.clear();
    int idx = 0;
    for (int e = 1; e <= m_num; e++)
    {
        params.push_back(Parameter(idx++, "amp" + stringify(e), DistParams(1, 1e5), DistParams(1, 1.5), PRIOR_NORMAL, TRANSFORM_LOG()));
        params.push_back(Parameter(idx++, "r" + stringify(e), DistParams(1, 1e5), DistParams(1, 1.5), PRIOR_NORMAL, TRANSFORM_LOG()));
    }
}

void ExpFwdModel::EvaluateModel(const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const string &) const
{
    const int T = data.Nrows();
    result.ReSize(T);
    result = 0;
    for (int e = 0; e < m_num; e++)
    {
        const double amp = params(2 * e + 1), rate = params(2 * e + 2);
        for (int k = 0; k < T; k++)
            result(k + 1) += amp * exp(-rate * (double(k) * m_dt));
    }
}

void ExpFwdModel::InitVoxelPosterior(MVNDist &posterior) const
{
    const double peak = data.Maximum();
    for (int e = 0; e < m_num; e++)
        posterior.means(2 * e + 1) = peak / (m_num + e);
}

bool ExpFwdModel::GetDeviceModel(DeviceModelSpec &spec) const
{
    spec.model = FVB_MODEL_EXP;
    spec.iopt[0] = m_num;
    spec.dopt[0] = m_dt;
    return true;
}
