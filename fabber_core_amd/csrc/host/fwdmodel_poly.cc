// fwdmodel_poly.cc - polynomial model; device body = fvb::PolyModel (../vb_models.h)
#include "fwdmodel_poly.h"

#include "version.h"

#include "../../../include/fabber_vb.h"

using namespace std;

FwdModel *PolynomialFwdModel::NewInstance()
{
    return new PolynomialFwdModel();
}

void PolynomialFwdModel::GetOptions(vector<OptionSpec> &opts) const
{
    OptionSpec degree = { "degree", OPT_INT, "Maximum power in the polynomial function", OPT_REQ, "" };
    opts.push_back(degree);
}

string PolynomialFwdModel::GetDescription() const
{
    return "Model which fits data to a simple polynomial function: c0 + c1x + c2x^2 ... etc";
}
string PolynomialFwdModel::ModelVersion() const
{
    return fabber_version();
}

void PolynomialFwdModel::Initialize(FabberRunData &args)
{
    FwdModel::Initialize(args);
    m_degree = convertTo<int>(args.GetString("degree"));
    if (m_degree < 0)
        throw InvalidOptionValue("degree", stringify(m_degree), "Must be >= 0");
}

void PolynomialFwdModel::GetParameterDefaults(vector<Parameter> &params) const
{
    params.clear();
    for (int i = 0; i <= m_degree; i++)
        params.push_back(Parameter(i, "c" + stringify(i), DistParams(0, 1e12), DistParams(0, 1e12)));
}

void PolynomialFwdModel::EvaluateModel(const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const string &) const
{
    // x runs over 1..T; powers are accumulated in an int like the reference (fwdmodel_poly.cc:72-76)
    result.ReSize(data.Nrows());
    for (int x = 1; x <= result.Nrows(); x++)
    {
        double sum = 0;
        int xn = 1;
        for (int n = 0; n <= m_degree; n++, xn *= x)
            sum += params(n + 1) * xn;
        result(x) = sum;
    }
}

bool PolynomialFwdModel::GetDeviceModel(DeviceModelSpec &spec) const
{
    spec.model = FVB_MODEL_POLY;
    spec.iopt[0] = m_degree;
    return true;
}
