// fwdmodel_bump.cc - a small model library written purely against the public plugin surface
// (fabber_core/fwdmodel.h + the three extern "C" hooks of fwdmodel.cc:25-27), as a third party
// would write one for the reference: no device body, no knowledge of the GPU engine.
//
//   "bump"   : y(t) = amp * exp(-(t - mu)^2 / (2 width^2)),  t = 1..T   (amp LOG-transformed)
//   "mypoly" : the built-in polynomial model re-implemented, to compare host-evaluated and
//              device-evaluated runs of the same model
#include "fabber_core/fwdmodel.h"
#include "fabber_core/priors.h"

#include <cmath>
#include <string>
#include <vector>

class BumpFwdModel : public FwdModel
{
public:
    static FwdModel *NewInstance()
    {
        return new BumpFwdModel();
    }
    void GetOptions(std::vector<OptionSpec> &) const
    {
    }
    std::string GetDescription() const
    {
        return "Gaussian bump: amp * exp(-(t - mu)^2 / (2 width^2))";
    }
    std::string ModelVersion() const
    {
        return "test";
    }
    void EvaluateModel(const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string & = "") const
    {
        result.ReSize(data.Nrows());
        const double amp = params(1), mu = params(2), w = params(3);
        for (int t = 1; t <= result.Nrows(); t++)
            result(t) = amp * std::exp(-(t - mu) * (t - mu) / (2 * w * w));
    }
    // data-dependent initial posterior, like the reference's example models
    void InitVoxelPosterior(MVNDist &posterior) const
    {
        int best = 1;
        for (int t = 2; t <= data.Nrows(); t++)
            if (data(t) > data(best))
                best = t;
        posterior.means(1) = data(best) > 0.1 ? data(best) : 0.1;
        posterior.means(2) = best;
    }

protected:
    void GetParameterDefaults(std::vector<Parameter> &params) const
    {
        params.clear();
        params.push_back(Parameter(0, "amp", DistParams(1, 1e6), DistParams(1, 10), PRIOR_NORMAL, TRANSFORM_LOG()));
        params.push_back(Parameter(1, "mu", DistParams(10, 1e4), DistParams(10, 25)));
        params.push_back(Parameter(2, "width", DistParams(3, 100), DistParams(3, 4)));
    }
};

class MyPolyFwdModel : public FwdModel
{
public:
    static FwdModel *NewInstance()
    {
        return new MyPolyFwdModel();
    }
    void GetOptions(std::vector<OptionSpec> &opts) const
    {
        OptionSpec degree = { "degree", OPT_INT, "Maximum power", OPT_REQ, "" };
        opts.push_back(degree);
    }
    std::string GetDescription() const
    {
        return "c0 + c1 x + c2 x^2 ...";
    }
    std::string ModelVersion() const
    {
        return "test";
    }
    void Initialize(FabberRunData &args)
    {
        FwdModel::Initialize(args);
        m_degree = args.GetInt("degree", 0);
    }
    void EvaluateModel(const NEWMAT::ColumnVector &params, NEWMAT::ColumnVector &result, const std::string & = "") const
    {
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

protected:
    void GetParameterDefaults(std::vector<Parameter> &params) const
    {
        params.clear();
        for (int i = 0; i <= m_degree; i++)
            params.push_back(Parameter(i, "c" + stringify(i), DistParams(0, 1e12), DistParams(0, 1e12)));
    }
    int m_degree;
};

extern "C" {
int get_num_models()
{
    return 2;
}
const char *get_model_name(int index)
{
    return index == 0 ? "bump" : (index == 1 ? "mypoly" : 0);
}
NewInstanceFptr get_new_instance_func(const char *name)
{
    const std::string n(name);
    if (n == "bump")
        return BumpFwdModel::NewInstance;
    if (n == "mypoly")
        return MyPolyFwdModel::NewInstance;
    return 0;
}
}
