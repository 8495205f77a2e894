// fwdmodel_linear.cc - design-matrix model and host-side linearisation helper.
#include "fwdmodel_linear.h"

#include "tools.h"
#include "version.h"

#include "../../../include/fabber_vb.h"

using namespace std;
using namespace NEWMAT;

FwdModel *LinearFwdModel::NewInstance()
{
    return new LinearFwdModel();
}

void LinearFwdModel::GetOptions(std::vector<OptionSpec> &opts) const
{
    OptionSpec basis = { "basis", OPT_MATRIX, "Design matrix", OPT_REQ, "" };
    OptionSpec ones = { "add-ones-regressor", OPT_BOOL, "Append a constant regressor to the design matrix", OPT_NONREQ, "" };
    opts.push_back(basis);
    opts.push_back(ones);
}

std::string LinearFwdModel::GetDescription() const
{
    return "Model in which output is a linear combination of input parameters";
}
string LinearFwdModel::ModelVersion() const
{
    return fabber_version();
}

void LinearFwdModel::Initialize(FabberRunData &args)
{
    FwdModel::Initialize(args);
    const string designFile = args.GetString("basis");
    LOG << "LinearFwdModel::Reading design file: " << designFile << endl;
    m_jacobian = fabber::read_matrix_file(designFile);
    if (args.GetBool("add-ones-regressor"))
    {
        ColumnVector ones(m_jacobian.Nrows());
        ones = 1.0;
        m_jacobian = m_jacobian | ones;
    }
    LOG << "LinearFwdModel::Loaded " << m_jacobian.Ncols() << " basis functions of length " << m_jacobian.Nrows() << endl;
    m_centre.ReSize(m_jacobian.Ncols());
    m_centre = 0;
    m_offset.ReSize(m_jacobian.Nrows());
    m_offset = 0;
}

void LinearFwdModel::GetParameterDefaults(std::vector<Parameter> &params) const
{
    params.clear();
    for (int i = 0; i < m_centre.Nrows(); i++)
        params.push_back(Parameter(i, "Parameter_" + stringify(i + 1), DistParams(0, 1e12), DistParams(0, 1e12)));
}

void LinearFwdModel::EvaluateModel(const ColumnVector &params, ColumnVector &result, const std::string &) const
{
    result = m_jacobian * (params - m_centre) + m_offset;
}

bool LinearFwdModel::GetDeviceModel(DeviceModelSpec &spec) const
{
    // valid for the plain design-matrix model only (centre = offset = 0)
    if (!m_centre.IsZero() || !m_offset.IsZero())
        return false;
    spec.model = FVB_MODEL_LINEAR;
    spec.design = m_jacobian;
    return true;
}

ReturnMatrix LinearFwdModel::Jacobian() const
{
    return m_jacobian;
}
ReturnMatrix LinearFwdModel::Centre() const
{
    return m_centre;
}
ReturnMatrix LinearFwdModel::Offset() const
{
    return m_offset;
}

LinearizedFwdModel::LinearizedFwdModel(const FwdModel *model)
    : m_model(model)
{
    SetLogger(model->GetLogger());
}

LinearizedFwdModel::LinearizedFwdModel(const LinearizedFwdModel &from)
    : LinearFwdModel(from)
    , m_model(from.m_model)
{
    SetLogger(from.GetLogger());
}

static bool all_finite(const Matrix &m)
{
    const double *p = m.Store();
    for (int k = 0; k < m.Storage(); k++)
        if (!((p[k] - p[k]) == 0.0))
            return false;
    return true;
}

// Host re-linearisation for models WITHOUT a device body: the model's own EvaluateModel has to
// be called on the host, 1 + 2P times (fwdmodel_linear.cc:126-182).
void LinearizedFwdModel::ReCentre(const ColumnVector &about)
{
    m_centre = about;
    m_model->EvaluateFabber(m_centre, m_offset);
    if (!all_finite(m_offset))
        throw FabberInternalError("LinearizedFwdModel::ReCentre: Non-finite values found in offset");
    const int P = m_centre.Nrows();
    m_jacobian.ReSize(m_offset.Nrows(), P);
    ColumnVector plus, minus, fplus, fminus;
    for (int i = 1; i <= P; i++)
    {
        double step = m_centre(i) * 1e-5;
        if (step < 0)
            step = -step;
        if (step < 1e-10)
            step = 1e-10;
        plus = m_centre;
        minus = m_centre;
        plus(i) += step;
        minus(i) -= step;
        m_model->EvaluateFabber(plus, fplus);
        m_model->EvaluateFabber(minus, fminus);
        m_jacobian.Column(i) = (fplus - fminus) / (plus(i) - minus(i));
    }
    if (!all_finite(m_jacobian))
        throw FabberInternalError("LinearizedFwdModel::ReCentre: Non-finite values found in jacobian");
}
