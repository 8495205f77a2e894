// noisemodel.cc - noise-model base. The update equations live in the HIP kernels; the host
// versions of the reference's virtuals exist for interface compatibility only and refuse to run.
#include "noisemodel.h"

using namespace std;

NoiseModel *NoiseModel::NewFromName(const string &name)
{
    NoiseModel *noise = NoiseModelFactory::GetInstance()->Create(name);
    if (!noise)
        throw InvalidOptionValue("noise", name, "Unrecognized noise type");
    return noise;
}

void NoiseModel::Initialize(FabberRunData &rundata)
{
    m_log = rundata.GetLogger();
    m_masked_tpoints = rundata.GetIntList("mt", 1); // 1-based, noisemodel.cc:34-40
}

static void no_host_path(const char *what)
{
    throw FabberInternalError(string("NoiseModel::") + what
        + " has no host implementation in this library: the VB updates run in the MI355X engine (fabber_vb_run_*)");
}

void NoiseModel::UpdateNoise(
    NoiseParams &, const NoiseParams &, const MVNDist &, const LinearFwdModel &, const NEWMAT::ColumnVector &) const
{
    no_host_path("UpdateNoise");
}
void NoiseModel::UpdateTheta(const NoiseParams &, MVNDist &, const MVNDist &, const LinearFwdModel &,
    const NEWMAT::ColumnVector &, MVNDist *, float) const
{
    no_host_path("UpdateTheta");
}
double NoiseModel::CalcFreeEnergy(const NoiseParams &, const NoiseParams &, const MVNDist &, const MVNDist &,
    const LinearFwdModel &, const NEWMAT::ColumnVector &) const
{
    no_host_path("CalcFreeEnergy");
    return 0;
}
