// setup.cc - registration of built-in components under the reference's names (setup.cc:26-58)
#include "setup.h"

#include "convergence.h"
#include "fwdmodel.h"
#include "fwdmodel_exp.h"
#include "fwdmodel_linear.h"
#include "fwdmodel_poly.h"
#include "inference.h"
#include "inference_nlls.h"
#include "inference_vb.h"
#include "noisemodel.h"
#include "noisemodel_white.h"
#include "noisemodel_ar.h"


void FabberSetup::SetupDefaultInferenceTechniques()
{
    InferenceTechniqueFactory *f = InferenceTechniqueFactory::GetInstance();
    f->Add("vb", &Vb::NewInstance);
    f->Add("spatialvb", &Vb::NewInstance);
    f->Add("nlls", &NLLSInferenceTechnique::NewInstance);
}
void FabberSetup::SetupDefaultNoiseModels()
{
    NoiseModelFactory::GetInstance()->Add("white", &WhiteNoiseModel::NewInstance);
    NoiseModelFactory::GetInstance()->Add("ar", &Ar1cNoiseModel::NewInstance);
}
void FabberSetup::SetupDefaultFwdModels()
{
    FwdModelFactory *f = FwdModelFactory::GetInstance();
    f->Add("linear", &LinearFwdModel::NewInstance);
    f->Add("poly", &PolynomialFwdModel::NewInstance);
    if (!f->HasName("exp")) // a loaded plugin may already provide its own "exp"
        f->Add("exp", &ExpFwdModel::NewInstance);
}
void FabberSetup::SetupDefaultConvergenceDetectors()
{
    ConvergenceDetectorFactory *f = ConvergenceDetectorFactory::GetInstance();
    f->Add("maxits", &ConvergenceDetector::NewMaxIts);
    f->Add("pointzeroone", &ConvergenceDetector::NewFchange);
    f->Add("freduce", &ConvergenceDetector::NewFreduce);
    f->Add("trialmode", &ConvergenceDetector::NewTrialMode);
    f->Add("lm", &ConvergenceDetector::NewLM);
}
void FabberSetup::SetupDefaults()
{
    SetupDefaultInferenceTechniques();
    SetupDefaultNoiseModels();
    SetupDefaultFwdModels();
    SetupDefaultConvergenceDetectors();
}
void FabberSetup::Destroy()
{
    FwdModelFactory::Destroy();
    NoiseModelFactory::Destroy();
    InferenceTechniqueFactory::Destroy();
    ConvergenceDetectorFactory::Destroy();
}
