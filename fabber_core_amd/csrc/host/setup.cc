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

#include "armawrap/newmat.h"
#include "../../../include/fabber_vb.h"

#include <cstdlib>

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
// FVB_HOST_PINNED_IMAGES=1: the volumes of a run (blocks of 8 MB and more: newmat.h) are page-locked when they are created and
// unlocked before they are freed, by the library that owns the GPU (BigBlockHooks). Off by default: the copies of the
// pipelined engine call run at the same rate from pageable memory (bench.py: 16.6 ms either way on C3).
static void pin_block(void *p, std::size_t bytes)
{
    if (fabber_vb_device_count() > 0)
        (void)fabber_vb_pin_host_buffer(p, (uint64_t)bytes);
}
static void unpin_block(void *p, std::size_t)
{
    if (fabber_vb_device_count() > 0)
        (void)fabber_vb_unpin_host_buffer(p);
}

void FabberSetup::SetupDefaults()
{
    if (getenv("FVB_HOST_PINNED_IMAGES"))
    {
        NEWMAT::BigBlockHooks::created() = &pin_block;
        NEWMAT::BigBlockHooks::dying() = &unpin_block;
    }
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
