/* inference_nlls.h - "nlls" inference technique: non-linear least squares on the MI355X engine
 * (reference: inference_nlls.h, NLLSInferenceTechnique). */
#pragma once

#include "inference.h"

#include <string>
#include <vector>

class NLLSInferenceTechnique : public InferenceTechnique
{
public:
    static InferenceTechnique *NewInstance();
    NLLSInferenceTechnique();
    virtual ~NLLSInferenceTechnique();
    virtual void GetOptions(std::vector<OptionSpec> &opts) const;
    virtual std::string GetDescription() const;
    virtual std::string GetVersion() const;
    virtual void Initialize(FwdModel *fwd_model, FabberRunData &rundata);
    virtual void DoCalculations(FabberRunData &rundata);
    virtual void SaveResults(FabberRunData &rundata) const;

protected:
    /** Starting estimate: the means of the model's HardcodedInitialDists posterior, or of the
     *  MVN named by fwd-inital-posterior (inference_nlls.cc:68-82) */
    MVNDist *initialFwdPosterior;
    bool m_vbinit;
    bool m_lm;
    std::vector<int> m_status;

    struct EngineStorage;
    EngineStorage *m_store;
};
