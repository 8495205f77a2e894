/* inference.h - inference-technique plugin interface (reference: inference.h:26-166). */
#pragma once

#include "dist_mvn.h"
#include "easylog.h"
#include "factories.h"
#include "fwdmodel.h"
#include "noisemodel.h"
#include "rundata.h"

#include <string>
#include <vector>

struct fvb_config;

class InferenceTechnique : public Loggable
{
public:
    static std::vector<std::string> GetKnown();
    static InferenceTechnique *NewFromName(const std::string &name);
    static void UsageFromName(const std::string &name, std::ostream &stream);

    InferenceTechnique();
    virtual ~InferenceTechnique();
    virtual void GetOptions(std::vector<OptionSpec> &opts) const {};
    virtual std::string GetDescription() const = 0;
    virtual std::string GetVersion() const = 0;
    virtual void Initialize(FwdModel *fwd_model, FabberRunData &rundata);
    virtual void DoCalculations(FabberRunData &rundata) = 0;
    virtual void SaveResults(FabberRunData &rundata) const;

protected:
    /** finalMVN, mean_/std_/zstat_/var_<param>, modelfit, residuals, noise images and model extras
     *  from m_result_image (inference.cc:112-281), computed by the engine's post-processing kernel.
     *  n_noise = noise entries in the MVN, n_noise_saved = how many go to noise_means/_stdevs. */
    void SaveEngineResults(FabberRunData &rundata, const fvb_config &cfg, const std::vector<Parameter> &params,
        int n_noise, int n_noise_saved, bool host_model) const;

    FwdModel *m_model;
    int m_num_params;
    bool m_halt_bad_voxel;
    /** Per-voxel result MVNs. The MI355X techniques keep their results as one packed
     *  rows x voxels image (m_result_image) and only materialise this vector on request. */
    std::vector<MVNDist *> resultMVNs;
    NEWMAT::Matrix m_result_image;
    std::vector<int> m_masked_tpoints;
    bool m_debug;
};

typedef SingletonFactory<InferenceTechnique> InferenceTechniqueFactory;
