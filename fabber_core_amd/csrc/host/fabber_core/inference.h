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
