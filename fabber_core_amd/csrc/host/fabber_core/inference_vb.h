/* inference_vb.h - "vb" inference technique: the host driver of the MI355X voxelwise VB engine
 * (reference: inference_vb.h, Vb::Initialize/DoCalculations/SaveResults). */
#pragma once

#include "inference.h"

#include <memory>
#include <string>
#include <vector>

struct fvb_config;

class Vb : public InferenceTechnique
{
public:
    static InferenceTechnique *NewInstance();
    Vb();
    virtual ~Vb();
    virtual void GetOptions(std::vector<OptionSpec> &opts) const;
    virtual std::string GetDescription() const;
    virtual std::string GetVersion() const;
    virtual void Initialize(FwdModel *fwd_model, FabberRunData &rundata);
    virtual void DoCalculations(FabberRunData &rundata);
    virtual void SaveResults(FabberRunData &rundata) const;

protected:
    bool IsSpatial(FabberRunData &rundata) const;
    /** Resolve model / priors / noise / convergence options into the engine's problem block */
    void BuildEngineConfig(FabberRunData &rundata, fvb_config &cfg);
    /** Models evaluated on the host: initial posterior of every voxel as an MVN image, and the
     *  re-centre callback the engine calls (fvb_linearise_fn) */
    void BuildInitialMvn(FabberRunData &rundata, fvb_config &cfg);
    static int32_t LineariseCallback(void *user, int32_t n_active, const int32_t *ids, const double *means, double *lin);

    NoiseModel *m_noise;
    int m_noise_params;
    bool m_saveF, m_saveFsHistory, m_printF, m_needF;
    bool m_locked_linear;
    int m_nvoxels;

    // results of the last DoCalculations (host copies of the engine's outputs)
    std::vector<double> m_free_energy;
    NEWMAT::Matrix m_f_history;
    std::vector<int> m_status;

    // storage the engine's problem block points into
    struct EngineStorage;
    EngineStorage *m_store;
};
