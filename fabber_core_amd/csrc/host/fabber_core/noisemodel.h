/* noisemodel.h - noise-model plugin interface (reference: noisemodel.h:26-187).
 *
 * The per-voxel update equations (UpdateTheta / UpdateNoise / CalcFreeEnergy) run inside the
 * HIP kernels; a NoiseModel here is the HOST description of the noise model: option parsing,
 * initial distributions, and ConfigureEngine(), which writes that description into the engine's
 * problem block. The reference's update virtuals keep their signatures; their base
 * implementations throw, because this library has no CPU path to fall back on. */
#pragma once

#include "dist_mvn.h"
#include "factories.h"
#include "fwdmodel_linear.h"
#include "rundata.h"

#include "armawrap/newmat.h"

#include <ostream>
#include <string>
#include <vector>

struct fvb_config;

class NoiseParams
{
public:
    virtual ~NoiseParams()
    {
    }
    virtual NoiseParams *Clone() const = 0;
    virtual const NoiseParams &operator=(const NoiseParams &in) = 0;
    virtual const MVNDist OutputAsMVN() const = 0;
    virtual void InputFromMVN(const MVNDist &mvn) = 0;
    virtual void Dump(std::ostream &os) const = 0;
};

class NoiseModel : public Loggable
{
public:
    static NoiseModel *NewFromName(const std::string &name);
    virtual ~NoiseModel()
    {
    }
    virtual void Initialize(FabberRunData &args);
    virtual NoiseParams *NewParams() const = 0;
    virtual void HardcodedInitialDists(NoiseParams &prior, NoiseParams &posterior) const = 0;
    virtual void Precalculate(
        NoiseParams &noise, const NoiseParams &noisePrior, const NEWMAT::ColumnVector &sampleData) const
    {
    }
    virtual void UpdateNoise(NoiseParams &noise, const NoiseParams &noisePrior, const MVNDist &theta,
        const LinearFwdModel &model, const NEWMAT::ColumnVector &data) const;
    virtual void UpdateTheta(const NoiseParams &noise, MVNDist &theta, const MVNDist &thetaPrior,
        const LinearFwdModel &model, const NEWMAT::ColumnVector &data, MVNDist *thetaWithoutPrior = NULL,
        float LMalpha = 0) const;
    virtual double CalcFreeEnergy(const NoiseParams &noise, const NoiseParams &noisePrior, const MVNDist &theta,
        const MVNDist &thetaPrior, const LinearFwdModel &model, const NEWMAT::ColumnVector &data) const;
    virtual int NumParams() = 0;

    /** Number of entries this noise model contributes to the result MVN */
    virtual int NumOutputParams()
    {
        return NumParams();
    }
    /** MI355X: write the noise model into the engine's problem description for n_times samples.
     *  phi_index receives the per-timepoint noise-parameter index (255 = masked timepoint). */
    virtual void ConfigureEngine(fvb_config &cfg, int n_times, std::vector<unsigned char> &phi_index) const = 0;

protected:
    std::vector<int> m_masked_tpoints;
};

typedef SingletonFactory<NoiseModel> NoiseModelFactory;
