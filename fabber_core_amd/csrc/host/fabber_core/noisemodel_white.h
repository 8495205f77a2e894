/* noisemodel_white.h - white noise with an optional repeating pattern of independent noise
 * precisions (reference: noisemodel_white.h:22-95). */
#pragma once

#include "dist_gamma.h"
#include "noisemodel.h"

#include <string>
#include <vector>

class WhiteParams : public NoiseParams
{
public:
    explicit WhiteParams(int N);
    WhiteParams(const WhiteParams &from);
    virtual WhiteParams *Clone() const;
    virtual const WhiteParams &operator=(const NoiseParams &in);
    virtual const MVNDist OutputAsMVN() const;
    virtual void InputFromMVN(const MVNDist &mvn);
    virtual void Dump(std::ostream &os) const;

    const int nPhis;
    std::vector<GammaDist> phis;
};

class WhiteNoiseModel : public NoiseModel
{
public:
    static NoiseModel *NewInstance();
    virtual void Initialize(FabberRunData &args);
    virtual WhiteParams *NewParams() const;
    virtual void HardcodedInitialDists(NoiseParams &prior, NoiseParams &posterior) const;
    virtual int NumParams();
    virtual void ConfigureEngine(fvb_config &cfg, int n_times, std::vector<unsigned char> &phi_index) const;

protected:
    /** pattern string -> 0-based noise-parameter index for each of n_times samples */
    std::vector<int> ExpandPattern(int n_times) const;
    std::string phiPattern;
    int m_num_phis;
    double lockedNoiseStdev;
    double phiprior;
};
