/* noisemodel_ar.h - AR(1) noise model, host description (reference: noisemodel_ar.h).
 * num-echoes = 1 or 2, ar1-cross-terms = none / same / dual; the updates themselves run in the HIP
 * kernels (../../vb_lane_ar_kernel.h for one echo, ../../vb_wave_ar_kernel.h for every case). */
#pragma once

#include "dist_gamma.h"
#include "noisemodel.h"

#include <string>
#include <vector>

class Ar1cParams : public NoiseParams
{
public:
    Ar1cParams(int nAlpha, int nPhi);
    Ar1cParams(const Ar1cParams &from);
    virtual Ar1cParams *Clone() const;
    virtual const Ar1cParams &operator=(const NoiseParams &in);
    virtual const MVNDist OutputAsMVN() const;
    virtual void InputFromMVN(const MVNDist &mvn);
    virtual void Dump(std::ostream &os) const;

    MVNDist alpha;
    std::vector<GammaDist> phis;
};

class Ar1cNoiseModel : public NoiseModel
{
public:
    static NoiseModel *NewInstance();
    virtual void Initialize(FabberRunData &args);
    virtual Ar1cParams *NewParams() const;
    virtual void HardcodedInitialDists(NoiseParams &prior, NoiseParams &posterior) const;
    /** nPhis, as in the reference (noisemodel_ar.cc:362-365), although the result MVN carries
     *  nAlphas + nPhis noise entries: Vb::SaveResults therefore writes alpha_1 as "noise_means" */
    virtual int NumParams();
    virtual int NumOutputParams();
    virtual void ConfigureEngine(fvb_config &cfg, int n_times, std::vector<unsigned char> &phi_index) const;

protected:
    int NumAlphas() const;
    std::string ar1Type;
    int nPhis;
};
