// priors.cc - prior-type string grammar (host). Behaviour pinned by the reference's
// test/test_priors.cc:13-142, re-expressed in tests/test_host_boundary.py.
#include "priors.h"

#include "../../../include/fabber_vb.h"

#include <algorithm>

std::string Prior::ExpandPriorTypesString(std::string spec, unsigned int num_params)
{
    // The single '+' (if any) stands for "as many copies of the preceding type as needed"
    const size_t n_plus = std::count(spec.begin(), spec.end(), '+');
    if (n_plus > 1)
        throw InvalidOptionValue("param-spatial-priors", spec, "Only one + character allowed");
    const size_t n_given = spec.size() - n_plus;
    if (n_given > num_params)
        throw InvalidOptionValue("param-spatial-priors", spec, "Too many parameters");

    const size_t plus_at = spec.find('+');
    const char repeat = (plus_at == std::string::npos || plus_at == 0) ? PRIOR_DEFAULT : spec[plus_at - 1];
    std::string out;
    if (plus_at == std::string::npos)
    {
        out = spec;
        out.append(num_params - n_given, PRIOR_DEFAULT);
    }
    else
    {
        out = spec.substr(0, plus_at);
        out.append(num_params - n_given, repeat);
        out += spec.substr(plus_at + 1);
    }
    return out;
}

int Prior::DeviceCode(char prior_type)
{
    switch (prior_type)
    {
    case PRIOR_NORMAL:
    case PRIOR_DEFAULT:
        return FVB_PRIOR_NORMAL;
    case PRIOR_IMAGE:
        return FVB_PRIOR_IMAGE;
    case PRIOR_ARD:
        return FVB_PRIOR_ARD;
    case PRIOR_SPATIAL_M:
        return FVB_PRIOR_SPATIAL_M;
    case PRIOR_SPATIAL_m:
        return FVB_PRIOR_SPATIAL_m;
    case PRIOR_SPATIAL_P:
        return FVB_PRIOR_SPATIAL_P;
    case PRIOR_SPATIAL_p:
        return FVB_PRIOR_SPATIAL_p;
    default:
        throw InvalidOptionValue("Prior type", stringify(prior_type), "Supported types: NMmPpAI");
    }
}
