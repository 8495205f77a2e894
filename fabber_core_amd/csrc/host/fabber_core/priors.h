/* priors.h - prior type codes and the prior-type string grammar (reference: priors.h:22-155,
 * priors.cc:35-92). The priors themselves are applied inside the HIP kernels; the host only
 * resolves their description (type, mean, precision, image) per parameter. */
#pragma once

#include "rundata.h"

#include <string>

const char PRIOR_NORMAL = 'N';
const char PRIOR_IMAGE = 'I';
const char PRIOR_ARD = 'A';
const char PRIOR_SPATIAL_M = 'M';
const char PRIOR_SPATIAL_m = 'm';
const char PRIOR_SPATIAL_P = 'P';
const char PRIOR_SPATIAL_p = 'p';
const char PRIOR_DEFAULT = '-';

class Prior
{
public:
    /** "A+" style option string -> one type char per parameter ('+' repeats the previous type,
     *  missing entries become '-' = model default). Throws InvalidOptionValue. */
    static std::string ExpandPriorTypesString(std::string priors_str, unsigned int num_params);
    /** engine code (enum fvb_prior) for a prior type char; throws InvalidOptionValue */
    static int DeviceCode(char prior_type);
};
