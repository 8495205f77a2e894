#include "dist_gamma.h"

#include <math.h>

void GammaDist::Dump(std::ostream &os) const
{
    os << "Noise stdev == " << 1.0 / sqrt(b * c) << " (b==" << b << ", c==" << c << ")" << std::endl;
}
