/* dist_gamma.h - Gamma(scale b, shape c) distribution of a noise precision (dist_gamma.h:14-34) */
#pragma once

#include "easylog.h"

#include <ostream>

class GammaDist : public Loggable
{
public:
    explicit GammaDist(EasyLog *log = 0)
        : Loggable(log)
        , b(0)
        , c(0)
    {
    }
    double b;
    double c;
    double CalcMean() const
    {
        return b * c;
    }
    double CalcVariance() const
    {
        return b * b * c;
    }
    void SetMeanVariance(double m, double v)
    {
        b = v / m;
        c = m / b;
    }
    void Dump(std::ostream &os) const;
};
