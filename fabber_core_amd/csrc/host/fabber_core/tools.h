/* tools.h - small numerical / file helpers (reference: tools.h, tools.cc:27-98) */
#pragma once

#include "armawrap/newmat.h"

#include <string>

namespace fabber
{
/** Read a VEST (/NumWaves /NumPoints /Matrix) or plain ASCII ('#' comments) matrix file */
NEWMAT::Matrix read_matrix_file(const std::string &filename);
}
double gammaln(double x);
double digamma_fp64(double x);
