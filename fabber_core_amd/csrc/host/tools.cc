// tools.cc - matrix file reader and scalar special functions (host twins of vb_math.h)
#include "tools.h"

#include "rundata.h"

#include "../vb_math.h"

#include <fstream>
#include <sstream>

namespace fabber
{
NEWMAT::Matrix read_matrix_file(const std::string &filename)
{
    std::ifstream in(filename.c_str());
    if (!in.good())
        throw InvalidOptionValue("matrix file", filename, "Could not open file");
    std::vector<std::vector<double> > rows;
    std::string line;
    bool vest = false, in_matrix = false;
    while (std::getline(in, line))
    {
        size_t first = line.find_first_not_of(" \t\r");
        if (first == std::string::npos || line[first] == '#')
            continue;
        if (line[first] == '/')
        {
            // VEST header keywords; data start after /Matrix
            vest = true;
            if (line.compare(first, 7, "/Matrix") == 0)
                in_matrix = true;
            continue;
        }
        if (vest && !in_matrix)
            continue;
        std::istringstream is(line);
        std::vector<double> row;
        double v;
        while (is >> v)
            row.push_back(v);
        if (!row.empty())
            rows.push_back(row);
    }
    if (rows.empty())
        throw InvalidOptionValue("matrix file", filename, "No numeric data found");
    NEWMAT::Matrix m((int)rows.size(), (int)rows[0].size());
    for (size_t r = 0; r < rows.size(); r++)
    {
        if (rows[r].size() != rows[0].size())
            throw InvalidOptionValue("matrix file", filename, "Rows have different lengths");
        for (size_t c = 0; c < rows[r].size(); c++)
            m((int)r + 1, (int)c + 1) = rows[r][c];
    }
    return m;
}
}

double gammaln(double x)
{
    return fvb::gammaln(x);
}
double digamma_fp64(double x)
{
    return fvb::digamma(x);
}
