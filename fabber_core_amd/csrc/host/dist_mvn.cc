// dist_mvn.cc - MVN value type with lazy inversion. Host-side only (initial distributions,
// result packing, model-space conversion); the per-voxel algebra of the VB loop runs in the
// HIP kernels on the packed form. Semantics follow the reference's MVNDist (dist_mvn.cc).
#include "dist_mvn.h"

#include "tools.h"

#include <math.h>

using namespace NEWMAT;
using std::string;
using std::vector;

MVNDist::MVNDist(EasyLog *log)
    : Loggable(log)
    , m_size(-1)
    , precisionsValid(false)
    , covarianceValid(false)
{
}

MVNDist::MVNDist(int dim, EasyLog *log)
    : Loggable(log)
    , m_size(-1)
    , precisionsValid(false)
    , covarianceValid(false)
{
    SetSize(dim);
}

MVNDist::MVNDist(const MVNDist &from)
    : Loggable(from.m_log)
    , m_size(-1)
    , precisionsValid(false)
    , covarianceValid(false)
{
    *this = from;
}

MVNDist::MVNDist(const string filename, EasyLog *log)
    : Loggable(log)
    , m_size(-1)
    , precisionsValid(false)
    , covarianceValid(false)
{
    LoadFromMatrix(filename);
}

// Block-diagonal join. Covariances (not precisions) are copied so that exact zeros between the
// blocks stay exact; an un-invertible block contributes zeros (dist_mvn.cc:57-100).
MVNDist::MVNDist(const MVNDist &a, const MVNDist &b)
    : Loggable(a.m_log)
    , m_size(-1)
    , precisionsValid(false)
    , covarianceValid(false)
{
    SetSize(a.m_size + b.m_size);
    means = a.means & b.means;
    SymmetricMatrix cov(m_size);
    cov = 0;
    const MVNDist *parts[2] = { &a, &b };
    int off = 0;
    for (int p = 0; p < 2; p++)
    {
        const int n = parts[p]->m_size;
        try
        {
            const SymmetricMatrix &c = parts[p]->GetCovariance();
            for (int r = 1; r <= n; r++)
                for (int q = 1; q <= r; q++)
                    cov(off + r, off + q) = c(r, q);
        }
        catch (Exception &)
        {
        }
        off += n;
    }
    SetCovariance(cov);
}

MVNDist &MVNDist::operator=(const MVNDist &from)
{
    if (&from == this)
        return *this;
    m_log = from.m_log;
    if (from.m_size == -1)
    {
        m_size = -1;
        precisionsValid = covarianceValid = false;
        return *this;
    }
    SetSize(from.m_size);
    means = from.means;
    precisionsValid = from.precisionsValid;
    covarianceValid = from.covarianceValid;
    if (precisionsValid)
        precisions = from.precisions;
    if (covarianceValid)
        covariance = from.covariance;
    return *this;
}

MVNDist MVNDist::GetSubmatrix(int first, int last, bool checkIndependence)
{
    MVNDist ret;
    ret.CopyFromSubmatrix(*this, first, last, checkIndependence);
    return ret;
}

void MVNDist::CopyFromSubmatrix(const MVNDist &from, int first, int last, bool checkIndependence)
{
    SetSize(last - first + 1);
    means = ((const ColumnVector &)from.means).Rows(first, last);
    precisionsValid = from.precisionsValid;
    covarianceValid = from.covarianceValid;
    if (precisionsValid)
        precisions = ((const SymmetricMatrix &)from.precisions).SymSubMatrix(first, last);
    if (covarianceValid)
        covariance = ((const SymmetricMatrix &)from.covariance).SymSubMatrix(first, last);
    if (checkIndependence)
    {
        const SymmetricMatrix &c = from.GetCovariance();
        for (int r = first; r <= last; r++)
            for (int q = 1; q <= from.m_size; q++)
                if ((q < first || q > last) && c(r, q) != 0.0)
                    throw FabberRunDataError(
                        "Covariance found in part of MVN that should be independent from the rest!");
    }
}

int MVNDist::GetSize() const
{
    return m_size;
}

void MVNDist::SetSize(int dim)
{
    if (dim <= 0)
        throw FabberInternalError("MVNDist::SetSize dim<=0");
    if (m_size != dim)
    {
        m_size = dim;
        means.ReSize(dim);
        means = 0;
        precisions = IdentityMatrix(dim);
        covariance = IdentityMatrix(dim);
    }
    precisionsValid = true;
    covarianceValid = true;
}

// x -> x^-1, retrying once with 1e-10 added to the diagonal if exactly singular
static SymmetricMatrix invert_with_ridge(const SymmetricMatrix &m, EasyLog *m_log, int size)
{
    try
    {
        return m.i();
    }
    catch (Exception &)
    {
        WARN_ONCE("MVN precision (m_size==" + stringify(size) + ") was singular, adding 1e-10 to diagonal");
        SymmetricMatrix ridge(m);
        for (int k = 1; k <= size; k++)
            ridge(k, k) += 1e-10;
        return ridge.i();
    }
}

const SymmetricMatrix &MVNDist::GetPrecisions() const
{
    if (m_size == -1)
        throw FabberInternalError("MVNDist::GetPrecisions size = -1 (uninitialized)");
    if (!precisionsValid)
    {
        precisions = invert_with_ridge(covariance, m_log, m_size);
        precisionsValid = true;
    }
    return precisions;
}

const SymmetricMatrix &MVNDist::GetCovariance() const
{
    if (m_size == -1)
        throw FabberInternalError("MVNDist::GetCovariance size = -1 (uninitialized)");
    if (!covarianceValid)
    {
        covariance = invert_with_ridge(precisions, m_log, m_size);
        covarianceValid = true;
    }
    return covariance;
}

void MVNDist::SetPrecisions(const SymmetricMatrix &from)
{
    if (from.Nrows() != m_size)
        throw FabberInternalError("MVNDist::SetPrecisions size mismatch");
    precisions = from;
    precisionsValid = true;
    covarianceValid = false;
}

void MVNDist::SetCovariance(const SymmetricMatrix &from)
{
    if (from.Nrows() != m_size)
        throw FabberInternalError("MVNDist::SetCovariance size mismatch");
    covariance = from;
    covarianceValid = true;
    precisionsValid = false;
}

void MVNDist::LoadFromMatrix(const string &filename)
{
    Matrix mat = fabber::read_matrix_file(filename);
    const int N = mat.Nrows() - 1;
    if (N < 1 || mat.Ncols() != N + 1 || mat != mat.t() || mat(N + 1, N + 1) != 1.0)
        throw InvalidOptionValue(
            filename, "", "MVNs must be symmetric matrices (format = [covariance means(:); means(:) 1.0])");
    SetSize(N);
    for (int k = 1; k <= N; k++)
        means(k) = mat(k, N + 1);
    SymmetricMatrix sym(N);
    for (int r = 1; r <= N; r++)
        for (int c = 1; c <= r; c++)
            sym(r, c) = mat(r, c);
    SetCovariance(sym);
}

// ---- packed per-voxel image: n(n+1)/2 lower-triangle rows (row-major), n means, 1.0 ----
void MVNDist::PackInto(Matrix &image, int column) const
{
    const SymmetricMatrix &cov = GetCovariance();
    int row = 1;
    for (int r = 1; r <= m_size; r++)
        for (int c = 1; c <= r; c++)
            image(row++, column) = cov(r, c);
    for (int k = 1; k <= m_size; k++)
        image(row++, column) = means(k);
    image(row, column) = 1.0;
}

void MVNDist::UnpackFrom(const Matrix &image, int column, int n)
{
    SetSize(n);
    SymmetricMatrix cov(n);
    int row = 1;
    for (int r = 1; r <= n; r++)
        for (int c = 1; c <= r; c++)
            cov(r, c) = image(row++, column);
    for (int k = 1; k <= n; k++)
        means(k) = image(row++, column);
    if (image(row, column) != 1)
        throw FabberRunDataError("MVNDist::Load - Voxel data does not contain a valid MVN - last value != 1");
    SetCovariance(cov);
}

void MVNDist::Load(vector<MVNDist *> &mvns, const string &filename, FabberRunData &data, EasyLog *log)
{
    Matrix voxel_data = data.GetVoxelData(filename);
    MVNDist::Load(mvns, voxel_data, log);
}

void MVNDist::Load(vector<MVNDist *> &mvns, Matrix &voxel_data, EasyLog *log)
{
    const int nVoxels = voxel_data.Ncols();
    if (nVoxels == 0)
        throw FabberRunDataError("MVNDist::Load - Voxel data is empty");
    // rows = n(n+1)/2 + n + 1  =>  n = (sqrt(8 rows + 1) - 3) / 2
    const int n = ((int)sqrt(double(8 * voxel_data.Nrows() + 1)) - 3) / 2;
    if (voxel_data.Nrows() != n * (n + 1) / 2 + n + 1)
        throw FabberRunDataError("MVNDist::Load  - Incorrect number of rows for an MVN input");
    mvns.assign(nVoxels, (MVNDist *)NULL);
    for (int v = 1; v <= nVoxels; v++)
    {
        mvns[v - 1] = new MVNDist(log);
        mvns[v - 1]->UnpackFrom(voxel_data, v, n);
    }
}

void MVNDist::Save(const vector<MVNDist *> &mvns, const string &filename, FabberRunData &data)
{
    const int nVoxels = (int)mvns.size();
    const int n = nVoxels ? mvns[0]->means.Nrows() : 0;
    Matrix vols(n * (n + 1) / 2 + n + 1, nVoxels);
    for (int v = 1; v <= nVoxels; v++)
        mvns[v - 1]->PackInto(vols, v);
    data.SaveVoxelData(filename, vols, VDT_MVN);
}

void MVNDist::Dump(std::ostream &out) const
{
    out << "MVNDist, with m_size == " << m_size << ", precisionsValid == " << precisionsValid
        << ", covarianceValid == " << covarianceValid << std::endl;
    out << "  Means: " << means.t();
    if (precisionsValid || covarianceValid)
    {
        out << "  Covariance matrix:" << std::endl;
        const SymmetricMatrix &c = GetCovariance();
        for (int i = 1; i <= m_size; i++)
            out << "  " << ((const Matrix &)c).Row(i);
    }
    else
        out << "  Covariance undefined." << std::endl;
}
