/* dist_mvn.h - multivariate normal with lazily inverted precision / covariance.
 * Public surface of the reference's MVNDist (dist_mvn.h:22-226): `means` is a public
 * ColumnVector that model code writes to; Get/Set Precisions/Covariance flip validity flags and
 * invert on demand with the 1e-10 ridge retry; Load/Save use the packed per-voxel image layout
 * (lower triangle row-major, means, 1) that is also the HIP kernels' output layout. */
#pragma once

#include "easylog.h"
#include "rundata.h"

#include "armawrap/newmat.h"

#include <string>
#include <vector>

class MVNDist : public Loggable
{
public:
    explicit MVNDist(EasyLog *log = 0);
    explicit MVNDist(int dim, EasyLog *log = 0);
    MVNDist(const MVNDist &from);
    /** block-diagonal concatenation */
    MVNDist(const MVNDist &from1, const MVNDist &from2);
    /** from a (n+1)x(n+1) matrix file [cov means; means' 1] */
    MVNDist(const std::string filename, EasyLog *log = 0);

    MVNDist &operator=(const MVNDist &from);
    void CopyFromSubmatrix(const MVNDist &from, int first, int last, bool checkIndependence = true);
    MVNDist GetSubmatrix(int first, int last, bool checkIndependence = true);

    void SetSize(int dim);
    int GetSize() const;

    NEWMAT::ColumnVector means;

    const NEWMAT::SymmetricMatrix &GetPrecisions() const;
    const NEWMAT::SymmetricMatrix &GetCovariance() const;
    void SetPrecisions(const NEWMAT::SymmetricMatrix &from);
    void SetCovariance(const NEWMAT::SymmetricMatrix &from);

    void Dump(std::ostream &out) const;
    void LoadFromMatrix(const std::string &filename);

    static void Load(std::vector<MVNDist *> &mvns, const std::string &filename, FabberRunData &data, EasyLog *log = 0);
    static void Load(std::vector<MVNDist *> &mvns, NEWMAT::Matrix &voxel_data, EasyLog *log = 0);
    static void Save(const std::vector<MVNDist *> &mvns, const std::string &filename, FabberRunData &data);

    /** Pack into / unpack from one column of the per-voxel image (rows = n(n+1)/2 + n + 1) */
    void PackInto(NEWMAT::Matrix &image, int column) const;
    void UnpackFrom(const NEWMAT::Matrix &image, int column, int n_params);

protected:
    int m_size;

private:
    mutable NEWMAT::SymmetricMatrix precisions;
    mutable NEWMAT::SymmetricMatrix covariance;
    mutable bool precisionsValid;
    mutable bool covarianceValid;
};

inline std::ostream &operator<<(std::ostream &out, const MVNDist &dist)
{
    dist.Dump(out);
    return out;
}
