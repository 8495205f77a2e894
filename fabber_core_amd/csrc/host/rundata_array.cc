// rundata_array.cc - flat float arrays <-> per-voxel matrices for the C ABI.
// Volume order is x fastest, then y, z, and the 4th (time / row) dimension slowest
// (fabber_capi.h:100-105); voxel index = rank of the voxel among the non-zero mask entries in
// that scan (reference: rundata_array.cc:23-133).
#include "rundata_array.h"

using NEWMAT::Matrix;

void FabberRunDataArray::SetExtent(int nx, int ny, int nz, const int *mask)
{
    if (nx <= 0 || ny <= 0 || nz <= 0 || !mask)
        throw FabberRunDataError("SetExtent: dimensions must be positive and a mask must be given");
    FabberRunData::SetExtent(nx, ny, nz);
    const size_t nv = (size_t)nx * ny * nz;
    m_mask.assign(mask, mask + nv);
    size_t n_in = 0;
    for (size_t i = 0; i < nv; i++)
        n_in += (m_mask[i] != 0);
    Matrix coords(3, (int)n_in);
    size_t v = 0, idx = 0;
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++, idx++)
                if (m_mask[idx] != 0)
                {
                    coords.at0(0, (int)v) = x;
                    coords.at0(1, (int)v) = y;
                    coords.at0(2, (int)v) = z;
                    ++v;
                }
    SetVoxelCoords(coords);
}

void FabberRunDataArray::GetVoxelDataArray(std::string key, float *out)
{
    const Matrix &m = FabberRunData::GetVoxelData(key);
    const size_t nv = m_mask.size();
    const int rows = m.Nrows();
    for (int r = 0; r < rows; r++)
    {
        float *dst = out + (size_t)r * nv;
        int v = 0;
        for (size_t i = 0; i < nv; i++)
            dst[i] = (m_mask[i] != 0) ? (float)m.at0(r, v++) : 0.0f;
    }
}

void FabberRunDataArray::SetVoxelDataArray(std::string key, int data_size, const float *in)
{
    const size_t nv = m_mask.size();
    size_t n_in = 0;
    for (size_t i = 0; i < nv; i++)
        n_in += (m_mask[i] != 0);
    Matrix m(data_size, (int)n_in);
    for (int r = 0; r < data_size; r++)
    {
        const float *src = in + (size_t)r * nv;
        int v = 0;
        for (size_t i = 0; i < nv; i++)
            if (m_mask[i] != 0)
                m.at0(r, v++) = src[i];
    }
    FabberRunData::SetVoxelData(key, m);
}
