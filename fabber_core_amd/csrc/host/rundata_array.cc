// rundata_array.cc - flat float arrays <-> per-voxel matrices for the C ABI.
// Volume order is x fastest, then y, z, and the 4th (time / row) dimension slowest
// (fabber_capi.h:100-105); voxel index = rank of the voxel among the non-zero mask entries in
// that scan (reference: rundata_array.cc:23-133).
#include "rundata_array.h"

#include "tools.h"

#include <string.h>

using NEWMAT::Matrix;
using std::endl;

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

// Both directions are passes over whole volumes (rows x nx ny nz elements): rows are independent, so they are
// shared out over host threads, and a mask that keeps every voxel - the usual case when the caller has masked
// already - is a plain conversion loop without the gather / scatter.
void FabberRunDataArray::GetVoxelDataArray(std::string key, float *out)
{
    const Matrix &m = FabberRunData::GetVoxelData(key);
    const size_t nv = m_mask.size();
    const int rows = m.Nrows();
    const size_t cols = (size_t)m.Ncols();
    const bool dense = cols == nv;
    const double *src = m.Store();
    const int *mask = m_mask.data();
    // (large rows are cut into pieces too: one-row images - every mean_ / std_ image - would be one thread's otherwise)
    const int pieces = (nv >= (1u << 18)) ? 8 : 1;
    fabber_parallel_for(rows * pieces, [=](int job) {
        const int r = job / pieces, piece = job % pieces;
        const size_t i0 = nv * piece / pieces, i1 = nv * (piece + 1) / pieces;
        float *dst = out + (size_t)r * nv;
        const double *row = src + (size_t)r * cols;
        if (dense)
        {
            for (size_t i = i0; i < i1; i++)
                dst[i] = (float)row[i];
            return;
        }
        size_t v = 0;
        for (size_t i = 0; i < i0; i++)
            v += (mask[i] != 0);
        for (size_t i = i0; i < i1; i++)
            dst[i] = (mask[i] != 0) ? (float)row[v++] : 0.0f;
    });
}

void FabberRunDataArray::SetVoxelDataArray(std::string key, int data_size, const float *in)
{
    const size_t nv = m_mask.size();
    size_t n_in = 0;
    for (size_t i = 0; i < nv; i++)
        n_in += (m_mask[i] != 0);
    // kept as float32 [rows][masked voxels] (FabberRunData::SetVoxelDataF32): what the engine reads, and a Matrix
    // only when something asks for one
    FabberF32Values values;
    values.resize((size_t)std::max(data_size, 0) * n_in); // (not written here: the threads below touch their own rows first)
    float *dst0 = values.data();
    const int *mask = m_mask.data();
    const bool dense = n_in == nv;
    fabber_parallel_for(data_size, [=](int r) {
        const float *src = in + (size_t)r * nv;
        float *dst = dst0 + (size_t)r * n_in;
        if (dense)
        {
            memcpy(dst, src, sizeof(float) * nv);
            return;
        }
        size_t v = 0;
        for (size_t i = 0; i < nv; i++)
            if (mask[i] != 0)
                dst[v++] = src[i];
    });
    FabberRunData::SetVoxelDataF32(key, data_size, std::move(values));
}

void FabberRunDataArray::SaveVoxelDataMove(const std::string &filename, Matrix &data, VoxelDataType)
{
    LOG << "FabberRunData::Saving to memory: " << filename << endl;
    CheckSize(filename, data);
    m_voxel_data_f32.erase(filename);
    m_voxel_data[filename].SwapContents(data);
}
