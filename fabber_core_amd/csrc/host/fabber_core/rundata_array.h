/* rundata_array.h - run data fed from / read back into flat float arrays: the storage behind
 * the C ABI (reference: rundata_array.h, rundata_array.cc:23-133). */
#pragma once

#include "rundata.h"

#include <string>
#include <vector>

class FabberRunDataArray : public FabberRunData
{
public:
    explicit FabberRunDataArray(bool compat_options = true)
        : FabberRunData(compat_options)
    {
    }
    /** mask: nx*ny*nz ints, x fastest; non-zero = voxel included */
    void SetExtent(int nx, int ny, int nz, const int *mask);
    /** scatter a named result into a nx*ny*nz*size float volume, zeros outside the mask */
    void GetVoxelDataArray(std::string key, float *data);
    /** gather a nx*ny*nz*data_size float volume through the mask */
    void SetVoxelDataArray(std::string key, int data_size, const float *data);
    /** results stay in memory here: the matrix is taken over, not copied */
    void SaveVoxelDataMove(const std::string &filename, NEWMAT::Matrix &data, VoxelDataType data_type = VDT_SCALAR);

private:
    std::vector<int> m_mask;
};
