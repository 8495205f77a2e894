/* rundata_newimage.h - FabberRunData that loads and saves NIfTI files (the reference's class of
 * the same name, rundata_newimage.h:29-43, there on top of FSL NEWIMAGE; here on the small reader /
 * writer of nifti_io.h). */
#pragma once

#include "nifti_io.h"
#include "rundata.h"

#include "armawrap/newmat.h"

#include <string>
#include <vector>

class FabberRunDataNewimage : public FabberRunData
{
public:
    FabberRunDataNewimage(bool compat_options = true);

    /** Extent, co-ordinates and mask from --mask, or from the main data if there is no mask
     *  (rundata_newimage.cc:67-101) */
    void SetExtentFromData();
    const NEWMAT::Matrix &LoadVoxelData(const std::string &filename);
    virtual void SaveVoxelData(const std::string &filename, NEWMAT::Matrix &data, VoxelDataType data_type = VDT_SCALAR);

private:
    void SetCoordsFromExtent(int nx, int ny, int nz);
    void SetMask(const fabber_nifti::Volume &vol, bool all_ones);
    fabber_nifti::Header m_ref_header; /* geometry every output inherits */
    std::vector<unsigned char> m_mask; /* [z][y][x], 1 = inside */
    bool m_have_mask;
};
