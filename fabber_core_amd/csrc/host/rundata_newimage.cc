/* rundata_newimage.cc - NIfTI-backed run data, following rundata_newimage.cc:60-225 of the
 * reference (mask binarised at > 1e-16 :80, masked [t][voxel] matrices in x-fastest voxel order
 * :145, SYMMATRIX intent for MVN outputs :163-171, co-ordinates = grid indices :197-225). */
#include "fabber_core/rundata_newimage.h"

#include "fabber_core/easylog.h"

#include <ostream>
#include <stdexcept>

using namespace std;
using NEWMAT::Matrix;
using fabber_nifti::Volume;

static void DumpVolumeInfo(const Volume &v, ostream &out)
{
    out << "FabberRunDataNewimage::Dimensions: x=" << v.nx << ", y=" << v.ny << ", z=" << v.nz << ", vols=" << v.nt << endl;
    out << "FabberRunDataNewimage::Voxel size: x=" << v.hdr.pixdim[1] << "mm, y=" << v.hdr.pixdim[2] << "mm, z=" << v.hdr.pixdim[3]
        << "mm, TR=" << v.hdr.pixdim[4] << " sec\n";
    out << "FabberRunDataNewimage::Intents: " << v.hdr.intent_code << ", " << v.hdr.intent_p1 << ", " << v.hdr.intent_p2 << ", "
        << v.hdr.intent_p3 << endl;
}

FabberRunDataNewimage::FabberRunDataNewimage(bool compat_options)
    : FabberRunData(compat_options)
    , m_ref_header(fabber_nifti::default_header(1, 1, 1, 1))
    , m_have_mask(false)
{
}

void FabberRunDataNewimage::SetMask(const Volume &vol, bool all_ones)
{
    m_ref_header = vol.hdr;
    const size_t n = (size_t)vol.nx * vol.ny * vol.nz;
    m_mask.assign(n, 1);
    if (!all_ones)
        for (size_t i = 0; i < n; i++)
            m_mask[i] = vol.data[i] > 1e-16f ? 1 : 0; // binarise(1e-16, max + 1, exclusive)
    m_have_mask = true;
}

void FabberRunDataNewimage::SetExtentFromData()
{
    string mask_fname = GetStringDefault("mask", "");
    if (mask_fname != "")
    {
        LOG << "FabberRunDataNewimage::Loading mask data from '" + mask_fname << "'" << endl;
        const string path = fabber_nifti::find_image(mask_fname);
        if (path == "")
            throw DataNotFound(mask_fname, "File is invalid or does not exist");
        Volume vol;
        fabber_nifti::read_volume(path, vol);
        vol.nt = 1; // a 4D mask contributes its first volume, as read_volume(volume<float>&) does
        SetMask(vol, false);
        DumpVolumeInfo(vol, LOG);
        SetCoordsFromExtent(vol.nx, vol.ny, vol.nz);
    }
    else
    {
        LOG << "FabberRunDataNewimage::No mask, using data for extent" << endl;
        string data_fname = GetStringDefault("data", GetStringDefault("data1", ""));
        const string path = fabber_nifti::find_image(data_fname);
        if (path == "")
            throw DataNotFound(data_fname, "File is invalid or does not exist");
        Volume vol;
        fabber_nifti::read_volume(path, vol);
        m_have_mask = false;
        m_ref_header = vol.hdr;
        SetCoordsFromExtent(vol.nx, vol.ny, vol.nz);
    }
}

const Matrix &FabberRunDataNewimage::LoadVoxelData(const std::string &filename)
{
    if (m_voxel_data.find(filename) == m_voxel_data.end())
    {
        const string path = fabber_nifti::find_image(filename);
        if (path == "")
            throw DataNotFound(filename, "File is invalid or does not exist");
        LOG << "FabberRunDataNewimage::Loading data from '" + filename << "'" << endl;
        Volume vol;
        try
        {
            fabber_nifti::read_volume(path, vol);
        }
        catch (std::exception &e)
        {
            throw DataNotFound(filename, string("Error loading file: ") + e.what());
        }
        if (!m_have_mask)
            SetMask(vol, true); // outputs take their geometry from the first data set loaded
        DumpVolumeInfo(vol, LOG);
        const size_t nvox = (size_t)vol.nx * vol.ny * vol.nz;
        if (nvox != m_mask.size())
        {
            LOG << "NEWMAT error while applying mask... Most likely a dimension mismatch. ***\n";
            throw FabberRunDataError("Data set '" + filename + "' does not have the dimensions of the mask / main data");
        }
        LOG << "FabberRunDataNewimage::Applying mask to data..." << endl;
        size_t inside = 0;
        for (size_t i = 0; i < nvox; i++)
            inside += m_mask[i];
        Matrix m(vol.nt, (int)inside);
        double sum = 0;
        for (int t = 0; t < vol.nt; t++)
        {
            const float *src = vol.data.data() + (size_t)t * nvox;
            int col = 0;
            for (size_t i = 0; i < nvox; i++)
                if (m_mask[i])
                {
                    m.at0(t, col++) = src[i];
                    sum += src[i];
                }
        }
        m_voxel_data[filename] = m;
        const double count = (double)m.Nrows() * m.Ncols();
        LOG << "FabberRunDataNewimage::GetVoxelData: " << filename << " mean value=" << (count > 0 ? sum / count : 0.0) << endl;
    }
    return m_voxel_data[filename];
}

void FabberRunDataNewimage::SaveVoxelData(const std::string &filename, NEWMAT::Matrix &data, VoxelDataType data_type)
{
    LOG << "FabberRunDataNewimage::Saving to nifti: " << filename << endl;
    Volume out;
    out.hdr = m_ref_header;
    out.nx = m_extent[0];
    out.ny = m_extent[1];
    out.nz = m_extent[2];
    out.nt = data.Nrows();
    const size_t nvox = (size_t)out.nx * out.ny * out.nz;
    out.data.assign(nvox * out.nt, 0.0f);
    float vmin = 0, vmax = 0;
    bool first = true;
    for (int t = 0; t < out.nt; t++)
    {
        float *dst = out.data.data() + (size_t)t * nvox;
        int col = 0;
        for (size_t i = 0; i < nvox; i++)
            if (!m_have_mask || m_mask[i])
            {
                if (col >= data.Ncols())
                    throw FabberRunDataError("SaveVoxelData: '" + filename + "' has fewer voxels than the mask");
                const float v = (float)data.at0(t, col++);
                dst[i] = v;
                if (first || v < vmin)
                    vmin = v;
                if (first || v > vmax)
                    vmax = v;
                first = false;
            }
    }
    out.hdr.intent_code = (short)(data_type == VDT_MVN ? fabber_nifti::INTENT_SYMMATRIX : fabber_nifti::INTENT_NONE);
    out.hdr.intent_p1 = out.hdr.intent_p2 = out.hdr.intent_p3 = 0;
    out.hdr.cal_max = vmax; // setDisplayMaximumMinimum
    out.hdr.cal_min = vmin;
    const string target = (filename[0] == '/') ? filename : GetOutputDir() + "/" + filename;
    fabber_nifti::write_volume(fabber_nifti::output_path(target), out);
}

void FabberRunDataNewimage::SetCoordsFromExtent(int nx, int ny, int nz)
{
    LOG << "FabberRunDataNewimage::Setting coordinates from extent" << endl;
    FabberRunData::SetExtent(nx, ny, nz);
    const size_t nvox = (size_t)nx * ny * nz;
    size_t inside = nvox;
    if (m_have_mask)
    {
        inside = 0;
        for (size_t i = 0; i < nvox; i++)
            inside += m_mask[i];
    }
    Matrix coords(3, (int)inside);
    int col = 0;
    size_t i = 0;
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int ii = 0; ii < nx; ii++, i++)
                if (!m_have_mask || m_mask[i])
                {
                    coords.at0(0, col) = ii;
                    coords.at0(1, col) = j;
                    coords.at0(2, col) = k;
                    col++;
                }
    SetVoxelCoords(coords);
}
