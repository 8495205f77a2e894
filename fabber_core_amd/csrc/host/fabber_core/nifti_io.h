/* nifti_io.h - minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer.
 *
 * Plays the role FSL's NEWIMAGE / NewNifti play for the reference's FabberRunDataNewimage
 * (rundata_newimage.cc:67-225): read a 3D/4D volume as float, write a float32 volume with the
 * geometry of a reference header. Own implementation of the published NIfTI-1 layout (348-byte
 * header + 4 bytes of extension flag, `n+1` magic), gzip through zlib. Host-side only. */
#pragma once

#include <string>
#include <vector>

namespace fabber_nifti
{
const int INTENT_NONE = 0;
const int INTENT_SYMMATRIX = 1005; /* NIFTI_INTENT_SYMMATRIX, used for finalMVN */

/* The 348-byte NIfTI-1 header, field for field */
#pragma pack(push, 1)
struct Header
{
    int sizeof_hdr;
    char data_type[10];
    char db_name[18];
    int extents;
    short session_error;
    char regular;
    char dim_info;
    short dim[8];
    float intent_p1, intent_p2, intent_p3;
    short intent_code;
    short datatype;
    short bitpix;
    short slice_start;
    float pixdim[8];
    float vox_offset;
    float scl_slope, scl_inter;
    short slice_end;
    char slice_code;
    char xyzt_units;
    float cal_max, cal_min;
    float slice_duration;
    float toffset;
    int glmax, glmin;
    char descrip[80];
    char aux_file[24];
    short qform_code, sform_code;
    float quatern_b, quatern_c, quatern_d;
    float qoffset_x, qoffset_y, qoffset_z;
    float srow_x[4], srow_y[4], srow_z[4];
    char intent_name[16];
    char magic[4];
};
#pragma pack(pop)
static_assert(sizeof(Header) == 348, "NIfTI-1 header is 348 bytes");

struct Volume
{
    Header hdr;              /* as read (native byte order), or to be written */
    int nx, ny, nz, nt;      /* nt = product of dim[4..7] */
    std::vector<float> data; /* [t][z][y][x], x fastest; scl_slope / scl_inter applied */
    Volume()
        : nx(0)
        , ny(0)
        , nz(0)
        , nt(0)
    {
    }
};

/* `name` with or without extension: tries name, name.nii.gz, name.nii (fsl_imageexists).
 * Returns the path that exists or "". */
std::string find_image(const std::string &name);

/* Throws std::runtime_error with a message naming the file. */
void read_volume(const std::string &path, Volume &vol);

/* Default header for an nx x ny x nz x nt float32 volume with voxel sizes (1, 1, 1). */
Header default_header(int nx, int ny, int nz, int nt);

/* Writes float32. `path` decides the compression (.gz); dims are taken from nx..nt of `vol`, every
 * other field (geometry, units, intent) from vol.hdr. */
void write_volume(const std::string &path, const Volume &vol);

/* FSL's convention for names without extension: $FSLOUTPUTTYPE NIFTI -> .nii, anything else
 * (default NIFTI_GZ) -> .nii.gz. A name that already ends in .nii / .nii.gz is kept. */
std::string output_path(const std::string &name);
}

/* C entry points (for FFI users and the unit tests): 0 on success, -1 with a message in err_buf
 * (>= 256 bytes, may be NULL) on failure. Volumes are float [t][z][y][x], x fastest. */
extern "C" {
/* dims[4] receives nx, ny, nz, nt. With buf == NULL only the dimensions are read. */
int fabber_nifti_read(const char *path, int *dims, float *buf, unsigned long long buf_elems, char *err_buf);
/* intent_code: 0 or 1005 (SYMMATRIX); voxel sizes pixdim[3] may be NULL (1 mm). */
int fabber_nifti_write(const char *path, const int *dims, const float *data, int intent_code, const float *pixdim, char *err_buf);
}
