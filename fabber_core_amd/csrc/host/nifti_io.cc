/* nifti_io.cc - see fabber_core/nifti_io.h */
#include "fabber_core/nifti_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace fabber_nifti
{
namespace
{
bool ends_with(const std::string &s, const std::string &suffix)
{
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

bool file_exists(const std::string &path)
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f)
        return false;
    gzclose(f);
    return true;
}

template <typename T>
T swapped(T v)
{
    unsigned char *p = reinterpret_cast<unsigned char *>(&v);
    for (size_t i = 0; i < sizeof(T) / 2; i++)
        std::swap(p[i], p[sizeof(T) - 1 - i]);
    return v;
}

void swap_header(Header &h)
{
#define SW(f) h.f = swapped(h.f)
    SW(sizeof_hdr);
    SW(extents);
    SW(session_error);
    for (int i = 0; i < 8; i++)
    {
        SW(dim[i]);
        SW(pixdim[i]);
    }
    SW(intent_p1);
    SW(intent_p2);
    SW(intent_p3);
    SW(intent_code);
    SW(datatype);
    SW(bitpix);
    SW(slice_start);
    SW(vox_offset);
    SW(scl_slope);
    SW(scl_inter);
    SW(slice_end);
    SW(cal_max);
    SW(cal_min);
    SW(slice_duration);
    SW(toffset);
    SW(glmax);
    SW(glmin);
    SW(qform_code);
    SW(sform_code);
    SW(quatern_b);
    SW(quatern_c);
    SW(quatern_d);
    SW(qoffset_x);
    SW(qoffset_y);
    SW(qoffset_z);
    for (int i = 0; i < 4; i++)
    {
        SW(srow_x[i]);
        SW(srow_y[i]);
        SW(srow_z[i]);
    }
#undef SW
}

void read_exact(gzFile f, void *dst, size_t n, const std::string &path)
{
    char *p = static_cast<char *>(dst);
    while (n > 0)
    {
        const unsigned chunk = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        const int got = gzread(f, p, chunk);
        if (got <= 0)
            throw std::runtime_error("NIfTI: unexpected end of file in " + path);
        p += got;
        n -= (size_t)got;
    }
}

template <typename T>
void convert(const std::vector<unsigned char> &raw, bool swap, double slope, double inter, std::vector<float> &out)
{
    const size_t n = out.size();
    const T *src = reinterpret_cast<const T *>(raw.data());
    for (size_t i = 0; i < n; i++)
    {
        T v = src[i];
        if (swap)
            v = swapped(v);
        out[i] = (float)((double)v * slope + inter);
    }
}
} // namespace

std::string find_image(const std::string &name)
{
    if (name.empty())
        return "";
    const char *tries[] = { "", ".nii.gz", ".nii" };
    for (const char *ext : tries)
    {
        const std::string p = name + ext;
        if ((ends_with(p, ".nii") || ends_with(p, ".nii.gz")) && file_exists(p))
            return p;
    }
    return "";
}

void read_volume(const std::string &path, Volume &vol)
{
    gzFile f = gzopen(path.c_str(), "rb"); // reads plain files transparently
    if (!f)
        throw std::runtime_error("NIfTI: cannot open " + path);
    try
    {
        Header &h = vol.hdr;
        read_exact(f, &h, sizeof(Header), path);
        bool swap = false;
        if (h.sizeof_hdr != 348)
        {
            swap_header(h);
            swap = true;
            if (h.sizeof_hdr != 348)
                throw std::runtime_error("NIfTI: " + path + " is not a NIfTI-1 file (sizeof_hdr != 348; NIfTI-2 and ANALYZE pairs are not supported)");
        }
        if (std::strncmp(h.magic, "n+1", 3) != 0)
            throw std::runtime_error("NIfTI: " + path + " is not a single-file NIfTI-1 image (magic '" + std::string(h.magic, 3) + "')");
        const int nd = h.dim[0];
        if (nd < 1 || nd > 7)
            throw std::runtime_error("NIfTI: bad dim[0] in " + path);
        vol.nx = h.dim[1];
        vol.ny = nd >= 2 ? h.dim[2] : 1;
        vol.nz = nd >= 3 ? h.dim[3] : 1;
        vol.nt = 1;
        for (int i = 4; i <= nd; i++)
            vol.nt *= (h.dim[i] > 0 ? h.dim[i] : 1);
        if (vol.nx <= 0 || vol.ny <= 0 || vol.nz <= 0 || vol.nt <= 0)
            throw std::runtime_error("NIfTI: non-positive dimension in " + path);
        const size_t n = (size_t)vol.nx * vol.ny * vol.nz * vol.nt;
        size_t esz = 0;
        switch (h.datatype)
        {
        case 2: case 256: esz = 1; break;
        case 4: case 512: esz = 2; break;
        case 8: case 16: case 768: esz = 4; break;
        case 64: esz = 8; break;
        default:
            throw std::runtime_error("NIfTI: unsupported datatype " + std::to_string(h.datatype) + " in " + path);
        }
        // skip to vox_offset (header + extensions)
        size_t skip = (size_t)(h.vox_offset > 348 ? h.vox_offset : 352) - sizeof(Header);
        std::vector<unsigned char> junk(skip);
        if (skip)
            read_exact(f, junk.data(), skip, path);
        std::vector<unsigned char> raw(n * esz);
        read_exact(f, raw.data(), raw.size(), path);
        double slope = h.scl_slope, inter = h.scl_inter;
        if (slope == 0 || !std::isfinite(slope))
        {
            slope = 1;
            inter = 0;
        }
        vol.data.assign(n, 0.0f);
        switch (h.datatype)
        {
        case 2: convert<uint8_t>(raw, swap, slope, inter, vol.data); break;
        case 256: convert<int8_t>(raw, swap, slope, inter, vol.data); break;
        case 4: convert<int16_t>(raw, swap, slope, inter, vol.data); break;
        case 512: convert<uint16_t>(raw, swap, slope, inter, vol.data); break;
        case 8: convert<int32_t>(raw, swap, slope, inter, vol.data); break;
        case 768: convert<uint32_t>(raw, swap, slope, inter, vol.data); break;
        case 16: convert<float>(raw, swap, slope, inter, vol.data); break;
        case 64: convert<double>(raw, swap, slope, inter, vol.data); break;
        }
    }
    catch (...)
    {
        gzclose(f);
        throw;
    }
    gzclose(f);
}

Header default_header(int nx, int ny, int nz, int nt)
{
    Header h;
    std::memset(&h, 0, sizeof(h));
    h.sizeof_hdr = 348;
    h.regular = 'r';
    h.dim[0] = nt > 1 ? 4 : 3;
    h.dim[1] = (short)nx;
    h.dim[2] = (short)ny;
    h.dim[3] = (short)nz;
    h.dim[4] = (short)nt;
    h.dim[5] = h.dim[6] = h.dim[7] = 1;
    h.datatype = 16;
    h.bitpix = 32;
    h.pixdim[0] = 1;
    for (int i = 1; i < 8; i++)
        h.pixdim[i] = 1;
    h.vox_offset = 352;
    h.scl_slope = 1;
    h.xyzt_units = 2 | 8; // mm, s
    h.sform_code = 1;     // scanner-anat, identity voxel-to-world
    h.srow_x[0] = 1;
    h.srow_y[1] = 1;
    h.srow_z[2] = 1;
    h.qform_code = 1;
    std::memcpy(h.magic, "n+1", 4);
    return h;
}

void write_volume(const std::string &path, const Volume &vol)
{
    const size_t n = (size_t)vol.nx * vol.ny * vol.nz * vol.nt;
    if (vol.data.size() != n)
        throw std::runtime_error("NIfTI: data size does not match the dimensions for " + path);
    if (vol.nx > 32767 || vol.ny > 32767 || vol.nz > 32767 || vol.nt > 32767)
        throw std::runtime_error("NIfTI-1 cannot hold a dimension above 32767 (" + path + ")");
    Header h = vol.hdr;
    h.sizeof_hdr = 348;
    h.dim[0] = 4;
    h.dim[1] = (short)vol.nx;
    h.dim[2] = (short)vol.ny;
    h.dim[3] = (short)vol.nz;
    h.dim[4] = (short)vol.nt;
    h.dim[5] = h.dim[6] = h.dim[7] = 1;
    h.datatype = 16;
    h.bitpix = 32;
    h.vox_offset = 352;
    h.scl_slope = 1;
    h.scl_inter = 0;
    std::memcpy(h.magic, "n+1", 4);
    const char ext[4] = { 0, 0, 0, 0 };
    if (ends_with(path, ".gz"))
    {
        gzFile f = gzopen(path.c_str(), "wb1"); // fast level: these are large float volumes
        if (!f)
            throw std::runtime_error("NIfTI: cannot create " + path);
        bool ok = gzwrite(f, &h, sizeof(h)) == (int)sizeof(h) && gzwrite(f, ext, 4) == 4;
        const char *p = reinterpret_cast<const char *>(vol.data.data());
        size_t left = n * sizeof(float);
        while (ok && left > 0)
        {
            const unsigned chunk = left > (1u << 30) ? (1u << 30) : (unsigned)left;
            ok = gzwrite(f, p, chunk) == (int)chunk;
            p += chunk;
            left -= chunk;
        }
        ok = (gzclose(f) == Z_OK) && ok;
        if (!ok)
            throw std::runtime_error("NIfTI: write error on " + path);
    }
    else
    {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f)
            throw std::runtime_error("NIfTI: cannot create " + path);
        bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(ext, 4, 1, f) == 1
            && std::fwrite(vol.data.data(), sizeof(float), n, f) == n;
        ok = (std::fclose(f) == 0) && ok;
        if (!ok)
            throw std::runtime_error("NIfTI: write error on " + path);
    }
}

std::string output_path(const std::string &name)
{
    if (ends_with(name, ".nii") || ends_with(name, ".nii.gz"))
        return name;
    const char *t = std::getenv("FSLOUTPUTTYPE");
    if (t && std::string(t) == "NIFTI")
        return name + ".nii";
    return name + ".nii.gz";
}
} // namespace fabber_nifti

namespace
{
int nifti_fail(char *err_buf, const std::string &msg)
{
    if (err_buf)
    {
        std::strncpy(err_buf, msg.c_str(), 254);
        err_buf[254] = 0;
    }
    return -1;
}
} // namespace

extern "C" int fabber_nifti_read(const char *path, int *dims, float *buf, unsigned long long buf_elems, char *err_buf)
{
    try
    {
        const std::string found = fabber_nifti::find_image(path ? path : "");
        if (found.empty())
            return nifti_fail(err_buf, std::string("NIfTI: no such image: ") + (path ? path : "(null)"));
        fabber_nifti::Volume vol;
        fabber_nifti::read_volume(found, vol);
        if (dims)
        {
            dims[0] = vol.nx;
            dims[1] = vol.ny;
            dims[2] = vol.nz;
            dims[3] = vol.nt;
        }
        if (buf)
        {
            if (buf_elems < vol.data.size())
                return nifti_fail(err_buf, "NIfTI: buffer too small");
            std::memcpy(buf, vol.data.data(), vol.data.size() * sizeof(float));
        }
        return 0;
    }
    catch (std::exception &e)
    {
        return nifti_fail(err_buf, e.what());
    }
}

extern "C" int fabber_nifti_write(const char *path, const int *dims, const float *data, int intent_code, const float *pixdim, char *err_buf)
{
    try
    {
        if (!path || !dims || !data)
            return nifti_fail(err_buf, "NIfTI: null argument");
        fabber_nifti::Volume vol;
        vol.nx = dims[0];
        vol.ny = dims[1];
        vol.nz = dims[2];
        vol.nt = dims[3];
        vol.hdr = fabber_nifti::default_header(vol.nx, vol.ny, vol.nz, vol.nt);
        vol.hdr.intent_code = (short)intent_code;
        if (pixdim)
            for (int i = 0; i < 3; i++)
            {
                vol.hdr.pixdim[1 + i] = pixdim[i];
                (i == 0 ? vol.hdr.srow_x : (i == 1 ? vol.hdr.srow_y : vol.hdr.srow_z))[i] = pixdim[i];
            }
        vol.data.assign(data, data + (size_t)vol.nx * vol.ny * vol.nz * vol.nt);
        fabber_nifti::write_volume(path, vol);
        return 0;
    }
    catch (std::exception &e)
    {
        return nifti_fail(err_buf, e.what());
    }
}
