// Spatial-VB kernel instantiations for several noise precisions and AR(1) noise (vb_spatial_noise.h), host model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_host(int P, bool need_f, int kind)
{
    switch (P)
    {
        FVB_SPATIAL_NZ_CASE(HostLinModel, "host", 1)
        FVB_SPATIAL_NZ_CASE(HostLinModel, "host", 2)
        FVB_SPATIAL_NZ_CASE(HostLinModel, "host", 3)
        FVB_SPATIAL_NZ_CASE(HostLinModel, "host", 4)
        FVB_SPATIAL_NZ_CASE(HostLinModel, "host", 5)
        FVB_SPATIAL_NZ_CASE(HostLinModel, "host", 6)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
