// Spatial VB kernels for models that are evaluated on the host (HostLinModel, vb_models.h)
#include "vb_spatial.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_host(int P, bool need_f)
{
    switch (P)
    {
        FVB_SPATIAL_CASE(HostLinModel, "host", 1)
        FVB_SPATIAL_CASE(HostLinModel, "host", 2)
        FVB_SPATIAL_CASE(HostLinModel, "host", 3)
        FVB_SPATIAL_CASE(HostLinModel, "host", 4)
        FVB_SPATIAL_CASE(HostLinModel, "host", 5)
        FVB_SPATIAL_CASE(HostLinModel, "host", 6)
        FVB_SPATIAL_CASE(HostLinModel, "host", 7)
        FVB_SPATIAL_CASE(HostLinModel, "host", 8)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
