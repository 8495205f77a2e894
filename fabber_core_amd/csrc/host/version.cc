#include "version.h"
std::string fabber_version()
{
    return "fabber_core_amd 0.1 (MI355X voxelwise VB engine; API level of fabber_core v4)";
}
std::string fabber_source_date()
{
    return __DATE__;
}
