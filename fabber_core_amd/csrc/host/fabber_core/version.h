/* version.h */
#pragma once
#include <string>
std::string fabber_version();
std::string fabber_source_date();
