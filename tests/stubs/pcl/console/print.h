// TEST STUB (not PCL), see registration/registration.h
#pragma once
#include <cstdarg>
#include <cstdio>
namespace pcl { namespace console { inline void print_highlight(const char* fmt, ...) { va_list a; va_start(a, fmt); std::vfprintf(stdout, fmt, a); va_end(a); } } }
