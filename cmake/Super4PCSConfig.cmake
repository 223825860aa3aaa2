# Package configuration of the MI355X-native Super4PCS drop-in.
#
# The reference installs lib/cmake/Super4PCSConfig.cmake (generated from cmake/Config.cmake.in:34-39) which defines
#   Super4PCS_INCLUDE_DIR   include directory (parent of super4pcs/...)
#   Super4PCS_LIB_DIR       directory of the libraries
#   Super4PCS_LIBRARIES     super4pcs_accel super4pcs_io super4pcs_algo
# and an application does (tests/externalAppTest/CMakeLists.txt:6-11)
#   find_package(Super4PCS REQUIRED)
#   include_directories(${Super4PCS_INCLUDE_DIR})
#   link_directories(${Super4PCS_LIB_DIR})
#   target_link_libraries(app ${Super4PCS_LIBRARIES})
#
# Here the whole hot path lives in ONE shared library (libsuper4pcs_amd.so: HIP kernels + C ABI + host engine); the
# accelerators and the IO of the reference are header-only in this tree (include/super4pcs/**).  So that the reference's
# three library names keep working unchanged, they are defined as imported targets: super4pcs_algo is the shared library,
# super4pcs_accel and super4pcs_io are interface targets that forward to it.  The file is relocatable: every path is derived
# from its own location (<prefix>/cmake/ in the source tree, <prefix>/lib/cmake/ when installed by tools/install.sh).
#
# Usage:  -DCMAKE_PREFIX_PATH=<source tree>  /  -DCMAKE_PREFIX_PATH=<install prefix>/lib/cmake (as the reference's
# tests/CMakeLists.txt:46 does)  /  -DSuper4PCS_DIR=<directory of this file>

if(DEFINED Super4PCS_FOUND AND Super4PCS_FOUND AND TARGET super4pcs_algo)
  return()
endif()

get_filename_component(_s4p_cfg_dir "${CMAKE_CURRENT_LIST_FILE}" DIRECTORY)
get_filename_component(_s4p_prefix "${_s4p_cfg_dir}/.." ABSOLUTE)
if(NOT EXISTS "${_s4p_prefix}/include/super4pcs/shared4pcs.h")        # installed layout: <prefix>/lib/cmake
  get_filename_component(_s4p_prefix "${_s4p_cfg_dir}/../.." ABSOLUTE)
endif()

set(Super4PCS_VERSION 1.1.3)                                           # the reference release this tree is a drop-in for
set(Super4PCS_INCLUDE_DIR  "${_s4p_prefix}/include/")
set(Super4PCS_INCLUDE_DIRS "${Super4PCS_INCLUDE_DIR}")                 # (the reference's header comment names both spellings)

# the shared library: source tree (super4pcs_amd/lib) or installed (lib)
find_library(Super4PCS_AMD_LIBRARY NAMES super4pcs_amd
             PATHS "${_s4p_prefix}/super4pcs_amd/lib" "${_s4p_prefix}/lib" NO_DEFAULT_PATH)
if(NOT Super4PCS_AMD_LIBRARY)
  set(Super4PCS_FOUND FALSE)
  set(Super4PCS_NOT_FOUND_MESSAGE "libsuper4pcs_amd.so not found under ${_s4p_prefix}: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first")
  return()
endif()
get_filename_component(Super4PCS_LIB_DIR "${Super4PCS_AMD_LIBRARY}" DIRECTORY)
set(Super4PCS_LIB_DIR "${Super4PCS_LIB_DIR}/")

if(NOT TARGET super4pcs_algo)
  add_library(super4pcs_algo SHARED IMPORTED)
  set_target_properties(super4pcs_algo PROPERTIES
    IMPORTED_LOCATION "${Super4PCS_AMD_LIBRARY}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${Super4PCS_INCLUDE_DIR}")
endif()
foreach(_s4p_t super4pcs_accel super4pcs_io)
  if(NOT TARGET ${_s4p_t})
    add_library(${_s4p_t} INTERFACE IMPORTED)
    set_target_properties(${_s4p_t} PROPERTIES
      INTERFACE_INCLUDE_DIRECTORIES "${Super4PCS_INCLUDE_DIR}"
      INTERFACE_LINK_LIBRARIES super4pcs_algo)
  endif()
endforeach()
set(Super4PCS_LIBRARIES super4pcs_accel super4pcs_io super4pcs_algo)

# Eigen is optional here: with Eigen on the include path the public types are Eigen's (as in the reference); without it the
# headers fall back to their own fixed-size types (include/super4pcs/shared4pcs.h).  The reference leaves finding Eigen to
# the application as well (Config.cmake.in:15-16).
set(Super4PCS_FOUND TRUE)
unset(_s4p_cfg_dir)
unset(_s4p_t)
