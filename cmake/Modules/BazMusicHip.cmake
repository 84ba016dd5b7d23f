# BazMusicHip.cmake -- the HIP gate of the MUSIC-DoA block (gfx950 / MI355X).
#
# Replaces gr-baz's Armadillo gate (CMakeLists.txt:195-202 `find_package(Armadillo)` ... and
# lib/CMakeLists.txt:166-170,217): the block no longer needs Armadillo / LAPACK, it needs a HIP toolchain that can
# build the kernel library.  Include this file AFTER `project(... CXX)`; it
#   * enables the HIP language for CMAKE_HIP_ARCHITECTURES = gfx950 (the kernels are written for gfx950 only),
#   * defines BAZ_MUSIC_HIP_FOUND (also written to config.h, see config.h.in / swig/baz_music.i),
#   * defines the kernel libraries  baz_music_hip  baz_agc_hip  baz_resamp_hip  (C-ABI: include/baz_*_hip.h),
#   * sets   BAZ_MUSIC_HIP_HOST_SOURCES   host-block sources to append to `baz_sources`
#            BAZ_MUSIC_HIP_LIBRARIES      to append to the link line in place of ${ARMADILLO_LIBRARIES}
#            BAZ_MUSIC_HIP_INCLUDE_DIRS   include/ (the ABI) and the host-block headers.
# In a gr-baz tree:   set(BAZ_MUSIC_HIP_ROOT <path to this repository>)  before including it.
include(CheckLanguage)

if(NOT DEFINED BAZ_MUSIC_HIP_ROOT)
    get_filename_component(BAZ_MUSIC_HIP_ROOT "${CMAKE_CURRENT_LIST_DIR}/../.." ABSOLUTE)
endif()
if(NOT DEFINED CMAKE_HIP_ARCHITECTURES)
    set(CMAKE_HIP_ARCHITECTURES gfx950)
endif()
if(NOT CMAKE_HIP_ARCHITECTURES STREQUAL "gfx950")
    message(FATAL_ERROR "BazMusicHip: the MUSIC-DoA kernels are written for gfx950 (MI355X) only, not '${CMAKE_HIP_ARCHITECTURES}'")
endif()

check_language(HIP)
if(CMAKE_HIP_COMPILER)
    enable_language(HIP)
    set(BAZ_MUSIC_HIP_FOUND TRUE)
    message(STATUS "HIP found (${CMAKE_HIP_COMPILER}) - compiling MUSIC DOA estimator block for gfx950")
else()
    set(BAZ_MUSIC_HIP_FOUND FALSE)
    message(STATUS "HIP NOT found! NOT compiling MUSIC DOA estimator block")
    return()
endif()

set(_baz_csrc ${BAZ_MUSIC_HIP_ROOT}/gr_baz_amd/csrc)
set(_baz_host ${BAZ_MUSIC_HIP_ROOT}/gr_baz_amd/host)
set(BAZ_MUSIC_HIP_INCLUDE_DIRS ${BAZ_MUSIC_HIP_ROOT}/include ${_baz_host})

# -amdgpu-mfma-vgpr-form: MFMA accumulators in VGPRs (gfx950 has a unified register file)
set(_baz_hip_flags -O3 -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form)
foreach(_k music agc resamp)
    add_library(baz_${_k}_hip SHARED ${_baz_csrc}/baz_${_k}_hip.hip)
    set_source_files_properties(${_baz_csrc}/baz_${_k}_hip.hip PROPERTIES LANGUAGE HIP)
    target_include_directories(baz_${_k}_hip PUBLIC ${BAZ_MUSIC_HIP_ROOT}/include)
    target_compile_options(baz_${_k}_hip PRIVATE ${_baz_hip_flags})
    set_target_properties(baz_${_k}_hip PROPERTIES HIP_STANDARD 17 POSITION_INDEPENDENT_CODE ON)
endforeach()

set(BAZ_MUSIC_HIP_LIBRARIES baz_music_hip baz_agc_hip baz_resamp_hip)
set(BAZ_MUSIC_HIP_HOST_SOURCES
    ${_baz_host}/baz_music_doa.cc
    ${_baz_host}/baz_agc_cc.cc
    ${_baz_host}/baz_fractional_resampler_cc.cc)
set(BAZ_MUSIC_HIP_HOST_HEADERS
    ${_baz_host}/baz_music_doa.h
    ${_baz_host}/baz_agc_cc.h
    ${_baz_host}/baz_fractional_resampler_cc.h)
set(BAZ_MUSIC_HIP_ABI_HEADERS
    ${BAZ_MUSIC_HIP_ROOT}/include/baz_music_hip.h
    ${BAZ_MUSIC_HIP_ROOT}/include/baz_agc_hip.h
    ${BAZ_MUSIC_HIP_ROOT}/include/baz_resamp_hip.h)
