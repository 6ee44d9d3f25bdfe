# libs/hip.cmake of the lsbench tree -- ENABLE_HIP=ON.  Counterpart of
# libs/cusparse.cmake: where that one finds the CUDA toolkit, this one finds the
# backend library (built by `make -C lsbench_amd/csrc` of the hip_cdna4
# repository: gcc for the C host side, hipcc --offload-arch=gfx950 for the
# kernels) and links it; the library brings libamdhip64 and librccl with it.
#
#   cmake -DENABLE_HIP=ON -DENABLE_CHOLMOD=OFF \
#         -DLSBENCH_HIP_ROOT=/path/to/hip_cdna4-repo ..
set(LSBENCH_HIP_ROOT "" CACHE PATH "checkout of the hip_cdna4 backend (holds lsbench_amd/csrc and include/)")
find_library(LSBENCH_HIP_LIBRARY NAMES lsbench_hip
  HINTS ${LSBENCH_HIP_ROOT}/lsbench_amd/csrc ${LSBENCH_HIP_ROOT}/lib ENV LSBENCH_HIP_ROOT
  REQUIRED)
get_filename_component(LSBENCH_HIP_LIBDIR ${LSBENCH_HIP_LIBRARY} DIRECTORY)
target_link_libraries(lsbench PRIVATE ${LSBENCH_HIP_LIBRARY})
# the backend is found at run time next to where it was built
set_property(TARGET lsbench APPEND PROPERTY INSTALL_RPATH ${LSBENCH_HIP_LIBDIR})
set_property(TARGET lsbench APPEND PROPERTY BUILD_RPATH ${LSBENCH_HIP_LIBDIR})
