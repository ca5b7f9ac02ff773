#!/bin/bash
# ORACLE recipe: compile the pieces of the reference that build from their own few source files, where they lie
# under /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
#   iou3d_cpu.cpp  -> libiou3d_ref.so   (rotated BEV IoU on the CPU; pins oracle/iou3d_oracle.c)
#   roiaware_pool3d.cpp -> roiaware_pool3d_ref<EXT_SUFFIX>   (the reference's own pybind module, unmodified: its
#       points_in_boxes_cpu / check_pt_in_box3d_cpu pin the rotation + extent test of oracle/pointnet2_oracle.c. The file
#       also declares three CUDA launchers that nothing here defines or calls: the module is linked -z lazy and imported
#       with RTLD_LAZY (oracle.ref_points_in_boxes_cpu), so they stay unresolved and untouched. Nothing is stubbed.)
# cuda.h / cuda_runtime_api.h (included but unused by that file) come from the NVIDIA headers already shipped in this
# image with triton; nothing is stubbed. Needs: g++, torch headers. Skips quietly when /root/reference is absent.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/pcdet/ops/iou3d_nms/src
[ -d "$REF" ] || { echo "build_ref: reference not mounted, keeping prebuilt oracle/_ref"; exit 0; }
OUT="$HERE/_ref"; mkdir -p "$OUT"
PY=python3
TORCH_INC=$($PY -c "import torch.utils.cpp_extension as c; print(' '.join('-I'+p for p in c.include_paths()))")
TORCH_LIB=$($PY -c "import torch.utils.cpp_extension as c; print(c.library_paths()[0])")
PY_INC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NV_INC=$($PY -c "import triton, os; print(os.path.join(os.path.dirname(triton.__file__), 'backends/nvidia/include'))")
ABI=$($PY -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
EXT=$($PY -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
REF2=/root/reference/pcdet/ops/roiaware_pool3d/src
if [ ! "$OUT/roiaware_pool3d_ref$EXT" -nt "$REF2/roiaware_pool3d.cpp" ]; then
g++ -O2 -fPIC -shared -std=c++17 -D_GLIBCXX_USE_CXX11_ABI=$ABI -DTORCH_EXTENSION_NAME=roiaware_pool3d_ref \
    $TORCH_INC -I"$PY_INC" "$REF2/roiaware_pool3d.cpp" \
    -L"$TORCH_LIB" -ltorch -ltorch_cpu -ltorch_python -lc10 -Wl,-rpath,"$TORCH_LIB" -Wl,-z,lazy \
    -o "$OUT/roiaware_pool3d_ref$EXT" -w
echo "build_ref: built $OUT/roiaware_pool3d_ref$EXT"
fi
if [ "$OUT/libiou3d_ref.so" -nt "$REF/iou3d_cpu.cpp" ] && [ "$OUT/libiou3d_ref.so" -nt "$HERE/ref_bind_iou3d.cpp" ]; then exit 0; fi
g++ -O2 -fPIC -shared -std=c++17 -D_GLIBCXX_USE_CXX11_ABI=$ABI -DTORCH_EXTENSION_NAME=iou3d_ref \
    $TORCH_INC -I"$PY_INC" -I"$NV_INC" -I"$REF" \
    "$REF/iou3d_cpu.cpp" "$HERE/ref_bind_iou3d.cpp" \
    -L"$TORCH_LIB" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TORCH_LIB" -o "$OUT/libiou3d_ref.so" -w
echo "build_ref: built $OUT/libiou3d_ref.so"
