#!/bin/bash
# tools/build_variant.sh <tag> <extra compiler flags...>: a differently compiled build of the product sources under
# supersurfel_fusion_amd/csrc/variants/<tag>/libssf_hip.so (git-ignored, travels with gpurun), selected at run time with
# SSF_PRODUCT_VARIANT=<tag> (binding.load_product).  E.g.  tools/build_variant.sh prof -DSSF_PASSES_PROFILE
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:?tag}; shift
D=$R/supersurfel_fusion_amd/csrc/variants/$TAG
mkdir -p $D
cp $R/supersurfel_fusion_amd/csrc/*.hip $R/supersurfel_fusion_amd/csrc/*.hpp $R/supersurfel_fusion_amd/csrc/*.inc $R/supersurfel_fusion_amd/csrc/Makefile $D/
cp -r $R/supersurfel_fusion_amd/csrc/lab $D/
sed -i 's#\.\./\.\./include/#../../../../include/#' $D/*.hip $D/*.hpp $D/Makefile
make -C $D -j4 libssf_hip.so EXTRA="$*" 2>&1 | grep -E "error|warning" || true
rm -rf $D/*.hip $D/*.hpp $D/*.inc $D/Makefile $D/*.o $D/lab
ls -la $D/libssf_hip.so
