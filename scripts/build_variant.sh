#!/bin/bash
# dev: a second build of libpt_amd.so beside the product's, for an A/B on one box (PT_LIB_AMD=build/variants/NAME/libpt_amd.so).
#   scripts/build_variant.sh NAME ["-DMACRO=... extra hipcc flags"] [git revision to take csrc/ and include/ from]
# build/ is git-ignored and travels to the GPU box with gpurun.
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; rev=$3
d=build/variants/$name
rm -rf "$d"; mkdir -p "$d/pkg"
if [ -n "$rev" ]; then
  git archive "$rev" single-file-vulkan-pathtracing_amd/csrc include | tar -x -C "$d"
  mv "$d/single-file-vulkan-pathtracing_amd/csrc" "$d/pkg/csrc"; rmdir "$d/single-file-vulkan-pathtracing_amd"
else
  mkdir -p "$d/pkg/csrc" "$d/include"
  cp single-file-vulkan-pathtracing_amd/csrc/*.h single-file-vulkan-pathtracing_amd/csrc/*.hip single-file-vulkan-pathtracing_amd/csrc/Makefile "$d/pkg/csrc/"
  cp include/*.h "$d/include/"
fi
# (the Makefile reaches the public header through ../../include: pkg/csrc -> $d/include)
make -C "$d/pkg/csrc" -j6 -s EXTRA="$extra"
mv "$d/pkg/libpt_amd.so" "$d/libpt_amd.so"
rm -f "$d"/pkg/csrc/*.o
echo "$d/libpt_amd.so"
