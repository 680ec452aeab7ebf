#!/bin/bash
# Registers, spills and scratch of every kernel at HEAD: each translation unit of the library compiled (device only, no GPU needed) with the product's
# flags + -Rpass-analysis=kernel-resource-usage, summarised by profiles/kernel_resources.py.   usage: bash profiles/kernel_resources_all.sh > profiles/rNN_kernel_resources.txt
T=$(mktemp -d)
for f in bbtools_amd/csrc/*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Rpass-analysis=kernel-resource-usage -x hip --cuda-device-only -c $f -o $T/$(basename $f).o 2> $T/$(basename $f).txt ) &
done
wait
echo "# git $(git rev-parse --short HEAD)$(git diff --quiet || echo ' + uncommitted changes'); hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, one line per kernel"
for f in bbtools_amd/csrc/*.hip; do
  echo "## $(basename $f)"
  python profiles/kernel_resources.py $T/$(basename $f).txt | sort
done
rm -rf $T
