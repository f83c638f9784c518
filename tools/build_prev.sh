#!/bin/bash
# builds the library of the last commit into scratch/ab/liblatte_prev.so (for tools/ab_bench.sh); run in the build container
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/lt_prev && mkdir -p /tmp/lt_prev scratch/ab
git archive HEAD latte_amd/csrc include | tar -x -C /tmp/lt_prev
make -C /tmp/lt_prev/latte_amd/csrc -j8 ../liblatte_hip.so > /tmp/lt_prev/build.log 2>&1
cp /tmp/lt_prev/latte_amd/liblatte_hip.so scratch/ab/liblatte_prev.so
echo built scratch/ab/liblatte_prev.so
