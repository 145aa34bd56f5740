#!/bin/bash
# usage: tools/variants.sh <workloads...>   (runs kernel_times for the default build and every variants/*.so)
echo "== default"; timeout -k 10 100 python -u tools/kernel_times.py "$@" | grep "shade median" || exit 1
for f in variants/*.so; do echo "== $f"; RTX_AMD_LIB=$PWD/$f timeout -k 10 100 python -u tools/kernel_times.py "$@" | grep "shade median" || exit 1; done
