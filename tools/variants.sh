#!/bin/bash
# usage: tools/variants.sh <pattern to grep> <workloads...>   (runs kernel_times for the default build and every variants/*.so)
PAT="$1"; shift
echo "== default"; timeout -k 10 100 python -u tools/kernel_times.py "$@" | grep "$PAT" || exit 1
for f in variants/*.so; do echo "== $f"; RTX_AMD_LIB=$PWD/$f timeout -k 10 100 python -u tools/kernel_times.py "$@" | grep "$PAT" || exit 1; done
