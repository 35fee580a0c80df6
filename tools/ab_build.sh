#!/bin/bash
# build the product library into a side directory with extra -D flags: tools/ab_build.sh <name> <flags...>
set -e
name=$1; shift
out=/root/repo/gpurun_ab/$name; mkdir -p $out
cd /root/repo/bbs_sign_amd/csrc
ls *.hip | xargs -P 8 -I{} sh -c "hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden $* -c {} -o $out/{}.o"
hipcc -shared -fPIC --offload-arch=gfx950 $out/*.o -o $out/libbbs_sign_amd.so
rm -f $out/*.o
echo built $out/libbbs_sign_amd.so
