#!/bin/bash
# build a variant of the library with extra -D flags for ONE source file:  tools/build_variant.sh <name> <file.hip> <flags...>
cd /root/repo/i-dqn_amd || exit 1
name=$1; src=$2; shift 2
vo=_build/${src%.hip}__$name.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-pass-failed -Wno-inline-asm "$@" -c csrc/$src -o $vo || exit 1
objs=$(ls _build/*.o | grep -v "__" | grep -v "_build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libidqn_$name.so $objs $vo && echo built libidqn_$name.so
