#!/bin/bash
# Host-side AddressSanitizer / UBSan pass over libuavx's HOST code (handle, argument checks, launch plumbing): the device code is
# built as usual.  Step 1 (build container):  tools/asan_host.sh build      -> tools/ab/asan/{libuavx.so, abi_client, abi_client_ext}
#                 Step 2 (GPU box):          tools/asan_host.sh run
set -e
cd "$(dirname "$0")/.."
D=tools/ab/asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
if [ "$1" = build ]; then
  mkdir -p $D
  F="-O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=14 -fPIC -Wall -Wno-unused-function"
  hipcc $F -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -shared -shared-libsan \
        -o $D/libuavx.so gym_uav_collision_avoidance_amd/csrc/uavx_multi.hip gym_uav_collision_avoidance_amd/csrc/uavx_uw.hip
  python3 -c "import oracle; print(oracle.build())" > /dev/null
  for c in abi_client abi_client_ext; do
    gcc -O1 -g -std=gnu11 -D__HIP_PLATFORM_AMD__ tests/abi/$c.c -Iinclude -Ioracle -I/opt/rocm/include -L$D -luavx -Loracle/_build -luavx_oracle \
        -L/opt/rocm/lib -lamdhip64 -lm -Wl,--allow-shlib-undefined -Wl,-rpath,'$ORIGIN' -Wl,-rpath,$PWD/oracle/_build -Wl,-rpath,/opt/rocm/lib -o $D/$c
  done
  ls -la $D
else
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  LD_PRELOAD=$RT $D/abi_client 512 4 200
  LD_PRELOAD=$RT $D/abi_client 96 9 120
  LD_PRELOAD=$RT $D/abi_client 96 12 250                      # three wavefronts per workgroup (pick_group_waves)
  UAVX_TILES=2 LD_PRELOAD=$RT $D/abi_client 512 8 100         # two one-wavefront tiles per workgroup
  LD_PRELOAD=$RT $D/abi_client_ext 256 8 16 120
  LD_PRELOAD=$RT $D/abi_client_ext 64 24 0 60
  echo "asan host pass done"
fi
