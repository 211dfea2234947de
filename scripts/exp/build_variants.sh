#!/bin/bash
# A/B builds of libelliot_hip.so: one source file recompiled with -D overrides, linked with the other objects of the regular build
# into elliot_amd/csrc/variants/libelliot_hip_<name>.so (git-ignored; travels to the GPU box).  Select one with EL_LIB_PATH.
#   usage: scripts/exp/build_variants.sh <source.hip> <name> "<-D flags>" [<name> "<flags>" ...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/elliot_amd/csrc
SRC=$1; shift
mkdir -p $C/variants
python -c "import sys; sys.path.insert(0, '$R'); from elliot_amd import build; build.build()" > /dev/null
BASE=$(basename $SRC .hip)
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result $FLAGS -c $C/$BASE.hip -o $C/variants/${BASE}_$NAME.o
  OBJS=$(ls $C/*.o | grep -v "/$BASE.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/libelliot_hip_$NAME.so $OBJS $C/variants/${BASE}_$NAME.o
  echo "built $C/variants/libelliot_hip_$NAME.so  ($FLAGS)"
done
