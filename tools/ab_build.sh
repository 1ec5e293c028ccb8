#!/bin/bash
# Builds the product library of an EARLIER commit next to the current one, for same-box kernel A/Bs (runs here: needs git):
#   tools/ab_build.sh <git-rev> [name]      -> tools/bin/libois_<name>.so (git-ignored; travels to the GPU box)
#   gpurun -- 'for v in <name> current; do OIS_LIB_OVERRIDE=$PWD/tools/bin/libois_$v.so python tools/inloop_probe.py; done'
# `current` = a copy of options_in_simulation_amd/lib/libois_hip.so. Entry points the old library lacks become stubs (_native.py).
set -e -o pipefail
REV=$1; NAME=${2:-$1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/options_in_simulation_amd/csrc $T/include $ROOT/tools/bin
for f in ois_kernels.hip ois_gae.hip ois_device.h ois_physics.h ois_physics_pk.h; do git -C $ROOT show $REV:options_in_simulation_amd/csrc/$f > $T/options_in_simulation_amd/csrc/$f; done
git -C $ROOT show $REV:include/ois.h > $T/include/ois.h
FLAGS=$(git -C $ROOT show $REV:options_in_simulation_amd/csrc/Makefile | python3 -c "
import re, sys
m = re.search(r'^FLAGS = ((?:.*\\\\\\n)*.*)$', sys.stdin.read(), re.M)
print(m.group(1).replace('\\\\\\n', ' ').replace('\$(ARCH)', 'gfx950'))")
echo "flags: $FLAGS"
(cd $T/options_in_simulation_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -o $ROOT/tools/bin/libois_$NAME.so ois_kernels.hip ois_gae.hip)
cp $ROOT/options_in_simulation_amd/lib/libois_hip.so $ROOT/tools/bin/libois_current.so
rm -rf $T
ls -la $ROOT/tools/bin/libois_$NAME.so $ROOT/tools/bin/libois_current.so
