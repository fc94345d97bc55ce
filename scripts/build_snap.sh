#!/bin/bash
# Build libsnmf_hip.so (and, with "prof", the -DSNMF_PROF diagnostic library) from a SNAPSHOT of the sources, so that
# the tree can be edited while a 3-minute hipcc run is in flight (hipcc reads the sources once per device/host pass).
# usage: scripts/build_snap.sh [prod] [prof]   -> logs /tmp/build_prod.log /tmp/build_prof.log
cd "$(dirname "$0")/.."
SNAP=$(mktemp -d /tmp/snmf_snap.XXXXXX)
cp -r se_snmf_nat_amd/csrc include "$SNAP/"
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-value -I$SNAP/include -I$SNAP/csrc"
for what in "$@"; do
  case $what in
    prod) ( /opt/rocm/bin/hipcc $FLAGS -o "$SNAP/prod.so" "$SNAP/csrc/snmf_api.hip" > /tmp/build_prod.log 2>&1 && mv "$SNAP/prod.so" se_snmf_nat_amd/libsnmf_hip.so && echo done >> /tmp/build_prod.log || echo FAILED >> /tmp/build_prod.log ) & ;;
    prof) ( mkdir -p scripts/prof_build; /opt/rocm/bin/hipcc $FLAGS -DSNMF_PROF -o "$SNAP/prof.so" "$SNAP/csrc/snmf_api.hip" > /tmp/build_prof.log 2>&1 && mv "$SNAP/prof.so" scripts/prof_build/libsnmf_hip_prof.so && echo done >> /tmp/build_prof.log || echo FAILED >> /tmp/build_prof.log ) & ;;
  esac
done
wait
