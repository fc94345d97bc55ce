#!/bin/bash
# Build libsnmf_hip.so (and, with "prof", the -DSNMF_PROF diagnostic library) from a SNAPSHOT of the sources, so that
# the tree can be edited while a 3-minute hipcc run is in flight (hipcc reads the sources once per device/host pass).
# usage: scripts/build_snap.sh [prod] [prof] [var:NAME:"-DFOO=1 -DBAR=2"]...
#   prod -> se_snmf_nat_amd/libsnmf_hip.so, prof -> scripts/prof_build/libsnmf_hip_prof.so,
#   var:NAME:FLAGS -> scripts/prof_build/libsnmf_NAME.so (an experiment build, selected with SNMF_LIB_PATH)
#   logs: /tmp/build_<what>.log; all builds run in parallel
cd "$(dirname "$0")/.."
SNAP=$(mktemp -d /tmp/snmf_snap.XXXXXX)
cp -r se_snmf_nat_amd/csrc include "$SNAP/"
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-value -I$SNAP/include -I$SNAP/csrc"
mkdir -p scripts/prof_build
one() {  # name, extra flags, destination
  ( /opt/rocm/bin/hipcc $FLAGS $2 -o "$SNAP/$1.so" "$SNAP/csrc/snmf_api.hip" > /tmp/build_$1.log 2>&1 && mv "$SNAP/$1.so" "$3" && echo done >> /tmp/build_$1.log || echo FAILED >> /tmp/build_$1.log ) &
}
for what in "$@"; do
  case $what in
    prod) one prod "" se_snmf_nat_amd/libsnmf_hip.so ;;
    prof) one prof "-DSNMF_PROF" scripts/prof_build/libsnmf_hip_prof.so ;;
    var:*) name=$(echo "$what" | cut -d: -f2); fl=$(echo "$what" | cut -d: -f3-); one "$name" "$fl" "scripts/prof_build/libsnmf_$name.so" ;;
  esac
done
wait
for what in "$@"; do n=$what; case $what in var:*) n=$(echo "$what" | cut -d: -f2);; esac; echo "$n: $(tail -1 /tmp/build_$n.log)"; done
