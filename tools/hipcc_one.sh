#!/bin/bash
# compile ONE csrc file with the library's flags into /tmp and print the register / LDS / scratch use of its kernels
#   tools/hipcc_one.sh decode_nms [kernel-name-substring]
set -e
src=${PPY_ROOT:-/root/repo}/pytorch-ppyolo_amd/ppyolo_hip/csrc/$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops $PPY_EXTRA_HIPCC_FLAGS -c "$src" -o /tmp/$1.o -save-temps=obj 2>&1 | grep -v "not a recognized feature" || true
asm=/tmp/$1-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v pat="$2" '/^_Z.*:/ {name=$1} /^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize)/ { if (pat == "" || index(name, pat)) printf "%s %s\n", name, $0 }' "$asm" | sed 's/: ; / /' | awk '{k=$1; $1=""; a[k]=a[k] " |" $0} END {for (k in a) print substr(k,1,70), a[k]}'
