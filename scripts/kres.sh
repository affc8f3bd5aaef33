#!/bin/bash
# Compact per-kernel resource table of one .hip file (cross-compiles for gfx950, no GPU needed):
#   scripts/kres.sh iif_amd/csrc/conv_igemm.hip [name-filter] [extra hipcc flags...]
f=$1; filt=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I"$(dirname $f)" "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | \
  awk '/Function Name:/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       /SGPRs:/{s=$0; sub(/.*SGPRs: /,"",s); sub(/ \[.*/,"",s)}
       / VGPRs:/{v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /AGPRs:/{a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
       /ScratchSize/{c=$0; sub(/.*: /,"",c); sub(/ \[.*/,"",c)}
       /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-90s sgpr %3s vgpr %3s agpr %3s scratch %4s occ %s lds %s\n", n,s,v,a,c,o,l}' | \
  (command -v c++filt >/dev/null && c++filt || cat) | grep -E "$filt"
